import torch
dev='cuda'
for M,N,K in [(203720,128,128),(203720,256,128),(18401,1920,128),(203720,128,148)]:
    X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)
    for _ in range(3): Y=X@W.t()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): Y=X@W.t()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f'torch.mm fp32 M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TF/s')
