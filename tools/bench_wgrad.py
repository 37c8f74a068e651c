import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phoregen_amd import training as tr
dev='cuda'; g=torch.Generator(device=dev).manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for M, N, K in ((163346, 256, 148), (163346, 256, 128), (163346, 128, 128), (163346, 256, 20), (34000, 1920, 128), (34000, 128, 128), (34000, 1280, 128), (6478, 128, 128)):
    gY = torch.randn(M, N, device=dev, generator=g); X = torch.randn(M, K, device=dev, generator=g)
    ms = timeit(lambda: tr._wgrad(gY, X, N, K, True))
    ms2 = timeit(lambda: torch.mm(gY.t(), X))
    gW, gb = tr._wgrad(gY, X, N, K, True)
    ref = gY.double().t() @ X.double()
    err = float((gW.double() - ref).abs().max() / ref.abs().max()); errb = float((gb.double() - gY.double().sum(0)).abs().max() / gY.double().sum(0).abs().max())
    gf = 2.0 * M * N * K / 1e9
    print('M=%d N=%d K=%d  pg_gemm_wgrad %.3f ms (%.1f TF/s, %.2f TB/s of operands)   rocBLAS %.3f ms   max err vs fp64 %.1e (bias %.1e)' % (M, N, K, ms, gf / ms, (M * (N + K) * 4 / 1e9) / ms, ms2, err, errb))
