"""GEMM micro-benchmark on the shapes the denoiser uses (GPU box)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip
lib = hip.lib()
lib.pg_debug_gemm_streaming(int(os.environ.get('PG_GEMM_SP', '1')))
dev = 'cuda'
def run(M, N, K1, K2=0, ln=False, gather=0, act=0, reps=10, ldx=None):
    ldx = ldx or K1
    X = torch.randn(M, ldx, device=dev); W = torch.randn(N, K1 + K2, device=dev) * 0.1; b = torch.randn(N, device=dev)
    X2 = torch.randn(M, max(K2, 1), device=dev); Y = torch.empty(M, N, device=dev)
    A = torch.randn(20000, 1920, device=dev); idx = torch.randint(0, 20000, (M,), device=dev, dtype=torch.int32)
    gam = torch.randn(128, device=dev); bet = torch.randn(128, device=dev)
    p = hip.PgGemm()
    p.X, p.ldx, p.K1 = X.data_ptr(), ldx, K1
    p.X2, p.ldx2, p.K2 = (X2.data_ptr(), X2.stride(0), K2) if K2 else (None, 0, 0)
    p.W, p.ldw, p.bias = W.data_ptr(), K1 + K2, b.data_ptr()
    if ln: p.ln_gamma, p.ln_beta = gam.data_ptr(), bet.data_ptr()
    if gather >= 1: p.add1, p.ld_add1, p.idx1, p.add_rows = A.data_ptr(), 1920, idx.data_ptr(), 20000
    if gather >= 2: p.add2, p.ld_add2, p.idx2 = A[:, 256:].data_ptr(), 1920, idx.data_ptr()
    p.out_scale, p.act = 1.0, act
    p.Y, p.ldy, p.M, p.N = Y.data_ptr(), N, M, N
    s = hip.stream_ptr()
    for _ in range(2): lib.pg_gemm(C.byref(p), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.pg_gemm(C.byref(p), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'M={M} N={N} K={K1}+{K2} ln={ln} gather={gather} act={act} ldx={ldx}: {ms*1e3:.1f} us  {2*M*N*(K1+K2)/ms/1e9:.1f} TF/s')
E, n = 203720, 18401
run(E, 128, 128); run(E, 128, 128, gather=1); run(E, 128, 128, ln=True); run(E, 128, 128, act=1)
run(E, 256, 128); run(E, 256, 128, gather=1); run(E, 256, 128, K2=20, gather=2); run(E, 256, 128, K2=20, gather=1)
run(n, 1920, 128); run(n, 1280, 128); run(n, 128, 128, ln=True, ldx=1920); run(n, 128, 256)
run(E, 256, 20); run(E, 256, 20, gather=1)
