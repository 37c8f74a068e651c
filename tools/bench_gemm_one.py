"""One tall GEMM of the bond-row shape through pg_gemm, timed with HIP events (PG_GEMM_ABLATE = timing-only ablations:
1 no global loads, 2 one MFMA k-step per chunk, 4 store one row per tile)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip
lib = hip.lib(); dev = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 203720
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 128
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev); Y = torch.empty(M, N, device=dev)
p = hip.PgGemm(); p.X, p.ldx, p.K1 = X.data_ptr(), K, K; p.W, p.ldw, p.bias = W.data_ptr(), K, b.data_ptr()
p.out_scale, p.act = 1.0, 0; p.Y, p.ldy, p.M, p.N = Y.data_ptr(), N, M, N
s = hip.stream_ptr()
for _ in range(6): lib.pg_gemm(C.byref(p), s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): lib.pg_gemm(C.byref(p), s)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print('ablate=%s M=%d N=%d K=%d: %.1f us  %.1f TF/s' % (os.environ.get('PG_GEMM_ABLATE', '0'), M, N, K, us, 2.0 * M * N * K / us / 1e6))
