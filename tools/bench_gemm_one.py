import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip
lib = hip.lib(); dev = 'cuda'
M, N, K = 203720, 128, 128
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev); Y = torch.empty(M, N, device=dev)
p = hip.PgGemm(); p.X, p.ldx, p.K1 = X.data_ptr(), K, K; p.W, p.ldw, p.bias = W.data_ptr(), K, b.data_ptr()
p.out_scale, p.act = 1.0, 0; p.Y, p.ldy, p.M, p.N = Y.data_ptr(), N, M, N
s = hip.stream_ptr()
for _ in range(6): lib.pg_gemm(C.byref(p), s)
torch.cuda.synchronize()
