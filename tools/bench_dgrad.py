"""Input-gradient products of the training step, gX[M,128] = gY[M,K] W[K,128] with K = 256 ... 1920: the tiled pg_gemm (on W^T, as
`training._dgrad` calls it with `dgrad_mm=False`) against the library GEMM through torch.mm."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip
lib = hip.lib(); dev = 'cuda'; s = hip.stream_ptr()


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, K in ((34000, 1920), (34000, 1280), (34000, 640), (163346, 256), (163346, 384), (6478, 1280)):
    gY = torch.randn(M, K, device=dev); W = torch.randn(K, 128, device=dev) * 0.1; Wt = W.t().contiguous(); gX = torch.empty(M, 128, device=dev)
    p = hip.PgGemm(); p.X, p.ldx, p.K1 = gY.data_ptr(), K, K; p.W, p.ldw = Wt.data_ptr(), K
    p.out_scale, p.act = 1.0, 0; p.Y, p.ldy, p.M, p.N = gX.data_ptr(), 128, M, 128
    a = timed(lambda: lib.pg_gemm(C.byref(p), s))
    ref = gX.clone()
    b = timed(lambda: torch.mm(gY, W, out=gX))
    err = float((ref - gX).abs().max() / gX.abs().max())
    print(f'M={M:7d} K={K:5d}: pg_gemm {a:7.1f} us ({2.0 * M * 128 * K / a / 1e6:5.1f} TF/s)   torch.mm {b:7.1f} us ({2.0 * M * 128 * K / b / 1e6:5.1f} TF/s)   max rel diff {err:.1e}')
