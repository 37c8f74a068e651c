"""Isolated timings of the streaming kernels around the attention core of the training path at the config-5 size
(163 346 segments x 8 KB): pg_attn_fold_query (writes U), pg_attn_unfold_value (reads S), pg_attn_fold_wgrad (reads T)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phoregen_amd import hip
from phoregen_amd import training as tr

dev = 'cuda'
E = int(sys.argv[1]) if len(sys.argv) > 1 else 163346
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(E, 128, device=dev, generator=g)
W = torch.randn(64, 64, 4, device=dev, generator=g)
b2 = torch.randn(128, device=dev, generator=g)
U = torch.empty(E, 2048, device=dev)
S = torch.randn(E, 2048, device=dev, generator=g)
sw = torch.rand(E, 16, device=dev, generator=g)
out = torch.empty(E, 128, device=dev)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


gb = E * 2048 * 4 / 1e9
for name, fn in (('fold_query  (write U)', lambda: tr._fold(q, W, None, E, U)),
                 ('unfold_value (read S)', lambda: tr._unfold(S, sw, W, b2, None, E, out)),
                 ('fold_wgrad   (read T)', lambda: tr._fold_wgrad(q, S, None, E, W))):
    ms = timeit(fn)
    print('%s  %.3f ms  %.2f TB/s of the %.2f GB array' % (name, ms, gb / ms, gb))
