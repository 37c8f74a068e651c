"""Does the kind of kernel that runs BEFORE the triplet kernel change the triplet kernel's own duration (clock / power state)?
Per repetition: N GEMM launches of one kind (a bond-row product, 203 720 x 256 x 128), then the triplet kernel between two events.
Prints the mean triplet time after: nothing, fp32-MFMA GEMMs, split-bf16 GEMMs, an HBM copy of the same bytes.
NEEDS a library built WITH the experiment kernel gemm_emu_bf16x6.hip (commit 3a8c0f7 of this repository had it as the default; there
`pg_debug_gemm_streaming(2)` selected the fp32-MFMA streaming kernel): with the product library both GEMM kinds are the fp32 kernel.
Recorded output: profiles/r03_micro_clock_after_bf16_mfma.txt."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import ligphore_workload
from phoregen_amd import hip
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_

reps, n_gemm = 15, int(sys.argv[1]) if len(sys.argv) > 1 else 5
model = init_deterministic_(PhoreDiff(default_model_config(), 'zinc_300'), 0).eval().to('cuda')
work = ligphore_workload(128)
st = model.begin_sampling(work['h_phore'], work['pos_phore'], work['phore_norm'], work['batch_phore'], work['num_atoms'],
                          torch.zeros(128, 3), rng='device', seed=0, return_traj=False, num_steps=2)
model.reverse_step(st, 0, 999)
eng = st.eng
fn, args, _ = eng.prog_fwd[eng.tri_calls[0]]
lib, s = hip.lib(), hip.stream_ptr()
M = 203720
X, W, Y = torch.randn(M, 128, device='cuda'), torch.randn(256, 128, device='cuda') * 0.1, torch.empty(M, 256, device='cuda')
p = hip.PgGemm()
p.X, p.ldx, p.K1, p.W, p.ldw = X.data_ptr(), 128, 128, W.data_ptr(), 128
p.out_scale, p.act, p.Y, p.ldy, p.M, p.N = 1.0, 0, Y.data_ptr(), 256, M, 256
big = torch.empty(M * 96, device='cuda')


def run(kind):
    tri, pre = [], []
    for r in range(reps + 3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(n_gemm):
            if kind == 'copy':
                big[:M * 48].copy_(big[M * 48:])
            elif kind != 'none':
                lib.pg_debug_gemm_streaming(1 if kind == 'bf16x6' else 2)
                lib.pg_gemm(C.byref(p), s)
        e1.record()
        fn(*args, s)
        e2.record()
        torch.cuda.synchronize()
        if r >= 3:
            pre.append(e0.elapsed_time(e1)); tri.append(e1.elapsed_time(e2))
    lib.pg_debug_gemm_streaming(1)
    print(f'{kind:8s}: {n_gemm} launches before = {sum(pre)/len(pre):7.3f} ms, triplet kernel after them = {sum(tri)/len(tri):.3f} ms')


for kind in ('none', 'fp32', 'bf16x6', 'copy', 'none', 'bf16x6', 'fp32'):
    run(kind)
