"""Training-path report on the GPU box: HIP forward + adjoints (phoregen_amd/training.py) against autograd through the
oracle on the CPU, per output and per parameter.  LAYERS=n limits the depth (bisecting), T0/T1 pick the time steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F
from oracle import phoregen_oracle as po
from oracle.make_inputs import synthetic_batch
from phoregen_amd.config import default_model_config
from phoregen_amd.models.diffusion import PhoreDiff
from phoregen_amd.weights import init_deterministic_
from phoregen_amd.plan import BatchPlan
from phoregen_amd import training as tr
from helpers import DIFF_CFG

torch.manual_seed(0)
dev = 'cuda'
L = int(os.environ.get('LAYERS', '6'))

# ---- op-level checks ----
def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.double().abs().max()), 1e-30))

X = torch.randn(777, 148, device=dev, requires_grad=True); W = torch.randn(130, 148, device=dev, requires_grad=True)
b = torch.randn(130, device=dev, requires_grad=True); R = torch.randn(777, 130, device=dev)
y = tr.linear(X, W, b); (y * R).sum().backward()
g1 = [X.grad.clone(), W.grad.clone(), b.grad.clone()]; X.grad = W.grad = b.grad = None
y2 = F.linear(X, W, b); (y2 * R).sum().backward()
print('linear fwd %.1e gX %.1e gW %.1e gb %.1e' % (rel(y, y2), rel(g1[0], X.grad), rel(g1[1], W.grad), rel(g1[2], b.grad)))
X = torch.randn(1001, 128, device=dev, requires_grad=True); ga = torch.randn(128, device=dev, requires_grad=True)
be = torch.randn(128, device=dev, requires_grad=True); R = torch.randn(1001, 128, device=dev)
y = tr.LnReluFn.apply(X, ga, be); (y * R).sum().backward()
g1 = [X.grad.clone(), ga.grad.clone(), be.grad.clone()]; X.grad = ga.grad = be.grad = None
y2 = F.relu(F.layer_norm(X, (128,), ga, be, 1e-5)); (y2 * R).sum().backward()
print('ln_relu fwd %.1e gX %.1e gg %.1e gb %.1e' % (rel(y, y2), rel(g1[0], X.grad), rel(g1[1], ga.grad), rel(g1[2], be.grad)))
from phoregen_amd.packing import lane_fixed_w2
n = 50
q = torch.randn(n, 128, device=dev, requires_grad=True); W2 = torch.randn(128, 128, device=dev, requires_grad=True)
ids = torch.arange(0, n, 2, device=dev, dtype=torch.int32); R = torch.randn(n, 2048, device=dev)
U = tr.FoldFn.apply(q, lane_fixed_w2(W2), ids, ids.numel()); (U * R).sum().backward()
g1 = [q.grad.clone(), W2.grad.clone()]; q.grad = W2.grad = None
Ud = torch.einsum('shd,hdc->sch', q.view(n, 16, 8), W2.view(16, 8, 128))            # [s, c, h]
c = torch.arange(128, device=dev); h = torch.arange(16, device=dev)
idx = (((c >> 4) * 4 + (c & 3)) * 64 + ((c >> 2) & 3) * 16)[:, None] + h[None, :]       # lane-fixed position of (c, h)
U2 = torch.zeros(n, 2048, device=dev).index_put((ids.long()[:, None, None], idx[None]), Ud[ids.long()])
(U2 * R).sum().backward()
print('fold fwd %.1e gq %.1e gW2 %.1e' % (rel(U[ids.long()], U2[ids.long()]), rel(g1[0], q.grad), rel(g1[1], W2.grad)))
S = torch.randn(n, 2048, device=dev, requires_grad=True); sw = torch.rand(n, 16, device=dev, requires_grad=True)
b2 = torch.randn(128, device=dev, requires_grad=True); R = torch.randn(n, 128, device=dev); q.grad = W2.grad = None
o = tr.UnfoldFn.apply(S, sw, lane_fixed_w2(W2), b2, ids, ids.numel()); (o * R).sum().backward()
g1 = [S.grad.clone(), sw.grad.clone(), W2.grad.clone(), b2.grad.clone()]; S.grad = sw.grad = W2.grad = b2.grad = None
Sd = S[:, idx]                                                                        # [s, c, h]
od = torch.einsum('sch,hdc->shd', Sd, W2.view(16, 8, 128)).reshape(n, 128) + b2 * sw.repeat_interleave(8, 1)
mask = torch.zeros(n, 1, device=dev); mask[ids.long()] = 1
(od * mask * R).sum().backward()
print('unfold fwd %.1e gS %.1e gsw %.1e gW2 %.1e gb2 %.1e' % (rel(o, od * mask), rel(g1[0][ids.long()], S.grad[ids.long()]), rel(g1[1], sw.grad),
                                                              rel(g1[2], W2.grad), rel(g1[3], b2.grad)))

# ---- whole forward ----
cfg = default_model_config()
cfg.denoiser.num_layers = L
model = init_deterministic_(PhoreDiff(cfg, 'zinc_300'), 0).to(dev)
inp = synthetic_batch(3, [5, 9], [6, 11], [int(os.environ.get('T0', '950')), int(os.environ.get('T1', '990'))])
sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
orc = po.Oracle(sd_cpu, num_layers=L, diff_cfg=DIFF_CFG)
names = [k for k, p in model.named_parameters()]
for k in names:
    orc.sd[k].requires_grad_(True)
v, x0, bond, (cl, cu) = orc.forward(**inp)
gen = torch.Generator().manual_seed(11)
Rs = [torch.randn(o.shape, generator=gen) for o in (v, x0, bond, cl, cu)]
loss_ref = sum((o * r).sum() for o, r in zip((v, x0, bond, cl, cu), Rs))
loss_ref.backward()

params = {**{k: b_ for k, b_ in model.named_buffers()}, **{k: p for k, p in model.named_parameters()}}
plan = BatchPlan(inp['batch_node'], inp['batch_phore'], inp['edge_index'], inp['batch_edge'], 2, dev)
tf = tr.TrainForward(params, plan, knn_k=32, num_layers=L)
g = {k: (t.to(dev) if torch.is_tensor(t) else t) for k, t in inp.items()}
v2, x02, bond2, (cl2, cu2) = tf.forward(g['h_node_pert'], g['pos_pert'], g['h_edge_pert'], g['time_step'], g['h_phore'],
                                        g['pos_phore'], g['phore_norm'], g['batch_phore'])
outs2 = (v2, x02, bond2, cl2, cu2)
for nm, a, b_ in zip(('v', 'x0', 'bond', 'count_l', 'count_u'), outs2, (v, x0, bond, cl, cu)):
    print('fwd %-8s %.2e' % (nm, rel(a.detach(), b_.detach())))
loss = sum((o * r.to(dev)).sum() for o, r in zip(outs2, Rs))
loss.backward()
torch.cuda.synchronize()
rows = []
for k in names:
    gr, gh = orc.sd[k].grad, dict(model.named_parameters())[k].grad
    if gr is None and gh is None:
        continue
    gr = torch.zeros_like(orc.sd[k]) if gr is None else gr
    gh = torch.zeros_like(gr) if gh is None else gh.cpu()
    nr = float(gr.norm())
    rows.append((float((gh - gr).norm()) / max(nr, 1e-12), nr, k))
rows.sort(reverse=True)
print('worst parameter gradients (rel L2 err, |ref|, name):')
for e, nr, k in rows[:int(os.environ.get('TOP', '40'))]:
    print('  %.2e  %.2e  %s' % (e, nr, k))
print('median rel err %.2e over %d tensors; loss %.6f vs %.6f' % (sorted(r[0] for r in rows)[len(rows) // 2], len(rows), float(loss), float(loss_ref)))
