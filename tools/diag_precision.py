"""Diagnostic (GPU box): per-tensor error table of the HIP path vs the CPU oracle for a golden case and a precision."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import misa_oracle as orc
from golden_util import batch_of, load_case, SIDE
from mmda_amd import make_config, MISA

name = sys.argv[1] if len(sys.argv) > 1 else "real_b32_t50_full"
precision = sys.argv[2] if len(sys.argv) > 2 else "bf16"
z, meta, cfg = load_case(name)
P = orc.synth_params(cfg, meta["seed"])
batch = batch_of(z)
o, L, G = orc.loss_and_grads(P, cfg, batch)
m = MISA(make_config(precision=precision, device="cuda:0", **vars(cfg)))
m.load_state_dict(P); m.to("cuda:0")
b = {k: (v.to("cuda:0") if k != "l" else v) for k, v in batch.items()}
m.train_step(b["t"], b["v"], b["a"], b["l"], b["emo"], lr=1e-4, clip=1.0, do_adam=False, training=False)
pub = m._public()
print(f"== {name} {precision}")
for k in ["scores", "tcp"] + SIDE:
    ref = getattr(o, k).detach(); got = pub[k].cpu()
    print(f"out {k:16s} maxrel {float((got-ref).abs().max()/ref.abs().max()):.2e}")
for k, v in m.read_losses().items():
    print(f"loss {k:6s} {v:.6f} ref {float(getattr(L,k)):.6f}")
m._assign_grad_views()
rows = []
for k, p in m.named_parameters():
    if G[k] is None: continue
    g = p.grad.cpu().double(); r = G[k].double()
    mx = float((g-r).abs().max()/r.abs().max().clamp_min(1e-30))
    l2 = float((g-r).norm()/r.norm().clamp_min(1e-30))
    cos = float((g.flatten()@r.flatten())/(g.norm()*r.norm()).clamp_min(1e-30))
    rows.append((mx, l2, cos, k, float(r.abs().max())))
for mx, l2, cos, k, rm in sorted(rows, reverse=True)[:100]:
    print(f"grad maxrel {mx:.2e} l2rel {l2:.2e} cos {cos:.6f} |ref|max {rm:.2e} {k}")
