"""Diagnostic (GPU box): relative L2 error of every gradient of the bf16 path against (a) the exact fp32 oracle and (b) the
bf16-emulating oracle (oracle/bf16_emul.py), per golden fixture.  python tools/diag_bf16_emul.py [fixture ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import misa_oracle as orc
from oracle import bf16_emul as emu
from golden_util import batch_of, load_case
from mmda_amd import make_config, MISA

names = sys.argv[1:] or ["real_b8_t12_ragged", "real_b32_t50_full", "real_b16_t20_adv_confid", "real_gru_b8_t12_ragged", "real_gru_b16_t20_adv"]
for name in names:
    z, meta, cfg = load_case(name)
    P = orc.synth_params(cfg, meta["seed"])
    batch = batch_of(z)
    m = MISA(make_config(precision="bf16", device="cuda:0", **vars(cfg))); m.load_state_dict(P); m.to("cuda:0")
    b = {k: (v.to("cuda:0") if k != "l" else v) for k, v in batch.items()}
    _, _, G = orc.loss_and_grads(P, cfg, batch)
    for resident in (True, False):
        m.set_recurrence(resident)
        m.train_step(b["t"], b["v"], b["a"], b["l"], b["emo"], lr=1e-4, clip=1.0, do_adam=False, training=False)
        m._assign_grad_views()
        oq, Lq, Gq = emu.loss_and_grads(P, cfg, batch, rounding=True, tile_partials=resident)
        pub = m._public()
        print(f"== {name} resident={resident} aborted={m.cluster_aborted()} scores vs emul {float((pub['scores'].cpu() - oq.scores.detach()).abs().max()):.2e}")
        Lg = m.read_losses()
        print("   losses rel vs emul:", {k: f"{abs(Lg[k] - float(getattr(Lq, k))) / abs(float(getattr(Lq, k))):.1e}" for k in ("cls", "diff", "sim", "recon", "total")})
        rows = []
        for k, p in m.named_parameters():
            if G[k] is None or k.endswith("in_proj_bias"):
                continue
            g = p.grad.cpu().double()
            e32 = float((g - G[k].double()).norm() / G[k].double().norm())
            eq = float((g - Gq[k].double()).norm() / Gq[k].double().norm())
            rows.append((eq, e32, k))
        rows.sort(reverse=True)
        for eq, e32, k in rows[:6]:
            print(f"   {k:45s} vs emul {eq:.2e}   vs fp32 {e32:.2e}")
        print(f"   worst vs emul {rows[0][0]:.2e}; worst vs fp32 {max(r[1] for r in rows):.2e}")
