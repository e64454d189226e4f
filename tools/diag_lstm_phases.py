"""Diagnostic (GPU box): where a step of the resident-weights forward recurrence spends its cycles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops, _lib
lib = _lib.load()
d = torch.device("cuda:0")
T, B, H = 50, 32, 300
torch.manual_seed(0)
pre = torch.randn(T, B, 2, 4 * H, device=d)
wf = (torch.rand(4 * H, H, device=d) - 0.5) * 0.1; wr = (torch.rand(4 * H, H, device=d) - 0.5) * 0.1
lengths = torch.full((B,), T)
dbg = torch.zeros(128 * 8, dtype=torch.int64, device=d)
wave_names = ["poll", "issue fragment loads + flush previous stash", "fragment wait + MFMA", "cell update + LDS tile", "publish + drain", "-", "-", "-"]
names = ["cell update+stash stores+lds write", "barrier1", "publish+drain", "barrier2+flag", "poll+barrier3", "gather issue+side traffic", "gather wait+ldswrite", "h fragments from LDS + MFMA"]
for it in range(3):
    lib.mmda_debug_set_lstm_stamps(dbg.data_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    fw = ops.lstm_bidir_fwd(pre, wf, wr, lengths, mode="bf16", resident=True, gate_minor=True)   # the stamped kernel instances are gate-minor
    e1.record(); torch.cuda.synchronize()
    lib.mmda_debug_set_lstm_stamps(None)
    print(f"iter {it}: whole op (pack + kernel) {e0.elapsed_time(e1)*1e3:.0f} us; aborted={ops.lstm_aborted(fw)}")
raw = dbg.view(128, 8)[:20].cpu()
v = raw.double()
tot = v.sum(1)
print("per-WG total cycles/step:", (tot / T).tolist())
for i, n in enumerate(wave_names if os.environ.get("MMDA_LSTM_WAVE_FWD") else names):
    print(f"{n:28s} mean {float(v[:, i].mean())/T:8.0f} cyc/step  min {float(v[:, i].min())/T:8.0f} max {float(v[:, i].max())/T:8.0f}")

# ---- backward (wave-autonomous form)
dbg.zero_()
d_utt = torch.zeros(B, 4 * H, device=d); d_out = torch.randn(T, B, 2 * H, device=d)
lib.mmda_debug_set_lstm_stamps(dbg.data_ptr())
ops.lstm_bidir_bwd(fw, d_utt, d_out, mode="bf16")
torch.cuda.synchronize()
lib.mmda_debug_set_lstm_stamps(None)
vb = dbg.view(128, 8)[:76].cpu().double() if dbg.numel() >= 76 * 8 else dbg.view(128, 8).cpu().double()
print("backward per-wave total cycles/step:", [round(float(x) / T) for x in vb.sum(1)[:8].tolist()])
for i, n in enumerate(["poll", "issue gather loads + flush previous dG", "gather wait + fp32 sum", "cell backward + LDS tile", "A fragments + 38 MFMA + 19 publish stores", "drain", "-", "-"]):
    nz = vb[:, i][vb[:, i] > 0]
    if len(nz): print(f"{n:44s} mean {float(nz.mean())/T:8.0f} cyc/step  min {float(nz.min())/T:8.0f} max {float(nz.max())/T:8.0f}")
