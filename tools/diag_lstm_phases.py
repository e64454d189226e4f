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
dbg = torch.zeros(64 * 8, dtype=torch.int64, device=d)
wave_names = ["poll", "issue fragment loads + flush previous stash", "fragment wait + MFMA", "cell update + LDS tile", "publish + drain", "-", "-", "-"]
names = ["cell update+stash stores+lds write", "barrier1", "publish+drain", "barrier2+flag", "poll+barrier3", "gather issue+side traffic", "gather wait+ldswrite", "h fragments from LDS + MFMA"]
for it in range(3):
    lib.mmda_debug_set_lstm_stamps(dbg.data_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    fw = ops.lstm_bidir_fwd(pre, wf, wr, lengths, mode="bf16", resident=True)
    e1.record(); torch.cuda.synchronize()
    lib.mmda_debug_set_lstm_stamps(None)
    print(f"iter {it}: whole op (pack + kernel) {e0.elapsed_time(e1)*1e3:.0f} us; aborted={ops.lstm_aborted(fw)}")
raw = dbg.view(64, 8)[:20].cpu()
v = raw.double()
tot = v.sum(1)
print("per-WG total cycles/step:", (tot / T).tolist())
for i, n in enumerate(wave_names if os.environ.get("MMDA_LSTM_WAVE_FWD") else names):
    print(f"{n:28s} mean {float(v[:, i].mean())/T:8.0f} cyc/step  min {float(v[:, i].min())/T:8.0f} max {float(v[:, i].max())/T:8.0f}")
