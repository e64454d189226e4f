"""Diagnostic (GPU box): per-kernel GPU time of small library kernels, each repeated alone and in rotation (queue pre-filled)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, ctypes as C
from mmda_amd import _lib
lib = _lib.load()
d = torch.device("cuda:0")
B, ncls, hs = 32, 6, 128
s_ = torch.rand(B, ncls, device=d) * 0.8 + 0.1; y = (torch.rand(B, ncls, device=d) > 0.5).float(); tcp = torch.rand(B, 6, device=d)
loss = torch.zeros(8, device=d); ds = torch.zeros(B, ncls, device=d); dt = torch.zeros(B, 6, device=d)
rec = torch.randn(3, B, hs, device=d); org = torch.randn(3, B, hs, device=d); drec = torch.zeros_like(rec); dorg = torch.zeros_like(org)
x6 = torch.rand(6, B, hs, device=d); dx6 = torch.zeros_like(x6)
a = torch.rand(12288, device=d); b = torch.rand(12288, device=d); c = torch.rand(12288, device=d)
st = torch.cuda.current_stream().cuda_stream
ops = {
    "cls": lambda: lib.mmda_loss_cls(s_.data_ptr(), y.data_ptr(), B, ncls, 1.0, loss.data_ptr(), ds.data_ptr(), st),
    "conf": lambda: lib.mmda_loss_conf(s_.data_ptr(), tcp.data_ptr(), y.data_ptr(), B, ncls, 0.3, loss.data_ptr() + 16, ds.data_ptr(), dt.data_ptr(), st),
    "recon": lambda: lib.mmda_loss_recon(rec.data_ptr(), org.data_ptr(), B * hs, B, hs, 0.7, loss.data_ptr() + 12, drec.data_ptr(), dorg.data_ptr(), st),
    "cmd": lambda: lib.mmda_loss_cmd(x6.data_ptr() + 3 * B * hs * 4, B * hs, B, hs, 0.7, loss.data_ptr() + 8, dx6.data_ptr() + 3 * B * hs * 4, st),
    "add": lambda: lib.mmda_add(a.data_ptr(), b.data_ptr(), c.data_ptr(), 12288, st),
    "sigmoid_bwd": lambda: lib.mmda_sigmoid_bwd_inplace(c.data_ptr(), a.data_ptr(), 12288, st),
}
if os.environ.get("SIDE"):
    side = [torch.cuda.Stream() for _ in range(int(os.environ["SIDE"]))]
    for sd in side:
        with torch.cuda.stream(sd):
            lib.mmda_add(a.data_ptr(), b.data_ptr(), c.data_ptr(), 12288, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("created and used", len(side), "extra stream(s)")
def timed(label, fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(8_000_000); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for k, f in ops.items():
    print(f"{k:12s} alone: {timed(k, f, 200):6.2f} us")
rot = [ops[k] for k in ("cls", "recon", "conf", "add", "sigmoid_bwd")]
def rotation():
    for f in rot: f()
print(f"rotation of 5 (cls, recon, conf, add, sigmoid_bwd): {timed('rot', rotation, 60) / 5:6.2f} us per kernel")

big = torch.zeros(64 << 20, device=d)           # 256 MB: a pass over it leaves every L2 full of fresh lines
def big_only():
    big.add_(1.0)
def big_then_tiny():
    big.add_(1.0)
    for _ in range(20):
        ops["add"](); ops["sigmoid_bwd"]()
tb = timed("big", big_only, 20)
tt = timed("big+40tiny", big_then_tiny, 20)
print(f"256 MB pass alone {tb:.1f} us; followed by 40 tiny dependent kernels {tt:.1f} us -> {(tt - tb) / 40:.2f} us per tiny kernel")
med = torch.zeros(1 << 20, device=d)            # 4 MB
def med_then_tiny():
    med.add_(1.0)
    for _ in range(4):
        ops["add"](); ops["sigmoid_bwd"]()
def med_only():
    med.add_(1.0)
tb = timed("med", med_only, 50); tt = timed("med+8tiny", med_then_tiny, 50)
print(f"4 MB pass alone {tb:.1f} us; followed by 8 tiny dependent kernels {tt:.1f} us -> {(tt - tb) / 8:.2f} us per tiny kernel")
