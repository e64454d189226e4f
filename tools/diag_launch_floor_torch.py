"""Diagnostic (GPU box): GPU-side cost of back-to-back tiny library kernels inside the torch process, on the default (null)
stream and on a created stream, with the queue pre-filled behind torch.cuda._sleep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops, _lib
d = torch.device("cuda:0")
a = torch.zeros(12288, device=d); b = torch.ones(12288, device=d); y = torch.zeros(12288, device=d)
lib = _lib.load()

def run(label, n=300):
    s = torch.cuda.current_stream()
    sp = s.cuda_stream
    for _ in range(10):
        lib.mmda_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), 12288, sp)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    e0.record()
    for _ in range(n):
        lib.mmda_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), 12288, sp)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per mmda_add (stream handle {sp})")

run("default stream")
with torch.cuda.stream(torch.cuda.Stream()):
    run("torch side stream")
run("default stream again")
