"""Diagnostic (GPU box): GPU time of forward / losses / backward / Adam of one step, each bracketed by events with the queue
pre-filled (torch.cuda._sleep), i.e. independent of host launch speed and of any profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import make_config, _lib
from mmda_amd.solver import Solver
from mmda_amd.data import synth_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = make_config(vocab_size=20000, precision="bf16", device=str(dev), batch_size=32, seq_len=50, use_confidNet=False,
                  pretrained_emb=torch.randn(20000, 300))
solver = Solver(cfg, cfg, cfg, None, None, None, is_train=True).build()
m = solver.model
m.train()
t, v, a, y, emo, lengths, *_ = synth_batch(cfg, 32, 50, seed=0, ragged=False, device=dev)
for _ in range(10):
    m.train_step(t, v, a, lengths, emo, lr=1e-4, clip=1.0)
torch.cuda.synchronize()
tt, vv, aa, ld = m._prepare(t, v, a, lengths)
emo = emo.float().contiguous()
lib, h = m._lib, m._h
s = _lib.stream_ptr()
import itertools
for overlap in (1,):
  _lib.check(lib.mmda_misa_set_overlap(h, overlap), "overlap")
  acc = [0.0] * 5
  N = 20
  for it in range(N):
      ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
      torch.cuda._sleep(8_000_000)
      ev[0].record()
      _lib.check(lib.mmda_misa_zero_grad(h, s), "zg")
      ev[1].record()
      _lib.check(lib.mmda_misa_forward(h, tt.data_ptr(), vv.data_ptr(), aa.data_ptr(), ld.data_ptr(), 1, 1234 + it, s), "fwd")
      ev[2].record()
      _lib.check(lib.mmda_misa_losses(h, emo.data_ptr(), 1, s), "loss")
      ev[3].record()
      _lib.check(lib.mmda_misa_backward(h, tt.data_ptr(), vv.data_ptr(), aa.data_ptr(), ld.data_ptr(), s), "bwd")
      ev[4].record()
      _lib.check(lib.mmda_misa_adam_step(h, 1e-4, 1.0, 1.0, 11 + it, s), "adam")
      ev[5].record()
      torch.cuda.synchronize()
      for k in range(5):
          acc[k] += ev[k].elapsed_time(ev[k + 1])
  names = ["zero_grad", "forward", "losses", "backward", "adam"]
  print(f"overlap={overlap}", "  ".join(f"{n}={1e3 * x / N:.1f}us" for n, x in zip(names, acc)), f" total={1e3 * sum(acc) / N:.1f}us")

# dependent chain of three different trivial library kernels on the same buffers
x = torch.rand(12288, device=dev); yb = torch.rand(12288, device=dev); z = torch.rand(12288, device=dev)
def chain(n):
    for _ in range(n):
        lib.mmda_add(x.data_ptr(), yb.data_ptr(), z.data_ptr(), 12288, s)
        lib.mmda_sigmoid_bwd_inplace(z.data_ptr(), x.data_ptr(), 12288, s)
        lib.mmda_act_dropout_fwd(z.data_ptr(), yb.data_ptr(), 12288, 1, 0.0, 0, 0, s)
chain(5); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(8_000_000); e0.record(); chain(100); e1.record(); torch.cuda.synchronize()
print(f"dependent chain of 3 different trivial kernels: {e0.elapsed_time(e1) * 1e3 / 300:.2f} us per kernel")
# the losses pass alone, repeated
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(8_000_000); e0.record()
for _ in range(20):
    lib.mmda_misa_losses(h, emo.data_ptr(), 1, s)
e1.record(); torch.cuda.synchronize()
print(f"losses pass repeated: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per pass")
