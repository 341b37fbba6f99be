"""Diagnostics: run a benchmark workload's training loop (eager, or graph replay with `graph`), stop at the first
step whose gradients are not finite and print the range words, the maxima of every gradient tensor and which
tensors hold non-finite values.    python tools/nan_hunt.py [workload] [graph] [steps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks

dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'dsprites_betavae_b256'
use_graph = len(sys.argv) > 2 and sys.argv[2] == 'graph'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
ds, kw, B, beta, kind = bench.WORKLOADS[wl]
nets = get_networks(ds, **kw)
enc, dec = nets['encoder'].layers, nets['decoder'].layers
in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation=nets['observation'].posterior,
                tc=kind if kind == 'betatc' else None, seed=1)
bench.init_params_(eng, 1)
x = bench.synthetic_batch(wl, B, in_shape, dev, seed=100)
if use_graph:
  xb = eng.input_buffer()
  xb.copy_(x)
  x = xb
every = 1 if not use_graph else 10
for t in range(steps):
  eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=use_graph)
  if t % every == every - 1 or t < 3:
    f = eng.flag.item()
    if f or t % 100 == 99 or t < 3:
      print('step', t + 1, 'loss', eng.out4[0].item(), 'flag', f, flush=True)
    if f:
      break
torch.cuda.synchronize()
words = eng.range_words.cpu().numpy().view(np.float32).reshape(-1, 2048).max(1)
print('range words (cleared by the step itself)', words)
for name, prog in (('enc', eng.enc), ('dec', eng.dec)):
  for i, g in enumerate(prog.gouts):
    print(name, i, prog.recs[i].kind, tuple(g.shape), 'gout absmax', float(g.abs().max()), 'finite',
          bool(torch.isfinite(g).all()), 'out absmax', float(prog.outs[i].abs().max()), 'word',
          prog.dy_word[i] is not None)
for k, v in eng.grad_views().items():
  if not torch.isfinite(v).all():
    print('non-finite grad', k, int((~torch.isfinite(v)).sum()), 'of', v.numel())
