"""Diagnostics: run the benchmark's training loop eagerly, stop at the first step whose gradients are not finite
and print the range words, the maxima of every gradient tensor and which tensors hold non-finite values."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from odin_ai_amd.engine import VAEEngine
from oracle import vae_oracle as vo

dev = torch.device('cuda:0')
enc, dec, in_shape, zdim = vo.dsprites_spec(1)
B = 256
eng = VAEEngine(enc, dec, in_shape, zdim, B, dev)
bench.init_params_(eng, 1)
x = bench.synthetic_batch('dsprites_betavae_b256', B, in_shape, dev, seed=100)
use_graph = len(sys.argv) > 1 and sys.argv[1] == 'graph'
for t in range(3000):
  eng.train_step(x, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=use_graph)
  if t % 50 == 49 or t < 3:
    f = eng.flag.item()
    print('step', t + 1, 'loss', eng.out4[0].item(), 'flag', f, flush=True)
    if f:
      break
torch.cuda.synchronize()
words = eng.range_words.cpu().numpy().view(np.float32).reshape(-1, 2048).max(1)
print('range words', words)
for name, prog in (('enc', eng.enc), ('dec', eng.dec)):
  for i, g in enumerate(prog.gouts):
    print(name, i, 'gout absmax', float(g.abs().max()), 'finite', bool(torch.isfinite(g).all()),
          'out absmax', float(prog.outs[i].abs().max()))
for k, v in eng.grad_views().items():
  if not torch.isfinite(v).all():
    print('non-finite grad', k)
