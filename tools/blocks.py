#!/usr/bin/env python3
"""Diagnostic: per-block step time of the default workload inside ONE process (is a slow run a
per-process state or a transient?)."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks
ds, kw, B, beta, kind = bench.WORKLOADS['dsprites_betavae_b256']
nets = get_networks(ds, **kw)
dev = torch.device('cuda:0')
eng = VAEEngine(nets['encoder'].layers, nets['decoder'].layers, nets['encoder'].input_shape,
                nets['latents'].event_shape[0], B, dev, observation=nets['observation'].posterior, tc=kind)
bench.init_params_(eng, seed=1)
x = bench.synthetic_batch('dsprites_betavae_b256', B, nets['encoder'].input_shape, dev, seed=100)
for _ in range(20):
  eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=True)
torch.cuda.synchronize()
out = []
for blk in range(12):
  t0 = time.perf_counter()
  for _ in range(100):
    eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=True)
  torch.cuda.synchronize()
  out.append((time.perf_counter() - t0) * 10)
print(' '.join(f'{v:.4f}' for v in out))
