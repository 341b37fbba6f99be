#!/usr/bin/env python3
"""Weight-gradient slab bytes of one training step and the duration of the slab reduction alone.
  python tools/slabstat.py [workload]     (default dsprites_betavae_b256)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
import bench
from odin_ai_amd.engine import VAEEngine, ReduceJob
from odin_ai_amd.networks import get_networks

wl = sys.argv[1] if len(sys.argv) > 1 else 'dsprites_betavae_b256'
ds, kw, B, beta, kind = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
nets = get_networks(ds, **kw)
eng = VAEEngine(nets['encoder'].layers, nets['decoder'].layers, nets['encoder'].input_shape,
                nets['latents'].event_shape[0], B, dev, observation=nets['observation'].posterior,
                tc=kind if kind == 'betatc' else None)
bench.init_params_(eng, seed=1)
x = bench.synthetic_batch(wl, B, nets['encoder'].input_shape, dev, seed=100)
for _ in range(3):
  eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=False)
torch.cuda.synchronize()
arr = eng._jobs_keepalive
tot = 0
for j in arr:
  st = j.stride if j.stride > 0 else j.n
  by = j.rows * j.n * 4
  tot += by
  print(f'  job n={j.n:7d} rows={j.rows:4d} stride={st:7d}  {by / 1e6:8.2f} MB')
print(f'{len(arr)} jobs, {tot / 1e6:.1f} MB of slab rows read')
L = eng.lib
st = eng.stream()
def timed(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
t = timed(lambda: L.odin_slab_reduce(arr, len(arr), st))
print(f'slab_reduce alone: {t:.1f} us = {tot / t * 1e-6:.2f} TB/s')
