#!/usr/bin/env python3
"""Does the ~8 us between two step graphs shrink when K steps share one graph?  (possible since the hyper-parameter
ring: no host work between steps).  Prints ms per step for K = 1, 2, 4, 8 on dsprites_betavae_b256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks
from odin_ai_amd.dist import SegmentedGraph
from bench import init_params_, synthetic_batch

dev = torch.device('cuda:0')
nets = get_networks('dsprites')
enc, dec = nets['encoder'].layers, nets['decoder'].layers
in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
B = 256
for K in (1, 2, 4, 8, 1):
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation='bernoulli', seed=1)
  init_params_(eng, seed=1)
  xb = eng.input_buffer()
  xb.copy_(synthetic_batch('dsprites_betavae_b256', B, in_shape, dev, seed=100))
  pol = (100.0, None, None, None, True, True)
  # prime the ring far ahead: constant hyper-parameters
  for _ in range(3):
    eng.train_step(xb, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=True)
  torch.cuda.synchronize()
  prog = []
  for _ in range(K):
    prog += eng.step_program(xb, None, pol)
  sg = SegmentedGraph(dev, prog)
  cap = torch.cuda.Stream(dev)
  cap.wait_stream(torch.cuda.current_stream(dev))
  with torch.cuda.stream(cap):
    sg.run_eager()
  torch.cuda.current_stream(dev).wait_stream(cap)
  torch.cuda.synchronize()
  sg.capture(cap)

  def block(n):
    # n graph replays = n * K steps; the host keeps the ring filled (as train_step does)
    for _ in range(n):
      for _ in range(K):
        eng.step_count += 1
        eng._ring_step(eng.step_count, dict(lr=1e-3, beta=4.0, when_skip_update=0, capacity=None), None)
      sg.replay()
  t_end = time.perf_counter() + 0.6
  while time.perf_counter() < t_end:
    block(8)
    torch.cuda.synchronize()
  block(20)
  torch.cuda.synchronize()
  n = 200 // K
  t0 = time.perf_counter()
  block(n)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  assert eng.flag.item() == 0
  print(f'K = {K}: {dt / (n * K) * 1e3:.4f} ms per step ({n} replays)', flush=True)
