#!/usr/bin/env python3
"""same-process A/B: igemm_h weight slices of up to 32 (round 5) / 64 steps in LDS, on the MNIST conv and speech steps"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
for wl in ('mnist_conv_b128', 'celeba_betatcvae_b512'):
  ds, kw, B, beta, kind = bench.WORKLOADS[wl]
  nets = get_networks(ds, **kw)
  for steps in (32, 64, 32, 64):
    L.odin_debug_igemm_h_ldsw_steps(steps)
    eng = VAEEngine(nets['encoder'].layers, nets['decoder'].layers, nets['encoder'].input_shape,
                    nets['latents'].event_shape[0], B, dev, observation=nets['observation'].posterior, seed=1,
                    tc='betatc' if kind == 'betatc' else None)
    bench.init_params_(eng, seed=1)
    x = eng.input_buffer(); x.copy_(bench.synthetic_batch(wl, B, nets['encoder'].input_shape, dev, seed=3))
    for _ in range(30): eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{wl} ldsw_steps={steps}: {dt / 100 * 1e3:.4f} ms/step', flush=True)
    del eng
L.odin_debug_igemm_h_ldsw_steps(32)
