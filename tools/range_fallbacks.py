"""Fallback range passes (odin_absmax over a whole tensor: a producer that does not track its outputs, or a consumer
without a word) per training step of every benchmark workload: must be 0 on the benchmark paths -- a pass over an
activation tensor costs as much as the layer that wrote it.  usage: python tools/range_fallbacks.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks
dev = torch.device('cuda:0')
for name in ('dsprites_betavae_b256', 'shapes3d_vae_b256', 'celeba_betatcvae_b512', 'mnist_dense_b128', 'mnist_conv_b128',
             'speech_vae_b256'):
  ds, kw, B, beta, kind = bench.WORKLOADS[name]
  nets = get_networks(ds, **kw)
  enc, dec = nets['encoder'].layers, nets['decoder'].layers
  in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation=nets['observation'].posterior,
                  tc=kind if kind == 'betatc' else None, seed=1)
  bench.init_params_(eng, seed=1)
  x = bench.synthetic_batch(name, B, in_shape, dev, seed=100)
  L = eng.lib
  eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=False)
  torch.cuda.synchronize()
  f0 = L.odin_debug_absmax_fallbacks()
  for _ in range(3):
    eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=False)
  torch.cuda.synchronize()
  n = (L.odin_debug_absmax_fallbacks() - f0) / 3
  words = sum(w is not None for p in (eng.enc, eng.dec) for w in p.x_word)
  print(f'{name:26s} fallback passes per step {n:.1f}   layers reading an activation word {words}   flag {eng.flag.item()}')
  del eng
  torch.cuda.empty_cache()
