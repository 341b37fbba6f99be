import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks
dev = torch.device('cuda:0')
nets = get_networks('dsprites')
enc, dec = nets['encoder'].layers, nets['decoder'].layers
in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
eng = VAEEngine(enc, dec, in_shape, zdim, 256, dev, observation=nets['observation'].posterior, seed=1)
bench.init_params_(eng, seed=1)
x = bench.synthetic_batch('dsprites_betavae_b256', 256, in_shape, dev, seed=100)
L = eng.lib
print('fallbacks before', L.odin_debug_absmax_fallbacks())
for _ in range(3):
  eng.train_step(x, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=False)
torch.cuda.synchronize()
print('fallbacks after 3 eager steps', L.odin_debug_absmax_fallbacks(), 'flag', eng.flag.item())
ops = bench.profile_ops(eng)
print('fallbacks after profile_ops', L.odin_debug_absmax_fallbacks())
for o in ops:
  print(f"{o['layer']:14s} {o['op']:6s} {o['us']:8.1f} us  {o['path']}")
print('sum', sum(o['us'] for o in ops))
import time
for ug in (False, True):
  for _ in range(50): eng.train_step(x, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=ug)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(200): eng.train_step(x, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=ug)
  torch.cuda.synchronize(); print('graph', ug, (time.perf_counter() - t0) / 200 * 1e3, 'ms/step', 'flag', eng.flag.item())
