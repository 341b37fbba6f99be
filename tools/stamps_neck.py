#!/usr/bin/env python3
"""Phase times of the neck launches (neck.hip) from in-kernel wall-clock stamps of workgroup 0 (100 MHz):
usage: tools/stamps_neck.py [workload]  (default dsprites_betavae_b256)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks

wl = sys.argv[1] if len(sys.argv) > 1 else 'dsprites_betavae_b256'
ds, kw, B, beta, kind = bench.WORKLOADS[wl]
if kind == 'factor':
  B //= 2
nets = get_networks(ds, **kw)
dev = torch.device('cuda:0')
eng = VAEEngine(nets['encoder'].layers, nets['decoder'].layers, nets['encoder'].input_shape,
                nets['latents'].event_shape[0], B, dev, observation=nets['observation'].posterior, seed=1)
bench.init_params_(eng, seed=1)
x = bench.synthetic_batch(wl, B, nets['encoder'].input_shape, dev, seed=3)
buf = torch.zeros(32, dtype=torch.int64, device=dev)
for it in range(6):
  if it == 3:
    eng.lib.odin_debug_set_neck_stamps(buf.data_ptr())
  eng.train_step(x, None, lr=1e-3, beta=beta, global_clipnorm=100.0)
  torch.cuda.synchronize()
  if it >= 3:
    s = buf.cpu().tolist()
    names_f = ['loads+stage', 'conv3 mfma', 'partials', 'reduce', 'projection', 'latent', 'deconv1']
    f = [(s[i + 1] - s[i]) / 100.0 for i in range(6)]
    names_b = ['loads', 'dW1', 'dx1', 'latent', 'dh4+slabs', 'commit', 'dy3 (W4)', 'planes+W3', 'mfma', 'epilogue']
    b = [(s[i + 1] - s[i]) / 100.0 for i in range(8, 17)]
    print('fwd us:', ' '.join(f'{n}={v:.2f}' for n, v in zip(['stage', 'mfma', 'part', 'proj', 'latent', 'deconv1'], f)), 'total', (s[6] - s[0]) / 100.0)
    print('bwd us:', ' '.join(f'{n}={v:.2f}' for n, v in zip(['loads', 'dW1', 'dx1', 'latent', 'dh4+slabs', 'dy3(W4)', 'max+planes', 'mfma', 'epilogue'], b)), 'total', (s[17] - s[8]) / 100.0)
eng.lib.odin_debug_set_neck_stamps(None)
