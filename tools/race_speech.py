#!/usr/bin/env python3
"""run-to-run determinism of the audio VAE's forward + backward (same inputs, same parameters): every gradient tensor and
output bit-identical across repetitions -- a difference means a race in a kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from odin_ai_amd import _lib
from odin_ai_amd.engine import VAEEngine
from odin_ai_amd.networks import get_networks
from oracle import vae_oracle as vo
L = _lib.load(); dev = torch.device('cuda:0')
if os.environ.get('NO_BLK'):
  L.odin_debug_blk_planes(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
DS = sys.argv[3] if len(sys.argv) > 3 else 'speech'
nets = get_networks('speech', n_frames=96, n_mels=80) if DS == 'speech' else get_networks(DS)
OBS = 'gaussian_softplus1' if DS == 'speech' else 'bernoulli'
enc, dec = nets['encoder'].layers, nets['decoder'].layers
in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
rng = np.random.default_rng(5)
x = np.clip(rng.random((B,) + tuple(in_shape)), 1e-6, 1 - 1e-6).astype(np.float32)
eps = rng.standard_normal((B, zdim)).astype(np.float32)
om = vo.OracleVAE(enc, dec, in_shape, zdim, observation=OBS, beta=1.0)
P = om.init_params(seed=9)
eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation=OBS, lib=L)
eng.load_params({k: v.astype(np.float32).astype(np.float64) for k, v in P.items()})
eng.step_count = 1
eng.set_hyper(lr=1e-3, beta=1.0)
tx, te = torch.tensor(x, device=dev), torch.tensor(eps, device=dev)
ref = None
bad = 0
for it in range(N):
  eng.forward(tx, te)
  eng.backward()
  torch.cuda.synchronize()
  cur = {k: v.clone() for k, v in eng.grad_views().items()}
  cur['llk'] = eng.llk.clone(); cur['out4'] = eng.out4.clone()
  for i, g in enumerate(eng.enc.outs):
    cur[f'A.enc.out{i}'] = g.clone()
  cur['A.z'] = eng.z.clone()
  for i, g in enumerate(eng.dec.outs):
    cur[f'B.dec.out{i}'] = g.clone()
  for i, g in enumerate(eng.dec.gouts):
    cur[f'dec.gout{i}'] = g.clone()
  for i, g in enumerate(eng.enc.gouts):
    cur[f'enc.gout{i}'] = g.clone()
  if ref is None:
    ref = cur
    continue
  diffs = [(k, float((cur[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in ref if not torch.equal(cur[k], ref[k])]
  if diffs:
    bad += 1
    k0 = sorted(k for k, d, m in diffs if not isinstance(k, tuple))[0]
    dd = (cur[k0] - ref[k0]).abs()
    idx = torch.nonzero(dd > 0)
    print('  first differing tensor', k0, tuple(cur[k0].shape), 'elements', idx.shape[0], 'first', idx[:4].tolist(), 'last', idx[-2:].tolist())
    print('iteration', it, 'differs:', ' '.join(f"{k}:{d / max(m, 1e-30):.1e}" for k, d, m in diffs if not isinstance(k, tuple)), '| params:', len([1 for k, d, m in diffs if isinstance(k, tuple)]))
print('fused_tail', eng.fused_tail, 'iterations', N, 'with differences', bad)
