import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from odin_ai_amd import _lib
from odin_ai_amd.networks import get_networks
from odin_ai_amd.vae import FactorVAE
import tests.factor_util as fu
L = _lib.load()
dev = torch.device('cuda:0')
nets = get_networks('shapes3d')
B1, D, units = 128, 6, (1000,) * 5
fv = FactorVAE(discriminator_units=units, tc_coef=7.0, device=dev, lib=L, **nets)
rng = np.random.default_rng(33)
x = np.clip(rng.random((2 * B1, 64, 64, 3)), 1e-6, 1 - 1e-6).astype(np.float32)
eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
try:
  rep = fu.check_factor_vae_full_size(fv, nets, units, B1, x, eps, eps2, perm, lr=1e-3, clip=100.0, threads=32, tol=1.0)
except AssertionError as e:
  rep = e.args[0][2]
for k, v in rep.items():
  if 'dgrad' in k or 'grad' not in k: print(k, f'{float(v):.2e}')
disc = fv._discriminator(B1)
for name, pr in (('prog1', disc.prog1), ('prog2', disc.prog2)):
  w = pr.range_words.view(torch.float32).view(len(pr.recs), -1).max(1).values.cpu().numpy()
  g = [float(t.abs().max()) for t in pr.gouts]
  print(name, 'words', w, 'gouts max', g, 'dy_word', [x is not None for x in pr.dy_word])
