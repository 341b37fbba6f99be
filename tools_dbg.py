import sys, torch, numpy as np
sys.path.insert(0, '.')
from odin_ai_amd.networks import get_networks
from odin_ai_amd.vae import BetaTCVAE, FactorVAE
dev = torch.device('cuda:0')
which = sys.argv[1]
if which == 'factor':
  fv = FactorVAE(device=dev, **get_networks('shapes3d'))
  x = torch.rand(256, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
  for i in range(3):
    loss, m = fv.optimize(x, learning_rate=2e-4, global_clipnorm=100.0)
    torch.cuda.synchronize(); print('factor step', i, float(loss), {k: float(v) for k, v in m.items()}, flush=True)
else:
  tcv = BetaTCVAE(beta=4.0, device=dev, **get_networks('celeba'))
  xc = torch.rand(512, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
  for i in range(3):
    loss, m = tcv.optimize(xc, learning_rate=2e-4, global_clipnorm=100.0)
    torch.cuda.synchronize(); print('tc step', i, float(loss), {k: float(v) for k, v in m.items()}, flush=True)
