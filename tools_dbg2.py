import sys, torch, numpy as np
sys.path.insert(0, '.')
from odin_ai_amd.networks import get_networks
from odin_ai_amd.vae import BetaTCVAE, FactorVAE
dev = torch.device('cuda:0')
fv = FactorVAE(device=dev, **get_networks('shapes3d'))
x = torch.rand(256, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
for i in range(2):
  loss, m = fv.optimize(x, learning_rate=2e-4, global_clipnorm=100.0)
  torch.cuda.synchronize(); print('factor step', i, float(loss), flush=True)
tcv = BetaTCVAE(beta=4.0, device=dev, **get_networks('celeba'))
xc = torch.rand(512, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
eng = tcv._engine(512)
eng.step_count = 0
import ctypes as C
for i in range(2):
  eng.step_count += 1
  eng.set_hyper(lr=2e-4, beta=4.0)
  torch.cuda.synchronize(); print('hyper ok', flush=True)
  eng.forward(xc, None)
  torch.cuda.synchronize(); print('fwd ok', flush=True)
  eng.backward()
  torch.cuda.synchronize(); print('bwd ok', flush=True)
  eng.adam(global_clipnorm=100.0)
  torch.cuda.synchronize(); print('adam ok', float(eng.out4[0]), flush=True)
