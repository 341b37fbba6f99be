import os, sys, time, torch
sys.path.insert(0, '.')
from oracle.torch_ref import TorchTrainer, TorchVAE
from oracle import vae_oracle as vo
enc, dec, shp, D = vo.dsprites_spec(1)
m = vo.OracleVAE(enc, dec, shp, D, beta=4.0)
P = m.init_params(1)
x = torch.rand(256, 64, 64, 1); e = torch.randn(256, 10)
print('cpu_count', os.cpu_count())
for th in (8, 16, 32, 64, 128):
  if th > (os.cpu_count() or 1): break
  tr = TorchTrainer(TorchVAE(enc, dec, shp, D, beta=4.0, dtype=torch.float32), P, threads=th)
  tr.step(x, e)
  t0 = time.perf_counter(); n = 2
  for _ in range(n): tr.step(x, e)
  dt = (time.perf_counter() - t0) / n
  print(th, 'threads:', round(dt, 3), 's/step', round(256 / dt, 1), 'img/s', flush=True)
