"""Diagnostics: fused ELBO kernels, loads-in-flight sweep (ODIN_ELBO_U) at 64x64x3 / 64x64x1, B=256."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odin_ai_amd import _lib

L = _lib.load()
dev = torch.device('cuda:0')
sc = torch.tensor([1 / 256.], device=dev)
npart = C.c_int(0)


def timeit(fn, reps=100):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


for (B, npix, Cc) in ((256, 4096, 3), (256, 4096, 1), (512, 4096, 3)):
  n = npix * Cc
  lg, x = torch.randn(B, n, device=dev), torch.rand(B, n, device=dev)
  dl = torch.empty_like(lg)
  part = torch.empty(B * 64, device=dev)
  for u in ('0', '1', '2', '3', '4'):
    os.environ['ODIN_ELBO_U'] = u
    os.putenv('ODIN_ELBO_U', u)
    t = timeit(lambda: L.odin_elbo_bernoulli_fwd_bwd(lg.data_ptr(), x.data_ptr(), part.data_ptr(),
                                                     dl.data_ptr(), sc.data_ptr(), B, n,
                                                     C.byref(npart), None))
    print(f'bernoulli B={B} C={Cc} U={u}: {t:7.2f} us  {12.0 * B * n / t * 1e-6:8.1f} GB/s', flush=True)
  os.unsetenv('ODIN_ELBO_U')
  del os.environ['ODIN_ELBO_U']
  h = torch.randn(B, npix, 2 * Cc, device=dev)
  dh = torch.empty_like(h)
  for sp1 in (0, 1):
    t = timeit(lambda: L.odin_elbo_gaussian_fwd_bwd(h.data_ptr(), x.data_ptr(), part.data_ptr(),
                                                    dh.data_ptr(), sc.data_ptr(), B, npix, Cc, sp1,
                                                    C.byref(npart), None))
    print(f'gaussian  B={B} C={Cc} sp1={sp1}: {t:7.2f} us  {20.0 * B * n / t * 1e-6:8.1f} GB/s', flush=True)
# device-to-device copy of the same byte count for reference (read 25.2 MB + write 12.6 MB ~ a
# 18.9 MB copy moves 37.7 MB)
a = torch.empty(18874368 // 4, device=dev)
b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a))
print(f'torch copy 18.9 MB (37.7 MB moved): {t:7.2f} us {2 * a.numel() * 4 / t * 1e-6:8.1f} GB/s')
