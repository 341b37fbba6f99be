#!/usr/bin/env python3
"""Round-6 sweep of the persistent Bernoulli ELBO kernel's launch shape (workgroups x chunks in flight x pipelined) on the
two shapes bench.py prices (64x64x3 at batch 256 and 512), cold (rotating buffer sets > 256 MB), beside the stream probe."""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
n = 64 * 64 * 3
for B in (256, 512):
  N = B * n
  nsets = int(math.ceil(1.25 * 256 * 2**20 / (12.0 * N)))
  sets = [(torch.randn(N, device=dev), torch.rand(N, device=dev), torch.empty(N, device=dev)) for _ in range(nsets)]
  sc = torch.tensor([1.0 / B], device=dev)
  npart = C.c_int(0)
  L.odin_elbo_bernoulli_fwd_bwd(None, None, None, None, None, B, n, C.byref(npart), None)
  part = torch.empty(B * npart.value, device=dev)
  def timeit(fns, reps=8):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e-3
  t = timeit([lambda a=a, b=b, c=c: L.odin_debug_stream_probe(a.data_ptr(), b.data_ptr(), c.data_ptr(), N, 8, 512, None) for a, b, c in sets])
  print(f'B={B} probe 8 (512 blocks, nt stores): {t*1e6:.2f} us {12.0*N/t*1e-12/8.0:.3f}')
  for pipe in (0, 1):
    for blocks in (512, 1024, 2048, 4096):
      for U in (1, 2, 3, 4):
        if pipe == 0 and U == 1: continue
        L.odin_debug_elbo_shape(blocks, U, pipe)
        t = timeit([lambda a=a, b=b, c=c: L.odin_elbo_bernoulli_fwd_bwd(a.data_ptr(), b.data_ptr(), part.data_ptr(), c.data_ptr(), sc.data_ptr(), B, n, C.byref(npart), None) for a, b, c in sets])
        print(f'B={B} pipe={pipe} blocks={blocks} U={U}: {t*1e6:.2f} us  {12.0*N/t*1e-12/8.0:.3f} of 8 TB/s')
L.odin_debug_elbo_shape(0, 1, 1)
