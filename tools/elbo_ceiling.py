#!/usr/bin/env python3
"""Diagnostics: what a 2-arrays-in / 1-array-out stream of the fused ELBO kernel's length reaches from cold HBM on this
box (launches rotate through buffer sets larger than the 256 MB Infinity Cache), in several launch shapes, beside the
ELBO kernel itself and torch.add."""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
B, n = 256, 64 * 64 * 3
N = B * n
nsets = int(math.ceil(1.25 * 256 * 2**20 / (12.0 * N)))
sets = [(torch.randn(N, device=dev), torch.rand(N, device=dev), torch.empty(N, device=dev)) for _ in range(nsets)]
sc = torch.tensor([1.0 / B], device=dev)
npart = C.c_int(0)
L.odin_elbo_bernoulli_fwd_bwd(None, None, None, None, None, B, n, C.byref(npart), None)
part = torch.empty(B * npart.value, device=dev)


def timeit(fns, reps=6):
  for f in fns: f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    for f in fns: f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / (reps * len(fns)) * 1e-3


def report(name, t):
  print(f'{name:60s} {t*1e6:7.2f} us  {12.0*N/t*1e-9:7.1f} GB/s  {12.0*N/t*1e-12/8.0:5.3f} of 8 TB/s')


report('elbo_bernoulli_fwd_bwd', timeit([lambda a=a, b=b, c=c: L.odin_elbo_bernoulli_fwd_bwd(a.data_ptr(), b.data_ptr(), part.data_ptr(), c.data_ptr(), sc.data_ptr(), B, n, C.byref(npart), None) for a, b, c in sets]))
report('torch.add(a, b, out=c)', timeit([lambda a=a, b=b, c=c: torch.add(a, b, out=c) for a, b, c in sets]))
for variant, nm in ((0, 'wave-chunked U=3 (the ELBO kernel shape)'), (1, 'grid-stride U=4'), (2, 'grid-stride U=4 non-temporal'),
                    (3, 'grid-stride U=8'), (4, 'grid-stride U=8 non-temporal'), (5, 'grid-stride U=2'),
                    (6, 'U=4, b non-temporal'), (7, 'U=4, a and b non-temporal'), (8, 'U=4, stores non-temporal'),
                    (9, 'U=4, b and stores non-temporal')):
  for blocks in ((0,) if variant == 0 else (512, 2048) if variant >= 6 else (256, 512, 1024, 2048, 4096)):
    report(f'probe {variant} {nm}, blocks {blocks or "-"}',
           timeit([lambda a=a, b=b, c=c: L.odin_debug_stream_probe(a.data_ptr(), b.data_ptr(), c.data_ptr(), N, variant, blocks, None) for a, b, c in sets]))
