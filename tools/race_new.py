#!/usr/bin/env python3
"""run-to-run determinism of this round's later kernels at their benchmark sizes: dense_hw / dense_hd / the paired launch,
the discriminator head (fixed-point mean through one atomic), the folded Adam, the n_fft 512 front-end -- N launches each on
the same inputs, every output compared bit for bit with the first launch's.  usage: tools/race_new.py [N]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from odin_ai_amd import _lib
from odin_ai_amd._lib import AdamFold
L = _lib.load()
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)


def word(t):
  w = torch.zeros(2048, dtype=torch.int32, device=dev)
  L.odin_absmax(t.data_ptr(), t.numel(), w.data_ptr(), None)
  return w


def stress(name, fn, outs):
  ref, bad = None, 0
  for i in range(N):
    for o in outs: o.fill_(float('nan')) if o.is_floating_point() else o.zero_()
    fn()
    torch.cuda.synchronize()
    cur = [o.clone() for o in outs]
    if ref is None: ref = cur
    elif not all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, cur)): bad += 1
  print(f'{name:44s} {bad}/{N - 1} launches differ', flush=True)
  return bad


tot = 0
for B, K, Nn in [(256, 1000, 1000), (128, 1000, 1000), (512, 4096, 512)]:
  x, dy = torch.randn(B, K, device=dev), torch.randn(B, Nn, device=dev) * 1e-3
  w, aux = torch.randn(K, Nn, device=dev) / K ** 0.5, torch.randn(B, K, device=dev)
  dx, slab = torch.empty(B, K, device=dev), torch.empty(1, K * Nn + Nn, device=dev)
  xw, dyw, dxw = word(x), word(dy), torch.zeros(2048, dtype=torch.int32, device=dev)
  rows = C.c_int(0)
  pair = lambda: L.odin_dense_bwd_ranged(x.data_ptr(), dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), None, None,
                                         slab.data_ptr(), C.byref(rows), B, K, Nn, 1, 1, dyw.data_ptr(), dxw.data_ptr(), xw.data_ptr(), None)
  wg = lambda: L.odin_dense_bwd_ranged(x.data_ptr(), dy.data_ptr(), None, None, 0, None, None, None, slab.data_ptr(),
                                       C.byref(rows), B, K, Nn, 1, 0, dyw.data_ptr(), None, xw.data_ptr(), None)
  dg = lambda: L.odin_dense_bwd_ranged(None, dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), None, None, None,
                                       None, B, K, Nn, 0, 1, dyw.data_ptr(), dxw.data_ptr(), None, None)
  for nm, fn, outs in (('pair', pair, [dx, slab, dxw]), ('wgrad', wg, [slab]), ('dgrad', dg, [dx, dxw])):
    fn(); p = L.odin_debug_last_path().decode()
    tot += stress(f'dense {nm} [{B} x {K} x {Nn}] {p}', fn, outs)
# discriminator head, both modes
for B, mode in ((128, 0), (256, 1)):
  K = 1000
  h, w, b = torch.relu(torch.randn(B, K, device=dev)), torch.randn(K, device=dev) / 30, torch.randn(1, device=dev)
  dl = torch.full((B,), 7.0 / B, device=dev)
  logit, dlo, out = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(1, device=dev)
  dh, slab = torch.empty(B, K, device=dev), torch.empty(L.odin_disc_head_rows(B, K), K + 1, device=dev)
  wd, ws, r = torch.zeros(2048, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev), C.c_int(0)
  fn = lambda: L.odin_disc_head_fwd_bwd(h.data_ptr(), w.data_ptr(), b.data_ptr(), logit.data_ptr(), mode, dl.data_ptr(), dlo.data_ptr(),
                                        out.data_ptr(), 1, dh.data_ptr(), wd.data_ptr(), slab.data_ptr(), C.byref(r), ws.data_ptr(), B, K, None)
  outs = [logit, out, dh, slab, wd] + ([dlo] if mode == 1 else [])
  tot += stress(f'disc_head mode {mode} [{B} x {K}]', fn, outs)
  assert int(ws.abs().sum()) == 0
# folded Adam at the discriminator's size
n = 4012004
th0, g0 = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-2
m0, v0 = torch.randn(n, device=dev) * 1e-2, torch.rand(n, device=dev) * 1e-3
x, dy = torch.randn(256, 6, device=dev), torch.randn(256, 1000, device=dev) * 1e-2
slab = torch.randn(32, 1001, device=dev) * 1e-2
hy = torch.tensor([1e-5, 0.5, 0.9, 1e-7, 1.0], device=dev)
th, g, m, v = th0.clone(), g0.clone(), m0.clone(), v0.clone()
words = torch.ones(4096, dtype=torch.int32, device=dev)
fo = AdamFold(x.data_ptr(), dy.data_ptr(), 256, 6, 1000, 0, slab.data_ptr(), 32, 1001, 1001, 4011000, words.data_ptr(), 4096)
def fold():
  th.copy_(th0); g.copy_(g0); m.copy_(m0); v.copy_(v0)
  L.odin_adam_step_fold(th.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, hy.data_ptr(), C.byref(fo), None)
ref = None; bad = 0
for i in range(N):
  fold(); torch.cuda.synchronize()
  cur = [t.clone() for t in (th, g, m, v)]
  if ref is None: ref = cur
  elif not all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, cur)): bad += 1
print(f'{"adam_fold [4012004]":44s} {bad}/{N - 1} launches differ', flush=True); tot += bad
# front-end
from odin_ai_amd.mel import MelsSpecExtractor
y = torch.randn(256, 8000, device=dev) * 0.1
ex = MelsSpecExtractor(device=dev, lib=L, unit_range=True)
buf = torch.empty(256, 96, 80, 1, device=dev)
tot += stress('stft_mel512 [256 x 8000] -> [256, 96, 80]', lambda: ex(y, out=buf), [buf])
print('TOTAL', tot)
sys.exit(1 if tot else 0)
