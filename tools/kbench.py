#!/usr/bin/env python3
"""Stand-alone launches of the three dominant plane kernels (dSprites / Shapes3D decoder4 shapes, B=256):
fused tail with 1 and 3 logit maps, decoder4 data gradient (fconv_planes), decoder4 weight gradient
(wgrad_planes).  Prints HIP-event times; run under `rocprofv3 --pmc ...` for counters."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB') or None)
dev = torch.device('cuda:0')
REPS = int(os.environ.get('KB_REPS', '20'))
WHICH = os.environ.get('KB_WHICH', 'tail1,tail3,dgrad,wgrad').split(',')


def timed(fn, n=REPS):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


B, H, W = int(os.environ.get('KB_B', '256')), 32, 32
d = _lib.conv_desc(B, H, W, 32, 2 * H, 2 * W, 32, 4, 2, 1, 1, 'elu')
x = torch.randn(B, H, W, 32, device=dev); w = torch.randn(4, 4, 32, 32, device=dev) * 0.1
b = torch.randn(32, device=dev) * 0.1
g = torch.randn(B, 2 * H, 2 * W, 32, device=dev)
if os.environ.get('KB_ZERO'):  # all-zero operands: the same instruction stream at minimal switching power
  x.zero_(); w.zero_(); g.zero_()
rows, npart = C.c_int(0), C.c_int(0)
# range word of the gradient tensor g (include/odin_hip.h: odin_conv_desc.dy_amax), as the engine's step keeps it
words = torch.zeros(2048, dtype=torch.int32, device=dev)
L.odin_absmax(g.data_ptr(), g.numel(), words.data_ptr(), None)
d.dy_amax = words.data_ptr()
for C1 in (1, 3):
  if f'tail{C1}' not in WHICH: continue
  w1 = torch.randn(32, C1, device=dev) * 0.3; b1 = torch.randn(C1, device=dev)
  tgt = torch.rand(B, 2 * H, 2 * W, C1, device=dev); sc = torch.tensor([1.0 / B], device=dev)
  lg = torch.empty(B, 2 * H, 2 * W, C1, device=dev)
  L.odin_bernoulli_tail_fwd_bwd(1, None, None, None, None, None, None, None, None, None, C.byref(npart),
                                None, C.byref(rows), None, C.byref(d), C1, None)
  part = torch.empty(B * npart.value, device=dev); slab = torch.empty(rows.value, 32 * C1 + C1 + 32, device=dev)
  fn = lambda: L.odin_bernoulli_tail_fwd_bwd(1, x.data_ptr(), w.data_ptr(), b.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                             tgt.data_ptr(), lg.data_ptr(), g.data_ptr(), part.data_ptr(), C.byref(npart),
                                             slab.data_ptr(), C.byref(rows), sc.data_ptr(), C.byref(d), C1, None)
  print(f'tail C1={C1}: {timed(fn):.1f} us  [{L.odin_debug_last_path().decode()}]')
if 'dgrad' in WHICH:
  dx = torch.empty(B, H, W, 32, device=dev); aux = torch.randn(B, H, W, 32, device=dev)
  bs = torch.empty(L.odin_max_slab_rows(), 32, device=dev)
  fn = lambda: L.odin_deconv2d_dgrad(g.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), bs.data_ptr(),
                                     C.byref(rows), C.byref(d), None)
  print(f'dec4 dgrad: {timed(fn):.1f} us  [{L.odin_debug_last_path().decode()}]')
if 'wgrad' in WHICH:
  L.odin_deconv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
  ws = torch.empty(rows.value, 16 * 32 * 32, device=dev)
  fn = lambda: L.odin_deconv2d_wgrad(x.data_ptr(), g.data_ptr(), ws.data_ptr(), C.byref(rows), C.byref(d), None)
  print(f'dec4 wgrad: {timed(fn):.1f} us  [{L.odin_debug_last_path().decode()}]')
if 'bwd' in WHICH or 'wgrad' in WHICH:
  # the same layer's two gradients in ONE launch (bwd_planes.hip: dy fetched and split once), and dec3's pair
  for (hh, cin, cout, name) in ((32, 32, 32, 'dec4'), (16, 64, 32, 'dec3'), (8, 64, 64, 'dec2')):
    dd = _lib.conv_desc(B, hh, hh, cin, 2 * hh, 2 * hh, cout, 4, 2, 1, 1, 'elu')
    xx = torch.randn(B, hh, hh, cin, device=dev); ww = torch.randn(4, 4, cout, cin, device=dev) * 0.1
    gg = torch.randn(B, 2 * hh, 2 * hh, cout, device=dev); aux = torch.randn(B, hh, hh, cin, device=dev)
    L.odin_absmax(gg.data_ptr(), gg.numel(), words.data_ptr(), None)
    dd.dy_amax = words.data_ptr()
    dx = torch.empty(B, hh, hh, cin, device=dev)
    bs = torch.empty(L.odin_max_slab_rows() * 2, cin, device=dev)
    r1, r2 = C.c_int(0), C.c_int(0)
    L.odin_deconv2d_wgrad(None, None, None, C.byref(r1), C.byref(dd), None)
    r3, r4 = C.c_int(0), C.c_int(0)
    L.odin_deconv2d_bwd(None, None, None, None, 1, None, None, C.byref(r4), None, C.byref(r3), C.byref(dd), None)
    ws = torch.empty(max(r1.value, r3.value, 1), 16 * cout * cin, device=dev)
    f2 = lambda: (L.odin_deconv2d_wgrad(xx.data_ptr(), gg.data_ptr(), ws.data_ptr(), C.byref(r1), C.byref(dd), None),
                  L.odin_deconv2d_dgrad(gg.data_ptr(), ww.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), bs.data_ptr(),
                                        C.byref(r2), C.byref(dd), None))
    t2 = timed(f2)
    f1 = lambda: L.odin_deconv2d_bwd(xx.data_ptr(), gg.data_ptr(), ww.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(),
                                     bs.data_ptr(), C.byref(r2), ws.data_ptr(), C.byref(r1), C.byref(dd), None)
    t1 = timed(f1)
    print(f'{name} wgrad + dgrad, two launches: {t2:.1f} us; one launch: {t1:.1f} us  [{L.odin_debug_last_path().decode()}]')
