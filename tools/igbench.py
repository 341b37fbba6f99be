#!/usr/bin/env python3
"""Stand-alone timings of the small-spatial layers (encoder3 / decoder1 of the dSprites and Shapes3D stacks,
the speech stack's 6x5 layers) through the C ABI: forward, data gradient, weight gradient.
ODIN_NOIGEMM=1 times the tiled paths instead; ODIN_IG_NW / ODIN_IG_R override the launch geometry."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB') or None)
dev = torch.device('cuda:0')
REPS = int(os.environ.get('KB_REPS', '50'))


def timed(fn, n=REPS):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def same_pads(n, k, s):
  out = -(-n // s); total = max((out - 1) * s + k - n, 0)
  return out, total // 2


CASES = [  # kind, B, H, W, Ci, Co, K, S
    ('conv', 256, 8, 8, 64, 64, 4, 2), ('deconv', 256, 4, 4, 8, 64, 4, 2), ('deconv', 256, 4, 4, 16, 64, 4, 2),
    ('conv', 256, 12, 10, 64, 64, 4, 2), ('deconv', 256, 6, 5, 8, 64, 4, 2)]
if os.environ.get('IG_MID'):  # the 8- and 16-pixel layers of the image stacks (plane kernels by default)
  CASES = [('conv', 256, 16, 16, 32, 64, 4, 2), ('deconv', 256, 8, 8, 64, 64, 4, 2), ('deconv', 256, 16, 16, 64, 32, 4, 2),
           ('conv', 256, 32, 32, 32, 32, 4, 2)]
for kind, B, H, W, Ci, Co, K, S in CASES:
  if kind == 'conv':
    OH, pt = same_pads(H, K, S); OW, pl = same_pads(W, K, S)
    wshape, nb = (K, K, Ci, Co), Co
  else:
    OH, OW = H * S, W * S
    _, pt = same_pads(OH, K, S); _, pl = same_pads(OW, K, S)
    wshape, nb = (K, K, Co, Ci), 0
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  x = torch.randn(B, H, W, Ci, device=dev); w = torch.randn(*wshape, device=dev) * 0.1
  b = torch.randn(Co, device=dev) * 0.1; y = torch.empty(B, OH, OW, Co, device=dev)
  dy = torch.randn(B, OH, OW, Co, device=dev); dx = torch.empty(B, H, W, Ci, device=dev)
  aux = torch.randn(B, H, W, Ci, device=dev); bs = torch.empty(L.odin_max_slab_rows(), Ci, device=dev)
  rows = C.c_int(0)
  fwd = getattr(L, f'odin_{kind}2d_fwd'); dg = getattr(L, f'odin_{kind}2d_dgrad'); wg = getattr(L, f'odin_{kind}2d_wgrad')
  t_f = timed(lambda: fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), None))
  p_f = L.odin_debug_last_path().decode()
  t_d = timed(lambda: dg(dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), bs.data_ptr(), C.byref(rows),
                         C.byref(d), None))
  p_d = L.odin_debug_last_path().decode()
  wg(None, None, None, C.byref(rows), C.byref(d), None)
  slab = torch.empty(rows.value, K * K * Ci * Co + nb, device=dev)
  t_w = timed(lambda: wg(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None))
  p_w = L.odin_debug_last_path().decode()
  gf = 2.0 * B * (OH * OW if kind == 'conv' else H * W) * K * K * Ci * Co / 1e9
  print(f'{kind:6s} B{B} {H}x{W}x{Ci}->{OH}x{OW}x{Co}  {gf:.3f} GF  fwd {t_f:5.1f} [{p_f}]  dgrad {t_d:5.1f} [{p_d}]  '
        f'wgrad {t_w:5.1f} [{p_w}] rows {rows.value}')
