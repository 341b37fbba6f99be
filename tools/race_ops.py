#!/usr/bin/env python3
"""run-to-run determinism of every block-window kernel at the audio / MNIST layer sizes: each op launched N times on the same
inputs, every output compared BIT FOR BIT with the first launch (a difference = a race or a hardware hazard; cf. the
bk_mfma16 note in odin_ai_amd/csrc/blk_common.h).  usage: tools/race_ops.py [N]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odin_ai_amd import _lib
L = _lib.load(); dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device='cpu').manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)


def word(t):
  w = torch.zeros(2048, dtype=torch.int32, device=dev)
  L.odin_absmax(t.data_ptr(), t.numel(), w.data_ptr(), None)
  return w


def repeat(name, fn, outs):
  ref, bad, path = None, 0, ''
  for it in range(N):
    for o in outs:
      o.fill_(float('nan'))
    fn()
    torch.cuda.synchronize()
    path = L.odin_debug_last_path().decode()
    cur = [o.clone() for o in outs]
    if ref is None:
      ref = cur
    elif any(not torch.equal(a, b) for a, b in zip(cur, ref)):   # (NaN != NaN: an element left unwritten counts too)
      bad += 1
  print(f'{name:44s} [{path}] launches {N}  differing {bad}', flush=True)
  return bad


total = 0
LAYERS = [('deconv', 256, 12, 10, 64, 64, 4), ('deconv', 256, 24, 20, 64, 32, 4), ('deconv', 256, 48, 40, 32, 32, 4),
          ('conv', 256, 48, 40, 32, 32, 4), ('conv', 256, 24, 20, 32, 64, 4), ('conv', 256, 12, 10, 64, 64, 4),
          ('conv', 128, 14, 14, 32, 64, 5), ('conv', 128, 14, 14, 64, 64, 5), ('conv', 128, 28, 28, 32, 32, 5)]
for kind, B, H, W, Ci, Co, K in LAYERS:
  S = 2 if K == 4 else 1
  if kind == 'conv':
    OH, OW = H // S, W // S
    wt = R(K, K, Ci, Co, sc=0.05)
  else:
    OH, OW = 2 * H, 2 * W
    wt = R(K, K, Co, Ci, sc=0.05)
  pad = 1 if K == 4 else 2
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pad, pad, 'elu')
  x, b = R(B, H, W, Ci), R(Co, sc=0.1)
  y, dy = torch.empty(B, OH, OW, Co, device=dev), R(B, OH, OW, Co, sc=1e-3)
  dx, aux = torch.empty_like(x), R(B, H, W, Ci)
  xw, yw, dyw, dxw = word(x), torch.zeros(2048, dtype=torch.int32, device=dev), word(dy), torch.zeros(2048, dtype=torch.int32, device=dev)
  d.x_amax, d.y_amax, d.dy_amax, d.dx_amax = xw.data_ptr(), yw.data_ptr(), dyw.data_ptr(), dxw.data_ptr()
  n = K * K * Ci * Co + (Co if kind == 'conv' else 0)
  slab = torch.empty(L.odin_max_slab_rows(), n, device=dev)
  cs = torch.empty(L.odin_max_slab_rows(), Ci, device=dev)
  rows, r2 = C.c_int(0), C.c_int(0)
  fwd = getattr(L, f'odin_{kind}2d_fwd'); dg = getattr(L, f'odin_{kind}2d_dgrad'); wg = getattr(L, f'odin_{kind}2d_wgrad')
  bw = getattr(L, f'odin_{kind}2d_bwd')
  tag = f'{kind} k{K} {H}x{W} {Ci}->{Co}'
  total += repeat(tag + ' fwd', lambda: fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), st), [y])
  total += repeat(tag + ' dgrad', lambda: dg(dy.data_ptr(), wt.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), cs.data_ptr(), C.byref(rows), C.byref(d), st), [dx])
  total += repeat(tag + ' wgrad', lambda: wg(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), st), [slab[:1]])
  total += repeat(tag + ' bwd', lambda: bw(x.data_ptr(), dy.data_ptr(), wt.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), cs.data_ptr(), C.byref(r2), slab.data_ptr(), C.byref(rows), C.byref(d), st), [dx, slab[:1]])
# the fused Gaussian tail
B, H, W = 256, 48, 40
d = _lib.conv_desc(B, H, W, 32, 2 * H, 2 * W, 32, 4, 2, 1, 1, 'elu')
x, wt, b, w1, b1, t = R(B, H, W, 32), R(4, 4, 32, 32, sc=0.05), R(32, sc=0.1), R(32, 2, sc=0.2), torch.tensor([0.1, 0.7], device=dev), R(B, 2 * H, 2 * W, 1)
gw = torch.zeros(2048, dtype=torch.int32, device=dev); xw = word(x)
d.x_amax, d.dy_amax = xw.data_ptr(), gw.data_ptr()
rows, npart = C.c_int(0), C.c_int(0)
L.odin_gaussian_tail_fwd_bwd(None, None, None, None, None, None, None, None, None, C.byref(npart), None, C.byref(rows), None, C.byref(d), 1, 1, None)
lg, gg = torch.empty(B, 2 * H, 2 * W, 2, device=dev), torch.empty(B, 2 * H, 2 * W, 32, device=dev)
llk, sl, sc = torch.empty(B * npart.value, device=dev), torch.empty(rows.value, 98, device=dev), torch.tensor([1.0 / B], device=dev)
total += repeat('gaussian tail 48x40', lambda: L.odin_gaussian_tail_fwd_bwd(
    x.data_ptr(), wt.data_ptr(), b.data_ptr(), w1.data_ptr(), b1.data_ptr(), t.data_ptr(), lg.data_ptr(), gg.data_ptr(), llk.data_ptr(),
    C.byref(npart), sl.data_ptr(), C.byref(rows), sc.data_ptr(), C.byref(d), 1, 1, st), [lg, gg, llk, sl])
print('TOTAL differing launches:', total)
