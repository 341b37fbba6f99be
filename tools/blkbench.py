#!/usr/bin/env python3
"""stand-alone times of the 4x4 / stride-2 layers of the audio VAE (batch 256; speech_networks) with the block-window
plane kernels (blk_planes.hip) on and off: every role of every layer, which kernel family ran, GB/s of the
algorithmic traffic.  usage: tools/blkbench.py [batch]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odin_ai_amd import _lib
from oracle import vae_oracle as vo
L = _lib.load()
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def t(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def word(tn):
  w = torch.zeros(2048, dtype=torch.int32, device=dev)
  L.odin_absmax(tn.data_ptr(), tn.numel(), w.data_ptr(), None)
  return w


LAYERS = [  # kind, H, W (layer input), Cin, Cout
    ('conv', 48, 40, 32, 32), ('conv', 24, 20, 32, 64), ('conv', 12, 10, 64, 64),
    ('deconv', 12, 10, 64, 64), ('deconv', 24, 20, 64, 32), ('deconv', 48, 40, 32, 32)]
if os.environ.get('BLK_CELEBA'):
  # CelebA (batch 512): the 8 x 8 -> 4 x 4 encoder layer and the 4 x 4 -> 8 x 8 decoder layer
  B = 512
  LAYERS = [('conv', 8, 8, 64, 64), ('deconv', 4, 4, 64, 64)]
if os.environ.get('BLK_FIRST'):
  # the 64 x 64 image stacks' mid layers: block-window kernels (blk=1) against the row-window plane kernels (blk=0)
  L.odin_debug_blk_first(1)
  LAYERS = [('conv', 32, 32, 32, 32), ('conv', 16, 16, 32, 64), ('deconv', 8, 8, 64, 64), ('deconv', 16, 16, 64, 32),
            ('deconv', 32, 32, 32, 32)]
st = torch.cuda.current_stream().cuda_stream
if os.environ.get('BLK_MIN_FLOP'):
  L.odin_debug_blk_min_flop(float(os.environ['BLK_MIN_FLOP']))
for kind, H, W, Ci, Co in LAYERS:
  if kind == 'conv':
    OH, OW = H // 2, W // 2
    wt = torch.randn(4, 4, Ci, Co, device=dev) * 0.05
  else:
    OH, OW = 2 * H, 2 * W
    wt = torch.randn(4, 4, Co, Ci, device=dev) * 0.05
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, 4, 2, 1, 1, 'elu')
  x = torch.randn(B, H, W, Ci, device=dev); b = torch.randn(Co, device=dev) * 0.1
  y = torch.empty(B, OH, OW, Co, device=dev); dy = torch.randn(B, OH, OW, Co, device=dev) * 1e-3
  dx = torch.empty_like(x); aux = torch.randn_like(x)
  xw, yw, dyw, dxw = word(x), torch.zeros(2048, dtype=torch.int32, device=dev), word(dy), torch.zeros(2048, dtype=torch.int32, device=dev)
  d.x_amax, d.y_amax, d.dy_amax, d.dx_amax = xw.data_ptr(), yw.data_ptr(), dyw.data_ptr(), dxw.data_ptr()
  n = 16 * Ci * Co + (Co if kind == 'conv' else 0)
  slab = torch.empty(L.odin_max_slab_rows(), n, device=dev)
  cs = torch.empty(L.odin_max_slab_rows(), Ci, device=dev)
  rows = C.c_int(0)
  fwd = getattr(L, f'odin_{kind}2d_fwd'); dg = getattr(L, f'odin_{kind}2d_dgrad'); wg = getattr(L, f'odin_{kind}2d_wgrad')
  bw = getattr(L, f'odin_{kind}2d_bwd')
  gf = 2.0 * B * max(H * W, OH * OW) / (4 if kind == 'conv' else 1) * 16 * Ci * Co / 1e9
  gf = 2.0 * B * (OH * OW if kind == 'conv' else H * W) * 16 * Ci * Co / 1e9
  mb_f = (x.numel() + y.numel()) * 4 / 1e6
  for on in (1, 0):
    L.odin_debug_blk_planes(on)
    res = []
    tf = t(lambda: fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), st)); pf = L.odin_debug_last_path().decode()
    td = t(lambda: dg(dy.data_ptr(), wt.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), cs.data_ptr(), C.byref(rows), C.byref(d), st)); pd = L.odin_debug_last_path().decode()
    tw = t(lambda: wg(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), st)); pw = L.odin_debug_last_path().decode()
    r2 = C.c_int(0)
    tb = t(lambda: bw(x.data_ptr(), dy.data_ptr(), wt.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), cs.data_ptr(), C.byref(r2), slab.data_ptr(), C.byref(rows), C.byref(d), st)); pb = L.odin_debug_last_path().decode()
    print(f'{kind} {H}x{W} {Ci}->{Co} ({gf:.1f} GF, {mb_f:.0f} MB) blk={on}: fwd {tf:6.1f} us {mb_f / tf * 1e-3:5.2f} TB/s [{pf}]  dgrad {td:6.1f} [{pd}]  '
          f'wgrad {tw:6.1f} [{pw}] rows {rows.value}  bwd {tb:6.1f} [{pb}]', flush=True)
L.odin_debug_blk_planes(1)
