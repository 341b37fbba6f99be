#!/usr/bin/env python3
"""Diagnostics: dec4 weight gradient (wgrad_planes, 32-pixel coarse rows) with parts switched off (ODIN_WP_DBG)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
B, H, W = 256, 32, 32
d = _lib.conv_desc(B, H, W, 32, 2 * H, 2 * W, 32, 4, 2, 1, 1, 'elu')
x = torch.randn(B, H, W, 32, device=dev); dy = torch.randn(B, 2 * H, 2 * W, 32, device=dev)
rows = C.c_int(0)
L.odin_deconv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
slab = torch.empty(rows.value, 16 * 32 * 32, device=dev)
fn = lambda: L.odin_deconv2d_wgrad(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
for dbg in ('0', '1', '2', '4', '8', '16'):
  os.putenv('ODIN_WP_DBG', dbg)
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): fn()
  e1.record(); torch.cuda.synchronize()
  print(f'ODIN_WP_DBG={dbg}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  (1 no MFMA, 2 no LDS reads, 4 no fills, 8 no global loads, 16 loads but no split / LDS stores)')
