#!/usr/bin/env python3
"""Diagnostic: in-kernel stamps (workgroup 0; consumer wave 0 and producer wave 4) of the
transposed rolling-window kernel (tconv_ring.hip): fused decoder tail and the conv data gradient."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
names = {1: 'start', 2: 'setup', 10: 'C:barrier wait', 12: 'mfma 0 issued', 13: 'mfma 11', 14: 'mfma 23', 15: 'mfma 35', 16: 'mfma 47', 11: 'C:group / tile (mfma + prev epilogue)', 12: 'C:tile end',
         20: 'P:after-barrier', 21: 'P:issued', 22: 'P:landed'}


def timed(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def stamps(fn, title):
  us = timed(fn)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  for it in range(2):
    st.zero_()
    L.odin_debug_set_stamps(st.data_ptr())
    fn()
    torch.cuda.synchronize()
  L.odin_debug_set_stamps(None)
  va = st.cpu().numpy()
  print(f'--- {title}: {us:.1f} us')
  for half in (va[:32], va[32:]):
    v = half[half != 0]
    if len(v) == 0: continue
    ks, ts = (v >> 56), (v & ((1 << 56) - 1))
    print(f'   [{len(ks)} stamps, span {ts[-1]-ts[0]} ticks; steady period {(ts[-1]-ts[4])/((len(ks)-5)/2):.0f} ticks per tile]')
    for i in range(1, min(len(ks), 24)):
      print(f'   {names[int(ks[i])]:32s} +{ts[i]-ts[i-1]}')


def tail(B, H, W, C1):
  d = _lib.conv_desc(B, H, W, 32, 2 * H, 2 * W, 32, 4, 2, 1, 1, 'elu')
  x = torch.randn(B, H, W, 32, device=dev); w = torch.randn(4, 4, 32, 32, device=dev) * 0.1
  b = torch.randn(32, device=dev) * 0.1; w1 = torch.randn(32, C1, device=dev) * 0.3; b1 = torch.randn(C1, device=dev)
  tgt = torch.rand(B, 2 * H, 2 * W, C1, device=dev); sc = torch.tensor([1.0 / B], device=dev)
  lg = torch.empty(B, 2 * H, 2 * W, C1, device=dev); g = torch.empty(B, 2 * H, 2 * W, 32, device=dev)
  rows, npart = C.c_int(0), C.c_int(0)
  L.odin_bernoulli_tail_fwd_bwd(1, None, None, None, None, None, None, None, None, None, C.byref(npart),
                                None, C.byref(rows), None, C.byref(d), C1, None)
  part = torch.empty(B * npart.value, device=dev); slab = torch.empty(rows.value, 32 * C1 + C1 + 32, device=dev)
  fn = lambda: L.odin_bernoulli_tail_fwd_bwd(1, x.data_ptr(), w.data_ptr(), b.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                             tgt.data_ptr(), lg.data_ptr(), g.data_ptr(), part.data_ptr(), C.byref(npart),
                                             slab.data_ptr(), C.byref(rows), sc.data_ptr(), C.byref(d), C1, None)
  stamps(fn, f'fused tail B{B} {H}x{W}x32 -> {2*H}x{2*W}x32 -> {C1} maps')


def dgrad(B, H, W):
  """Conv2D (H,W,32) -> (H/2,W/2,32), k4 s2: data gradient reads dy, writes dx [B,H,W,32]"""
  d = _lib.conv_desc(B, H, W, 32, H // 2, W // 2, 32, 4, 2, 1, 1, 'elu')
  w = torch.randn(4, 4, 32, 32, device=dev) * 0.1
  dy = torch.randn(B, H // 2, W // 2, 32, device=dev); aux = torch.randn(B, H, W, 32, device=dev)
  dx = torch.empty(B, H, W, 32, device=dev); slab = torch.empty(L.odin_max_slab_rows(), 32, device=dev)
  rows = C.c_int(0)
  fn = lambda: L.odin_conv2d_dgrad(dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), slab.data_ptr(),
                                   C.byref(rows), C.byref(d), None)
  stamps(fn, f'conv dgrad B{B} {H}x{W}x32 <- {H//2}x{W//2}x32')


tail(256, 32, 32, 1)
import os
if os.environ.get("ODIN_TP_DBG"): sys.exit(0)
dgrad(256, 32, 32)
