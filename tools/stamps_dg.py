#!/usr/bin/env python3
"""Diagnostic: in-kernel stamps (workgroup 0, wave 0) of the deconv data-gradient launches."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
from odin_ai_amd.engine import same_pads
L = _lib.load()
dev = torch.device('cuda:0')
names = {1: 'start', 2: 'weights', 3: 'prefetch0', 4: 'tile(epilogue)', 5: 'commit+sync', 6: 'issue-next', 7: 'mfma', 8: 'end',
         10: 'C:tile-start(barrier wait)', 11: 'C:mfma', 12: 'C:epilogue', 20: 'P:after-barrier', 21: 'P:issued', 22: 'P:landed'}


def run(B, H, W, Ci, Co, K, S):
  """deconv (H,W,Ci)->(H*S,W*S,Co): dgrad reads dy [B,OH,OW,Co], writes dx [B,H,W,Ci]"""
  OH, OW = H * S, W * S
  _, pt, _ = same_pads(OH, K, S); _, pl, _ = same_pads(OW, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  w = torch.randn(K, K, Co, Ci, device=dev) * 0.1
  dy = torch.randn(B, OH, OW, Co, device=dev); aux = torch.randn(B, H, W, Ci, device=dev)
  dx = torch.empty(B, H, W, Ci, device=dev)
  slab = torch.empty(L.odin_max_slab_rows(), Ci, device=dev)
  rows = C.c_int(0)
  fn = lambda: L.odin_deconv2d_dgrad(dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(),
                                     slab.data_ptr(), C.byref(rows), C.byref(d), None)
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): fn()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 20 * 1e3
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  for it in range(2):
    st.zero_()
    L.odin_debug_set_stamps(st.data_ptr())
    fn()
    torch.cuda.synchronize()
  L.odin_debug_set_stamps(None)
  va = st.cpu().numpy()
  print(f'--- deconv dgrad B{B} out {H}x{W}x{Ci} <- {OH}x{OW}x{Co}: {us:.1f} us, rows {rows.value}')
  for half in (va[:32], va[32:]):
    v = half[half != 0]
    if len(v) == 0: continue
    ks, ts = (v >> 56), (v & ((1 << 56) - 1))
    print(f'   [{len(ks)} stamps, span {ts[-1]-ts[0]} ticks]')
    for i in range(1, min(len(ks), 16)):
      print(f'   {names[int(ks[i])]:26s} +{ts[i]-ts[i-1]}')


run(256, 32, 32, 32, 32, 4, 2)   # dec4 dgrad
run(256, 16, 16, 64, 32, 4, 2)   # dec3 dgrad
