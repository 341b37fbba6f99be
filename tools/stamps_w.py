#!/usr/bin/env python3
"""Diagnostic: in-kernel stamps of the small weight-gradient launches (workgroup 0, wave 0)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
from odin_ai_amd.engine import same_pads
L = _lib.load()
dev = torch.device('cuda:0')
nm = {1: 'start', 3: 'prefetch0', 4: 'tile(mfma+top)', 5: 'commit+sync', 6: 'issue-next', 8: 'end(loop)', 9: 'written'}


def run(kind, B, H, W, Ci, Co, K, S):
  if kind == 'deconv':
    OH, OW = H * S, W * S
    _, pt, _ = same_pads(OH, K, S); _, pl, _ = same_pads(OW, K, S)
    n = K * K * Co * Ci
    fn = L.odin_deconv2d_wgrad
  else:
    OH, pt, _ = same_pads(H, K, S); OW, pl, _ = same_pads(W, K, S)
    n = K * K * Ci * Co + Co
    fn = L.odin_conv2d_wgrad
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  x = torch.randn(B, H, W, Ci, device=dev); g = torch.randn(B, OH, OW, Co, device=dev)
  rows = C.c_int(0)
  fn(None, None, None, C.byref(rows), C.byref(d), None)
  slab = torch.empty(rows.value, n, device=dev)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  for it in range(3):
    fn(x.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  torch.cuda.synchronize(); e0.record()
  for it in range(20):
    fn(x.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 20 * 1e3
  for it in range(2):
    st.zero_()
    L.odin_debug_set_wgrad_stamps(st.data_ptr())
    fn(x.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
    torch.cuda.synchronize()
  L.odin_debug_set_wgrad_stamps(None)
  v = st.cpu().numpy()[:32]; v = v[v != 0]
  ks, ts = (v >> 56), (v & ((1 << 56) - 1))
  print(f'--- {kind} wgrad B{B} {H}x{W}x{Ci}->{Co} k{K}s{S}: {us:.1f} us, rows {rows.value}, wg0 total {ts[-1]-ts[0]} ticks')
  for i in range(1, min(len(ks), 24)):
    print(f'   {nm.get(int(ks[i]), str(int(ks[i]))):14s} +{ts[i]-ts[i-1]}')


run('conv', 256, 8, 8, 64, 64, 4, 2)     # enc3
run('deconv', 256, 4, 4, 8, 64, 4, 2)    # dec1
run('conv', 256, 16, 16, 32, 64, 4, 2)   # enc2
run('deconv', 256, 32, 32, 32, 32, 4, 2)  # dec4 (producer/consumer kernel)
