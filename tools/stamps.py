#!/usr/bin/env python3
"""Diagnostic: in-kernel cycle stamps of one conv launch (workgroup 0, wave 0)."""
import ctypes as C, sys, torch, numpy as np
sys.path.insert(0, '.')
from odin_ai_amd import _lib
from odin_ai_amd.engine import same_pads
L = _lib.load()
dev = torch.device('cuda:0')
names = {1: 'start', 2: 'weights', 3: 'prefetch0', 4: 'tile', 5: 'commit+sync', 6: 'issue-next', 7: 'mfma', 8: 'end'}
def run(kind, B, H, W, Ci, Co, K, S, act='elu'):
  if kind == 'deconv':
    OH, OW = H * S, W * S
    _, pt, _ = same_pads(OH, K, S); _, pl, _ = same_pads(OW, K, S)
    w = torch.randn(K, K, Co, Ci, device=dev) * 0.1
  else:
    OH, pt, _ = same_pads(H, K, S); OW, pl, _ = same_pads(W, K, S)
    w = torch.randn(K, K, Ci, Co, device=dev) * 0.1
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, act)
  x = torch.randn(B, H, W, Ci, device=dev); b = torch.zeros(Co, device=dev)
  y = torch.empty(B, OH, OW, Co, device=dev)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  fn = L.odin_deconv2d_fwd if kind == 'deconv' else L.odin_conv2d_fwd
  for it in range(3):
    st.zero_()
    L.odin_debug_set_stamps(st.data_ptr())
    fn(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), None)
    torch.cuda.synchronize()
  L.odin_debug_set_stamps(None)
  v = st.cpu().numpy()
  v = v[v != 0]
  ks, ts = (v >> 56), (v & ((1 << 56) - 1))
  print(f'--- {kind} B{B} {H}x{W}x{Ci}->{Co} k{K}s{S}: total {ts[-1]-ts[0]} cycles')
  for i in range(1, min(len(ks), 30)):
    print(f'   {names[int(ks[i])]:12s} +{ts[i]-ts[i-1]}')
run('deconv', 256, 32, 32, 32, 32, 4, 2)


def run_tail(B, H, W, Ci, Co, K, S, C1=1):
  """fused decoder tail: deconv(Co, elu) -> conv1x1(C1) -> Bernoulli fwd+bwd"""
  OH, OW = H * S, W * S
  _, pt, _ = same_pads(OH, K, S); _, pl, _ = same_pads(OW, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  x = torch.randn(B, H, W, Ci, device=dev); w = torch.randn(K, K, Co, Ci, device=dev) * 0.1
  b = torch.zeros(Co, device=dev); w1 = torch.randn(Co, C1, device=dev) * 0.1; b1 = torch.zeros(C1, device=dev)
  tgt = torch.rand(B, OH, OW, C1, device=dev); g = torch.empty(B, OH, OW, Co, device=dev)
  npart, rows = C.c_int(0), C.c_int(0)
  L.odin_bernoulli_tail_fwd_bwd(1, None, None, None, None, None, None, None, None, None, C.byref(npart),
                                None, C.byref(rows), None, C.byref(d), C1, None)
  llk = torch.empty(B * npart.value, device=dev); slab = torch.empty(rows.value, Co * C1 + C1 + Co, device=dev)
  scale = torch.full((1,), 1.0 / B, device=dev)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  for it in range(3):
    st.zero_()
    L.odin_debug_set_stamps(st.data_ptr())
    L.odin_bernoulli_tail_fwd_bwd(1, x.data_ptr(), w.data_ptr(), b.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                  tgt.data_ptr(), None, g.data_ptr(), llk.data_ptr(), C.byref(npart),
                                  slab.data_ptr(), C.byref(rows), scale.data_ptr(), C.byref(d), C1, None)
    torch.cuda.synchronize()
  L.odin_debug_set_stamps(None)
  v = st.cpu().numpy(); v = v[v != 0]
  ks, ts = (v >> 56), (v & ((1 << 56) - 1))
  print(f'--- fused tail B{B} {H}x{W}x{Ci}->{Co}->{C1}: total {ts[-1]-ts[0]} cycles, rows {rows.value}')
  for i in range(1, min(len(ks), 30)):
    print(f'   {names[int(ks[i])]:12s} +{ts[i]-ts[i-1]}')
run_tail(256, 32, 32, 32, 32, 4, 2)
run('conv', 256, 32, 32, 32, 32, 4, 2)
run('conv', 256, 1, 1, 128, 20, 1, 1, 'linear')


def run_w(B, H, W, Ci, Co, K, S):
  """deconv wgrad: x [B,H,W,Ci], dy [B,H*S,W*S,Co]"""
  OH, OW = H * S, W * S
  _, pt, _ = same_pads(OH, K, S); _, pl, _ = same_pads(OW, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  x = torch.randn(B, H, W, Ci, device=dev); g = torch.randn(B, OH, OW, Co, device=dev)
  rows = C.c_int(0)
  L.odin_deconv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
  slab = torch.empty(rows.value, K * K * Co * Ci, device=dev)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  for it in range(3):
    st.zero_()
    L.odin_debug_set_wgrad_stamps(st.data_ptr())
    L.odin_deconv2d_wgrad(x.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
    torch.cuda.synchronize()
  L.odin_debug_set_wgrad_stamps(None)
  va = st.cpu().numpy()
  nm = {1: 'start', 3: 'prefetch0', 4: 'tile(mfma+top)', 5: 'commit+sync', 6: 'issue-next', 8: 'end',
        10: 'C:phase-start', 11: 'C:mfma-done', 20: 'P:phase-start', 21: 'P:loads-issued', 22: 'P:committed', 23: 'P:data-arrived'}
  t0 = None
  for half, lab in ((va[:32], 'consumer wave 0 / single-role'), (va[32:], 'producer wave 4')):
    v = half[half != 0]
    if len(v) == 0: continue
    ks, ts = (v >> 56), (v & ((1 << 56) - 1))
    t0 = ts[0] if t0 is None else t0
    print(f'--- deconv wgrad B{B} {H}x{W}x{Ci}->{Co} [{lab}]: total {ts[-1]-ts[0]} cycles, rows {rows.value}')
    for i in range(0, min(len(ks), 24)):
      print(f'   {nm[int(ks[i])]:16s} @{ts[i]-t0:8d}  +{ts[i]-ts[i-1] if i else 0}')
run_w(256, 32, 32, 32, 32, 4, 2)


def run_wd(B, K, N):
  x = torch.randn(B, K, device=dev); g = torch.randn(B, N, device=dev)
  rows = C.c_int(0)
  L.odin_dense_wgrad(None, None, None, C.byref(rows), B, K, N, None)
  slab = torch.empty(rows.value, K * N + N, device=dev)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  for it in range(3):
    st.zero_()
    L.odin_debug_set_wgrad_stamps(st.data_ptr())
    L.odin_dense_wgrad(x.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(rows), B, K, N, None)
    torch.cuda.synchronize()
  L.odin_debug_set_wgrad_stamps(None)
  v = st.cpu().numpy(); v = v[v != 0]
  ks, ts = (v >> 56), (v & ((1 << 56) - 1))
  nm = {1: 'start', 3: 'prefetch0', 4: 'tile(mfma+top)', 5: 'commit+sync', 6: 'issue-next', 8: 'end'}
  print(f'--- dense wgrad B{B} {K}->{N}: total {ts[-1]-ts[0]} cycles, rows {rows.value}')
  for i in range(1, min(len(ks), 24)):
    print(f'   {nm[int(ks[i])]:14s} +{ts[i]-ts[i-1]}')
run_wd(256, 10, 128)
run_wd(256, 1024, 128)
