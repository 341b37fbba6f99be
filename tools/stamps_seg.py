#!/usr/bin/env python3
"""Diagnostic (needs `make -C odin_ai_amd/csrc diag` and ODIN_DIAG_LIB=tools/diag/libodin_hip_diag.so):
per-segment s_memtime deltas inside one tile of tconv_planes' fused tail (waves 0 and 4 of workgroup 0)
and the shader clock the chip holds (s_memtime / s_memrealtime)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB') or None)
dev = torch.device('cuda:0')
names = {1: 'start', 2: 'setup done', 10: 'barrier released', 11: 'tile done (at barrier)', 12: 'mfma 0 issued',
         13: 'mfma 11', 14: 'mfma 23', 15: 'mfma 35', 16: 'mfma 47'}


def timed(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def run(C1, B=256, H=32, W=32):
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
  us = timed(fn, 200)
  st = torch.zeros(72, dtype=torch.int64, device=dev)
  for it in range(2):
    st.zero_()
    L.odin_debug_set_stamps(st.data_ptr())
    fn()
    torch.cuda.synchronize()
  L.odin_debug_set_stamps(None)
  va = st.cpu().numpy()
  print(f'--- fused tail C1={C1} B{B} {H}x{W}: {us:.1f} us (dbg={os.environ.get("ODIN_TP_DBG", "0")})')
  if va[67] > va[65]:
    ghz = (va[66] - va[64]) / ((va[67] - va[65]) * 10.0)
    print(f'   workgroup 0 body: {va[66]-va[64]} shader ticks in {(va[67]-va[65])/100.0:.2f} us -> {ghz:.3f} GHz')
  for wv, half in ((0, va[:32]), (4, va[32:64])):
    v = half[half != 0]
    if len(v) == 0: continue
    ks, ts = (v >> 56), (v & ((1 << 56) - 1))
    out = [f'{names.get(int(ks[i]), int(ks[i]))} +{ts[i]-ts[i-1]}' for i in range(1, len(ks))]
    print(f'   wave {wv}: ' + ' | '.join(out))


for c1 in (1, 3):
  run(c1)
