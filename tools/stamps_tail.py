#!/usr/bin/env python3
"""Diagnostic (diagnostics build, `make -C odin_ai_amd/csrc diag`): in-kernel stamps of the fused decoder tail
(tconv_planes.hip, EPI 3), workgroup 0, waves 0 and 4 (the two waves of SIMD 0): per tile the barrier exit, MFMA 0 / 5 /
11 / 17 / 23 issued, the barrier entry."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB') or 'tools/diag/libodin_hip_diag.so')
dev = torch.device('cuda:0')
names = {1: 'kernel start', 2: 'prologue stores done', 10: 'behind the barrier', 11: 'MFMAs done (before barrier)',
         12: 'mfma 0 issued', 13: 'mfma 5', 14: 'mfma 11', 15: 'mfma 17', 16: 'mfma 23'}
C1 = int(os.environ.get('C1', '1'))
B, H, W = 256, 32, 32
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
for _ in range(5): fn()
torch.cuda.synchronize()
st = torch.zeros(72, dtype=torch.int64, device=dev)
for it in range(2):
  st.zero_()
  L.odin_debug_set_stamps(st.data_ptr())
  fn()
  torch.cuda.synchronize()
L.odin_debug_set_stamps(None)
print('path', L.odin_debug_last_path().decode())
va = st.cpu().numpy()
M = (1 << 56) - 1
t0 = int(va[0] & M)
for wv, half in ((0, va[:32]), (4, va[32:64])):
  print(f'--- wave {wv}')
  prev = None
  for v in half:
    if v == 0: break
    k, t = int(v >> 56), int(v & M) - t0
    print(f'  {names.get(k, k):32s} {t:8d}' + ('' if prev is None else f'   +{t - prev}'))
    prev = t
print('kernel clocks', int(va[66] - va[64]), 'wall ticks', int(va[67] - va[65]))
