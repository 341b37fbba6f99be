#!/usr/bin/env python3
"""Stand-alone timings of the CelebA stack's layers that the plane kernels do not cover (batch 512):
encoder3 Conv2D(64, k4, s1) on 8x8x64, encoder_proj Dense(4096 -> 512), decoder1 Conv2DTranspose(64, k4, s1)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(); dev = torch.device('cuda:0')
def timed(fn, n=30):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
B = 512
# Dense 4096 -> 512
K, N = 4096, 512
x = torch.randn(B, K, device=dev); w = torch.randn(K, N, device=dev) * 0.02; b = torch.zeros(N, device=dev)
y = torch.empty(B, N, device=dev); dy = torch.randn(B, N, device=dev); dx = torch.empty(B, K, device=dev)
rows = C.c_int(0)
tf = timed(lambda: L.odin_dense_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, K, N, 0, None)); pf = L.odin_debug_last_path().decode()
td = timed(lambda: L.odin_dense_dgrad(dy.data_ptr(), w.data_ptr(), None, 0, dx.data_ptr(), None, C.byref(rows), B, K, N, None)); pd = L.odin_debug_last_path().decode()
L.odin_dense_wgrad(None, None, None, C.byref(rows), B, K, N, None)
slab = torch.empty(rows.value, K * N + N, device=dev)
tw = timed(lambda: L.odin_dense_wgrad(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), B, K, N, None)); pw = L.odin_debug_last_path().decode()
print(f'dense 4096->512 B512 (2.15 GF): fwd {tf:.1f} [{pf}] dgrad {td:.1f} [{pd}] wgrad {tw:.1f} [{pw}] rows {rows.value}')
ref = x @ w
print('  fwd max err', float((y - ref).abs().max() / ref.abs().max()))
# Conv 64 -> 64 k4 s1 on 8x8
d = _lib.conv_desc(B, 8, 8, 64, 8, 8, 64, 4, 1, 1, 1, 'elu')
xi = torch.randn(B, 8, 8, 64, device=dev); wc = torch.randn(4, 4, 64, 64, device=dev) * 0.05; bc = torch.zeros(64, device=dev)
yo = torch.empty(B, 8, 8, 64, device=dev); dyo = torch.randn(B, 8, 8, 64, device=dev); dxi = torch.empty(B, 8, 8, 64, device=dev)
tf = timed(lambda: L.odin_conv2d_fwd(xi.data_ptr(), wc.data_ptr(), bc.data_ptr(), yo.data_ptr(), C.byref(d), None)); pf = L.odin_debug_last_path().decode()
td = timed(lambda: L.odin_conv2d_dgrad(dyo.data_ptr(), wc.data_ptr(), xi.data_ptr(), 1, dxi.data_ptr(), None, C.byref(rows), C.byref(d), None)); pd = L.odin_debug_last_path().decode()
L.odin_conv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
slab = torch.empty(rows.value, 16 * 64 * 64 + 64, device=dev)
tw = timed(lambda: L.odin_conv2d_wgrad(xi.data_ptr(), dyo.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)); pw = L.odin_debug_last_path().decode()
print(f'conv 8x8x64->8x8x64 k4 s1 B512 (4.3 GF): fwd {tf:.1f} [{pf}] dgrad {td:.1f} [{pd}] wgrad {tw:.1f} [{pw}] rows {rows.value}')
