#!/usr/bin/env python3
"""stand-alone times of the thin Dense kernels (thin_dense.hip) at FactorVAE's shapes"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')
def t(fn, n=200):
  for _ in range(20): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for B, K, N in ((128, 6, 1000), (256, 6, 1000), (128, 1000, 1), (256, 1000, 1)):
  x = torch.randn(B, K, device=dev); w = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
  y = torch.empty(B, N, device=dev); dy = torch.randn(B, N, device=dev); dx = torch.empty(B, K, device=dev)
  rows = C.c_int(0)
  L.odin_dense_wgrad(None, None, None, C.byref(rows), B, K, N, None)
  slab = torch.empty(max(rows.value, 1), K * N + N, device=dev)
  st = torch.cuda.current_stream().cuda_stream
  f = t(lambda: L.odin_dense_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, K, N, 2, st))
  p1 = L.odin_debug_last_path().decode()
  d = t(lambda: L.odin_dense_dgrad(dy.data_ptr(), w.data_ptr(), x.data_ptr(), 2, dx.data_ptr(), None, None, B, K, N, st))
  g = t(lambda: L.odin_dense_wgrad(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), B, K, N, st))
  print(f'B={B} K={K} N={N}: fwd {f:.2f} us  dgrad {d:.2f} us  wgrad {g:.2f} us (rows {rows.value})  [{p1}]')
