#!/usr/bin/env python3
"""stand-alone times of a Dense layer's backward pass (weight + data gradient in one call, and the weight gradient
alone) with the LDS-staged weight gradient (dense_h.hip: dense_hw) on and off.  usage: tools/densebench.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odin_ai_amd import _lib
L = _lib.load()
dev = torch.device('cuda:0')


def t(fn, n=100):
  for _ in range(10): fn()
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


st = torch.cuda.current_stream().cuda_stream
# FactorVAE's discriminator (factor_vae.py:150-153) at 128 / 256 rows, CelebA's projection (image_networks.py:688), the
# 512-unit default nets at batch 128
for B, K, N in [(256, 1000, 1000), (128, 1000, 1000), (512, 4096, 512), (512, 1024, 1024), (128, 784, 512), (128, 512, 512)]:
  x = torch.randn(B, K, device=dev); dy = torch.randn(B, N, device=dev) * 1e-3
  w = torch.randn(K, N, device=dev) / K ** 0.5; aux = torch.randn(B, K, device=dev)
  dx = torch.empty(B, K, device=dev); slab = torch.empty(1, K * N + N, device=dev)
  xw, dyw, dxw = word(x), word(dy), torch.zeros(2048, dtype=torch.int32, device=dev)
  rows = C.c_int(0)
  for tiles in (128, 1 << 30):
    L.odin_debug_dense_hw_min_tiles(tiles)
    pair = lambda: L.odin_dense_bwd_ranged(x.data_ptr(), dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), None, None,
                                           slab.data_ptr(), C.byref(rows), B, K, N, 1, 1, dyw.data_ptr(), dxw.data_ptr(), xw.data_ptr(), st)
    wg = lambda: L.odin_dense_bwd_ranged(x.data_ptr(), dy.data_ptr(), None, None, 0, None, None, None, slab.data_ptr(),
                                         C.byref(rows), B, K, N, 1, 0, dyw.data_ptr(), None, xw.data_ptr(), st)
    dg = lambda: L.odin_dense_bwd_ranged(None, dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), None, None, None,
                                         None, B, K, N, 0, 1, dyw.data_ptr(), dxw.data_ptr(), None, st)
    y = torch.empty(B, N, device=dev); bias = torch.zeros(N, device=dev); yw = torch.zeros(2048, dtype=torch.int32, device=dev)
    fw = lambda: L.odin_dense_fwd_ranged(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, K, N, 1, xw.data_ptr(), yw.data_ptr(), st)
    tf = t(fw); pf = L.odin_debug_last_path().decode()
    tp = t(pair); pp = L.odin_debug_last_path().decode()
    tw = t(wg); pw = L.odin_debug_last_path().decode()
    td = t(dg); pd = L.odin_debug_last_path().decode()
    gf = 2.0 * B * K * N / 1e9
    print(f'[{B} x {K} x {N}] ({gf:.2f} GF per half) min_tiles={tiles}: pair {tp:6.1f} us [{pp}]  wgrad {tw:6.1f} us '
          f'{gf / tw * 1e3:6.1f} TF/s [{pw}]  dgrad {td:6.1f} us [{pd}]  fwd {tf:6.1f} us [{pf}]', flush=True)
L.odin_debug_dense_hw_min_tiles(128)
