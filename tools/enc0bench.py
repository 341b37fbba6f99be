import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(); dev = torch.device('cuda:0')
def timed(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for Ci in (1, 3):
  B, H, W, Co = 256, 64, 64, 32
  d = _lib.conv_desc(B, H, W, Ci, 32, 32, Co, 4, 2, 1, 1, 'elu', True)
  x = torch.rand(B, H, W, Ci, device=dev); w = torch.randn(4, 4, Ci, Co, device=dev) * 0.1
  b = torch.randn(Co, device=dev) * 0.1; y = torch.empty(B, 32, 32, Co, device=dev)
  dy = torch.randn(B, 32, 32, Co, device=dev); rows = C.c_int(0)
  t = timed(lambda: L.odin_conv2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), None))
  L.odin_conv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
  slab = torch.empty(rows.value, 16 * Ci * Co + Co, device=dev)
  tw = timed(lambda: L.odin_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None))
  print(f'Cin={Ci}: fwd {t:.1f} us  wgrad {tw:.1f} us rows {rows.value}')
