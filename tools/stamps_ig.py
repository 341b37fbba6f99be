#!/usr/bin/env python3
"""Diagnostic (diag build: make -C odin_ai_amd/csrc diag): clock stamps of workgroup (0,0) wave 0 of igemm_kernel:
0 entry, 1 set-up done, 2 first batch issued, 3 reduction loop done, 4 barrier passed, 5 partial tiles summed, 6 end."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB', 'tools/diag/libodin_hip_diag.so'))
dev = torch.device('cuda:0')
def same_pads(n, k, s):
  out = -(-n // s); total = max((out - 1) * s + k - n, 0); return out, total // 2
for kind, B, H, W, Ci, Co, K, S in [('conv', 256, 8, 8, 64, 64, 4, 2), ('deconv', 256, 4, 4, 8, 64, 4, 2)]:
  if kind == 'conv':
    OH, pt = same_pads(H, K, S); OW, pl = same_pads(W, K, S); wshape = (K, K, Ci, Co)
  else:
    OH, OW = H * S, W * S; _, pt = same_pads(OH, K, S); _, pl = same_pads(OW, K, S); wshape = (K, K, Co, Ci)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  x = torch.randn(B, H, W, Ci, device=dev); w = torch.randn(*wshape, device=dev) * 0.1
  b = torch.randn(Co, device=dev) * 0.1; y = torch.empty(B, OH, OW, Co, device=dev)
  dy = torch.randn(B, OH, OW, Co, device=dev); dx = torch.empty(B, H, W, Ci, device=dev)
  aux = torch.randn(B, H, W, Ci, device=dev); bs = torch.empty(L.odin_max_slab_rows(), Ci, device=dev)
  rows = C.c_int(0)
  st = torch.zeros(64, dtype=torch.int64, device=dev)
  for name, fn in [('fwd', lambda: getattr(L, f'odin_{kind}2d_fwd')(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), None)),
                   ('dgrad', lambda: getattr(L, f'odin_{kind}2d_dgrad')(dy.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), bs.data_ptr(), C.byref(rows), C.byref(d), None))]:
    for rep in range(3):
      L.odin_debug_set_stamps(st.data_ptr()); st.zero_(); torch.cuda.synchronize()
      fn(); torch.cuda.synchronize()
      full = st.cpu().numpy(); t = full[:7]
      print(kind, name, 'deltas (ticks):', [int(t[i + 1] - t[i]) for i in range(6)], 'total', int(t[6] - t[0]), 'wall(100MHz ticks)', int(full[9] - full[8]))
    L.odin_debug_set_stamps(None)
