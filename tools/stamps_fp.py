#!/usr/bin/env python3
"""Diagnostic (diagnostics build, `make -C odin_ai_amd/csrc diag`): in-kernel stamps of fconv_planes.hip, workgroup 0,
all eight waves -- per tile: start (behind the barrier), first MFMA issued (its fragments have arrived), last MFMA issued,
partial tile written (in front of the barrier)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB') or 'tools/diag/libodin_hip_diag.so')
dev = torch.device('cuda:0')
names = {1: 'kernel start', 2: 'tile start', 3: 'mfma 0 issued', 4: 'mfma 11 issued', 5: 'partials written',
         6: 'zero fills + tables done', 7: 'behind the prologue barrier', 8: 'weights split', 9: 'first rows stored'}
B, H, W = 256, 32, 32
d = _lib.conv_desc(B, H, W, 32, 2 * H, 2 * W, 32, 4, 2, 1, 1, 'elu')
w = torch.randn(4, 4, 32, 32, device=dev) * 0.1
g = torch.randn(B, 2 * H, 2 * W, 32, device=dev)
dx = torch.empty(B, H, W, 32, device=dev); aux = torch.randn(B, H, W, 32, device=dev)
bs = torch.empty(L.odin_max_slab_rows(), 32, device=dev)
rows = C.c_int(0)
words = torch.zeros(2048, dtype=torch.int32, device=dev)   # range word of g (odin_conv_desc.dy_amax)
L.odin_absmax(g.data_ptr(), g.numel(), words.data_ptr(), None)
d.dy_amax = words.data_ptr()
fn = lambda: L.odin_deconv2d_dgrad(g.data_ptr(), w.data_ptr(), aux.data_ptr(), 1, dx.data_ptr(), bs.data_ptr(),
                                   C.byref(rows), C.byref(d), None)
for _ in range(5): fn()
torch.cuda.synchronize()
st = torch.zeros(8 * 32, dtype=torch.int64, device=dev)
for it in range(2):
  st.zero_()
  L.odin_debug_set_stamps(st.data_ptr())
  fn()
  torch.cuda.synchronize()
L.odin_debug_set_stamps(None)
print('path', L.odin_debug_last_path().decode())
va = st.cpu().numpy().reshape(8, 32)
t0 = min(int(va[w][0] & ((1 << 56) - 1)) for w in range(8))
print('workgroup 0: shader cycles since its first wave started; rows = stamps, columns = waves 0..7')
for i in range(32):
  k = int(va[0][i] >> 56)
  if k == 0: break
  print(f'{names[k]:18s}' + ''.join(f'{int(va[w][i] & ((1 << 56) - 1)) - t0:8d}' for w in range(8)))
