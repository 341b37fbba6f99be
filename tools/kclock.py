#!/usr/bin/env python3
"""In-kernel clock of the fused decoder tail (tconv_planes<3,1,32>) under sustained load, as MI355X_MICROARCH.md
prescribes ('DVFS give-back', item 6): after >= 2 s of back-to-back launches on random data one launch records
s_memtime and s_memrealtime (100 MHz) at the start and the end of workgroup 0; clock = d(memtime) / d(memrealtime)
* 100 MHz.  Diagnostics build only (make -C odin_ai_amd/csrc diag; the product kernel executes no stamp)."""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, '.')
from odin_ai_amd import _lib
L = _lib.load(os.environ.get('ODIN_DIAG_LIB', 'tools/diag/libodin_hip_diag.so'))
dev = torch.device('cuda:0')
B, H, W, C1 = 256, 32, 32, 1
d = _lib.conv_desc(B, H, W, 32, 2 * H, 2 * W, 32, 4, 2, 1, 1, 'elu')
x = torch.randn(B, H, W, 32, device=dev); w = torch.randn(4, 4, 32, 32, device=dev) * 0.1
b = torch.randn(32, device=dev) * 0.1; w1 = torch.randn(32, C1, device=dev) * 0.3; b1 = torch.randn(C1, device=dev)
tgt = torch.rand(B, 2 * H, 2 * W, C1, device=dev); sc = torch.tensor([1.0 / B], device=dev)
lg = torch.empty(B, 2 * H, 2 * W, C1, device=dev); g = torch.empty(B, 2 * H, 2 * W, 32, device=dev)
rows, npart = C.c_int(0), C.c_int(0)
L.odin_bernoulli_tail_fwd_bwd(1, None, None, None, None, None, None, None, None, None, C.byref(npart), None,
                              C.byref(rows), None, C.byref(d), C1, None)
part = torch.empty(B * npart.value, device=dev); slab = torch.empty(rows.value, 32 * C1 + C1 + 32, device=dev)
fn = lambda: L.odin_bernoulli_tail_fwd_bwd(1, x.data_ptr(), w.data_ptr(), b.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                                           tgt.data_ptr(), lg.data_ptr(), g.data_ptr(), part.data_ptr(), C.byref(npart),
                                           slab.data_ptr(), C.byref(rows), sc.data_ptr(), C.byref(d), C1, None)
st = torch.zeros(72, dtype=torch.int64, device=dev)
for rep in range(3):
  t0 = time.time()
  while time.time() - t0 < 2.0:
    for _ in range(200): fn()
    torch.cuda.synchronize()
  st.zero_()
  L.odin_debug_set_stamps(st.data_ptr())
  fn()
  torch.cuda.synchronize()
  L.odin_debug_set_stamps(None)
  v = st.cpu().numpy()
  dc, dw = int(v[66] - v[64]), int(v[67] - v[65])
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(50): fn()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 50 * 1e3
  print(f'workgroup 0: {dc} shader cycles in {dw * 10} ns -> in-kernel clock {dc / (dw * 10.0):.3f} GHz; launch {us:.1f} us; '
        f'MFMA-busy cycles per SIMD 49152 -> pipe busy {49152.0 / dc:.2f} of the kernel; '
        f'bf16 FLOP/s at this clock and full pipe: {1024 * 1024 * dc / (dw * 10.0) / 1e3:.0f} TFLOP/s')
