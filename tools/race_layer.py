#!/usr/bin/env python3
"""repeat ONE launch of the audio decoder2 forward (tconv_blk, 64 -> 64 channels, 12 x 10 -> 24 x 20) and compare every run
with the float64 oracle: which elements go wrong, how often, in which tile of a workgroup"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from odin_ai_amd import _lib
L = _lib.load(); dev = torch.device('cuda:0')
B, H, W, Ci, Co = 256, 12, 10, 64, 64
rng = np.random.default_rng(1)
x = torch.tensor(rng.standard_normal((B, H, W, Ci)), dtype=torch.float32, device=dev)
w = torch.tensor(rng.standard_normal((4, 4, Co, Ci)) * 0.05, dtype=torch.float32, device=dev)
b = torch.tensor(rng.standard_normal(Co) * 0.1, dtype=torch.float32, device=dev)
d = _lib.conv_desc(B, H, W, Ci, 2 * H, 2 * W, Co, 4, 2, 1, 1, 'elu')
ref = torch.nn.functional.elu(torch.nn.functional.conv_transpose2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), stride=2, padding=1)).permute(0, 2, 3, 1)
st = torch.cuda.current_stream().cuda_stream
nbad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
  y = torch.full((B, 2 * H, 2 * W, Co), float('nan'), device=dev)
  L.odin_deconv2d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), st)
  torch.cuda.synchronize()
  err = (y.double() - ref).abs()
  bad = torch.nonzero(err > 1e-3)
  if bad.shape[0]:
    nbad += 1
    if nbad <= 6:
      i = bad[0].tolist()
      print('run', it, L.odin_debug_last_path().decode(), 'bad elements', bad.shape[0], 'samples', sorted(set(bad[:, 0].tolist()))[:8], 'rows', sorted(set(bad[:, 1].tolist())),
            'cols', sorted(set(bad[:, 2].tolist())), 'channels', sorted(set(bad[:, 3].tolist())), 'got', float(y[tuple(i)]), 'ref', float(ref[tuple(i)]))
print('runs with wrong elements:', nbad)
