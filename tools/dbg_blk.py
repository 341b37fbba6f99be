import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from odin_ai_amd import _lib
from oracle import vae_oracle as vo
L = _lib.load(); dev = torch.device('cuda:0')
L.odin_debug_blk_min_flop(0.0)
for act in ('elu', 'relu'):
  B, H, W, Ci, Co = 2, 12, 10, 32, 32
  rng = np.random.default_rng(1)
  x = rng.standard_normal((B, H, W, Ci)); w = rng.standard_normal((4, 4, Co, Ci)) * 0.2; b = rng.standard_normal(Co) * 0.1
  d = _lib.conv_desc(B, H, W, Ci, 2 * H, 2 * W, Co, 4, 2, 1, 1, act)
  y_ref = vo._ACT[act](vo.conv2d_transpose(x, w, b, 2))
  T = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
  tx, tw, tb = T(x), T(w), T(b)
  ty = torch.full((B, 2 * H, 2 * W, Co), float('nan'), device=dev)
  L.odin_deconv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  torch.cuda.synchronize()
  err = np.abs(ty.cpu().numpy() - y_ref)
  print(act, L.odin_debug_last_path().decode(), 'max err', err.max(), 'nan', np.isnan(err).sum())
  bad = err > 1e-3
  print(' bad count', bad.sum(), 'of', bad.size)
  if bad.sum():
    idx = np.argwhere(bad)
    print(' first bad', idx[:5].tolist(), 'channels bad hist', np.bincount(idx[:, 3], minlength=32).tolist())
    print(' rows bad hist', np.bincount(idx[:, 1], minlength=24).tolist())
    print(' cols bad hist', np.bincount(idx[:, 2], minlength=20).tolist())
    i = idx[0]; print(' got', ty.cpu().numpy()[tuple(i)], 'ref', y_ref[tuple(i)])
