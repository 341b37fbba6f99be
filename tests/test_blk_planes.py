"""blk_planes.hip -- the block-window plane kernels (4x4 / stride-2 layers of ANY image size: the audio VAE's 96 x 80 ...
12 x 10 maps) through the C ABI against the float64 oracle, on both backends (fixture `bk`).  The launch-size threshold
is lowered to 0 so that these small shapes take the path the product sends the 2-16 GFLOP layers through.

Reference semantics: image_networks.py:460-513 (Conv2D / Conv2DTranspose k4 s2 `SAME`), base_networks.py:549
(tape.gradient)."""
import ctypes as C

import numpy as np
import pytest
import torch

from odin_ai_amd import _lib
from oracle import vae_oracle as vo


def reduce_slab(bk, slab, rows, n):
  L = bk.L
  out = bk.zeros(n)
  job = (_lib.ReduceJob * 1)(_lib.ReduceJob(slab.data_ptr(), out.data_ptr(), n, rows, slab.shape[1], 0))
  L.odin_slab_reduce(job, 1, None)
  return out.cpu().numpy()


def close(a, b, tol=2e-5):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  err = np.abs(a - b).max()
  ref = max(1.0, np.abs(b).max())
  assert err <= tol * ref, (err, ref)


@pytest.fixture
def blk(bk, request):
  L = bk.L
  request.addfinalizer(lambda old=L.odin_debug_blk_min_flop(0.0): L.odin_debug_blk_min_flop(old))
  return bk


def word_of(bk, t):
  w = bk.zeros(2048, dtype=torch.int32)
  bk.L.odin_absmax(t.data_ptr(), t.numel(), w.data_ptr(), None)
  return w


def word_max(w):
  return float(w.cpu().view(torch.float32).max())


DECONV = [
    # B, H, W, Cin, Cout, act, scale of x, scale of dy
    (2, 12, 10, 32, 32, 'elu', 1.0, 1.0),       # audio decoder geometry: ragged tiles in both directions
    (1, 6, 5, 64, 64, 'elu', 1.0, 1.0),         # 64 reduction channels, two output blocks, one (ragged) tile per image
    (2, 9, 20, 64, 32, 'relu', 1.0, 1e-9),      # odd height, 2.5 tiles per row; tiny gradients
    (2, 24, 8, 32, 64, 'linear', 3e4, 1.0),     # activations beyond the f16 window
    (5, 3, 3, 32, 32, 'elu', 1.0, 1.0),         # an image smaller than a tile
]


@pytest.mark.parametrize('B,H,W,Ci,Co,act,xs,gs', DECONV)
def test_deconv(blk, B, H, W, Ci, Co, act, xs, gs):
  bk = blk
  L, T = bk.L, bk.T
  rng = np.random.default_rng(B * 100 + H)
  K, S = 4, 2
  x = rng.standard_normal((B, H, W, Ci)) * xs
  w = rng.standard_normal((K, K, Co, Ci)) * 0.2 / xs
  b = rng.standard_normal(Co) * 0.1
  OH, OW = H * S, W * S
  _, pt, _ = vo.same_pads(OH, K, S)
  _, pl, _ = vo.same_pads(OW, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, act)
  y_ref = vo._ACT[act](vo.conv2d_transpose(x, w, b, S))
  tx, tw, tb = T(x), T(w), T(b)
  xw, yw = word_of(bk, tx), bk.zeros(2048, dtype=torch.int32)
  d.x_amax, d.y_amax = xw.data_ptr(), yw.data_ptr()
  ty = bk.full((B, OH, OW, Co), float('nan'))
  L.odin_deconv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == 'tconv_blk(f16x2)'
  close(ty.cpu().numpy(), y_ref)
  assert word_max(yw) >= float(np.abs(y_ref).max()) * (1 - 1e-5) and word_max(yw) <= float(np.abs(y_ref).max()) * 1.001
  # data gradient (strided gather over dy) and weight gradient
  dy = rng.standard_normal((B, OH, OW, Co)) * gs
  tdy = T(dy)
  dyw, dxw = word_of(bk, tdy), bk.zeros(2048, dtype=torch.int32)
  d.dy_amax, d.dx_amax = dyw.data_ptr(), dxw.data_ptr()
  dx_ref, dw_ref, db_ref = vo.conv2d_transpose_bwd(x, w, dy, S)
  aux = rng.standard_normal((B, H, W, Ci))
  taux = T(aux)
  g_ref = dx_ref * vo.elu_grad_from_output(aux.astype(np.float32).astype(np.float64))
  tdx = bk.full((B, H, W, Ci), float('nan'))
  rows = C.c_int(0)
  slab = bk.full((L.odin_max_slab_rows(), Ci), float('nan'))
  L.odin_deconv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx.data_ptr(), slab.data_ptr(),
                        C.byref(rows), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == 'fconv_blk(f16x2)' or Co != 32
  close(tdx.cpu().numpy() / gs, g_ref / gs)
  close(reduce_slab(bk, slab, rows.value, Ci) / gs, g_ref.sum((0, 1, 2)) / gs, 1e-4)
  assert word_max(dxw) >= float(np.abs(g_ref).max()) * (1 - 1e-5)
  n = K * K * Co * Ci
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_deconv2d_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == ('wgrad_planes(f16x2)' if W == 8 else 'wgrad_blk(f16x2)')   # (8-pixel rows: the row-window kernel)
  close(reduce_slab(bk, slab, rows.value, n).reshape(K, K, Co, Ci) / (gs * xs), dw_ref / (gs * xs), 1e-4)
  # the whole backward pass in one call: ONE launch where the layer has 32 output channels (bwd_blk)
  tdx3 = bk.full((B, H, W, Ci), float('nan'))
  slab3, cs3 = bk.full((L.odin_max_slab_rows(), n), float('nan')), bk.full((L.odin_max_slab_rows(), Ci), float('nan'))
  r_w, r_c = C.c_int(0), C.c_int(0)
  dxw.zero_()
  L.odin_deconv2d_bwd(tx.data_ptr(), tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx3.data_ptr(), cs3.data_ptr(),
                      C.byref(r_c), slab3.data_ptr(), C.byref(r_w), C.byref(d), None)
  if Co == 32:
    assert L.odin_debug_last_path().decode() == 'bwd_blk(f16x2)'
    d_w, d_c = C.c_int(0), C.c_int(0)   # the dry run reports the same rows
    L.odin_deconv2d_bwd(None, None, None, None, 1, None, None, C.byref(d_c), None, C.byref(d_w), C.byref(d), None)
    assert (d_w.value, d_c.value) == (r_w.value, r_c.value)
  close(tdx3.cpu().numpy() / gs, g_ref / gs)
  close(reduce_slab(bk, cs3, r_c.value, Ci) / gs, g_ref.sum((0, 1, 2)) / gs, 1e-4)
  close(reduce_slab(bk, slab3, r_w.value, n).reshape(K, K, Co, Ci) / (gs * xs), dw_ref / (gs * xs), 1e-4)
  assert word_max(dxw) >= float(np.abs(g_ref).max()) * (1 - 1e-5)


CONV = [
    # B, H, W, Cin, Cout, act, scale of x, scale of dy
    (2, 24, 20, 32, 32, 'elu', 1.0, 1.0),       # audio encoder geometry
    (1, 12, 10, 32, 64, 'elu', 1.0, 1e-9),      # two output blocks forward = 64 reduction channels backward; tiny gradients
    (3, 18, 40, 32, 32, 'relu', 3e4, 1.0),      # odd coarse height; activations beyond the f16 window
    (4, 6, 6, 32, 32, 'elu', 1.0, 1.0),         # an image smaller than a tile
]


@pytest.mark.parametrize('B,H,W,Ci,Co,act,xs,gs', CONV)
def test_conv(blk, B, H, W, Ci, Co, act, xs, gs):
  bk = blk
  L, T = bk.L, bk.T
  rng = np.random.default_rng(B * 100 + H)
  K, S = 4, 2
  x = rng.standard_normal((B, H, W, Ci)) * xs
  w = rng.standard_normal((K, K, Ci, Co)) * 0.2 / xs
  b = rng.standard_normal(Co) * 0.1
  OH, pt, _ = vo.same_pads(H, K, S)
  OW, pl, _ = vo.same_pads(W, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, act)
  y_ref = vo._ACT[act](vo.conv2d(x, w, b, S))
  tx, tw, tb = T(x), T(w), T(b)
  xw, yw = word_of(bk, tx), bk.zeros(2048, dtype=torch.int32)
  d.x_amax, d.y_amax = xw.data_ptr(), yw.data_ptr()
  ty = bk.full((B, OH, OW, Co), float('nan'))
  L.odin_conv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == 'fconv_blk(f16x2)'
  close(ty.cpu().numpy(), y_ref)
  assert word_max(yw) >= float(np.abs(y_ref).max()) * (1 - 1e-5)
  dy = rng.standard_normal((B, OH, OW, Co)) * gs
  tdy = T(dy)
  dyw, dxw = word_of(bk, tdy), bk.zeros(2048, dtype=torch.int32)
  d.dy_amax, d.dx_amax = dyw.data_ptr(), dxw.data_ptr()
  dx_ref, dw_ref, db_ref = vo.conv2d_bwd(x, w, dy, S)
  aux = rng.standard_normal((B, H, W, Ci))
  taux = T(aux)
  g_ref = dx_ref * vo.elu_grad_from_output(aux.astype(np.float32).astype(np.float64))
  tdx = bk.full((B, H, W, Ci), float('nan'))
  rows = C.c_int(0)
  slab = bk.full((L.odin_max_slab_rows(), Ci), float('nan'))
  L.odin_conv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx.data_ptr(), slab.data_ptr(),
                      C.byref(rows), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == 'tconv_blk(f16x2)'
  close(tdx.cpu().numpy() / gs, g_ref / gs)
  close(reduce_slab(bk, slab, rows.value, Ci) / gs, g_ref.sum((0, 1, 2)) / gs, 1e-4)
  assert word_max(dxw) >= float(np.abs(g_ref).max()) * (1 - 1e-5) and word_max(dxw) <= float(np.abs(g_ref).max()) * 1.001
  # without an activation below (aux_act = linear) and without a column-sum slab
  tdx2 = bk.full((B, H, W, Ci), float('nan'))
  L.odin_conv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), None, 0, tdx2.data_ptr(), None, None, C.byref(d), None)
  assert L.odin_debug_last_path().decode() == 'tconv_blk(f16x2)'
  close(tdx2.cpu().numpy() / gs, dx_ref / gs)
  n = K * K * Ci * Co + Co
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_conv2d_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == 'wgrad_blk(f16x2)'
  g = reduce_slab(bk, slab, rows.value, n)
  close(g[:-Co].reshape(K, K, Ci, Co) / (gs * xs), dw_ref / (gs * xs), 1e-4)
  close(g[-Co:] / gs, db_ref / gs, 1e-4)


@pytest.mark.parametrize('B,H,W,act,sp1', [(2, 12, 10, 'elu', 1), (3, 5, 9, 'relu', 0), (1, 16, 8, 'elu', 1)])
def test_gaussian_tail(blk, B, H, W, act, sp1):
  """odin_gaussian_tail_fwd_bwd (Conv2DTranspose 32 -> 32 -> 1x1 head of 2 maps -> Normal log-prob, forward + backward in
  one launch) against torch float64 autograd of the same three steps (the style of oracle/torch_ref.py)."""
  import torch.nn.functional as F
  bk = blk
  L, T = bk.L, bk.T
  rng = np.random.default_rng(7 * B + H)
  OH, OW = 2 * H, 2 * W
  N = dict(x=rng.standard_normal((B, H, W, 32)), w=rng.standard_normal((4, 4, 32, 32)) * 0.1,
           b=rng.standard_normal(32) * 0.1, w1=rng.standard_normal((32, 2)) * 0.2, b1=np.array([0.1, 0.7 if sp1 else 1.5]),
           t=rng.standard_normal((B, OH, OW, 1)))
  if not sp1:
    N['w1'][:, 1] *= 0.05     # (a raw scale must stay positive)
  d = _lib.conv_desc(B, H, W, 32, OH, OW, 32, 4, 2, 1, 1, act)
  tn = {k: T(v) for k, v in N.items()}
  gw, xw = bk.zeros(2048, dtype=torch.int32), word_of(bk, tn['x'])
  d.x_amax, d.dy_amax = xw.data_ptr(), gw.data_ptr()
  assert L.odin_gaussian_tail_applicable(C.byref(d), 1) == 1
  rows, npart = C.c_int(0), C.c_int(0)
  L.odin_gaussian_tail_fwd_bwd(None, None, None, None, None, None, None, None, None, C.byref(npart), None, C.byref(rows),
                               None, C.byref(d), 1, sp1, None)
  assert npart.value == ((H + 7) // 8) * ((W + 7) // 8) and rows.value >= 1
  logits, g = bk.full((B, OH, OW, 2), float('nan')), bk.full((B, OH, OW, 32), float('nan'))
  llk_part, slab = bk.full((B * npart.value,), float('nan')), bk.full((rows.value, 98), float('nan'))
  scale = T(np.array([1.0 / B]))
  r2, n2 = C.c_int(0), C.c_int(0)
  L.odin_gaussian_tail_fwd_bwd(tn['x'].data_ptr(), tn['w'].data_ptr(), tn['b'].data_ptr(), tn['w1'].data_ptr(),
                               tn['b1'].data_ptr(), tn['t'].data_ptr(), logits.data_ptr(), g.data_ptr(),
                               llk_part.data_ptr(), C.byref(n2), slab.data_ptr(), C.byref(r2), scale.data_ptr(),
                               C.byref(d), 1, sp1, None)
  assert (r2.value, n2.value) == (rows.value, npart.value)
  assert L.odin_debug_last_path().decode() == 'tconv_blk_gtail(f16x2)'
  # ---- float64 restatement
  actf = {'elu': F.elu, 'relu': F.relu}[act]
  tt = {k: torch.tensor(v, dtype=torch.float64) for k, v in N.items()}
  for k in ('w1', 'b1', 'b'):
    tt[k].requires_grad_(True)
  pre = F.conv_transpose2d(tt['x'].permute(0, 3, 1, 2), tt['w'].permute(3, 2, 0, 1), tt['b'], stride=2, padding=1)
  pre.retain_grad()
  y = actf(pre).permute(0, 2, 3, 1)
  lg = y @ tt['w1'] + tt['b1']
  loc, raw = lg[..., :1], lg[..., 1:]
  sd = F.softplus(raw + np.log(np.e - 1.0)) if sp1 else raw
  llk = (-0.5 * ((tt['t'] - loc) / sd) ** 2 - torch.log(sd) - 0.5 * np.log(2 * np.pi)).sum((1, 2, 3))
  loss = -(llk.sum() / B)
  loss.backward()
  close(logits.cpu().numpy(), lg.detach().numpy())
  close(llk_part.cpu().numpy().reshape(B, -1).sum(1), llk.detach().numpy(), 1e-5)
  g_ref = pre.grad.permute(0, 2, 3, 1).numpy()
  close(g.cpu().numpy(), g_ref, 1e-4 * max(1e-30, float(np.abs(g_ref).max())) / max(1.0, float(np.abs(g_ref).max())))
  row = slab.cpu().numpy().astype(np.float64).sum(0)
  close(row[:64].reshape(32, 2), tt['w1'].grad.numpy(), 1e-4)
  close(row[64:66], tt['b1'].grad.numpy(), 1e-4)
  close(row[66:], tt['b'].grad.numpy(), 1e-4)
  assert word_max(gw) >= float(np.abs(g_ref).max()) * (1 - 1e-5) and word_max(gw) <= float(np.abs(g_ref).max()) * 1.001


CONV5 = [
    # K, B, H, W, Cin, Cout, act, scale of x, scale of dy   (K x K / stride 1, `SAME`: blk5_planes.hip)
    (5, 2, 14, 14, 32, 64, 'elu', 1.0, 1.0),       # MNIST encoder2: ragged 8 x 8 tiles, two output blocks
    (5, 1, 28, 28, 32, 32, 'relu', 1.0, 1.0),      # MNIST decoder4
    (5, 1, 14, 14, 64, 64, 'elu', 1.0, 1e-9),      # MNIST decoder2: 64 reduction channels = two passes; tiny gradients
    (5, 2, 9, 11, 64, 32, 'linear', 3e4, 1.0),     # odd sizes; activations beyond the f16 window
    (4, 3, 8, 8, 64, 64, 'elu', 1.0, 1.0),         # CelebA encoder3: 4 x 4 / stride 1, `SAME` pads (1, 2)
    (4, 2, 9, 13, 32, 64, 'relu', 1.0, 1e-9),      # the same kernel size on odd maps, 32 reduction channels forward
]


@pytest.mark.parametrize('K,B,H,W,Ci,Co,act,xs,gs', CONV5)
def test_conv5(blk, K, B, H, W, Ci, Co, act, xs, gs):
  bk = blk
  L, T = bk.L, bk.T
  rng = np.random.default_rng(B * 100 + H)
  S = 1
  x = rng.standard_normal((B, H, W, Ci)) * xs
  w = rng.standard_normal((K, K, Ci, Co)) * 0.1 / xs
  b = rng.standard_normal(Co) * 0.1
  OH, pt, _ = vo.same_pads(H, K, S)
  OW, pl, _ = vo.same_pads(W, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, act)
  y_ref = vo._ACT[act](vo.conv2d(x, w, b, S))
  tx, tw, tb = T(x), T(w), T(b)
  xw, yw = word_of(bk, tx), bk.zeros(2048, dtype=torch.int32)
  d.x_amax, d.y_amax = xw.data_ptr(), yw.data_ptr()
  ty = bk.full((B, OH, OW, Co), float('nan'))
  L.odin_conv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == ('conv5_blk(f16x2)' if K == 5 else 'conv4s1_blk(f16x2)')
  close(ty.cpu().numpy(), y_ref)
  assert word_max(yw) >= float(np.abs(y_ref).max()) * (1 - 1e-5) and word_max(yw) <= float(np.abs(y_ref).max()) * 1.001
  dy = rng.standard_normal((B, OH, OW, Co)) * gs
  tdy = T(dy)
  dyw, dxw = word_of(bk, tdy), bk.zeros(2048, dtype=torch.int32)
  d.dy_amax, d.dx_amax = dyw.data_ptr(), dxw.data_ptr()
  dx_ref, dw_ref, db_ref = vo.conv2d_bwd(x, w, dy, S)
  aux = rng.standard_normal((B, H, W, Ci))
  taux = T(aux)
  g_ref = dx_ref * vo.elu_grad_from_output(aux.astype(np.float32).astype(np.float64))
  tdx = bk.full((B, H, W, Ci), float('nan'))
  rows = C.c_int(0)
  slab = bk.full((L.odin_max_slab_rows(), Ci), float('nan'))
  L.odin_conv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx.data_ptr(), slab.data_ptr(),
                      C.byref(rows), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == ('conv5_blk(f16x2)' if K == 5 else 'conv4s1_blk(f16x2)')
  close(tdx.cpu().numpy() / gs, g_ref / gs)
  close(reduce_slab(bk, slab, rows.value, Ci) / gs, g_ref.sum((0, 1, 2)) / gs, 1e-4)
  assert word_max(dxw) >= float(np.abs(g_ref).max()) * (1 - 1e-5) and word_max(dxw) <= float(np.abs(g_ref).max()) * 1.001
  tdx2 = bk.full((B, H, W, Ci), float('nan'))
  L.odin_conv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), None, 0, tdx2.data_ptr(), None, None, C.byref(d), None)
  close(tdx2.cpu().numpy() / gs, dx_ref / gs)
  n = K * K * Ci * Co + Co
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_conv2d_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  assert L.odin_debug_last_path().decode() == ('wgrad5_blk(f16x2)' if K == 5 else 'wgrad4s1_blk(f16x2)')
  g = reduce_slab(bk, slab, rows.value, n)
  close(g[:-Co].reshape(K, K, Ci, Co) / (gs * xs), dw_ref / (gs * xs), 1e-4)
  close(g[-Co:] / gs, db_ref / gs, 1e-4)


@pytest.mark.gpu
def test_run_to_run_determinism_at_layer_size():
  """Every launch of a block-window kernel on the same inputs gives the same bits.  The first form of these kernels issued
  two v_mfma_f32_16x16x32_f16 into one accumulator with a single independent MFMA between them: 7 % of the launches of the
  audio decoder2 forward came out wrong in a few hundred elements of a workgroup's first tile (blk_common.h, bk_mfma16 note;
  tools/race_ops.py runs the whole family 150 times)."""
  from odin_ai_amd import _lib as lib_mod
  L = lib_mod.load()
  dev = torch.device('cuda:0')
  g = torch.Generator(device='cpu').manual_seed(3)
  R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
  st = torch.cuda.current_stream().cuda_stream
  for B, H, W, Ci, Co in ((256, 12, 10, 64, 64), (256, 24, 20, 64, 32)):
    d = _lib.conv_desc(B, H, W, Ci, 2 * H, 2 * W, Co, 4, 2, 1, 1, 'elu')
    x, wt, b = R(B, H, W, Ci), R(4, 4, Co, Ci, sc=0.05), R(Co, sc=0.1)
    ref = None
    for it in range(40):
      y = torch.full((B, 2 * H, 2 * W, Co), float('nan'), device=dev)
      L.odin_deconv2d_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), st)
      torch.cuda.synchronize()
      assert L.odin_debug_last_path().decode() == 'tconv_blk(f16x2)'
      if ref is None:
        ref = y
        yr = torch.nn.functional.elu(torch.nn.functional.conv_transpose2d(
            x.double().permute(0, 3, 1, 2), wt.double().permute(3, 2, 0, 1), b.double(), stride=2, padding=1)).permute(0, 2, 3, 1)
        assert float((y.double() - yr).abs().max()) <= 2e-5 * max(1.0, float(yr.abs().max()))
      else:
        assert torch.equal(y, ref), (it, int((y != ref).sum()))
