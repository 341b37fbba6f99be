"""Op-level parity of every conv / deconv / dense / fused-tail entry point of the C ABI vs the
float64 oracle, on two backends (fixture `bk`, tests/conftest.py):
  'sim' -- the kernel SOURCES built for the CPU fiber simulator (debugs tiling / indexing in the
           build container, `-m "not gpu"`);
  'hip' -- the hipcc gfx950 build on an MI355X (`-m gpu`): the parity tests proper.
"""
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


CONV_CASES = [
    # B, H, W, Cin, Cout, K, S, act, center
    (2, 16, 16, 1, 32, 4, 2, 'elu', True),
    (2, 16, 16, 3, 32, 4, 2, 'elu', True),     # small-Cin VALU kernels, 48 taps
    (1, 64, 64, 1, 32, 4, 2, 'elu', True),     # first-layer matrix-core kernels (32-pixel row blocks)
    (1, 64, 64, 3, 64, 4, 2, 'elu', True),     # first-layer matrix-core kernels, 48 taps, 64 channels
    (2, 64, 64, 3, 32, 4, 2, 'elu', True),     # RGB first layer: rows staged in LDS (forward and weight gradient), image seam
    (2, 64, 64, 1, 32, 4, 2, 'elu', False),    # the same with one channel, no centring
    (2, 96, 80, 1, 32, 4, 2, 'elu', False),    # the audio VAE's first layer: 40-pixel output rows = a ragged second 32-pixel block
    (1, 16, 88, 3, 32, 4, 2, 'elu', True),     # RGB on two f16 planes with a ragged block (44-pixel rows)
    (3, 12, 12, 1, 64, 5, 1, 'elu', False),    # small-Cin VALU kernels, 25 taps, 64 lanes per pixel
    (3, 8, 8, 32, 32, 4, 2, 'elu', False),
    (2, 8, 8, 32, 64, 4, 2, 'elu', False),
    (5, 4, 4, 64, 64, 4, 2, 'linear', False),
    (2, 8, 8, 64, 64, 4, 1, 'elu', False),
    (2, 14, 14, 3, 8, 5, 2, 'relu', False),
    (2, 7, 7, 8, 16, 5, 1, 'elu', False),
    (2, 16, 16, 32, 3, 1, 1, 'linear', False),  # streaming 1x1 kernels (pw1x1.hip)
    (3, 12, 20, 32, 2, 1, 1, 'linear', False),  # Gaussian head: 2 maps, ragged pixel count
    (2, 9, 7, 64, 6, 1, 1, 'linear', False),    # CelebA two-parameter head: 6 maps over 64 channels
    (1, 8, 8, 16, 8, 1, 1, 'elu', False),
    (5, 4, 4, 8, 1, 1, 1, 'relu', False),
    (16, 8, 8, 64, 64, 4, 2, 'elu', False),    # small-M: waves split the reduction (KS=4)
    (37, 4, 4, 8, 8, 4, 2, 'elu', False),      # several whole images per tile, ragged last tile
    (1, 8, 8, 160, 40, 4, 2, 'linear', False),  # channel-chunked reduction
    (3, 16, 16, 32, 32, 4, 2, 'elu', False),   # wgrad row-chunk loop (S*P = 64), two images per tile, ragged
    (2, 16, 16, 64, 32, 4, 2, 'elu', False),   # wgrad row-chunk loop (S*P = 128)
    (1, 16, 16, 32, 32, 3, 1, 'relu', False),  # wgrad row-chunk loop (S*P = 32), 16-wide rows
    (1, 32, 32, 32, 32, 4, 2, 'elu', False),   # two-workgroup data-gradient instance (EPI 2); fwd: rolling-window kernel
    (5, 32, 32, 32, 32, 4, 2, 'elu', False),   # fconv_ring forward: several tiles per workgroup across image boundaries
    (1, 64, 64, 32, 32, 4, 2, 'elu', False),   # 32-pixel output rows: fconv_planes / fconv_ring forward, wgrad_planes, tconv_planes data gradient
    (3, 32, 32, 32, 64, 4, 2, 'elu', False),   # fconv_ring forward, two 32-channel output blocks
    (3, 16, 16, 32, 64, 4, 2, 'elu', False),   # fconv_ring forward, 8-pixel output rows (a wave spans two rows; encoder2)
    (2, 32, 32, 64, 32, 4, 2, 'elu', False),   # fconv_ring forward over 64 channels: two reduction passes
    (2, 32, 32, 32, 64, 4, 2, 'elu', False),   # data gradient = tconv_planes over 64 reduction channels (two passes)
    (1, 24, 80, 32, 32, 4, 2, 'elu', False),   # speech stack (80-pixel rows): wide-row instances of the 4x4/s2 gathers
    # row widths that are not powers of two: the two-plane implicit GEMM (igemm_h.hip), all three roles
    (3, 24, 20, 32, 64, 4, 2, 'elu', False),   # speech encoder3 geometry
    (5, 12, 40, 32, 32, 4, 2, 'elu', False),   # ragged last tile, 16-channel steps over 32 channels
    (7, 8, 8, 64, 64, 4, 2, 'elu', False),     # encoder3 of the image stacks (8 x 8 -> 4 x 4)
    (3, 10, 12, 48, 48, 3, 1, 'relu', False),  # stride 1, 3 x 3, ragged channel tile
]
IGEMM_H_CONV = {(3, 24, 20, 32, 64), (5, 12, 40, 32, 32), (7, 8, 8, 64, 64), (3, 10, 12, 48, 48)}


@pytest.mark.parametrize('B,H,W,Ci,Co,K,S,act,center', CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(bk, request, B, H, W, Ci, Co, K, S, act, center):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(0)
  x = rng.random((B, H, W, Ci))
  w = rng.standard_normal((K, K, Ci, Co)) * 0.2
  b = rng.standard_normal(Co) * 0.1
  OH, pt, _ = vo.same_pads(H, K, S)
  OW, pl, _ = vo.same_pads(W, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, act, center)
  xin = 2 * x - 1 if center else x
  y_ref = vo._ACT[act](vo.conv2d(xin, w, b, S))
  tx, tw, tb = T(x), T(w), T(b)
  ty = bk.full((B, OH, OW, Co), float('nan'))
  if (B, H, W, Ci, Co) in IGEMM_H_CONV:  # (test sizes are far below the launch size the product sends there)
    request.addfinalizer(lambda old=L.odin_debug_igemm_h_min_flop(0.0): L.odin_debug_igemm_h_min_flop(old))
  L.odin_conv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  close(ty.cpu().numpy(), y_ref)
  want_h = (B, H, W, Ci, Co) in IGEMM_H_CONV
  print('conv', (B, H, W, Ci, Co), L.odin_debug_last_path().decode())
  if want_h:
    assert L.odin_debug_last_path().decode() == 'igemm_h(f16x2)'
  # backward
  dy = rng.standard_normal((B, OH, OW, Co))
  tdy = T(dy)
  dx_ref, dw_ref, db_ref = vo.conv2d_bwd(xin, w, dy, S)
  if H % S == 0 and W % S == 0:
    aux = rng.standard_normal((B, H, W, Ci))
    taux = T(aux)
    tdx = bk.full((B, H, W, Ci), float('nan'))
    rows = C.c_int(0)
    slab = bk.full((L.odin_max_slab_rows(), Ci), float('nan'))
    L.odin_conv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx.data_ptr(),
                        slab.data_ptr(), C.byref(rows), C.byref(d), None)
    path = L.odin_debug_last_path().decode()
    g_ref = dx_ref * vo.elu_grad_from_output(aux.astype(np.float32).astype(np.float64))
    close(tdx.cpu().numpy(), g_ref)
    close(reduce_slab(bk, slab, rows.value, Ci), g_ref.sum((0, 1, 2)), 1e-4)
    if want_h:
      assert path == 'igemm_h(f16x2)', path
  rows = C.c_int(0)
  n = K * K * Ci * Co + Co
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_conv2d_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d),
                      None)
  path = L.odin_debug_last_path().decode()
  g = reduce_slab(bk, slab, rows.value, n)
  close(g[:-Co].reshape(K, K, Ci, Co), dw_ref, 1e-4)
  close(g[-Co:], db_ref, 1e-4)
  if want_h:
    assert path == 'igemm_h_wgrad(f16x2)', path


DECONV_CASES = [
    (1, 16, 16, 32, 32, 4, 2, 'elu'),          # two-workgroup forward instance (EPI 1); dgrad: rolling-window kernel
    (5, 16, 16, 32, 32, 4, 2, 'elu'),          # fconv_ring data gradient across image boundaries
    (2, 32, 32, 32, 32, 4, 2, 'elu'),          # 32-pixel rows, two images (tiles across the image seam): tconv_planes forward, fconv_planes / fconv_ring data gradient, wgrad_planes (decoder4)
    (3, 16, 16, 64, 32, 4, 2, 'elu'),          # fconv_ring data gradient into 64 channels; forward: tconv_planes over 64 channels in two passes (decoder3)
    (3, 8, 8, 64, 64, 4, 2, 'elu'),            # fconv_ring data gradient: 8-pixel rows, 64 reduction channels in two passes; forward: tconv_planes, 8-pixel input rows (decoder2)
    (5, 8, 8, 32, 32, 4, 2, 'elu'),            # tconv_planes forward, 8-pixel input rows, one pass
    (3, 4, 4, 8, 64, 4, 2, 'elu'),             # the decoders' first Conv2DTranspose: smalldeconv.hip (dSprites: 8 channels)
    (5, 4, 4, 16, 64, 4, 2, 'elu'),            # ... Shapes3D: 16 channels, an odd batch (the last workgroup holds one sample)
    (2, 6, 5, 8, 64, 4, 2, 'relu'),            # ... the audio decoder's 6 x 5 image
    (3, 7, 7, 4, 64, 5, 2, 'elu'),             # MNIST's first Conv2DTranspose (4 channels, 5 x 5): the generic forward of smalldeconv.hip
    (2, 5, 3, 12, 64, 3, 2, 'relu'),           # ... 12 channels, 3 x 3, a ragged image
    (2, 8, 8, 64, 32, 4, 2, 'elu'),
    (1, 16, 16, 32, 32, 4, 2, 'linear'),
    (2, 16, 16, 32, 64, 4, 2, 'elu'),          # tconv_planes forward, two 32-channel output blocks
    (1, 32, 32, 64, 32, 4, 2, 'elu'),          # tconv_planes forward over 64 channels (two reduction passes), 32-pixel rows
    (2, 8, 8, 8, 64, 4, 1, 'elu'),
    (3, 7, 7, 4, 16, 5, 2, 'elu'),
    (1, 12, 40, 32, 32, 4, 2, 'elu'),          # speech decoder4 geometry (40 -> 80 pixels per row): wide-row instances
    (1, 12, 20, 64, 32, 4, 2, 'elu'),          # speech decoder3 geometry, 64 reduction channels
    # the two-plane implicit GEMM (igemm_h.hip) on row widths that are not powers of two
    (2, 12, 20, 64, 32, 4, 2, 'elu'),          # speech decoder3
    (2, 24, 40, 32, 32, 4, 2, 'elu'),          # speech decoder4 (40 -> 80 pixels per row)
    (7, 4, 4, 64, 64, 4, 2, 'elu'),            # 4 x 4 -> 8 x 8 over 64 channels
]
IGEMM_H_DECONV = {(2, 12, 20, 64, 32), (2, 24, 40, 32, 32), (7, 4, 4, 64, 64)}


@pytest.mark.parametrize('B,H,W,Ci,Co,K,S,act', DECONV_CASES)
def test_deconv2d_fwd_dgrad_wgrad(bk, request, B, H, W, Ci, Co, K, S, act):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(1)
  x = rng.standard_normal((B, H, W, Ci))
  w = rng.standard_normal((K, K, Co, Ci)) * 0.2
  b = rng.standard_normal(Co) * 0.1
  OH, OW = H * S, W * S
  _, pt, _ = vo.same_pads(OH, K, S)
  _, pl, _ = vo.same_pads(OW, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, act)
  y_ref = vo._ACT[act](vo.conv2d_transpose(x, w, b, S))
  tx, tw, tb = T(x), T(w), T(b)
  ty = bk.full((B, OH, OW, Co), float('nan'))
  if (B, H, W, Ci, Co) in IGEMM_H_DECONV:  # (test sizes are far below the launch size the product sends there)
    request.addfinalizer(lambda old=L.odin_debug_igemm_h_min_flop(0.0): L.odin_debug_igemm_h_min_flop(old))
  L.odin_deconv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  close(ty.cpu().numpy(), y_ref)
  want_h = (B, H, W, Ci, Co) in IGEMM_H_DECONV
  print('deconv', (B, H, W, Ci, Co), L.odin_debug_last_path().decode())
  if want_h:
    assert L.odin_debug_last_path().decode() == 'igemm_h(f16x2)'
  dy = rng.standard_normal((B, OH, OW, Co))
  tdy = T(dy)
  dx_ref, dw_ref, db_ref = vo.conv2d_transpose_bwd(x, w, dy, S)
  aux = rng.standard_normal((B, H, W, Ci))
  taux = T(aux)
  tdx = bk.full((B, H, W, Ci), float('nan'))
  rows = C.c_int(0)
  slab = bk.full((L.odin_max_slab_rows(), Ci), float('nan'))
  L.odin_deconv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx.data_ptr(),
                        slab.data_ptr(), C.byref(rows), C.byref(d), None)
  path = L.odin_debug_last_path().decode()
  g_ref = dx_ref * vo.elu_grad_from_output(aux.astype(np.float32).astype(np.float64))
  close(tdx.cpu().numpy(), g_ref)
  close(reduce_slab(bk, slab, rows.value, Ci), g_ref.sum((0, 1, 2)), 1e-4)
  if want_h:
    assert path == 'igemm_h(f16x2)', path
  # without a column-sum slab (the layer below is no Conv2DTranspose): other kernel families take the call
  # (smalldeconv.hip for the decoders' first Conv2DTranspose)
  tdx2 = bk.full((B, H, W, Ci), float('nan'))
  L.odin_deconv2d_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, tdx2.data_ptr(), None, None, C.byref(d), None)
  print('deconv dgrad, no column sums', (B, H, W, Ci, Co), L.odin_debug_last_path().decode())
  if (H, W, Ci, Co, K, S) in ((4, 4, 8, 64, 4, 2), (4, 4, 16, 64, 4, 2), (6, 5, 8, 64, 4, 2)):
    assert L.odin_debug_last_path().decode() == 'smalldeconv_bwd'
  close(tdx2.cpu().numpy(), g_ref)
  n = K * K * Co * Ci
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_deconv2d_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows),
                        C.byref(d), None)
  path = L.odin_debug_last_path().decode()
  close(reduce_slab(bk, slab, rows.value, n).reshape(K, K, Co, Ci), dw_ref, 1e-4)
  if want_h:
    assert path == 'igemm_h_wgrad(f16x2)', path


@pytest.mark.parametrize('B,K,N,act', [(5, 1024, 128, 'linear'), (130, 10, 128, 'linear'),
                                       (7, 128, 20, 'linear'), (33, 100, 70, 'relu'),
                                       (3, 4096, 40, 'linear'), (150, 36, 33, 'relu'),
                                       (100, 256, 40, 'linear'), (70, 512, 64, 'relu'),
                                       # narrow heads on the tiny-Dense path (staged rows capped to
                                       # 32 KB of LDS): FactorVAE(discriminator_units=(128,128)) ends
                                       # in Dense(128 -> 1)
                                       (128, 128, 1, 'linear'), (256, 256, 1, 'linear'),
                                       (200, 256, 4, 'relu'), (64, 128, 2, 'linear'),
                                       (300, 1, 256, 'linear'), (256, 2, 200, 'relu'),
                                       # small matrix-core GEMMs straight from L2 (dense_gemm.hip): ragged
                                       # M / N / K, 1..16 waves per tile
                                       (256, 1024, 128, 'linear'), (37, 1000, 75, 'relu'),
                                       (128, 70, 1000, 'relu'), (64, 784, 512, 'relu'), (9, 2052, 33, 'linear'),
                                       # both widths >= 256, batch a multiple of 8: the two-plane GEMM (dense_h.hip), ragged N
                                       (40, 264, 296, 'relu'), (32, 512, 256, 'linear'),
                                       # one thin side (thin_dense.hip): FactorVAE's first / last discriminator layers,
                                       # ragged batches, every K bucket of the thin-K kernels, 1..4 thin-N outputs
                                       (130, 6, 1000, 'relu'), (33, 10, 64, 'elu'), (7, 16, 260, 'relu'), (66, 31, 128, 'linear'),
                                       (129, 1000, 1, 'linear'), (5, 260, 3, 'relu'), (64, 1024, 4, 'linear')])
def test_dense(bk, B, K, N, act):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(2)
  x = rng.standard_normal((B, K))
  w = rng.standard_normal((K, N)) / np.sqrt(K)
  b = rng.standard_normal(N) * 0.1
  y_ref = vo._ACT[act](vo.dense(x, w, b))
  ty = bk.full((B, N), float('nan'))
  tx, tw, tb = T(x), T(w), T(b)
  L.odin_dense_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), B, K, N,
                   _lib.ACT[act], None)
  close(ty.cpu().numpy(), y_ref)
  dy = rng.standard_normal((B, N))
  tdy = T(dy)
  dx_ref, dw_ref, db_ref = vo.dense_bwd(x, w, dy)
  tdx = bk.full((B, K), float('nan'))
  L.odin_dense_dgrad(tdy.data_ptr(), tw.data_ptr(), None, 0, tdx.data_ptr(), None, None, B, K, N,
                     None)
  close(tdx.cpu().numpy(), dx_ref)
  rows = C.c_int(0)
  n = K * N + N
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_dense_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), B, K, N, None)
  g = reduce_slab(bk, slab, rows.value, n)
  close(g[:-N].reshape(K, N), dw_ref, 1e-4)
  close(g[-N:], db_ref, 1e-4)


@pytest.mark.parametrize('B,K,N,scale,ranged', [
    # one short chunk; two chunks, the second short; three (the third prefetched behind the first); five and six (the
    # loop's own prefetches, odd and even counts); ragged widths (264 = 4 tiles + 8 columns, 296 = 4 tiles + 40)
    (40, 264, 296, 1.0, False), (72, 264, 296, 1.0, True), (136, 296, 264, 1.0, False), (264, 256, 320, 1.0, True),
    (328, 264, 264, 1.0, False),
    # activations outside the f16 window with and without their range word (without: |x| <= 65504 is the contract)
    (72, 256, 264, 3e4, True), (72, 256, 264, 1e-6, True), (64, 264, 256, 2e3, False)])
def test_dense_weight_gradient_lds_staged(bk, request, B, K, N, scale, ranged):
  """dense_hw (dense_h.hip): the weight gradient of a Dense layer with both widths >= 256 over 64 x 64 tiles with the
  operands staged through LDS -- alone and as the second role of the paired launch (odin_dense_bwd), against the
  oracle (odin/networks/base_networks.py:1002-1014 under tf.GradientTape: dW = x^T dy, db = column sums of dy)."""
  L, T = bk.L, bk.T
  request.addfinalizer(lambda old=L.odin_debug_dense_hw_min_tiles(1): L.odin_debug_dense_hw_min_tiles(old))
  rng = np.random.default_rng(B + K)
  x = rng.standard_normal((B, K)) * scale
  w = rng.standard_normal((K, N)) / np.sqrt(K)
  dy = rng.standard_normal((B, N)) * 1e-3
  aux = rng.standard_normal((B, K))
  tx, tw, tdy, taux = T(x), T(w), T(dy), T(aux)
  xf, dyf = x.astype(np.float32).astype(np.float64), dy.astype(np.float32).astype(np.float64)
  dx_ref, dw_ref, db_ref = vo.dense_bwd(xf, w.astype(np.float32).astype(np.float64), dyf)
  xw = bk.zeros(2048, dtype=torch.int32)
  L.odin_absmax(tx.data_ptr(), tx.numel(), xw.data_ptr(), None)
  n = K * N + N

  def rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max()

  rows = C.c_int(0)
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_dense_bwd_ranged(tx.data_ptr(), tdy.data_ptr(), None, None, 0, None, None, None, slab.data_ptr(), C.byref(rows),
                          B, K, N, 1, 0, None, None, xw.data_ptr() if ranged else None, None)
  assert 'dense_hw' in L.odin_debug_last_path().decode()
  assert rows.value == 1
  g = slab[0].cpu().numpy()
  assert rel(g[:K * N].reshape(K, N), dw_ref) <= 2e-6
  assert rel(g[K * N:], db_ref) <= 2e-6
  # both halves in one call: the same tiles, bit for bit
  rows2 = C.c_int(0)
  slab2 = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  dx = bk.full((B, K), float('nan'))
  L.odin_dense_bwd_ranged(tx.data_ptr(), tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), _lib.ACT['relu'], dx.data_ptr(), None,
                          None, slab2.data_ptr(), C.byref(rows2), B, K, N, 1, 1, None, None, xw.data_ptr() if ranged else None,
                          None)
  assert torch.equal(slab2[0], slab[0])
  assert rel(dx.cpu().numpy(), dx_ref * (aux.astype(np.float32) > 0)) <= 2e-6
  # the data gradient alone: dense_hd, the LDS-staged form for two k-contiguous operands (with the tests' setting every
  # shape runs on it); its range word is max |dx| exactly
  word, dx2 = bk.zeros(2048, dtype=torch.int32), bk.full((B, K), float('nan'))
  L.odin_dense_bwd_ranged(None, tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), _lib.ACT['relu'], dx2.data_ptr(), None, None,
                          None, None, B, K, N, 0, 1, None, word.data_ptr(), None, None)
  assert 'dense_hd' in L.odin_debug_last_path().decode()
  assert rel(dx2.cpu().numpy(), dx_ref * (aux.astype(np.float32) > 0)) <= 2e-6
  assert float(word.view(torch.float32).max()) == float(dx2.abs().max())


@pytest.mark.parametrize('B,K,mode,act', [(128, 1000, 0, 'relu'), (256, 1000, 1, 'relu'), (20, 64, 1, 'elu'), (13, 1284, 0, 'relu'),
                                          (66, 2048, 1, 'linear')])
def test_discriminator_head_in_one_launch(bk, B, K, mode, act):
  """odin_disc_head_fwd_bwd: FactorDiscriminator's last Dense(K -> 1) (factor_discriminator.py:60-95) with the mean it
  feeds -- total_correlation (:169-198, mode 0) or dtc_loss (:200-235, mode 1) -- its data gradient and its weight
  gradient rows, against the oracle; twice on one workspace (every launch leaves it zero), bit for bit."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(B + K + mode)
  h = vo._ACT[act](rng.standard_normal((B, K)))
  w = rng.standard_normal((K, 1)) / np.sqrt(K)
  b = rng.standard_normal(1) * 0.1
  hf, wf, bf = (a.astype(np.float32).astype(np.float64) for a in (h, w, b))
  logit = (hf @ wf + bf)[:, 0]
  n = B // 2
  if mode == 0:
    dl = np.full(B, 7.0 / B)
    out_ref = logit.mean()
  else:
    dz, dp = vo.dtc_loss_bwd(logit[:n], logit[n:])
    dl = np.concatenate([dz, dp])
    out_ref = vo.dtc_loss(logit[:n], logit[n:])
  dgrad = {'relu': (hf > 0).astype(np.float64), 'linear': np.ones_like(hf), 'elu': np.where(hf > 0, 1.0, 1.0 + hf)}[act]
  dh_ref = dl[:, None] * wf[:, 0][None, :] * dgrad
  dw_ref, db_ref = hf.T @ dl, dl.sum()
  rows = L.odin_disc_head_rows(B, K)
  assert rows == (B + 7) // 8
  th, tw, tb, tdl = T(h), T(w), T(b), T(dl)
  ws = bk.zeros(4, dtype=torch.int32)
  res = []
  for _ in range(2):
    tlogit, tdlo, tout = bk.full((B,), float('nan')), bk.full((B,), float('nan')), bk.full((1,), float('nan'))
    tdh, slab = bk.full((B, K), float('nan')), bk.full((rows, K + 1), float('nan'))
    word, r_out = bk.zeros(2048, dtype=torch.int32), C.c_int(0)
    L.odin_disc_head_fwd_bwd(th.data_ptr(), tw.data_ptr(), tb.data_ptr(), tlogit.data_ptr(), mode,
                             tdl.data_ptr() if mode == 0 else None, tdlo.data_ptr() if mode == 1 else None, tout.data_ptr(),
                             _lib.ACT[act], tdh.data_ptr(), word.data_ptr(), slab.data_ptr(), C.byref(r_out), ws.data_ptr(),
                             B, K, None)
    assert r_out.value == rows and int(ws.abs().sum()) == 0
    res.append((tlogit.clone(), tout.clone(), tdh.clone(), slab.clone()))
  assert all(torch.equal(a, b_) for a, b_ in zip(*res))
  close(tlogit.cpu().numpy(), logit)
  assert abs(float(tout) - out_ref) <= 2e-6 * max(1.0, abs(out_ref))
  if mode == 1:
    close(tdlo.cpu().numpy() * B, dl * B)
  close(tdh.cpu().numpy() * B, dh_ref * B)
  g = reduce_slab(bk, slab, rows, K + 1)
  close(g[:K] * B / np.sqrt(K), dw_ref * B / np.sqrt(K))
  close(g[K:] * B, [db_ref * B])
  assert float(word.view(torch.float32).max()) == float(tdh.abs().max())
  # forward + loss only (evaluation)
  tout2 = bk.full((1,), float('nan'))
  L.odin_disc_head_fwd_bwd(th.data_ptr(), tw.data_ptr(), tb.data_ptr(), tlogit.data_ptr(), mode,
                           tdl.data_ptr() if mode == 0 else None, None, tout2.data_ptr(), _lib.ACT[act], None, None, None, None,
                           ws.data_ptr(), B, K, None)
  assert torch.equal(tout2, tout)


@pytest.mark.parametrize('kind,shape', [
    ('conv', (5, 8, 8, 64, 64, 4, 2)), ('conv', (16, 8, 8, 32, 64, 4, 2)), ('deconv', (6, 4, 4, 8, 64, 4, 2)),
    ('deconv', (257, 4, 4, 16, 64, 4, 2)), ('dense', (100, 256, 40)), ('dense', (64, 1024, 128)), ('conv', (3, 16, 16, 32, 32, 4, 2)),
    ('dense', (48, 256, 320)), ('dense', (128, 1000, 2)), ('dense', (64, 6, 1000)),
    # bwd_planes.hip: Conv2DTranspose over 32 output channels, weight + data gradient in one launch (rows of 32 / 16 / 8
    # pixels; 64 input channels = two workgroup columns; a batch that leaves the last workgroup short)
    ('deconv', (3, 32, 32, 32, 32, 4, 2)), ('deconv', (5, 16, 16, 64, 32, 4, 2)), ('deconv', (7, 8, 8, 32, 32, 4, 2)),
    # ... over 64 output channels: two passes inside the launch; its slab rows are partitioned like the data
    # gradient's tiles, not like odin_deconv2d_wgrad's (sums equal to rounding, not bit for bit)
    ('deconv', (5, 8, 8, 64, 64, 4, 2)), ('deconv', (3, 16, 16, 32, 64, 4, 2))])
def test_layer_bwd_in_one_call(bk, kind, shape):
  """odin_conv2d_bwd / odin_deconv2d_bwd / odin_dense_bwd = the weight gradient + the data gradient of a layer in
  one call (small layers: ONE launch shared by the two implicit-GEMM kernels): results identical, bit for bit, to
  the two separate calls -- the same workgroups run the same arithmetic."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(9)
  rows, rows2 = C.c_int(0), C.c_int(0)
  if kind == 'dense':
    B, K, N = shape
    x, w, dy, aux = (rng.standard_normal(s_) for s_ in ((B, K), (K, N), (B, N), (B, K)))
    tx, tw, tdy, taux = T(x), T(w / np.sqrt(K)), T(dy), T(aux)
    n = K * N + N
    dx1, dx2 = bk.full((B, K), float('nan')), bk.full((B, K), float('nan'))
    s1, s2 = bk.full((L.odin_max_slab_rows(), n), float('nan')), bk.full((L.odin_max_slab_rows(), n), float('nan'))
    L.odin_dense_wgrad(tx.data_ptr(), tdy.data_ptr(), s1.data_ptr(), C.byref(rows), B, K, N, None)
    L.odin_dense_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx1.data_ptr(), None, None, B, K, N, None)
    L.odin_dense_bwd(tx.data_ptr(), tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx2.data_ptr(), None, None,
                     s2.data_ptr(), C.byref(rows2), B, K, N, 1, 1, None, None, None)
    # the range word of dx: valid whatever family ran (the range contract, include/odin_hip.h) -- from the epilogue
    # where odin_dense_dgrad_keeps_range says so, by one counted extra pass otherwise
    word, dx3 = bk.zeros(2048, dtype=torch.int32), bk.full((B, K), float('nan'))
    f0 = L.odin_debug_absmax_fallbacks()
    L.odin_dense_bwd(tx.data_ptr(), tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx3.data_ptr(), None, None,
                     None, None, B, K, N, 0, 1, None, word.data_ptr(), None)
    assert torch.equal(dx3, dx2)
    kept = float(word.view(torch.float32).max())
    assert kept == float(dx2.abs().max())
    if L.odin_dense_dgrad_keeps_range(B, K, N):
      assert L.odin_debug_absmax_fallbacks() - f0 <= 1   # (at most the bound of a dy that came without a word)
  else:
    B, H, W, Ci, Co, K, S = shape
    if kind == 'conv':
      OH, pt, _ = vo.same_pads(H, K, S)
      OW, pl, _ = vo.same_pads(W, K, S)
      wshape = (K, K, Ci, Co)
    else:
      OH, OW = H * S, W * S
      _, pt, _ = vo.same_pads(OH, K, S)
      _, pl, _ = vo.same_pads(OW, K, S)
      wshape = (K, K, Co, Ci)
    d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
    x, w, dy, aux = (rng.standard_normal(s_) for s_ in ((B, H, W, Ci), wshape, (B, OH, OW, Co), (B, H, W, Ci)))
    tx, tw, tdy, taux = T(x), T(w * 0.1), T(dy), T(aux)
    n = K * K * Ci * Co + (Co if kind == 'conv' else 0)
    dx1, dx2 = bk.full((B, H, W, Ci), float('nan')), bk.full((B, H, W, Ci), float('nan'))
    s1, s2 = bk.full((L.odin_max_slab_rows(), n), float('nan')), bk.full((L.odin_max_slab_rows(), n), float('nan'))
    wg, dg, bw = ((L.odin_conv2d_wgrad, L.odin_conv2d_dgrad, L.odin_conv2d_bwd) if kind == 'conv' else
                  (L.odin_deconv2d_wgrad, L.odin_deconv2d_dgrad, L.odin_deconv2d_bwd))
    fused = kind == 'deconv' and Co in (32, 64) and Ci % 32 == 0
    if fused:
      # with the column sums of dx (the bias gradient of the layer below) and every range word
      words = bk.zeros(4 * 2048, dtype=torch.int32)
      wp = [words[i * 2048:].data_ptr() for i in range(4)]
      L.odin_absmax(tdy.data_ptr(), tdy.numel(), wp[0], None)
      L.odin_absmax(tx.data_ptr(), tx.numel(), wp[1], None)
      d.dy_amax, d.x_amax = wp[0], wp[1]
      c1, c2 = bk.full((L.odin_max_slab_rows() * 2, Ci), float('nan')), bk.full((L.odin_max_slab_rows() * 2, Ci), float('nan'))
      cr1, cr2 = C.c_int(0), C.c_int(0)
      d.dx_amax = wp[2]
      wg(tx.data_ptr(), tdy.data_ptr(), s1.data_ptr(), C.byref(rows), C.byref(d), None)
      dg(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx1.data_ptr(), c1.data_ptr(), C.byref(cr1), C.byref(d), None)
      assert L.odin_debug_last_path().decode().startswith('fconv_planes')
      d.dx_amax = wp[3]
      bw(tx.data_ptr(), tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx2.data_ptr(), c2.data_ptr(), C.byref(cr2),
         s2.data_ptr(), C.byref(rows2), C.byref(d), None)
      assert L.odin_debug_last_path().decode().startswith('bwd_planes')
      assert cr1.value == cr2.value > 0 and torch.equal(c1[:cr1.value], c2[:cr2.value])
      assert torch.equal(words[2 * 2048:3 * 2048], words[3 * 2048:])
      assert float(words[3 * 2048:].view(torch.float32).max()) == float(dx2.abs().max())
      # ... and against the float64 oracle
      dx_ref, dw_ref, _ = vo.conv2d_transpose_bwd(x, w * 0.1, dy, S)
      dx_ref = dx_ref * np.where(aux > 0, 1.0, np.minimum(aux, 0.0) + 1.0)
      close(dx2.cpu().numpy(), dx_ref, 1e-4)
      close(reduce_slab(bk, s2, rows2.value, n).reshape(wshape), dw_ref, 1e-4)
    else:
      wg(tx.data_ptr(), tdy.data_ptr(), s1.data_ptr(), C.byref(rows), C.byref(d), None)
      dg(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx1.data_ptr(), None, None, C.byref(d), None)
      bw(tx.data_ptr(), tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx2.data_ptr(), None, None,
         s2.data_ptr(), C.byref(rows2), C.byref(d), None)
  path = L.odin_debug_last_path().decode()
  print(kind, shape, path)
  assert torch.equal(dx1, dx2)
  if kind == 'deconv' and Co == 64 and fused:
    assert rows.value > 0 and rows2.value > 0
    g1, g2 = reduce_slab(bk, s1, rows.value, n), reduce_slab(bk, s2, rows2.value, n)
    assert float(np.abs(g1 - g2).max()) <= 2e-6 * float(np.abs(g1).max())
  else:
    assert rows.value == rows2.value and rows.value > 0
    assert torch.equal(s1[:rows.value], s2[:rows.value])


@pytest.mark.parametrize('kind,shape,colsum', [
    # the advisor's round-4 repro: Conv2D 4x4 s1 32->32 on 32x32 at B=32 below a Conv2DTranspose (column-sum slab):
    # 32768 tiles > ODIN_MAX_COLSUM_BLOCKS sends the data gradient to the generic gather kernel
    ('conv', (32, 32, 32, 32, 32, 4, 1), True),
    ('conv', (32, 32, 32, 32, 32, 4, 1), False),
    ('conv', (4, 16, 16, 32, 64, 4, 2), False),      # tconv_planes
    ('conv', (3, 14, 14, 32, 64, 5, 2), True),       # MNIST's 5x5 stack
    ('conv', (2, 28, 28, 32, 32, 5, 1), True),
    ('conv', (6, 8, 8, 64, 64, 4, 2), False),        # igemm
    ('conv', (2, 64, 64, 32, 3, 1, 1), True),        # 1x1 head: pw1x1
    ('deconv', (4, 16, 16, 32, 32, 4, 2), True),     # fconv_planes
    ('deconv', (5, 7, 7, 64, 32, 5, 2), True),
    ('deconv', (3, 8, 8, 16, 24, 3, 1), False),      # generic
    ('deconv', (2, 4, 4, 8, 64, 4, 2), False)])
def test_data_gradient_range_contract(bk, kind, shape, colsum):
  """The range contract (include/odin_hip.h, round 5): a data gradient that is handed a dx_amax word leaves a bound
  >= max|dx| in it whatever kernel family the dispatch picked -- with and without a column-sum slab (the
  predicate said 'kept' where a slab sent the layer to a family that did not track, and the zero word overflowed the
  plane kernels below).  Where the predicate says 'kept without an extra pass' no absmax pass may run."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(77)
  B, H, W, Ci, Co, K, S = shape
  if kind == 'conv':
    OH, pt, _ = vo.same_pads(H, K, S)
    OW, pl, _ = vo.same_pads(W, K, S)
    wshape = (K, K, Ci, Co)
  else:
    OH, OW = H * S, W * S
    _, pt, _ = vo.same_pads(OH, K, S)
    _, pl, _ = vo.same_pads(OW, K, S)
    wshape = (K, K, Co, Ci)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  w, dy, aux = (rng.standard_normal(s_) for s_ in (wshape, (B, OH, OW, Co), (B, H, W, Ci)))
  tw, tdy, taux = T(w * 0.1), T(dy * 3e-4), T(aux)
  dyw, dxw = bk.zeros(2048, dtype=torch.int32), bk.zeros(2048, dtype=torch.int32)
  L.odin_absmax(tdy.data_ptr(), tdy.numel(), dyw.data_ptr(), None)
  d.dy_amax, d.dx_amax = dyw.data_ptr(), dxw.data_ptr()
  dx = bk.full((B, H, W, Ci), float('nan'))
  slab = bk.full((L.odin_max_slab_rows(), Ci), float('nan')) if colsum else None
  rows = C.c_int(0)
  dg = L.odin_conv2d_dgrad if kind == 'conv' else L.odin_deconv2d_dgrad
  keeps = (L.odin_conv2d_dgrad_keeps_range if kind == 'conv' else L.odin_deconv2d_dgrad_keeps_range)(C.byref(d), 1)
  f0 = L.odin_debug_absmax_fallbacks()
  dg(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), 1, dx.data_ptr(), slab.data_ptr() if colsum else None,
     C.byref(rows), C.byref(d), None)
  path = L.odin_debug_last_path().decode()
  passes = L.odin_debug_absmax_fallbacks() - f0
  bound = float(dxw.view(torch.float32).max())
  m = float(dx.abs().max())
  print(kind, shape, colsum, path, 'keeps', keeps, 'passes', passes, bound, m)
  assert np.isfinite(m) and m > 0
  assert bound >= m and bound <= 1.0001 * m, (bound, m)
  if passes == 0:
    assert path != 'absmax'
  if colsum:
    assert 0 < rows.value <= L.odin_max_slab_rows()


@pytest.mark.parametrize('scale', [1e5, 1e-30, 3e9, 1.0])
@pytest.mark.parametrize('family', ['fconv_planes', 'tconv_planes', 'igemm_h', 'dense_h'])
def test_plane_kernels_beyond_f16_range(bk, request, family, scale):
  """activations beyond what an f16 plane holds (|x| > 65504) or far below it (1e-30) through the
  forward and weight-gradient launches of every two-plane family, against the float64 oracle at 1e-4 of the result's
  maximum.  The layer input comes with its range word (odin_conv_desc.x_amax / odin_dense_*_ranged): the plane
  kernels carry it times the exact power of two that brings its bound to [2^14, 2^15) -- fp32 has no such limit
  (image_networks.py:157-174 are plain fp32 Keras layers), so neither may this.  scale 1.0: the word lies inside the
  safe window and the kernel takes its unscaled body (same result)."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(13)

  def word_of(t):
    w = bk.zeros(2048, dtype=torch.int32)
    L.odin_absmax(t.data_ptr(), t.numel(), w.data_ptr(), None)
    return w

  def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()

  if family == 'dense_h':
    B, K, N = 32, 256, 256
    x = rng.standard_normal((B, K)) * scale
    w, b = rng.standard_normal((K, N)) / np.sqrt(K), rng.standard_normal(N) * 0.1 * scale
    dy = rng.standard_normal((B, N)) * 1e-3
    tx, tw, tb, tdy = T(x), T(w), T(b), T(dy)
    xw, yw = word_of(tx), bk.zeros(2048, dtype=torch.int32)
    ty = bk.full((B, N), float('nan'))
    L.odin_dense_fwd_ranged(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), B, K, N, 0, xw.data_ptr(),
                            yw.data_ptr(), None)
    assert 'dense_h' in L.odin_debug_last_path().decode()
    xf = x.astype(np.float32).astype(np.float64)
    y_ref = vo.dense(xf, w.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64))
    assert rel(ty.cpu().numpy(), y_ref) <= 1e-4
    assert float(yw.view(torch.float32).max()) == float(ty.abs().max())
    rows = C.c_int(0)
    slab = bk.full((L.odin_max_slab_rows(), K * N + N), float('nan'))
    L.odin_dense_bwd_ranged(tx.data_ptr(), tdy.data_ptr(), None, None, 0, None, None, None, slab.data_ptr(),
                            C.byref(rows), B, K, N, 1, 0, None, None, xw.data_ptr(), None)
    assert 'dense_h' in L.odin_debug_last_path().decode()
    g = reduce_slab(bk, slab, rows.value, K * N + N)
    _, dw_ref, _ = vo.dense_bwd(xf, w, dy.astype(np.float32).astype(np.float64), need_dx=False)
    assert rel(g[:K * N].reshape(K, N), dw_ref) <= 1e-4
    return

  if family == 'igemm_h':
    request.addfinalizer(lambda old=L.odin_debug_igemm_h_min_flop(0.0): L.odin_debug_igemm_h_min_flop(old))
    kind, (B, H, W, Ci, Co, K, S) = 'conv', (2, 12, 10, 32, 32, 4, 2)   # (rows of 10 / 5 pixels: no plane kernel)
  elif family == 'fconv_planes':
    kind, (B, H, W, Ci, Co, K, S) = 'conv', (2, 32, 32, 32, 32, 4, 2)
  else:
    kind, (B, H, W, Ci, Co, K, S) = 'deconv', (2, 16, 16, 32, 32, 4, 2)
  if kind == 'conv':
    OH, pt, _ = vo.same_pads(H, K, S)
    OW, pl, _ = vo.same_pads(W, K, S)
    wshape = (K, K, Ci, Co)
  else:
    OH, OW = H * S, W * S
    _, pt, _ = vo.same_pads(OH, K, S)
    _, pl, _ = vo.same_pads(OW, K, S)
    wshape = (K, K, Co, Ci)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  x = rng.standard_normal((B, H, W, Ci)) * scale
  w, b = rng.standard_normal(wshape) * 0.1, rng.standard_normal(Co) * 0.1 * scale
  if scale < 1e-10:
    # (the kernels' ELU is exp2(t log2 e) - 1, 6e-8 absolute -- nothing at unit scale, everything at 1e-30: keep the
    # pre-activations of this case positive, where ELU is the identity)
    b = np.abs(b) + 40.0 * scale
  dy = rng.standard_normal((B, OH, OW, Co)) * 1e-3
  tx, tw, tb, tdy = T(x), T(w), T(b), T(dy)
  xw, yw, dyw = word_of(tx), bk.zeros(2048, dtype=torch.int32), word_of(tdy)
  d.x_amax, d.y_amax, d.dy_amax = xw.data_ptr(), yw.data_ptr(), dyw.data_ptr()
  ty = bk.full((B, OH, OW, Co), float('nan'))
  fwd = L.odin_conv2d_fwd if kind == 'conv' else L.odin_deconv2d_fwd
  fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
  path = L.odin_debug_last_path().decode()
  assert family in path, path
  xf, wf, bf = (a.astype(np.float32).astype(np.float64) for a in (x, w, b))
  pre = (vo.conv2d if kind == 'conv' else vo.conv2d_transpose)(xf, wf, bf, S)
  y_ref = vo.elu(pre)
  assert rel(ty.cpu().numpy(), y_ref) <= 1e-4, (path, rel(ty.cpu().numpy(), y_ref))
  assert float(yw.view(torch.float32).max()) == float(ty.abs().max())
  # the weight gradient: the activation is its other operand
  wg = L.odin_conv2d_wgrad if kind == 'conv' else L.odin_deconv2d_wgrad
  n = K * K * Ci * Co + (Co if kind == 'conv' else 0)
  rows = C.c_int(0)
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  wg(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  wpath = L.odin_debug_last_path().decode()
  assert '(f16x2)' in wpath, wpath
  g = reduce_slab(bk, slab, rows.value, n)
  dyf = dy.astype(np.float32).astype(np.float64)
  dw_ref = (vo.conv2d_bwd if kind == 'conv' else vo.conv2d_transpose_bwd)(xf, wf, dyf, S, need_dx=False)[1]
  assert rel(g[:K * K * Ci * Co].reshape(wshape), dw_ref) <= 1e-4, (wpath, rel(g[:K * K * Ci * Co].reshape(wshape), dw_ref))


def test_deferred_plane_weight_gradients(bk):
  """odin_wgrad_planes_defer_begin / _end: the plane weight gradients of several layers as ONE launch, bit-identical
  to the separate launches; odin_slab_reduce flushes what is still pending."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(31)
  cases = [('conv', (2, 64, 64, 32, 32)), ('conv', (3, 32, 32, 32, 64)), ('deconv', (3, 8, 8, 64, 32)),
           ('deconv', (2, 16, 16, 32, 32)), ('conv', (3, 16, 16, 64, 64))]
  jobs, words = [], []
  for kind, (B, H, W, Ci, Co) in cases:
    K, S = 4, 2
    if kind == 'conv':
      OH, pt, _ = vo.same_pads(H, K, S)
      OW, pl, _ = vo.same_pads(W, K, S)
      n = K * K * Ci * Co + Co
    else:
      OH, OW = H * S, W * S
      _, pt, _ = vo.same_pads(OH, K, S)
      _, pl, _ = vo.same_pads(OW, K, S)
      n = K * K * Ci * Co
    d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
    x, dy = T(rng.standard_normal((B, H, W, Ci))), T(rng.standard_normal((B, OH, OW, Co)) * 1e-3)
    # (the caller's own range word of dy: a job that would need a library scratch word is never deferred)
    word = bk.zeros(2048, dtype=torch.int32)
    L.odin_absmax(dy.data_ptr(), dy.numel(), word.data_ptr(), None)
    d.dy_amax = word.data_ptr()
    words.append(word)
    fn = L.odin_conv2d_wgrad if kind == 'conv' else L.odin_deconv2d_wgrad
    s1 = bk.full((L.odin_max_slab_rows(), n), float('nan'))
    s2 = bk.full((L.odin_max_slab_rows(), n), float('nan'))
    jobs.append((fn, x, dy, d, s1, s2, n))
  rows1, rows2 = [], []
  for fn, x, dy, d, s1, s2, n in jobs:
    r = C.c_int(0)
    fn(x.data_ptr(), dy.data_ptr(), s1.data_ptr(), C.byref(r), C.byref(d), None)
    assert L.odin_debug_last_path().decode() == 'wgrad_planes(f16x2)'
    rows1.append(r.value)
  L.odin_wgrad_planes_defer_begin()
  for fn, x, dy, d, s1, s2, n in jobs:
    r = C.c_int(0)
    fn(x.data_ptr(), dy.data_ptr(), s2.data_ptr(), C.byref(r), C.byref(d), None)
    rows2.append(r.value)
  assert all(bool(torch.isnan(j[5][0, 0])) for j in jobs)   # nothing has run yet
  L.odin_wgrad_planes_defer_end(None)
  assert L.odin_debug_last_path().decode() == 'wgrad_planes_multi(f16x2)'
  assert rows1 == rows2
  for (fn, x, dy, d, s1, s2, n), r in zip(jobs, rows1):
    assert torch.equal(s1[:r], s2[:r])
  # a pending job is flushed by the slab reduction that reads it
  fn, x, dy, d, s1, s2, n = jobs[0]
  s3, out = bk.full((L.odin_max_slab_rows(), n), float('nan')), bk.zeros(n)
  r = C.c_int(0)
  L.odin_wgrad_planes_defer_begin()
  fn(x.data_ptr(), dy.data_ptr(), s3.data_ptr(), C.byref(r), C.byref(d), None)
  job = (_lib.ReduceJob * 1)(_lib.ReduceJob(s3.data_ptr(), out.data_ptr(), n, r.value, n, 0))
  L.odin_slab_reduce(job, 1, None)
  L.odin_wgrad_planes_defer_end(None)
  assert torch.equal(s3[:r.value], s1[:r.value]) and bool(torch.isfinite(out).all())


@pytest.fixture(scope='module')
def hipbk():
  import torch
  from tests.conftest import Backend
  assert torch.cuda.is_available(), 'the hip backend needs an MI355X'
  return Backend('hip', _lib.load(), 'cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('B,K,N,act,family', [
    # FactorVAE's discriminator layers at BASELINE config 3's half batches (factor_vae.py:150-153) and the whole
    # batch of the discriminator step, MNIST's dense stack at batch 128 (variational_autoencoder.py:181-185),
    # CelebA's encoder head: the instances the benchmarks run, held to the float64 oracle (too large for the
    # CPU simulator) -- `family` is the kernel family the dispatcher must have picked (both widths >= 256: the
    # two-plane GEMM of dense_h.hip; narrower layers: the fp32 implicit GEMM)
    (128, 1000, 1000, 'relu', 'dense_h'), (256, 1000, 1000, 'relu', 'dense_h'), (128, 6, 1000, 'relu', 'thin_dense'),
    (256, 6, 1000, 'relu', 'thin_dense'), (128, 1000, 1, 'linear', 'thin_dense'), (256, 1000, 1, 'linear', 'thin_dense'), (128, 784, 512, 'relu', 'dense_h'), (128, 512, 784, 'linear', 'dense_h'),
    (128, 512, 512, 'relu', 'dense_h'), (512, 4096, 512, 'linear', 'dense_h'), (256, 1024, 256, 'linear', 'dense_h'),
    (256, 1024, 128, 'linear', 'igemm')])
def test_dense_at_benchmark_sizes(hipbk, B, K, N, act, family):
  bk = hipbk
  L, T = bk.L, bk.T
  rng = np.random.default_rng(4)
  x = rng.standard_normal((B, K))
  w = rng.standard_normal((K, N)) / np.sqrt(K)
  b = rng.standard_normal(N) * 0.1
  y_ref = vo._ACT[act](vo.dense(x, w, b))
  ty = bk.full((B, N), float('nan'))
  tx, tw, tb = T(x), T(w), T(b)
  paths = []
  L.odin_dense_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), B, K, N, _lib.ACT[act], None)
  paths.append(L.odin_debug_last_path().decode())
  close(ty.cpu().numpy(), y_ref)
  dy = rng.standard_normal((B, N))
  aux = rng.standard_normal((B, K))
  tdy, taux = T(dy), T(aux)
  dx_ref, dw_ref, db_ref = vo.dense_bwd(x, w, dy)
  tdx = bk.full((B, K), float('nan'))
  rows = C.c_int(0)
  L.odin_dense_dgrad(tdy.data_ptr(), tw.data_ptr(), taux.data_ptr(), _lib.ACT['relu'], tdx.data_ptr(),
                     None, None, B, K, N, None)
  paths.append(L.odin_debug_last_path().decode())
  close(tdx.cpu().numpy(), dx_ref * (aux.astype(np.float32) > 0))
  n = K * N + N
  slab = bk.full((L.odin_max_slab_rows(), n), float('nan'))
  L.odin_dense_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), B, K, N, None)
  paths.append(L.odin_debug_last_path().decode())
  g = reduce_slab(bk, slab, rows.value, n)
  close(g[:-N].reshape(K, N), dw_ref, 1e-4)
  close(g[-N:], db_ref, 1e-4)
  print('dense', (B, K, N), paths)
  if family is not None:
    assert all(family in q for q in paths), paths


@pytest.mark.parametrize('is_deconv,B,H,W,Ci,Co,K,S,C1', [
    (1, 2, 8, 8, 32, 32, 4, 2, 1),     # specialised <T,4,2,32> instance
    (1, 1, 16, 16, 32, 32, 4, 2, 1),   # two-workgroup fused tail instance (also the shape of the opt-in bf16-plane path)
    (1, 3, 8, 8, 8, 16, 4, 2, 3),      # generic transposed
    (0, 2, 16, 16, 8, 24, 5, 1, 1),    # generic gather conv (MNIST-style decoder tail)
    (1, 3, 16, 16, 32, 32, 4, 2, 1),   # tconv_planes fused tail (bf16 planes), 16-pixel rows, tiles across image seams
    (1, 3, 16, 16, 32, 32, 4, 2, 3),   # 3 logit maps, 16-pixel rows, tiles across image seams (tconv_planes; tconv_ring when opted in)
    (1, 2, 32, 32, 32, 32, 4, 2, 1),   # tconv_planes fused tail, 32-pixel input rows (dSprites decoder4)
    (1, 2, 32, 32, 32, 32, 4, 2, 3),   # tconv_planes fused tail with 3 logit maps (Shapes3D / CelebA decoder4 -> 64x64x3)
])
def test_bernoulli_tail(bk, is_deconv, B, H, W, Ci, Co, K, S, C1):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(3)
  x = rng.standard_normal((B, H, W, Ci))
  b = rng.standard_normal(Co) * 0.1
  w1 = rng.standard_normal((1, 1, Co, C1)) * 0.3
  b1 = rng.standard_normal(C1) * 0.1
  if is_deconv:
    w = rng.standard_normal((K, K, Co, Ci)) * 0.2
    OH, OW = H * S, W * S
    _, pt, _ = vo.same_pads(OH, K, S)
    _, pl, _ = vo.same_pads(OW, K, S)
    d = vo.elu(vo.conv2d_transpose(x, w, b, S))
  else:
    w = rng.standard_normal((K, K, Ci, Co)) * 0.2
    OH, pt, _ = vo.same_pads(H, K, S)
    OW, pl, _ = vo.same_pads(W, K, S)
    d = vo.elu(vo.conv2d(x, w, b, S))
  desc = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  tgt = np.clip(rng.random((B, OH, OW, C1)), 1e-6, 1 - 1e-6)
  scale = 1.0 / B
  lg = vo.conv2d(d, w1, b1, 1)
  llk_ref = vo.bernoulli_log_prob(lg, tgt)
  dl = -(vo.bernoulli_log_prob_grad(lg, tgt)) * scale
  dd, dw1_ref, db1_ref = vo.conv2d_bwd(d, w1, dl, 1)
  g_ref = dd * vo.elu_grad_from_output(d)
  tx, tw, tb, tw1, tb1, tt, tsc = T(x), T(w), T(b), T(w1), T(b1), T(tgt), T([scale])
  logits = bk.full((B, OH, OW, C1), float('nan'))
  g = bk.full((B, OH, OW, Co), float('nan'))
  rows, npart = C.c_int(0), C.c_int(0)
  L.odin_bernoulli_tail_fwd_bwd(is_deconv, None, None, None, None, None, None, None, None, None,
                                C.byref(npart), None, C.byref(rows), None, C.byref(desc), C1, None)
  part = bk.full((B * npart.value,), float('nan'))
  n = Co * C1 + C1 + Co
  slab = bk.full((rows.value, n), float('nan'))
  L.odin_bernoulli_tail_fwd_bwd(is_deconv, tx.data_ptr(), tw.data_ptr(), tb.data_ptr(),
                                tw1.data_ptr(), tb1.data_ptr(), tt.data_ptr(), logits.data_ptr(),
                                g.data_ptr(), part.data_ptr(), C.byref(npart), slab.data_ptr(),
                                C.byref(rows), tsc.data_ptr(), C.byref(desc), C1, None)
  close(logits.cpu().numpy(), lg)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref)
  close(g.cpu().numpy(), g_ref)
  red = reduce_slab(bk, slab, rows.value, n)
  close(red[:Co * C1].reshape(Co, C1), dw1_ref[0, 0], 1e-4)
  close(red[Co * C1:Co * C1 + C1], db1_ref, 1e-4)
  close(red[Co * C1 + C1:], g_ref.sum((0, 1, 2)), 1e-4)


def test_exact_fp32_switch_matches_the_plane_kernels(bk, monkeypatch):
  """ODIN_EXACT_FP32 (the one arithmetic switch of the product library, odin_internal.h) routes every convolution
  through the exact fp32 matrix-core kernels; the default carries the 4x4/s2 32-channel layers through the f16 matrix
  pipe as two planes (3 MFMAs per 16 k-values, <= 3 * 2^-22 per product): results must agree to fp32-class accuracy
  (<= 4e-6 of the tensor maximum) and differ in their rounding (the plane path really ran)."""
  import os
  L, T = bk.L, bk.T
  rng = np.random.default_rng(11)
  B, H, W, Ci, Co, K, S = 4, 16, 16, 32, 32, 4, 2
  OH, OW = H * S, W * S
  _, pt, _ = vo.same_pads(OH, K, S)
  _, pl, _ = vo.same_pads(OW, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu')
  tx, tw = T(rng.standard_normal((B, H, W, Ci))), T(rng.standard_normal((K, K, Co, Ci)) * 0.1)
  tb = T(rng.standard_normal(Co) * 0.1)
  outs, paths = [], []
  for flag in (None, '1'):
    if flag is None:
      monkeypatch.delenv('ODIN_EXACT_FP32', raising=False)
      os.unsetenv('ODIN_EXACT_FP32')
    else:
      monkeypatch.setenv('ODIN_EXACT_FP32', flag)
      os.putenv('ODIN_EXACT_FP32', flag)
    ty = bk.zeros(B, OH, OW, Co)
    L.odin_deconv2d_fwd(tx.data_ptr(), tw.data_ptr(), tb.data_ptr(), ty.data_ptr(), C.byref(d), None)
    outs.append(ty.cpu().numpy().copy())
    paths.append(L.odin_debug_last_path().decode())
  os.unsetenv('ODIN_EXACT_FP32')
  assert paths[0].endswith('(f16x2)') and not paths[1].endswith('(f16x2)'), paths
  assert np.abs(outs[0]).max() > 0.5
  assert np.abs(outs[0] - outs[1]).max() <= 4e-6 * np.abs(outs[0]).max()
  assert not np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize('B,H,W,Ci,Co', [(2, 16, 16, 32, 48), (1, 32, 32, 32, 80), (3, 16, 16, 32, 32),
                                         (2, 64, 64, 32, 32)])
def test_wgrad_dry_run_reports_the_rows_the_launch_writes(bk, B, H, W, Ci, Co):
  """The slab is sized from the dry run (NULL slab pointer): the real launch must pick the same
  plan -- also for channel counts that disqualify the producer/consumer kernel (Cout % 32 != 0)."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(5)
  K, S = 4, 2
  OH, pt, _ = vo.same_pads(H, K, S)
  OW, pl, _ = vo.same_pads(W, K, S)
  d = _lib.conv_desc(B, H, W, Ci, OH, OW, Co, K, S, pt, pl, 'elu', False)
  dry = C.c_int(0)
  L.odin_conv2d_wgrad(None, None, None, C.byref(dry), C.byref(d), None)
  x, dy = rng.random((B, H, W, Ci)), rng.standard_normal((B, OH, OW, Co))
  n = K * K * Ci * Co + Co
  slab = bk.full((dry.value, n), float('nan'))   # exactly the rows the dry run promised
  rows = C.c_int(0)
  tx, tdy = T(x), T(dy)
  L.odin_conv2d_wgrad(tx.data_ptr(), tdy.data_ptr(), slab.data_ptr(), C.byref(rows), C.byref(d), None)
  assert rows.value == dry.value
  _, dw_ref, db_ref = vo.conv2d_bwd(x, rng.standard_normal((K, K, Ci, Co)), dy, S)
  g = reduce_slab(bk, slab, rows.value, n)
  close(g[:-Co].reshape(K, K, Ci, Co), dw_ref, 1e-4)
  close(g[-Co:], db_ref, 1e-4)


def test_slab_reduce_sumsq(bk):
  """odin_slab_reduce_sumsq (the gradient norm's stage-1 launch inside the slab reduction): results bit-identical to
  odin_slab_reduce; the partials -- one per ACTIVE workgroup of the jobs that write into the gradient buffer -- sum to
  the squared norm of what was written; a job outside the gradient buffer (the range-word reset) is reduced but not
  counted; the dry run reports the same number of partials; the staging copy arrives."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(11)
  # (n, rows): wide vector path, narrow vector path, odd sizes, a single-row job, an empty job, a zero-row reset
  shapes = [(16 * 32 * 32, 37), (2048, 256), (65, 100), (33, 7), (4096 + 64, 1), (24, 3)]
  total = sum(n for n, _ in shapes)
  g1, g2 = bk.full((total + 3,), float('nan')), bk.full((total + 3,), float('nan'))
  other1, other2 = bk.full((512,), float('nan')), bk.full((512,), float('nan'))
  slabs, jobs1, jobs2, off = [], [], [], 0
  for n, rows in shapes:
    stride = n + (4 if n % 4 == 0 else 3)
    sl = T(rng.standard_normal((rows, stride)))
    slabs.append(sl)
    for jobs, g in ((jobs1, g1), (jobs2, g2)):
      jobs.append(_lib.ReduceJob(sl.data_ptr(), g[off:].data_ptr(), n, rows, stride, 0))
    off += n
  zsl = T(np.zeros((1, 512)))
  jobs1.append(_lib.ReduceJob(zsl.data_ptr(), other1.data_ptr(), 512, 0, 512, 0))
  jobs2.append(_lib.ReduceJob(zsl.data_ptr(), other2.data_ptr(), 512, 0, 512, 0))
  a1 = (_lib.ReduceJob * len(jobs1))(*jobs1)
  a2 = (_lib.ReduceJob * len(jobs2))(*jobs2)
  L.odin_slab_reduce(a1, len(jobs1), None)
  nd, nr = C.c_int(0), C.c_int(0)
  L.odin_slab_reduce_sumsq(a2, len(jobs2), g2.data_ptr(), total, None, C.byref(nd), None, None, 0, None)
  assert 0 < nd.value < 4096 and torch.isnan(g2).all()   # (dry run: nothing written)
  part = bk.full((nd.value + 8,), float('nan'))
  src, dst = T(rng.standard_normal(24)), bk.full((32,), float('nan'))
  L.odin_slab_reduce_sumsq(a2, len(jobs2), g2.data_ptr(), total, part.data_ptr(), C.byref(nr), src.data_ptr(),
                           dst.data_ptr(), 24, None)
  assert nr.value == nd.value
  assert torch.equal(g1[:total], g2[:total]) and torch.isnan(g2[total:]).all()
  assert torch.equal(other1, other2) and float(other2.abs().max()) == 0.0
  assert torch.equal(dst[:24], src) and torch.isnan(dst[24:]).all()
  assert torch.isfinite(part[:nr.value]).all() and torch.isnan(part[nr.value:]).all()
  want = float((g2[:total].double() ** 2).sum())
  got = float(part[:nr.value].double().sum())
  assert abs(got - want) <= 1e-6 * want, (got, want)
