"""CPU ORACLE (test infrastructure, NOT product code) for the VAE training step.

A float64 numpy restatement of the reference's VAE hot path, with a hand-written
backward pass and Keras-Adam.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
(``odin_ai_amd``) never does.

PARITY STATUS: **parity unpinned** against the TensorFlow reference for the VAE
arithmetic -- the reference's own tests hold no golden vectors for ELBO / KL /
log-prob / conv stacks / Adam and TensorFlow 2.5 / TFP 0.13 are
not installable here, so the reference cannot be executed.  The oracle is instead
(1) cross-checked against an independent torch-autograd restatement
(``oracle/torch_ref.py``), (2) pinned on closed-form known answers
(tests/test_oracle_kat.py) and (3) pinned on the portable *properties* of the only
hot-path test the reference has (``tests/bayesian/test_vae.py:112-125``,
permute_dims).  The mel front-end oracle (``oracle/mel_oracle.py``) IS pinned against
the reference's own numpy code executed in the build container.

All file:line citations are relative to /root/reference.

Conventions: activations NHWC; Conv2D kernel (kh,kw,Cin,Cout);
Conv2DTranspose kernel (kh,kw,Cout,Cin); Dense kernel (in,out); Flatten is row-major
over (H,W,C); TF ``SAME`` padding.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F64 = np.float64
LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# padding / activations
# --------------------------------------------------------------------------------------
def same_pads(n: int, k: int, s: int) -> Tuple[int, int, int]:
  """TF ``SAME``: returns (out, pad_before, pad_after).  Keras Conv2D(padding='same')
  as used by odin/networks/image_networks.py:157-174."""
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  before = total // 2
  return out, before, total - before


def elu(x):
  """tf.nn.elu (odin/networks/image_networks.py:157)."""
  return np.where(x > 0, x, np.expm1(np.minimum(x, 0.0)))


def elu_grad_from_output(y):
  """d elu / d pre-activation expressed from the OUTPUT y: 1 if y>0 else y+1."""
  return np.where(y > 0, 1.0, y + 1.0)


def relu(x):
  return np.maximum(x, 0.0)


def softplus(x):
  """tf.nn.softplus (odin/bay/layers/continuous.py:478-479)."""
  return np.logaddexp(0.0, x)


def sigmoid(x):
  return np.where(x >= 0, 1.0 / (1.0 + np.exp(-np.abs(x))),
                  np.exp(-np.abs(x)) / (1.0 + np.exp(-np.abs(x))))


SOFTPLUS_INV_1 = math.log(math.e - 1.0)  # softplus^-1(1) = 0.541324...


def softplus1(x):
  """odin/backend/maths.py:279-281: softplus(x + softplus_inverse(1))."""
  return softplus(x + SOFTPLUS_INV_1)


_ACT = {'linear': (lambda x: x), 'elu': elu, 'relu': relu}


def act_grad_from_output(name: str, y):
  if name == 'linear':
    return np.ones_like(y)
  if name == 'elu':
    return elu_grad_from_output(y)
  if name == 'relu':
    return (y > 0).astype(F64)
  raise ValueError(name)


# --------------------------------------------------------------------------------------
# conv / deconv / dense, forward + backward
# --------------------------------------------------------------------------------------
def _windows(xp, kh, kw, s, oh, ow):
  """[B,OH,OW,kh,kw,C] view of the padded input."""
  B, Hp, Wp, C = xp.shape
  sb, sh, sw, sc = xp.strides
  return np.lib.stride_tricks.as_strided(
      xp, shape=(B, oh, ow, kh, kw, C),
      strides=(sb, sh * s, sw * s, sh, sw, sc), writeable=False)


def conv2d(x, w, b, stride: int):
  """keras.layers.Conv2D(padding='same') forward (pre-activation).
  x [B,H,W,Cin], w [kh,kw,Cin,Cout], b [Cout] or None.
  Call sites: odin/networks/image_networks.py:166-169,463-466."""
  B, H, W, C = x.shape
  kh, kw, ci, co = w.shape
  assert ci == C
  oh, pt, pb = same_pads(H, kh, stride)
  ow, pl, pr = same_pads(W, kw, stride)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  cols = _windows(xp, kh, kw, stride, oh, ow).reshape(B * oh * ow, kh * kw * C)
  y = cols @ w.reshape(kh * kw * C, co)
  if b is not None:
    y = y + b
  return y.reshape(B, oh, ow, co)


def conv2d_bwd(x, w, dy, stride: int, need_dx: bool = True):
  """Gradients of conv2d wrt (x, w, b) given dy = dL/d(pre-activation)."""
  B, H, W, C = x.shape
  kh, kw, ci, co = w.shape
  oh, pt, pb = same_pads(H, kh, stride)
  ow, pl, pr = same_pads(W, kw, stride)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  cols = _windows(xp, kh, kw, stride, oh, ow).reshape(B * oh * ow, kh * kw * C)
  dy2 = dy.reshape(B * oh * ow, co)
  dw = (cols.T @ dy2).reshape(kh, kw, C, co)
  db = dy2.sum(0)
  dx = None
  if need_dx:
    dxp = np.zeros_like(xp)
    for i in range(kh):
      for j in range(kw):
        dxp[:, i:i + stride * oh:stride, j:j + stride * ow:stride, :] += dy @ w[i, j].T
    dx = dxp[:, pt:pt + H, pl:pl + W, :]
  return dx, dw, db


def conv2d_transpose(x, w, b, stride: int):
  """keras.layers.Conv2DTranspose(padding='same') forward (pre-activation).
  x [B,H,W,Cin], w [kh,kw,Cout,Cin].  out = H*stride; equals
  conv2d_backprop_input of a SAME conv on an input of size H*stride
  (call sites odin/networks/image_networks.py:170-173,497-505)."""
  B, H, W, C = x.shape
  kh, kw, co, ci = w.shape
  assert ci == C
  OH, OW = H * stride, W * stride
  _, pt, _ = same_pads(OH, kh, stride)
  _, pl, _ = same_pads(OW, kw, stride)
  full = np.zeros((B, (H - 1) * stride + kh, (W - 1) * stride + kw, co), F64)
  for i in range(kh):
    for j in range(kw):
      full[:, i:i + stride * H:stride, j:j + stride * W:stride, :] += x @ w[i, j].T
  # the full transposed conv may be smaller than pad+out when k < s (never here)
  y = full[:, pt:pt + OH, pl:pl + OW, :]
  assert y.shape[1] == OH and y.shape[2] == OW
  if b is not None:
    y = y + b
  return y


def conv2d_transpose_bwd(x, w, dy, stride: int, need_dx: bool = True):
  B, H, W, C = x.shape
  kh, kw, co, ci = w.shape
  OH, OW = H * stride, W * stride
  _, pt, _ = same_pads(OH, kh, stride)
  _, pl, _ = same_pads(OW, kw, stride)
  fh, fw = (H - 1) * stride + kh, (W - 1) * stride + kw
  dfull = np.zeros((B, fh, fw, co), F64)
  dfull[:, pt:pt + OH, pl:pl + OW, :] = dy
  dw = np.zeros_like(w)
  dx = np.zeros_like(x) if need_dx else None
  for i in range(kh):
    for j in range(kw):
      g = dfull[:, i:i + stride * H:stride, j:j + stride * W:stride, :]  # [B,H,W,co]
      dw[i, j] = np.tensordot(g, x, axes=([0, 1, 2], [0, 1, 2]))  # [co,ci]
      if need_dx:
        dx += g @ w[i, j]
  db = dy.sum((0, 1, 2))
  return dx, dw, db


def dense(x, w, b):
  """keras Dense: x [B,in] @ w [in,out] + b."""
  y = x @ w
  return y if b is None else y + b


def dense_bwd(x, w, dy, need_dx: bool = True):
  return (dy @ w.T if need_dx else None), x.T @ dy, dy.sum(0)


# --------------------------------------------------------------------------------------
# distributions
# --------------------------------------------------------------------------------------
def bernoulli_log_prob(logits, x):
  """Independent(Bernoulli(logits), 3).log_prob(x) -> [B]
  (odin/networks/image_networks.py:87-93, odin/bay/layers/discrete.py:63-98).
  = sum x*log sigmoid(l) + (1-x)*log sigmoid(-l) = sum [x*l - softplus(l)]."""
  e = x * logits - softplus(logits)
  return e.reshape(e.shape[0], -1).sum(1)


def bernoulli_log_prob_grad(logits, x):
  """d log_prob / d logits, elementwise: x - sigmoid(l)."""
  return x - sigmoid(logits)


def gaussian_log_prob(loc, scale, x):
  """Independent(Normal(loc, scale), n).log_prob(x) -> [B]
  (odin/networks/image_networks.py:95-102)."""
  e = -0.5 * ((x - loc) / scale) ** 2 - np.log(scale) - 0.5 * LOG2PI
  return e.reshape(e.shape[0], -1).sum(1)


# ---- MixtureQuantizedLogistic (odin/bay/distributions/quantized.py:206-349): PixelCNN++ mixture of
# K discretised logistics per pixel; with C > 1 channels the component means of channel i receive a
# linear term in the (transformed) values of the channels j < i.  `h` [..., K * n_out] is reshaped to
# [..., K, n_out] and split into (logit 1 | loc C | raw scale C | coefficient C (C - 1) / 2) (:266-282).
def mixql_n_out(C: int) -> int:
  return 2 * C + C * (C - 1) // 2 + 1


def mixql_split(h, C: int, K: int):
  no = mixql_n_out(C)
  hh = np.asarray(h, F64).reshape(h.shape[:-1] + (K, no))
  return hh[..., 0], hh[..., 1:1 + C], hh[..., 1 + C:1 + 2 * C], hh[..., 1 + 2 * C:]


def _mixql_locs(locs, coefs, xt, C):
  """loc_i += sum_{j < i} x_j * coef[count], count running over (i, j) in the reference's loop order
  (:315-320); x_j the pixel value mapped to [-1, 1] (`_switch_domain`, 'sigmoid': 2 v - 1)."""
  if C == 1:
    return locs
  out = [locs[..., i] for i in range(C)]
  cnt = 0
  for i in range(C):
    for j in range(i):
      out[i] = out[i] + xt[..., None, j] * coefs[..., cnt]
      cnt += 1
  return np.stack(out, -1)


def mixql_log_prob_pix(h, x, C: int, K: int = 10):
  """log-probability of every pixel [..]: logsumexp_k(log_softmax(logits)_k + sum_c log QL_kc(x_c))."""
  logits, locs, raw, coefs = mixql_split(h, C, K)
  x = np.asarray(x, F64)
  xt = 2.0 * x - 1.0
  le = _mixql_locs(locs, coefs, xt, C)
  q = qlogistic_log_prob_elem(le, raw, np.broadcast_to(x[..., None, :], le.shape))
  comp = logits - _logsumexp(logits, -1)[..., None] + q.sum(-1)
  return _logsumexp(comp, -1)


def mixql_log_prob(h, x, C: int, K: int = 10):
  e = mixql_log_prob_pix(h, x, C, K)
  return e.reshape(e.shape[0], -1).sum(1)


def mixql_log_prob_grad(h, x, C: int, K: int = 10):
  """d log_prob / d h, elementwise, same shape as h."""
  logits, locs, raw, coefs = mixql_split(h, C, K)
  x = np.asarray(x, F64)
  xt = 2.0 * x - 1.0
  le = _mixql_locs(locs, coefs, xt, C)
  xb = np.broadcast_to(x[..., None, :], le.shape)
  q = qlogistic_log_prob_elem(le, raw, xb)
  gl, gr = qlogistic_log_prob_grad(le, raw, xb)
  la = logits - _logsumexp(logits, -1)[..., None]
  comp = la + q.sum(-1)
  r = np.exp(comp - _logsumexp(comp, -1)[..., None])     # responsibilities
  g = np.zeros(h.shape[:-1] + (K, mixql_n_out(C)))
  g[..., 0] = r - np.exp(la)
  g[..., 1:1 + C] = r[..., None] * gl
  g[..., 1 + C:1 + 2 * C] = r[..., None] * gr
  cnt = 0
  for i in range(C):
    for j in range(i):
      g[..., 1 + 2 * C + cnt] = r * gl[..., i] * xt[..., None, j]
      cnt += 1
  return g.reshape(h.shape)


def mixql_mean(h, C: int, K: int = 10):
  """MixtureQuantizedLogistic._mean (:351-381): the channel chain runs on the component MEANS, the
  Shift(-0.5) of the base distribution stays, `_pixels_to`(sigmoid) divides by `high`."""
  logits, locs, raw, coefs = mixql_split(h, C, K)
  out = [locs[..., i] for i in range(C)]
  cnt = 0
  for i in range(C):
    for j in range(i):
      out[i] = out[i] + out[j] * coefs[..., cnt]
      cnt += 1
  m = 127.5 * (np.stack(out, -1) + 1.0) - 0.5
  pi = np.exp(logits - _logsumexp(logits, -1)[..., None])
  return (pi[..., None] * m).sum(-2) / 255.0


# ---- QuantizedLogistic (odin/bay/distributions/quantized.py:50-204; TFP 0.13
# QuantizedDistribution._log_prob_with_logsf_and_logcdf, third-party, restated from its published
# source): PixelCNN-style discretised logistic over the pixel values low..high ------------------
QL_MIN_SCALE = math.exp(-7.0)


def qlogistic_params(loc, raw, low=0.0, high=255.0):
  """image_networks.py:55-71 + quantized.py:104-108: scale = softplus(raw) + e^-7, then both are
  mapped to pixel units: m = low + (high-low)/2 * (loc + 1), s = scale * (high-low)/2."""
  support = 0.5 * (high - low)
  return low + support * (loc + 1.0), (softplus(raw) + QL_MIN_SCALE) * support


def _ql_terms(m, s, x, low, high):
  # y = x * high is formed in FLOAT32 by the reference (`_switch_domain`, 'sigmoid'): the
  # floor / ceil below are discontinuous, so the oracle rounds exactly where TF does
  y = (np.asarray(x, np.float32) * np.float32(high)).astype(F64)
  ninf = -np.inf

  def logcdf(j):  # P[Y <= j], QuantizedDistribution._log_cdf on X = Logistic - 0.5
    r = -softplus(-(j + 0.5 - m) / s)
    r = np.where(j < low, ninf, r)
    return np.where(j < high, r, 0.0)

  def logsf(j):   # P[Y > j], QuantizedDistribution._log_survival_function
    r = -softplus((j + 0.5 - m) / s)
    r = np.where(j < low, 0.0, r)
    return np.where(j < high, r, ninf)

  jf, jc = np.floor(y), np.ceil(y)
  logsf_y, logsf_y1 = logsf(jc), logsf(np.ceil(y - 1.0))
  logcdf_y, logcdf_y1 = logcdf(jf), logcdf(np.floor(y - 1.0))
  use_sf = logsf_y < logcdf_y
  return y, jf, jc, use_sf, logsf_y, logsf_y1, logcdf_y, logcdf_y1


def qlogistic_log_prob_elem(loc, raw, x, low=0.0, high=255.0):
  """Per-element log P[Y = y]: log(exp(big) - exp(small)), on the survival side right of the
  median and on the cdf side left of it (TFP's numerically safe choice)."""
  m, s = qlogistic_params(loc, raw, low, high)
  _, _, _, use_sf, lsy, lsy1, lcy, lcy1 = _ql_terms(m, s, x, low, high)
  big = np.where(use_sf, lsy1, lcy)
  small = np.where(use_sf, lsy, lcy1)
  with np.errstate(divide='ignore', invalid='ignore'):
    d = big - small  # >= 0 (inf when small = -inf)
    l1m = np.where(d < math.log(2.0), np.log(-np.expm1(-d)), np.log1p(-np.exp(-d)))
  return big + l1m


def qlogistic_log_prob(loc, raw, x, low=0.0, high=255.0):
  e = qlogistic_log_prob_elem(loc, raw, x, low, high)
  return e.reshape(e.shape[0], -1).sum(1)


def qlogistic_log_prob_grad(loc, raw, x, low=0.0, high=255.0):
  """(d log_prob / d loc, d log_prob / d raw), elementwise."""
  m, s = qlogistic_params(loc, raw, low, high)
  y, jf, jc, use_sf, lsy, lsy1, lcy, lcy1 = _ql_terms(m, s, x, low, high)
  support = 0.5 * (high - low)

  def d_logcdf(j):  # d/du of -softplus(-u) = sigmoid(-u); zero on the clamped branches
    u = (j + 0.5 - m) / s
    live = (j >= low) & (j < high)
    return np.where(live, sigmoid(-u), 0.0), u

  def d_logsf(j):   # d/du of -softplus(u) = -sigmoid(u)
    u = (j + 0.5 - m) / s
    live = (j >= low) & (j < high)
    return np.where(live, -sigmoid(u), 0.0), u

  gb_sf, ub_sf = d_logsf(np.ceil(y - 1.0))
  gs_sf, us_sf = d_logsf(jc)
  gb_cf, ub_cf = d_logcdf(jf)
  gs_cf, us_cf = d_logcdf(np.floor(y - 1.0))
  big = np.where(use_sf, lsy1, lcy)
  small = np.where(use_sf, lsy, lcy1)
  gb, ub = np.where(use_sf, gb_sf, gb_cf), np.where(use_sf, ub_sf, ub_cf)
  gs, us = np.where(use_sf, gs_sf, gs_cf), np.where(use_sf, us_sf, us_cf)
  with np.errstate(invalid='ignore'):
    r = np.where(np.isneginf(small), 0.0, np.exp(small - big))  # e^small / e^big in [0, 1)
  wb, ws = 1.0 / (1.0 - r), -r / (1.0 - r)  # d result / d big, d result / d small
  # du/dm = -1/s, du/ds = -u/s
  dm = (wb * gb + ws * gs) * (-1.0 / s)
  ds = (wb * gb * ub + ws * gs * us) * (-1.0 / s)
  return dm * support, ds * support * sigmoid(raw)


def mvn_diag_params(p, D: int):
  """MultivariateNormalLayer.new: loc = p[..., :D], scale = softplus(p[..., D:])
  (odin/bay/layers/continuous.py:459-483)."""
  return p[..., :D], softplus(p[..., D:])


def kl_mc(loc, scale, z):
  """kl_divergence(analytic=False): log q(z|x) - log p(z), p = N(0,I)
  (odin/bay/helpers.py:267-276; prior odin/bay/random_variable.py:80-86)."""
  D = loc.shape[-1]
  lq = (-0.5 * ((z - loc) / scale) ** 2 - np.log(scale)).sum(-1) - 0.5 * D * LOG2PI
  lp = (-0.5 * z ** 2).sum(-1) - 0.5 * D * LOG2PI
  return lq - lp


def kl_analytic(loc, scale):
  """tfd.kl_divergence(MVNDiag(loc,scale) || N(0,I)) (odin/bay/helpers.py:264-265)."""
  return 0.5 * (scale ** 2 + loc ** 2 - 1.0 - 2.0 * np.log(scale)).sum(-1)


def kl_analytic_reverse(loc, scale):
  """reverse=False swaps the arguments (odin/bay/helpers.py:261-262): tfd.kl_divergence(prior,
  posterior) = KL(N(0,I) || N(loc, scale)) = sum log s + (1 + m^2) / (2 s^2) - 1/2."""
  return (np.log(scale) + 0.5 * (1.0 + loc ** 2) / scale ** 2 - 0.5).sum(-1)


def free_bits_clamp(kl, free_bits: Optional[float], D: int):
  """max(kl, free_bits * D) on the per-sample summed KL (odin/bay/helpers.py:278-280).
  Returns (kl, mask) with mask=1 where the gradient flows."""
  if free_bits is None:
    return kl, np.ones_like(kl)
  thr = free_bits * D
  return np.maximum(kl, thr), (kl > thr).astype(F64)


# --------------------------------------------------------------------------------------
# schedules, TC, permute, discriminator losses
# --------------------------------------------------------------------------------------
def interp_linear(step: float, vmin=1e-6, vmax=1.0, steps=2000, delay_in=0.0):
  """Interpolation.apply + linear (odin/backend/interpolation.py:82-99,119-122),
  non-cyclical branch; AnnealingVAE defaults odin/bay/vi/autoencoder/beta_vae.py:99-107."""
  a = max(float(step), 1e-8)
  a = (a - delay_in) / steps
  a = min(max(a, 0.0), 1.0)
  return (vmax - vmin) * a + vmin


def _logsumexp(a, axis):
  m = a.max(axis=axis, keepdims=True)
  return (m + np.log(np.exp(a - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def total_correlation(z, loc, scale):
  """odin/bay/vi/losses.py:101-157 (minibatch estimator, constants kept exactly as
  the reference keeps them: Normal.log_prob includes -0.5*log(2pi))."""
  lp = (-0.5 * ((z[:, None, :] - loc[None, :, :]) / scale[None, :, :]) ** 2
        - np.log(scale[None, :, :]) - 0.5 * LOG2PI)  # [j,i,l]
  log_qz_product = _logsumexp(lp, 1).sum(1)
  log_qz = _logsumexp(lp.sum(2), 1)
  return float(np.mean(log_qz - log_qz_product))


def total_correlation_bwd(z, loc, scale):
  """Gradients of total_correlation wrt (z, loc, scale)."""
  Bn = z.shape[0]
  d = (z[:, None, :] - loc[None, :, :]) / scale[None, :, :]  # [j,i,l]
  lp = -0.5 * d ** 2 - np.log(scale[None, :, :]) - 0.5 * LOG2PI
  s = lp.sum(2)  # [j,i]
  wj = np.exp(s - _logsumexp(s, 1)[:, None])  # softmax over i of joint     [j,i]
  wl = np.exp(lp - _logsumexp(lp, 1)[:, None, :])  # softmax over i per latent [j,i,l]
  g = (wj[:, :, None] - wl) / Bn  # dTC/dlp[j,i,l]
  dlp_dz = -d / scale[None]
  dlp_dloc = d / scale[None]
  dlp_dscale = (d ** 2 - 1.0) / scale[None]
  return (g * dlp_dz).sum(1), (g * dlp_dloc).sum(0), (g * dlp_dscale).sum(0)


def permute_dims(z, perm):
  """odin/bay/vi/utils.py:233-269 with an EXPLICIT permutation: perm [B,D] int, column l
  of the output is z[perm[:, l], l].  (The reference draws perm with tf.random.shuffle.)"""
  return np.take_along_axis(z, perm, axis=0)


def dtc_loss(logit_z, logit_zperm):
  """odin/bay/vi/autoencoder/factor_discriminator.py:200-235:
  0.5*(mean(-log_sigmoid(l_z)) + mean(l_perm - log_sigmoid(l_perm)))
  = 0.5*(mean softplus(-l_z) + mean softplus(l_perm))."""
  return 0.5 * (np.mean(softplus(-logit_z)) + np.mean(softplus(logit_zperm)))


def dtc_loss_bwd(logit_z, logit_zperm):
  n1, n2 = logit_z.size, logit_zperm.size
  return -0.5 * sigmoid(-logit_z) / n1, 0.5 * sigmoid(logit_zperm) / n2


# --------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------
def adam_keras(theta, g, m, v, t: int, lr: float, b1=0.9, b2=0.999, eps=1e-7):
  """tf.optimizers.Adam (TF 2.5, non-amsgrad) dense update, created at
  odin/networks/base_networks.py:85-112: epsilon OUTSIDE the bias-corrected sqrt."""
  m = b1 * m + (1.0 - b1) * g
  v = b2 * v + (1.0 - b2) * g * g
  a = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
  return theta - a * m / (np.sqrt(v) + eps), m, v


def global_norm(grads: Sequence[np.ndarray]) -> float:
  return math.sqrt(sum(float((g.astype(F64) ** 2).sum()) for g in grads))


def clip_by_global_norm(grads, clip: float):
  """tf.clip_by_global_norm (odin/networks/base_networks.py:588)."""
  n = global_norm(grads)
  sc = clip / max(n, clip)
  return [g * sc for g in grads], n


def gradient_policies(grads: Sequence[np.ndarray], clipnorm=None, global_clipnorm=None,
                      clipvalue=None, skip_update_threshold=None, skip_enabled=True):
  """Networks.optimize between tape.gradient and apply_gradients
  (odin/networks/base_networks.py:549-596), in the reference's order.  Returns (grads, skipped)."""
  g = [np.asarray(x, F64) for x in grads]
  skipped = False
  if skip_update_threshold is not None:
    skipped = bool(any((x >= skip_update_threshold).any() for x in g)) and skip_enabled
    if skipped:
      g = [x - x for x in g]                                  # :561-569
  if clipnorm is not None:                                    # tf.clip_by_norm per variable
    g = [x * (clipnorm / max(math.sqrt(float((x ** 2).sum())), clipnorm)) for x in g]
  if global_clipnorm is not None:
    g, _ = clip_by_global_norm(g, global_clipnorm)
  if clipvalue is not None:
    g = [np.clip(x, -clipvalue, clipvalue) for x in g]
  return g, skipped


def exponential_decay(step, init_lr, decay_steps=10000, rate=0.996, staircase=True):
  """tf.optimizers.schedules.ExponentialDecay (odin/networks/image_networks.py:1010-1013)."""
  p = step / decay_steps
  if staircase:
    p = math.floor(p)
  return init_lr * rate ** p


# --------------------------------------------------------------------------------------
# Sequential networks described by plain tuples
#   ('center',)                      CenterAt0: 2x-1   (image_networks.py:121-126)
#   ('conv',   cout, k, s, act)      Conv2D SAME
#   ('deconv', cout, k, s, act)      Conv2DTranspose SAME
#   ('flatten',)
#   ('dense',  units, act)
#   ('reshape', (h, w, c))
# --------------------------------------------------------------------------------------
Layer = tuple


def layer_param_shapes(layers: Sequence[Layer], in_shape: Tuple[int, ...]):
  """Returns [(layer_index, 'w'|'b', shape)], out_shape.  in_shape excludes batch."""
  shp = tuple(in_shape)
  out = []
  for li, L in enumerate(layers):
    kind = L[0]
    if kind == 'center':
      pass
    elif kind == 'conv':
      _, co, k, s, _a = L
      H, W, C = shp
      out += [(li, 'w', (k, k, C, co)), (li, 'b', (co,))]
      shp = (same_pads(H, k, s)[0], same_pads(W, k, s)[0], co)
    elif kind == 'deconv':
      _, co, k, s, _a = L
      H, W, C = shp
      out += [(li, 'w', (k, k, co, C)), (li, 'b', (co,))]
      shp = (H * s, W * s, co)
    elif kind == 'flatten':
      shp = (int(np.prod(shp)),)
    elif kind == 'dense':
      _, u, _a = L
      out += [(li, 'w', (shp[0], u)), (li, 'b', (u,))]
      shp = (u,)
    elif kind == 'reshape':
      shp = tuple(L[1])
    else:
      raise ValueError(kind)
  return out, shp


def seq_forward(layers, params: Dict[Tuple[int, str], np.ndarray], x):
  """Returns (output, cache) -- cache[i] = input to layer i; cache[-1] = output."""
  acts = [x]
  h = x
  for li, L in enumerate(layers):
    kind = L[0]
    if kind == 'center':
      h = 2.0 * h - 1.0
    elif kind == 'conv':
      h = _ACT[L[4]](conv2d(h, params[(li, 'w')], params[(li, 'b')], L[3]))
    elif kind == 'deconv':
      h = _ACT[L[4]](conv2d_transpose(h, params[(li, 'w')], params[(li, 'b')], L[3]))
    elif kind == 'flatten':
      h = h.reshape(h.shape[0], -1)
    elif kind == 'dense':
      h = _ACT[L[2]](dense(h, params[(li, 'w')], params[(li, 'b')]))
    elif kind == 'reshape':
      h = h.reshape((h.shape[0],) + tuple(L[1]))
    acts.append(h)
  return h, acts


def seq_backward(layers, params, acts, dout, need_dx: bool = False):
  """dout = dL/d(output post-activation).  Returns (dx or None, grads dict)."""
  grads = {}
  g = dout
  n = len(layers)
  for li in range(n - 1, -1, -1):
    L = layers[li]
    kind = L[0]
    xin, yout = acts[li], acts[li + 1]
    last = (li == 0) or all(layers[j][0] in ('center',) for j in range(li))
    ndx = need_dx or not last
    if kind == 'center':
      g = 2.0 * g if g is not None else None
    elif kind == 'conv':
      gp = g * act_grad_from_output(L[4], yout)
      g, grads[(li, 'w')], grads[(li, 'b')] = conv2d_bwd(xin, params[(li, 'w')], gp, L[3], ndx)
    elif kind == 'deconv':
      gp = g * act_grad_from_output(L[4], yout)
      g, grads[(li, 'w')], grads[(li, 'b')] = conv2d_transpose_bwd(
          xin, params[(li, 'w')], gp, L[3], ndx)
    elif kind == 'flatten':
      g = g.reshape(xin.shape) if g is not None else None
    elif kind == 'dense':
      gp = g * act_grad_from_output(L[2], yout)
      g, grads[(li, 'w')], grads[(li, 'b')] = dense_bwd(xin, params[(li, 'w')], gp, ndx)
    elif kind == 'reshape':
      g = g.reshape(xin.shape) if g is not None else None
  return g, grads


# --------------------------------------------------------------------------------------
# The VAE step
# --------------------------------------------------------------------------------------
class OracleVAE:
  """encode -> reparameterise -> decode -> ELBO -> backward, float64.

  Follows VariationalAutoencoder.call/elbo_components
  (odin/bay/vi/autoencoder/variational_autoencoder.py:288-394,515-542), VAEStep.call
  (:117-126), VariationalModel.elbo (odin/bay/vi/_base.py:151-194), BetaVAE
  (beta_vae.py:38-43), BetaTCVAE (beta_vae.py:123-129).

  params keys: ('enc', li, 'w'|'b'), ('dec', li, 'w'|'b'), ('lat','w'|'b')
  (DistributionDense projection, odin/bay/layers/dense_distribution.py:339-380).
  observation: 'bernoulli' (decoder emits C logits) | 'gaussian' (raw loc,scale split,
  image_networks.py:95-102) | 'gaussian_softplus1' (GaussianLayer,
  odin/bay/layers/continuous.py:196-260) | 'qlogistic' | 'mixqlogistic' (decoder emits
  n_components * (2C + C(C-1)/2 + 1) maps, image_networks.py:72-85).
  """

  def __init__(self, enc_layers, dec_layers, in_shape, zdim, observation='bernoulli',
               analytic=False, free_bits=None, beta=1.0, tc_beta: Optional[float] = None,
               reverse: bool = True, n_components: int = 10, capacity: Optional[float] = None):
    # capacity: BetaCapacityVAE (odin/bay/vi/autoencoder/beta_vae.py:132-177): every KL term becomes
    # gamma * |kl - C(step)| -- `beta` plays gamma, `capacity` is the value C(step) of the schedule
    self.capacity = None if capacity is None else float(capacity)
    self.reverse = bool(reverse)
    self.n_components = int(n_components)  # 'mixqlogistic' (image_networks.py:48, default 10)
    self.enc, self.dec = list(enc_layers), list(dec_layers)
    self.in_shape, self.D = tuple(in_shape), int(zdim)
    self.observation, self.analytic, self.free_bits = observation, analytic, free_bits
    self.beta, self.tc_beta = float(beta), tc_beta
    self.enc_shapes, eo = layer_param_shapes(self.enc, self.in_shape)
    assert len(eo) == 1, 'encoder must end flat'
    self.hdim = eo[0]
    self.dec_shapes, do = layer_param_shapes(self.dec, (self.D,))
    self.out_shape = do

  def param_shapes(self):
    out = [(('enc', li, k), s) for li, k, s in self.enc_shapes]
    out += [(('lat', 'w'), (self.hdim, 2 * self.D)), (('lat', 'b'), (2 * self.D,))]
    out += [(('dec', li, k), s) for li, k, s in self.dec_shapes]
    return out

  def init_params(self, seed=1, scale=1.0):
    """Seeded He-normal-ish init (values only matter for being shared with the DUT)."""
    rng = np.random.default_rng(seed)
    P = {}
    for key, shp in self.param_shapes():
      if key[-1] == 'b':
        P[key] = 0.05 * rng.standard_normal(shp)
      else:
        if len(shp) == 4:
          fan_in = shp[0] * shp[1] * (shp[2] if self._is_conv(key) else shp[3])
        else:
          fan_in = shp[0]
        P[key] = scale * math.sqrt(2.0 / fan_in) * rng.standard_normal(shp)
    return P

  def _is_conv(self, key):
    net = self.enc if key[0] == 'enc' else self.dec
    return net[key[1]][0] == 'conv'

  @staticmethod
  def _sub(P, net):
    return {(k[1], k[2]): v for k, v in P.items() if k[0] == net}

  # ---- forward ----
  def forward(self, P, x, eps):
    x = np.asarray(x, F64)
    eps = np.asarray(eps, F64)
    B = x.shape[0]
    h_e, enc_acts = seq_forward(self.enc, self._sub(P, 'enc'), x)
    p = dense(h_e, P[('lat', 'w')], P[('lat', 'b')])
    loc, scale = mvn_diag_params(p, self.D)
    z = loc + scale * eps
    h_d, dec_acts = seq_forward(self.dec, self._sub(P, 'dec'), z)
    if self.observation == 'bernoulli':
      llk = bernoulli_log_prob(h_d, x)
      recon = sigmoid(h_d)
    elif self.observation == 'qlogistic':
      C = x.shape[-1]
      oloc, raw = h_d[..., :C], h_d[..., C:]
      llk = qlogistic_log_prob(oloc, raw, x)
      recon = qlogistic_params(oloc, raw)[0] / 255.0  # QuantizedLogistic.mean (quantized.py:185-187)
    elif self.observation == 'mixqlogistic':
      C = x.shape[-1]
      llk = mixql_log_prob(h_d, x, C, self.n_components)
      recon = mixql_mean(h_d, C, self.n_components)
    else:
      C = x.shape[-1]
      oloc, raw = h_d[..., :C], h_d[..., C:]
      oscale = softplus1(raw) if self.observation == 'gaussian_softplus1' else raw
      llk = gaussian_log_prob(oloc, oscale, x)
      recon = oloc
    if not self.reverse:
      # reverse=False with the Monte-Carlo form cannot be evaluated by the reference either:
      # after the swap `tf.convert_to_tensor(prior)` is called on a plain tfd.Independent
      # (helpers.py:267-276 with q_sample=None from variational_autoencoder.py:535-539)
      assert self.analytic, 'reverse=False needs analytic=True'
      kl_raw = kl_analytic_reverse(loc, scale)
    else:
      kl_raw = kl_analytic(loc, scale) if self.analytic else kl_mc(loc, scale, z)
    kl_c, fb_mask = free_bits_clamp(kl_raw, self.free_bits, self.D)
    if self.capacity is not None:  # beta_vae.py:174-176: tf.math.abs(val - c); d|x| = sign(x)
      fb_mask = fb_mask * np.sign(kl_c - self.capacity)
      kl_c = np.abs(kl_c - self.capacity)
    kl = self.beta * kl_c
    elbo = llk - kl
    out = dict(h_e=h_e, p=p, loc=loc, raw_scale=p[:, self.D:], scale=scale, z=z, h_d=h_d,
               recon=recon, llk=llk, kl_raw=kl_raw, kl=kl, fb_mask=fb_mask,
               enc_acts=enc_acts, dec_acts=dec_acts)
    if self.tc_beta is not None:
      tc = total_correlation(z, loc, scale)
      out['tc'] = (self.tc_beta - 1.0) * tc
      elbo = elbo - out['tc']
    out['elbo'] = elbo
    out['loss'] = float(-np.mean(elbo))
    out['metrics'] = dict(llk_image=float(llk.mean()), kl_latents=float(kl.mean()))
    if self.tc_beta is not None:
      out['metrics']['tc_latents'] = float(out['tc'])
    return out

  # ---- backward of loss = -mean(elbo) ----
  def backward(self, P, x, eps, fwd=None, extra_dz=None):
    """extra_dz: optional dL/dz added from outside (FactorVAE discriminator TC term)."""
    x = np.asarray(x, F64)
    eps = np.asarray(eps, F64)
    f = fwd if fwd is not None else self.forward(P, x, eps)
    B = x.shape[0]
    D = self.D
    h_d = f['h_d']
    # dL/dh_d, L = -(1/B) sum llk + ...
    if self.observation == 'bernoulli':
      dh_d = -(bernoulli_log_prob_grad(h_d, x)) / B
    elif self.observation == 'qlogistic':
      C = x.shape[-1]
      gl, gr = qlogistic_log_prob_grad(h_d[..., :C], h_d[..., C:], x)
      dh_d = -np.concatenate([gl, gr], -1) / B
    elif self.observation == 'mixqlogistic':
      dh_d = -mixql_log_prob_grad(h_d, x, x.shape[-1], self.n_components) / B
    else:
      C = x.shape[-1]
      oloc, raw = h_d[..., :C], h_d[..., C:]
      if self.observation == 'gaussian_softplus1':
        oscale = softplus1(raw)
        dscale_draw = sigmoid(raw + SOFTPLUS_INV_1)
      else:
        oscale = raw
        dscale_draw = np.ones_like(raw)
      d = (x - oloc) / oscale
      dllk_dloc = d / oscale
      dllk_dscale = (d ** 2 - 1.0) / oscale
      dh_d = -np.concatenate([dllk_dloc, dllk_dscale * dscale_draw], -1) / B
    dz, gdec = seq_backward(self.dec, self._sub(P, 'dec'), f['dec_acts'], dh_d, need_dx=True)
    loc, scale, z = f['loc'], f['scale'], f['z']
    # KL term: L += (beta/B) * sum_b clamp(kl_raw_b)
    wkl = (self.beta / B) * f['fb_mask'][:, None]
    if not self.reverse:
      dloc = wkl * loc / scale ** 2
      dscale = wkl * (1.0 / scale - (1.0 + loc ** 2) / scale ** 3)
    elif self.analytic:
      dloc = wkl * loc
      dscale = wkl * (scale - 1.0 / scale)
    else:
      # kl = -0.5*eps^2 - log(scale) + 0.5 z^2 (+const); z = loc + scale*eps
      dloc = wkl * z
      dscale = wkl * (z * eps - 1.0 / scale)
    if extra_dz is not None:
      dz = dz + extra_dz
    if self.tc_beta is not None:
      tz, tl, ts = total_correlation_bwd(z, loc, scale)
      c = (self.tc_beta - 1.0)
      dz = dz + c * tz
      dloc = dloc + c * tl
      dscale = dscale + c * ts
    dloc = dloc + dz
    dscale = dscale + dz * eps
    draw = dscale * sigmoid(f['raw_scale'])
    dp = np.concatenate([dloc, draw], -1)
    dh_e, gw, gb = dense_bwd(f['h_e'], P[('lat', 'w')], dp)
    _, genc = seq_backward(self.enc, self._sub(P, 'enc'), f['enc_acts'], dh_e)
    G = {('lat', 'w'): gw, ('lat', 'b'): gb}
    G.update({('enc',) + k: v for k, v in genc.items()})
    G.update({('dec',) + k: v for k, v in gdec.items()})
    return G, dict(dh_d=dh_d, dz=dz, dp=dp, dh_e=dh_e)


def marginal_log_prob(model: OracleVAE, P, x, eps_n):
  """VariationalAutoencoder.marginal_log_prob (variational_autoencoder.py:396-513) with
  reduce=None: one encoder pass, z = q.sample(n) from the given eps_n [n,B,D], decode n*B codes;
  returns (llk [B] = logsumexp_k log p(x|z_k) - log n,
           (lq [B], lp [B]) = the same log-mean-exp of log q(z_k|x) and of log p(z_k))."""
  x = np.asarray(x, F64)
  eps_n = np.asarray(eps_n, F64)
  n, B, D = eps_n.shape
  h_e, _ = seq_forward(model.enc, model._sub(P, 'enc'), x)
  loc, scale = mvn_diag_params(dense(h_e, P[('lat', 'w')], P[('lat', 'b')]), D)
  z = loc[None] + scale[None] * eps_n
  h_d, _ = seq_forward(model.dec, model._sub(P, 'dec'), z.reshape(n * B, D))
  xt = np.tile(x, (n,) + (1,) * (x.ndim - 1))
  if model.observation == 'bernoulli':
    llk = bernoulli_log_prob(h_d, xt)
  else:
    C = x.shape[-1]
    oloc, raw = h_d[..., :C], h_d[..., C:]
    llk = gaussian_log_prob(oloc, softplus1(raw) if model.observation == 'gaussian_softplus1'
                            else raw, xt)
  lq = (-0.5 * eps_n ** 2 - np.log(scale)[None]).sum(-1) - 0.5 * D * LOG2PI
  lp = (-0.5 * z ** 2).sum(-1) - 0.5 * D * LOG2PI
  lme = lambda a: _logsumexp(a, 0) - math.log(n)
  return lme(llk.reshape(n, B)), (lme(lq), lme(lp))


def train_step(model: OracleVAE, P, M, V, t, x, eps, lr, global_clipnorm=None):
  """Networks.optimize (odin/networks/base_networks.py:415-624), one VAEStep:
  grads -> optional clip_by_global_norm -> Keras Adam.  Returns (P, M, V, fwd, G)."""
  f = model.forward(P, x, eps)
  G, _ = model.backward(P, x, eps, f)
  keys = [k for k, _ in model.param_shapes()]
  if global_clipnorm is not None:
    gl, _ = clip_by_global_norm([G[k] for k in keys], global_clipnorm)
    G = dict(zip(keys, gl))
  P2, M2, V2 = {}, {}, {}
  for k in keys:
    P2[k], M2[k], V2[k] = adam_keras(P[k], G[k], M[k], V[k], t, lr)
  return P2, M2, V2, f, G


# --------------------------------------------------------------------------------------
# FactorVAE discriminator (dense relu MLP -> 1 logit)
# odin/bay/vi/autoencoder/factor_vae.py:149-176; factor_discriminator.py:66-235
# --------------------------------------------------------------------------------------
def disc_layers(units=(1000,) * 5):
  return [('dense', u, 'relu') for u in units] + [('dense', 1, 'linear')]


def disc_forward(layers, DP, z):
  out, acts = seq_forward(layers, DP, z)
  return out[:, 0], acts


def disc_backward(layers, DP, acts, dlogit, need_dz=False):
  return seq_backward(layers, DP, acts, dlogit[:, None], need_dx=need_dz)


def factor_vae_iteration(model: 'OracleVAE', P, M, V, t: int, dlayers, DP, DM, DV, td: int, x,
                         eps1, eps2, perm, lr: float, tc_coef: float = 7.0,
                         global_clipnorm=None, pretraining: bool = False, disc_lr=1e-5,
                         disc_b1=0.5, disc_b2=0.9):
  """One FactorVAE training iteration = the two TrainSteps of
  odin/bay/vi/autoencoder/factor_vae.py:239-287 run by Networks.optimize
  (odin/networks/base_networks.py:415-624) one after the other:

  step 1 (VAEStep on x1, VAE parameters): loss = -mean(llk - beta*kl) + tc_coef*mean(D(z))
      (factor_vae.py:202-223; factor_discriminator.py:169-198: the gradient flows through D
      into z, D's own parameters are not in this step's parameter list), Adam of `fit`.
  step 2 (FactorDiscriminatorStep on x2, discriminator parameters, factor_vae.py:66-93):
      z' = encode(x2) with the ALREADY UPDATED encoder, z = the cached sample of step 1;
      dtc_loss (factor_discriminator.py:200-235) with stop_gradient on both;
      Adam(1e-5, beta_1=.5, beta_2=.9) (factor_vae.py:173-176).

  `model.beta` must hold the annealed beta of this step (AnnealingVAE, beta_vae.py:99-107).
  Returns a dict with every intermediate the parity tests compare."""
  x = np.asarray(x, F64)
  B1 = x.shape[0] // 2
  x1, x2 = x[:B1], x[B1:]
  f = model.forward(P, x1, eps1)
  out = dict(fwd=f)
  extra = None
  tc = 0.0
  if not pretraining:
    logit, acts = disc_forward(dlayers, DP, f['z'])
    tc = tc_coef * float(np.mean(logit))
    extra, _ = disc_backward(dlayers, DP, acts, np.full(B1, tc_coef / B1), need_dz=True)
  out['tc'], out['loss'], out['extra_dz'] = tc, f['loss'] + tc, extra
  G, _ = model.backward(P, x1, eps1, f, extra_dz=extra)
  out['G'] = G
  keys = [k for k, _ in model.param_shapes()]
  Gc = G
  if global_clipnorm is not None:
    gl, _ = clip_by_global_norm([G[k] for k in keys], global_clipnorm)
    Gc = dict(zip(keys, gl))
  P2, M2, V2 = {}, {}, {}
  for k in keys:
    P2[k], M2[k], V2[k] = adam_keras(P[k], Gc[k], M[k], V[k], t, lr)
  out.update(P=P2, M=M2, V=V2)
  if pretraining:
    return out
  f2 = model.forward(P2, x2, eps2)
  zp = permute_dims(f2['z'], np.asarray(perm, np.int64))
  l1, a1 = disc_forward(dlayers, DP, f['z'])
  l2, a2 = disc_forward(dlayers, DP, zp)
  out['z2'], out['zperm'], out['dtc_loss'] = f2['z'], zp, dtc_loss(l1, l2)
  d1, d2 = dtc_loss_bwd(l1, l2)
  _, g1 = disc_backward(dlayers, DP, a1, d1)
  _, g2 = disc_backward(dlayers, DP, a2, d2)
  DG = {k: g1[k] + g2[k] for k in g1}
  out['DG'] = DG
  DP2, DM2, DV2 = {}, {}, {}
  for k in DG:
    DP2[k], DM2[k], DV2[k] = adam_keras(DP[k], DG[k], DM[k], DV[k], td, disc_lr, b1=disc_b1,
                                        b2=disc_b2)
  out.update(DP=DP2, DM=DM2, DV=DV2)
  return out


# --------------------------------------------------------------------------------------
# Network specs restating odin/networks/image_networks.py (used by tests to drive both
# the oracle and the product with identical architectures)
# --------------------------------------------------------------------------------------
def dsprites_spec(n_channels=1, zdim=None, proj_dim=None, n_out_params=1):
  """dsprites_networks / shapes3d_networks (image_networks.py:436-534,560-597)."""
  if zdim is None:
    zdim = 10 if n_channels == 1 else 6
  if proj_dim is None:
    proj_dim = 128 if n_channels == 1 else 256
  enc = [('center',), ('conv', 32, 4, 2, 'elu'), ('conv', 32, 4, 2, 'elu'),
         ('conv', 64, 4, 2, 'elu'), ('conv', 64, 4, 2, 'elu'), ('flatten',),
         ('dense', proj_dim, 'linear')]
  dec = [('dense', proj_dim, 'linear'), ('reshape', (4, 4, proj_dim // 16)),
         ('deconv', 64, 4, 2, 'elu'), ('deconv', 64, 4, 2, 'elu'),
         ('deconv', 32, 4, 2, 'elu'), ('deconv', 32, 4, 2, 'elu'),
         ('conv', n_channels * n_out_params, 1, 1, 'linear')]
  return enc, dec, (64, 64, n_channels), zdim


def celeba_spec(zdim=45, n_out=3):
  """celeba_networks (image_networks.py:661-725); observation choice per SURVEY a12."""
  enc = [('center',), ('conv', 32, 4, 2, 'elu'), ('conv', 32, 4, 2, 'elu'),
         ('conv', 64, 4, 2, 'elu'), ('conv', 64, 4, 1, 'elu'), ('flatten',),
         ('dense', 512, 'linear')]
  dec = [('dense', 512, 'linear'), ('reshape', (8, 8, 8)),
         ('deconv', 64, 4, 1, 'elu'), ('deconv', 64, 4, 2, 'elu'),
         ('deconv', 32, 4, 2, 'elu'), ('deconv', 32, 4, 2, 'elu'),
         ('conv', n_out, 1, 1, 'linear')]
  return enc, dec, (64, 64, 3), zdim


def mnist_conv_spec(zdim=32):
  """mnist_networks (image_networks.py:223-292)."""
  enc = [('center',), ('conv', 32, 5, 1, 'elu'), ('conv', 32, 5, 2, 'elu'),
         ('conv', 64, 5, 1, 'elu'), ('conv', 64, 5, 2, 'elu'), ('flatten',),
         ('dense', 196, 'linear')]
  dec = [('dense', 196, 'linear'), ('reshape', (7, 7, 4)),
         ('deconv', 64, 5, 2, 'elu'), ('conv', 64, 5, 1, 'elu'),
         ('deconv', 32, 5, 2, 'elu'), ('conv', 32, 5, 1, 'elu'),
         ('conv', 1, 1, 1, 'linear')]
  return enc, dec, (28, 28, 1), zdim


def mnist_dense_spec(zdim=16):
  """VariationalAutoencoder defaults (variational_autoencoder.py:181-185) +
  dense_network (base_networks.py:965-1022)."""
  enc = [('flatten',), ('dense', 512, 'relu'), ('dense', 512, 'relu')]
  dec = [('dense', 512, 'relu'), ('dense', 512, 'relu'), ('dense', 784, 'linear'),
         ('reshape', (28, 28, 1))]
  return enc, dec, (28, 28, 1), zdim
