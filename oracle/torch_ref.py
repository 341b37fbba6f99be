"""CPU ORACLE #2 (test infrastructure, NOT product code): torch-CPU restatement.

An independent second restatement of the reference VAE step that uses torch's own
conv kernels (explicit ``F.pad`` + ``F.conv2d`` / ``F.conv_transpose2d`` + crop) and
``torch.autograd`` for every gradient, so that it shares no arithmetic code with
``oracle/vae_oracle.py``.  Uses:

* tests: float64 cross-check of the numpy oracle's forward and hand-written backward;
* ``bench.py``: the ``cpu_baseline`` leg (``kind: "port"``) -- the same training step in
  fp32 on all host cores (the reference's TF-CPU path cannot run: no TensorFlow on the
  box).

PARITY STATUS: parity unpinned vs the TF reference (see oracle/vae_oracle.py header).
Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .vae_oracle import same_pads, layer_param_shapes, LOG2PI, SOFTPLUS_INV_1


def _act(name, x):
  if name == 'linear':
    return x
  if name == 'elu':
    return F.elu(x)
  if name == 'relu':
    return F.relu(x)
  raise ValueError(name)


def t_conv2d(x, w, b, s):
  """x NHWC, w HWIO (Keras Conv2D SAME; odin/networks/image_networks.py:166-169)."""
  B, H, W, C = x.shape
  kh, kw, _, co = w.shape
  _, pt, pb = same_pads(H, kh, s)
  _, pl, pr = same_pads(W, kw, s)
  xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
  y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=s)
  return y.permute(0, 2, 3, 1)


def t_deconv2d(x, w, b, s):
  """x NHWC, w (kh,kw,Cout,Cin) (Keras Conv2DTranspose SAME; image_networks.py:170-173)."""
  B, H, W, C = x.shape
  kh, kw, co, ci = w.shape
  _, pt, _ = same_pads(H * s, kh, s)
  _, pl, _ = same_pads(W * s, kw, s)
  # torch weight for conv_transpose2d: [Cin, Cout, kh, kw]
  y = F.conv_transpose2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), b, stride=s)
  y = y[:, :, pt:pt + H * s, pl:pl + W * s]
  return y.permute(0, 2, 3, 1)


def t_seq(layers, params, x, relu_ties=None, tie_tol=1e-5):
  """relu_ties: {layer index: the checked implementation's OUTPUT of that relu Dense layer}.  relu'(0) is a tie that
  rounding decides: where the two sides disagree on the sign of a pre-activation that is zero to within tie_tol
  (relative to the layer's largest), the reference takes the other side's branch -- one flipped unit otherwise
  shows up as an O(1e-3) error of the gradients below it.  A disagreement anywhere else is an error."""
  h = x
  for li, L in enumerate(layers):
    if relu_ties is not None and li in relu_ties and L[0] == 'dense' and L[2] == 'relu':
      pre = h @ params[(li, 'w')] + params[(li, 'b')]
      mine, theirs = pre.detach() > 0, relu_ties[li] > 0
      diff = mine != theirs
      if bool(diff.any()):
        worst = float(pre.detach()[diff].abs().max() / pre.detach().abs().max())
        assert worst <= tie_tol, ('relu masks disagree away from zero', li, int(diff.sum()), worst)
      h = pre * theirs.to(pre.dtype)
      continue
    k = L[0]
    if k == 'center':
      h = 2.0 * h - 1.0
    elif k == 'conv':
      h = _act(L[4], t_conv2d(h, params[(li, 'w')], params[(li, 'b')], L[3]))
    elif k == 'deconv':
      h = _act(L[4], t_deconv2d(h, params[(li, 'w')], params[(li, 'b')], L[3]))
    elif k == 'flatten':
      h = h.reshape(h.shape[0], -1)
    elif k == 'dense':
      h = _act(L[2], h @ params[(li, 'w')] + params[(li, 'b')])
    elif k == 'reshape':
      h = h.reshape((h.shape[0],) + tuple(L[1]))
  return h


def t_total_correlation(z, loc, scale):
  """odin/bay/vi/losses.py:101-157."""
  lp = (-0.5 * ((z[:, None, :] - loc[None]) / scale[None]) ** 2 - torch.log(scale[None])
        - 0.5 * LOG2PI)
  return (torch.logsumexp(lp.sum(2), 1) - torch.logsumexp(lp, 1).sum(1)).mean()


def t_qlogistic_log_prob(loc, raw, x, low=0.0, high=255.0):
  """QuantizedLogistic.log_prob (odin/bay/distributions/quantized.py:50-204 over TFP's
  QuantizedDistribution), torch autograd version; x*high is rounded to float32 like the reference."""
  support = 0.5 * (high - low)
  m = low + support * (loc + 1.0)
  s = (F.softplus(raw) + math.exp(-7.0)) * support
  y = (x.to(torch.float32) * torch.tensor(high, dtype=torch.float32)).to(loc.dtype)
  ninf = torch.full_like(m, -float('inf'))
  zero = torch.zeros_like(m)

  def logcdf(j):
    r = -F.softplus(-(j + 0.5 - m) / s)
    return torch.where(j < high, torch.where(j < low, ninf, r), zero)

  def logsf(j):
    r = -F.softplus((j + 0.5 - m) / s)
    return torch.where(j < high, torch.where(j < low, zero, r), ninf)

  lsy, lsy1 = logsf(torch.ceil(y)), logsf(torch.ceil(y - 1.0))
  lcy, lcy1 = logcdf(torch.floor(y)), logcdf(torch.floor(y - 1.0))
  use_sf = lsy < lcy
  big = torch.where(use_sf, lsy1, lcy)
  small = torch.where(use_sf, lsy, lcy1)
  # log(exp(big) - exp(small)); small = -inf contributes nothing (and no gradient)
  fin = torch.isfinite(small)
  d = torch.where(fin, big - small, torch.ones_like(big))
  l1m = torch.where(d < math.log(2.0), torch.log(-torch.expm1(-d)), torch.log1p(-torch.exp(-d)))
  return big + torch.where(fin, l1m, torch.zeros_like(l1m))


def t_mixql_log_prob_pix(h, x, C, K=10):
  """MixtureQuantizedLogistic.log_prob per pixel (odin/bay/distributions/quantized.py:284-349): K
  components, channel chain on the transformed values, MixtureSameFamily over Independent(QL, 1)."""
  no = 2 * C + C * (C - 1) // 2 + 1
  hh = h.reshape(h.shape[:-1] + (K, no))
  logits, locs, raw, coefs = hh[..., 0], hh[..., 1:1 + C], hh[..., 1 + C:1 + 2 * C], hh[..., 1 + 2 * C:]
  xt = 2.0 * x - 1.0
  cols = [locs[..., i] for i in range(C)]
  cnt = 0
  for i in range(C):
    for j in range(i):
      cols[i] = cols[i] + xt[..., None, j] * coefs[..., cnt]
      cnt += 1
  le = torch.stack(cols, -1)
  q = t_qlogistic_log_prob(le, raw, x[..., None, :].expand(le.shape))
  comp = torch.log_softmax(logits, -1) + q.sum(-1)
  return torch.logsumexp(comp, -1)


class TorchVAE:
  """Same constructor arguments as oracle.vae_oracle.OracleVAE."""

  def __init__(self, enc_layers, dec_layers, in_shape, zdim, observation='bernoulli',
               analytic=False, free_bits=None, beta=1.0, tc_beta=None,
               dtype=torch.float64, reverse=True, n_components=10, capacity=None):
    self.capacity = capacity  # BetaCapacityVAE (beta_vae.py:132-177): beta * |kl - capacity|
    self.reverse = bool(reverse)
    self.n_components = int(n_components)
    self.enc, self.dec = list(enc_layers), list(dec_layers)
    self.in_shape, self.D = tuple(in_shape), int(zdim)
    self.observation, self.analytic, self.free_bits = observation, analytic, free_bits
    self.beta, self.tc_beta, self.dtype = float(beta), tc_beta, dtype

  def tensors(self, P: Dict, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=self.dtype, requires_grad=requires_grad)
            for k, v in P.items()}

  @staticmethod
  def _sub(T, net):
    return {(k[1], k[2]): v for k, v in T.items() if k[0] == net}

  def forward(self, T, x, eps, extra_loss_fn=None):
    D = self.D
    B = x.shape[0]
    h_e = t_seq(self.enc, self._sub(T, 'enc'), x)
    p = h_e @ T[('lat', 'w')] + T[('lat', 'b')]
    loc, scale = p[:, :D], F.softplus(p[:, D:])
    z = loc + scale * eps
    h_d = t_seq(self.dec, self._sub(T, 'dec'), z)
    if self.observation == 'bernoulli':
      llk = (x * h_d - F.softplus(h_d)).reshape(B, -1).sum(1)
      recon = torch.sigmoid(h_d)
    elif self.observation == 'qlogistic':
      C = x.shape[-1]
      llk = t_qlogistic_log_prob(h_d[..., :C], h_d[..., C:], x).reshape(B, -1).sum(1)
      recon = (127.5 * (h_d[..., :C] + 1.0)) / 255.0
    elif self.observation == 'mixqlogistic':
      C = x.shape[-1]
      llk = t_mixql_log_prob_pix(h_d, x, C, self.n_components).reshape(B, -1).sum(1)
      recon = None
    else:
      C = x.shape[-1]
      oloc, raw = h_d[..., :C], h_d[..., C:]
      osc = F.softplus(raw + SOFTPLUS_INV_1) if self.observation == 'gaussian_softplus1' else raw
      llk = (-0.5 * ((x - oloc) / osc) ** 2 - torch.log(osc) - 0.5 * LOG2PI).reshape(B, -1).sum(1)
      recon = oloc
    if not self.reverse:  # KL(p || q), closed form (odin/bay/helpers.py:261-265)
      kl_raw = (torch.log(scale) + 0.5 * (1.0 + loc ** 2) / scale ** 2 - 0.5).sum(-1)
    elif self.analytic:
      kl_raw = 0.5 * (scale ** 2 + loc ** 2 - 1.0 - 2.0 * torch.log(scale)).sum(-1)
    else:
      lq = (-0.5 * ((z - loc) / scale) ** 2 - torch.log(scale)).sum(-1) - 0.5 * D * LOG2PI
      lp = (-0.5 * z ** 2).sum(-1) - 0.5 * D * LOG2PI
      kl_raw = lq - lp
    kl = kl_raw
    if self.free_bits is not None:
      kl = torch.clamp(kl, min=self.free_bits * D)
    if self.capacity is not None:
      kl = torch.abs(kl - self.capacity)
    kl = self.beta * kl
    elbo = llk - kl
    out = dict(h_e=h_e, p=p, loc=loc, scale=scale, z=z, h_d=h_d, recon=recon, llk=llk,
               kl_raw=kl_raw, kl=kl)
    if self.tc_beta is not None:
      out['tc'] = (self.tc_beta - 1.0) * t_total_correlation(z, loc, scale)
      elbo = elbo - out['tc']
    loss = -elbo.mean()
    if extra_loss_fn is not None:
      loss = loss + extra_loss_fn(out)
    out['elbo'], out['loss'] = elbo, loss
    return out

  def loss_and_grads(self, P, x, eps, extra_loss_fn=None):
    T = self.tensors(P)
    xt = torch.tensor(np.asarray(x), dtype=self.dtype)
    et = torch.tensor(np.asarray(eps), dtype=self.dtype)
    out = self.forward(T, xt, et, extra_loss_fn)
    out['loss'].backward()
    G = {k: v.grad.detach().numpy() for k, v in T.items()}
    return {k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}, G


class TorchTrainer:
  """fp32 CPU training step used as bench.py's cpu_baseline ("port"): forward, autograd
  backward, Keras-Adam (epsilon outside sqrt; odin/networks/base_networks.py:85-112,604)."""

  def __init__(self, model: TorchVAE, P: Dict, lr=1e-3, threads: Optional[int] = None):
    if threads:
      torch.set_num_threads(threads)
    self.model = model
    self.T = {k: torch.tensor(np.asarray(v), dtype=model.dtype, requires_grad=True)
              for k, v in P.items()}
    self.M = {k: torch.zeros_like(v) for k, v in self.T.items()}
    self.V = {k: torch.zeros_like(v) for k, v in self.T.items()}
    self.t, self.lr = 0, lr

  def step(self, x: torch.Tensor, eps: torch.Tensor) -> float:
    for v in self.T.values():
      v.grad = None
    out = self.model.forward(self.T, x, eps)
    out['loss'].backward()
    self.t += 1
    b1, b2, e = 0.9, 0.999, 1e-7
    a = self.lr * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
    with torch.no_grad():
      for k, p in self.T.items():
        g = p.grad
        self.M[k].mul_(b1).add_(g, alpha=1 - b1)
        self.V[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(self.M[k], self.V[k].sqrt().add_(e), value=-a)
    return float(out['loss'])


class TorchFactorTrainer:
  """fp32 CPU port of ONE FactorVAE iteration (odin/bay/vi/autoencoder/factor_vae.py:239-287, restated in
  oracle.vae_oracle.factor_vae_iteration): VAE step on the first half of the batch with the discriminator's
  tc_coef * mean(D(z)) term (D's parameters not updated), Keras-Adam; then the discriminator step on the second
  half -- z' = encode(x2) with the ALREADY updated encoder, permute_dims, dtc_loss on stopped gradients,
  Adam(1e-5, 0.5, 0.9).  bench.py's cpu_baseline for the FactorVAE workload."""

  def __init__(self, model: TorchVAE, P: Dict, disc_layers, DP: Dict, lr=1e-3, tc_coef=7.0,
               threads: Optional[int] = None):
    self.vae = TorchTrainer(model, P, lr=lr, threads=threads)
    self.dl = list(disc_layers)
    self.D = {k: torch.tensor(np.asarray(v), dtype=model.dtype, requires_grad=True) for k, v in DP.items()}
    self.DM = {k: torch.zeros_like(v) for k, v in self.D.items()}
    self.DV = {k: torch.zeros_like(v) for k, v in self.D.items()}
    self.td, self.tc_coef = 0, float(tc_coef)

  def _disc(self, z, params):
    return t_seq(self.dl, params, z)[:, 0]

  def step(self, x: torch.Tensor, eps1: torch.Tensor, eps2: torch.Tensor, perm: torch.Tensor) -> float:
    B1 = x.shape[0] // 2
    x1, x2 = x[:B1], x[B1:]
    tr, m = self.vae, self.vae.model
    Dfix = {(k[1], k[2]): v.detach() for k, v in self.D.items()}
    for v in tr.T.values():
      v.grad = None
    out = m.forward(tr.T, x1, eps1, extra_loss_fn=lambda o: self.tc_coef * self._disc(o['z'], Dfix).mean())
    out['loss'].backward()
    tr.t += 1
    b1, b2, e = 0.9, 0.999, 1e-7
    a = tr.lr * math.sqrt(1 - b2 ** tr.t) / (1 - b1 ** tr.t)
    with torch.no_grad():
      for k, p in tr.T.items():
        g = p.grad
        tr.M[k].mul_(b1).add_(g, alpha=1 - b1)
        tr.V[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(tr.M[k], tr.V[k].sqrt().add_(e), value=-a)
    # ---- discriminator step ----
    with torch.no_grad():
      z1 = out['z'].detach()
      z2 = m.forward(tr.T, x2, eps2)['z']
      zp = torch.gather(z2, 0, perm)
    for v in self.D.values():
      v.grad = None
    Dp = {(k[1], k[2]): v for k, v in self.D.items()}
    l1, l2 = self._disc(z1, Dp), self._disc(zp, Dp)
    dloss = 0.5 * (F.softplus(-l1).mean() + F.softplus(l2).mean())
    dloss.backward()
    self.td += 1
    b1, b2 = 0.5, 0.9
    a = 1e-5 * math.sqrt(1 - b2 ** self.td) / (1 - b1 ** self.td)
    with torch.no_grad():
      for k, p in self.D.items():
        g = p.grad
        self.DM[k].mul_(b1).add_(g, alpha=1 - b1)
        self.DV[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(self.DM[k], self.DV[k].sqrt().add_(e), value=-a)
    return float(out['loss'].detach())
