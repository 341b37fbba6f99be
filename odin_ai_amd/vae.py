"""VariationalAutoencoder / BetaVAE / AnnealingVAE / BetaTCVAE / FactorVAE on the HIP engine.

Drop-in for the reference's model API on this path (same class names, constructor
arguments, method names, returned structures and error behaviour):

  odin/bay/vi/autoencoder/variational_autoencoder.py:132 (VariationalAutoencoder, VAEStep)
  odin/bay/vi/_base.py:21-194                             (ELBO configuration, elbo())
  odin/bay/vi/autoencoder/beta_vae.py:11,83,110           (BetaVAE, AnnealingVAE, BetaTCVAE)
  odin/bay/vi/autoencoder/factor_vae.py:99                (FactorVAE, two-step training)
  odin/networks/base_networks.py:415-812                  (optimize(), fit())

Tensors are torch tensors on the model's device; returned "distributions" are light
objects exposing what callers of the reference use (mean / stddev / sample / log_prob /
event_shape / batch_shape / KL_divergence).  Every FLOP of encode / decode / ELBO /
backward / Adam runs in libodin_hip.so through ``VAEEngine``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _lib
from .engine import (ACT, H_ALPHA, N_HYPER, RANGE_WORDS, NetProgram, ParamLayout, ReduceJob, VAEEngine,
                     build_layers)
from .interpolation import Interpolation, linear
from .networks import RVconf, SequentialNetwork, get_networks, layer_names

LOG2PI = math.log(2.0 * math.pi)


# ======================================================================================
# light distribution objects
# ======================================================================================
class KLdivergence:
  """odin/bay/helpers.py:285-372: callable attached to the posterior; `.prior` attribute."""

  def __init__(self, posterior: 'MVNDiagPosterior', prior='Independent(Normal(0,1))'):
    self.posterior, self.prior = posterior, prior

  def __call__(self, analytic=False, reverse=True, free_bits=None, sample_shape=None,
               keepdims=False):
    q = self.posterior
    if not reverse and not analytic:
      raise TypeError('reverse=False needs analytic=True (helpers.py:261-276: the swapped '
                      '"posterior" is the prior, which carries no cached sample)')
    if not reverse:  # KL(p || q), helpers.py:261-265
      kl = (torch.log(q.scale) + 0.5 * (1.0 + q.loc ** 2) / q.scale ** 2 - 0.5).sum(-1)
    elif analytic:
      kl = 0.5 * (q.scale ** 2 + q.loc ** 2 - 1.0 - 2.0 * torch.log(q.scale)).sum(-1)
    else:
      z = q.z
      lq = (-0.5 * ((z - q.loc) / q.scale) ** 2 - torch.log(q.scale)).sum(-1)
      lp = (-0.5 * z ** 2).sum(-1)
      kl = lq - lp
    if free_bits is not None:
      kl = torch.clamp(kl, min=free_bits * q.loc.shape[-1])
    if analytic and keepdims:
      kl = kl.unsqueeze(0)  # odin/bay/helpers.py:370-371
    return kl


class MVNDiagPosterior:
  """MultivariateNormalDiag(loc, softplus(raw)) with its cached sample
  (odin/bay/layers/continuous.py:459-483; dense_distribution.py:339-380)."""

  def __init__(self, p: torch.Tensor, z: torch.Tensor, D: int):
    self.loc = p[:, :D]
    self.raw_scale = p[:, D:]
    self.scale = torch.nn.functional.softplus(self.raw_scale)
    self.z = z
    self.KL_divergence = KLdivergence(self)

  event_shape = property(lambda self: (self.loc.shape[-1],))
  batch_shape = property(lambda self: tuple(self.loc.shape[:-1]))

  def mean(self):
    return self.loc

  def stddev(self):
    return self.scale

  def sample(self, n=None, seed=None):
    g = None
    if seed is not None:
      g = torch.Generator(device=self.loc.device).manual_seed(int(seed))
    shp = tuple(self.loc.shape) if n is None else (int(n),) + tuple(self.loc.shape)
    return self.loc + self.scale * torch.randn(shp, device=self.loc.device, generator=g)

  def log_prob(self, z):
    D = self.loc.shape[-1]
    return (-0.5 * ((z - self.loc) / self.scale) ** 2 - torch.log(self.scale)).sum(-1) \
        - 0.5 * D * LOG2PI

  def __array__(self):
    return self.z.detach().cpu().numpy()

  def tensor(self):
    """tf.convert_to_tensor(q): the cached sample."""
    return self.z


class BernoulliObservation:
  """Independent(Bernoulli(logits), 3) (odin/networks/image_networks.py:87-93)."""

  def __init__(self, logits: torch.Tensor):
    self.logits = logits

  event_shape = property(lambda self: tuple(self.logits.shape[1:]))
  batch_shape = property(lambda self: (self.logits.shape[0],))

  def mean(self):
    return torch.sigmoid(self.logits)

  def log_prob(self, x):
    e = x * self.logits - torch.nn.functional.softplus(self.logits)
    return e.reshape(e.shape[0], -1).sum(1)

  def sample(self, n=None):
    p = self.mean()
    if n is not None:
      p = p.unsqueeze(0).expand((int(n),) + tuple(p.shape))
    return torch.bernoulli(p)


class GaussianObservation:
  """Independent(Normal(loc, scale), 3), params split on the channel axis
  (image_networks.py:95-102); softplus1 scale as GaussianLayer (continuous.py:196-260)."""

  def __init__(self, h: torch.Tensor, softplus1: bool):
    Cc = h.shape[-1] // 2
    self.loc = h[..., :Cc]
    raw = h[..., Cc:]
    self.scale = torch.nn.functional.softplus(raw + math.log(math.e - 1.0)) if softplus1 else raw

  event_shape = property(lambda self: tuple(self.loc.shape[1:]))
  batch_shape = property(lambda self: (self.loc.shape[0],))

  def mean(self):
    return self.loc

  def stddev(self):
    return self.scale

  def log_prob(self, x):
    e = -0.5 * ((x - self.loc) / self.scale) ** 2 - torch.log(self.scale) - 0.5 * LOG2PI
    return e.reshape(e.shape[0], -1).sum(1)

  def sample(self, n=None):
    shp = tuple(self.loc.shape) if n is None else (int(n),) + tuple(self.loc.shape)
    return self.loc + self.scale * torch.randn(shp, device=self.loc.device)


class QuantizedLogisticObservation:
  """QuantizedLogistic(loc, softplus(raw) + e^-7, low=0, high=255, inputs_domain='sigmoid',
  reinterpreted_batch_ndims=3) (image_networks.py:55-71; distributions/quantized.py:50-204)."""

  def __init__(self, h: torch.Tensor):
    Cc = h.shape[-1] // 2
    self.loc = 127.5 * (h[..., :Cc] + 1.0)                                      # pixel units
    self.scale = (torch.nn.functional.softplus(h[..., Cc:]) + math.exp(-7.0)) * 127.5

  event_shape = property(lambda self: tuple(self.loc.shape[1:]))
  batch_shape = property(lambda self: (self.loc.shape[0],))

  def mean(self):
    return self.loc / 255.0      # quantized.py:185-187 (`_pixels_to`, sigmoid domain)

  def stddev(self):
    return self.scale * (math.pi / math.sqrt(3.0))

  def _logcdf(self, j):
    r = -torch.nn.functional.softplus(-(j + 0.5 - self.loc) / self.scale)
    r = torch.where(j < 0.0, torch.full_like(r, -float('inf')), r)
    return torch.where(j < 255.0, r, torch.zeros_like(r))

  def _logsf(self, j):
    r = -torch.nn.functional.softplus((j + 0.5 - self.loc) / self.scale)
    r = torch.where(j < 0.0, torch.zeros_like(r), r)
    return torch.where(j < 255.0, r, torch.full_like(r, -float('inf')))

  def log_prob(self, x):
    e = self.log_prob_elem(x)
    return e.reshape(e.shape[0], -1).sum(1)

  def log_prob_elem(self, x):
    y = x * 255.0
    lsy, lsy1 = self._logsf(torch.ceil(y)), self._logsf(torch.ceil(y - 1.0))
    lcy, lcy1 = self._logcdf(torch.floor(y)), self._logcdf(torch.floor(y - 1.0))
    use_sf = lsy < lcy
    big, small = torch.where(use_sf, lsy1, lcy), torch.where(use_sf, lsy, lcy1)
    fin = torch.isfinite(small)
    d = torch.where(fin, big - small, torch.ones_like(big))
    l1m = torch.where(d < math.log(2.0), torch.log(-torch.expm1(-d)), torch.log1p(-torch.exp(-d)))
    return big + torch.where(fin, l1m, torch.zeros_like(l1m))

  def sample(self, n=None):
    shp = tuple(self.loc.shape) if n is None else (int(n),) + tuple(self.loc.shape)
    u = torch.rand(shp, device=self.loc.device).clamp_(1e-6, 1 - 1e-6)
    xs = self.loc + self.scale * (torch.log(u) - torch.log1p(-u)) - 0.5  # Logistic shifted by -1/2
    return torch.ceil(xs).clamp_(0.0, 255.0) / 255.0


class MixtureQuantizedLogisticObservation:
  """MixtureQuantizedLogistic(params, n_components=10, n_channels=C, low=0, high=255,
  inputs_domain='sigmoid') (distributions/quantized.py:206-381; image_networks.py:72-85): the
  decoder's K * (1 + 2C + C(C-1)/2) maps are (mixture logit | loc | raw scale | channel coefficients)
  per component."""

  def __init__(self, h: torch.Tensor, n_channels: int, n_components: int = 10):
    C, K = int(n_channels), int(n_components)
    no = 2 * C + C * (C - 1) // 2 + 1
    assert h.shape[-1] == K * no, (h.shape, K, no)
    hh = h.reshape(tuple(h.shape[:-1]) + (K, no))
    self.C, self.K = C, K
    self.logits, self.locs = hh[..., 0], hh[..., 1:1 + C]
    self.scales = torch.nn.functional.softplus(hh[..., 1 + C:1 + 2 * C]) + math.exp(-7.0)
    self.coefs = hh[..., 1 + 2 * C:]
    self._shape = tuple(h.shape[1:-1]) + (C,)

  event_shape = property(lambda self: self._shape)
  batch_shape = property(lambda self: (self.logits.shape[0],))

  def _chain(self, base):
    """loc_i += sum_{j<i} base_j * coef (quantized.py:315-320 with the transformed pixel values,
    :360-364 with the component means)."""
    cols = [self.locs[..., i] for i in range(self.C)]
    cnt = 0
    for i in range(self.C):
      for j in range(i):
        cols[i] = cols[i] + (base[..., None, j] if base is not None else cols[j]) * self.coefs[..., cnt]
        cnt += 1
    return torch.stack(cols, -1)

  def mean(self):
    m = 127.5 * (self._chain(None) + 1.0) - 0.5     # Shift(-0.5) of the base logistic (:373-375)
    pi = torch.softmax(self.logits, -1)
    return (pi[..., None] * m).sum(-2) / 255.0      # `_pixels_to`, sigmoid domain

  def log_prob(self, x):
    le = self._chain(2.0 * x - 1.0)
    comp = QuantizedLogisticObservation.__new__(QuantizedLogisticObservation)
    comp.loc, comp.scale = 127.5 * (le + 1.0), self.scales * 127.5
    xb = x[..., None, :].expand(le.shape)
    B = x.shape[0]
    # per (pixel, component, channel) terms, then the mixture over components, then the pixels
    q = comp.log_prob_elem(xb)
    pix = torch.logsumexp(torch.log_softmax(self.logits, -1) + q.sum(-1), -1)
    return pix.reshape(B, -1).sum(1)

  def sample(self, n=None):
    """The reference leaves sampling as a TODO that returns uniform noise of the image shape
    (quantized.py:383-386); restated as such."""
    shp = (self.logits.shape[0],) + self._shape
    if n is not None:
      shp = (int(n),) + shp
    return torch.rand(shp, device=self.logits.device)


# ======================================================================================
# training step objects
# ======================================================================================
@dataclass
class TrainStep:
  """odin/networks/base_networks.py:130-173"""
  inputs: Any = None
  training: bool = True
  mask: Any = None
  parameters: Any = None
  optimizer: Any = None
  name: str = ''

  def call(self):
    raise NotImplementedError

  def __call__(self):
    return self.call()


@dataclass
class VAEStep(TrainStep):
  """variational_autoencoder.py:111-126: loss = -mean(elbo(llk, kl)); metrics = means."""
  vae: 'VariationalAutoencoder' = None
  call_kw: Dict[str, Any] = field(default_factory=dict)

  def call(self):
    llk, kl = self.vae.elbo_components(self.inputs, training=self.training, mask=self.mask,
                                       **self.call_kw)
    loss = -torch.mean(self.vae.elbo(llk, kl))
    metrics = {k: torch.mean(v) for k, v in llk.items()}
    metrics.update({k: torch.mean(v) if torch.is_tensor(v) else v for k, v in kl.items()})
    return loss, metrics


# ======================================================================================
# the model
# ======================================================================================
def _as_tensor(x, device):
  if torch.is_tensor(x):
    return x.to(device=device, dtype=torch.float32).contiguous()
  return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=device).contiguous()


class VariationalAutoencoder:
  """See module docstring.  Construction mirrors
  ``vae_cls(**get_networks(ds_name))`` of the reference (examples/vae/utils.py:231-256)."""

  def __init__(self, observation: RVconf = None, latents: RVconf = None,
               encoder: SequentialNetwork = None, decoder: SequentialNetwork = None,
               analytic: bool = False, reverse: bool = True, free_bits: Optional[float] = None,
               sample_shape=(), allow_negative_kl: bool = True, path: Optional[str] = None,
               step: int = 0, name: str = 'VariationalAutoencoder', device=None, seed: int = 1,
               lib=None, **kwargs):
    if encoder is None or decoder is None or latents is None or observation is None:
      d = get_networks('dense')
      encoder = encoder or d['encoder']
      decoder = decoder or d['decoder']
      latents = latents or d['latents']
      observation = observation or d['observation']
    for n, net in (('encoder', encoder), ('decoder', decoder)):
      if not isinstance(net, SequentialNetwork):
        raise ValueError(f'{n} must be a SequentialNetwork description, got {type(net)}')
    if latents.posterior not in ('mvndiag', 'diag', 'normaldiag'):
      raise ValueError(f"latents posterior {latents.posterior!r} is outside the HIP path "
                       f"(supported: 'mvndiag')")
    sample_shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
    if len(sample_shape) > 1:
      raise ValueError(f'sample_shape={sample_shape}: at most one sample axis')
    if not reverse and not analytic:
      raise TypeError('reverse=False needs analytic=True (the reference cannot evaluate the '
                      'Monte-Carlo form of KL(p||q) either, helpers.py:261-276)')
    self.observation, self.latents, self.encoder, self.decoder = observation, latents, encoder, decoder
    self.analytic, self.reverse, self.free_bits = bool(analytic), bool(reverse), free_bits
    self.sample_shape, self.allow_negative_kl = tuple(sample_shape), allow_negative_kl
    self.path, self.name = path, name
    self._step = int(step)
    self.seed = int(seed)
    self.device = torch.device(device if device is not None else
                               ('cuda' if torch.cuda.is_available() else 'cpu'))
    self._lib = lib
    self._engines: Dict[int, VAEEngine] = {}
    self._params: Optional[torch.Tensor] = None
    self._optim_state = None
    self._last_outputs = None
    self._tc_mode = None
    self._capacity_mode = False
    self.trainer = None
    self.input_shape = tuple(encoder.input_shape) if encoder.input_shape else None
    self.zdim = int(latents.event_size)
    if self.input_shape is not None:
      self.build((None,) + self.input_shape)

  # ------------------------------------------------------------------ construction
  def build(self, input_shape):
    shape = tuple(input_shape)[1:]
    self.input_shape = shape
    eng = self._engine(1)  # validates shapes, allocates + initialises parameters
    return self

  def input_buffer(self, batch_size: int) -> torch.Tensor:
    """The float32 [B, H, W, C] tensor the captured training-step graph for this batch size
    reads: hand it to `odin_ai_amd.data.DeviceImageDataset(out=...)` (or fill it in place) and
    pass the batches it yields to `fit` / `optimize` -- no per-step input copy."""
    return self._engine(int(batch_size)).input_buffer()

  def _engine(self, B: int) -> VAEEngine:
    if self.input_shape is None:
      raise RuntimeError('model is not built: call build((None, H, W, C)) first')
    if B not in self._engines:
      eng = VAEEngine(self.encoder.layers, self.decoder.layers, self.input_shape, self.zdim, B,
                      self.device, observation=self.observation.posterior,
                      analytic=self.analytic, reverse=self.reverse, free_bits=self.free_bits,
                      tc=self._tc_mode, capacity=self._capacity_mode,
                      lib=self._lib, params=self._params, seed=self.seed + self._rank(),
                      optim_state=self._optim_state, world_size=self._world_size(),
                      force_dp=bool(getattr(self, 'force_dp', False)),
                      **getattr(self, 'engine_options', {}))
      if self._params is None:
        self._params = eng.params
        self._optim_state = (eng.m, eng.v)
        self._init_parameters(eng)
        # data parallel: every replica starts from rank 0's weights; the noise streams differ
        # per rank (seed + rank above), the weights must not
        from .dist import broadcast_parameters
        broadcast_parameters(eng.params, src=0)
      self._engines[B] = eng
    return self._engines[B]

  def _world_size(self) -> int:
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

  def _rank(self) -> int:
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0

  def _init_parameters(self, eng: VAEEngine):
    """HeNormal for elu convs, glorot_uniform for Dense, glorot_normal for the latent
    projection, zero biases (image_networks.py:157-174; dense_distribution.py:124)."""
    g = torch.Generator(device='cpu').manual_seed(self.seed)
    deconv = {r.key for r in eng.enc_recs + eng.dec_recs if r.kind == 'deconv'}
    for key, shp, off in eng.layout.entries:
      n = int(np.prod(shp))
      if key[-1] == 'b':
        eng.params[off:off + n].zero_()
        continue
      if len(shp) == 4:
        fan_in = shp[0] * shp[1] * (shp[3] if key[:2] in deconv else shp[2])
        w = torch.nn.init.trunc_normal_(torch.empty(n), 0.0, 1.0, -2.0, 2.0, generator=g)
        w = w * math.sqrt(2.0 / fan_in) / 0.87962566
      elif key[0] == 'lat':
        w = torch.randn(n, generator=g) * math.sqrt(2.0 / (shp[0] + shp[1]))
      else:
        lim = math.sqrt(6.0 / (shp[0] + shp[1]))
        w = (torch.rand(n, generator=g) * 2 - 1) * lim
      eng.params[off:off + n] = w.to(self.device)

  # ------------------------------------------------------------------ properties
  @property
  def step(self) -> int:
    return self._step

  @property
  def n_parameters(self) -> int:
    return self._engine(1).n_params

  @property
  def last_outputs(self):
    return self._last_outputs

  @property
  def trainable_variables(self) -> Dict[tuple, torch.Tensor]:
    return self._engine(1).param_views()

  def set_elbo_configs(self, analytic=None, reverse=None, free_bits=None, sample_shape=None):
    """odin/bay/vi/_base.py:51-89"""
    if analytic is not None:
      self.analytic = bool(analytic)
    if reverse is not None:
      self.reverse = bool(reverse)
    if free_bits is not None:
      self.free_bits = free_bits
    if sample_shape is not None:
      self.sample_shape = (sample_shape,) if isinstance(sample_shape, int) else tuple(sample_shape)
    for e in self._engines.values():
      e.set_kl_form(self.analytic, self.reverse)
      e.free_bits = -1.0 if self.free_bits is None else float(self.free_bits)
    return self

  @property
  def n_samples(self) -> int:
    """prod(sample_shape): MC samples of z per input (variational_autoencoder.py:288-314)."""
    n = 1
    for i in self.sample_shape:
      n *= int(i)
    return n

  def _tile(self, x: torch.Tensor) -> torch.Tensor:
    """sample_shape = (n,): the reference draws z of shape [n, B, D] from one encoder pass and
    decodes n*B codes; llk / kl come out as [n, B] and the loss is their overall mean.  Here
    the batch is tiled n times (each copy gets its own noise): the same [n, B] values and the
    same gradients, at the price of n encoder passes."""
    n = self.n_samples
    return x if n == 1 else x.repeat((n,) + (1,) * (x.dim() - 1))

  @property
  def beta(self) -> float:
    return 1.0

  def _hyper_extra(self) -> dict:
    """per-step scalars beyond (lr, beta) that a subclass hands to the engine (BetaCapacityVAE: the capacity)"""
    return {}

  # ------------------------------------------------------------------ forward API
  def _posterior(self, eng: VAEEngine) -> MVNDiagPosterior:
    return MVNDiagPosterior(eng.p.clone(), eng.z.clone(), eng.D)

  def _observation_dist(self, h: torch.Tensor):
    if self.observation.posterior == 'bernoulli':
      return BernoulliObservation(h)
    if self.observation.posterior == 'qlogistic':
      return QuantizedLogisticObservation(h)
    if self.observation.posterior == 'mixqlogistic':
      return MixtureQuantizedLogisticObservation(h, self.input_shape[-1],
                                                 self.observation.kwargs.get('n_components', 10))
    return GaussianObservation(h, self.observation.posterior == 'gaussian_softplus1')

  def encode(self, inputs, training=None, mask=None, only_encoding=False, eps=None, **kwargs):
    """variational_autoencoder.py:288-314"""
    x = _as_tensor(inputs, self.device)
    eng = self._engine(x.shape[0])
    eng.set_hyper(beta=self.beta, t=self._step, **self._hyper_extra())
    eng.run_encoder(x, None if eps is None else _as_tensor(eps, self.device))
    if only_encoding:
      return eng.enc.outs[-1].clone()
    return self._posterior(eng)

  def decode(self, latents, training=None, mask=None, only_decoding=False, **kwargs):
    """variational_autoencoder.py:316-360"""
    z = latents.tensor() if isinstance(latents, MVNDiagPosterior) else latents
    z = _as_tensor(z, self.device)
    eng = self._engine(z.shape[0])
    h = eng.run_decoder(z).clone()
    return h if only_decoding else self._observation_dist(h)

  def call(self, inputs, training=None, mask=None, eps=None, **kwargs):
    """variational_autoencoder.py:362-394 -> (p(x|z), q(z|x))"""
    qz_x = self.encode(inputs, training=training, mask=mask, eps=eps)
    px_z = self.decode(qz_x, training=training, mask=mask)
    self._last_outputs = (px_z, qz_x)
    return px_z, qz_x

  __call__ = call

  def elbo_components(self, inputs, training=None, mask=None, eps=None, **kwargs):
    """variational_autoencoder.py:515-542.  One fused engine pass: llk [B], kl [B]
    (already multiplied by beta for the Beta family, beta_vae.py:38-43)."""
    x = self._tile(_as_tensor(inputs, self.device))
    n = self.n_samples
    eng = self._engine(x.shape[0])
    eng.set_hyper(beta=self.beta, t=self._step, **self._hyper_extra())
    eng.forward(x, None if eps is None else _as_tensor(eps, self.device).reshape(x.shape[0], -1))
    qz_x = self._posterior(eng)
    px_z = self._observation_dist(eng.dec.outs[-1].clone())
    self._last_outputs = (px_z, qz_x)
    shp = (lambda t: t.reshape(n, -1)) if n > 1 else (lambda t: t)
    llk = {f'llk_{self.observation.name}': shp(eng.llk.clone())}
    klv = shp(eng.kl.clone() * self.beta)
    if self.analytic:
      klv = klv[:1] if n > 1 else klv.unsqueeze(0)  # [1, B] (helpers.py:370-371)
    kl = {f'kl_{self.latents.name}': klv}
    if self._tc_mode == 'betatc':
      kl[f'tc_{self.latents.name}'] = (self.beta - 1.0) * eng.tc_ws[0].clone()
    return llk, kl

  def elbo(self, llk: Dict[str, torch.Tensor], kl: Dict[str, torch.Tensor]) -> torch.Tensor:
    """odin/bay/vi/_base.py:151-194: sum(llk) - sum(kl), broadcasting scalars / [1,B]."""
    L = 0.0
    for v in llk.values():
      L = L + v
    K = 0.0
    for k, v in kl.items():
      if not self.allow_negative_kl and torch.is_tensor(v):
        if bool((v < -1e-3).any()):
          raise AssertionError(f'negative KL for {k}')
      K = K + v
    return L - K

  def sample_prior(self, n: int = 1, seed: int = 1) -> torch.Tensor:
    g = torch.Generator(device='cpu').manual_seed(int(seed))
    return torch.randn(int(n), self.zdim, generator=g).to(self.device)

  def sample_observation(self, n: int = 1, seed: int = 1, training=False, **kwargs):
    return self.decode(self.sample_prior(n, seed=seed), training=training)

  def marginal_log_prob(self, inputs, training=None, n_mcmc: Optional[int] = 100,
                        reduce: Optional[Callable] = torch.mean, batch_size: int = 32,
                        verbose: bool = False, eps=None, **kwargs):
    """variational_autoencoder.py:396-513: per batch ONE encoder pass, n_mcmc posterior samples
    per input, one decoder pass over the n_mcmc*B codes, log-mean-exp over the samples on
    device.  Returns the reference's structure: ({name: llk}, {name: (log q, log p)}) with
    llk = logsumexp_k log p(x|z_k) - log n and the same log-mean-exp of log q(z_k|x), log p(z_k);
    `reduce` (default mean over inputs) is applied to each, None keeps the per-input vectors.
    `eps` ([n_mcmc, N, D], optional) fixes the noise for parity tests."""
    x_all = _as_tensor(inputs, self.device)
    n = int(n_mcmc) if n_mcmc is not None else self.n_samples
    N, D = x_all.shape[0], self.zdim
    outs = {'llk': [], 'lq': [], 'lp': []}
    f32 = dict(dtype=torch.float32, device=self.device)
    for i0 in range(0, N, int(batch_size)):
      x = x_all[i0:i0 + int(batch_size)].contiguous()
      B = x.shape[0]
      eng = self._engine(B)
      lib, st = eng.lib, eng.stream()
      eng.set_hyper(beta=1.0, t=self._step, **self._hyper_extra())
      eng.run_encoder(x)  # fills eng.p = (loc | raw scale)
      if eps is None:
        e = torch.empty(n, B, D, **f32)
        lib.odin_rng_normal(e.data_ptr(), n * B * D, self.seed + 7919 + i0, eng.hp(N_HYPER), st)
      else:
        e = _as_tensor(eps, self.device)[:, i0:i0 + B].contiguous()
      z = torch.empty(n, B, D, **f32)
      lq, lp, llk = torch.empty(n, B, **f32), torch.empty(n, B, **f32), torch.empty(n, B, **f32)
      lib.odin_latent_sample_logprob(eng.p.data_ptr(), e.data_ptr(), z.data_ptr(), lq.data_ptr(),
                                     lp.data_ptr(), n, B, D, st)
      # decode the n*B codes in chunks of whole sample-rows (bounded activation memory)
      rows = max(1, min(n, 2048 // max(B, 1)))
      for k0 in range(0, n, rows):
        r = min(rows, n - k0)
        de = self._engine(r * B)
        de.set_hyper(beta=1.0, t=self._step, **self._hyper_extra())
        h = de.run_decoder(z[k0:k0 + r].reshape(r * B, D))
        xt = x.repeat((r,) + (1,) * (x.dim() - 1))
        de.observation_llk(h, xt, llk[k0:k0 + r].reshape(r * B))
      for key, src in (('llk', llk), ('lq', lq), ('lp', lp)):
        o = torch.empty(B, **f32)
        lib.odin_logmeanexp_rows(src.data_ptr(), o.data_ptr(), n, B, st)
        outs[key].append(o)
    cat = {k: torch.cat(v) for k, v in outs.items()}
    if reduce is not None:
      cat = {k: reduce(v) for k, v in cat.items()}
    return ({self.observation.name: cat['llk']},
            {self.latents.name: (cat['lq'], cat['lp'])})

  # ------------------------------------------------------------------ training
  def train_steps(self, inputs, training=None, mask=None, name: str = '', **kwargs
                  ) -> Iterator[VAEStep]:
    """variational_autoencoder.py:545-558"""
    yield VAEStep(vae=self, parameters=self.trainable_variables, inputs=inputs,
                  training=training, mask=mask, name=name, call_kw=kwargs)

  def _lr(self, learning_rate) -> float:
    return float(learning_rate(self._step)) if callable(learning_rate) else float(learning_rate)

  def optimize(self, inputs, training: bool = True, optimizer=None, learning_rate=1e-4,
               clipnorm=None, clipvalue=None, global_clipnorm=None, skip_update_threshold=None,
               when_skip_update: int = 0, nan_gradients_policy: str = 'skip',
               allow_none_gradients=False, aggregate_gradients=False, track_gradients=False,
               eps=None, use_graph: bool = False):
    """Networks.optimize (base_networks.py:415-624): step += 1; forward; backward; NaN policy;
    skip_update_threshold -> per-variable clipnorm -> clip_by_global_norm -> clipvalue (the
    reference's order, :549-596); Adam.  Returns (loss, metrics) as device scalars.

    nan_gradients_policy: 'ignore' applies the update whatever the gradients hold; every other
    policy leaves parameters and optimiser state untouched on device when a gradient is not
    finite and raises `nan_flag` (polled by `fit`, which stops / raises / restores the last
    checkpoint).  (The reference zeroes the gradients with `g - g`, which is NaN again for a NaN
    gradient, and hands them to Adam: a step it cannot survive.  Not reproduced.)
    aggregate_gradients only matters for multi-step models (FactorVAE); track_gradients adds the
    gradient tensors as `_grad/<variable>` metrics (:609-611)."""
    if nan_gradients_policy not in ('stop', 'skip', 'raise', 'ignore', 'restore'):
      raise ValueError(f'nan_gradients_policy={nan_gradients_policy!r}')
    x = self._tile(_as_tensor(inputs, self.device))
    eng = self._engine(x.shape[0])
    if eps is not None:
      eps = _as_tensor(eps, self.device).reshape(x.shape[0], -1)
    if training:
      self._step += 1
    eng.step_count = self._step - 1 if training else self._step
    if not training:
      eng.set_hyper(beta=self.beta, t=self._step, **self._hyper_extra())
      eng.forward(x, eps)
    else:
      eng.train_step(x, eps,
                     lr=self._lr(learning_rate), beta=self.beta,
                     global_clipnorm=global_clipnorm, use_graph=use_graph, clipnorm=clipnorm,
                     clipvalue=clipvalue, skip_update_threshold=skip_update_threshold,
                     when_skip_update=when_skip_update,
                     check_nan=nan_gradients_policy != 'ignore', **self._hyper_extra())
      self._step = eng.step_count
    out = eng.out4.clone()
    metrics = {f'llk_{self.observation.name}': out[1], f'kl_{self.latents.name}': out[2]}
    if self._tc_mode == 'betatc':
      metrics[f'tc_{self.latents.name}'] = out[3]
    if training and track_gradients:
      for k, g in eng.grad_views().items():
        metrics['_grad/' + self.variable_name(k)] = g.clone()
    return out[0], metrics

  @property
  def nan_flag(self) -> bool:
    """True when a training step met non-finite gradients and skipped its update (host sync)."""
    return any(int(e.flag.item()) != 0 for e in self._engines.values())

  @property
  def skipped_update(self) -> int:
    """Networks.skipped_update (base_networks.py:573-577): updates zeroed by skip_update_threshold."""
    return sum(int(e.skipped_update.item()) for e in self._engines.values())

  def variable_name(self, key: tuple) -> str:
    """Keras variable name of a parameter key: the reference's layer names
    (image_networks.py:248-268,463-511: encoder0.., encoder_proj, decoder_proj, decoder1..;
    DistributionDense 'latents', dense_distribution.py:229-238) + /kernel | /bias."""
    suffix = 'kernel' if key[-1] == 'w' else 'bias'
    if key[0] == 'lat':
      return f'{self.latents.name}/{suffix}'
    net = self.encoder if key[0] == 'enc' else self.decoder
    names = layer_names(net, 'encoder' if key[0] == 'enc' else 'decoder')
    return f'{names[key[1]]}/{suffix}'

  def fit(self, train, *, valid=None, valid_freq: int = 500, valid_interval: float = 0,
          optimizer='adam', learning_rate=1e-4, clipnorm=None, global_clipnorm=None,
          clipvalue=None, skip_update_threshold=None, when_skip_update=None, epochs: int = -1,
          max_iter: int = 1000, batch_size: int = 32, on_batch_end=None, on_valid_end=None,
          compile_graph: bool = True, autograph: bool = False, logging_interval: float = 5,
          skip_fitted: Union[bool, int] = False, nan_gradients_policy: str = 'stop',
          logdir=None, allow_none_gradients=False, track_gradients=False, seed: int = 1,
          nan_check_interval: int = 50):
    """Networks.fit (base_networks.py:642-812) with Trainer.fit's cadence (training/trainer.py:536-738).
    `train` / `valid`: array / tensor [N,H,W,C] or an iterable of batches.  compile_graph -> the step is
    replayed as one HIP graph.  Validation (mean loss / metrics of the training=False step over `valid`,
    `last_valid_loss`, `last_valid_metrics`, `valid/*` scalars) and the `on_valid_end` callback run at the first
    iteration, whenever step % valid_freq == 0 and `valid_interval` seconds have passed (`valid_interval` > 0
    makes valid_freq 1), and once more when training ends; `on_batch_end` after every step; `train/*` scalars
    at most every `logging_interval` seconds (metrics whose name starts with '_' stay hidden)."""
    if optimizer not in ('adam', None) and not callable(optimizer):
      raise RuntimeError(f'No support for optimizer {optimizer!r} on the HIP path (adam only)')
    if nan_gradients_policy not in ('stop', 'skip', 'raise', 'ignore', 'restore'):
      raise ValueError(nan_gradients_policy)
    if skip_fitted and self._step >= (max_iter if skip_fitted is True else int(skip_fitted)):
      return self
    history = []

    def batches():
      if torch.is_tensor(train) or isinstance(train, np.ndarray):
        data = _as_tensor(train, self.device).contiguous()
        N = data.shape[0]
        g = torch.Generator(device='cpu').manual_seed(seed)
        ep = 0
        n_per = int(np.prod(data.shape[1:]))
        # the shuffled batch is gathered by ONE launch straight into the tensor the step graph reads (no torch
        # advanced-indexing kernel, no per-step copy into the static buffer)
        direct = (data.dtype == torch.float32 and n_per % 4 == 0 and tuple(data.shape[1:]) == tuple(self.input_shape)
                  and N >= batch_size)
        if direct:
          eng0 = self._engine(int(batch_size))
          xb0 = eng0.input_buffer()
        while epochs < 0 or ep < epochs:
          perm = torch.randperm(N, generator=g).to(device=self.device, dtype=torch.int32)
          for i in range(0, N - batch_size + 1, batch_size):
            if direct:
              eng0.lib.odin_gather_rows_f32(data.data_ptr(), perm.data_ptr() + 4 * i, xb0.data_ptr(),
                                            int(batch_size), n_per, eng0.stream())
              yield xb0
            else:
              yield data[perm[i:i + batch_size].long()].contiguous()
          ep += 1
      else:
        ep = 0
        while epochs < 0 or ep < epochs:
          for b in train:
            yield _as_tensor(b, self.device)
          ep += 1

    # ---- validation / logging cadence of Trainer.fit (training/trainer.py:607-700): `valid_interval` > 0 takes
    # precedence over `valid_freq`; validation (and ALWAYS the on_valid_end callback) runs at the first
    # iteration, then whenever step % valid_freq == 0 and valid_interval seconds have passed, and once more
    # when training ends; train summaries are written at most every `logging_interval` seconds
    import time as _time
    valid_freq = max(1, int(valid_freq))
    valid_interval = float(valid_interval)
    if valid_interval > 0:
      valid_freq = 1

    def valid_batches():
      if torch.is_tensor(valid) or isinstance(valid, np.ndarray):
        data = _as_tensor(valid, self.device)
        N = data.shape[0]
        bs = min(batch_size, N)
        for i in range(0, N - bs + 1, bs):
          yield data[i:i + bs].contiguous()
      else:
        for b in valid:
          yield _as_tensor(b, self.device)

    def run_valid():
      losses, mets = [], {}
      for vb in valid_batches():
        l, m = next(iter(self.train_steps(vb, training=False)))()
        losses.append(float(l))
        for k, v in m.items():
          mets.setdefault(k, []).append(float(v))
      if not losses:
        return None, {}
      return float(np.mean(losses)), {k: float(np.mean(v)) for k, v in mets.items()}

    def events(eng):
      if getattr(self, '_events', None) is None or self._events_dir != logdir:
        from .tf_checkpoint import ScalarEventWriter
        self._events, self._events_dir = ScalarEventWriter(logdir, lib=eng.lib), logdir
      return self._events

    def validate(eng):
      if valid is not None:
        vl, vm = run_valid()
        self.last_valid_loss, self.last_valid_metrics = vl, vm
        if vl is not None:
          self.valid_history.append((self._step, vl))
          if logdir is not None:
            ev = events(eng)
            ev.scalar('valid/loss', vl, self._step)
            for mk, mv in vm.items():
              if not mk.startswith('_'):
                ev.scalar(f'valid/{mk}', mv, self._step)
            ev.flush()
      if on_valid_end is not None:  # (the callback is always called, with or without a validation set)
        on_valid_end()

    self.valid_history = getattr(self, 'valid_history', [])
    self.last_valid_loss, self.last_valid_metrics = None, {}
    t_start = _time.monotonic()
    last_log, last_valid = -float('inf'), t_start
    it = 0
    eng = None
    for xb in batches():
      if it >= max_iter:
        break
      loss, metrics = self.optimize(xb, training=True, learning_rate=learning_rate,
                                    clipnorm=clipnorm, clipvalue=clipvalue,
                                    global_clipnorm=global_clipnorm,
                                    skip_update_threshold=skip_update_threshold,
                                    when_skip_update=when_skip_update or 0,
                                    nan_gradients_policy=nan_gradients_policy,
                                    track_gradients=track_gradients,
                                    use_graph=compile_graph and self.device.type == 'cuda')
      it += 1
      eng = self._engine(xb.shape[0])
      self.last_train_loss, self.last_train_metrics = loss, metrics
      if on_batch_end is not None:
        on_batch_end()
      now = _time.monotonic()
      # the NaN flag is sticky on device; polling it costs a host sync, so it is read every
      # `nan_check_interval` iterations (and at the end) unless the policy must act at once
      if it % nan_check_interval == 0 or it == max_iter:
        if int(eng.flag.item()) != 0:  # non-finite gradients: the update was skipped on device
          if nan_gradients_policy == 'raise':
            raise RuntimeError(f'NaNs gradient! (step {self._step})')
          if nan_gradients_policy == 'stop':
            break
          if nan_gradients_policy == 'restore':  # fall back to the last checkpoint (:543-545)
            self.load_weights(raise_notfound=False)
          eng.flag.zero_()
        history.append((self._step, float(loss)))
        # Trainer's TensorBoard scalars (training/trainer.py:52-71), at most every logging_interval seconds
        if logdir is not None and now - last_log >= float(logging_interval):
          ev = events(eng)
          ev.scalar('train/loss', float(loss), self._step)
          for mk, mv in metrics.items():
            if not mk.startswith('_'):   # (metrics hidden with a leading '_', trainer.py:662)
              ev.scalar(f'train/{mk}', float(mv), self._step)
          ev.flush()
          last_log = now
      if it == 1 or (self._step % valid_freq == 0 and now - last_valid >= valid_interval):
        validate(eng)
        last_valid = _time.monotonic()
    if eng is not None:
      validate(eng)  # final callback: training ended (trainer.py:704-709)
    self.history = history
    return self

  # ------------------------------------------------------------------ checkpoints
  def save_weights(self, filepath: Optional[str] = None, overwrite: bool = True,
                   save_format: str = 'tf'):
    """base_networks.py:373-390: weights + step (optimizer state deliberately not tracked).
    save_format='tf' (the reference's, :386): a TensorFlow checkpoint `<filepath>.index` +
    `<filepath>.data-00000-of-00001` whose object graph names every variable by its Keras name
    (`encoder0/kernel`, ..., `latents/bias`, `Step`) -- readable by `tf.train.load_checkpoint`;
    save_format='npz': one plain `.npz` with the same names.  No pickle either way."""
    filepath = filepath or self.path
    if filepath is None:
      raise ValueError('No path is given for saving weights')
    eng = self._engine(1)
    W = {self.variable_name(k): v.detach().cpu().numpy() for k, v in eng.param_views().items()}
    W.update(self._extra_checkpoint_variables())
    if save_format == 'tf':
      if os.path.exists(filepath + '.index') and not overwrite:
        raise RuntimeError(f'{filepath} exists')
      from . import tf_checkpoint
      W['Step'] = np.asarray(self._step, np.int64)   # base_networks.py:212
      tf_checkpoint.save_checkpoint(filepath, W, lib=eng.lib)
    elif save_format == 'npz':
      if os.path.exists(self._npz_path(filepath)) and not overwrite:
        raise RuntimeError(f'{filepath} exists')
      W['__step__'] = np.asarray(self._step, np.int64)
      with open(self._npz_path(filepath), 'wb') as f:
        np.savez(f, **W)
    else:
      raise ValueError(f"save_format={save_format!r} ('tf' | 'npz')")
    return self

  @staticmethod
  def _npz_path(filepath: str) -> str:
    return filepath if filepath.endswith('.npz') else filepath + '.npz'

  def load_weights(self, filepath: Optional[str] = None, raise_notfound: bool = False):
    """base_networks.py:338-371.  Reads a TensorFlow checkpoint (also one written by the
    reference: variables are found by their Keras names inside the checkpoint's object graph,
    whatever the object paths) or the `.npz` form."""
    filepath = filepath or self.path
    if filepath is not None and os.path.exists(filepath + '.index'):
      from . import tf_checkpoint
      d = tf_checkpoint.load_checkpoint(filepath, lib=self._engine(1).lib)
      step_keys = [k for k in d if k == 'Step' or k.endswith('/Step')]
      step = int(d[step_keys[0]]) if step_keys else self._step
    elif filepath is not None and os.path.exists(self._npz_path(filepath)):
      d = dict(np.load(self._npz_path(filepath), allow_pickle=False))
      step = int(d.pop('__step__'))
    else:
      if raise_notfound:
        raise FileNotFoundError(f'Cannot find saved weights at path: {filepath}')
      return self
    eng = self._engine(1)
    for k, v in eng.param_views().items():
      name = self.variable_name(k)
      a = self._find_variable(d, name)
      if tuple(a.shape) != tuple(v.shape):
        raise ValueError(f'{name}: checkpoint shape {a.shape} != {tuple(v.shape)}')
      v.copy_(torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device))
    self._load_extra_checkpoint_variables(d)
    self._step = step
    return self

  def _extra_checkpoint_variables(self) -> Dict[str, np.ndarray]:
    """Variables beyond encoder / latents / decoder that Keras would track on this model
    (sub-layers and optimizers held as attributes): {checkpoint name: array}."""
    return {}

  def _load_extra_checkpoint_variables(self, d: Dict[str, np.ndarray]):
    pass

  @staticmethod
  def _find_variable(d: Dict[str, np.ndarray], name: str):
    """exact name first, then a unique '<prefix>/name' (a model nested in another)."""
    if name in d:
      return d[name]
    hits = [n for n in d if n.endswith('/' + name)]
    if len(hits) != 1:
      raise KeyError(f'{name}: {len(hits)} matching variables in the checkpoint ({hits[:3]})')
    return d[hits[0]]

  def __str__(self):
    return (f'{self.name}(input={self.input_shape}, zdim={self.zdim}, '
            f'params={self.n_parameters}, step={self._step}, device={self.device})')


# ======================================================================================
# Beta family
# ======================================================================================
class BetaVAE(VariationalAutoencoder):
  """beta_vae.py:11-43: every KL term multiplied by beta (constant or an Interpolation of
  the training step)."""

  def __init__(self, beta: Union[float, Interpolation] = 1.0, name='BetaVAE', **kwargs):
    self._beta = beta
    super().__init__(name=name, **kwargs)

  @property
  def beta(self) -> float:
    if isinstance(self._beta, Interpolation):
      return float(self._beta(self._step))
    return float(self._beta)

  @beta.setter
  def beta(self, b):
    self._beta = b


class BetaCapacityVAE(VariationalAutoencoder):
  """beta_vae.py:132-177 (Burgess et al. 2018, eq. 8): every KL term becomes gamma * |kl - C(step)| with the
  capacity C raised from c_min to c_max over n_steps by `interpolation` (backend/interpolation.py); `step` is the
  training step, incremented before the loss is evaluated (base_networks.py:478-479).  The latent kernels emit
  |kl - C| and its sign (odin_latent_fwd's `capacity` argument), so gamma takes the place beta has in BetaVAE."""

  def __init__(self, gamma: float = 10.0, c_min: float = 0.01, c_max: float = 25.0, n_steps: int = 10000,
               interpolation: str = 'linear', name='BetaCapacityVAE', **kwargs):
    from . import interpolation as interp
    self.gamma = float(gamma)
    self.interpolation = interp.get(str(interpolation))(vmin=float(c_min), vmax=float(c_max), steps=int(n_steps))
    super().__init__(name=name, **kwargs)
    self._capacity_mode = True
    self._engines.clear()

  @property
  def beta(self) -> float:  # the weight the engine multiplies the (capacity-shifted) KL by
    return self.gamma

  @property
  def capacity(self) -> float:
    return float(self.interpolation(self._step))

  def _hyper_extra(self) -> dict:
    return dict(capacity=self.capacity)


class AnnealingVAE(BetaVAE):
  """beta_vae.py:83-107: beta = linear(vmin=1e-6, vmax=1, steps=2000)(step)."""

  def __init__(self, vmin: float = 1e-6, vmax: float = 1.0, steps: int = 2000,
               name='AnnealingVAE', **kwargs):
    super().__init__(beta=linear(vmin=vmin, vmax=vmax, steps=steps), name=name, **kwargs)


class BetaTCVAE(BetaVAE):
  """beta_vae.py:110-129: + (beta-1) * total_correlation(z, q(z|x)) (losses.py:101-157);
  the KL is ALSO scaled by beta through BetaVAE.elbo_components (:42)."""

  def __init__(self, beta: float = 1.0, name='BetaTCVAE', **kwargs):
    super().__init__(beta=beta, name=name, **kwargs)
    self._tc_mode = 'betatc'
    self._engines.clear()


# ======================================================================================
# FactorVAE
# ======================================================================================
class FactorDiscriminator:
  """factor_discriminator.py:16-235: Flatten -> [Dense(units, relu)]*n -> Dense(1) logit.
  ONE set of flat parameter / gradient / Adam buffers (and one Adam iteration counter) for the
  model; `bind(B1)` adds the launch programs of a batch size: one over B1 samples (TC term
  inside the VAE step: data-gradient only) and one over 2*B1 samples ([z ; permute_dims(z')],
  discriminator step)."""

  def __init__(self, lib, zdim: int, units: Sequence[int], activation: str, device, seed: int):
    self.lib, self.device, self.D = lib, device, zdim
    layers = [('dense', int(u), activation) for u in units] + [('dense', 1, 'linear')]
    self.layout = ParamLayout()
    self.recs, out = build_layers('disc', layers, (zdim,), self.layout)
    self.layout.pad_to(4)
    f32 = dict(dtype=torch.float32, device=device)
    n = self.layout.size
    self.params, self.grads = torch.zeros(n, **f32), torch.zeros(n, **f32)
    self.m, self.v = torch.zeros(n, **f32), torch.zeros(n, **f32)
    g = torch.Generator(device='cpu').manual_seed(seed + 7)
    for key, shp, off in self.layout.entries:
      if key[-1] == 'w':  # glorot_uniform (dense_network default, base_networks.py:968)
        lim = math.sqrt(6.0 / (shp[0] + shp[1]))
        self.params[off:off + int(np.prod(shp))] = \
            ((torch.rand(int(np.prod(shp)), generator=g) * 2 - 1) * lim).to(device)
    self.flag = torch.zeros(1, dtype=torch.int32, device=device)
    self.t = 0  # iterations of the discriminator's own Adam
    self.bound: Dict[int, 'DiscPrograms'] = {}

  @property
  def n_parameters(self):
    return sum(int(np.prod(s)) for _, s, _ in self.layout.entries)

  def bind(self, B1: int, force_dp: bool = False) -> 'DiscPrograms':
    if B1 not in self.bound:
      self.bound[B1] = DiscPrograms(self, B1, force_dp)
    return self.bound[B1]


class DiscPrograms:
  """Launch programs + activation buffers of the discriminator for one half-batch size."""

  def __init__(self, disc: FactorDiscriminator, B1: int, force_dp: bool = False):
    lib, device, zdim = disc.lib, disc.device, disc.D
    f32 = dict(dtype=torch.float32, device=device)
    self.disc, self.B1 = disc, B1
    self.params, self.grads, self.layout = disc.params, disc.grads, disc.layout
    mr = lib.odin_max_slab_rows()
    # (the range words of both programs' gradient tensors in one buffer: cleared once per iteration, reset_ranges)
    nw = len(disc.recs) * RANGE_WORDS
    # (gradient words of the two programs | their activation words, round 5: the layers' inputs keep their 22 bits
    # at any magnitude)
    self.range_words = torch.zeros(4 * nw, dtype=torch.int32, device=device)
    self.prog1 = NetProgram(lib, disc.recs, B1, device, disc.params, disc.grads, mr,
                            range_words=self.range_words[:nw], act_words=self.range_words[2 * nw:3 * nw])
    self.prog2 = NetProgram(lib, disc.recs, 2 * B1, device, disc.params, disc.grads, mr,
                            range_words=self.range_words[nw:2 * nw], act_words=self.range_words[3 * nw:],
                            direct_wgrad=True)
    self.tc = torch.zeros(1, **f32)
    self.dlogit1 = torch.zeros(B1, 1, **f32)
    self.dlogit1_value = None
    self.dz = torch.zeros(B1, zdim, **f32)
    self.zcat = torch.zeros(2 * B1, zdim, **f32)  # [z (step 1's sample) ; permute_dims(z')]
    self.zperm = self.zcat[B1:]
    self.perm = torch.zeros(B1, zdim, dtype=torch.int32, device=device)
    self.world, self.rank = 1, 0
    self.gather = False  # z' is all-gathered before permute_dims (data parallel; also at world size 1 under force_dp)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_dp):
      self.gather = True
      # permute_dims shuffles across the GLOBAL batch (vi/utils.py:262-267): all ranks draw the
      # same [B1_global, D] permutation, z' is all-gathered and every rank gathers its own rows
      self.world, self.rank = dist.get_world_size(), dist.get_rank()
      self.perm = torch.zeros(B1 * self.world, zdim, dtype=torch.int32, device=device)
      self.z2_all = torch.zeros(B1 * self.world, zdim, **f32)
    self.dlogit2 = torch.zeros(2 * B1, 1, **f32)
    self.dtc = torch.zeros(1, **f32)
    self._keep = None
    # the head Dense(K -> 1) with the mean it feeds and its backward as ONE launch per pass (include/odin_hip.h:
    # odin_disc_head_fwd_bwd) instead of forward | mean or dtc_loss | weight gradient | data gradient
    self.fused_head = False
    if len(disc.recs) >= 2 and disc.recs[-1].N == 1 and disc.recs[-1].act == 'linear':
      K = disc.recs[-1].K
      rows = int(lib.odin_disc_head_rows(2 * B1, K))
      if rows > 0 and int(lib.odin_disc_head_rows(B1, K)) > 0:
        self.fused_head = True   # (FactorVAE.fuse_discriminator = False before the first step: the separate launches)
        self.head_slab = torch.empty(rows, K + 1, **f32)
        self.head_ws = torch.zeros(8, dtype=torch.int32, device=device)   # 16 bytes per pass, zero between launches

  m = property(lambda self: self.disc.m)
  v = property(lambda self: self.disc.v)

  def reset_job(self) -> ReduceJob:
    """zero the range words (their producers fold in with atomicMax): a reduction over zero slab rows, riding on
    the iteration's last slab reduction (engine.py: VAEEngine.backward does the same for its own words)"""
    rw = self.range_words
    return ReduceJob(rw.data_ptr(), rw.data_ptr(), rw.numel(), 0, rw.numel(), 0)


H_DALPHA = 10  # discriminator Adam block {alpha_t, beta1, beta2, eps, grad_scale} in the hyper buffer


class FactorVAE(AnnealingVAE):
  """factor_vae.py:99-293.  Each iteration splits the batch in halves (x1, x2):
    step 1 (VAE params, fit's Adam): loss = -mean(llk - beta_t*kl) + tc_coef*mean(D(z)),
            the gradient flows through D into z (total_correlation, factor_discriminator.py:169-198);
    step 2 (discriminator params, Adam(1e-5, .5, .9)): dtc_loss of D(z) and
            D(permute_dims(z')) with z' = encode(x2) (factor_discriminator.py:200-235).
  On one GPU the whole iteration (both steps, both optimisers) is replayed as ONE HIP graph
  when `use_graph` / `fit(compile_graph=True)`."""

  def __init__(self, discriminator_units: Sequence[int] = (1000,) * 5, discriminator_optim=None,
               activation: str = 'relu', batchnorm: bool = False, tc_coef: float = 7.0,
               maximize_tc: bool = False, name='FactorVAE', **kwargs):
    if batchnorm:
      raise NotImplementedError('batchnorm=True is outside the HIP path (reference default False)')
    super().__init__(name=name, **kwargs)
    self.tc_coef = float(tc_coef) * (-1.0 if maximize_tc else 1.0)
    self.disc_units, self.disc_act = tuple(discriminator_units), activation
    self.disc_lr, self.disc_b1, self.disc_b2 = 1e-5, 0.5, 0.9
    if discriminator_optim is not None:  # dict(learning_rate=, beta_1=, beta_2=) accepted
      self.disc_lr = float(discriminator_optim.get('learning_rate', self.disc_lr))
      self.disc_b1 = float(discriminator_optim.get('beta_1', self.disc_b1))
      self.disc_b2 = float(discriminator_optim.get('beta_2', self.disc_b2))
    self._is_pretraining = False
    self._disc_state: Optional[FactorDiscriminator] = None
    self._fgraphs: Dict[tuple, Any] = {}

  @property
  def is_pretraining(self):
    return self._is_pretraining

  def pretrain(self):
    self._is_pretraining = True
    return self

  def finetune(self):
    self._is_pretraining = False
    return self

  @property
  def discriminator(self) -> FactorDiscriminator:
    if self._disc_state is None:
      eng = self._engine(1)
      self._disc_state = FactorDiscriminator(eng.lib, self.zdim, self.disc_units, self.disc_act,
                                             self.device, self.seed)
      from .dist import broadcast_parameters
      broadcast_parameters(self._disc_state.params, src=0)
    return self._disc_state

  def _disc_names(self):
    """[(checkpoint name, flat offset, shape)] of the discriminator's variables: Keras names of
    `dense_network` inside the `FactorDiscriminator` sub-layer (factor_discriminator.py:60-95)."""
    D = self.discriminator
    return [(f"discriminator/dense_{key[1]}/{'kernel' if key[-1] == 'w' else 'bias'}", off, shp)
            for key, shp, off in D.layout.entries]

  def _extra_checkpoint_variables(self) -> Dict[str, np.ndarray]:
    """The reference keeps `self.discriminator` (a Keras layer) and `self.disc_optimizer` as
    attributes of the model (factor_vae.py:137-160), so `save_weights` tracks both: the
    discriminator's kernels / biases, its Adam's iteration count and moment slots.  A restored
    FactorVAE continues with the SAME discriminator and bias correction."""
    D = self.discriminator
    W = {}
    for name, off, shp in self._disc_names():
      n = int(np.prod(shp))
      W[name] = D.params[off:off + n].view(shp).detach().cpu().numpy()
      W[f'disc_optimizer/{name}/m'] = D.m[off:off + n].view(shp).detach().cpu().numpy()
      W[f'disc_optimizer/{name}/v'] = D.v[off:off + n].view(shp).detach().cpu().numpy()
    W['disc_optimizer/iter'] = np.asarray(D.t, np.int64)
    return W

  def _load_extra_checkpoint_variables(self, d: Dict[str, np.ndarray]):
    names = self._disc_names()
    have = [any(n == nm or n.endswith('/' + nm) for n in d) for nm, _, _ in names]
    if not any(have):
      if self._is_pretraining:
        return  # a checkpoint of the pretraining phase may predate the discriminator
      raise KeyError('the checkpoint holds no discriminator variables (discriminator/dense_*/kernel): '
                     'a FactorVAE restored from it would continue with a re-initialised discriminator; '
                     'call pretrain() first if that is intended')
    D = self.discriminator
    as_t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device)
    for name, off, shp in names:
      n = int(np.prod(shp))
      a = self._find_variable(d, name)
      if tuple(a.shape) != tuple(shp):
        raise ValueError(f'{name}: checkpoint shape {a.shape} != {tuple(shp)}')
      D.params[off:off + n].view(shp).copy_(as_t(a))
      for slot, buf in (('m', D.m), ('v', D.v)):
        buf[off:off + n].view(shp).copy_(as_t(self._find_variable(d, f'disc_optimizer/{name}/{slot}')))
    D.t = int(self._find_variable(d, 'disc_optimizer/iter'))

  def _discriminator(self, B1: int) -> DiscPrograms:
    dp = self.discriminator.bind(B1, bool(getattr(self, 'force_dp', False)))
    eng = self._engine(B1)
    if eng.z.data_ptr() != dp.zcat.data_ptr():
      # step 1's cached sample IS the first half of the discriminator step's input: the engine
      # writes z there directly (no per-iteration copy); must happen before any graph capture
      assert not getattr(eng, '_graphs', None), 'bind the discriminator before capturing graphs'
      eng.z = dp.zcat[:B1]
    return dp

  # -- the two steps, returning device scalars ------------------------------------------
  def _set_hyper(self, eng, disc, lr, use_tc, training, when_skip_update: int = 0):
    h_extra = None
    if training and not self._is_pretraining:
      t = disc.disc.t + 1
      a = self.disc_lr * math.sqrt(1 - self.disc_b2 ** t) / (1 - self.disc_b1 ** t)
      # (grad_scale 1 / world: the discriminator's bucket is SUMMED over the ranks, its loss is a mean
      # over the global batch)
      h_extra = (a, self.disc_b1, self.disc_b2, 1e-7, 1.0 / eng.world_size)
    eng.set_hyper(lr=lr, beta=self.beta, tc_coef=self.tc_coef if use_tc else 0.0, extra=h_extra,
                  skip_enable=self._step >= int(when_skip_update or 0))

  def _program(self, eng, eng2, disc, x1, x2, eps, eps2, perm, training, use_tc, pol,
               aggregate_gradients):
    """The launch program of one FactorVAE iteration: [('k', fn) kernels | ('c', fn) collective]
    (engine.VAEEngine.step_program).  On one GPU there is no 'c' entry: one graph per iteration.  Data
    parallel: A | all-reduce (VAE bucket) | B | all-gather (z' for the global permute_dims) | C |
    all-reduce (discriminator bucket) | D."""
    lib, B1 = eng.lib, disc.B1
    dp = eng.is_dp
    P = []
    # A/B switch (bench.py --no-fused-disc): the head's launches and permute_dims as round 5 issued them
    fuse = bool(getattr(self, 'fuse_discriminator', True))
    fused_head = disc.fused_head and fuse
    # the discriminator's update with its last small gradient pieces formed inside the Adam launch (odin_adam_step_fold:
    # the first layer's thin-K weight gradient, the sum of the head's slab rows, the range-word reset) -- one GPU only: a
    # data-parallel step all-reduces the finished gradient buffer first
    r0, rh = disc.prog2.recs[0], disc.prog2.recs[-1]
    fold_adam = bool(fused_head and training and not dp and r0.kind == 'dense' and r0.K <= 32 and
                     ((r0.K + 1) * r0.N) % 4 == 0 and r0.w_off % 4 == 0 and r0.b_off == r0.w_off + r0.K * r0.N and
                     rh.w_off % 4 == 0 and rh.b_off == rh.w_off + rh.K)

    # The discriminator's pass over z (TC estimate + its gradient wrt z: twelve small launches, ~96 us at batch 128, none
    # of which fills the chip) depends on z alone and is needed again only by the ENCODER's backward pass: `overlap_disc`
    # runs it on a side stream beside the decoder's forward pass and fused tail.  Measured (round 6, same-call A/B,
    # `bench.py --overlap-disc`): 0.755 ms per iteration against 0.735 on one stream -- the decoder's launches hold
    # every CU with 8 waves and ~100 KB of LDS, the forked branch of the captured graph waits for them and then delays the
    # join -- so it is OFF by default, like the engine's other overlap options.
    side = getattr(eng, 'side_stream', None)
    overlap = bool(getattr(self, 'overlap_disc', False)) and use_tc and training and side is not None

    def disc_pass(st):
      P1 = disc.prog1
      if fused_head:
        n = len(P1.recs)
        h = P1.forward(eng.z, st, upto=n - 1)
        lib.odin_disc_head_fwd_bwd(h.data_ptr(), P1.w(n - 1).data_ptr(), P1.b(n - 1).data_ptr(), P1.outs[n - 1].data_ptr(),
                                   0, disc.dlogit1.data_ptr(), None, disc.tc.data_ptr(), ACT[P1.recs[n - 2].act],
                                   P1.gouts[n - 2].data_ptr(), P1.word(n - 2), None, None, disc.head_ws.data_ptr(), B1,
                                   P1.recs[n - 1].K, st)
        P1.backward(eng.z, P1.gouts[n - 2], st, dx_out=disc.dz, data_only=True, last=n - 2)
        return
      lg = P1.forward(eng.z, st)
      lib.odin_mean(lg.data_ptr(), B1, disc.tc.data_ptr(), st)
      P1.backward(eng.z, disc.dlogit1, st, dx_out=disc.dz, data_only=True)

    def step1():  # ELBO with the discriminator's TC estimate, backward
      st = eng.stream()
      if overlap:
        cur = torch.cuda.current_stream(self.device)

        def fork_disc():
          ev = torch.cuda.Event()
          ev.record(cur)
          side.wait_event(ev)
          disc_pass(side.cuda_stream)

        eng.forward(x1, eps, finalize=False, after_latent=fork_disc)
        cur.wait_stream(side)   # (dz and the TC mean are complete before the backward pass / the finalisation)
      else:
        eng.forward(x1, eps, finalize=False)
      extra = None
      # the ELBO finalisation (llk[B], loss, mean terms: nothing in the backward pass reads them) rides in the first
      # launch of the VAE optimiser's update when one follows (engine.adam: _fin_pending), as in the plain VAE step
      ride = training
      if use_tc:
        if not overlap:
          disc_pass(st)
        if ride:
          eng._fin_pending = (eng._llk_part_used.data_ptr(), eng.n_part, disc.tc.data_ptr())
        else:
          eng.finalize(tc_ptr=disc.tc.data_ptr())
        extra = disc.dz
        if not training or self._is_pretraining:  # (no discriminator step behind this one to clear the words)
          lib.odin_range_reset(disc.range_words.data_ptr(), disc.range_words.numel() // RANGE_WORDS, st)
      elif ride:
        eng._fin_pending = (eng._llk_part_used.data_ptr(), eng.n_part, None)
      else:
        eng.finalize()
      if training:
        # the gradient norm's stage-1 sums ride in the slab reduction when the update follows at once with no per-tensor
        # policy in between (engine.VAEEngine.step_program does the same for the plain step)
        eng._fuse_norm_now = bool(fuse and not dp and not aggregate_gradients and pol[1] is None and pol[2] is None and
                                  pol[3] is None and (pol[0] is not None or pol[4]) and eng.fuse_norm)
        try:
          eng.backward(extra_dz=extra)
        finally:
          eng._fuse_norm_now = False

    P.append(('k', step1))
    if training and dp:
      P.append(('c', eng.allreduce))
    if training and not aggregate_gradients:
      P.append(('k', lambda: eng._update(pol)))
    # ---- step 2: discriminator (skipped while pretraining, factor_vae.py:279) ----
    if not self._is_pretraining:
      def encode2():
        st = eng.stream()
        eng2.run_encoder(x2, eps2)  # z' with the encoder as step 1 left it
        if perm is None and not disc.gather and fuse:   # (the permutation and permute_dims(z') in one launch)
          lib.odin_random_permute_dims(disc.perm.data_ptr(), eng2.z.data_ptr(), disc.zperm.data_ptr(), B1, self.zdim,
                                       self.seed + 11, eng.hp(N_HYPER), st)
        elif perm is None:
          lib.odin_random_perm(disc.perm.data_ptr(), B1 * disc.world, self.zdim, self.seed + 11,
                               eng.hp(N_HYPER), st)

      P.append(('k', encode2))
      if disc.gather:
        P.append(('c', lambda: eng._comm().all_gather(disc.z2_all.view(-1), eng2.z.view(-1))))

      def disc_step():
        st = eng.stream()
        if disc.gather:
          src, prm = disc.z2_all, disc.perm[disc.rank * B1:(disc.rank + 1) * B1]
        else:
          src, prm = eng2.z, disc.perm
        if perm is not None or disc.gather or not fuse:
          lib.odin_permute_dims(src.data_ptr(), prm.data_ptr(), disc.zperm.data_ptr(), B1,
                                self.zdim, st)
        P2 = disc.prog2
        if fused_head:
          n = len(P2.recs)
          r = P2.recs[n - 1]
          h = P2.forward(disc.zcat, st, upto=n - 1)
          hrows = C.c_int(0)
          lib.odin_disc_head_fwd_bwd(h.data_ptr(), P2.w(n - 1).data_ptr(), P2.b(n - 1).data_ptr(), P2.outs[n - 1].data_ptr(),
                                     1, None, disc.dlogit2.data_ptr(), disc.dtc.data_ptr(), ACT[P2.recs[n - 2].act],
                                     P2.gouts[n - 2].data_ptr() if training else None, P2.word(n - 2),
                                     disc.head_slab.data_ptr(), C.byref(hrows), disc.head_ws[4:].data_ptr(), 2 * B1, r.K, st)
          if training and fold_adam:
            jobs = P2.backward(disc.zcat, P2.gouts[n - 2], st, last=n - 2, first=1)
            if jobs:   # (none with the 1000-unit stack: its hidden layers write their gradients themselves)
              arr = (ReduceJob * len(jobs))(*jobs)
              disc._keep = arr
              lib.odin_slab_reduce(arr, len(jobs), st)
            return
          if training:
            jobs = P2.backward(disc.zcat, P2.gouts[n - 2], st, last=n - 2)
            jobs.append(ReduceJob(disc.head_slab.data_ptr(), disc.grads[r.w_off:].data_ptr(), r.K + 1, hrows.value, r.K + 1, 0))
        else:
          lg2 = P2.forward(disc.zcat, st)
          lib.odin_dtc_loss_fwd_bwd(lg2.data_ptr(), lg2[B1:].data_ptr(), disc.dtc.data_ptr(),
                                    disc.dlogit2.data_ptr(), disc.dlogit2[B1:].data_ptr(), B1, st)
          if training:
            jobs = P2.backward(disc.zcat, disc.dlogit2, st)
        if training:
          jobs.append(disc.reset_job())
          arr = (ReduceJob * len(jobs))(*jobs)
          disc._keep = arr
          lib.odin_slab_reduce(arr, len(jobs), st)

      P.append(('k', disc_step))
      if training:
        if dp:
          P.append(('c', lambda: eng._comm().all_reduce(disc.grads)))
        def disc_update():
          if fold_adam:
            P2 = disc.prog2
            fo = _lib.AdamFold(disc.zcat.data_ptr(), P2.gouts[0].data_ptr(), 2 * B1, r0.K, r0.N, r0.w_off,
                               disc.head_slab.data_ptr(), disc.head_slab.shape[0], rh.K + 1, rh.K + 1, rh.w_off,
                               disc.range_words.data_ptr(), disc.range_words.numel())
            disc._fold_keep = fo
            lib.odin_adam_step_fold(disc.params.data_ptr(), disc.grads.data_ptr(), disc.m.data_ptr(), disc.v.data_ptr(),
                                    disc.params.numel(), eng.hp(H_DALPHA), C.byref(fo), eng.stream())
            return
          lib.odin_adam_step_flat(disc.params.data_ptr(), disc.grads.data_ptr(), disc.m.data_ptr(), disc.v.data_ptr(),
                                  disc.params.numel(), eng.hp(H_DALPHA), None, 0.0, None, eng.stream())

        P.append(('k', disc_update))
    if training and aggregate_gradients:
      P.append(('k', lambda: eng._update(pol)))
    return P

  def _iteration(self, *args):
    """eager launch of one iteration on the current stream"""
    for _, fn in self._program(*args):
      fn()

  def optimize(self, inputs, training: bool = True, optimizer=None, learning_rate=1e-4,
               clipnorm=None, clipvalue=None, global_clipnorm=None, skip_update_threshold=None,
               when_skip_update: int = 0, nan_gradients_policy: str = 'skip',
               allow_none_gradients=False, aggregate_gradients=False, track_gradients=False,
               eps=None, eps2=None, perm=None, use_graph: bool = False, snapshot: bool = True, **kwargs):
    """Networks.optimize over FactorVAE.train_steps (factor_vae.py:239-287): the gradient
    policies apply to the VAE step's gradients (the discriminator step has its own optimiser
    and, as in the reference's default call, no clipping is configured for it separately --
    the same clip arguments are NOT applied to it here; state otherwise if you need them)."""
    x = _as_tensor(inputs, self.device)
    assert x.shape[0] % 2 == 0, 'FactorVAE splits the batch in two halves'
    if self.n_samples != 1:
      raise NotImplementedError('FactorVAE on the HIP path draws one posterior sample per input '
                                '(sample_shape=()); the batch-tiling of VariationalAutoencoder.optimize '
                                'is not wired through the two-step iteration')
    B1 = x.shape[0] // 2
    eng, disc = self._engine(B1), self._discriminator(B1)
    eng2 = self._engine_x2(B1)
    disc.dtc = eng.out8[4:5]   # (beside out4: one device-to-device copy snapshots the iteration's scalars)
    if training:
      self._step += 1
    eng.step_count = self._step
    eng2.hyper = eng.hyper  # one hyper buffer (RNG step, beta) for both halves
    use_tc = not (self._is_pretraining and training)
    # (Round 6, tried and dropped: two device rows for the iteration's scalars, one captured graph per row, the row of
    # iteration t + 1 copied on a side stream while iteration t runs and the main stream waiting on its event only --
    # to take the 80-byte copy and the gaps around it, 13 us, off the critical path.  Same-call A/B: 0.684 ms against
    # 0.649 with the plain stream-ordered copy; the cross-stream waits cost more than the copy.)
    self._set_hyper(eng, disc, self._lr(learning_rate), use_tc, training, when_skip_update)
    val = self.tc_coef / (B1 * eng.world_size)
    if disc.dlogit1_value != val:
      disc.dlogit1.fill_(val)
      disc.dlogit1_value = val
    if perm is not None:
      disc.perm.copy_(torch.as_tensor(perm, dtype=torch.int32, device=self.device))
    pol = (global_clipnorm, clipnorm, clipvalue, skip_update_threshold,
           nan_gradients_policy != 'ignore')
    te = None if eps is None else _as_tensor(eps, self.device)
    te2 = None if eps2 is None else _as_tensor(eps2, self.device)
    graphable = use_graph and self.device.type == 'cuda'
    if graphable:
      self._graph_iteration(eng, eng2, disc, x, te, te2, perm is not None, training, use_tc, pol,
                            aggregate_gradients)
    else:
      x1, x2 = x[:B1], x[B1:]
      self._iteration(eng, eng2, disc, x1, x2, te, te2, perm, training, use_tc, pol,
                      aggregate_gradients)
    if training and not self._is_pretraining:
      disc.disc.t += 1
    # (snapshot=False: the returned scalars are views of the buffer the NEXT iteration overwrites -- what
    # VAEEngine.train_step returns; saves the per-iteration device-to-device copy of a tight training loop)
    out = eng.out8.clone() if snapshot else eng.out8
    metrics = {f'elbo/llk_{self.observation.name}': out[1],
               f'elbo/kl_{self.latents.name}': out[2], 'elbo/tc': out[3]}
    if not self._is_pretraining:
      metrics['disc/dtc_loss'] = out[4]
    if training and track_gradients:
      for k, g in eng.grad_views().items():
        metrics['_grad/elbo/' + self.variable_name(k)] = g.clone()
    return out[0], metrics

  def _graph_iteration(self, eng, eng2, disc, x, eps, eps2, explicit_perm, training, use_tc, pol,
                       aggregate_gradients):
    """HIP graphs per (batch size, configuration): both steps, both Adams -- ONE graph on one GPU, the
    kernel segments between the collectives under data parallelism (dist.SegmentedGraph)."""
    from .dist import SegmentedGraph
    B1 = disc.B1
    key = (B1, pol, eps is not None, eps2 is not None, explicit_perm, training, use_tc,
           aggregate_gradients, self._is_pretraining, eng.analytic, eng.free_bits,
           bool(getattr(self, 'fuse_discriminator', True)))
    if key not in self._fgraphs:
      xs = self.input_buffer(x.shape[0])
      if x.data_ptr() != xs.data_ptr():
        xs.copy_(x)
      if eps is not None:
        eng.eps.copy_(eps)
      if eps2 is not None:
        eng2.eps.copy_(eps2)
      sg = SegmentedGraph(self.device, self._program(
          eng, eng2, disc, xs[:B1], xs[B1:], eng.eps if eps is not None else None,
          eng2.eps if eps2 is not None else None, True if explicit_perm else None, training, use_tc,
          pol, aggregate_gradients))
      cap = torch.cuda.Stream(self.device)
      cap.wait_stream(torch.cuda.current_stream(self.device))
      D = disc.disc
      saved = [t.clone() for t in (eng.params, eng.m, eng.v, D.params, D.m, D.v, eng.flag,
                                   eng.skipped_update)]
      with torch.cuda.stream(cap):
        sg.run_eager()  # warm-up outside capture; must not count
        for t, sv in zip((eng.params, eng.m, eng.v, D.params, D.m, D.v, eng.flag,
                          eng.skipped_update), saved):
          t.copy_(sv)
      torch.cuda.current_stream(self.device).wait_stream(cap)
      sg.capture(cap)
      self._fgraphs[key] = (sg, xs)
    sg, xs = self._fgraphs[key]
    if x.data_ptr() != xs.data_ptr():
      xs.copy_(x, non_blocking=True)
    if eps is not None:
      eng.eps.copy_(eps, non_blocking=True)
    if eps2 is not None:
      eng2.eps.copy_(eps2, non_blocking=True)
    sg.replay()

  def input_buffer(self, batch_size: int) -> torch.Tensor:
    """Static [B, H, W, C] tensor every captured iteration graph of this batch size reads: a data pipeline that
    writes the next batch straight into it (and passes it to optimize) saves the per-iteration device-to-device copy."""
    if not hasattr(self, '_xs'):
      self._xs = {}
    B = int(batch_size)
    if B not in self._xs:
      self._xs[B] = torch.empty((B,) + tuple(self.input_shape), dtype=torch.float32, device=self.device)
    return self._xs[B]

  def _engine_x2(self, B1: int) -> VAEEngine:
    key = -B1
    if key not in self._engines:
      self._engine(B1)
      self._engines[key] = VAEEngine(self.encoder.layers, self.decoder.layers, self.input_shape,
                                     self.zdim, B1, self.device,
                                     observation=self.observation.posterior,
                                     analytic=self.analytic, free_bits=self.free_bits,
                                     lib=self._lib, params=self._params,
                                     seed=self.seed + 1000003 + self._rank(),
                                     range_words=self._engines[B1].range_words)
    return self._engines[key]

  def total_correlation(self, qz_x, training=None):
    z = _as_tensor(qz_x.tensor() if isinstance(qz_x, MVNDiagPosterior) else qz_x, self.device)
    disc = self._discriminator(z.shape[0])
    lg = disc.prog1.forward(z, self._engine(z.shape[0]).stream())
    return self.tc_coef * lg.mean()

  def dtc_loss(self, qz_x, qz_xprime=None, training=None, perm=None):
    z = _as_tensor(qz_x.tensor() if isinstance(qz_x, MVNDiagPosterior) else qz_x, self.device)
    zp = z if qz_xprime is None else _as_tensor(
        qz_xprime.tensor() if isinstance(qz_xprime, MVNDiagPosterior) else qz_xprime, self.device)
    B1 = z.shape[0]
    disc, eng = self._discriminator(B1), self._engine(B1)
    lib, st = eng.lib, eng.stream()
    if perm is None:
      lib.odin_random_perm(disc.perm.data_ptr(), B1, self.zdim, self.seed + 11, None, st)
    else:
      disc.perm.copy_(torch.as_tensor(perm, dtype=torch.int32, device=self.device))
    zp = zp.clone()  # (zp may alias the buffer permute_dims writes)
    disc.zcat[:B1].copy_(z)
    lib.odin_permute_dims(zp.data_ptr(), disc.perm.data_ptr(), disc.zperm.data_ptr(), B1,
                          self.zdim, st)
    lg2 = disc.prog2.forward(disc.zcat, st)
    lib.odin_dtc_loss_fwd_bwd(lg2.data_ptr(), lg2[B1:].data_ptr(), disc.dtc.data_ptr(),
                              disc.dlogit2.data_ptr(), disc.dlogit2[B1:].data_ptr(), B1, st)
    return disc.dtc[0].clone()


def get_vae(name: str):
  """odin/bay/vi/autoencoder/__init__.py:28"""
  table = {c.__name__.lower(): c for c in (VariationalAutoencoder, BetaVAE, AnnealingVAE,
                                           BetaTCVAE, FactorVAE, BetaCapacityVAE)}
  table['vae'] = VariationalAutoencoder
  key = str(name).lower().replace('_', '')
  if key not in table:
    raise ValueError(f'Cannot find VAE {name!r}; the HIP path offers {sorted(table)}')
  return table[key]
