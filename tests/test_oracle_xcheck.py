"""numpy float64 oracle vs the independent torch-autograd restatement (CPU only)."""
import numpy as np
import pytest

from oracle import vae_oracle as vo
from oracle.torch_ref import TorchVAE


def _small_spec(kind):
  if kind == 'dsprites':
    return vo.dsprites_spec(1), 'bernoulli'
  if kind == 'shapes3d':
    return vo.dsprites_spec(3), 'bernoulli'
  if kind == 'celeba_qlogistic':
    return vo.celeba_spec(45, 6), 'qlogistic'
  if kind == 'celeba_gauss':
    return vo.celeba_spec(45, 6), 'gaussian_softplus1'
  if kind == 'mixql_1ch':   # mnist-style single channel: 10 x (1 + 1 + 1) parameter maps
    return vo.dsprites_spec(1, n_out_params=30), 'mixqlogistic'
  if kind == 'mixql_3ch':   # RGB: 10 x (1 + 3 + 3 + 3) maps with the channel chain
    e, d, s_, z = vo.dsprites_spec(3)
    return (e, d[:-1] + [('conv', 100, 1, 1, 'linear')], s_, z), 'mixqlogistic'
  if kind == 'mnist_conv':
    return vo.mnist_conv_spec(), 'bernoulli'
  if kind == 'mnist_dense':
    return vo.mnist_dense_spec(), 'bernoulli'
  raise ValueError(kind)


CASES = [
    ('dsprites', dict(beta=4.0)),
    ('dsprites', dict(beta=1.0, analytic=True)),
    ('dsprites', dict(beta=2.0, free_bits=0.5)),
    ('dsprites', dict(beta=3.0, analytic=True, reverse=False)),
    ('shapes3d', dict(beta=1.0, tc_beta=4.0)),
    ('celeba_gauss', dict(beta=4.0, tc_beta=4.0)),
    ('celeba_qlogistic', dict(beta=2.0)),
    ('mixql_1ch', dict(beta=2.0)),
    ('mixql_3ch', dict(beta=1.0)),
    ('mnist_conv', dict()),
    ('mnist_dense', dict(analytic=True)),
]


@pytest.mark.parametrize('kind,kw', CASES)
def test_oracle_matches_torch_autograd(kind, kw):
  (enc, dec, in_shape, zdim), obs = _small_spec(kind)
  B = 3
  rng = np.random.default_rng(7)
  x = np.clip(rng.random((B,) + in_shape), 1e-6, 1 - 1e-6)
  eps = rng.standard_normal((B, zdim))
  m = vo.OracleVAE(enc, dec, in_shape, zdim, observation=obs, **kw)
  P = m.init_params(seed=3)
  f = m.forward(P, x, eps)
  G, _ = m.backward(P, x, eps, f)
  tm = TorchVAE(enc, dec, in_shape, zdim, observation=obs, **kw)
  tf_, TG = tm.loss_and_grads(P, x, eps)
  assert abs(f['loss'] - float(tf_['loss'])) <= 1e-9 * max(1.0, abs(f['loss']))
  for k in ('loc', 'scale', 'z', 'llk', 'kl', 'h_d'):
    np.testing.assert_allclose(f[k], tf_[k], rtol=1e-9, atol=1e-9)
  for k in G:
    scale = max(1e-12, np.abs(TG[k]).max())
    assert np.abs(G[k] - TG[k]).max() <= 1e-9 * max(scale, 1.0), k


def test_tc_grad_matches_autograd():
  import torch
  from oracle.torch_ref import t_total_correlation
  rng = np.random.default_rng(0)
  B, D = 16, 5
  z, loc = rng.standard_normal((B, D)), rng.standard_normal((B, D))
  sc = 0.3 + rng.random((B, D))
  tz, tl, ts = (torch.tensor(a, requires_grad=True) for a in (z, loc, sc))
  tc = t_total_correlation(tz, tl, ts)
  tc.backward()
  assert abs(float(tc) - vo.total_correlation(z, loc, sc)) < 1e-12
  gz, gl, gs = vo.total_correlation_bwd(z, loc, sc)
  np.testing.assert_allclose(gz, tz.grad.numpy(), atol=1e-12)
  np.testing.assert_allclose(gl, tl.grad.numpy(), atol=1e-12)
  np.testing.assert_allclose(gs, ts.grad.numpy(), atol=1e-12)


def test_torch_factor_iteration_matches_numpy_oracle():
  """oracle/torch_ref.TorchFactorTrainer (bench.py's cpu_baseline of the FactorVAE workload) against
  vo.factor_vae_iteration: both optimisers' parameters after one iteration, float64."""
  import torch
  from oracle.torch_ref import TorchFactorTrainer, TorchVAE
  enc = [('flatten',), ('dense', 12, 'relu')]
  dec = [('dense', 12, 'relu'), ('dense', 16, 'linear'), ('reshape', (4, 4, 1))]
  in_shape, zdim, B = (4, 4, 1), 3, 8
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation='bernoulli', beta=1.0)
  P = model.init_params(seed=1)
  dl = [('dense', 6, 'relu'), ('dense', 1, 'linear')]
  rng = np.random.default_rng(0)
  DP = {('disc', 0, 'w'): rng.standard_normal((3, 6)) * 0.5, ('disc', 0, 'b'): np.zeros(6),
        ('disc', 1, 'w'): rng.standard_normal((6, 1)) * 0.5, ('disc', 1, 'b'): np.zeros(1)}
  x = (rng.random((B, 4, 4, 1)) < 0.4).astype(np.float64)
  e1, e2 = rng.standard_normal((4, 3)), rng.standard_normal((4, 3))
  perm = np.stack([rng.permutation(4) for _ in range(3)], 1)
  M = {k: np.zeros_like(v) for k, v in P.items()}
  V = {k: np.zeros_like(v) for k, v in P.items()}
  DPo = {(k[1], k[2]): v for k, v in DP.items()}
  DM = {k: np.zeros_like(v) for k, v in DPo.items()}
  DV = {k: np.zeros_like(v) for k, v in DPo.items()}
  o = vo.factor_vae_iteration(model, P, M, V, 1, dl, DPo, DM, DV, 1, x, e1, e2, perm, 1e-3, tc_coef=7.0)
  tm = TorchVAE(enc, dec, in_shape, zdim, observation='bernoulli', beta=1.0, dtype=torch.float64)
  tr = TorchFactorTrainer(tm, P, dl, DP, lr=1e-3, tc_coef=7.0)
  loss = tr.step(torch.tensor(x), torch.tensor(e1), torch.tensor(e2), torch.tensor(perm))
  assert abs(loss - o['loss']) < 1e-10
  assert max(np.abs(tr.vae.T[k].detach().numpy() - o['P'][k]).max() for k in P) < 1e-12
  assert max(np.abs(tr.D[k].detach().numpy() - o['DP'][(k[1], k[2])]).max() for k in DP) < 1e-12


# ---------------------------------------------------------------------------------------------------------
# A THIRD restatement that the builder did not write: torch.distributions (the PyTorch project's own
# Normal / Independent / Bernoulli / kl_divergence / Logistic-by-transform implementations) and torch.nn.functional's
# convolutions, against the numpy oracle's distribution arithmetic.  It pins nothing against TensorFlow -- the oracle
# stays "parity unpinned" -- but a formula slip shared by oracle/vae_oracle.py and oracle/torch_ref.py (both written
# for this repository) would show here.
# ---------------------------------------------------------------------------------------------------------
def test_distribution_arithmetic_matches_torch_distributions():
  import torch
  import torch.distributions as td
  rng = np.random.default_rng(11)
  B, D = 7, 10
  loc, raw, eps = rng.standard_normal((B, D)), rng.standard_normal((B, D)), rng.standard_normal((B, D))
  p = np.concatenate([loc, raw], 1)
  loc_o, sc_o = vo.mvn_diag_params(p, D)
  T = lambda a: torch.tensor(a, dtype=torch.float64)
  q = td.Independent(td.Normal(T(loc), torch.nn.functional.softplus(T(raw))), 1)   # continuous.py:478-483
  prior = td.Independent(td.Normal(torch.zeros(D, dtype=torch.float64), torch.ones(D, dtype=torch.float64)), 1)
  np.testing.assert_allclose(sc_o, q.base_dist.scale.numpy(), rtol=1e-12)
  z = loc_o + sc_o * eps
  # helpers.py:267-276 Monte-Carlo KL = log q(z) - log p(z); :265 analytic KL(q||p); :261-262 swapped
  np.testing.assert_allclose(vo.kl_mc(loc_o, sc_o, z), (q.log_prob(T(z)) - prior.log_prob(T(z))).numpy(), rtol=1e-10,
                             atol=1e-10)
  np.testing.assert_allclose(vo.kl_analytic(loc_o, sc_o), td.kl_divergence(q, prior).numpy(), rtol=1e-11)
  prior_b = td.Independent(td.Normal(torch.zeros(B, D, dtype=torch.float64), torch.ones(B, D, dtype=torch.float64)), 1)
  np.testing.assert_allclose(vo.kl_analytic_reverse(loc_o, sc_o), td.kl_divergence(prior_b, q).numpy(), rtol=1e-11)
  # Independent(Bernoulli(logits), 3).log_prob(x) for non-binary x in (0, 1) (discrete.py:88-95, image_networks.py:91)
  lg, x = rng.standard_normal((B, 6, 5, 3)) * 3, np.clip(rng.random((B, 6, 5, 3)), 1e-6, 1 - 1e-6)
  bern = td.Independent(td.Bernoulli(logits=T(lg), validate_args=False), 3)
  np.testing.assert_allclose(vo.bernoulli_log_prob(lg, x), bern.log_prob(T(x)).numpy(), rtol=1e-11)
  # Independent(Normal(loc, scale), 3) (image_networks.py:95-102) and the softplus1 scale (backend/maths.py:279-281)
  mu, rw = rng.standard_normal((B, 6, 5, 3)), rng.standard_normal((B, 6, 5, 3))
  sc = vo.softplus1(rw)
  np.testing.assert_allclose(sc, torch.nn.functional.softplus(T(rw) + float(np.log(np.expm1(1.0)))).numpy(), rtol=1e-12)
  gauss = td.Independent(td.Normal(T(mu), T(sc)), 3)
  np.testing.assert_allclose(vo.gaussian_log_prob(mu, sc, x), gauss.log_prob(T(x)).numpy(), rtol=1e-11)
  # free bits on the summed KL (helpers.py:278-280)
  kl = vo.kl_analytic(loc_o, sc_o)
  clamped, mask = vo.free_bits_clamp(kl, 0.8, D)
  np.testing.assert_allclose(clamped, torch.clamp(T(kl), min=0.8 * D).numpy())
  assert ((kl > 0.8 * D) == (mask == 1.0)).all()


def test_quantized_logistic_matches_a_transformed_torch_distribution():
  """QuantizedDistribution(Logistic(loc, scale), low=0, high=255) (bay/distributions/quantized.py:50-170 -> TFP): the
  probability of pixel level j is CDF(j) - CDF(j - 1) of a Logistic built from torch's Uniform through the
  sigmoid / affine transforms, with the two edge bins open-ended."""
  import torch
  import torch.distributions as td
  rng = np.random.default_rng(12)
  loc, raw = rng.standard_normal((4, 3)) * 0.5, rng.standard_normal((4, 3))
  x = rng.integers(0, 256, size=(4, 3)) / 255.0
  m, s = vo.qlogistic_params(loc, raw)
  T = lambda a: torch.tensor(a, dtype=torch.float64)
  base = td.Uniform(torch.zeros_like(T(m)), torch.ones_like(T(m)))
  logistic = td.TransformedDistribution(base, [td.SigmoidTransform().inv, td.AffineTransform(T(m), T(s))])
  j = T(np.round(x * 255.0))
  # quantized.py:107-113: the quantised variable is ceil(Logistic - 0.5) clipped to [0, 255], so
  # P(j) = cdf(j + 1/2) - cdf(j - 1/2) of the Logistic itself, P(0) = cdf(1/2), P(255) = 1 - cdf(254.5)
  hi, lo = logistic.cdf(j + 0.5), logistic.cdf(j - 0.5)
  prob = torch.where(j <= 0, hi, torch.where(j >= 255, 1.0 - lo, hi - lo))
  np.testing.assert_allclose(vo.qlogistic_log_prob_elem(loc, raw, x), torch.log(prob).numpy(), rtol=1e-8, atol=1e-10)


def test_conv_stack_matches_torch_functional_convolutions():
  """TF `SAME` Conv2D / Conv2DTranspose as explicit pads + F.conv2d and F.conv_transpose2d + crop,
  written out here independently of oracle/torch_ref.py."""
  import torch
  import torch.nn.functional as F
  rng = np.random.default_rng(13)
  for (H, W, Ci, Co, K, S) in ((9, 12, 3, 5, 4, 2), (8, 8, 4, 6, 5, 1), (7, 10, 2, 3, 5, 2), (8, 8, 6, 4, 4, 1)):
    x, w, b = rng.standard_normal((2, H, W, Ci)), rng.standard_normal((K, K, Ci, Co)), rng.standard_normal(Co)
    oh, pt, pb = vo.same_pads(H, K, S)
    ow, pl, pr = vo.same_pads(W, K, S)
    xt = F.pad(torch.tensor(x).permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xt, torch.tensor(w).permute(3, 2, 0, 1), torch.tensor(b), stride=S).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(vo.conv2d(x, w, b, S), y, rtol=1e-10, atol=1e-10)
    # the transposed layer mapping (H, W, Co) -> (H S, W S, Ci) with Keras' (kh, kw, out, in) kernel
    wt = rng.standard_normal((K, K, Ci, Co))
    xin = rng.standard_normal((2, H, W, Co))
    full = F.conv_transpose2d(torch.tensor(xin).permute(0, 3, 1, 2), torch.tensor(wt).permute(3, 2, 0, 1), stride=S)
    _, qt, _ = vo.same_pads(H * S, K, S)
    _, ql, _ = vo.same_pads(W * S, K, S)
    yt = full[:, :, qt:qt + H * S, ql:ql + W * S].permute(0, 2, 3, 1).numpy() + rng.standard_normal(0).sum()
    np.testing.assert_allclose(vo.conv2d_transpose(xin, wt, np.zeros(Ci), S), yt, rtol=1e-10, atol=1e-10)


def test_keras_adam_known_answers_by_hand():
  """tf.optimizers.Adam (base_networks.py:101): epsilon OUTSIDE the bias-corrected square root -- three steps of a
  two-element problem worked by hand in plain Python floats."""
  lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
  theta, m, v = [1.0, -2.0], [0.0, 0.0], [0.0, 0.0]
  P, M, V = np.array(theta), np.zeros(2), np.zeros(2)
  grads = [[0.5, -0.25], [0.1, 0.4], [-0.3, 0.0]]
  for t, g in enumerate(grads, start=1):
    a = lr * (1.0 - b2 ** t) ** 0.5 / (1.0 - b1 ** t)
    for i in range(2):
      m[i] = b1 * m[i] + (1 - b1) * g[i]
      v[i] = b2 * v[i] + (1 - b2) * g[i] * g[i]
      theta[i] -= a * m[i] / (v[i] ** 0.5 + eps)
    P, M, V = vo.adam_keras(P, np.array(g), M, V, t, lr)
    np.testing.assert_allclose(P, theta, rtol=1e-13)
  # first step from zero state: |delta| = lr * |g| / (|g| + eps / sqrt(1 - b2))  
  p1, _, _ = vo.adam_keras(np.zeros(1), np.array([0.02]), 0.0, 0.0, 1, lr)
  assert abs(-p1[0] - lr * 0.02 / (0.02 + eps / (1 - b2) ** 0.5)) < 1e-15
