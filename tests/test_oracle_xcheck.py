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
