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
