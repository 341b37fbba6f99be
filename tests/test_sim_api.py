"""Model API (VariationalAutoencoder & friends) on the CPU simulator vs the oracle."""
import numpy as np
import pytest
import torch

from odin_ai_amd.networks import RVconf, SequentialNetwork, get_networks
from odin_ai_amd.vae import (AnnealingVAE, BetaTCVAE, BetaVAE, FactorVAE, VariationalAutoencoder,
                             get_vae)
from oracle import vae_oracle as vo
from tests.simutil import sim_lib


@pytest.fixture(scope='module')
def L():
  return sim_lib()


def tiny_nets(C=1, zdim=4, hw=8):
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',),
         ('dense', 24, 'linear')]
  dec = [('dense', 32, 'linear'), ('reshape', (2, 2, 8)), ('deconv', 16, 4, 2, 'elu'),
         ('deconv', 8, 4, 2, 'elu'), ('conv', C, 1, 1, 'linear')]
  return dict(encoder=SequentialNetwork(enc, 'Encoder', (hw, hw, C)),
              decoder=SequentialNetwork(dec, 'Decoder', (zdim,)),
              observation=RVconf((hw, hw, C), 'bernoulli', projection=False, name='image'),
              latents=RVconf((zdim,), 'mvndiag', projection=True, name='latents'))


def oracle_params(vae):
  return {k: v.detach().cpu().numpy().astype(np.float64) for k, v in vae.trainable_variables.items()}


def test_api_call_elbo_and_optimize(L):
  nets = tiny_nets()
  vae = BetaVAE(beta=4.0, device='cpu', lib=L, **nets)
  B = 6
  rng = np.random.default_rng(0)
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, 4)).astype(np.float32)
  px, qz = vae(x, eps=eps)
  assert px.mean().shape == (B, 8, 8, 1) and qz.mean().shape == (B, 4)
  assert qz.event_shape == (4,) and px.batch_shape == (B,)
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), 4, beta=4.0)
  P = oracle_params(vae)
  f = model.forward(P, x.astype(np.float64), eps.astype(np.float64))
  np.testing.assert_allclose(qz.mean().numpy(), f['loc'], atol=1e-5)
  np.testing.assert_allclose(qz.stddev().numpy(), f['scale'], atol=1e-5)
  np.testing.assert_allclose(px.mean().numpy(), f['recon'], atol=1e-5)
  llk, kl = vae.elbo_components(x, eps=eps)
  assert set(llk) == {'llk_image'} and set(kl) == {'kl_latents'}
  np.testing.assert_allclose(llk['llk_image'].numpy(), f['llk'], rtol=1e-5)
  np.testing.assert_allclose(kl['kl_latents'].numpy(), f['kl'], rtol=1e-4, atol=1e-4)
  np.testing.assert_allclose(vae.elbo(llk, kl).numpy(), f['elbo'], rtol=1e-5)
  # the reference's KL_divergence closure on the posterior object
  np.testing.assert_allclose(qz.KL_divergence(analytic=False).numpy() * 4.0, f['kl'], rtol=1e-4,
                             atol=1e-4)
  assert qz.KL_divergence(analytic=True, keepdims=True).shape == (1, B)
  step = next(iter(vae.train_steps(x, training=True, eps=eps)))
  loss, metrics = step()
  assert abs(float(loss) - f['loss']) < 1e-4 * abs(f['loss'])
  # one optimisation step: loss returned is the pre-update loss; step counter increments
  l0, m = vae.optimize(x, eps=eps, learning_rate=1e-3)
  assert vae.step == 1 and abs(float(l0) - f['loss']) < 1e-4 * abs(f['loss'])
  assert set(m) == {'llk_image', 'kl_latents'}
  l1, _ = vae.optimize(x, eps=eps, learning_rate=1e-3)
  assert float(l1) < float(l0)


def test_fit_save_load_and_errors(L, tmp_path):
  nets = tiny_nets()
  vae = VariationalAutoencoder(device='cpu', lib=L, path=str(tmp_path / 'w.pkl'), **nets)
  x = (np.random.default_rng(1).random((24, 8, 8, 1)) < 0.3).astype(np.float32)
  vae.fit(x, max_iter=6, batch_size=8, learning_rate=1e-3, compile_graph=False)
  assert vae.step == 6 and len(vae.history) >= 1
  vae.save_weights()
  vae2 = VariationalAutoencoder(device='cpu', lib=L, path=str(tmp_path / 'w.pkl'), **tiny_nets())
  vae2.load_weights()
  assert vae2.step == 6
  for k, v in vae.trainable_variables.items():
    assert torch.equal(v, vae2.trainable_variables[k])
  with pytest.raises(ValueError):
    get_networks('no_such_dataset')
  with pytest.raises(ValueError):
    VariationalAutoencoder(device='cpu', lib=L, **dict(tiny_nets(), encoder='not a network'))
  with pytest.raises(RuntimeError):
    vae.fit(x, optimizer='sgd', max_iter=1)
  with pytest.raises(ValueError):
    get_vae('nope')
  assert get_vae('betavae') is BetaVAE


def test_annealing_and_betatc(L):
  a = AnnealingVAE(device='cpu', lib=L, **tiny_nets())
  a._step = 1000
  assert abs(a.beta - 0.5000005) < 1e-9  # linear(1e-6, 1, 2000) at step 1000 (SURVEY KAT)
  nets = tiny_nets()
  tcv = BetaTCVAE(beta=3.0, device='cpu', lib=L, **nets)
  B = 6
  rng = np.random.default_rng(2)
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, 4)).astype(np.float32)
  llk, kl = tcv.elbo_components(x, eps=eps)
  assert set(kl) == {'kl_latents', 'tc_latents'}
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), 4, beta=3.0,
                       tc_beta=3.0)
  f = model.forward(oracle_params(tcv), x.astype(np.float64), eps.astype(np.float64))
  assert abs(float(kl['tc_latents']) - f['tc']) < 1e-4 * max(1.0, abs(f['tc']))
  np.testing.assert_allclose(tcv.elbo(llk, kl).numpy(), f['elbo'], rtol=1e-5)


def test_factor_vae_two_steps_match_oracle(L):
  nets = tiny_nets()
  B1, D = 4, 4
  fv = FactorVAE(discriminator_units=(16, 16), tc_coef=7.0, device='cpu', lib=L, **nets)
  rng = np.random.default_rng(3)
  x = np.clip(rng.random((2 * B1, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  P = oracle_params(fv)
  disc = fv._discriminator(B1)
  DP = {(k[1], k[2]): v.detach().numpy().astype(np.float64)
        for k, v in disc.layout.views(disc.params).items()}
  loss, metrics = fv.optimize(x, training=False, eps=eps, eps2=eps2, perm=perm)
  # oracle: step 1
  beta = vo.interp_linear(0)  # step 0 (training=False): beta = 1e-6 + ...
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), D, beta=beta)
  x64 = x.astype(np.float64)
  f = model.forward(P, x64[:B1], eps.astype(np.float64))
  dl = vo.disc_layers((16, 16))
  lg, _ = vo.disc_forward(dl, DP, f['z'])
  tc = 7.0 * lg.mean()
  assert abs(float(loss) - (f['loss'] + tc)) < 1e-4 * max(1.0, abs(f['loss']))
  assert abs(float(metrics['elbo/tc']) - tc) < 1e-4 * max(1.0, abs(tc))
  # step 2: dtc loss with an explicit permutation
  f2 = model.forward(P, x64[B1:], eps2.astype(np.float64))
  zp = vo.permute_dims(f2['z'], perm.astype(np.int64))
  l1, _ = vo.disc_forward(dl, DP, f['z'])
  l2, _ = vo.disc_forward(dl, DP, zp)
  assert abs(float(metrics['disc/dtc_loss']) - vo.dtc_loss(l1, l2)) < 1e-5
  # a real training iteration moves both parameter sets
  p_before, d_before = fv._params.clone(), disc.params.clone()
  fv.optimize(x, training=True, learning_rate=1e-3, eps=eps, eps2=eps2, perm=perm)
  assert fv.step == 1 and not torch.equal(p_before, fv._params)
  assert not torch.equal(d_before, disc.params)


@pytest.mark.parametrize('units', [(16, 16), (40,)])
def test_factor_vae_iteration_gradients_and_both_adams(L, units):
  """the step-1 dz through D, the discriminator gradients and the
  Adam(1e-5, .5, .9) update against the oracle (simulator build of the kernels)."""
  from tests.factor_util import check_factor_vae_iteration
  nets = tiny_nets()
  B1, D = 4, 4
  fv = FactorVAE(discriminator_units=units, tc_coef=7.0, device='cpu', lib=L, **nets)
  rng = np.random.default_rng(11)
  x = np.clip(rng.random((2 * B1, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  rep = check_factor_vae_iteration(fv, nets, units, B1, x, eps, eps2, perm, clip=100.0)
  assert fv.step == 1000
