"""Model API (VariationalAutoencoder & friends) vs the oracle, on both backends of the `bk` fixture
(tests/conftest.py): the CPU fiber simulator build of the kernel sources (`-m "not gpu"`) and the
gfx950 build on an MI355X through the C ABI (`-m gpu`): marginal_log_prob (f2), sample_shape,
every Networks.optimize gradient policy, track_gradients, checkpoints, FactorVAE iterations."""
import numpy as np
import pytest
import torch

from odin_ai_amd.networks import RVconf, SequentialNetwork, get_networks
from odin_ai_amd.vae import (AnnealingVAE, BetaCapacityVAE, BetaTCVAE, BetaVAE, FactorVAE,
                             VariationalAutoencoder, get_vae)
from oracle import vae_oracle as vo


@pytest.fixture(scope='module')
def L(bk):
  return bk.L


@pytest.fixture(scope='module')
def DEV(bk):
  return bk.dev


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
  return {k: v.detach().cpu().numpy(force=True).astype(np.float64) for k, v in vae.trainable_variables.items()}


def test_api_call_elbo_and_optimize(L, DEV):
  nets = tiny_nets()
  vae = BetaVAE(beta=4.0, device=DEV, lib=L, **nets)
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
  np.testing.assert_allclose(qz.mean().numpy(force=True), f['loc'], atol=1e-5)
  np.testing.assert_allclose(qz.stddev().numpy(force=True), f['scale'], atol=1e-5)
  np.testing.assert_allclose(px.mean().numpy(force=True), f['recon'], atol=1e-5)
  llk, kl = vae.elbo_components(x, eps=eps)
  assert set(llk) == {'llk_image'} and set(kl) == {'kl_latents'}
  np.testing.assert_allclose(llk['llk_image'].numpy(force=True), f['llk'], rtol=1e-5)
  np.testing.assert_allclose(kl['kl_latents'].numpy(force=True), f['kl'], rtol=1e-4, atol=1e-4)
  np.testing.assert_allclose(vae.elbo(llk, kl).numpy(force=True), f['elbo'], rtol=1e-5)
  # the reference's KL_divergence closure on the posterior object
  np.testing.assert_allclose(qz.KL_divergence(analytic=False).numpy(force=True) * 4.0, f['kl'], rtol=1e-4,
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


def test_fit_save_load_and_errors(L, DEV, tmp_path):
  nets = tiny_nets()
  vae = VariationalAutoencoder(device=DEV, lib=L, path=str(tmp_path / 'w'), **nets)
  x = (np.random.default_rng(1).random((24, 8, 8, 1)) < 0.3).astype(np.float32)
  vae.fit(x, max_iter=6, batch_size=8, learning_rate=1e-3, compile_graph=False)
  assert vae.step == 6 and len(vae.history) >= 1
  vae.save_weights()
  vae2 = VariationalAutoencoder(device=DEV, lib=L, path=str(tmp_path / 'w'), **tiny_nets())
  vae2.load_weights()
  assert vae2.step == 6
  for k, v in vae.trainable_variables.items():
    assert torch.equal(v, vae2.trainable_variables[k])
  with pytest.raises(ValueError):
    get_networks('no_such_dataset')
  with pytest.raises(ValueError):
    VariationalAutoencoder(device=DEV, lib=L, **dict(tiny_nets(), encoder='not a network'))
  with pytest.raises(RuntimeError):
    vae.fit(x, optimizer='sgd', max_iter=1)
  with pytest.raises(ValueError):
    get_vae('nope')
  assert get_vae('betavae') is BetaVAE


def test_annealing_and_betatc(L, DEV):
  a = AnnealingVAE(device=DEV, lib=L, **tiny_nets())
  a._step = 1000
  assert abs(a.beta - 0.5000005) < 1e-9  # linear(1e-6, 1, 2000) at step 1000 (SURVEY KAT)
  nets = tiny_nets()
  tcv = BetaTCVAE(beta=3.0, device=DEV, lib=L, **nets)
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
  np.testing.assert_allclose(tcv.elbo(llk, kl).numpy(force=True), f['elbo'], rtol=1e-5)


def test_factor_vae_two_steps_match_oracle(L, DEV):
  nets = tiny_nets()
  B1, D = 4, 4
  fv = FactorVAE(discriminator_units=(16, 16), tc_coef=7.0, device=DEV, lib=L, **nets)
  rng = np.random.default_rng(3)
  x = np.clip(rng.random((2 * B1, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  P = oracle_params(fv)
  disc = fv._discriminator(B1)
  DP = {(k[1], k[2]): v.detach().numpy(force=True).astype(np.float64)
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


@pytest.mark.parametrize('units', [(16, 16), (40,), (64, 64)])   # (64: the head fused with its loss, odin_disc_head_fwd_bwd)
def test_factor_vae_iteration_gradients_and_both_adams(L, DEV, units):
  """the step-1 dz through D, the discriminator gradients and the
  Adam(1e-5, .5, .9) update against the oracle (simulator build of the kernels)."""
  from tests.factor_util import check_factor_vae_iteration
  nets = tiny_nets()
  B1, D = 4, 4
  fv = FactorVAE(discriminator_units=units, tc_coef=7.0, device=DEV, lib=L, **nets)
  rng = np.random.default_rng(11)
  x = np.clip(rng.random((2 * B1, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  rep = check_factor_vae_iteration(fv, nets, units, B1, x, eps, eps2, perm, clip=100.0)
  assert fv.step == 1000


def test_beta_capacity_vae_matches_oracle(L, DEV):
  """BetaCapacityVAE (beta_vae.py:132-177): kl <- gamma * |kl - C(step)| with C = linear(c_min, c_max, n_steps)
  evaluated at the ALREADY incremented step: elbo_components, the loss and every gradient of one optimize()
  against the oracle, on both sides of the capacity (kl > C early, kl < C late)."""
  from odin_ai_amd.interpolation import linear
  nets = tiny_nets()
  B, D = 6, 4
  rng = np.random.default_rng(5)
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, D)).astype(np.float32)
  for start_step, c_max in ((0, 0.02), (7, 400.0), (3, 3.0)):  # kl > C everywhere, kl < C everywhere, mixed
    vae = BetaCapacityVAE(gamma=10.0, c_min=0.01, c_max=c_max, n_steps=10, analytic=True, device=DEV, lib=L,
                          **nets)
    vae._step = start_step
    P = oracle_params(vae)
    sched = linear(vmin=0.01, vmax=c_max, steps=10)
    # forward API at the current step
    c_now = float(sched(vae.step))
    assert abs(vae.capacity - c_now) < 1e-12
    m0 = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), D, beta=10.0, capacity=c_now,
                      analytic=True)
    f0 = m0.forward(P, x.astype(np.float64), eps.astype(np.float64))
    llk, kl = vae.elbo_components(x, eps=eps)
    assert tuple(kl['kl_latents'].shape) == (1, B)  # analytic KL keeps its leading axis (helpers.py:370-371)
    np.testing.assert_allclose(kl['kl_latents'].numpy(force=True)[0], f0['kl'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(llk['llk_image'].numpy(force=True), f0['llk'], rtol=1e-5)
    # one training step: the schedule sees step + 1 (base_networks.py:478-479)
    c_next = float(sched(start_step + 1))
    m1 = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), D, beta=10.0, capacity=c_next,
                      analytic=True)
    f1 = m1.forward(P, x.astype(np.float64), eps.astype(np.float64))
    G, _ = m1.backward(P, x.astype(np.float64), eps.astype(np.float64), f1)
    side = np.sign(f1['kl_raw'] - c_next)
    if start_step == 0: assert (side > 0).all(), side   # both branches of |.| are exercised
    if start_step == 7: assert (side < 0).all(), side
    loss, metrics = vae.optimize(x, eps=eps, learning_rate=1e-3, track_gradients=True)
    assert vae.step == start_step + 1
    assert abs(float(loss) - f1['loss']) <= 1e-4 * max(1.0, abs(f1['loss']))
    assert abs(float(metrics['kl_latents']) - f1['kl'].mean()) <= 1e-4 * max(1.0, abs(f1['kl'].mean()))
    eng = vae._engine(B)
    gv = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
    for k, g in G.items():
      err = np.abs(gv[k] - g).max() / max(1e-30, np.abs(g).max())
      assert err <= 1e-4, (k, err)


def test_save_best_llk_callback(L, DEV, tmp_path):
  """examples/vae/utils.py:446-467: mean log p(x|z) over the validation batches, `valid/llk`, checkpoint on
  improvement only -- through fit(on_valid_end=...)."""
  import os
  from odin_ai_amd.callbacks import Callback
  nets = tiny_nets()
  path = str(tmp_path / 'best')
  vae = BetaVAE(beta=1.0, device=DEV, lib=L, path=path, **nets)
  rng = np.random.default_rng(6)
  x = (rng.random((48, 8, 8, 1)) < 0.3).astype(np.float32).clip(1e-6, 1 - 1e-6)
  xv = x[:16]
  # the value itself: mean over samples of px.log_prob(x) with the model's own posterior sample
  said = []
  llk0 = Callback.save_best_llk(vae, xv, batch_size=8, log=said.append)
  assert said[-1].startswith('best llk') and os.path.exists(path + '.index')
  t0 = os.path.getmtime(path + '.index')
  px, _ = vae(xv[:8])
  assert np.isfinite(llk0) and abs(llk0) > 1.0 and px.log_prob(torch.as_tensor(xv[:8]).to(DEV)).shape == (8,)
  # an untrained copy scores no better than the best -> no new checkpoint
  Callback._best[id(vae)] = llk0 + 1e9
  Callback.save_best_llk(vae, xv, batch_size=8, log=said.append)
  assert said[-1].startswith('worse llk') and os.path.getmtime(path + '.index') == t0
  Callback._best[id(vae)] = llk0
  # inside fit: training improves the validation likelihood and the callback checkpoints it
  seen = []
  vae.fit(x, valid=xv, valid_freq=4, max_iter=9, batch_size=16, learning_rate=5e-3, compile_graph=False,
          on_valid_end=lambda: seen.append(Callback.save_best_llk(vae, xv, batch_size=8, log=None)))
  assert len(seen) >= 3 and max(seen) > llk0 and Callback._best[id(vae)] == max(seen)
  # the checkpoint on disk is the best model: reloading it reproduces the weights of that moment
  w = {k: v.clone() for k, v in vae.trainable_variables.items()}
  vae2 = BetaVAE(beta=1.0, device=DEV, lib=L, path=path, **nets).load_weights(raise_notfound=True)
  if seen[-1] == max(seen):
    for k, v in vae2.trainable_variables.items():
      assert torch.equal(v, w[k]), k


def test_factor_vae_torch_autograd_cross_check(L, DEV):
  """The float64 torch-autograd form of the FactorVAE check that the GPU suite runs at BASELINE config 3's size
  (tests/factor_util.check_factor_vae_full_size), here on the simulator build at toy size."""
  from tests.factor_util import check_factor_vae_full_size
  nets = tiny_nets()
  B1, D, units = 4, 4, (16, 16)
  fv = FactorVAE(discriminator_units=units, tc_coef=7.0, device=DEV, lib=L, **nets)
  rng = np.random.default_rng(12)
  x = np.clip(rng.random((2 * B1, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  check_factor_vae_full_size(fv, nets, units, B1, x, eps, eps2, perm, clip=100.0)


@pytest.mark.parametrize('kw', [dict(clipnorm=0.05), dict(clipvalue=0.002),
                                dict(global_clipnorm=0.5, clipvalue=0.01),
                                dict(clipnorm=0.2, global_clipnorm=0.3),
                                dict(skip_update_threshold=1e-3),
                                dict(skip_update_threshold=1e-3, when_skip_update=5),
                                dict(skip_update_threshold=1e9)])
def test_optimize_gradient_policies_match_oracle(L, DEV, kw):
  """Every clipping / skipping argument of Networks.optimize (base_networks.py:549-596) changes
  the Adam update exactly as the oracle's restatement says -- none is silently ignored."""
  nets = tiny_nets()
  vae = BetaVAE(beta=2.0, device=DEV, lib=L, **nets)
  B = 6
  rng = np.random.default_rng(4)
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, 4)).astype(np.float32)
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), 4, beta=2.0)
  P = oracle_params(vae)
  f = model.forward(P, x.astype(np.float64), eps.astype(np.float64))
  G, _ = model.backward(P, x.astype(np.float64), eps.astype(np.float64), f)
  keys = [k for k, _ in model.param_shapes()]
  okw = {k: v for k, v in kw.items() if k != 'when_skip_update'}
  gl, skipped = vo.gradient_policies([G[k] for k in keys],
                                     skip_enabled=1 >= kw.get('when_skip_update', 0), **okw)
  lr = 1e-3
  vae.optimize(x, eps=eps, learning_rate=lr, **kw)
  assert vae.skipped_update == int(skipped)
  pv = {k: v.numpy(force=True) for k, v in vae.trainable_variables.items()}
  for k, g in zip(keys, gl):
    want, _, _ = vo.adam_keras(P[k], g, 0.0, 0.0, 1, lr)
    well = np.abs(g) > 1e-3 * max(np.abs(g).max(), 1e-30)
    d = np.abs(pv[k] - want)
    assert d[well].max() <= 2e-5 if well.any() else True, (k, d.max())
    if skipped:
      assert np.array_equal(pv[k], P[k].astype(np.float32)), k  # Adam on zeros from zero state
  with pytest.raises(ValueError):
    vae.optimize(x, nan_gradients_policy='nope')


def test_track_gradients_and_checkpoint_names(L, DEV, tmp_path):
  nets = get_networks('dsprites')
  names = ['encoder0', 'encoder3', 'encoder_proj', 'decoder_proj', 'decoder1', 'decoder6']
  from odin_ai_amd.networks import layer_names
  en, dn = layer_names(nets['encoder'], 'encoder'), layer_names(nets['decoder'], 'decoder')
  assert set(names) <= set(en.values()) | set(dn.values())
  vae = VariationalAutoencoder(device=DEV, lib=L, path=str(tmp_path / 'ck'), **tiny_nets())
  x = (np.random.default_rng(1).random((4, 8, 8, 1)) < 0.3).astype(np.float32)
  _, m = vae.optimize(x, track_gradients=True)
  gk = [k for k in m if k.startswith('_grad/')]
  assert len(gk) == len(vae.trainable_variables) and '_grad/latents/kernel' in gk
  vae.save_weights(save_format='npz')
  d = np.load(str(tmp_path / 'ck.npz'), allow_pickle=False)
  assert 'latents/kernel' in d.files and int(d['__step__']) == 1
  # the reference's format: a TensorFlow checkpoint, variables under their Keras names
  from odin_ai_amd import tf_checkpoint
  vae.save_weights(str(tmp_path / 'tfck' / 'model'))
  ck = tf_checkpoint.load_checkpoint(str(tmp_path / 'tfck' / 'model'))
  assert int(ck['Step']) == 1 and ck['latents/kernel'].shape == (24, 8)
  v2 = VariationalAutoencoder(device=DEV, lib=L, **tiny_nets()).load_weights(str(tmp_path / 'tfck' / 'model'))
  assert v2.step == 1
  for k, v in vae.trainable_variables.items():
    assert torch.equal(v, v2.trainable_variables[k])
  # a checkpoint whose object graph prefixes the names (as a Keras model nested in another would)
  W = {'vae/' + k: v for k, v in ck.items()}
  tf_checkpoint.save_checkpoint(str(tmp_path / 'nested'), W)
  v3 = VariationalAutoencoder(device=DEV, lib=L, **tiny_nets()).load_weights(str(tmp_path / 'nested'))
  assert torch.equal(v3.trainable_variables[('lat', 'w')], vae.trainable_variables[('lat', 'w')])
  # fit(logdir=...) writes the scalars the reference's Trainer logs
  vae.fit(x, max_iter=4, batch_size=4, compile_graph=False, logdir=str(tmp_path / 'tb'),
          nan_check_interval=2)
  ev = tf_checkpoint.read_scalar_events(vae._events.path)
  assert {t for _, t, _ in ev} >= {'train/loss', 'train/llk_image', 'train/kl_latents'}


def test_nan_gradients_policy_skip_vs_ignore(L, DEV):
  """Networks.optimize (base_networks.py:519-547): with a non-finite gradient 'skip' leaves parameters and
  optimiser state untouched and raises the flag; 'ignore' applies the update whatever the gradients hold -- also
  when global_clipnorm routes the update through the fused norm + Adam launches."""
  x = (np.random.default_rng(8).random((4, 8, 8, 1)) < 0.3).astype(np.float32)
  xbad = x.copy()
  xbad[1, 2, 3, 0] = float('nan')
  for gclip in (None, 10.0):
    vae = VariationalAutoencoder(device=DEV, lib=L, **tiny_nets())
    P0 = {k: v.clone() for k, v in vae.trainable_variables.items()}
    vae.optimize(xbad, nan_gradients_policy='skip', global_clipnorm=gclip)
    eng = vae._engine(4)
    assert int(eng.flag.item()) == 1
    for k, v in vae.trainable_variables.items():
      assert torch.equal(v, P0[k]), k
    eng.flag.zero_()
    vae.optimize(xbad, nan_gradients_policy='ignore', global_clipnorm=gclip)
    assert int(eng.flag.item()) == 0
    assert any(not torch.isfinite(v).all() for v in vae.trainable_variables.values())


def test_fit_validation_cadence_and_callbacks(L, DEV, tmp_path):
  """Trainer.fit (training/trainer.py:607-709): validation at the first iteration, then every `valid_freq`
  steps, and once more when training ends; on_valid_end is called every time (with or without a validation
  set), on_batch_end after every step; valid/ scalars land in the event file; evaluation does not train."""
  from odin_ai_amd import tf_checkpoint
  rng = np.random.default_rng(3)
  xtr = (rng.random((16, 8, 8, 1)) < 0.3).astype(np.float32)
  xva = (rng.random((8, 8, 8, 1)) < 0.3).astype(np.float32)
  vae = VariationalAutoencoder(device=DEV, lib=L, **tiny_nets())
  calls = dict(batch=0, valid=0)
  vae.fit(xtr, valid=xva, valid_freq=2, max_iter=4, batch_size=4, compile_graph=False, logdir=str(tmp_path / 'tb'),
          logging_interval=0, nan_check_interval=1,
          on_batch_end=lambda: calls.__setitem__('batch', calls['batch'] + 1),
          on_valid_end=lambda: calls.__setitem__('valid', calls['valid'] + 1))
  assert calls == dict(batch=4, valid=4)          # it = 1, steps 2 and 4, end of training
  assert vae.step == 4 and np.isfinite(vae.last_valid_loss) and 'llk_image' in vae.last_valid_metrics
  assert [s for s, _ in vae.valid_history] == [1, 2, 4, 4]
  tags = {t for _, t, _ in tf_checkpoint.read_scalar_events(vae._events.path)}
  assert {'train/loss', 'valid/loss', 'valid/llk_image', 'valid/kl_latents'} <= tags
  # the validation loss is the mean over the validation batches of the training=False step
  P0 = {k: v.clone() for k, v in vae.trainable_variables.items()}
  ref = np.mean([float(next(iter(vae.train_steps(xva[i:i + 4], training=False)))()[0]) for i in (0, 4)])
  for k, v in vae.trainable_variables.items():
    assert torch.equal(v, P0[k])
  # (same data, same parameters; the posterior sample differs from call to call: compare loosely)
  assert abs(vae.last_valid_loss - ref) < 0.2 * abs(ref)
  # valid_interval > 0 takes precedence over valid_freq: nothing but the first and the final call within 1000 s
  calls['valid'] = 0
  vae.fit(xtr, valid=xva, valid_freq=1, valid_interval=1000.0, max_iter=3, batch_size=4, compile_graph=False,
          on_valid_end=lambda: calls.__setitem__('valid', calls['valid'] + 1))
  assert calls['valid'] == 2


def test_marginal_log_prob_matches_oracle(L, DEV):
  """variational_autoencoder.py:396-513: one encoder pass, n posterior samples, n*B decodes,
  log-mean-exp over the samples -- against the oracle with the same noise."""
  nets = tiny_nets()
  vae = VariationalAutoencoder(device=DEV, lib=L, **nets)
  rng = np.random.default_rng(9)
  N, n, D = 10, 7, 4
  x = np.clip(rng.random((N, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((n, N, D)).astype(np.float32)
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), D)
  llk_ref, (lq_ref, lp_ref) = vo.marginal_log_prob(model, oracle_params(vae), x, eps)
  llk, kl = vae.marginal_log_prob(x, n_mcmc=n, reduce=None, batch_size=4, eps=eps)
  assert set(llk) == {'image'} and set(kl) == {'latents'}
  np.testing.assert_allclose(llk['image'].numpy(force=True), llk_ref, rtol=2e-5)
  np.testing.assert_allclose(kl['latents'][0].numpy(force=True), lq_ref, rtol=2e-5, atol=2e-5)
  np.testing.assert_allclose(kl['latents'][1].numpy(force=True), lp_ref, rtol=2e-5, atol=2e-5)
  llk_m, _ = vae.marginal_log_prob(x, n_mcmc=n, batch_size=4, eps=eps)  # reduce = mean
  assert abs(float(llk_m['image']) - llk_ref.mean()) < 1e-4 * abs(llk_ref.mean())
  # without explicit noise: finite, and more samples tighten the bound on average
  a, _ = vae.marginal_log_prob(x, n_mcmc=2, batch_size=5)
  b, _ = vae.marginal_log_prob(x, n_mcmc=50, batch_size=5)
  assert np.isfinite(float(a['image'])) and float(b['image']) >= float(a['image']) - 1.0


def test_reverse_kl_and_sample_shape(L, DEV):
  nets = tiny_nets()
  rng = np.random.default_rng(10)
  B, D = 5, 4
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, D)).astype(np.float32)
  with pytest.raises(TypeError):
    BetaVAE(beta=2.0, reverse=False, device=DEV, lib=L, **nets)
  vae = BetaVAE(beta=2.0, reverse=False, analytic=True, device=DEV, lib=L, **nets)
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), D, beta=2.0,
                       analytic=True, reverse=False)
  P = oracle_params(vae)
  f = model.forward(P, x.astype(np.float64), eps.astype(np.float64))
  G, _ = model.backward(P, x.astype(np.float64), eps.astype(np.float64), f)
  llk, kl = vae.elbo_components(x, eps=eps)
  assert kl['kl_latents'].shape == (1, B)
  np.testing.assert_allclose(kl['kl_latents'][0].numpy(force=True), f['kl'], rtol=1e-4, atol=1e-4)
  px, qz = vae.last_outputs
  np.testing.assert_allclose(qz.KL_divergence(analytic=True, reverse=False).numpy(force=True) * 2.0, f['kl'],
                             rtol=1e-4, atol=1e-4)
  with pytest.raises(TypeError):
    qz.KL_divergence(analytic=False, reverse=False)
  _, m = vae.optimize(x, eps=eps, track_gradients=True, learning_rate=1e-3)
  for k, g in G.items():
    got = m['_grad/' + vae.variable_name(k)].numpy(force=True)
    assert np.abs(got - g).max() <= 1e-4 * np.abs(g).max(), k
  # sample_shape=(3,): [3, B] components, loss = overall mean
  n = 3
  v3 = VariationalAutoencoder(sample_shape=(n,), device=DEV, lib=L, **tiny_nets())
  v3._engine(1).load_params(P)
  eps3 = rng.standard_normal((n, B, D)).astype(np.float32)
  llk3, kl3 = v3.elbo_components(x, eps=eps3)
  assert llk3['llk_image'].shape == (n, B) and kl3['kl_latents'].shape == (n, B)
  m1 = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, 1), D)
  for k in range(n):
    fk = m1.forward(P, x.astype(np.float64), eps3[k].astype(np.float64))
    np.testing.assert_allclose(llk3['llk_image'][k].numpy(force=True), fk['llk'], rtol=1e-5)
    np.testing.assert_allclose(kl3['kl_latents'][k].numpy(force=True), fk['kl'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('C', [1, 3])
def test_mixture_quantized_logistic_model(L, DEV, C):
  """f1: _parse_distribution('mixqlogistic') (image_networks.py:72-85) -> MixtureQuantizedLogistic head
  (quantized.py:206-381): decoder head sizes, call / elbo / mean against the oracle, and the model trains."""
  assert get_networks('mnist', distribution='mixqlogistic')['decoder'].layers[-1][1] == 30
  assert get_networks('celeba', distribution='mixqlogistic')['decoder'].layers[-1][1] == 100
  with pytest.raises(ValueError):
    get_networks('mnist', distribution='mixqlogistic', n_components=5)
  nets = tiny_nets(C=C)
  n_maps = 10 * vo.mixql_n_out(C)
  nets['decoder'] = SequentialNetwork(nets['decoder'].layers[:-1] + [('conv', n_maps, 1, 1, 'linear')],
                                      'Decoder', (4,))
  nets['observation'] = RVconf((8, 8, C), 'mixqlogistic', projection=False, name='image',
                               kwargs=dict(n_components=10))
  vae = BetaVAE(beta=2.0, device=DEV, lib=L, **nets)
  B = 6
  rng = np.random.default_rng(5)
  x = np.clip(rng.random((B, 8, 8, C)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, 4)).astype(np.float32)
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, (8, 8, C), 4, beta=2.0,
                       observation='mixqlogistic')
  P = oracle_params(vae)
  f = model.forward(P, x.astype(np.float64), eps.astype(np.float64))
  px, qz = vae(x, eps=eps)
  assert px.event_shape == (8, 8, C) and px.mean().shape == (B, 8, 8, C)
  np.testing.assert_allclose(px.mean().numpy(force=True), f['recon'], atol=2e-5)
  import torch as _t
  np.testing.assert_allclose(px.log_prob(_t.as_tensor(x, device=DEV)).numpy(force=True), f['llk'], rtol=2e-5)
  llk, kl = vae.elbo_components(x, eps=eps)
  np.testing.assert_allclose(llk['llk_image'].numpy(force=True), f['llk'], rtol=2e-5)
  l0, _ = vae.optimize(x, eps=eps, learning_rate=1e-3)
  assert abs(float(l0) - f['loss']) < 1e-4 * abs(f['loss'])
  for _ in range(3):
    l1, _ = vae.optimize(x, eps=eps, learning_rate=1e-3)
  assert float(l1) < float(l0)
  assert px.sample().shape == (B, 8, 8, C)
