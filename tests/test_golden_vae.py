"""Frozen float64-oracle vectors (tests/golden/vae_golden_*.npz, oracle/gen_vae_golden.py):
  * `-m "not gpu"`: the oracle re-derives every stored value from the seeds (the oracle cannot
    change silently);
  * `-m gpu`: the HIP engine is held to the FROZEN numbers, not to a live oracle call.
These fixtures freeze the oracle; they do not pin it against TensorFlow (see the oracle header)."""
import os

import numpy as np
import pytest
import torch

from oracle import gen_vae_golden as gen
from oracle import vae_oracle as vo

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
NAMES = list(gen.CASES)


def load(name):
  return np.load(os.path.join(GOLD, f'vae_golden_{name}.npz'), allow_pickle=False)


@pytest.mark.parametrize('name', NAMES)
def test_oracle_reproduces_frozen_vectors(name):
  g, o = load(name), gen.run_case(name)
  assert set(g.files) == set(o.keys())
  for k in g.files:
    a, b = np.asarray(g[k], np.float64), np.asarray(o[k], np.float64)
    assert a.shape == b.shape, k
    assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(a).max()), (name, k)


def _digest_close(got, want, tol):
  d = gen.digest(got)
  scale = max(1.0, want[2])
  assert abs(d[2] - want[2]) <= tol * scale                       # max |.|
  assert np.abs(d[3:] - want[3:]).max() <= tol * scale            # strided samples
  assert abs(d[0] - want[0]) <= tol * scale * np.sqrt(np.size(got)) * 4   # sum (random-walk bound)


@pytest.mark.gpu
@pytest.mark.parametrize('name', [n for n in NAMES if n != 'shapes3d_factor'])
def test_hip_engine_matches_frozen_vectors(name):
  from odin_ai_amd import _lib
  from odin_ai_amd.engine import VAEEngine
  g = load(name)
  spec, obs, B, kw, _ = gen.CASES[name]
  enc, dec, in_shape, zdim = spec()
  dev = torch.device('cuda:0')
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation=obs, **kw)
  P = {k: v.astype(np.float32).astype(np.float64) for k, v in model.init_params(seed=17).items()}
  for k, v in P.items():  # the seeded parameters are the frozen ones
    assert np.abs(gen.digest(v) - g['param0/' + '/'.join(map(str, k))]).max() < 1e-12
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation=obs,
                  tc='betatc' if 'tc_beta' in kw else None, lib=_lib.load())
  eng.load_params(P)
  beta = kw.get('beta', 1.0)
  eng.step_count = 1
  eng.set_hyper(lr=1e-3, beta=beta)
  eng.forward(torch.tensor(g['x'], device=dev), torch.tensor(g['eps'], device=dev))
  eng.backward()
  torch.cuda.synchronize()
  p = eng.p.cpu().numpy()
  assert np.abs(p[:, :zdim] - g['loc']).max() <= 1e-4
  assert np.abs(p[:, zdim:] - g['raw_scale']).max() <= 1e-4
  assert np.abs(eng.z.cpu().numpy() - g['z']).max() <= 1e-4
  assert np.abs(eng.llk.cpu().numpy() - g['llk']).max() <= 1e-4 * np.abs(g['llk']).max()
  assert np.abs(eng.kl.cpu().numpy() * beta - g['kl']).max() <= 1e-4 * max(1.0, np.abs(g['kl']).max())
  assert abs(eng.out4[0].item() - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
  if 'tc' in g.files:
    assert abs(eng.out4[3].item() - float(g['tc'])) <= 1e-4 * max(1.0, abs(float(g['tc'])))
  _digest_close(eng.dec.outs[-1].cpu().numpy(), g['h_d'], 1e-4)
  gtol = 1e-3 if 'tc_beta' in kw else 1e-4   # (TC conditioning: tests/test_gpu_parity.py)
  for k, v in eng.grad_views().items():
    _digest_close(v.cpu().numpy(), g['grad/' + '/'.join(map(str, k))], gtol)
  # The optimiser, in three checks of decreasing sharpness:
  #  (1) THE test of the Adam kernel: Keras-Adam in float64 on the ENGINE's own gradients must reproduce the
  #      engine's parameters to 2e-6 -- a sign error or a wrong epsilon placement cannot pass this;
  #  (2) against the frozen float64 trajectory, elements whose gradient is well conditioned
  #      (|g| > 1e-3 max|g|: Adam normalises every update to ~lr, so a tiny gradient amplifies fp32 rounding to a
  #      whole step) must agree to 1e-4, the north-star tolerance;
  #  (3) the rest is bounded by the worst case of (2)'s exclusion -- a flipped normalised update, 2 lr -- and the
  #      mean error over the samples by 0.5 % of lr.
  lr = 1e-3
  gv = {k: v.cpu().numpy().astype(np.float64) for k, v in eng.grad_views().items()}
  p0 = {k: v.cpu().numpy().astype(np.float64) for k, v in eng.param_views().items()}
  keys = list(p0)
  gn = vo.global_norm([gv[k] for k in keys])
  gs = 100.0 / max(gn, 100.0)
  eng.adam(global_clipnorm=100.0)
  for k, v in eng.param_views().items():
    pk, _, _ = vo.adam_keras(p0[k], gv[k] * gs, np.zeros_like(p0[k]), np.zeros_like(p0[k]), 1, lr)
    assert np.abs(v.cpu().numpy() - pk).max() <= 2e-6 * max(1.0, np.abs(pk).max()), ('adam-kernel', k)
    want = g['param/' + '/'.join(map(str, k))]
    gwant = g['grad/' + '/'.join(map(str, k))]
    got = gen.digest(v.cpu().numpy())
    err = np.abs(got[3:] - want[3:])
    good = np.abs(gwant[3:]) > 1e-3 * gwant[2]
    tol = 1e-3 if 'tc_beta' in kw else 1e-4
    if good.any():
      assert err[good].max() <= tol, (k, err[good].max())
    assert err.max() <= 2.0 * lr + 1e-6, (k, err.max())
    assert err.mean() <= (5e-5 if 'tc_beta' in kw else 5e-6), (k, err.mean())


@pytest.mark.gpu
def test_hip_factor_vae_matches_frozen_vectors():
  from odin_ai_amd import _lib
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import FactorVAE
  g = load('shapes3d_factor')
  dev = torch.device('cuda:0')
  nets = get_networks('shapes3d')
  fv = FactorVAE(discriminator_units=(64, 64), tc_coef=7.0, device=dev, lib=_lib.load(), **nets)
  B1 = 2
  eng, disc = fv._engine(B1), fv._discriminator(B1)
  enc, dec, in_shape, zdim = vo.dsprites_spec(3)
  model = vo.OracleVAE(enc, dec, in_shape, zdim)
  eng.load_params({k: v.astype(np.float32) for k, v in model.init_params(seed=17).items()})
  for k, v in disc.layout.views(disc.params).items():
    v.copy_(torch.tensor(g[f'disc0/{k[1]}/{k[2]}'], device=dev))
  fv._step = int(g['t']) - 1
  loss, m = fv.optimize(g['x'], learning_rate=1e-3, eps=g['eps'], eps2=g['eps2'], perm=g['perm'],
                        global_clipnorm=100.0)
  assert abs(float(loss) - float(g['loss'])) <= 1e-4 * abs(float(g['loss']))
  assert abs(float(m['elbo/tc']) - float(g['tc'])) <= 1e-4 * max(1.0, abs(float(g['tc'])))
  assert abs(float(m['disc/dtc_loss']) - float(g['dtc_loss'])) <= 1e-5
  assert np.array_equal(disc.zperm.cpu().numpy(),
                        np.take_along_axis(fv._engine_x2(B1).z.cpu().numpy(), g['perm'].astype(np.int64), 0))
  assert np.abs(disc.zperm.cpu().numpy() - g['zperm']).max() <= 1e-4
  for k, v in eng.grad_views().items():
    _digest_close(v.cpu().numpy(), g['grad/' + '/'.join(map(str, k))], 1e-4)
  for k, v in disc.layout.views(disc.grads).items():
    _digest_close(v.cpu().numpy(), g[f'dgrad/{k[1]}/{k[2]}'], 1e-4)
