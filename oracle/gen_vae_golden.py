"""Generates tests/golden/vae_golden_*.npz: frozen outputs of the float64 VAE oracle
(oracle/vae_oracle.py) on seeded inputs, SURVEY.md section 8c step 3.

The VAE path of the reference cannot run here (TensorFlow absent) and its tests hold no vectors,
so these fixtures do NOT pin the oracle against the reference ("parity unpinned" stays in the
oracle's header).  What they do: freeze today's oracle, so that the oracle and the kernels can no
longer drift together -- tests/test_golden_vae.py re-derives every value from the seeds and
compares, and the `-m gpu` tests hold the HIP engine to the frozen numbers.

Per case: inputs (x, eps, perm) in full; small outputs (loc, raw scale, z, llk[B], kl[B], loss, tc)
in full; large tensors (decoder output, every gradient, post-Adam parameters) as a digest
[sum, sum of squares, max |.|] plus 64 strided samples.

    python oracle/gen_vae_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vae_oracle as vo  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def digest(a):
  a = np.asarray(a, np.float64).ravel()
  idx = np.linspace(0, a.size - 1, min(64, a.size)).astype(np.int64)
  return np.concatenate([[a.sum(), (a * a).sum(), np.abs(a).max()], a[idx]])


CASES = {
    # name: (spec, observation, B, model kwargs, binary data)
    'mnist_dense': (vo.mnist_dense_spec, 'bernoulli', 4, dict(), True),
    'dsprites_beta4': (lambda: vo.dsprites_spec(1), 'bernoulli', 2, dict(beta=4.0), False),
    'shapes3d_factor': (lambda: vo.dsprites_spec(3), 'bernoulli', 4, dict(), False),
    'celeba_betatc': (lambda: vo.celeba_spec(45, 3), 'bernoulli', 4, dict(beta=4.0, tc_beta=4.0), False),
    'celeba_qlogistic': (lambda: vo.celeba_spec(45, 6), 'qlogistic', 2, dict(beta=2.0), False),
    # f1: MixtureQuantizedLogistic head, 1 channel (30 maps), dSprites-shaped conv stack
    'dsprites_mixql': (lambda: vo.dsprites_spec(1, n_out_params=30), 'mixqlogistic', 2, dict(beta=2.0), False),
}


def run_case(name):
  spec, obs, B, kw, binary = CASES[name]
  enc, dec, in_shape, zdim = spec()
  rng = np.random.default_rng(abs(hash(name)) % 1000 if False else sum(map(ord, name)))
  if binary:
    x = (rng.random((B,) + tuple(in_shape)) < 0.13).astype(np.float64)
  else:
    x = np.clip(rng.random((B,) + tuple(in_shape)), 1e-6, 1 - 1e-6).astype(np.float32).astype(np.float64)
  out = dict(x=x.astype(np.float32))
  lr = 1e-3
  if name == 'shapes3d_factor':
    B1 = B // 2
    eps, eps2 = rng.standard_normal((B1, zdim)), rng.standard_normal((B1, zdim))
    perm = np.stack([rng.permutation(B1) for _ in range(zdim)], 1).astype(np.int32)
    t = 1000
    model = vo.OracleVAE(enc, dec, in_shape, zdim, beta=vo.interp_linear(t))
    P = model.init_params(seed=17)
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    units = (64, 64)
    dl = vo.disc_layers(units)
    drng = np.random.default_rng(23)
    DP, shp = {}, zdim
    for li, L in enumerate(dl):
      DP[(li, 'w')] = (drng.standard_normal((shp, L[1])) * np.sqrt(2.0 / shp)).astype(np.float32).astype(np.float64)
      DP[(li, 'b')] = np.zeros(L[1])
      shp = L[1]
    zero = lambda d: {k: np.zeros_like(v) for k, v in d.items()}
    r = vo.factor_vae_iteration(model, P, zero(P), zero(P), t, dl, DP, zero(DP), zero(DP), 1, x, eps,
                                eps2, perm, lr, tc_coef=7.0, global_clipnorm=100.0)
    f = r['fwd']
    out.update(eps=eps.astype(np.float32), eps2=eps2.astype(np.float32), perm=perm, t=np.int64(t),
               loss=r['loss'], tc=r['tc'], dtc_loss=r['dtc_loss'], loc=f['loc'], raw_scale=f['raw_scale'],
               z=f['z'], llk=f['llk'], kl=f['kl'], z2=r['z2'], zperm=r['zperm'])
    for k, g in r['G'].items():
      out['grad/' + '/'.join(map(str, k))] = digest(g)
    for k, g in r['DG'].items():
      out['dgrad/' + '/'.join(map(str, k))] = digest(g)
    for k, v in r['P'].items():
      out['param/' + '/'.join(map(str, k))] = digest(v)
    for k, v in r['DP'].items():
      out['dparam/' + '/'.join(map(str, k))] = digest(v)
    for k, v in DP.items():
      out['disc0/' + '/'.join(map(str, k))] = v.astype(np.float32)
  else:
    eps = rng.standard_normal((B, zdim))
    model = vo.OracleVAE(enc, dec, in_shape, zdim, observation=obs, **kw)
    P = model.init_params(seed=17)
    P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
    zero = {k: np.zeros_like(v) for k, v in P.items()}
    P2, M2, V2, f, G = vo.train_step(model, P, zero, dict(zero), 1, x, eps, lr, global_clipnorm=100.0)
    out.update(eps=eps.astype(np.float32), loss=f['loss'], loc=f['loc'], raw_scale=f['raw_scale'],
               z=f['z'], llk=f['llk'], kl=f['kl'], h_d=digest(f['h_d']))
    if 'tc' in f:
      out['tc'] = f['tc']
    Gu, _ = model.backward(P, x, eps, f)
    for k, g in Gu.items():
      out['grad/' + '/'.join(map(str, k))] = digest(g)
    for k, v in P2.items():
      out['param/' + '/'.join(map(str, k))] = digest(v)
  for k, v in P.items():
    out['param0/' + '/'.join(map(str, k))] = digest(v)
  return out


if __name__ == '__main__':
  os.makedirs(OUT, exist_ok=True)
  for name in CASES:
    o = run_case(name)
    path = os.path.join(OUT, f'vae_golden_{name}.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in o.items()})
    print(name, os.path.getsize(path), 'bytes', 'loss', float(o['loss']))
