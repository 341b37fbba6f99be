"""Consumes golden vectors emitted by the REFERENCE on TensorFlow (tools/emit_tf_golden.py, run by a maintainer who has
TF 2.5 / TFP 0.13): tests/golden/tf_<config>.npz.  None is committed yet -- this container and the GPU box have no
TensorFlow -- so these tests skip; the day a file is present the numpy oracle (CPU suite) and the HIP step (GPU
suite) are held to the reference's own loss, llk[B], kl[B], posterior parameters and gradients at 1e-4."""
import glob
import os
import re

import numpy as np
import pytest

from oracle import vae_oracle as vo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'tf_*.npz')))

SPECS = {
    'dsprites_beta4': (lambda: vo.dsprites_spec(1), dict(beta=4.0)),
    'dsprites_analytic_fb': (lambda: vo.dsprites_spec(1), dict(analytic=True, free_bits=0.5)),
    'shapes3d': (lambda: vo.dsprites_spec(3), dict()),
    'shapes3d_betatc': (lambda: vo.dsprites_spec(3), dict(beta=4.0, tc_beta=4.0)),
    'mnist_dense': (lambda: vo.mnist_dense_spec(), dict()),
    'mnist_conv': (lambda: vo.mnist_conv_spec(), dict()),
}


def keras_to_oracle_params(G, model, prefix='var/'):
  """Keras variable names -> oracle keys, in network order: the encoder's kernels / biases in the order Keras lists
  them, then the `latents` DistributionDense, then the decoder's (image_networks.py:460-513 layer names)."""
  names = [k[len(prefix):] for k in G.files if k.startswith(prefix)]
  enc = [n for n in names if n.split('/')[0].lower().startswith('encoder')]
  dec = [n for n in names if n.split('/')[0].lower().startswith('decoder') or n.split('/')[0].lower().startswith('image')]
  lat = [n for n in names if n.split('/')[0].lower().startswith('latents')]
  P = {}
  keys = [k for k, _ in model.param_shapes()]
  for net, group in (('enc', enc), ('dec', dec)):
    want = [k for k in keys if k[0] == net]
    kernels = [n for n in group if 'kernel' in n]
    biases = [n for n in group if 'bias' in n]
    assert len(kernels) == len([k for k in want if k[2] == 'w']), (net, kernels, want)
    for k, n in zip([k for k in want if k[2] == 'w'], kernels):
      P[k] = G[prefix + n].astype(np.float64)
    for k, n in zip([k for k in want if k[2] == 'b'], biases):
      P[k] = G[prefix + n].astype(np.float64)
  P[('lat', 'w')] = G[prefix + [n for n in lat if 'kernel' in n][0]].astype(np.float64)
  P[('lat', 'b')] = G[prefix + [n for n in lat if 'bias' in n][0]].astype(np.float64)
  return P


@pytest.mark.skipif(not GOLDEN, reason='no tests/golden/tf_*.npz: run tools/emit_tf_golden.py where TensorFlow 2.5 is')
@pytest.mark.parametrize('path', GOLDEN or ['-'], ids=[os.path.basename(p) for p in GOLDEN] or ['none'])
def test_oracle_matches_the_tensorflow_reference(path):
  G = np.load(path)
  name = re.sub(r'^tf_|\.npz$', '', os.path.basename(path))
  spec, kw = SPECS[name]
  enc, dec, in_shape, zdim = spec()
  model = vo.OracleVAE(enc, dec, in_shape, zdim, **kw)
  P = keras_to_oracle_params(G, model)
  x = G['x'].astype(np.float64)
  eps = (G['z'].astype(np.float64) - G['loc']) / G['scale']
  f = model.forward(P, x, eps)
  assert abs(f['loss'] - float(G['loss'])) <= 1e-4 * max(1.0, abs(float(G['loss'])))
  np.testing.assert_allclose(f['loc'], G['loc'], atol=1e-4)
  np.testing.assert_allclose(f['scale'], G['scale'], atol=1e-4)
  llk = [G[k] for k in G.files if k.startswith('llk/')][0]
  np.testing.assert_allclose(f['llk'], llk.reshape(-1), rtol=1e-4)
  grads, _ = model.backward(P, x, eps, f)
  GP = keras_to_oracle_params(G, model, prefix='grad/')
  for k, g in grads.items():
    assert np.abs(g - GP[k]).max() <= 1e-4 * max(1e-30, np.abs(GP[k]).max()), k
