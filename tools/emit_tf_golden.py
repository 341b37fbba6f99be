#!/usr/bin/env python3
"""Emit golden vectors from the REFERENCE itself (trungnt13/odin-ai on TensorFlow 2.5 / TFP 0.13).

This script is NOT run by this repository (no TensorFlow here or on the GPU box): a maintainer with the reference
installed (conda env of its `odin.yml`, Python 3.7) runs it once,

    python tools/emit_tf_golden.py --out tests/golden --configs dsprites_beta4 shapes3d mnist_dense

and commits the `tests/golden/tf_*.npz` it writes; `tests/test_tf_golden.py` then holds the numpy oracle (CPU suite)
and the HIP path (GPU suite) to the reference's own numbers -- the step that turns `parity: unpinned`
into pinned.  Nothing here is copied from the reference: it only CALLS its public API
(odin.networks.get_networks, odin.bay.get_vae, VariationalAutoencoder.elbo_components / elbo,
variational_autoencoder.py:515-542, _base.py:151-194).

Per config one file with: every Keras variable by name (`encoder0/kernel:0` ...), the batch x, and what one
VAEStep (variational_autoencoder.py:117-126) produces on it: loc, scale, the sample z the reference drew (so that
eps = (z - loc) / scale can be fed to the other implementations), decoder output, llk[B], kl[B], loss and the
gradient of the loss with respect to every variable.
"""
import argparse
import os

import numpy as np

CONFIGS = {
    # name: (vae name for odin.bay.get_vae, dataset name for get_networks, input shape, batch, vae kwargs)
    'dsprites_beta4': ('betavae', 'dsprites', (64, 64, 1), 8, dict(beta=4.0)),
    'dsprites_analytic_fb': ('vae', 'dsprites', (64, 64, 1), 5, dict(analytic=True, free_bits=0.5)),
    'shapes3d': ('vae', 'shapes3d', (64, 64, 3), 6, dict()),
    'shapes3d_betatc': ('betatcvae', 'shapes3d', (64, 64, 3), 8, dict(beta=4.0)),
    'mnist_dense': ('vae', None, (28, 28, 1), 16, dict()),   # the class defaults (variational_autoencoder.py:181-185)
    'mnist_conv': ('vae', 'mnist', (28, 28, 1), 6, dict()),
}


def emit(name, out_dir, seed=7):
  import tensorflow as tf
  from odin.bay import get_vae
  from odin.networks import get_networks
  vae_name, ds, shape, B, kw = CONFIGS[name]
  tf.random.set_seed(seed)
  cls = get_vae(vae_name)
  if ds is None:
    from odin.bay.random_variable import RVconf
    vae = cls(observation=RVconf(shape, 'bernoulli', projection=True, name='image'), **kw)
  else:
    nets = get_networks(ds, is_semi_supervised=False, is_hierarchical=False)
    vae = cls(**nets, **kw)
  vae.build((None,) + shape)
  rng = np.random.default_rng(seed)
  if ds in (None, 'mnist'):
    x = (rng.random((B,) + shape) < 0.13).astype(np.float32)
  else:
    x = np.clip(rng.random((B,) + shape), 1e-6, 1 - 1e-6).astype(np.float32)
  with tf.GradientTape() as tape:
    llk, kl = vae.elbo_components(x, training=True)
    elbo = vae.elbo(llk, kl)
    loss = -tf.reduce_mean(elbo)
  px, qz = vae.last_outputs
  grads = tape.gradient(loss, vae.trainable_variables)
  out = dict(x=x, loss=np.asarray(loss), elbo=np.asarray(elbo),
             loc=np.asarray(qz.mean()), scale=np.asarray(qz.stddev()), z=np.asarray(tf.convert_to_tensor(qz)),
             recon=np.asarray(px.mean()), step=np.asarray(int(vae.step)),
             tf_version=np.asarray(tf.__version__))
  for k, v in llk.items():
    out['llk/' + k] = np.asarray(v)
  for k, v in kl.items():
    out['kl/' + k] = np.asarray(v)
  for v, g in zip(vae.trainable_variables, grads):
    out['var/' + v.name] = v.numpy()
    out['grad/' + v.name] = np.zeros_like(v.numpy()) if g is None else np.asarray(g)
  path = os.path.join(out_dir, f'tf_{name}.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, 'loss', float(loss))


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--out', default='tests/golden')
  ap.add_argument('--configs', nargs='*', default=sorted(CONFIGS))
  a = ap.parse_args()
  for c in a.configs:
    emit(c, a.out)
