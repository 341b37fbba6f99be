"""TensorFlow checkpoint / event-file containers written and read without TensorFlow."""
import os
import struct

import numpy as np
import pytest

from odin_ai_amd import tf_checkpoint as tfc


def test_crc32c_known_answers():
  # RFC 3720 B.4 test vectors for CRC-32C
  assert tfc.crc32c(b'123456789') == 0xE3069283
  assert tfc.crc32c(bytes(32)) == 0x8A9136AA
  assert tfc.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
  assert tfc.crc32c(bytes(range(32))) == 0x46DD794E
  assert tfc.unmask(tfc.mask(0x12345678)) == 0x12345678
  # the library's host routine agrees with the pure-python one, including continuation
  from odin_ai_amd import _lib
  L = _lib.Lib(_lib.DEFAULT_LIB) if os.path.exists(_lib.DEFAULT_LIB) else None
  if L is not None:
    data = bytes(np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8))
    assert tfc.crc32c(data, lib=L) == tfc.crc32c(data)
    assert tfc.crc32c(data[5000:], tfc.crc32c(data[:5000]), lib=L) == tfc.crc32c(data)


def test_bundle_round_trip_and_corruption(tmp_path):
  rng = np.random.default_rng(1)
  V = {'encoder0/kernel': rng.standard_normal((4, 4, 1, 32)).astype(np.float32),
       'encoder0/bias': np.zeros(32, np.float32),
       'latents/kernel': rng.standard_normal((128, 20)).astype(np.float32),
       'decoder_proj/kernel': rng.standard_normal((10, 128)).astype(np.float32),
       'big/kernel': rng.standard_normal((300, 257)).astype(np.float32),   # several table blocks
       'Step': np.asarray(1234, np.int64)}
  for i in range(40):  # many keys: prefix compression + restarts + more than one data block
    V[f'layer{i:02d}/bias'] = rng.standard_normal(3).astype(np.float32)
  prefix = str(tmp_path / 'ck' / 'model')
  tfc.save_checkpoint(prefix, V)
  assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
  R = tfc.load_checkpoint(prefix)
  assert set(R) == set(V)
  for k in V:
    assert R[k].dtype == V[k].dtype and R[k].shape == V[k].shape and np.array_equal(R[k], V[k])
  # the index is a valid table: magic, sorted keys, header entry first
  items = tfc.read_table(prefix + '.index')
  keys = [k for k, _ in items]
  assert keys == sorted(keys) and keys[0] == b'' and tfc.OBJECT_GRAPH_KEY.encode() in keys
  assert struct.unpack('<Q', open(prefix + '.index', 'rb').read()[-8:])[0] == tfc.TABLE_MAGIC
  # a flipped data byte is caught by the per-tensor checksum
  p = prefix + '.data-00000-of-00001'
  b = bytearray(open(p, 'rb').read())
  b[len(b) // 2] ^= 0x40
  open(p, 'wb').write(bytes(b))
  with pytest.raises(ValueError):
    tfc.load_checkpoint(prefix)


def test_scalar_event_file(tmp_path):
  w = tfc.ScalarEventWriter(str(tmp_path / 'logs'))
  for step in range(1, 4):
    w.scalar('train/loss', 100.0 / step, step)
    w.scalar('train/llk_image', -50.0 * step, step)
  w.close()
  ev = tfc.read_scalar_events(w.path)
  assert len(ev) == 6 and ev[0] == (1, 'train/loss', 100.0) and ev[-1][:2] == (3, 'train/llk_image')


def _keras_fixture():
  """the committed Keras-shaped checkpoint and the values it holds (regenerated from the seeded script)"""
  import importlib.util
  root = os.path.dirname(os.path.abspath(__file__))
  spec = importlib.util.spec_from_file_location('gen_keras_ckpt', os.path.join(root, 'golden', 'gen_keras_ckpt.py'))
  gen = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(gen)
  return gen, os.path.join(root, 'golden', 'keras_ckpt', 'model')


def test_keras_shaped_checkpoint_reader(tmp_path):
  """A checkpoint in the shape keras.Model.save_weights(save_format='tf') writes -- nested object graph
  (`encoder/layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE`), `save_counter`, optimizer hyper-parameters and
  Adam slot variables, two data shards, several index blocks -- assembled by tests/golden/gen_keras_ckpt.py without
  this package's writer: every variable is found under its Keras name."""
  gen, prefix = _keras_fixture()
  expected = gen.build(str(tmp_path / 'regen'))
  for suffix in ('.index', '.data-00000-of-00002', '.data-00001-of-00002'):   # the committed files ARE the script's output
    assert open(prefix + suffix, 'rb').read() == open(str(tmp_path / 'regen' / 'model') + suffix, 'rb').read(), suffix
  got = tfc.load_checkpoint(prefix)
  assert set(got) == set(expected)
  for k, v in expected.items():
    assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
  assert int(got['Step']) == 4321 and int(got['save_counter']) == 3
  assert 'Adam/encoder0/kernel/m' in got and 'Adam/iter' in got
  # the index really has several data blocks + prefix-compressed keys
  keys = [k for k, _ in tfc.read_table(prefix + '.index')]
  assert len(keys) == len(expected) + 2 and keys == sorted(keys)
  # a snappy-compressed block (trailer type 1) is refused with a clear message
  raw = bytearray(open(prefix + '.index', 'rb').read())
  foot = bytes(raw[-48:])
  _, p = tfc._read_varint(foot, 0)
  _, p = tfc._read_varint(foot, p)
  ioff, p = tfc._read_varint(foot, p)
  inn, p = tfc._read_varint(foot, p)
  raw[ioff + inn] = 1
  bad = str(tmp_path / 'snappy')
  open(bad + '.index', 'wb').write(bytes(raw))
  with pytest.raises(ValueError, match='snappy'):
    tfc.read_table(bad + '.index')
  # names with '.' are escaped the way tf.train.Checkpoint escapes local names
  assert tfc.checkpoint_key('a.b/kernel') == 'a..b.Skernel/.ATTRIBUTES/VARIABLE_VALUE'


def test_model_loads_keras_shaped_checkpoint(bk):
  """VariationalAutoencoder.load_weights on that file: encoder / latents / decoder parameters by their Keras
  names (the optimizer's slot variables share suffixes with them and must not be confused), step from `Step`."""
  import torch
  from odin_ai_amd.networks import RVconf, SequentialNetwork
  from odin_ai_amd.vae import VariationalAutoencoder
  gen, prefix = _keras_fixture()
  import tempfile
  with tempfile.TemporaryDirectory() as d:
    expected = gen.build(d)
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',), ('dense', 24, 'linear')]
  dec = [('dense', 32, 'linear'), ('reshape', (2, 2, 8)), ('deconv', 16, 4, 2, 'elu'), ('deconv', 8, 4, 2, 'elu'),
         ('conv', 1, 1, 1, 'linear')]
  nets = dict(encoder=SequentialNetwork(enc, 'Encoder', (8, 8, 1), ['encoder0', 'encoder1', 'encoder_proj']),
              decoder=SequentialNetwork(dec, 'Decoder', (4,), ['decoder_proj', 'decoder1', 'decoder2', 'decoder6']),
              observation=RVconf((8, 8, 1), 'bernoulli', projection=False, name='image'),
              latents=RVconf((4,), 'mvndiag', projection=True, name='latents'))
  vae = VariationalAutoencoder(device=bk.dev, lib=bk.L, **nets).load_weights(prefix, raise_notfound=True)
  assert vae.step == 4321
  for key, v in vae.trainable_variables.items():
    name = vae.variable_name(key)
    assert torch.equal(v.cpu(), torch.as_tensor(expected[name])), name
  # and the loaded model computes: one forward pass, finite ELBO
  x = (np.random.default_rng(0).random((4, 8, 8, 1)) < 0.4).astype(np.float32)
  llk, kl = vae.elbo_components(x)
  assert np.isfinite(llk['llk_image'].numpy(force=True)).all() and np.isfinite(kl['kl_latents'].numpy(force=True)).all()
