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
