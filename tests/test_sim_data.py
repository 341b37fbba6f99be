"""Input-pipeline kernel (uint8 gather + ImageDataset.normalize) on the CPU simulator build,
bit-exact against the numpy restatement."""
import numpy as np
import pytest
import torch

from odin_ai_amd.data import DeviceImageDataset
from oracle import data_oracle as do
from tests.simutil import sim_lib


@pytest.fixture(scope='module')
def L():
  return sim_lib()


@pytest.mark.parametrize('mode,premul', [('probs', 1.0), ('tanh', 1.0), ('raster', 1.0),
                                         ('probs', 255.0), ('binarized', 1.0)])
def test_gather_normalize_bit_exact(L, mode, premul):
  rng = np.random.default_rng(0)
  hi = 2 if (premul != 1.0 or mode == 'binarized') else 256
  imgs = rng.integers(0, hi, size=(11, 8, 8, 3), dtype=np.uint8)
  ds = DeviceImageDataset(imgs, batch_size=4, normalize=mode, premul=premul, device='cpu', lib=L)
  idx = np.array([10, 0, 3, 3], np.int32)
  got = ds.gather(torch.from_numpy(idx)).numpy()
  ref = do.gather_normalize(imgs, idx, mode, premul)
  assert got.dtype == np.float32 and got.shape == (4, 8, 8, 3)
  assert np.array_equal(got, ref)


def test_dataset_iteration_covers_an_epoch(L):
  imgs = (np.arange(10 * 4 * 4 * 1) % 251).astype(np.uint8).reshape(10, 4, 4, 1)
  out = torch.empty(3, 4, 4, 1)
  ds = DeviceImageDataset(imgs, batch_size=3, shuffle=True, seed=5, device='cpu', lib=L, out=out)
  seen = []
  for xb in ds:
    assert xb.data_ptr() == out.data_ptr()
    seen.append(xb.clone())
  assert len(seen) == len(ds) == 3
  ref = do.normalize(imgs, 'probs')
  flat = {r.tobytes() for r in ref}
  for xb in seen:
    for row in xb.numpy():
      assert row.tobytes() in flat  # every yielded image is a normalised dataset image
  # ragged / invalid arguments are rejected
  with pytest.raises(ValueError):
    DeviceImageDataset(imgs[:, :3], batch_size=2, device='cpu', lib=L)
  with pytest.raises(ValueError):
    DeviceImageDataset(imgs, batch_size=11, device='cpu', lib=L)
