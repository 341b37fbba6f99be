"""Mel front-end kernel sources on the CPU simulator vs the reference-generated goldens."""
import os

import numpy as np
import pytest

from odin_ai_amd.mel import MelsSpecExtractor, mel_filters
from tests.simutil import sim_lib

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mel_golden.npz'))


def test_product_filterbank_matches_reference():
  np.testing.assert_allclose(mel_filters(8000, 512, 80, 64, 4000), G['mel_basis'], atol=1e-13)


def test_mel_kernel_matches_reference_golden():
  L = sim_lib()
  ex = MelsSpecExtractor(device='cpu', lib=L)
  y = G['y'][:2, :2000]  # 23 frames per utterance keeps the simulator fast
  out = ex(y).numpy()
  assert out.shape == (2, 23, 80)
  from oracle import mel_oracle as mo
  for i in range(2):
    ref = mo.mel_frontend(y[i])  # oracle is pinned to the reference (tests/test_mel_oracle.py)
    # fp32 FFT vs float64 reference: tolerance 2e-3 dB
    assert np.abs(out[i] - ref).max() < 2e-3, np.abs(out[i] - ref).max()
  with pytest.raises(ValueError):
    MelsSpecExtractor(fmin=5000, fmax=4000, device='cpu', lib=L)
