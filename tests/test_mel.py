"""Mel front-end kernel vs the reference-generated goldens and the numpy restatements, on both
backends of the `bk` fixture (CPU simulator build / gfx950 build on an MI355X): numpy front-end
(a24) and the TF AudioFeatureLoader variant (a25)."""
import os

import numpy as np
import pytest

from odin_ai_amd.mel import MelsSpecExtractor, mel_filters

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mel_golden.npz'))


def test_product_filterbank_matches_reference():
  np.testing.assert_allclose(mel_filters(8000, 512, 80, 64, 4000), G['mel_basis'], atol=1e-13)


def test_mel_kernel_matches_reference_golden(bk):
  L, DEV = bk.L, bk.dev
  ex = MelsSpecExtractor(device=DEV, lib=L)
  y = G['y'][:2, :2000]  # 23 frames per utterance keeps the simulator fast
  out = ex(y).numpy(force=True)
  assert out.shape == (2, 23, 80)
  from oracle import mel_oracle as mo
  for i in range(2):
    ref = mo.mel_frontend(y[i])  # oracle is pinned to the reference (tests/test_mel_oracle.py)
    # float64 arithmetic like the reference, float32 storage: 2e-5 dB absolute on values up to
    # ~100 dB in magnitude (north_star's 1e-4 relative would allow 1e-2 dB there)
    assert np.abs(out[i] - ref).max() < 2e-5, np.abs(out[i] - ref).max()
  # other FFT sizes of the radix-4 family, no pre-emphasis, hann window, power output
  for n_fft, fl in ((128, 100), (2048, 400)):
    ex2 = MelsSpecExtractor(frame_length=fl, step_length=80, n_fft=n_fft, window='hann', n_mels=24,
                            preemphasis=None, log=False, device=DEV, lib=L)
    o2 = ex2(y[:1, :1200]).numpy(force=True)[0]
    r2 = mo.mel_frontend(y[0, :1200], frame_length=fl, n_fft=n_fft, n_mels=24, preemph=None,
                         window='hann', log=False)
    assert np.abs(o2 - r2).max() <= 1e-6 * np.abs(r2).max(), (n_fft, np.abs(o2 - r2).max())
  with pytest.raises(ValueError):
    MelsSpecExtractor(fmin=5000, fmax=4000, device=DEV, lib=L)


def test_unit_range_output_and_tf_variant(bk):
  """log_output=2 maps the floored dB into [0, 1]; the TF AudioFeatureLoader variant (fft 256:
  radix-4 stages + one radix-2 stage, HTK filterbank, un-normalised Hann window) matches its
  numpy restatement."""
  from odin_ai_amd.mel import AudioFeatureLoader
  from oracle import mel_oracle as mo
  L, DEV = bk.L, bk.dev
  y = G['y'][:2, :1600]
  ex = MelsSpecExtractor(device=DEV, lib=L, unit_range=True)
  out = ex(y).numpy(force=True)
  for i in range(2):
    ref = mo.mel_frontend(y[i])
    want = (ref - ref.max()) / 80.0 + 1.0
    assert np.abs(out[i] - want).max() < 1e-6 and out[i].min() >= 0.0 and out[i].max() == 1.0
  # the first T frames written straight into a [B, T, n_mels, 1] buffer (the VAE's input): the same values as
  # cropping the full spectrogram -- the top_db floor still spans all frames
  import torch
  buf = torch.full((2, 10, 80, 1), float('nan'), dtype=torch.float32, device=DEV)
  assert ex(y, out=buf) is buf
  assert np.array_equal(buf.numpy(force=True)[..., 0], out[:, :10])
  ex_db = MelsSpecExtractor(device=DEV, lib=L)
  full = ex_db(y).numpy(force=True)
  buf2 = torch.empty(2, 7, 80, dtype=torch.float32, device=DEV)
  ex_db(y, out=buf2)
  assert np.array_equal(buf2.numpy(force=True), full[:, :7])
  for log_mels in (False, True):
    al = AudioFeatureLoader(device=DEV, lib=L, log_mels=log_mels)
    o = al.melspec(y).numpy(force=True)
    assert o.shape == (2, 17, 20)
    for i in range(2):
      r = mo.tf_audio_melspec(y[i], log_mels=log_mels)
      assert np.abs(o[i] - r).max() < 2e-5, (log_mels, np.abs(o[i] - r).max())
  from odin_ai_amd.mel import htk_mel_weight_matrix
  np.testing.assert_allclose(htk_mel_weight_matrix(), mo.tf_linear_to_mel_weight_matrix(), atol=1e-14)


@pytest.mark.parametrize('kw,n', [(dict(), 5203), (dict(preemphasis=None, window='hann', log=False, n_mels=24), 4001),
                                  (dict(frame_length=512, step_length=100, n_mels=40), 3100)])
def test_fft512_register_kernel_against_the_general_one(bk, request, kw, n):
  """n_fft = 512 runs on stft_mel512_kernel (two radix-16 passes in registers, 16 lanes per frame, mel.hip); the general
  radix-4 kernel serves every other size.  Same float64 arithmetic: the two agree to the rounding of the twiddle powers
  (dB: 1e-6; power: 1e-11 relative), ragged last blocks of 16 frames, odd sample counts, frames split over workgroups,
  and both match the oracle (signal.py:955-967, 1442-1562, 1623-1691, 636-680)."""
  from oracle import mel_oracle as mo
  L, DEV = bk.L, bk.dev
  y = np.tile(G['y'][:3], (1, 2))[:, :n]
  ex = MelsSpecExtractor(device=DEV, lib=L, **kw)
  request.addfinalizer(lambda old=L.odin_debug_mel_r16(-1): L.odin_debug_mel_r16(old))
  L.odin_debug_mel_r16(1)
  a = ex(y).numpy(force=True)
  L.odin_debug_mel_r16(0)
  b = ex(y).numpy(force=True)
  assert a.shape == b.shape and a.shape[0] == 3
  if kw.get('log', True):
    assert np.abs(a - b).max() <= 1e-6, np.abs(a - b).max()
  else:
    assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max(), np.abs(a - b).max()
  okw = {k if k != 'preemphasis' else 'preemph': v for k, v in kw.items()}
  ref = mo.mel_frontend(y[1], **okw)
  if kw.get('log', True):
    assert np.abs(a[1] - ref).max() < 2e-5
  else:
    assert np.abs(a[1] - ref).max() <= 1e-6 * np.abs(ref).max()
