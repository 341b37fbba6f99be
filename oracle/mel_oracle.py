"""CPU ORACLE (test infrastructure, NOT product code) for the speech front-end.

float64 numpy restatement of the reference's offline feature path
  pre_emphasis        odin/preprocessing/signal.py:955-967
  stft                odin/preprocessing/signal.py:1442-1562 (framing :1532-1538,
                      periodic window * frames :1542-1549, rfft :1555, scale 1/sum(window))
  power_spectrogram   :1623-1648   (|S|**2)
  hz2mel / mel2hz     :489-568     (Slaney scale)
  mel_filters         :735-810     (area-normalised triangles)
  mels_spectrogram    :1650-1691   (basis . P^T, then power2db)
  power2db            :636-680     (10 log10 max(amin, .), clamp at global max - top_db)

PARITY STATUS: **pinned** -- tests/golden/mel_golden.npz holds outputs of the reference's
own signal.py executed in the build container (oracle/gen_mel_golden.py; two import
shims documented there), and tests/test_mel_oracle.py checks this restatement against
them to 1e-10.
"""
from __future__ import annotations

import numpy as np


def pre_emphasis(s, coeff=0.97):
  s = np.asarray(s, np.float64)
  if s.ndim == 1:
    return np.append(s[0], s[1:] - coeff * s[:-1])
  return s - np.c_[s[:, :1], s[:, :-1]] * coeff


def get_window(name: str, n: int):
  """scipy.signal.get_window(name, n, fftbins=True) for the two windows the reference's
  extractors default to ('hamm' speech.py:695, 'hann' speech.py:883): periodic forms."""
  k = np.arange(n, dtype=np.float64)
  if name in ('hamm', 'hamming'):
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * k / n)
  if name in ('hann', 'hanning'):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)
  raise ValueError(name)


def stft(y, frame_length, step_length, n_fft, window='hamm'):
  y = np.asarray(y, np.float64)
  n_frames = 1 + (y.shape[-1] - frame_length) // step_length
  idx = np.arange(frame_length)[None, :] + step_length * np.arange(n_frames)[:, None]
  frames = y[idx]
  w = get_window(window, frame_length)
  scale = np.sqrt(1.0 / w.sum() ** 2)
  return np.fft.rfft(frames * w[None, :], n=n_fft, axis=-1) * scale


def hz2mel(f):
  f = np.atleast_1d(np.asarray(f, np.float64)).copy()
  f_sp = 200.0 / 3
  mels = f / f_sp
  min_log_hz = 1000.0
  min_log_mel = min_log_hz / f_sp
  logstep = np.log(6.4) / 27.0
  t = f >= min_log_hz
  mels[t] = min_log_mel + np.log(f[t] / min_log_hz) / logstep
  return mels


def mel2hz(m):
  m = np.atleast_1d(np.asarray(m, np.float64))
  f_sp = 200.0 / 3
  freqs = f_sp * m
  min_log_hz = 1000.0
  min_log_mel = min_log_hz / f_sp
  logstep = np.log(6.4) / 27.0
  t = m >= min_log_mel
  freqs[t] = min_log_hz * np.exp(logstep * (m[t] - min_log_mel))
  return freqs


def mel_filters(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
  if fmax is None:
    fmax = float(sr) / 2
  weights = np.zeros((n_mels, 1 + n_fft // 2))
  fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2, endpoint=True)
  mel_f = mel2hz(np.linspace(float(hz2mel(fmin)[0]), float(hz2mel(fmax)[0]), n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = np.subtract.outer(mel_f, fftfreqs)
  for i in range(n_mels):
    lower = -ramps[i] / fdiff[i]
    upper = ramps[i + 2] / fdiff[i + 1]
    weights[i] = np.maximum(0, np.minimum(lower, upper))
  enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
  return weights * enorm[:, None]


def power2db(S, amin=1e-10, top_db=80.0):
  log_spec = 10.0 * np.log10(np.maximum(amin, np.abs(S)))
  if top_db is not None:
    log_spec = np.maximum(log_spec, log_spec.max() - top_db)
  return log_spec


def mel_frontend(y, sr=8000, frame_length=200, step_length=80, n_fft=512, n_mels=80, fmin=64,
                 fmax=4000, preemph=0.97, window='hamm', top_db=80.0, log=True):
  """One utterance y [n_samples] -> [n_frames, n_mels] (float64)."""
  if preemph is not None and preemph > 0:
    y = pre_emphasis(y, preemph)
  S = stft(y, frame_length, step_length, n_fft, window)
  P = np.abs(S) ** 2
  M = (mel_filters(sr, n_fft, n_mels, int(fmin), int(fmax)) @ P.T).T
  return power2db(M, top_db=top_db) if log else M


# ---- TF variant (odin/fuel/audio_data.py:17-101,210-270).  PARITY UNPINNED: the arithmetic lives
# in tensorflow==2.5.0 (tf.signal.stft / linear_to_mel_weight_matrix, absent here); restated from
# the published definitions of those ops and from the reference's own amplitude_to_DB (:258-268).
def tf_hann_window(n):
  """tf.signal.hann_window(n, periodic=True)."""
  return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def tf_linear_to_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                                   lower_edge_hertz=125.0, upper_edge_hertz=3800.0):
  def h2m(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)
  lin = np.linspace(0.0, sample_rate / 2.0, num_spectrogram_bins)[1:]
  sm = h2m(lin)[:, None]
  edges = np.linspace(h2m(lower_edge_hertz), h2m(upper_edge_hertz), num_mel_bins + 2)
  W = np.zeros((num_spectrogram_bins - 1, num_mel_bins))
  for i in range(num_mel_bins):
    lo, ce, up = edges[i], edges[i + 1], edges[i + 2]
    W[:, i] = np.maximum(0.0, np.minimum((sm[:, 0] - lo) / (ce - lo), (up - sm[:, 0]) / (up - ce)))
  return np.concatenate([np.zeros((1, num_mel_bins)), W], 0)


def tf_audio_melspec(y, frame_length=256, frame_step=80, fft_length=256, sample_rate=8000,
                     num_mel_bins=20, lower=125.0, upper=3800.0, top_db=80.0, log_mels=False):
  """AudioFeatureLoader.stft -> magnitude (power 2) -> melspec for one utterance."""
  y = np.asarray(y, np.float64)
  n_frames = 1 + (y.shape[-1] - frame_length) // frame_step
  idx = np.arange(frame_length)[None, :] + frame_step * np.arange(n_frames)[:, None]
  S = np.fft.rfft(y[idx] * tf_hann_window(frame_length)[None, :], n=fft_length, axis=-1)
  mel = (np.abs(S) ** 2) @ tf_linear_to_mel_weight_matrix(num_mel_bins, fft_length // 2 + 1,
                                                          sample_rate, lower, upper)
  if log_mels:
    return np.log(mel + 1e-6)
  db = 10.0 * np.log10(np.maximum(mel, 1e-10))
  return np.maximum(db, db.max() - top_db) if top_db is not None else db
