"""Speech front-end on the HIP path: STFT -> power -> Slaney mel -> dB.

Mirrors the reference's extractor surface (odin/preprocessing/speech.py:655-929
`STFTExtractor` / `MelsSpecExtractor` / `SpectraExtractor`, which call
odin/preprocessing/signal.py) for the configuration BASELINE.json names: 25 ms / 10 ms
frames at 8 kHz (200 / 80 samples), n_fft 512, periodic Hamming window, 80 Slaney mels
from 64 Hz to 4 kHz, dB with top_db 80.  The window and the filterbank are small host-side
constants; every per-sample FLOP runs in `odin_stft_mel_db`.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib


def _hz2mel(f):
  f = np.atleast_1d(np.asarray(f, np.float64)).copy()
  m = f / (200.0 / 3)
  t = f >= 1000.0
  m[t] = 1000.0 / (200.0 / 3) + np.log(f[t] / 1000.0) / (np.log(6.4) / 27.0)
  return m


def _mel2hz(m):
  m = np.atleast_1d(np.asarray(m, np.float64))
  f = (200.0 / 3) * m
  t = m >= 1000.0 / (200.0 / 3)
  f[t] = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m[t] - 1000.0 / (200.0 / 3)))
  return f


def mel_filters(sr: int, n_fft: int, n_mels: int = 128, fmin: float = 0.0,
                fmax: Optional[float] = None) -> np.ndarray:
  """Slaney-scale, area-normalised triangular filterbank [n_mels, n_fft//2+1]
  (odin/preprocessing/signal.py:735-810)."""
  fmax = float(sr) / 2 if fmax is None else float(fmax)
  freqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
  mel_f = _mel2hz(np.linspace(_hz2mel(fmin)[0], _hz2mel(fmax)[0], n_mels + 2))
  fdiff = np.diff(mel_f)
  ramps = mel_f[:, None] - freqs[None, :]
  W = np.zeros((n_mels, 1 + n_fft // 2))
  for i in range(n_mels):
    W[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
  return W * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def periodic_window(name: str, n: int) -> np.ndarray:
  k = np.arange(n, dtype=np.float64)
  if name in ('hamm', 'hamming'):
    return 0.54 - 0.46 * np.cos(2 * np.pi * k / n)
  if name in ('hann', 'hanning'):
    return 0.5 - 0.5 * np.cos(2 * np.pi * k / n)
  raise ValueError(f'window {name!r} is outside the HIP path (hamm / hann)')


class MelsSpecExtractor:
  """Batched log-mel spectrogram on the GPU.  Arguments follow
  speech.py `STFTExtractor(frame_length, step_length, n_fft, window)` +
  `MelsSpecExtractor(n_mels, fmin, fmax, top_db)`; `frame_length` / `step_length` are in
  seconds when < 1 (as in the reference) else in samples."""

  def __init__(self, sr: int = 8000, frame_length=0.025, step_length=0.010, n_fft: int = 512,
               window: str = 'hamm', n_mels: int = 80, fmin: float = 64, fmax: float = 4000,
               top_db: Optional[float] = 80.0, preemphasis: Optional[float] = 0.97,
               log: bool = True, device=None, lib=None, unit_range: bool = False,
               mel_basis: Optional[np.ndarray] = None, normalize_window: bool = True,
               natural_log: bool = False):
    self.sr = int(sr)
    self.frame_length = int(round(frame_length * sr)) if frame_length < 1 else int(frame_length)
    self.step_length = int(round(step_length * sr)) if step_length < 1 else int(step_length)
    self.n_fft, self.n_mels = int(n_fft), int(n_mels)
    if int(fmin) >= int(fmax):
      raise ValueError(f'fmin must < fmax, but fmin={int(fmin)} and fmax={int(fmax)}')
    self.top_db = -1.0 if top_db is None else float(top_db)
    self.preemph = 0.0 if not preemphasis else float(preemphasis)
    self.log = bool(log)
    # output mode of odin_stft_mel_db: 0 power, 1 dB, 2 dB mapped to [0, 1], 3 ln(mel + 1e-6)
    self.mode = 3 if natural_log else ((2 if unit_range else 1) if log else 0)
    if unit_range and (top_db is None or top_db <= 0 or not log):
      raise ValueError('unit_range=True needs log=True and top_db > 0')
    self.device = torch.device(device if device is not None else
                               ('cuda' if torch.cuda.is_available() else 'cpu'))
    self.lib = lib if lib is not None else _lib.load()
    w = periodic_window(window, self.frame_length)
    f64 = dict(dtype=torch.float64, device=self.device)
    self.window = torch.tensor(w / w.sum() if normalize_window else w, **f64)
    k = np.arange(self.n_fft // 2, dtype=np.float64)
    ang = 2.0 * np.pi * k / self.n_fft
    self.twiddles = torch.tensor(np.stack([np.cos(ang), -np.sin(ang)], 1), **f64).contiguous()
    fb = (mel_filters(self.sr, self.n_fft, self.n_mels, int(fmin), int(fmax))
          if mel_basis is None else np.asarray(mel_basis, np.float64))
    assert fb.shape == (self.n_mels, self.n_fft // 2 + 1), fb.shape
    self.mel_basis = fb
    # the kernel contracts each filter over its non-zero band only
    band, vals = [], []
    for m in range(self.n_mels):
      nz = np.nonzero(fb[m])[0]
      k0, k1 = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 0)
      band.append((k0, k1 - k0, len(vals)))
      vals.extend(fb[m, k0:k1].tolist())
    self.fb_vals = torch.tensor(np.asarray(vals if vals else [0.0]), **f64)
    self.fb_band = torch.tensor(np.asarray(band, np.int32), dtype=torch.int32, device=self.device)

  def n_frames(self, n_samples: int) -> int:
    return 1 + (n_samples - self.frame_length) // self.step_length

  def transform(self, y, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [B, n_samples] (or [n_samples]) -> [B, n_frames, n_mels] float32 on the device.  With `out` (a
    contiguous float32 device tensor of B * T * n_mels elements, T <= n_frames, e.g. the VAE's [B, T, n_mels, 1]
    input buffer) only the first T frames are stored, straight into it (the top_db floor still spans the whole
    utterance): no intermediate spectrogram, no crop copy."""
    y = torch.as_tensor(y, dtype=torch.float32, device=self.device)
    squeeze = y.ndim == 1
    if squeeze:
      y = y[None]
    y = y.contiguous()
    B, n = y.shape
    nf = self.n_frames(n)
    if out is None:
      out = torch.empty(B, nf, self.n_mels, dtype=torch.float32, device=self.device)
      T = nf
    else:
      assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() % (B * self.n_mels) == 0
      T = out.numel() // (B * self.n_mels)
    st = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == 'cuda' else None
    ws = getattr(self, '_ws', None)
    if ws is None or ws.numel() < 8 * B:  # block maxima of the split launch
      ws = self._ws = torch.empty(8 * B, dtype=torch.float32, device=self.device)
    self.lib.odin_stft_mel_db_frames(y.data_ptr(), self.window.data_ptr(), self.twiddles.data_ptr(),
                                     self.fb_vals.data_ptr(), self.fb_band.data_ptr(), out.data_ptr(), B,
                                     n, self.frame_length, self.step_length, self.n_fft, self.n_mels,
                                     self.preemph, self.top_db, self.mode, T, ws.data_ptr(), st)
    return out[0] if squeeze else out

  __call__ = transform


def spectrogram_batch(mel: torch.Tensor, n_frames: int, pad_value: Optional[float] = None
                      ) -> torch.Tensor:
  """[B, T, n_mels] log-mel -> the [B, n_frames, n_mels, 1] input of `speech_networks`: the first
  n_frames frames, or right-padded with `pad_value` (default: the batch minimum, i.e. the top_db
  floor) when the utterances are shorter (fuel/audio_data.py:236-260 crops / pads to max_length)."""
  B, T, M = mel.shape
  if T >= n_frames:
    return mel[:, :n_frames].reshape(B, n_frames, M, 1).contiguous()
  pv = float(mel.min()) if pad_value is None else float(pad_value)
  out = torch.full((B, n_frames, M), pv, dtype=mel.dtype, device=mel.device)
  out[:, :T] = mel
  return out.reshape(B, n_frames, M, 1)


def htk_mel_weight_matrix(num_mel_bins=20, num_spectrogram_bins=129, sample_rate=8000,
                          lower_edge_hertz=125.0, upper_edge_hertz=3800.0) -> np.ndarray:
  """tf.signal.linear_to_mel_weight_matrix ([num_spectrogram_bins, num_mel_bins]) as used by the
  reference's AudioFeatureLoader (odin/fuel/audio_data.py:94-100).  THIRD-PARTY algorithm
  (tensorflow==2.5.0, python/ops/signal/mel_ops.py) restated from its published definition: HTK
  mel scale 1127 ln(1 + f/700), triangles linear in mel between num_mel_bins + 2 equally spaced
  edges, NOT area-normalised, the DC bin zeroed."""
  hz2mel = lambda f: 1127.0 * np.log1p(np.asarray(f, np.float64) / 700.0)
  lin = np.linspace(0.0, sample_rate / 2.0, num_spectrogram_bins)[1:]
  bins_mel = hz2mel(lin)[:, None]
  edges = np.linspace(hz2mel(lower_edge_hertz), hz2mel(upper_edge_hertz), num_mel_bins + 2)
  lower, center, upper = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
  W = np.maximum(0.0, np.minimum((bins_mel - lower) / (center - lower),
                                 (upper - bins_mel) / (upper - center)))
  return np.pad(W, [[1, 0], [0, 0]])


class AudioFeatureLoader:
  """The TF front-end variant of the reference (odin/fuel/audio_data.py:17-101,210-270; used by
  examples/vae/vae_audio.py:64-79) on the same HIP kernel: tf.signal.stft(frame 256, step 80,
  fft 256, periodic Hann, no padding) -> |.|^power -> HTK mel (20 bins, 125-3800 Hz) -> dB with
  `top_DB` relative to the utterance's maximum (or ln(mel + 1e-6) when `log_mels`)."""

  def __init__(self, frame_length=256, frame_step=80, fft_length=None, sample_rate=8000,
               power=2.0, top_DB=80.0, pad_end=False, num_mel_bins=20, log_mels=False,
               lower_edge_hertz=125.0, upper_edge_hertz=3800.0, device=None, lib=None):
    if power != 2.0:
      raise NotImplementedError('power != 2.0 (magnitude spectrogram) is outside the HIP path')
    if pad_end:
      raise NotImplementedError('pad_end=True is outside the HIP path')
    if fft_length is None:
      fft_length = frame_length
    fft_length = 2 ** int(np.ceil(np.log2(fft_length)))                      # :73
    self.frame_length, self.frame_step, self.fft_length = int(frame_length), int(frame_step), fft_length
    self.sample_rate, self.top_DB, self.log_mels = int(sample_rate), top_DB, bool(log_mels)
    self.num_mel_bins = int(num_mel_bins)
    self.mel_weight = htk_mel_weight_matrix(num_mel_bins, fft_length // 2 + 1, sample_rate,
                                            lower_edge_hertz, upper_edge_hertz)
    self._ex = MelsSpecExtractor(sr=sample_rate, frame_length=frame_length, step_length=frame_step,
                                 n_fft=fft_length, window='hann', n_mels=num_mel_bins, fmin=0,
                                 fmax=sample_rate / 2, top_db=top_DB, preemphasis=None, log=True,
                                 device=device, lib=lib, mel_basis=self.mel_weight.T,
                                 normalize_window=False, natural_log=self.log_mels)

  def melspec(self, y) -> torch.Tensor:
    """y [B, n_samples] -> [B, n_frames, num_mel_bins] (:216-226)."""
    return self._ex(y)
