#!/usr/bin/env python3
"""Generates tests/golden/mel_golden.npz by RUNNING THE REFERENCE's own numpy code
(/root/reference/odin/preprocessing/signal.py) in the build container.  The reference file
is loaded in place -- nothing of it is copied into this repository; only the input signals
and the arrays it returns are stored.

Two shims are needed to import that file on Python 3.10 / numpy 2:
  1. fake `odin` / `odin.utils` modules exposing `cache_memory` / `cache_disk` decorators
     (signal.py:33 imports the py3.7-only odin.utils);
  2. `hz2mel` wrapped to return a Python scalar for scalar input, because np.atleast_1d
     (:514) + np.linspace(array, array, n) (:786) yields shape (n,1) on numpy >= 1.16.
`mels_spectrogram` is called directly (not `spectra()`, which drops fmin/fmax, :1818).
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/odin/preprocessing/signal.py'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                   'mel_golden.npz')


def load_reference():
  def cache(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
      return a[0]
    return lambda f: f
  odin = types.ModuleType('odin')
  utils = types.ModuleType('odin.utils')
  utils.cache_memory = cache
  utils.cache_disk = cache
  odin.utils = utils
  sys.modules.setdefault('odin', odin)
  sys.modules.setdefault('odin.utils', utils)
  spec = importlib.util.spec_from_file_location('ref_signal', REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  orig = mod.hz2mel
  mod.hz2mel = lambda f: (float(orig(f)[0]) if np.isscalar(f) else orig(f))
  return mod


def signals(sr=8000, n=8000):
  rng = np.random.default_rng(1)
  t = np.arange(n) / sr
  chirp = np.sin(2 * np.pi * (200 * t + 0.5 * 1500 * t ** 2))
  y0 = (0.1 * rng.standard_normal(n) + 0.5 * chirp).astype(np.float32)
  y1 = (0.1 * rng.standard_normal(n)).astype(np.float32)
  y2 = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)
  return np.stack([y0, y1, y2])


def main():
  ref = load_reference()
  Y = signals()
  out = dict(y=Y)
  for i, y in enumerate(Y):
    ye = ref.pre_emphasis(y.astype(np.float64), 0.97)
    S = ref.stft(ye, frame_length=200, step_length=80, n_fft=512, window='hamm')
    P = ref.power_spectrogram(S, power=2.0)
    M = ref.mels_spectrogram(P, sr=8000, n_mels=80, fmin=64, fmax=4000, top_db=80.0)
    if i == 0:
      out[f'stft_{i}'] = S
    out[f'mel_db_{i}'] = M
  out['mel_basis'] = ref.mel_filters(8000, 512, 80, 64, 4000)
  out['mel_basis_24'] = ref.mel_filters(8000, 512, 24, 64, 4000)
  os.makedirs(os.path.dirname(OUT), exist_ok=True)
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
  main()
