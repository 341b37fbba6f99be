"""Scalar schedules of the training step (host side).

Restates ``Interpolation.apply`` (odin/backend/interpolation.py:82-99) and the curve families
defined beside it (:105-240), used by AnnealingVAE / FactorVAE
(odin/bay/vi/autoencoder/beta_vae.py:99-107) and by cyclical-annealing configurations:

    a = max(step, 1e-8)
    cyclical:      a = a mod (delay_in + steps + delay_out) + 1;  a = clip(a - delay_in, 0, steps)
    non-cyclical:  a = a - delay_in
    a = clip(a / steps, 0, 1);   value = (vmax - vmin) * alpha(a) + vmin

(the `+ 1` phase of the cyclical branch and the `delay_out` plateau are the reference's; the
reference evaluates in float32, the host value here is a Python float -- the difference is below
1e-7 relative and far below what the device hyper buffer (float32) resolves).
The value is handed to the kernels through the device hyper-parameter buffer.
"""
from __future__ import annotations

import math
from dataclasses import dataclass


def _alpha(kind: str, a: float, power: float, inverse: bool, scale: float) -> float:
  if kind == 'linear':
    return a
  if kind in ('smooth', 'smooth2'):          # interpolation.py:124-134
    return a * a * (3 - 2 * a)
  if kind in ('fade', 'smoother'):           # :136-141
    return a * a * a * (a * (a * 6 - 15) + 10)
  if kind == 'power':                        # :146-169
    if a <= 0.5:
      return (a * 2) ** power / 2
    return ((a - 1) * 2) ** power / ((power % 2 - 0.5) * 4) + 1
  if kind == 'powerIn':                      # :172-179
    return a ** (1.0 / power) if inverse else a ** power
  if kind == 'powerOut':                     # :182-189
    if inverse:
      return 1 - (-(a - 1)) ** (1.0 / power)
    return (a - 1) ** power * (power % 2 - 0.5) * 2 + 1
  if kind == 'sine':                         # :195-198
    return (1 - math.cos(a * math.pi)) / 2
  if kind == 'sineIn':
    return 1 - math.cos(a * math.pi / 2)
  if kind == 'sineOut':
    return math.sin(a * math.pi / 2)
  if kind == 'circle':                       # :213-218
    if a <= 0.5:
      return (1 - math.sqrt(1 - (a * 2) ** 2)) / 2
    return (math.sqrt(1 - ((a - 1) * 2) ** 2) + 1) / 2
  if kind == 'circleIn':
    return 1 - math.sqrt(1 - a * a)
  if kind == 'circleOut':
    return math.sqrt(1 - (a - 1) ** 2)
  raise ValueError(f'unknown interpolation {kind!r}')


@dataclass
class Interpolation:
  vmin: float = 0.0
  vmax: float = 1.0
  steps: float = 1
  delay_in: float = 0.0
  delay_out: float = 0.0
  cyclical: bool = False
  kind: str = 'linear'
  power: float = 2.0
  inverse: bool = False
  scale: float = 3.0

  def __post_init__(self):
    self.delay_in = max(self.delay_in, 0)
    self.delay_out = max(self.delay_out, 0)

  @property
  def length(self):
    return self.steps

  def __call__(self, step) -> float:
    if self.kind == 'const':                 # interpolation.py:105-109
      return float(self.vmax)
    a = max(float(step), 1e-8)
    if self.cyclical:
      a = a % (self.delay_in + self.steps + self.delay_out) + 1
      a = a - self.delay_in
      a = max(min(a, self.steps), 0.0)
    else:
      a = a - self.delay_in
    a = a / self.steps
    a = max(0.0, min(a, 1.0))
    return (self.vmax - self.vmin) * _alpha(self.kind, a, self.power, self.inverse, self.scale) + self.vmin

  apply = __call__


def _make(kind):
  def ctor(vmin: float = 0.0, vmax: float = 1.0, steps: float = 1, delay_in: float = 0.0,
           delay_out: float = 0.0, cyclical: bool = False, **kw) -> Interpolation:
    if 'length' in kw:  # the power / swing families of the reference call it `length`
      steps = kw.pop('length')
    return Interpolation(vmin, vmax, steps, delay_in, delay_out, cyclical, kind, **kw)
  ctor.__name__ = kind
  ctor.__doc__ = f'odin.backend.interpolation.{kind}'
  return ctor


const = _make('const')
linear = _make('linear')
smooth = _make('smooth')
smooth2 = _make('smooth2')
fade = _make('fade')
smoother = fade
power = _make('power')
powerIn = _make('powerIn')
powerOut = _make('powerOut')
sine = _make('sine')
sineIn = _make('sineIn')
sineOut = _make('sineOut')
circle = _make('circle')
circleIn = _make('circleIn')
circleOut = _make('circleOut')


_KINDS = ('const', 'linear', 'smooth', 'smooth2', 'fade', 'smoother', 'power', 'powerIn', 'powerOut', 'sine', 'sineIn',
          'sineOut', 'circle', 'circleIn', 'circleOut')


def get(name=None):
  """odin.backend.interpolation.get (interpolation.py:420-428): the constructor registered under `name`, or all of
  them (sorted by name) for None; an unknown name raises (the reference: KeyError from its dictionary)."""
  table = {k: globals()[k] for k in _KINDS if k in globals()}
  if name is None:
    return [v for _, v in sorted(table.items())]
  if name not in table:
    raise KeyError(f"unknown interpolation {name!r} (known: {', '.join(sorted(table))})")
  return table[name]
