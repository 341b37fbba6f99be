"""Scalar schedules of the training step (host side).

Restates odin/backend/interpolation.py:82-122 (``Interpolation.apply`` + ``linear``) for
the non-cyclical branch used by AnnealingVAE / FactorVAE
(odin/bay/vi/autoencoder/beta_vae.py:99-107): ``a = max(step, 1e-8)``,
``a = clip((a - delay_in) / steps, 0, 1)``, ``value = (vmax - vmin) * a + vmin``.
The value is a host float handed to the kernels through the device hyper-parameter buffer.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class Interpolation:
  vmin: float = 0.0
  vmax: float = 1.0
  steps: int = 1000
  delay_in: float = 0.0
  cyclical: bool = False
  kind: str = 'linear'

  def _curve(self, a: float) -> float:
    if self.kind == 'linear':
      return a
    if self.kind == 'power':
      return a ** 2
    raise ValueError(self.kind)

  def __call__(self, step) -> float:
    a = max(float(step), 1e-8)
    if self.cyclical:
      period = self.steps + self.delay_in
      a = a % period
    a = (a - self.delay_in) / self.steps
    a = min(max(a, 0.0), 1.0)
    return (self.vmax - self.vmin) * self._curve(a) + self.vmin


def linear(vmin: float = 0.0, vmax: float = 1.0, steps: int = 1000, delay_in: float = 0.0,
           cyclical: bool = False) -> Interpolation:
  return Interpolation(vmin, vmax, steps, delay_in, cyclical, 'linear')
