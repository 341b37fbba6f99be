import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


class Backend:
  """One kernel library + the device its buffers live on.  'sim' = the kernel SOURCES built for
  the CPU fiber simulator (tests/sim, debugging aid, runs without a GPU); 'hip' = the hipcc
  gfx950 build on cuda:0 -- the op-level parity tests run unchanged on both."""

  def __init__(self, name, L, dev):
    import torch
    self.name, self.L, self.dev = name, L, torch.device(dev)

  def T(self, a, dt=None):
    import numpy as np
    import torch
    return torch.tensor(np.ascontiguousarray(a), dtype=dt or torch.float32, device=self.dev)

  def zeros(self, *shape, dtype=None):
    import torch
    if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
      shape = tuple(shape[0])
    return torch.zeros(*shape, dtype=dtype or torch.float32, device=self.dev)

  def full(self, shape, val):
    import torch
    return torch.full(tuple(shape), val, dtype=torch.float32, device=self.dev)


@pytest.fixture(scope='session', params=['sim', pytest.param('hip', marks=pytest.mark.gpu)])
def bk(request):
  if request.param == 'sim':
    from tests.simutil import sim_lib
    return Backend('sim', sim_lib(), 'cpu')
  import torch
  from odin_ai_amd import _lib
  assert torch.cuda.is_available(), 'the hip backend needs an MI355X'
  return Backend('hip', _lib.load(), 'cuda:0')
