"""Helpers for running the kernel sources through the CPU fiber simulator (tests only)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_LIB = os.path.join(ROOT, 'tests', 'sim', 'libodin_sim.so')


def sim_lib():
  """Build (if stale) and load tests/sim/libodin_sim.so; skip when no host clang++."""
  from odin_ai_amd import _lib
  mk = os.path.join(ROOT, 'odin_ai_amd', 'csrc')
  # (a stale library -- a header changed -- rebuilds 27 translation units: side by side, not one after the other)
  r = subprocess.run(['make', '-j', str(min(8, os.cpu_count() or 1)), '-C', mk, 'sim'], capture_output=True, text=True)
  if r.returncode != 0:
    pytest.skip('cannot build the simulator library: ' + r.stderr[-400:])
  return _lib.Lib(SIM_LIB)
