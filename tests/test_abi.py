"""The C-ABI library loads and exports every symbol include/odin_hip.h declares
(no compute calls: runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  txt = open(os.path.join(ROOT, 'include', 'odin_hip.h')).read()
  return sorted(set(re.findall(r'\b(odin_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
  from odin_ai_amd import _lib
  syms = set(declared_symbols())
  bound = set(_lib.SIGNATURES) | {'odin_last_error'}
  assert syms == bound, (sorted(syms - bound), sorted(bound - syms))


def test_product_library_exports_every_declared_symbol():
  from odin_ai_amd import _lib
  if not os.path.exists(_lib.DEFAULT_LIB):
    import __graft_entry__ as g
    g.build()
  import ctypes
  lib = ctypes.CDLL(_lib.DEFAULT_LIB)
  for s in declared_symbols():
    assert hasattr(lib, s), s
  L = _lib.Lib(_lib.DEFAULT_LIB)
  assert L.odin_version() >= 100 and L.odin_max_slab_rows() >= 64


def test_no_oracle_import_in_product():
  pkg = os.path.join(ROOT, 'odin_ai_amd')
  for fn in os.listdir(pkg):
    if fn.endswith('.py'):
      src = open(os.path.join(pkg, fn)).read()
      assert 'import oracle' not in src and 'from oracle' not in src, fn
