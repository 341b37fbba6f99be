"""N>1 path on CPU: two gloo ranks (kernel sources on the simulator) == one rank on the
full batch.  Covers the flat-bucket all-reduce, the 1/B_global loss scaling and the
replicated Adam step of `VAEEngine`."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spec():
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',),
         ('dense', 24, 'linear')]
  dec = [('dense', 32, 'linear'), ('reshape', (2, 2, 8)), ('deconv', 16, 4, 2, 'elu'),
         ('deconv', 8, 4, 2, 'elu'), ('conv', 1, 1, 1, 'linear')]
  return enc, dec, (8, 8, 1), 4


def _data(B):
  rng = np.random.default_rng(5)
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, 4)).astype(np.float32)
  return torch.tensor(x), torch.tensor(eps)


def _init(eng):
  g = torch.Generator().manual_seed(0)
  eng.params.copy_(torch.randn(eng.params.numel(), generator=g) * 0.1)


def _worker(rank, world, port, out_path):
  sys.path.insert(0, ROOT)
  from odin_ai_amd import _lib
  from odin_ai_amd.dist import shard_batch
  from odin_ai_amd.engine import VAEEngine
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  L = _lib.Lib(os.path.join(ROOT, 'tests', 'sim', 'libodin_sim.so'))
  enc, dec, shp, D = _spec()
  B = 8
  x, eps = _data(B)
  eng = VAEEngine(enc, dec, shp, D, B // world, 'cpu', lib=L, world_size=world)
  _init(eng)
  xs, es = shard_batch(x, rank, world), shard_batch(eps, rank, world)
  for _ in range(3):
    eng.train_step(xs, es, lr=1e-3, beta=4.0, global_clipnorm=100.0)
  if rank == 0:
    torch.save(eng.params.clone(), out_path)
  dist.destroy_process_group()


def test_two_gloo_ranks_match_single_rank(tmp_path):
  from odin_ai_amd.engine import VAEEngine
  from tests.simutil import sim_lib
  L = sim_lib()
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  out = str(tmp_path / 'p2.pt')
  mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
  p2 = torch.load(out)
  enc, dec, shp, D = _spec()
  x, eps = _data(8)
  eng = VAEEngine(enc, dec, shp, D, 8, 'cpu', lib=L, world_size=1)
  _init(eng)
  for _ in range(3):
    eng.train_step(x, eps, lr=1e-3, beta=4.0, global_clipnorm=100.0)
  d = (eng.params - p2).abs()
  # identical maths, different fp32 summation order (per-rank partial sums): Adam's
  # normalisation bounds the per-step drift by a fraction of lr
  assert d.max().item() < 2e-4 and d.mean().item() < 2e-6, (d.max().item(), d.mean().item())
