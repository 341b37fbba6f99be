"""Data-parallel helpers: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distributed training; the build shards the
minibatch across ranks, every rank computes the gradient of (1/B_global) * sum over its local
samples, and ONE sum all-reduce of the flat fp32 gradient bucket per optimiser step makes
the replicas identical (`VAEEngine.allreduce`).  `backend="nccl"` IS RCCL on ROCm; the CPU
tests use gloo.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
  """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).
  Returns (rank, local_rank, world_size)."""
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world > 1 and not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
      backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {}
    if backend == 'nccl':
      kw['device_id'] = torch.device('cuda', local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
  return rank, local_rank, world


def shard_batch(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
  """Contiguous shard of the global batch owned by `rank` (global batch % world == 0)."""
  assert x.shape[0] % world == 0
  n = x.shape[0] // world
  return x[rank * n:(rank + 1) * n].contiguous()


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, force: bool = False):
  """Replicas start from rank `src`'s parameters (never rely on identical seeds)."""
  if dist.is_initialized() and (dist.get_world_size() > 1 or force):
    dist.broadcast(flat_params, src=src)
