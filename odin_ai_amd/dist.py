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


class Comm:
  """The collectives of the data-parallel step, enqueued on the CURRENT stream.

  On a GPU the transport is RCCL through the C ABI (`odin_comm_init` / `odin_allreduce_flat` /
  `odin_allgather_flat` / `odin_reduce_scatter_flat`, include/odin_hip.h): a communicator of this
  process's own, whose 128-byte id travels from rank 0 through the torch.distributed process group
  that the launcher initialised.  `ODIN_DIST_COMM=torch` keeps torch.distributed's own RCCL
  communicator instead; the CPU tests (gloo) always take that path (gloo has no reduce-scatter: a sum
  all-reduce + slice stands in)."""

  def __init__(self, lib, device):
    self.lib, self.device = lib, torch.device(device)
    self.world = dist.get_world_size() if dist.is_initialized() else 1
    self.rank = dist.get_rank() if dist.is_initialized() else 0
    self.gloo = dist.is_initialized() and dist.get_backend() == 'gloo'
    self.handle = None
    want_native = self.device.type == 'cuda' and os.environ.get('ODIN_DIST_COMM', 'rccl') != 'torch'
    if want_native:
      import ctypes as C
      idt = torch.zeros(128, dtype=torch.uint8)
      if self.rank == 0:
        buf = (C.c_char * 128)()
        lib.odin_comm_unique_id(C.cast(buf, C.c_void_p))
        idt = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
      if self.world > 1:
        t = idt.to(self.device) if not self.gloo else idt
        dist.broadcast(t, src=0)
        idt = t.cpu()
      raw = bytes(idt.numpy().tobytes())
      h = C.c_void_p()
      lib.odin_comm_init(C.byref(h), C.c_char_p(raw), self.rank, self.world)
      self.handle = h
    self.native = self.handle is not None

  def _st(self):
    return torch.cuda.current_stream(self.device).cuda_stream

  def all_reduce(self, t: torch.Tensor):
    """in-place SUM"""
    if self.native:
      self.lib.odin_allreduce_flat(self.handle, t.data_ptr(), t.numel(), self._st())
    elif dist.is_initialized():
      dist.all_reduce(t, op=dist.ReduceOp.SUM)

  def all_gather(self, out: torch.Tensor, inp: torch.Tensor):
    """out [world * n] <- every rank's inp [n] (flat, contiguous)"""
    assert out.numel() == self.world * inp.numel() and out.is_contiguous() and inp.is_contiguous()
    if self.native:
      self.lib.odin_allgather_flat(self.handle, inp.data_ptr(), out.data_ptr(), inp.numel(), self._st())
    elif self.gloo:
      dist.all_gather(list(out.view(self.world, -1).unbind(0)), inp.reshape(-1))
    else:
      dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1))

  def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor):
    """out [n] <- sum over ranks of inp [rank * n : (rank + 1) * n]; `inp` may be clobbered"""
    assert inp.numel() == self.world * out.numel() and out.is_contiguous() and inp.is_contiguous()
    if self.native:
      self.lib.odin_reduce_scatter_flat(self.handle, inp.data_ptr(), out.data_ptr(), out.numel(), self._st())
    elif self.gloo:
      dist.all_reduce(inp, op=dist.ReduceOp.SUM)
      n = out.numel()
      out.view(-1).copy_(inp.view(-1)[self.rank * n:(self.rank + 1) * n])
    else:
      dist.reduce_scatter_tensor(out.view(-1), inp.view(-1), op=dist.ReduceOp.SUM)

  def close(self):
    if self.handle is not None:
      self.lib.odin_comm_destroy(self.handle)
      self.handle = None


class SegmentedGraph:
  """A step that contains collectives, replayed as HIP graphs: the launch program is a list of
  ('k', fn) -- fn enqueues kernels on the current stream, captured -- and ('c', fn) -- fn enqueues a
  collective, run eagerly between two replays on the same stream (RCCL launches stay outside the
  graphs).  Consecutive 'k' entries share one graph:  A | all-gather | B | reduce-scatter | C |
  all-reduce | D.  On the CPU (no graphs) `run` just calls every fn."""

  def __init__(self, device, program):
    self.device = torch.device(device)
    self.segs = []  # ('k', [fns]) | ('c', fn)
    for kind, fn in program:
      if kind == 'k' and self.segs and self.segs[-1][0] == 'k':
        self.segs[-1][1].append(fn)
      else:
        self.segs.append((kind, [fn] if kind == 'k' else fn))
    self.graphs = None

  def run_eager(self):
    for kind, f in self.segs:
      if kind == 'k':
        for fn in f:
          fn()
      else:
        f()

  def capture(self, cap_stream):
    """capture every kernel segment (no kernel runs); the caller has warmed the program up"""
    import gc
    self.graphs = []
    # No cyclic garbage collection while a stream is capturing: a collection that frees an object holding a HIP resource
    # (an earlier engine's graph, an event, a communicator) issues HIP calls that are illegal during capture and abort
    # the process (seen in round 6: `Garbage-collecting` inside _backward_enc under capture, "Fatal Python error:
    # Aborted").  torch.cuda.graph() collects once before the capture begins; the launch program itself allocates
    # Python objects (ctypes arrays, tuples), so a collection can still trigger in the middle.
    was_enabled = gc.isenabled()
    gc.disable()
    try:
      for kind, f in self.segs:
        if kind != 'k':
          self.graphs.append(None)
          continue
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap_stream, capture_error_mode='thread_local'):
          for fn in f:
            fn()
        self.graphs.append(g)
    finally:
      if was_enabled:
        gc.enable()

  def replay(self):
    for (kind, f), g in zip(self.segs, self.graphs):
      if g is not None:
        g.replay()
      else:
        f()

  @property
  def n_graphs(self):
    return sum(g is not None for g in (self.graphs or []))
