"""HIP execution engine for one VAE training step (host side of the C ABI).

`NetProgram` turns a sequential encoder / decoder description (the layer tuples of
``odin_ai_amd.networks``) into a static list of kernel launches over preallocated HBM
buffers; `VAEEngine` strings encoder -> latent -> decoder -> ELBO -> backward -> slab
reduction -> (all-reduce) -> Adam together and can replay the whole step as ONE captured
HIP graph.  PyTorch is used for device memory, streams and RCCL only: every FLOP of the
step runs in ``libodin_hip.so``.

Reference path restated: Networks.optimize (odin/networks/base_networks.py:415-624) ->
VAEStep.call (odin/bay/vi/autoencoder/variational_autoencoder.py:117-126) ->
elbo_components (:515-542).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import ACT, ConvDesc, ReduceJob


def same_pads(n: int, k: int, s: int) -> Tuple[int, int, int]:
  """TF SAME: (out, pad_before, pad_after)."""
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  return out, total // 2, total - total // 2


# --------------------------------------------------------------------------------------
# parameter layout
# --------------------------------------------------------------------------------------
class ParamLayout:
  """Flat fp32 parameter buffer: per trainable layer [W (Keras layout) | b] contiguous,
  so that one wgrad slab reduction writes (dW | db) in place and ONE RCCL all-reduce /
  ONE Adam launch covers the model."""

  def __init__(self):
    self.entries: List[Tuple[tuple, Tuple[int, ...], int]] = []  # (key, shape, offset)
    self.size = 0

  def add(self, key, shape) -> int:
    off = self.size
    self.entries.append((key, tuple(shape), off))
    self.size += int(np.prod(shape))
    return off

  def pad_to(self, mult=4):
    self.size = (self.size + mult - 1) // mult * mult

  def views(self, flat: torch.Tensor) -> Dict[tuple, torch.Tensor]:
    return {k: flat[o:o + int(np.prod(s))].view(s) for k, s, o in self.entries}


class LayerRec:
  __slots__ = ('kind', 'act', 'key', 'w_off', 'b_off', 'w_n', 'b_n', 'in_shape', 'out_shape',
               'desc', 'K', 'N', 'center')


def build_layers(net: str, layers: Sequence[tuple], in_shape: Tuple[int, ...],
                 layout: ParamLayout) -> Tuple[List[LayerRec], Tuple[int, ...]]:
  """Resolve shapes / SAME pads and register parameters.  Returns (records, out_shape).
  Layer tuples: ('center',) ('conv',co,k,s,act) ('deconv',co,k,s,act) ('flatten',)
  ('dense',units,act) ('reshape',(h,w,c)) -- cf. odin/networks/image_networks.py."""
  recs: List[LayerRec] = []
  shp = tuple(in_shape)
  center = False
  for li, L in enumerate(layers):
    kind = L[0]
    if kind == 'center':
      center = True
      continue
    if kind == 'flatten':
      shp = (int(np.prod(shp)),)
      continue
    if kind == 'reshape':
      assert int(np.prod(L[1])) == int(np.prod(shp))
      shp = tuple(L[1])
      continue
    r = LayerRec()
    r.kind, r.key, r.center = kind, (net, li), center
    r.in_shape = shp
    r.desc = None
    r.K = r.N = 0
    if kind == 'conv':
      _, co, k, s, act = L
      H, W, Ci = shp
      OH, pt, _ = same_pads(H, k, s)
      OW, pl, _ = same_pads(W, k, s)
      r.act = act
      r.out_shape = (OH, OW, co)
      r.desc = dict(H=H, W=W, Cin=Ci, OH=OH, OW=OW, Cout=co, K=k, stride=s, pad_t=pt, pad_l=pl)
      wshape = (k, k, Ci, co)
    elif kind == 'deconv':
      _, co, k, s, act = L
      H, W, Ci = shp
      OH, OW = H * s, W * s
      _, pt, _ = same_pads(OH, k, s)
      _, pl, _ = same_pads(OW, k, s)
      r.act = act
      r.out_shape = (OH, OW, co)
      r.desc = dict(H=H, W=W, Cin=Ci, OH=OH, OW=OW, Cout=co, K=k, stride=s, pad_t=pt, pad_l=pl)
      wshape = (k, k, co, Ci)
      assert not center
    elif kind == 'dense':
      _, u, act = L
      assert len(shp) == 1, 'dense needs a flat input (add a flatten layer)'
      r.act, r.K, r.N = act, shp[0], u
      r.out_shape = (u,)
      wshape = (shp[0], u)
      assert not center
    else:
      raise ValueError(f'unknown layer kind {kind!r}')
    r.w_n = int(np.prod(wshape))
    r.b_n = wshape[-1] if kind != 'deconv' else wshape[2]
    r.w_off = layout.add((net, li, 'w'), wshape)
    r.b_off = layout.add((net, li, 'b'), (r.b_n,))
    center = False
    shp = r.out_shape
    recs.append(r)
  return recs, shp


# --------------------------------------------------------------------------------------
# one sequential network bound to buffers for a fixed batch size
# --------------------------------------------------------------------------------------
RANGE_WORDS = 2048  # include/odin_hip.h: ODIN_RANGE_WORDS (uint32 per range word)


class NetProgram:

  def __init__(self, lib, recs: List[LayerRec], B: int, device, params: torch.Tensor,
               grads: torch.Tensor, max_rows: int, range_words: Optional[torch.Tensor] = None,
               act_words: Optional[torch.Tensor] = None, use_act_words: bool = True, small_wgrad_gf: float = 0.8,
               direct_wgrad: bool = False):
    self.lib, self.recs, self.B, self.device = lib, recs, B, device
    self.params, self.grads = params, grads
    f32 = dict(dtype=torch.float32, device=device)
    # one range word per outs[i] too (round 5, include/odin_hip.h: odin_conv_desc.x_amax / y_amax): max |activation|,
    # kept by the layer that produces the tensor when the layer above reads it (a two-plane kernel: forward or weight
    # gradient), so that activations beyond the f16 range -- or far below it -- keep their 22 bits; zeroed with the
    # gradient words
    self.act_words = (torch.zeros(len(recs) * RANGE_WORDS, dtype=torch.int32, device=device)
                      if act_words is None else act_words)
    assert self.act_words.numel() == len(recs) * RANGE_WORDS and self.act_words.is_contiguous()
    # one range word per gouts[i] (include/odin_hip.h: odin_conv_desc.dy_amax / dx_amax): max |gradient|, kept by the
    # kernel that produces the tensor and read by the f16-plane kernels that consume it; zeroed once per step
    self.range_words = (torch.zeros(len(recs) * RANGE_WORDS, dtype=torch.int32, device=device)
                        if range_words is None else range_words)
    assert self.range_words.numel() == len(recs) * RANGE_WORDS and self.range_words.is_contiguous()
    self.outs = [torch.empty((B,) + r.out_shape, **f32) for r in recs]
    # gradient wrt the PRE-activation output of every layer
    self.gouts = [torch.empty((B,) + r.out_shape, **f32) for r in recs]
    self.descs = []
    for r in recs:
      if r.desc is not None:
        d = r.desc
        cd = _lib.conv_desc(B, d['H'], d['W'], d['Cin'], d['OH'], d['OW'], d['Cout'],
                            d['K'], d['stride'], d['pad_t'], d['pad_l'], r.act, r.center)
        i = len(self.descs)
        cd.dx_amax = (self.range_words.data_ptr() + 4 * RANGE_WORDS * (i - 1)) if i > 0 else None
        self.descs.append(cd)
      else:
        self.descs.append(None)
    # gouts[i] travels with its word whenever a library data gradient produced it: the data gradient of layer i + 1 is
    # handed word(i) as dx_amax and leaves a valid bound there whatever kernel family ran (include/odin_hip.h: the
    # range contract of round 5; round 4 relied on the *_keeps_range predicates agreeing with the dispatch -- they
    # did not for a column-sum slab on a large layer, and a zero word overflowed the consumers).  The top gradient
    # (ELBO kernel / fused tail / head / latent block) is the caller's: set_top_word
    self.dy_word = [self.word(i) if i + 1 < len(recs) else None for i in range(len(recs))]
    self.dx_word = [self.word(i - 1) if i > 0 else None for i in range(len(recs))]   # (Dense layers: odin_dense_bwd)
    for i, cd in enumerate(self.descs):
      if cd is not None:
        cd.dy_amax = self.dy_word[i]
    # activation words: layer i reads the word of its input (outs[i - 1]) if one of its launches is a plane kernel;
    # layer i - 1 is asked to keep that word only then
    self.reads_x = [bool(use_act_words) and self._reads_x(i) for i in range(len(recs))]
    self.x_word: List[Optional[int]] = [None] * len(recs)
    self.y_word: List[Optional[int]] = [None] * len(recs)
    for i in range(1, len(recs)):
      if self.reads_x[i]:
        self.x_word[i] = self.y_word[i - 1] = self.aword(i - 1)
    for i, cd in enumerate(self.descs):
      if cd is not None:
        cd.x_amax, cd.y_amax = self.x_word[i], self.y_word[i]
    self.wslabs: List[Optional[torch.Tensor]] = [None] * len(recs)
    # `direct_wgrad`: a Dense weight gradient that arrives as ONE complete slab row is written straight into the flat
    # gradient buffer ([W | b] is contiguous there): no slab, no reduction job (FactorVAE's discriminator: four
    # 1000 x 1000 layers -- 16 MB copied by a 25 us reduction launch per iteration otherwise)
    self.direct_wgrad = bool(direct_wgrad)
    self.wdirect = [False] * len(recs)
    self.wrows = [0] * len(recs)
    self.bslabs: List[Optional[torch.Tensor]] = [None] * len(recs)  # deconv bias (colsum)
    self.brows = [0] * len(recs)
    # layers whose weight gradient is a small launch (few workgroups, long per-workgroup
    # latency): these can run beside the data-gradient chain on a side stream
    self.small_wgrad = []
    for r in recs:
      if r.desc is not None:
        d = r.desc
        mac = B * (d['OH'] * d['OW'] if r.kind == 'conv' else d['H'] * d['W']) * d['K'] ** 2 * \
            d['Cin'] * d['Cout']
      else:
        mac = B * r.K * r.N
      self.small_wgrad.append(2 * mac < float(small_wgrad_gf) * 1e9)
    self._plan_slabs()

  def word(self, i: int) -> int:
    """device address of the range word of gouts[i]"""
    return self.range_words.data_ptr() + 4 * RANGE_WORDS * i

  def aword(self, i: int) -> int:
    """device address of the range word of outs[i]"""
    return self.act_words.data_ptr() + 4 * RANGE_WORDS * i

  def _reads_x(self, i: int) -> bool:
    r = self.recs[i]
    if self.descs[i] is None:
      return bool(self.lib.odin_dense_reads_x_range(self.B, r.K, r.N))
    fn = self.lib.odin_conv2d_reads_x_range if r.kind == 'conv' else self.lib.odin_deconv2d_reads_x_range
    return bool(fn(C.byref(self.descs[i])))

  def set_x_word(self, i: int, valid: bool) -> None:
    """the producer of outs[i - 1] is outside this program's forward loop (the bottleneck launch writes outs[0]
    without a word): layer i then reads no word (unscaled, as without the side channel)"""
    w = self.aword(i - 1) if (valid and self.reads_x[i]) else None
    self.x_word[i] = w
    if self.descs[i] is not None:
      self.descs[i].x_amax = w

  def set_top_word(self, kept: bool, n: Optional[int] = None) -> Optional[int]:
    """the producer of gouts[n] (outside this program; n = the last layer by default) says whether it keeps the
    tensor's range word; -> the word"""
    n = len(self.recs) - 1 if n is None else n
    self.dy_word[n] = self.word(n) if kept else None
    if self.descs[n] is not None:
      self.descs[n].dy_amax = self.dy_word[n]
    return self.dy_word[n]

  def check_range_words(self, upto: Optional[int] = None, first: int = 0) -> None:
    """Debug / tests (synchronises): every word handed to a consumer bounds its tensor -- call between backward() and
    the slab reduction that clears the words (NetProgram.backward alone does not clear them)."""
    n = len(self.recs) if upto is None else upto
    for i in range(first, n):   # (gouts below `first` stay inside a fused bottleneck launch: never written)
      if self.dy_word[i] is None or self.dy_word[i] != self.word(i):   # (no word, or the caller's own static bound)
        continue
      blk = self.range_words[RANGE_WORDS * i:RANGE_WORDS * (i + 1)]
      bound = float(blk.view(torch.float32).max().item())
      t = self.gouts[i]
      m = float(torch.nan_to_num(t, nan=0.0, posinf=0.0, neginf=0.0).abs().max().item())
      assert bound >= m, f'range word of gouts[{i}] ({self.recs[i].kind}): {bound} < max|t| = {m}'

  def dgrad_keeps_range(self, j: int) -> bool:
    """does the data gradient of layer j fold max |gouts[j - 1]| into its range word itself?"""
    if j <= 0 or j >= len(self.recs):
      return False
    if self.descs[j] is None:  # Dense (its column-sum slab, if the previous layer wants one, rules the plane GEMM out)
      return self.bslabs_wanted(j - 1) is False and bool(
          self.lib.odin_dense_dgrad_keeps_range(self.B, self.recs[j].K, self.recs[j].N))
    fn = (self.lib.odin_conv2d_dgrad_keeps_range if self.recs[j].kind == 'conv'
          else self.lib.odin_deconv2d_dgrad_keeps_range)
    return bool(fn(C.byref(self.descs[j]), ACT[self.recs[j - 1].act]))

  def bslabs_wanted(self, i: int) -> bool:
    """does layer i take its bias gradient from the column sums of its successor's data gradient?"""
    return self.recs[i].kind == 'deconv'

  # -- planning (dry runs report how many slab rows each call will write) --------------
  def _plan_slabs(self):
    lib, B = self.lib, self.B
    f32 = dict(dtype=torch.float32, device=self.device)
    rows = C.c_int(0)
    for i, r in enumerate(self.recs):
      d = self.descs[i]
      if r.kind == 'conv':
        lib.odin_conv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
        n = r.w_n + r.b_n
      elif r.kind == 'deconv':
        lib.odin_deconv2d_wgrad(None, None, None, C.byref(rows), C.byref(d), None)
        # (the one-call backward may write more rows than the weight gradient alone: bwd_planes.hip with 64 output
        # channels -- include/odin_hip.h: odin_deconv2d_bwd, dry run)
        r1, r2 = C.c_int(0), C.c_int(0)
        lib.odin_deconv2d_bwd(None, None, None, None, ACT[self.recs[i - 1].act] if i > 0 else 0, None, None, C.byref(r2),
                              None, C.byref(r1), C.byref(d), None)
        rows.value = max(rows.value, r1.value)
        n = r.w_n
      else:
        lib.odin_dense_wgrad(None, None, None, C.byref(rows), B, r.K, r.N, None)
        n = r.w_n + r.b_n
      self.wrows[i] = rows.value
      if (self.direct_wgrad and r.kind == 'dense' and rows.value == 1 and r.b_off == r.w_off + r.w_n and
          self.grads[r.w_off:].data_ptr() % 16 == 0):
        self.wslabs[i] = self.grads[r.w_off:r.w_off + n].view(1, n)
        self.wdirect[i] = True
        continue
      self.wslabs[i] = torch.empty((rows.value, n), **f32)
      if r.kind == 'deconv':
        # bias gradient = column sums of gouts[i], emitted by whoever produces gouts[i]:
        # the data-gradient of layer i+1
        assert i + 1 < len(self.recs), 'a network may not END with a Conv2DTranspose'
        self.bslabs[i] = torch.empty((lib.odin_max_slab_rows(), r.b_n), **f32)

  def w(self, i):
    r = self.recs[i]
    return self.params[r.w_off:r.w_off + r.w_n]

  def b(self, i):
    r = self.recs[i]
    return self.params[r.b_off:r.b_off + r.b_n]

  # -- forward --------------------------------------------------------------------------
  def forward(self, x: torch.Tensor, st, upto: Optional[int] = None, start: int = 0):
    """run layers [start, upto) (to the end when None) on x = the input of layer `start`; returns the last
    output (or x)."""
    lib, B = self.lib, self.B
    h = x
    for i in range(start, len(self.recs) if upto is None else upto):
      r = self.recs[i]
      y = self.outs[i]
      if r.kind == 'conv':
        lib.odin_conv2d_fwd(h.data_ptr(), self.w(i).data_ptr(), self.b(i).data_ptr(),
                            y.data_ptr(), C.byref(self.descs[i]), st)
      elif r.kind == 'deconv':
        lib.odin_deconv2d_fwd(h.data_ptr(), self.w(i).data_ptr(), self.b(i).data_ptr(),
                              y.data_ptr(), C.byref(self.descs[i]), st)
      else:
        lib.odin_dense_fwd_ranged(h.data_ptr(), self.w(i).data_ptr(), self.b(i).data_ptr(),
                                  y.data_ptr(), B, r.K, r.N, ACT[r.act], self.x_word[i], self.y_word[i], st)
      h = y
    return h

  # -- backward -------------------------------------------------------------------------
  def backward(self, x: torch.Tensor, gout_last: torch.Tensor, st,
               dx_out: Optional[torch.Tensor] = None, last: Optional[int] = None,
               skip_bias_of_last: bool = False, data_only: bool = False,
               fork=None, side_jobs: Optional[list] = None, first: int = 0) -> List[ReduceJob]:
    """gout_last: dL/d(pre-activation output of the last layer).  If dx_out is given the
    gradient wrt the network input is written there.  `first` > 0 stops above layer `first`: its data
    gradient still lands in gouts[first - 1], layers below are the caller's.  Returns the slab-reduce jobs."""
    lib, B = self.lib, self.B
    n = len(self.recs) if last is None else last + 1
    jobs: List[ReduceJob] = []
    g = gout_last
    rows = C.c_int(0)
    for i in range(n - 1, first - 1, -1):
      r, d = self.recs[i], self.descs[i]
      xin = x if i == 0 else self.outs[i - 1]
      # ---- weight (and bias) gradient: independent of the data-gradient chain, so it is
      # issued on the side stream (fork) and overlaps the next layers' data-gradients ----
      slab = self.wslabs[i]
      wst = st if (fork is None or not fork.wants(self.small_wgrad[i])) else fork(i)
      # ---- data gradient -> pre-activation gradient of the previous layer ----
      if i > 0:
        prev = self.recs[i - 1]
        dst, aux, aux_act = self.gouts[i - 1], self.outs[i - 1], ACT[prev.act]
        bslab = self.bslabs[i - 1]
      elif dx_out is not None:
        dst, aux, aux_act, bslab = dx_out, None, 0, None
      else:
        dst = None
      auxp = aux.data_ptr() if (dst is not None and aux is not None and aux_act != 0) else None
      bsp = bslab.data_ptr() if (dst is not None and bslab is not None) else None
      # both halves on one stream: ONE call -- where both run on the small-layer implicit-GEMM kernels they share
      # a launch (include/odin_hip.h: odin_conv2d_bwd)
      both = (not data_only) and dst is not None and wst is st
      wrows = C.c_int(0)
      if both:
        if r.kind == 'conv':
          lib.odin_conv2d_bwd(xin.data_ptr(), g.data_ptr(), self.w(i).data_ptr(), auxp, aux_act, dst.data_ptr(),
                              bsp, C.byref(rows), slab.data_ptr(), C.byref(wrows), C.byref(d), st)
        elif r.kind == 'deconv':
          lib.odin_deconv2d_bwd(xin.data_ptr(), g.data_ptr(), self.w(i).data_ptr(), auxp, aux_act, dst.data_ptr(),
                                bsp, C.byref(rows), slab.data_ptr(), C.byref(wrows), C.byref(d), st)
        else:
          lib.odin_dense_bwd_ranged(xin.data_ptr(), g.data_ptr(), self.w(i).data_ptr(), auxp, aux_act, dst.data_ptr(),
                                    bsp, C.byref(rows), slab.data_ptr(), C.byref(wrows), B, r.K, r.N, 1, 1,
                                    self.dy_word[i], self.dx_word[i], self.x_word[i], st)
      elif data_only:
        pass
      elif r.kind == 'conv':
        lib.odin_conv2d_wgrad(xin.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(wrows),
                              C.byref(d), wst)
      elif r.kind == 'deconv':
        lib.odin_deconv2d_wgrad(xin.data_ptr(), g.data_ptr(), slab.data_ptr(), C.byref(wrows),
                                C.byref(d), wst)
      else:
        lib.odin_dense_bwd_ranged(xin.data_ptr(), g.data_ptr(), None, None, 0, None, None, None, slab.data_ptr(),
                                  C.byref(wrows), B, r.K, r.N, 1, 0, self.dy_word[i], None, self.x_word[i], wst)
      if not data_only:
        assert 0 < wrows.value <= self.wrows[i]
        n_red = slab.shape[1]
        if skip_bias_of_last and i == n - 1 and r.kind != 'deconv':
          n_red = r.w_n  # the fused tail already delivers this layer's bias gradient
        # (jobs whose slab was written on a side stream are kept apart: only the final reduction,
        # after the join, may read them)
        tgt = side_jobs if (side_jobs is not None and wst is not st) else jobs
        if self.wdirect[i]:
          assert wrows.value == 1   # (the launch wrote the gradient itself)
        else:
          tgt.append(ReduceJob(slab.data_ptr(), self.grads[r.w_off:].data_ptr(), n_red, wrows.value,
                               slab.shape[1], 0))
      if dst is None:
        break
      if not both:
        if r.kind == 'conv':
          lib.odin_conv2d_dgrad(g.data_ptr(), self.w(i).data_ptr(), auxp, aux_act, dst.data_ptr(),
                                bsp, C.byref(rows), C.byref(d), st)
        elif r.kind == 'deconv':
          lib.odin_deconv2d_dgrad(g.data_ptr(), self.w(i).data_ptr(), auxp, aux_act,
                                  dst.data_ptr(), bsp, C.byref(rows), C.byref(d), st)
        else:
          lib.odin_dense_bwd(None, g.data_ptr(), self.w(i).data_ptr(), auxp, aux_act, dst.data_ptr(), bsp,
                             C.byref(rows), None, None, B, r.K, r.N, 0, 1, self.dy_word[i], self.dx_word[i], st)
      if bslab is not None and not data_only:
        pr = self.recs[i - 1]
        jobs.append(ReduceJob(bslab.data_ptr(), self.grads[pr.b_off:].data_ptr(), pr.b_n,
                              rows.value, pr.b_n, 0))
      g = dst
    return jobs


# --------------------------------------------------------------------------------------
# the whole VAE step
# --------------------------------------------------------------------------------------
# `softplus1` argument of odin_elbo_gaussian_fwd_bwd per two-parameter observation
OBS_MODE = {'gaussian': 0, 'gaussian_softplus1': 1, 'qlogistic': 2}
MIXQL_K = 10  # components of the 'mixqlogistic' observation (image_networks.py:48)


def observation_maps(observation: str, C: int) -> int:
  """parameter maps the decoder emits per pixel for an observation over C channels."""
  if observation == 'bernoulli':
    return C
  if observation == 'mixqlogistic':
    return MIXQL_K * (2 * C + C * (C - 1) // 2 + 1)
  return 2 * C
H_ALPHA, H_B1, H_B2, H_EPS, H_GSCALE, H_INVB, H_KLW, H_BETA, H_TCCOEF, H_TCGRAD = range(10)
H_CAP = 15  # BetaCapacityVAE: the capacity C(step) (slots 10..14: the second optimiser's Adam block)
N_HYPER = 16


class VAEEngine:
  """encoder -> q(z|x) -> decoder -> ELBO -> backward -> Adam for a FIXED batch size.

  observation: 'bernoulli' | 'gaussian' | 'gaussian_softplus1' | 'qlogistic' | 'mixqlogistic'
  tc: None | 'betatc' (total_correlation, weight (beta-1))
  """

  def __init__(self, enc_layers, dec_layers, in_shape, zdim, batch_size, device,
               observation='bernoulli', analytic=False, free_bits=None, tc=None, lib=None,
               params: Optional[torch.Tensor] = None, world_size: int = 1, seed: int = 1,
               optim_state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               force_dp: bool = False, reverse: bool = True, capacity: bool = False,
               range_words: Optional[torch.Tensor] = None, *,
               act_words: bool = True, hyper_ring: bool = True, hyper_ring_rows: int = 128, fuse_norm: bool = True,
               overlap_wgrad: Optional[str] = None, early_reduce: bool = False, defer_wgrad: bool = False,
               side_streams: int = 2, small_wgrad_gf: float = 0.8, dp_buckets: Optional[int] = None,
               neck: bool = True, neck_bwd: Optional[bool] = None, static_top_word: bool = True):
    """The keyword-only arguments are the engine's launch-order / A-B options (tests and tools pass them; the engine
    reads no environment variable):
      act_words        activation range words for the two-plane consumers; False: unscaled planes
      hyper_ring       per-step scalars from the device-resident ring; False: one 80-byte copy per step
      hyper_ring_rows  rows of that ring (a power of two >= 8)
      fuse_norm        the gradient norm's stage-1 sums ride in the slab reduction (odin_slab_reduce_sumsq)
      static_top_word  the Bernoulli ELBO kernel's gradient travels with its a-priori bound 1 / B as a static range word; False:
                       one absmax pass per step bounds it (A/B)
      overlap_wgrad    None | 'small': the bottleneck layers' weight gradients on side streams (slower since round 2,
                       profiles/r05_ab_same_call.txt; kept for the multi-bucket DP step's tests)
      early_reduce     with overlap_wgrad: the decoder's slabs reduced on a side stream beside the encoder's backward
      defer_wgrad      the plane weight gradients of a step as ONE launch at the end of the backward pass (slower)
      side_streams     side streams the overlaps rotate over
      small_wgrad_gf   GFLOP below which a weight gradient counts as 'small' for overlap_wgrad
      dp_buckets       gradient buckets of the data-parallel step (default: 2 from 4 ranks and 8 MB up, else 1)
      neck             the encoder's last convolution + projection + latent block + the decoder's projection + first
                       Conv2DTranspose as ONE launch per direction where the shapes allow (neck.hip); False: round 5's
                       four launches per direction
      neck_bwd         the neck's backward launch (None: where it was measured faster)"""
    self.lib = lib if lib is not None else _lib.load()
    self.device = torch.device(device)
    self.B, self.D = int(batch_size), int(zdim)
    # BetaCapacityVAE (beta_vae.py:132-177): the KL kernels emit |kl - C(step)| and its sign; `beta` plays gamma
    self.capacity_on = bool(capacity)
    self.in_shape = tuple(in_shape)
    # KL form handed to the latent kernels: 0 = Monte-Carlo, 1 = closed-form KL(q||p),
    # 2 = closed-form KL(p||q) (`reverse=False`, odin/bay/helpers.py:261-265)
    self.observation = observation
    self.set_kl_form(analytic, reverse)
    self.free_bits = -1.0 if free_bits is None else float(free_bits)
    self.tc_mode, self.world_size, self.seed = tc, int(world_size), int(seed)
    # data-parallel step (gradient-bucket all-reduce between backward and Adam); `force_dp` runs
    # that path at world size 1 too (RCCL on a 1-GPU box)
    self.is_dp = self.world_size > 1 or bool(force_dp)
    # beta-TC: the global-batch estimator (all-gather | shard kernel | reduce-scatter) whenever the step is
    # data parallel -- also at world size 1 under `force_dp`, which is how a 1-GPU box exercises it
    self.tc_sharded = tc == 'betatc' and self.is_dp
    f32 = dict(dtype=torch.float32, device=self.device)
    # ---- parameters ----
    self.layout = ParamLayout()
    self.enc_recs, eo = build_layers('enc', enc_layers, self.in_shape, self.layout)
    assert len(eo) == 1, 'encoder must end with a flat output'
    self.hdim = eo[0]
    self.lat_w_off = self.layout.add(('lat', 'w'), (self.hdim, 2 * self.D))
    self.lat_b_off = self.layout.add(('lat', 'b'), (2 * self.D,))
    self.dec_recs, do = build_layers('dec', dec_layers, (self.D,), self.layout)
    self.layout.pad_to(4)
    self.out_shape = do
    C_in = self.in_shape[-1]
    assert tuple(do) == self.in_shape[:-1] + (observation_maps(observation, C_in),), (do, self.in_shape)
    n = self.layout.size
    self.params = params if params is not None else torch.zeros(n, **f32)
    assert self.params.numel() == n
    self.grads = torch.zeros(n, **f32)
    if optim_state is not None:
      self.m, self.v = optim_state
    else:
      self.m = torch.zeros(n, **f32)
      self.v = torch.zeros(n, **f32)
    self.n_params = sum(int(np.prod(s)) for _, s, _ in self.layout.entries)
    # ---- programs / buffers ----
    B, D = self.B, self.D
    mr = self.lib.odin_max_slab_rows()
    ne = len(self.enc_recs)
    nl = ne + len(self.dec_recs)
    # (gradient words | activation words: one buffer, cleared together by the step's last backward launch)
    # `range_words`: another engine's buffer (same networks): a forward-only engine that runs between that engine's
    # forward and backward passes (FactorVAE's second half batch) folds its maxima into the same words -- still upper
    # bounds -- and needs no clearing launch of its own
    self._shared_ranges = range_words is not None
    self.range_words = (range_words if range_words is not None else
                        torch.zeros(2 * nl * RANGE_WORDS, dtype=torch.int32, device=self.device))
    assert self.range_words.numel() == 2 * nl * RANGE_WORDS
    gw, aw = self.range_words[:nl * RANGE_WORDS], self.range_words[nl * RANGE_WORDS:]
    self.enc = NetProgram(self.lib, self.enc_recs, B, self.device, self.params, self.grads, mr,
                          range_words=gw[:ne * RANGE_WORDS], act_words=aw[:ne * RANGE_WORDS],
                          use_act_words=act_words, small_wgrad_gf=small_wgrad_gf)
    self.dec = NetProgram(self.lib, self.dec_recs, B, self.device, self.params, self.grads, mr,
                          range_words=gw[ne * RANGE_WORDS:], act_words=aw[ne * RANGE_WORDS:],
                          use_act_words=act_words, small_wgrad_gf=small_wgrad_gf)
    self._act_words_dirty = False   # a forward pass has written activation words that no backward pass has cleared
    # The top gradient of a Bernoulli observation that is NOT produced by a fused tail comes from the stand-alone ELBO kernel
    # (dlogits = (sigmoid(l) - x) / B_global): its bound is known a priori, |dlogits| <= 1 / B_global, so the last layer's
    # plane kernels read a STATIC range word holding that bound instead of paying one absmax pass per step for it (the dense
    # MNIST step: 5 of 125 us; a bound within a factor of two of the maximum costs at most one of the 22 bits).  The word is
    # outside the buffer the step clears.
    self._bern_word = None
    if observation == 'bernoulli' and static_top_word:
      self._bern_word = torch.zeros(RANGE_WORDS, dtype=torch.int32, device=self.device)
      self._bern_word[:1] = torch.tensor([1.0 / (B * self.world_size)], dtype=torch.float32).view(torch.int32).to(self.device)
      self.dec.dy_word[-1] = self._bern_word.data_ptr()
      if self.dec.descs[-1] is not None:
        self.dec.descs[-1].dy_amax = self._bern_word.data_ptr()
    self.p = torch.empty(B, 2 * D, **f32)
    self.dp = torch.empty(B, 2 * D, **f32)
    self.eps = torch.zeros(B, D, **f32)
    self.z = torch.empty(B, D, **f32)
    self.dz = torch.empty(B, D, **f32)
    self.kl = torch.empty(B, **f32)
    self.fbmask = torch.empty(B, **f32)
    self.dh_e = torch.empty(B, self.hdim, **f32)
    n_per = int(np.prod(self.in_shape))
    self.n_per = n_per
    # per-sample partial log-likelihoods: the kernel reports how many it writes per sample
    npart = C.c_int(0)
    if observation == 'bernoulli':
      self.lib.odin_elbo_bernoulli_fwd_bwd(None, None, None, None, None, B, n_per, C.byref(npart),
                                           None)
    elif observation == 'mixqlogistic':
      Cc = self.in_shape[-1]
      self.lib.odin_elbo_mixqlogistic_fwd_bwd(None, None, None, None, None, B, n_per // Cc, Cc,
                                              MIXQL_K, C.byref(npart), None)
    else:
      Cc = self.in_shape[-1]
      self.lib.odin_elbo_gaussian_fwd_bwd(None, None, None, None, None, B, n_per // Cc, Cc,
                                          OBS_MODE[observation],
                                          C.byref(npart), None)
    self.llk_part = torch.empty(B * max(npart.value, (n_per + 1023) // 1024), **f32)
    self.llk = torch.empty(B, **f32)
    # [loss, mean llk, mean beta*kl, tc | 4 spare words a model may place its own step scalars in (FactorVAE: dtc_loss),
    # so that ONE device-to-device copy snapshots them all]
    self.out8 = torch.zeros(8, **f32)
    self.out4 = self.out8[:4]
    self.n_part = 0
    rows = C.c_int(0)
    self.lib.odin_dense_wgrad(None, None, None, C.byref(rows), B, self.hdim, 2 * D, None)
    self.lat_slab = torch.empty(rows.value, self.hdim * 2 * D + 2 * D, **f32)
    self.lat_rows = rows.value
    if tc == 'betatc':
      self.tc_ws = torch.zeros(self.lib.odin_total_correlation_workspace(
          B, B * self.world_size if self.tc_sharded else B, D), **f32)
      self.tc_dz = torch.empty(B, D, **f32)
      self.tc_dloc = torch.empty(B, D, **f32)
      self.tc_dscale = torch.empty(B, D, **f32)
      if self.tc_sharded:
        # global-batch estimator under data parallelism: all-gather (p | z), this rank's
        # rows against every posterior, reduce-scatter of the posterior-side partial gradients
        Bg = B * self.world_size
        self.tc_pz_local = torch.empty(B, 3 * D, **f32)
        self.tc_pz_all = torch.empty(Bg, 3 * D, **f32)
        self.tc_p_all = torch.empty(Bg, 2 * D, **f32)
        self.tc_part_all = torch.empty(2, Bg, D, **f32)    # dloc / dscale partials for every i
    self._plan_fused_tail(f32)
    self._plan_gauss_head(f32)
    self._plan_latent_block(f32)
    self._neck_bwd_opt = neck_bwd
    self._plan_neck(f32, bool(neck))
    self.ws = torch.empty(4096, **f32)
    self.gnorm2 = torch.zeros(1, **f32)
    self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
    # Networks.optimize gradient policies (skip_update_threshold / clipnorm / clipvalue)
    self.skip_hit = torch.zeros(1, dtype=torch.int32, device=self.device)
    self.skipped_update = torch.zeros(1, dtype=torch.int32, device=self.device)
    offs = [o for _, _, o in self.layout.entries] + [self.n_params_end()]
    self.seg_offsets = torch.tensor(offs, dtype=torch.int64, device=self.device)
    # hyper-parameters that change per step live in device memory (graph replays)
    # ring of pinned staging buffers: the async H2D copy of step t may still be pending
    # when the host prepares step t+1
    self._ring = [torch.zeros(N_HYPER + 4, dtype=torch.float32) for _ in range(8)]
    self._ring_ev = [None] * 8
    if self.device.type == 'cuda':
      self._ring = [t.pin_memory() for t in self._ring]
    self._ring_i = 0
    self.hyper = torch.zeros(N_HYPER + 4, **f32)
    # Device-resident schedule (round 5): the rows of the coming steps in a device ring, loaded into `hyper` by the
    # step's last kernel (include/odin_hip.h: odin_sumsq_adam_ring) -- no per-step host copy while the caller's
    # (lr, beta, ...) follow the prediction (constant values, or the `schedule` callable of train_step); any other
    # call falls back to the explicit 80-byte copy of round 4 for that step (`hyper_ring=False`: always).
    self.ring_rows = int(hyper_ring_rows)   # (a power of two; tests: 16)
    assert self.ring_rows >= 8 and self.ring_rows & (self.ring_rows - 1) == 0
    self.hyper_ring = torch.zeros(self.ring_rows, N_HYPER + 4, **f32)
    self.fuse_norm = bool(fuse_norm)
    self.hyper_staged = torch.zeros(32, **f32)   # (odin_slab_reduce_sumsq stages the whole row: N_HYPER + 4 floats)
    self._ring_host = torch.zeros(self.ring_rows, N_HYPER + 4, dtype=torch.float32)   # host mirror of the ring
    if self.device.type == 'cuda':
      self._ring_host = self._ring_host.pin_memory()
    self._ring_filled_to = -1      # rows of steps <= this are in the device ring (as predicted)
    self._ring_last_t = None       # the step whose Adam advanced `hyper` last (it loaded row _ring_last_t + 1)
    self._ring_live = False        # `hyper` on the device holds (or will hold, by the previous step's Adam) this step's row
    self._ring_args = None         # the caller's arguments the prediction was made from
    self._ring_copy_ev = [None, None]   # events of the last two refill copies (the pinned mirror is reused)
    self.use_hyper_ring = bool(hyper_ring)
    self.step_count = 0
    self.side_stream = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None
    n_side = int(side_streams)
    self.side_streams = ([self.side_stream] + [torch.cuda.Stream(self.device) for _ in range(n_side - 1)]
                         if self.side_stream is not None else [])
    # weight gradients on a side stream ('small': only the launches that leave most CUs idle -- the bottleneck layers
    # -- run beside the data-gradient chain; the 'all' form of rounds 1-5 lost every A/B since round 2 and is gone).
    # OFF by default: it paid at 1.1 ms per step (+2-3 %), but each cross-stream dependency of the captured graph costs
    # 5-18 us of idle time on the main chain (profiles/r02_step_timeline.txt, r05_ab_same_call.txt: 0.547 vs 0.484 ms)
    assert overlap_wgrad in (None, 'small'), overlap_wgrad
    self.overlap_wgrad = overlap_wgrad
    self.early_reduce = bool(early_reduce)
    # the five plane weight gradients of a step as ONE multi-layer launch at the end of the backward pass
    # (odin_wgrad_planes_defer_begin / _end): OFF -- same-box A/B: dSprites 0.491 ms with it, 0.484 without
    # (r05_ab_same_call.txt).  Four launch floors are saved, but issued right behind the data gradient that produced its
    # dy a weight gradient still finds that tensor in the Infinity Cache; at the end of the pass it comes from HBM
    self.defer_wgrad = bool(defer_wgrad)
    self.graph = None
    # tests / debugging: verify every range word against its tensor before the slab reduction clears the words
    # (synchronises; NetProgram.check_range_words)
    self.debug_check_ranges = False
    self._jobs_keepalive = None
    self._jobs_cover_ok: Dict[tuple, bool] = {}
    # data parallel: collectives through `dist.Comm` (RCCL via the C ABI on a GPU); gradient buckets:
    # 2 = the decoder's share of the flat gradient buffer is all-reduced on a side stream while the
    # encoder's backward pass runs.  Splitting the step costs by itself (three graph segments, two slab
    # reductions, no fused bottleneck backward): +84 us on the dSprites step (1.5 MB bucket), +23 us on the CelebA
    # step (9.6 MB) at world size 1 (profiles/r03_dp_buckets.txt) -- more than the all-reduce of a small bucket
    # takes.  Default: two buckets only from 4 ranks up AND from 8 MB of gradients; `dp_buckets` overrides.
    self.comm = None
    big = self.grads.numel() * 4 >= (8 << 20)
    self.dp_buckets = int(dp_buckets or 0) or (2 if (self.world_size >= 4 and big) else 1)
    self.dec_start = min(o for k, _, o in self.layout.entries if k[0] == 'dec')

  def _comm(self):
    if self.comm is None:
      from .dist import Comm
      self.comm = Comm(self.lib, self.device)
    return self.comm

  def set_kl_form(self, analytic, reverse=True):
    if not reverse and not analytic:
      # the reference cannot evaluate this combination either: after swapping the arguments it
      # calls tf.convert_to_tensor on the prior (helpers.py:267-276 with q_sample=None)
      raise TypeError('reverse=False needs analytic=True: the Monte-Carlo form would draw the '
                      'cached sample from the prior, which has none')
    self.analytic = 2 if not reverse else int(bool(analytic))

  def n_params_end(self) -> int:
    k, shp, off = self.layout.entries[-1]
    return off + int(np.prod(shp))

  def _plan_fused_tail(self, f32):
    """Training-step fusion layer[-2](act) -> Conv2D 1x1 -> Bernoulli log-prob + backward
    (odin_bernoulli_tail_fwd_bwd) when the decoder ends that way."""
    self.fused_tail = False
    self.tail_mode = None   # None: Bernoulli (odin_bernoulli_tail_fwd_bwd); 0 / 1: Normal with a raw / softplus1 scale
    recs = self.dec_recs
    if len(recs) >= 2 and self.observation in ('gaussian', 'gaussian_softplus1'):
      self._plan_gauss_tail(f32)
      return
    if self.observation != 'bernoulli' or len(recs) < 2:
      return
    a, b = recs[-2], recs[-1]
    if not (b.kind == 'conv' and b.desc['K'] == 1 and b.desc['stride'] == 1 and b.act == 'linear'
            and b.desc['Cout'] <= 4 and a.kind in ('conv', 'deconv') and a.desc['Cout'] <= 32):
      return
    rows, npart = C.c_int(0), C.c_int(0)
    try:
      self.lib.odin_bernoulli_tail_fwd_bwd(int(a.kind == 'deconv'), None, None, None, None, None,
                                           None, None, None, None, C.byref(npart), None,
                                           C.byref(rows), None, C.byref(self.dec.descs[-2]),
                                           b.desc['Cout'], None)
    except _lib.OdinError:
      return
    # a dry run cannot see the one-image-per-tile requirement; it holds when the layer's
    # output image has more pixels than one tile (128)
    if a.desc['OH'] * a.desc['OW'] <= 128:
      return
    self.fused_tail = True
    self.tail_keeps_range = bool(self.lib.odin_bernoulli_tail_keeps_range(
        int(a.kind == 'deconv'), C.byref(self.dec.descs[-2]), b.desc['Cout']))
    self.tail_rows, self.tail_npart = rows.value, npart.value
    co, c1 = a.desc['Cout'], b.desc['Cout']
    self.tail_slab = torch.empty(rows.value, co * c1 + c1 + co, **f32)
    self.tail_llk_part = torch.empty(self.B * npart.value, **f32)

  def _plan_gauss_tail(self, f32):
    """Training-step fusion Conv2DTranspose(k4, s2, 32 -> 32) -> Conv2D 1x1 (loc | scale) -> Normal log-prob + backward
    in ONE launch (odin_gaussian_tail_fwd_bwd, blk_planes.hip): the audio VAE's decoder4 -> decoder6 -> observation
    (examples/vae/vae_audio.py:84-110) without the 32-channel activation ever reaching HBM."""
    a, b = self.dec_recs[-2], self.dec_recs[-1]
    Cc = self.in_shape[-1]
    if not (b.kind == 'conv' and b.desc['K'] == 1 and b.desc['stride'] == 1 and b.act == 'linear'
            and b.desc['Cout'] == 2 * Cc and a.kind == 'deconv'):
      return
    d = self.dec.descs[-2]
    if not self.lib.odin_gaussian_tail_applicable(C.byref(d), Cc):
      return
    rows, npart = C.c_int(0), C.c_int(0)
    self.lib.odin_gaussian_tail_fwd_bwd(None, None, None, None, None, None, None, None, None, C.byref(npart), None,
                                        C.byref(rows), None, C.byref(d), Cc, OBS_MODE[self.observation], None)
    self.fused_tail = True
    self.tail_mode = OBS_MODE[self.observation]
    self.tail_keeps_range = True
    self.tail_rows, self.tail_npart = rows.value, npart.value
    co, c1 = a.desc['Cout'], b.desc['Cout']
    self.tail_slab = torch.empty(rows.value, co * c1 + c1 + co, **f32)
    self.tail_llk_part = torch.empty(self.B * npart.value, **f32)

  def _plan_gauss_head(self, f32):
    """Training-step fusion Conv2D 1x1 -> Normal log-prob + backward (odin_gaussian_head_fwd_bwd) when the decoder
    ends in the 1x1 head of a Gaussian observation (the audio VAE, examples/vae/vae_audio.py:84-110): one pass over
    the activation below the head instead of three."""
    self.gauss_head = self._used_head = False
    recs = self.dec_recs
    if self.observation not in ('gaussian', 'gaussian_softplus1', 'bernoulli') or len(recs) < 2:
      return
    a, b = recs[-2], recs[-1]
    Cc = self.in_shape[-1]
    bern = self.observation == 'bernoulli'
    if not (b.kind == 'conv' and b.desc['K'] == 1 and b.desc['stride'] == 1 and b.act == 'linear'
            and b.desc['Cout'] == (Cc if bern else 2 * Cc) and a.kind in ('conv', 'deconv')):
      return
    if not bern and self.fused_tail:
      return   # (the Gaussian tail runs inside the layer's own launch: _plan_gauss_tail)
    if bern and self.fused_tail:
      # the Bernoulli decoders whose last two layers run as ONE plane-kernel launch keep it (dSprites, Shapes3D, CelebA);
      # the generic fused tail (fp32 gather kernel) gives way to layer + head where the layer is big enough for the
      # two-plane implicit GEMM (MNIST's 5x5 stack at batch 128: 141 us -> 58 + 10)
      d = a.desc
      mac = self.B * (d['OH'] * d['OW'] if a.kind == 'conv' else d['H'] * d['W']) * d['K'] ** 2 * d['Cin'] * d['Cout']
      if self.tail_keeps_range or 2.0 * mac < 1.2e9:
        return
    # (mode of odin_gaussian_head_fwd_bwd: 0 / 1 = Normal with a raw / softplus1 scale, 3 = Bernoulli)
    self.head_mode = 3 if bern else OBS_MODE[self.observation]
    rows, npart = C.c_int(0), C.c_int(0)
    try:
      self.lib.odin_gaussian_head_fwd_bwd(None, None, None, None, None, None, None, None, C.byref(npart), None,
                                          C.byref(rows), None, None, self.B, self.n_per // Cc, b.desc['Cin'], Cc,
                                          self.head_mode, ACT[a.act], None, None)
    except _lib.OdinError:
      return
    self.gauss_head = True
    self.fused_tail = False
    self.head_rows, self.head_npart = rows.value, npart.value
    cin, co = b.desc['Cin'], b.desc['Cout']
    self.head_slab = torch.empty(rows.value, cin * co + co, **f32)
    self.head_colsum = torch.empty(rows.value, cin, **f32) if a.kind == 'deconv' else None
    self.head_llk_part = torch.empty(self.B * npart.value, **f32)

  def _plan_latent_block(self, f32):
    """Training-step fusion of the bottleneck (latent_block.hip): noise + DistributionDense + reparameterise /
    KL + the decoder's first Dense as one launch, and their five backward launches as one, when the decoder
    starts with a Dense layer and both weight matrices fit in LDS."""
    self.lat_block = self._used_block = False
    recs = self.dec_recs
    if len(recs) < 2 or recs[0].kind != 'dense' or self.enc_recs[-1].kind == 'deconv':
      return
    rows = self.lib.odin_latent_block_rows(self.B, self.hdim, self.D, recs[0].N)
    if rows <= 0:
      return
    self.lat_block, self.lb_rows = True, rows
    D, N0 = self.D, recs[0].N
    self.lb_slab0 = torch.empty(rows, D * N0 + N0, **f32)
    self.lb_slabl = torch.empty(rows, self.hdim * 2 * D + 2 * D, **f32)

  def _plan_neck(self, f32, enabled: bool):
    """Training-step fusion of the whole neck (neck.hip): Conv2D(64, k4, s2) on [8, 8, 64] -> Flatten -> Dense(P) ->
    latent block -> Dense(D -> 16 C0) -> Reshape(4, 4, C0) -> Conv2DTranspose(64, k4, s2) as one launch per direction
    (image_networks.py:466-471, 494-502: the dSprites / Shapes3D stacks)."""
    self.neck = self._used_neck = False
    er, dr = self.enc_recs, self.dec_recs
    if not (enabled and self.lat_block and len(er) >= 3 and len(dr) >= 3):
      return
    c3, d4, d0, t1 = er[-2], er[-1], dr[0], dr[1]
    if not (c3.kind == 'conv' and d4.kind == 'dense' and d0.kind == 'dense' and t1.kind == 'deconv'):
      return
    g = c3.desc
    if not (g['H'] == 8 and g['W'] == 8 and g['Cin'] == 64 and g['Cout'] == 64 and g['K'] == 4 and g['stride'] == 2
            and g['pad_t'] == 1 and g['pad_l'] == 1 and not c3.center and d4.K == 1024):
      return
    t = t1.desc
    if not (t['H'] == 4 and t['W'] == 4 and t['Cout'] == 64 and t['K'] == 4 and t['stride'] == 2 and t['pad_t'] == 1
            and t['pad_l'] == 1 and d0.N == 16 * t['Cin']):
      return
    rows = self.lib.odin_neck_rows(self.B, d4.N, self.D, t['Cin'])
    if rows <= 0:
      return
    if self.params[d4.w_off:].data_ptr() % 16 != 0:   # (the projection's rows are read as 16-byte loads)
      return
    self.neck, self.nk_rows = True, rows
    ne = len(er)
    # the layer below the convolution keeps the activation word of its output (the neck's input planes read it)
    self.enc.y_word[ne - 3] = self.enc.aword(ne - 3)
    if self.enc.descs[ne - 3] is not None:
      self.enc.descs[ne - 3].y_amax = self.enc.y_word[ne - 3]
    A = _lib.NeckArgs()
    A.B, A.P, A.D, A.C0 = self.B, d4.N, self.D, t['Cin']
    A.act2, A.act3, A.act4, A.act0, A.act1 = (ACT[er[-3].act], ACT[c3.act], ACT[d4.act], ACT[d0.act], ACT[t1.act])
    A.seed = self.seed
    A.x, A.x_amax = self.enc.outs[ne - 3].data_ptr(), self.enc.y_word[ne - 3]
    A.w3, A.b3, A.y3 = self.enc.w(ne - 2).data_ptr(), self.enc.b(ne - 2).data_ptr(), self.enc.outs[ne - 2].data_ptr()
    A.w4, A.b4, A.y4 = self.enc.w(ne - 1).data_ptr(), self.enc.b(ne - 1).data_ptr(), self.enc.outs[ne - 1].data_ptr()
    A.wl, A.bl = self.params[self.lat_w_off:].data_ptr(), self.params[self.lat_b_off:].data_ptr()
    A.eps = self.eps.data_ptr()
    A.p, A.kl, A.fbmask = self.p.data_ptr(), self.kl.data_ptr(), self.fbmask.data_ptr()
    A.w0, A.b0, A.y0 = self.dec.w(0).data_ptr(), self.dec.b(0).data_ptr(), self.dec.outs[0].data_ptr()
    A.w1, A.b1, A.y1 = self.dec.w(1).data_ptr(), self.dec.b(1).data_ptr(), self.dec.outs[1].data_ptr()
    self._nk = A
    C0 = t['Cin']
    self.nk_slab1 = torch.empty(rows, 16 * 64 * C0, **f32)
    self.nk_slab0 = torch.empty(rows, self.D * 16 * C0 + 16 * C0, **f32)
    self.nk_slabl = torch.empty(rows, d4.N * 2 * self.D + 2 * self.D, **f32)

  def _neck_fwd(self, eps, st, y1_word: bool = True, enc_only: bool = False):
    """one launch: conv3 .. deconv1 (the caller has run the encoder up to the layer below conv3); `enc_only`: the launch
    stops behind the latent block (run_encoder: the decoder's first two layers are not wanted)"""
    A = self._nk
    A.y1 = None if enc_only else self.dec.outs[1].data_ptr()
    A.analytic, A.free_bits = int(self.analytic), float(self.free_bits)
    A.step_dev = self.hp(N_HYPER)
    A.eps_in = None if eps is None else self.eps.data_ptr()
    A.z = self.z.data_ptr()   # (FactorVAE re-points z into the discriminator's input buffer after construction)
    A.capacity = self.hp(H_CAP) if self.capacity_on else None
    A.y1_amax = self.dec.y_word[1] if y1_word else None
    self.lib.odin_neck_fwd(C.byref(A), st)

  def _dec_first(self) -> int:
    """the decoder layers below this index are back-propagated by the fused bottleneck launches"""
    return 2 if self._bwd_neck() else int(self._bwd_block())

  def _neck_bwd(self, dzx, tl, ts, st, jobs):
    """conv3 .. deconv1 backward: ONE launch for the data-gradient chain and the three small weight gradients, then the
    two large weight gradients (reductions over the whole batch) on their own kernels"""
    lib, A, ne = self.lib, self._nk, len(self.enc_recs)
    c3, d4, d0, t1 = self.enc_recs[-2], self.enc_recs[-1], self.dec_recs[0], self.dec_recs[1]
    A.dy1 = self.dec.gouts[1].data_ptr()
    A.klw = self.hp(H_KLW)
    A.dz_extra, A.dloc_x, A.dscale_x = dzx, tl, ts
    A.dz, A.dp = self.dz.data_ptr(), self.dp.data_ptr()
    A.dh4, A.dy3, A.dx = (self.enc.gouts[ne - 1].data_ptr(), self.enc.gouts[ne - 2].data_ptr(),
                          self.enc.gouts[ne - 3].data_ptr())
    A.dh4_amax, A.dy3_amax, A.dx_amax = self.enc.set_top_word(True), self.enc.word(ne - 2), self.enc.word(ne - 3)
    A.slab1, A.slab0, A.slabl = self.nk_slab1.data_ptr(), self.nk_slab0.data_ptr(), self.nk_slabl.data_ptr()
    lib.odin_neck_bwd(C.byref(A), st)
    rows = self.nk_rows
    for slab, off in ((self.nk_slab1, t1.w_off), (self.nk_slab0, d0.w_off), (self.nk_slabl, self.lat_w_off)):
      jobs.append(ReduceJob(slab.data_ptr(), self.grads[off:].data_ptr(), slab.shape[1], rows, slab.shape[1], 0))
    # conv3's weight gradient (a reduction over all B * 16 pixels) and the projection's (over the batch)
    wrows = C.c_int(0)
    lib.odin_wgrad_pair_begin()   # (independent of each other: one launch where both are small-layer implicit GEMMs)
    slab = self.enc.wslabs[ne - 2]
    lib.odin_conv2d_wgrad(self.enc.outs[ne - 3].data_ptr(), self.enc.gouts[ne - 2].data_ptr(), slab.data_ptr(),
                          C.byref(wrows), C.byref(self.enc.descs[ne - 2]), st)
    jobs.append(ReduceJob(slab.data_ptr(), self.grads[c3.w_off:].data_ptr(), slab.shape[1], wrows.value, slab.shape[1], 0))
    slab = self.enc.wslabs[ne - 1]
    lib.odin_dense_bwd_ranged(self.enc.outs[ne - 2].data_ptr(), self.enc.gouts[ne - 1].data_ptr(), None, None, 0, None,
                              None, None, slab.data_ptr(), C.byref(wrows), self.B, d4.K, d4.N, 1, 0,
                              self.enc.dy_word[ne - 1], None, None, st)
    lib.odin_wgrad_pair_end()
    jobs.append(ReduceJob(slab.data_ptr(), self.grads[d4.w_off:].data_ptr(), slab.shape[1], wrows.value, slab.shape[1], 0))

  def _bwd_neck(self) -> bool:
    if not (self.neck and self._used_neck) or (self.is_dp and self.dp_buckets >= 2):
      return False
    # (measured, same-call A/Bs of round 6: with the 128-wide projection of the dSprites stack the backward launch wins
    # 6 us per step; with the 256-wide one -- 1 MB of W4 streamed twice per workgroup -- round 5's launches are 2-3 us
    # faster: profiles/r06_neck.txt)
    return self._nk.P == 128 if self._neck_bwd_opt is None else bool(self._neck_bwd_opt)

  def _bwd_block(self) -> bool:
    # (two gradient buckets: the decoder's first Dense belongs to the bucket that is already being
    # all-reduced while the encoder's share runs -- keep the separate launches there)
    return self.lat_block and self._used_block and not (self.is_dp and self.dp_buckets >= 2)

  # ---- helpers -----------------------------------------------------------------------
  def hp(self, idx):  # device address of one hyper scalar
    return self.hyper.data_ptr() + 4 * idx

  def stream(self):
    if self.device.type == 'cuda':
      return torch.cuda.current_stream(self.device).cuda_stream
    return None

  def param_views(self) -> Dict[tuple, torch.Tensor]:
    return self.layout.views(self.params)

  def grad_views(self) -> Dict[tuple, torch.Tensor]:
    return self.layout.views(self.grads)

  def load_params(self, P: Dict[tuple, np.ndarray]):
    """P keyed like the oracle: ('enc', li, 'w'|'b'), ('lat', 'w'|'b'), ('dec', li, ...)."""
    views = self.param_views()
    assert set(P.keys()) == set(views.keys()), (sorted(P.keys()), sorted(views.keys()))
    for k, v in P.items():
      views[k].copy_(torch.as_tensor(np.asarray(v), dtype=torch.float32).to(self.device))

  def set_hyper(self, lr=1e-3, beta=1.0, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0,
                t: Optional[int] = None, tc_coef: Optional[float] = None,
                skip_enable: bool = True, extra: Optional[Sequence[float]] = None,
                capacity: Optional[float] = None):
    """Host scalars -> device (one small async H2D copy)."""
    t = self.step_count if t is None else t
    self._ring_live = False   # (an explicit copy: whatever the ring predicted for this step no longer counts)
    self._ring_i = (self._ring_i + 1) % len(self._ring)
    h = self._ring[self._ring_i]
    if self._ring_ev[self._ring_i] is not None:
      self._ring_ev[self._ring_i].synchronize()
    self._fill_row(h, t, lr=lr, beta=beta, b1=b1, b2=b2, eps=eps, grad_scale=grad_scale, tc_coef=tc_coef,
                   skip_enable=skip_enable, extra=extra, capacity=capacity)
    # (leaving this 80-byte copy out of the steady state was measured in round 4 at 0.72 ms per step: no difference;
    # at 0.52 ms it is worth 7-8 us -- profiles/r05_hyper_ring.txt -- hence the device ring of train_step)
    self.hyper.copy_(h, non_blocking=True)
    if self.device.type == 'cuda':
      ev = torch.cuda.Event()
      ev.record(torch.cuda.current_stream(self.device))
      self._ring_ev[self._ring_i] = ev

  def _fill_row(self, h, t, lr=1e-3, beta=1.0, b1=0.9, b2=0.999, eps=1e-7, grad_scale=1.0, tc_coef=None,
                skip_enable=True, extra=None, capacity=None):
    """the hyper-parameter row of step t into the host tensor h (N_HYPER + 4 floats)"""
    tt = max(int(t), 1)
    Bg = self.B * self.world_size
    h[H_ALPHA] = lr * math.sqrt(1.0 - b2 ** tt) / (1.0 - b1 ** tt)
    h[H_B1], h[H_B2], h[H_EPS], h[H_GSCALE] = b1, b2, eps, grad_scale
    h[H_INVB] = 1.0 / Bg
    h[H_KLW] = beta / Bg
    h[H_BETA] = beta
    h[H_TCCOEF] = (beta - 1.0) if self.tc_mode == 'betatc' else 0.0
    # the TC gradient coefficient: with the batch sharded, every rank back-propagates the FULL
    # d(TC_global)/d(its own z, loc, scale) -- the sum all-reduce of the parameter gradients then
    # assembles d/dtheta exactly once per sample
    h[H_TCGRAD] = (beta - 1.0) if self.tc_mode == 'betatc' else 0.0
    if tc_coef is not None:  # FactorVAE: tc term = tc_coef * mean(D(z))
      h[H_TCCOEF] = tc_coef
    if extra is not None:  # second optimiser's Adam block (FactorVAE discriminator), slots 10..14
      for i, val in enumerate(extra):
        h[10 + i] = float(val)
    if self.capacity_on:
      assert capacity is not None, 'this engine was built with capacity=True: pass the capacity C(step)'
      h[H_CAP] = float(capacity)
    h[N_HYPER:].view(torch.int32)[0] = int(t)
    h[N_HYPER:].view(torch.int32)[1] = int(skip_enable)

  def _ring_step(self, t: int, args: dict, schedule=None) -> None:
    """Make sure the device row `hyper` holds step t's scalars when the step runs -- without a copy when the ring
    already predicted exactly this row.  `args`: set_hyper's keyword arguments of this step; `schedule(step) -> dict`
    (optional) overrides them for FUTURE steps (a learning-rate / beta / capacity schedule known in advance)."""
    R = self.ring_rows
    row = self._ring_host[t % R]
    want = torch.zeros(N_HYPER + 4, dtype=torch.float32)
    wa = dict(args)
    if schedule is not None:
      wa.update(schedule(t))   # (a schedule rules the current step too: the rows it filled are the ones that run)
    if 'when_skip_update' in wa:
      wa['skip_enable'] = t >= int(wa.pop('when_skip_update'))
    self._fill_row(want, t, **wa)
    # a hit needs the DEVICE row to hold step t: the previous ring-advanced step must have been t - 1 (its Adam loaded
    # row t).  step_count may jump -- another batch-size engine ran steps in between, a checkpoint was restored -- and
    # the mirror alone cannot see that
    hit = (self._ring_live and self._ring_last_t == t - 1 and self._ring_filled_to >= t and
           torch.equal(want.view(torch.int32), row.view(torch.int32)))
    self._ring_last_t = t
    key = tuple(sorted((k, v if not isinstance(v, (list, tuple)) else tuple(v)) for k, v in args.items()))
    stable = schedule is not None or self._ring_args is None or key == self._ring_args
    self._ring_args = key
    if not hit:
      # the explicit copy of round 4 for this step; the ring is refilled from t + 1 with what is known now -- unless
      # the caller's values change from call to call without a schedule (then every step takes this path and a refill
      # per step would only add a second copy: behave exactly as round 4 until two consecutive calls agree)
      self.set_hyper(t=t, **wa)
      if not stable:
        return
      self._ring_filled_to = t
    # refill: whenever fewer than R / 4 predicted rows are left (or after a miss), rows up to t + R / 2 are written --
    # stream-ordered behind the previous step, whose Adam has consumed every slot that is overwritten
    if self._ring_filled_to - t < R // 4:
      lo, hi = self._ring_filled_to + 1, t + R // 2
      # the pinned mirror rows written below may still be the source of an earlier refill's async copy: wait for
      # every outstanding one (steady state: enqueued ~R / 4 steps ago, long done; after consecutive misses: the
      # previous step's -- the host writes must not race its DMA)
      for old in self._ring_copy_ev:
        if old is not None:
          old.synchronize()
      self._ring_copy_ev.append(None)
      self._ring_copy_ev.pop(0)
      for u in range(lo, hi + 1):
        a = dict(args)
        if schedule is not None:
          a.update(schedule(u))
        if 'when_skip_update' in a:
          a['skip_enable'] = u >= int(a.pop('when_skip_update'))
        self._fill_row(self._ring_host[u % R], u, **a)
      # (a contiguous span of the mirror, or two when it wraps)
      a0, a1 = lo % R, hi % R
      spans = [(a0, a1 + 1)] if a0 <= a1 else [(a0, R), (0, a1 + 1)]
      for x0, x1 in spans:
        self.hyper_ring[x0:x1].copy_(self._ring_host[x0:x1], non_blocking=True)
      if self.device.type == 'cuda':
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._ring_copy_ev[-1] = ev
      self._ring_filled_to = hi
    self._ring_live = True

  def _clear_stale_act_words(self, st):
    """The activation words are cleared by the step's last backward launch; a forward pass that follows another
    forward pass (encode / decode / evaluation) clears them itself -- the words fold maxima in, a stale larger bound
    would cost the planes precision."""
    if self._shared_ranges:
      return
    if self._act_words_dirty:
      nl = len(self.enc_recs) + len(self.dec_recs)
      self.lib.odin_range_reset(self.range_words.data_ptr() + 4 * RANGE_WORDS * nl, nl, st)
    self._act_words_dirty = True

  # ---- partial passes used by the model API (encode / decode) ---------------------------
  def run_encoder(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, st=None):
    """encoder -> Dense(2D) -> (loc, softplus(raw)) -> z = loc + scale*eps; fills p, z, kl."""
    lib, B, D = self.lib, self.B, self.D
    st = self.stream() if st is None else st
    assert x.shape == (B,) + self.in_shape and x.is_contiguous()
    self.x = x
    self._clear_stale_act_words(st)
    lw = self.params[self.lat_w_off:]
    lb = self.params[self.lat_b_off:]
    if self.neck:
      # conv3 .. deconv1 as ONE launch (neck.hip; the decoder's first two layers it also evaluates land in dec.outs[0 / 1],
      # unused here -- and without touching the decoder's activation word)
      if eps is not None and eps is not self.eps:
        self.eps.copy_(eps)
      self.enc.forward(x, st, upto=len(self.enc_recs) - 2)
      self._neck_fwd(eps, st, y1_word=False, enc_only=True)
      return self.p, self.z
    if self.lat_block:
      # noise + projection + reparameterisation + KL as ONE launch (latent_block.hip; the decoder's first Dense it
      # also evaluates lands in dec.outs[0], unused here): three launches less than the separate kernels
      if eps is not None and eps is not self.eps:
        self.eps.copy_(eps)
      h_e = self.enc.forward(x, st)
      r0 = self.dec_recs[0]
      lib.odin_latent_block_fwd(h_e.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                None if eps is None else self.eps.data_ptr(), self.eps.data_ptr(),
                                self.seed, self.hp(N_HYPER), self.p.data_ptr(), self.z.data_ptr(),
                                self.kl.data_ptr(), self.fbmask.data_ptr(), self.dec.w(0).data_ptr(),
                                self.dec.b(0).data_ptr(), self.dec.outs[0].data_ptr(), B, self.hdim, D,
                                r0.N, ACT[r0.act], int(self.analytic), self.free_bits,
                                self.hp(H_CAP) if self.capacity_on else None, st)
      return self.p, self.z
    if eps is None:
      lib.odin_rng_normal(self.eps.data_ptr(), B * D, self.seed, self.hp(N_HYPER), st)
    elif eps is not self.eps:
      self.eps.copy_(eps)
    h_e = self.enc.forward(x, st)
    lib.odin_dense_fwd(h_e.data_ptr(), lw.data_ptr(), lb.data_ptr(), self.p.data_ptr(), B,
                       self.hdim, 2 * D, 0, st)
    lib.odin_latent_fwd(self.p.data_ptr(), self.eps.data_ptr(), self.z.data_ptr(),
                        self.kl.data_ptr(), self.fbmask.data_ptr(), B, D, int(self.analytic),
                        self.free_bits, self.hp(H_CAP) if self.capacity_on else None, st)
    return self.p, self.z

  def run_decoder(self, z: torch.Tensor, st=None):
    st = self.stream() if st is None else st
    assert z.shape == (self.B, self.D) and z.is_contiguous()
    self._clear_stale_act_words(st)
    if len(self.dec_recs) > 1:
      self.dec.set_x_word(1, True)
    return self.dec.forward(z, st)

  def observation_llk(self, h_d: torch.Tensor, x: torch.Tensor, out: torch.Tensor, st=None):
    """out[b] = log p(x_b | decoder output h_d_b) through the fused observation kernel (its
    gradient output lands in the decoder's gradient buffer and is ignored)."""
    lib, B = self.lib, self.B
    st = self.stream() if st is None else st
    npart = C.c_int(0)
    gl = self.dec.gouts[-1]
    if self.observation == 'bernoulli':
      lib.odin_elbo_bernoulli_fwd_bwd(h_d.data_ptr(), x.data_ptr(), self.llk_part.data_ptr(),
                                      gl.data_ptr(), self.hp(H_INVB), B, self.n_per,
                                      C.byref(npart), st)
    elif self.observation == 'mixqlogistic':
      Cc = self.in_shape[-1]
      lib.odin_elbo_mixqlogistic_fwd_bwd(h_d.data_ptr(), x.data_ptr(), self.llk_part.data_ptr(),
                                         gl.data_ptr(), self.hp(H_INVB), B, self.n_per // Cc, Cc,
                                         MIXQL_K, C.byref(npart), st)
    else:
      Cc = self.in_shape[-1]
      lib.odin_elbo_gaussian_fwd_bwd(h_d.data_ptr(), x.data_ptr(), self.llk_part.data_ptr(),
                                     gl.data_ptr(), self.hp(H_INVB), B, self.n_per // Cc, Cc,
                                     OBS_MODE[self.observation],
                                     C.byref(npart), st)
    lib.odin_sum_parts(self.llk_part.data_ptr(), npart.value, out.data_ptr(), B, st)
    return out

  # ---- forward -----------------------------------------------------------------------
  def forward(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, st=None,
              fused: bool = True, tc_ptr: Optional[int] = None, finalize: bool = True,
              tc_split: bool = False, after_latent=None):
    """Runs encode -> reparameterise -> decode -> ELBO (+ dlogits).  `eps=None` draws the
    noise on device from the Philox stream (seed, step).  `after_latent()` (optional) is called once z is issued
    on the stream -- FactorVAE forks its discriminator pass onto a side stream there, beside the decoder."""
    lib, B, D = self.lib, self.B, self.D
    st = self.stream() if st is None else st
    assert x.shape == (B,) + self.in_shape and x.is_contiguous()
    self.x = x
    lw = self.params[self.lat_w_off:]
    lb = self.params[self.lat_b_off:]
    self._used_block = self.lat_block and fused
    self._used_neck = self.neck and fused
    self._clear_stale_act_words(st)
    # (outs[0] of the decoder comes from the bottleneck launch in the fused step: no word for layer 1's input then)
    if len(self.dec_recs) > 1:
      self.dec.set_x_word(1, not self._used_block)
    if self._used_neck:
      if eps is not None and eps is not self.eps:
        self.eps.copy_(eps)
      self.enc.forward(x, st, upto=len(self.enc_recs) - 2)
      self._neck_fwd(eps, st)
      dec_in, dec_start = self.dec.outs[1], 2
    elif self._used_block:
      if eps is not None and eps is not self.eps:
        self.eps.copy_(eps)
      h_e = self.enc.forward(x, st)
      r0 = self.dec_recs[0]
      lib.odin_latent_block_fwd(h_e.data_ptr(), lw.data_ptr(), lb.data_ptr(),
                                None if eps is None else self.eps.data_ptr(), self.eps.data_ptr(),
                                self.seed, self.hp(N_HYPER), self.p.data_ptr(), self.z.data_ptr(),
                                self.kl.data_ptr(), self.fbmask.data_ptr(), self.dec.w(0).data_ptr(),
                                self.dec.b(0).data_ptr(), self.dec.outs[0].data_ptr(), B, self.hdim, D,
                                r0.N, ACT[r0.act], int(self.analytic), self.free_bits,
                                self.hp(H_CAP) if self.capacity_on else None, st)
      dec_in, dec_start = self.dec.outs[0], 1
    else:
      if eps is None:
        lib.odin_rng_normal(self.eps.data_ptr(), B * D, self.seed, self.hp(N_HYPER), st)
      elif eps is not self.eps:
        self.eps.copy_(eps)
      h_e = self.enc.forward(x, st)
      lib.odin_dense_fwd(h_e.data_ptr(), lw.data_ptr(), lb.data_ptr(), self.p.data_ptr(), B,
                         self.hdim, 2 * D, 0, st)
      lib.odin_latent_fwd(self.p.data_ptr(), self.eps.data_ptr(), self.z.data_ptr(),
                          self.kl.data_ptr(), self.fbmask.data_ptr(), B, D, int(self.analytic),
                          self.free_bits, self.hp(H_CAP) if self.capacity_on else None, st)
      dec_in, dec_start = self.z, 0
    if after_latent is not None:
      after_latent()
    npart = C.c_int(0)
    if self.fused_tail:
      # gouts[-2] comes from the fused tail (which keeps the range word on the plane kernel) or, unfused, from the
      # 1x1 head's data gradient
      nd2 = len(self.dec_recs) - 2
      # (always kept: by the fused tail for its g_out, by the 1x1 head's data gradient otherwise -- the range contract)
      self.dec.descs[nd2].dy_amax = self.dec.word(nd2)
    self._used_head = False
    if self.gauss_head:
      # gouts[-2] comes from the fused head (which keeps its range word) or from the 1x1 head's data gradient
      nd2 = len(self.dec_recs) - 2
      self.dec.set_top_word(True, nd2)
    if self.gauss_head and fused:
      nd = len(self.dec_recs)
      hm = self.dec.forward(dec_in, st, upto=nd - 1, start=dec_start)
      a, b = self.dec_recs[-2], self.dec_recs[-1]
      Cc = self.in_shape[-1]
      rows = C.c_int(0)
      lib.odin_gaussian_head_fwd_bwd(
          hm.data_ptr(), self.dec.w(nd - 1).data_ptr(), self.dec.b(nd - 1).data_ptr(), x.data_ptr(),
          self.dec.outs[-1].data_ptr(), None, self.dec.gouts[-2].data_ptr(), self.head_llk_part.data_ptr(),
          C.byref(npart), self.head_slab.data_ptr(), C.byref(rows),
          self.head_colsum.data_ptr() if self.head_colsum is not None else None, self.hp(H_INVB), B,
          self.n_per // Cc, b.desc['Cin'], Cc, self.head_mode, ACT[a.act], self.dec.word(nd - 2), st)
      assert rows.value == self.head_rows and npart.value == self.head_npart
      self._used_fused, self._used_head = False, True
      llk_part = self.head_llk_part
      h_d = self.dec.outs[-1]
    elif self.fused_tail and fused:
      nd = len(self.dec_recs)
      h = self.dec.forward(dec_in, st, upto=nd - 2, start=dec_start)
      a, b = self.dec_recs[-2], self.dec_recs[-1]
      rows = C.c_int(0)
      if self.tail_mode is not None:
        lib.odin_gaussian_tail_fwd_bwd(
            h.data_ptr(), self.dec.w(nd - 2).data_ptr(), self.dec.b(nd - 2).data_ptr(),
            self.dec.w(nd - 1).data_ptr(), self.dec.b(nd - 1).data_ptr(), x.data_ptr(),
            self.dec.outs[-1].data_ptr(), self.dec.gouts[-2].data_ptr(), self.tail_llk_part.data_ptr(),
            C.byref(npart), self.tail_slab.data_ptr(), C.byref(rows), self.hp(H_INVB),
            C.byref(self.dec.descs[-2]), self.in_shape[-1], self.tail_mode, st)
        assert rows.value == self.tail_rows and npart.value == self.tail_npart
      else:
        lib.odin_bernoulli_tail_fwd_bwd(
            int(a.kind == 'deconv'), h.data_ptr(), self.dec.w(nd - 2).data_ptr(),
            self.dec.b(nd - 2).data_ptr(), self.dec.w(nd - 1).data_ptr(),
            self.dec.b(nd - 1).data_ptr(), x.data_ptr(), self.dec.outs[-1].data_ptr(),
            self.dec.gouts[-2].data_ptr(), self.tail_llk_part.data_ptr(), C.byref(npart),
            self.tail_slab.data_ptr(), C.byref(rows), self.hp(H_INVB), C.byref(self.dec.descs[-2]),
            b.desc['Cout'], st)
      self._used_fused = True
      llk_part = self.tail_llk_part
      h_d = self.dec.outs[-1]
    else:
      self._used_fused = False
      llk_part = self.llk_part
      h_d = self.dec.forward(dec_in, st, start=dec_start)
    gl = self.dec.gouts[-1]
    if self._used_fused or self._used_head:
      pass
    elif self.observation == 'bernoulli':
      lib.odin_elbo_bernoulli_fwd_bwd(h_d.data_ptr(), x.data_ptr(), self.llk_part.data_ptr(),
                                      gl.data_ptr(), self.hp(H_INVB), B, self.n_per,
                                      C.byref(npart), st)
    elif self.observation == 'mixqlogistic':
      Cc = self.in_shape[-1]
      lib.odin_elbo_mixqlogistic_fwd_bwd(h_d.data_ptr(), x.data_ptr(), self.llk_part.data_ptr(),
                                         gl.data_ptr(), self.hp(H_INVB), B, self.n_per // Cc, Cc,
                                         MIXQL_K, C.byref(npart), st)
    else:
      Cc = self.in_shape[-1]
      lib.odin_elbo_gaussian_fwd_bwd(h_d.data_ptr(), x.data_ptr(), self.llk_part.data_ptr(),
                                     gl.data_ptr(), self.hp(H_INVB), B, self.n_per // Cc, Cc,
                                     OBS_MODE[self.observation],
                                     C.byref(npart), st)
    self.n_part = npart.value
    tcp = None
    if self.tc_sharded and tc_split:
      # segmented step: only the local part here; the caller runs tc_gather | tc_shard | tc_scatter and
      # then finalize(tc_ptr) (train_step's program)
      self._tc_pack()
      self._llk_part_used = llk_part
      return h_d
    if self.tc_sharded:
      self._tc_pack()
      self._tc_gather()
      self._tc_shard(st)
      self._tc_scatter()
      tcp = self.tc_ws.data_ptr()
    elif self.tc_mode == 'betatc':
      lib.odin_total_correlation_fwd_bwd(self.z.data_ptr(), self.p.data_ptr(),
                                         self.tc_ws.data_ptr(), self.tc_dz.data_ptr(),
                                         self.tc_dloc.data_ptr(), self.tc_dscale.data_ptr(),
                                         self.hp(H_TCGRAD), B, D, st)
      tcp = self.tc_ws.data_ptr()
    self._llk_part_used = llk_part
    if tc_ptr is not None:
      tcp = tc_ptr
    if finalize:
      lib.odin_elbo_finalize(llk_part.data_ptr(), self.n_part, self.kl.data_ptr(),
                             self.hp(H_BETA), tcp, self.llk.data_ptr(), self.out4.data_ptr(), B,
                             st)
    return h_d

  # total_correlation over the GLOBAL batch (losses.py:136-157 couples all pairs): one all-gather of
  # [B, 3D], the shard kernel, a reduce-scatter of the two [B_global, D] posterior-side gradient planes and a
  # scalar all-reduce for the reported value.  Identical to a single GPU holding the whole batch.  Four
  # pieces so that the step can be replayed as graphs around the two collectives.
  def _tc_pack(self):
    D = self.D
    self.tc_pz_local[:, :2 * D].copy_(self.p)
    self.tc_pz_local[:, 2 * D:].copy_(self.z)

  def _tc_gather(self):
    self._comm().all_gather(self.tc_pz_all.view(-1), self.tc_pz_local.view(-1))

  def _tc_shard(self, st=None):
    lib, B, D, W = self.lib, self.B, self.D, self.world_size
    st = self.stream() if st is None else st
    self.tc_p_all.copy_(self.tc_pz_all[:, :2 * D])
    dl, ds = self.tc_part_all[0], self.tc_part_all[1]  # [Bg, D] planes
    lib.odin_total_correlation_shard(self.z.data_ptr(), self.tc_p_all.data_ptr(),
                                     self.tc_ws.data_ptr(), self.tc_dz.data_ptr(), dl.data_ptr(),
                                     ds.data_ptr(), self.hp(H_TCGRAD), B, B * W, D, st)

  def _tc_scatter(self):
    c = self._comm()
    c.reduce_scatter(self.tc_dloc.view(-1), self.tc_part_all[0].view(-1))
    c.reduce_scatter(self.tc_dscale.view(-1), self.tc_part_all[1].view(-1))
    c.all_reduce(self.tc_ws[:1])

  def finalize(self, tc_ptr: Optional[int] = None, st=None):
    st = self.stream() if st is None else st
    self.lib.odin_elbo_finalize(self._llk_part_used.data_ptr(), self.n_part, self.kl.data_ptr(),
                                self.hp(H_BETA), tc_ptr, self.llk.data_ptr(),
                                self.out4.data_ptr(), self.B, st)

  # ---- backward ----------------------------------------------------------------------
  def _fork(self):
    """Returns (fork(i) -> raw side-stream handle ordered after everything issued so far on
    the current stream, join()).  None on CPU / when overlap is disabled."""
    if self.side_stream is None or not (self.overlap_wgrad or self.early_reduce):
      return None, (lambda: None)
    cur = torch.cuda.current_stream(self.device)
    sides = self.side_streams
    nxt = [0]

    def fork(i):
      side = sides[nxt[0] % len(sides)]  # round robin: small launches also overlap each other
      nxt[0] += 1
      ev = torch.cuda.Event()
      ev.record(cur)
      side.wait_event(ev)
      return side.cuda_stream

    small_on = self.overlap_wgrad == 'small'   # (early_reduce alone forks only the decoder's slab reduction)
    fork.wants = lambda small: small_on and bool(small)

    def join():
      for side in sides[:max(1, min(nxt[0], len(sides)))]:   # (only the side streams that were forked onto)
        cur.wait_stream(side)

    return fork, join

  def backward(self, st=None, extra_dz: Optional[torch.Tensor] = None, phase: Optional[str] = None):
    """phase None: the whole backward pass and ONE slab reduction.  'dec' / 'enc': the decoder's share
    (down to dz) / the rest, each followed by the reduction of its own slabs -- the two-bucket
    data-parallel step all-reduces the decoder's gradients while 'enc' runs."""
    lib, B, D = self.lib, self.B, self.D
    st = self.stream() if st is None else st
    fork, join = self._fork() if phase is None else (None, (lambda: None))
    if phase is None and fork is None and self.defer_wgrad:
      # the plane layers' weight gradients depend on nothing but their own layer's tensors: collected here and issued
      # as ONE launch just before the slab reduction (include/odin_hip.h: odin_wgrad_planes_defer_begin)
      lib.odin_wgrad_planes_defer_begin()
    if phase == 'enc':
      jobs = []
      late_jobs = []
      early = False
    else:
      jobs, late_jobs, early = self._backward_dec(st, fork)
      if phase == 'dec':
        arr = (ReduceJob * len(jobs))(*jobs)
        self._jobs_keepalive0 = arr
        lib.odin_slab_reduce(arr, len(jobs), st)
        return
    self._backward_enc(st, extra_dz, fork, join, jobs, late_jobs, early)

  def _backward_dec(self, st, fork):
    lib, B, D = self.lib, self.B, self.D
    late_jobs: list = []  # slabs written on side streams
    early = fork is not None and self.early_reduce
    if self._used_fused:
      nd = len(self.dec_recs)
      a, b = self.dec_recs[-2], self.dec_recs[-1]
      co, c1 = a.desc['Cout'], b.desc['Cout']
      jobs = self.dec.backward(self.z, self.dec.gouts[-2], st, dx_out=self.dz, last=nd - 2,
                               skip_bias_of_last=True, fork=fork,
                               side_jobs=late_jobs if early else None, first=self._dec_first())
      ts, stride = self.tail_slab, self.tail_slab.shape[1]
      # (dW1 | db1) of the 1x1 conv, then the bias gradient of the fused layer
      jobs.append(ReduceJob(ts.data_ptr(), self.grads[b.w_off:].data_ptr(), co * c1 + c1,
                            self.tail_rows, stride, 0))
      jobs.append(ReduceJob(ts[:, co * c1 + c1:].data_ptr(), self.grads[a.b_off:].data_ptr(), co,
                            self.tail_rows, stride, 0))
    elif self._used_head:
      nd = len(self.dec_recs)
      a, b = self.dec_recs[-2], self.dec_recs[-1]
      jobs = self.dec.backward(self.z, self.dec.gouts[-2], st, dx_out=self.dz, last=nd - 2, fork=fork,
                               side_jobs=late_jobs if early else None, first=self._dec_first())
      hs = self.head_slab
      jobs.append(ReduceJob(hs.data_ptr(), self.grads[b.w_off:].data_ptr(), hs.shape[1], self.head_rows,
                            hs.shape[1], 0))  # (dW1 | db1) of the 1x1 head
      if self.head_colsum is not None:  # bias gradient of the Conv2DTranspose below = column sums of gouts[-2]
        hc = self.head_colsum
        jobs.append(ReduceJob(hc.data_ptr(), self.grads[a.b_off:].data_ptr(), a.b_n, self.head_rows, hc.shape[1], 0))
    else:
      jobs = self.dec.backward(self.z, self.dec.gouts[-1], st, dx_out=self.dz, fork=fork,
                               side_jobs=late_jobs if early else None, first=self._dec_first())
    if early and jobs:
      # the decoder's slabs (most of the slab bytes) are complete: reduce them on a side stream
      # while the encoder's backward pass keeps the matrix cores busy (the reduction is HBM-bound)
      arr0 = (ReduceJob * len(jobs))(*jobs)
      self._jobs_keepalive0 = arr0
      lib.odin_slab_reduce(arr0, len(jobs), fork(-3))
      jobs = []
    return jobs, late_jobs, early

  def _backward_enc(self, st, extra_dz, fork, join, jobs, late_jobs, early):
    lib, B, D = self.lib, self.B, self.D
    dzx = extra_dz.data_ptr() if extra_dz is not None else None
    if self.tc_mode == 'betatc':
      assert extra_dz is None
      dzx = self.tc_dz.data_ptr()
    tl = self.tc_dloc.data_ptr() if self.tc_mode == 'betatc' else None
    ts = self.tc_dscale.data_ptr() if self.tc_mode == 'betatc' else None
    h_e = self.enc.outs[-1]
    last = self.enc_recs[-1]
    aux_act = ACT[last.act]
    lw = self.params[self.lat_w_off:]
    if self._bwd_neck():
      ne = len(self.enc_recs)
      self._neck_bwd(dzx, tl, ts, st, jobs)
      jobs += self.enc.backward(self.x, self.enc.gouts[ne - 3], st, fork=fork, last=ne - 3)
    elif self._bwd_block():
      r0 = self.dec_recs[0]
      lib.odin_latent_block_bwd(self.dec.gouts[0].data_ptr(), self.dec.w(0).data_ptr(), self.z.data_ptr(),
                                self.p.data_ptr(), self.eps.data_ptr(), self.fbmask.data_ptr(),
                                self.hp(H_KLW), dzx, tl, ts, lw.data_ptr(), h_e.data_ptr(), aux_act,
                                self.dz.data_ptr(), self.dp.data_ptr(), self.enc.gouts[-1].data_ptr(),
                                self.lb_slab0.data_ptr(), self.lb_slabl.data_ptr(), B, self.hdim, D, r0.N,
                                int(self.analytic), self.enc.set_top_word(True), st)
      jobs.append(ReduceJob(self.lb_slab0.data_ptr(), self.grads[r0.w_off:].data_ptr(),
                            self.lb_slab0.shape[1], self.lb_rows, self.lb_slab0.shape[1], 0))
      jobs.append(ReduceJob(self.lb_slabl.data_ptr(), self.grads[self.lat_w_off:].data_ptr(),
                            self.lb_slabl.shape[1], self.lb_rows, self.lb_slabl.shape[1], 0))
    else:
      lib.odin_latent_bwd(self.p.data_ptr(), self.eps.data_ptr(), self.z.data_ptr(),
                          self.dz.data_ptr(), dzx, self.fbmask.data_ptr(), self.hp(H_KLW), tl, ts,
                          self.dp.data_ptr(), B, D, int(self.analytic), st)
      rows = C.c_int(0)
      lib.odin_dense_wgrad(h_e.data_ptr(), self.dp.data_ptr(), self.lat_slab.data_ptr(),
                           C.byref(rows), B, self.hdim, 2 * D,
                           st if (fork is None or not fork.wants(True)) else fork(-1))  # small
      jobs.append(ReduceJob(self.lat_slab.data_ptr(), self.grads[self.lat_w_off:].data_ptr(),
                            self.lat_slab.shape[1], rows.value, self.lat_slab.shape[1], 0))
      auxp = h_e.data_ptr() if aux_act != 0 else None
      bslab = self.enc.bslabs[-1]
      # (the projection's data gradient keeps the range word of the encoder's top gradient where its kernel family does)
      top = self.enc.set_top_word(True)
      lib.odin_dense_bwd(None, self.dp.data_ptr(), lw.data_ptr(), auxp, aux_act, self.enc.gouts[-1].data_ptr(),
                         bslab.data_ptr() if bslab is not None else None, C.byref(rows), None, None, B,
                         self.hdim, 2 * D, 0, 1, None, top, st)
      if bslab is not None:
        jobs.append(ReduceJob(bslab.data_ptr(), self.grads[last.b_off:].data_ptr(), last.b_n,
                              rows.value, last.b_n, 0))
    if not self._bwd_neck():
      jobs += self.enc.backward(self.x, self.enc.gouts[-1], st, fork=fork)
    jobs += late_jobs
    # the range words of the gradient tensors (their producers fold in with atomicMax, so every step starts from
    # zero): cleared by the step's LAST backward launch -- a reduction over zero slab rows writes zeros -- instead
    # of a memset node of its own at the top of the step (4.3 us in the step timeline)
    rw = self.range_words
    jobs.append(ReduceJob(rw.data_ptr(), rw.data_ptr(), rw.numel(), 0, rw.numel(), 0))
    self._act_words_dirty = False
    join()
    arr = (ReduceJob * len(jobs))(*jobs)
    self._jobs_keepalive = arr
    lib.odin_wgrad_planes_defer_end(st)   # (no-op unless backward() opened a collection)
    if self.debug_check_ranges:
      self.dec.check_range_words(first=self._dec_first() - 1 if self._bwd_neck() else 0)
      self.enc.check_range_words()
    self._norm_parts = 0
    if getattr(self, '_fuse_norm_now', False) and not early:
      # the gradient norm's stage-1 launch rides in the reduction (include/odin_hip.h: odin_slab_reduce_sumsq) -- its
      # jobs tile the flat gradient buffer exactly once (checked on the first call) -- and the launch stages this step's
      # hyper-parameter row for the Adam launch (odin_adam_ring_parts), which advances `hyper` itself
      nparts = C.c_int(0)
      # (checked once per job-list signature: the fused / head / block variants of a step build different lists)
      sig = tuple((jb.dst, jb.n) for jb in jobs)
      ok = self._jobs_cover_ok.get(sig)
      if ok is None:
        ok = self._jobs_cover(jobs)
        if ok:
          # every Adam workgroup re-sums the partials: beyond ~2 k of them (wide Dense layers: one per 256 result
          # elements) that costs more than the launch it saves -- such models keep the separate stage-1 launch
          lib.odin_slab_reduce_sumsq(arr, len(jobs), self.grads.data_ptr(), self.grads.numel(), None, C.byref(nparts),
                                     None, None, 0, st)
          ok = 0 < nparts.value <= 2048
        self._jobs_cover_ok[sig] = ok
      if not ok:
        # (a parameter whose gradient does not arrive through a slab job, or too many partials: the norm keeps its own
        # stage-1 launch -- odin_sumsq_adam_ring in adam())
        lib.odin_slab_reduce(arr, len(jobs), st)
        return
      lib.odin_slab_reduce_sumsq(arr, len(jobs), self.grads.data_ptr(), self.grads.numel(), self.ws.data_ptr(),
                                 C.byref(nparts), self.hyper.data_ptr(), self.hyper_staged.data_ptr(), N_HYPER + 4, st)
      assert 0 < nparts.value <= self.ws.numel(), (nparts.value, self.ws.numel())
      self._norm_parts = nparts.value
      return
    lib.odin_slab_reduce(arr, len(jobs), st)

  def _jobs_cover(self, jobs) -> bool:
    """do the reduction jobs that write into the flat gradient buffer tile it exactly once (so that the sum of their
    squares IS the squared gradient norm)?"""
    g0, n = self.grads.data_ptr(), self.grads.numel()
    cover = np.zeros(n, dtype=np.int32)
    for jb in jobs:
      off = (jb.dst - g0) // 4 if jb.dst is not None else -1
      if 0 <= off < n:
        if off + jb.n > n:
          return False
        cover[off:off + jb.n] += 1
    end = self.n_params_end()   # (beyond it: the padding of the flat buffers to a multiple of 4, zero for ever)
    return bool(int(cover[:end].min()) == 1 and int(cover.max()) == 1 and int(cover[end:].sum()) == 0)

  # ---- optimiser ---------------------------------------------------------------------
  def grad_policies(self, st=None, clipnorm: Optional[float] = None,
                    skip_update_threshold: Optional[float] = None):
    """skip_update_threshold, then per-variable clip_by_norm (base_networks.py:549-583), on the
    flat (already all-reduced) gradient buffer.  `step >= when_skip_update` is the device flag
    written by set_hyper(skip_enable=...)."""
    lib = self.lib
    st = self.stream() if st is None else st
    if skip_update_threshold is not None:
      lib.odin_grad_skip_threshold(self.grads.data_ptr(), self.grads.numel(),
                                   float(skip_update_threshold), self.hp(N_HYPER + 1),
                                   self.skip_hit.data_ptr(), self.skipped_update.data_ptr(), st)
    if clipnorm is not None:
      lib.odin_clip_by_norm_segments(self.grads.data_ptr(), self.seg_offsets.data_ptr(),
                                     self.seg_offsets.numel() - 1, float(clipnorm), st)

  def adam(self, st=None, global_clipnorm: Optional[float] = None, check_nan: bool = True,
           clipvalue: Optional[float] = None, ring: bool = False):
    """`ring`: this is the LAST launch of a train_step whose hyper-parameter rows live in the device ring: the Adam
    launch also loads the next step's row into `hyper` (odin_sumsq_adam_ring)."""
    lib = self.lib
    st = self.stream() if st is None else st
    fin, self._fin_pending = getattr(self, '_fin_pending', None), None
    if getattr(self, '_norm_parts', 0) > 0:
      # the norm's partials and the staged hyper-parameter row were left by this step's slab reduction: ONE launch
      # (`ring` False: the caller writes `hyper` itself every step -- FactorVAE's iteration -- nothing to advance)
      assert clipvalue is None and (global_clipnorm is not None or check_nan)
      lib.odin_adam_ring_parts(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                               self.params.numel(), self.hyper_staged.data_ptr(), H_ALPHA, H_BETA, self.ws.data_ptr(),
                               self._norm_parts, self.gnorm2.data_ptr(), float(global_clipnorm or 0.0),
                               self.flag.data_ptr() if check_nan else None,
                               fin[0] if fin is not None else None, fin[1] if fin is not None else 0,
                               self.kl.data_ptr(), fin[2] if fin is not None else None, self.llk.data_ptr(),
                               self.out4.data_ptr(), self.B, self.hyper_ring.data_ptr() if ring else None,
                               self.hyper.data_ptr(), self.ring_rows, N_HYPER + 4, N_HYPER, st)
      self._norm_parts = 0
      return
    if ring:
      assert clipvalue is None and (global_clipnorm is not None or check_nan)
      lib.odin_sumsq_adam_ring(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                               self.params.numel(), self.hp(H_ALPHA), self.ws.data_ptr(), self.gnorm2.data_ptr(),
                               float(global_clipnorm or 0.0), self.flag.data_ptr() if check_nan else None,
                               fin[0] if fin is not None else None, fin[1] if fin is not None else 0,
                               self.kl.data_ptr(), self.hp(H_BETA), fin[2] if fin is not None else None,
                               self.llk.data_ptr(), self.out4.data_ptr(), self.B, self.hyper_ring.data_ptr(),
                               self.hyper.data_ptr(), self.hyper_staged.data_ptr(), self.ring_rows, N_HYPER + 4,
                               N_HYPER, st)
      return
    if fin is not None and (clipvalue is not None or not (global_clipnorm is not None or check_nan)):
      # (an update path without the fused first launch: finalise on its own)
      lib.odin_elbo_finalize(fin[0], fin[1], self.kl.data_ptr(), self.hp(H_BETA), fin[2], self.llk.data_ptr(),
                             self.out4.data_ptr(), self.B, st)
      fin = None
    if clipvalue is not None:
      # reference order: clip_by_global_norm, THEN clip_by_value (base_networks.py:584-596): the
      # norm is taken first, one elementwise launch scales and clamps, Adam runs unscaled (its
      # NaN guard still reads the norm)
      lib.odin_sumsq_flat(self.grads.data_ptr(), self.grads.numel(), self.ws.data_ptr(),
                          self.gnorm2.data_ptr(), st)
      lib.odin_clip_by_value(self.grads.data_ptr(), self.grads.numel(), float(clipvalue),
                             self.gnorm2.data_ptr() if global_clipnorm else None,
                             float(global_clipnorm or 0.0), st)
      lib.odin_adam_step_flat(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                              self.v.data_ptr(), self.params.numel(), self.hp(H_ALPHA),
                              self.gnorm2.data_ptr() if check_nan else None, 0.0,
                              self.flag.data_ptr(), st)
      return
    if global_clipnorm is not None or check_nan:
      # gradient norm (clip scale, NaN guard) + update: stage-1 partial sums, then ONE launch that
      # finishes the norm and applies Adam
      if fin is not None and clipvalue is None:
        lib.odin_sumsq_adam_finalize_flat(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                          self.v.data_ptr(), self.params.numel(), self.hp(H_ALPHA),
                                          self.ws.data_ptr(), self.gnorm2.data_ptr(),
                                          float(global_clipnorm or 0.0),
                                          self.flag.data_ptr() if check_nan else None, fin[0], fin[1],
                                          self.kl.data_ptr(), self.hp(H_BETA), fin[2], self.llk.data_ptr(),
                                          self.out4.data_ptr(), self.B, st)
        return
      lib.odin_sumsq_adam_flat(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                               self.v.data_ptr(), self.params.numel(), self.hp(H_ALPHA),
                               self.ws.data_ptr(), self.gnorm2.data_ptr(),
                               float(global_clipnorm or 0.0), self.flag.data_ptr() if check_nan else None, st)
      return
    lib.odin_adam_step_flat(self.params.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                            self.v.data_ptr(), self.params.numel(), self.hp(H_ALPHA), None,
                            0.0, self.flag.data_ptr(), st)

  def allreduce(self, lo: int = 0, hi: Optional[int] = None):
    """SUM all-reduce of (a range of) the flat gradient buffer over the ranks"""
    if self.is_dp:
      self._comm().all_reduce(self.grads[lo:hi])

  # ---- one optimisation step ---------------------------------------------------------
  def step_program(self, x, eps, pol):
    """The launch program of one step: [('k', fn) kernels | ('c', fn) collective].  Single GPU: all
    kernels.  Data parallel: ... backward | all-reduce | update, with the decoder's gradient bucket
    reduced on a side stream beside the encoder's backward pass when dp_buckets == 2; beta-TC under
    data parallelism adds all-gather and reduce-scatter around the total-correlation shard kernel."""
    P = []
    tc_dp = self.tc_sharded
    if tc_dp:
      P.append(('k', lambda: self.forward(x, eps, tc_split=True)))
      P.append(('c', self._tc_gather))
      P.append(('k', self._tc_shard))
      P.append(('c', self._tc_scatter))
      P.append(('k', lambda: self.finalize(self.tc_ws.data_ptr())))
    else:
      # the ELBO finalisation (llk[B], loss, mean terms: nothing in the backward pass reads them) rides in the
      # update's first launch instead of being a launch of its own
      def fwd():
        self.forward(x, eps, finalize=False)
        tcp = self.tc_ws.data_ptr() if self.tc_mode == 'betatc' else None
        self._fin_pending = (self._llk_part_used.data_ptr(), self.n_part, tcp)
      P.append(('k', fwd))
    if self.is_dp and self.dp_buckets >= 2:
      cur = lambda: torch.cuda.current_stream(self.device)
      side = self.side_stream  # (None on the CPU: the two buckets reduce one after the other)

      def reduce_dec_async():
        if side is None:
          return self.allreduce(self.dec_start, None)
        side.wait_stream(cur())
        with torch.cuda.stream(side):
          self.allreduce(self.dec_start, None)

      def reduce_enc_and_join():
        self.allreduce(0, self.dec_start)
        if side is not None:
          cur().wait_stream(side)

      P.append(('k', lambda: self.backward(phase='dec')))
      P.append(('c', reduce_dec_async))
      P.append(('k', lambda: self.backward(phase='enc')))
      P.append(('c', reduce_enc_and_join))
    else:
      # single device, no per-tensor policies between the reduction and Adam: the norm rides in the reduction
      fuse = bool(len(pol) > 5 and pol[5] and not self.is_dp and pol[1] is None and pol[3] is None and self.fuse_norm)

      def bwd():
        self._fuse_norm_now = fuse
        try:
          self.backward()
        finally:
          self._fuse_norm_now = False
      P.append(('k', bwd))
      if self.is_dp:
        P.append(('c', self.allreduce))
    P.append(('k', lambda: self._update(pol)))
    return P

  def train_step(self, x: torch.Tensor, eps: Optional[torch.Tensor] = None, lr=1e-3, beta=1.0,
                 global_clipnorm: Optional[float] = None, use_graph: bool = False,
                 clipnorm: Optional[float] = None, clipvalue: Optional[float] = None,
                 skip_update_threshold: Optional[float] = None, when_skip_update: int = 0,
                 check_nan: bool = True, capacity: Optional[float] = None, schedule=None):
    """Networks.optimize for one VAEStep: step += 1, forward, backward, (all-reduce), gradient
    policies, Adam.  Returns the device tensor out4 = [loss, mean llk, mean beta*kl, tc] (no host
    sync)."""
    self.step_count += 1
    # the device ring serves the update path whose last two launches are odin_sumsq_adam_ring (norm + Adam)
    ring = bool(self.use_hyper_ring and clipvalue is None and (global_clipnorm is not None or check_nan))
    if ring:
      self._ring_step(self.step_count, dict(lr=lr, beta=beta, when_skip_update=int(when_skip_update),
                                            capacity=capacity), schedule)
    else:
      self.set_hyper(lr=lr, beta=beta, skip_enable=self.step_count >= int(when_skip_update), capacity=capacity)
    pol = (global_clipnorm, clipnorm, clipvalue, skip_update_threshold, bool(check_nan), ring)
    if use_graph and self.device.type == 'cuda':
      self._graph_step(x, eps, pol)
    else:
      for _, fn in self.step_program(x, eps, pol):
        fn()
    return self.out4

  def _update(self, pol):
    gclip, clipnorm, clipvalue, skip_thr, check_nan = pol[:5]
    self.grad_policies(clipnorm=clipnorm, skip_update_threshold=skip_thr)
    self.adam(global_clipnorm=gclip, clipvalue=clipvalue, check_nan=check_nan, ring=len(pol) > 5 and pol[5])

  def input_buffer(self) -> torch.Tensor:
    """The static [B, H, W, C] input tensor the captured step graph reads.  A data pipeline that
    writes the next batch straight into it (and passes it to train_step) saves the per-step
    device-to-device copy."""
    if getattr(self, 'x_static', None) is None:
      self.x_static = torch.empty((self.B,) + self.in_shape, dtype=torch.float32, device=self.device)
    return self.x_static

  def _graph_step(self, x, eps, pol):
    """Capture the step once into HIP graphs, replay afterwards.  Single GPU: ONE graph (forward +
    backward + slab reduction + policies + Adam).  Data parallel: the kernel segments of
    `step_program` as one graph each, the RCCL collectives between them eager on the same stream
    (A | all-reduce | B; with beta-TC: A | all-gather | B | reduce-scatter | C | all-reduce | D).  The input
    batch is read from a static buffer; eps comes from the on-device Philox stream unless given
    explicitly.  Everything a capture bakes in (clip values, KL form, free bits, explicit-eps mode,
    bucket count) is part of the graph's key: changing any of them captures a new graph instead of
    silently replaying the old one."""
    from .dist import SegmentedGraph
    key = (pol, self.analytic, self.free_bits, eps is not None, self.dp_buckets)
    if not hasattr(self, '_graphs'):
      self._graphs = {}
    if key not in self._graphs:
      self.input_buffer()
      if x.data_ptr() != self.x_static.data_ptr():
        self.x_static.copy_(x)
      if eps is not None:
        self.eps.copy_(eps)
      ex = eps is not None
      sg = SegmentedGraph(self.device, self.step_program(self.x_static, self.eps if ex else None, pol))
      # warm-up outside capture (first-call attribute setup, lazy allocations, communicator set-up)
      cap = torch.cuda.Stream(self.device)
      cap.wait_stream(torch.cuda.current_stream(self.device))
      saved = (self.params.clone(), self.m.clone(), self.v.clone(), self.flag.clone(),
               self.skipped_update.clone(), self.hyper.clone())
      with torch.cuda.stream(cap):
        sg.run_eager()
        # the warm-up step must not count: restore the optimiser state it touched (and the hyper-parameter row: with
        # the device ring the warm-up's Adam has already loaded the NEXT step's row)
        self.params.copy_(saved[0]); self.m.copy_(saved[1]); self.v.copy_(saved[2])
        self.flag.copy_(saved[3]); self.skipped_update.copy_(saved[4]); self.hyper.copy_(saved[5])
      torch.cuda.current_stream(self.device).wait_stream(cap)
      if self.side_stream is not None:
        torch.cuda.current_stream(self.device).wait_stream(self.side_stream)
      sg.capture(cap)
      self._graphs[key] = sg
      self.graph = sg.graphs[0]
    sg = self._graphs[key]
    if x.data_ptr() != self.x_static.data_ptr():  # a producer may write the static buffer directly
      self.x_static.copy_(x, non_blocking=True)
    if eps is not None:
      self.eps.copy_(eps, non_blocking=True)
    sg.replay()
