"""ctypes binding of ``libodin_hip.so`` (the C ABI declared in ``include/odin_hip.h``).

The product path has NO fallback: if the gfx950 library is missing or a symbol is absent
the import of the HIP backend raises.  ``load(path)`` with an explicit path exists so
that the test-suite can point the very same Python host code at the CPU-simulated build
of the kernel sources (``tests/sim/libodin_sim.so``) for debugging in containers without
a GPU; nothing in the package ever selects that library by itself.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, 'libodin_hip.so')

ACT = {'linear': 0, 'elu': 1, 'relu': 2, None: 0}


class ConvDesc(C.Structure):
  """mirror of ``odin_conv_desc`` (include/odin_hip.h)."""
  _fields_ = [(n, C.c_int) for n in ('B', 'H', 'W', 'Cin', 'OH', 'OW', 'Cout', 'KH', 'KW',
                                     'stride', 'pad_t', 'pad_l', 'act', 'center')] + \
             [('dy_amax', C.c_void_p), ('dx_amax', C.c_void_p),  # range words of dy / dx (backward; optional)
              ('x_amax', C.c_void_p), ('y_amax', C.c_void_p)]    # range words of x / y (forward; optional)


class ReduceJob(C.Structure):
  _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('n', C.c_int), ('rows', C.c_int),
              ('stride', C.c_int), ('pad_', C.c_int)]


class AdamFold(C.Structure):
  """mirror of ``odin_adam_fold`` (include/odin_hip.h): gradient pieces formed inside the Adam launch."""
  _fields_ = [('x', C.c_void_p), ('dy', C.c_void_p), ('B', C.c_int), ('K', C.c_int), ('N', C.c_int), ('w_off', C.c_size_t),
              ('slab', C.c_void_p), ('slab_rows', C.c_int), ('slab_stride', C.c_size_t), ('slab_n', C.c_size_t),
              ('slab_off', C.c_size_t), ('zero', C.c_void_p), ('zero_n', C.c_int)]


class NeckArgs(C.Structure):
  """mirror of ``odin_neck_args`` (include/odin_hip.h): the neck of the 64x64 stacks in one launch per direction."""
  _fields_ = [(n, C.c_int) for n in ('B', 'P', 'D', 'C0', 'act2', 'act3', 'act4', 'act0', 'act1', 'analytic')] + \
             [('free_bits', C.c_float), ('seed', C.c_uint64)] + \
             [(n, C.c_void_p) for n in (
                 'step_dev', 'x', 'x_amax', 'w3', 'b3', 'y3', 'w4', 'b4', 'y4', 'wl', 'bl', 'eps_in', 'eps', 'p', 'z',
                 'kl', 'fbmask', 'capacity', 'w0', 'b0', 'y0', 'w1', 'b1', 'y1', 'y1_amax',
                 'dy1', 'klw', 'dz_extra', 'dloc_x', 'dscale_x', 'dz', 'dp', 'dh4', 'dy3', 'dx',
                 'dh4_amax', 'dy3_amax', 'dx_amax', 'slab1', 'slab0', 'slabl')]


P = C.c_void_p
I = C.c_int
F = C.c_float
IP = C.POINTER(C.c_int)
DP = C.POINTER(ConvDesc)

# name -> argtypes, exactly the declarations of include/odin_hip.h
SIGNATURES = {
    'odin_version': [],
    'odin_debug_last_path': [],
    'odin_crc32c': [C.c_uint32, P, C.c_size_t],
    'odin_max_slab_rows': [],
    'odin_range_reset': [P, I, P],
    'odin_conv2d_dgrad_keeps_range': [DP, I],
    'odin_deconv2d_dgrad_keeps_range': [DP, I],
    'odin_bernoulli_tail_keeps_range': [I, DP, I],
    'odin_conv2d_bwd': [P, P, P, P, I, P, P, IP, P, IP, DP, P],
    'odin_deconv2d_bwd': [P, P, P, P, I, P, P, IP, P, IP, DP, P],
    'odin_dense_bwd': [P, P, P, P, I, P, P, IP, P, IP, I, I, I, I, I, P, P, P],
    'odin_dense_bwd_ranged': [P, P, P, P, I, P, P, IP, P, IP, I, I, I, I, I, P, P, P, P],
    'odin_dense_dgrad_keeps_range': [I, I, I],
    'odin_conv2d_reads_x_range': [DP],
    'odin_deconv2d_reads_x_range': [DP],
    'odin_dense_reads_x_range': [I, I, I],
    'odin_absmax': [P, C.c_size_t, P, P],
    'odin_debug_absmax_fallbacks': [],
    'odin_debug_stream_probe': [P, P, P, C.c_size_t, I, I, P],
    'odin_comm_unique_id': [P],
    'odin_comm_init': [C.POINTER(C.c_void_p), P, I, I],
    'odin_comm_destroy': [P],
    'odin_comm_library': [],
    'odin_allreduce_flat': [P, P, C.c_size_t, P],
    'odin_allgather_flat': [P, P, P, C.c_size_t, P],
    'odin_reduce_scatter_flat': [P, P, P, C.c_size_t, P],
    'odin_conv2d_fwd': [P, P, P, P, DP, P],
    'odin_conv2d_dgrad': [P, P, P, I, P, P, IP, DP, P],
    'odin_conv2d_wgrad': [P, P, P, IP, DP, P],
    'odin_deconv2d_fwd': [P, P, P, P, DP, P],
    'odin_deconv2d_dgrad': [P, P, P, I, P, P, IP, DP, P],
    'odin_deconv2d_wgrad': [P, P, P, IP, DP, P],
    'odin_bernoulli_tail_fwd_bwd': [I, P, P, P, P, P, P, P, P, P, IP, P, IP, P, DP, I, P],
    'odin_dense_fwd': [P, P, P, P, I, I, I, I, P],
    'odin_dense_fwd_ranged': [P, P, P, P, I, I, I, I, P, P, P],
    'odin_dense_dgrad': [P, P, P, I, P, P, IP, I, I, I, P],
    'odin_dense_wgrad': [P, P, P, IP, I, I, I, P],
    'odin_slab_reduce': [C.POINTER(ReduceJob), I, P],
    'odin_latent_fwd': [P, P, P, P, P, I, I, I, F, P, P],
    'odin_latent_bwd': [P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    'odin_latent_block_rows': [I, I, I, I],
    'odin_latent_block_fwd': [P, P, P, P, P, C.c_uint64, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, P, P],
    'odin_latent_block_bwd': [P, P, P, P, P, P, P, P, P, P, P, P, I, P, P, P, P, P, I, I, I, I, I, P, P],
    'odin_wgrad_pair_begin': [],
    'odin_wgrad_pair_end': [],
    'odin_neck_rows': [I, I, I, I],
    'odin_neck_fwd': [C.POINTER(NeckArgs), P],
    'odin_neck_bwd': [C.POINTER(NeckArgs), P],
    'odin_elbo_bernoulli_fwd_bwd': [P, P, P, P, P, I, I, IP, P],
    'odin_elbo_gaussian_fwd_bwd': [P, P, P, P, P, I, I, I, I, IP, P],
    'odin_gaussian_tail_applicable': [DP, I],
    'odin_gaussian_tail_fwd_bwd': [P, P, P, P, P, P, P, P, P, IP, P, IP, P, DP, I, I, P],
    'odin_gaussian_head_fwd_bwd': [P, P, P, P, P, P, P, P, IP, P, IP, P, P, I, I, I, I, I, I, P, P],
    'odin_disc_head_rows': [I, I],
    'odin_disc_head_fwd_bwd': [P, P, P, P, I, P, P, P, I, P, P, P, IP, P, I, I, P],
    'odin_debug_igemm_h_min_flop': [C.c_double],
    'odin_debug_blk_min_flop': [C.c_double],
    'odin_debug_blk_planes': [C.c_int],
    'odin_debug_blk_first': [C.c_int],
    'odin_debug_dense_hw_min_tiles': [C.c_int],
    'odin_debug_mel_r16': [C.c_int],
    'odin_debug_smallc_planes': [C.c_int],
    'odin_wgrad_planes_defer_begin': [],
    'odin_wgrad_planes_defer_end': [P],
    'odin_elbo_mixqlogistic_fwd_bwd': [P, P, P, P, P, I, I, I, I, IP, P],
    'odin_elbo_finalize': [P, I, P, P, P, P, P, I, P],
    'odin_mean': [P, I, P, P],
    'odin_total_correlation_workspace': [I, I, I],
    'odin_total_correlation_fwd_bwd': [P, P, P, P, P, P, P, I, I, P],
    'odin_total_correlation_shard': [P, P, P, P, P, P, P, I, I, I, P],
    'odin_permute_dims': [P, P, P, I, I, P],
    'odin_random_perm': [P, I, I, C.c_uint64, P, P],
    'odin_random_permute_dims': [P, P, P, I, I, C.c_uint64, P, P],
    'odin_dtc_loss_fwd_bwd': [P, P, P, P, P, I, P],
    'odin_adam_step_flat': [P, P, P, P, C.c_size_t, P, P, F, P, P],
    'odin_adam_step_fold': [P, P, P, P, C.c_size_t, P, C.POINTER(AdamFold), P],
    'odin_sumsq_flat': [P, C.c_size_t, P, P, P],
    'odin_latent_sample_logprob': [P, P, P, P, P, I, I, I, P],
    'odin_logmeanexp_rows': [P, P, I, I, P],
    'odin_sum_parts': [P, I, P, I, P],
    'odin_grad_skip_threshold': [P, C.c_size_t, F, P, P, P, P],
    'odin_clip_by_norm_segments': [P, P, I, F, P],
    'odin_clip_by_value': [P, C.c_size_t, F, P, F, P],
    'odin_sumsq_adam_flat': [P, P, P, P, C.c_size_t, P, P, P, F, P, P],
    'odin_sumsq_adam_finalize_flat': [P, P, P, P, C.c_size_t, P, P, P, F, P, P, I, P, P, P, P, P, I, P],
    'odin_sumsq_adam_ring': [P, P, P, P, C.c_size_t, P, P, P, F, P, P, I, P, P, P, P, P, I, P, P, P, I, I, I, P],
    'odin_slab_reduce_sumsq': [P, I, P, C.c_size_t, P, IP, P, P, I, P],
    'odin_adam_ring_parts': [P, P, P, P, C.c_size_t, P, I, I, P, I, P, F, P, P, I, P, P, P, P, I, P, P, I, I, I, P],
    'odin_rng_normal': [P, C.c_size_t, C.c_uint64, P, P],
    'odin_gather_normalize_u8': [P, P, P, I, I, F, I, P],
    'odin_gather_rows_f32': [P, P, P, I, I, P],
    'odin_stft_mel_db': [P, P, P, P, P, P, I, I, I, I, I, I, C.c_double, C.c_double, I, P],
    'odin_stft_mel_db_frames': [P, P, P, P, P, P, I, I, I, I, I, I, C.c_double, C.c_double, I, I, P, P],
    'odin_debug_set_stamps': [P],
    'odin_debug_set_wgrad_stamps': [P],
    'odin_debug_set_neck_stamps': [P],
    'odin_debug_elbo_shape': [I, I, I],
    'odin_debug_set_mel_stamps': [P],
    'odin_debug_igemm_h_ldsw_steps': [I],
    'odin_graph_begin': [P],
    'odin_graph_end': [P, C.POINTER(C.c_void_p)],
    'odin_graph_launch': [P, P],
    'odin_graph_destroy': [P],
}


# entry points whose return value is a result, not an error code
VALUE_RETURNING = ('odin_version', 'odin_comm_library', 'odin_conv2d_dgrad_keeps_range',
                   'odin_deconv2d_dgrad_keeps_range', 'odin_bernoulli_tail_keeps_range', 'odin_dense_dgrad_keeps_range', 'odin_conv2d_reads_x_range',
                   'odin_deconv2d_reads_x_range', 'odin_dense_reads_x_range', 'odin_max_slab_rows', 'odin_debug_absmax_fallbacks', 'odin_crc32c', 'odin_debug_last_path',
                   'odin_latent_block_rows', 'odin_neck_rows', 'odin_debug_igemm_h_ldsw_steps', 'odin_total_correlation_workspace', 'odin_debug_igemm_h_min_flop', 'odin_debug_blk_min_flop',
                   'odin_debug_blk_planes', 'odin_debug_blk_first', 'odin_debug_dense_hw_min_tiles', 'odin_gaussian_tail_applicable', 'odin_disc_head_rows', 'odin_debug_mel_r16', 'odin_debug_smallc_planes')
# entry points declared `void` in include/odin_hip.h
VOID_RETURNING = ('odin_wgrad_planes_defer_begin', 'odin_wgrad_pair_begin')


class OdinError(RuntimeError):
  pass


class Lib:

  def __init__(self, path: str):
    if not os.path.exists(path):
      raise OdinError(
          f"HIP extension not found: {path}. Build it with `python -c 'import "
          f"__graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). There is no "
          f"CPU fallback.")
    self.path = path
    self.c = C.CDLL(path)
    self.c.odin_last_error.restype = C.c_char_p
    self.c.odin_last_error.argtypes = []
    for name, args in SIGNATURES.items():
      fn = getattr(self.c, name)  # AttributeError if the symbol is missing: fail loudly
      fn.argtypes = args
      fn.restype = (C.c_char_p if name in ('odin_debug_last_path', 'odin_comm_library') else
                    C.c_double if name in ('odin_debug_igemm_h_min_flop', 'odin_debug_blk_min_flop') else
                    None if name in VOID_RETURNING else
                    C.c_uint32 if name in VALUE_RETURNING else C.c_int)

  def check(self, rc: int, what: str = ''):
    if rc != 0:
      raise OdinError(f"{what} failed: {self.c.odin_last_error().decode()} (rc={rc})")

  def __getattr__(self, name):
    if name.startswith('odin_'):
      fn = getattr(self.c, name)

      trace = os.environ.get('ODIN_TRACE', '0') == '1'

      def call(*a):
        if trace:  # diagnostics: serialise and name every launch
          import sys
          import torch
          print('odin>', name, [x if isinstance(x, int) else '.' for x in a][-6:], flush=True,
                file=sys.stderr)
        rc = fn(*a)
        if trace:
          import torch
          if torch.cuda.is_available():
            torch.cuda.synchronize()
        if name in VOID_RETURNING:
          return None
        if rc != 0 and name not in VALUE_RETURNING:
          raise OdinError(f"{name} failed: {self.c.odin_last_error().decode()} (rc={rc})")
        return rc

      self.__dict__[name] = call
      return call
    raise AttributeError(name)


_lib: Optional[Lib] = None


def load(path: Optional[str] = None) -> Lib:
  """Load (once) and return the library.  ``path=None`` -> the in-tree gfx950 build."""
  global _lib
  if path is not None:
    _lib = Lib(path)
  elif _lib is None:
    # ODIN_HIP_LIB: another build of the SAME library (A/B timing of two builds in one gpurun call)
    _lib = Lib(os.environ.get('ODIN_HIP_LIB') or DEFAULT_LIB)
  return _lib


def ptr(t) -> Optional[int]:
  """device (or, under the simulator, host) address of a torch tensor / None."""
  if t is None:
    return None
  assert t.is_contiguous(), 'odin kernels need contiguous buffers'
  return t.data_ptr()


def conv_desc(B, H, W, Cin, OH, OW, Cout, K, stride, pad_t, pad_l, act='linear',
              center=False) -> ConvDesc:
  return ConvDesc(B, H, W, Cin, OH, OW, Cout, K, K, stride, pad_t, pad_l, ACT[act], int(center))
