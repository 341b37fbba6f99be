#!/usr/bin/env python3
"""bench.py -- VAE train images/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload at N=1 = BASELINE.json configs[1]: beta-VAE (beta=4) on dSprites-shaped synthetic
data, 64x64x1, `dsprites_networks` conv encoder/decoder, batch 256 per GPU (weak scaling:
per-GPU batch fixed, gradients summed by ONE RCCL all-reduce of the flat fp32 bucket).
A "step" = forward + backward + slab reduction + (all-reduce) + Adam on one batch already
resident in HBM.  Prints ONE JSON line (rank 0).

Extra objects in the JSON line:
  roofline     -- the dominant kernel launch, timed live with HIP events on the launch
                  stream; algorithmic FLOPs / bytes per launch are stated in DESIGN.md.
  cpu_baseline -- the same training step as a torch-CPU fp32 port (oracle/torch_ref.py,
                  all host cores), rank 0 at N=1 only, on a bounded number of steps.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-in MFMA peak
PEAK_MFMA_BF16_TFLOPS = 2516.6  # dense bf16 MFMA peak: 16 x the fp32 rate (32x32x16_bf16 in 8 passes)
PEAK_HBM_GBS = 8000.0         # spec (6.3 TB/s achievable with float4 copies)

WORKLOADS = {
    # name: (get_networks name, kwargs, batch per GPU, beta, vae kind)
    'dsprites_betavae_b256': ('dsprites', {}, 256, 4.0, None),
    'shapes3d_vae_b256': ('shapes3d', {}, 256, 1.0, None),
    'celeba_betatcvae_b512': ('celeba', {}, 512, 4.0, 'betatc'),
    'mnist_dense_b128': ('dense', {}, 128, 1.0, None),
    # BASELINE config 1, the convolutional variant (mnist_networks, image_networks.py:244-271: 5x5 kernels on 28 x 28)
    'mnist_conv_b128': ('mnist', {}, 128, 1.0, None),
    # BASELINE config 3: both optimisers, batch split 128 + 128, whole iteration as one graph
    'factorvae_shapes3d_b256': ('shapes3d', {}, 256, None, 'factor'),
    # BASELINE config 5: audio [256, 8000] -> log-mel front-end -> conv VAE, front-end INSIDE the step
    'speech_vae_b256': ('speech', {}, 256, 1.0, 'speech'),
}


def init_params_(eng, seed):
  """Random-init weights of the reference's initialisers' scale (HeNormal for elu convs,
  glorot for dense; odin/networks/image_networks.py:157-174), biases zero."""
  g = torch.Generator(device='cpu').manual_seed(seed)
  for (key, shp, off) in eng.layout.entries:
    n = int(np.prod(shp))
    if key[-1] == 'b':
      eng.params[off:off + n].zero_()
      continue
    if len(shp) == 4:
      is_deconv = any(r.kind == 'deconv' and (r.key == key[:2]) for r in eng.enc_recs + eng.dec_recs)
      fan_in = shp[0] * shp[1] * (shp[3] if is_deconv else shp[2])
      std = math.sqrt(2.0 / fan_in)
    else:
      std = math.sqrt(2.0 / (shp[0] + shp[1]))
    eng.params[off:off + n] = (torch.randn(n, generator=g) * std).to(eng.device)


def synthetic_batch(name, B, in_shape, device, seed):
  """dSprites-like: sprite masks in {1e-6, 1-1e-6}, ~5% foreground; others U(0,1) clipped
  (odin/fuel/image_data/_base.py:130-147)."""
  g = torch.Generator(device='cpu').manual_seed(seed)
  if name.startswith('dsprites'):
    x = torch.zeros(B, *in_shape)
    ys = torch.randint(4, 44, (B,), generator=g)
    xs = torch.randint(4, 44, (B,), generator=g)
    hs = torch.randint(6, 18, (B,), generator=g)
    for b in range(B):
      x[b, ys[b]:ys[b] + hs[b], xs[b]:xs[b] + hs[b], :] = 1.0
  elif name.startswith('mnist'):
    x = (torch.rand(B, *in_shape, generator=g) < 0.13).float()
    return x.to(device)
  else:
    x = torch.rand(B, *in_shape, generator=g)
  return x.clamp_(1e-6, 1 - 1e-6).to(device)


def describe_nonfinite(eng):
  """stderr: which gradient tensors of the LAST step hold non-finite values (diagnostics of a failed run)"""
  for name, prog in (('enc', eng.enc), ('dec', eng.dec)):
    for i, g in enumerate(prog.gouts):
      print(f'  {name}[{i}] {prog.recs[i].kind}: |dL/dy| max {float(g.abs().max()):.4g} finite '
            f'{bool(torch.isfinite(g).all())}, |y| max {float(prog.outs[i].abs().max()):.4g}', file=sys.stderr)
  for k, v in eng.grad_views().items():
    if not torch.isfinite(v).all():
      print(f'  non-finite gradient {k}: {int((~torch.isfinite(v)).sum())} of {v.numel()}', file=sys.stderr)


def conv_flops(rec, B):
  d = rec.desc
  if rec.kind == 'conv':
    return 2.0 * B * d['OH'] * d['OW'] * d['K'] * d['K'] * d['Cin'] * d['Cout']
  if rec.kind == 'deconv':
    return 2.0 * B * d['H'] * d['W'] * d['K'] * d['K'] * d['Cin'] * d['Cout']
  return 2.0 * B * rec.K * rec.N


def profile_ops(eng, reps=20):
  """Per-launch timing of every kernel of the step with HIP events on the launch stream."""
  import ctypes as C
  from odin_ai_amd._lib import ACT
  lib, B = eng.lib, eng.B
  st = eng.stream()
  rows = C.c_int(0)
  out = []

  def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

  nd = len(eng.dec.recs)
  ne = len(eng.enc.recs)
  fused = getattr(eng, 'fused_tail', False)
  head = getattr(eng, 'gauss_head', False)   # the Gaussian 1x1 head runs as ONE launch inside the step
  neck = bool(getattr(eng, 'neck', False))   # conv3 .. deconv1 run as ONE launch per direction (neck.hip)
  neck_bwd = neck and (eng._nk.P == 128 if eng._neck_bwd_opt is None else bool(eng._neck_bwd_opt))
  neck_fl = [0.0, 0.0]                       # FLOPs of the launches the neck replaces (forward, backward)
  for net, prog, x0 in (('enc', eng.enc, eng.x), ('dec', eng.dec, eng.z)):
    for i, r in enumerate(prog.recs):
      xin = x0 if i == 0 else prog.outs[i - 1]
      y, g, d = prog.outs[i], prog.gouts[i], prog.descs[i]
      w, b = prog.w(i), prog.b(i)
      fl = conv_flops(r, B)
      in_tail = fused and net == 'dec' and i >= nd - 2
      if r.kind == 'conv':
        f_fwd = lambda: lib.odin_conv2d_fwd(xin.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), st)
        f_wg = lambda: lib.odin_conv2d_wgrad(xin.data_ptr(), g.data_ptr(), prog.wslabs[i].data_ptr(), C.byref(rows), C.byref(d), st)
        f_dg = None
        if i > 0:
          dst, aux = prog.gouts[i - 1], prog.outs[i - 1]
          f_dg = lambda: lib.odin_conv2d_dgrad(g.data_ptr(), w.data_ptr(), aux.data_ptr(), ACT[prog.recs[i - 1].act], dst.data_ptr(), None, None, C.byref(d), st)
      elif r.kind == 'deconv':
        f_fwd = lambda: lib.odin_deconv2d_fwd(xin.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), st)
        f_wg = lambda: lib.odin_deconv2d_wgrad(xin.data_ptr(), g.data_ptr(), prog.wslabs[i].data_ptr(), C.byref(rows), C.byref(d), st)
        f_dg = None
        if i > 0:
          dst, aux = prog.gouts[i - 1], prog.outs[i - 1]
          f_dg = lambda: lib.odin_deconv2d_dgrad(g.data_ptr(), w.data_ptr(), aux.data_ptr(), ACT[prog.recs[i - 1].act], dst.data_ptr(), None, None, C.byref(d), st)
      else:
        f_fwd = lambda: lib.odin_dense_fwd(xin.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, r.K, r.N, ACT[r.act], st)
        f_wg = lambda: lib.odin_dense_wgrad(xin.data_ptr(), g.data_ptr(), prog.wslabs[i].data_ptr(), C.byref(rows), B, r.K, r.N, st)
        f_dg = None
        if i > 0:
          dst, aux = prog.gouts[i - 1], prog.outs[i - 1]
          f_dg = lambda: lib.odin_dense_dgrad(g.data_ptr(), w.data_ptr(), aux.data_ptr(), ACT[prog.recs[i - 1].act], dst.data_ptr(), None, None, B, r.K, r.N, st)
      ops = [('fwd', f_fwd), ('wgrad', f_wg), ('dgrad', f_dg)]
      if r.kind == 'deconv' and i > 0:
        # the step back-propagates a Conv2DTranspose through ONE call; where that is one launch (bwd_planes.hip: dy
        # fetched and split once for both gradients) the pair is timed -- and priced -- as the launch it is
        dst, aux = prog.gouts[i - 1], prog.outs[i - 1]
        bsl = prog.bslabs[i - 1]
        wrows = C.c_int(0)
        f_bw = lambda: lib.odin_deconv2d_bwd(xin.data_ptr(), g.data_ptr(), w.data_ptr(), aux.data_ptr(),
                                             ACT[prog.recs[i - 1].act], dst.data_ptr(),
                                             bsl.data_ptr() if bsl is not None else None, C.byref(rows),
                                             prog.wslabs[i].data_ptr(), C.byref(wrows), C.byref(d), st)
        f_bw()
        if lib.odin_debug_last_path().decode().startswith(('bwd_planes', 'bwd_blk')):
          ops = [('fwd', f_fwd), ('bwd', f_bw)]
      if in_tail:
        # inside the training step these launches are replaced by the fused tail kernel
        ops = [] if i == nd - 1 else [o for o in ops if o[0] != 'fwd']
      if head and net == 'dec' and i == nd - 1:
        ops = []  # (replaced by odin_gaussian_head_fwd_bwd, timed below)
      if neck and ((net == 'enc' and i >= ne - 2) or (net == 'dec' and i <= 1)):
        # inside the step these launches are the neck's: forward always; backward (data gradients of all four layers,
        # weight gradients of the decoder's two) where the engine uses the neck's backward launch
        neck_fl[0] += fl
        keep = []
        if neck_bwd:
          neck_fl[1] += fl * (2 if net == 'dec' else 1)
          keep = [o for o in ops if o[0] == 'wgrad'] if net == 'enc' else []
        else:
          keep = [o for o in ops if o[0] != 'fwd']
        ops = keep
      for tag, fn in ops:
        if fn is None:
          continue
        t = timeit(fn)
        flo = 2 * fl if tag == 'bwd' else fl   # (weight + data gradient)
        out.append(dict(layer=f'{net}{i}:{r.kind}', op=tag, us=t * 1e6, gflop=flo * 1e-9,
                        tflops=flo / t * 1e-12, path=lib.odin_debug_last_path().decode()))
  if fused:
    a, bb = eng.dec.recs[-2], eng.dec.recs[-1]
    h = eng.z if nd == 2 else eng.dec.outs[nd - 3]
    npart = C.c_int(0)
    fn = lambda: lib.odin_bernoulli_tail_fwd_bwd(
        int(a.kind == 'deconv'), h.data_ptr(), eng.dec.w(nd - 2).data_ptr(),
        eng.dec.b(nd - 2).data_ptr(), eng.dec.w(nd - 1).data_ptr(), eng.dec.b(nd - 1).data_ptr(),
        eng.x.data_ptr(), eng.dec.outs[-1].data_ptr(), eng.dec.gouts[-2].data_ptr(),
        eng.tail_llk_part.data_ptr(), C.byref(npart), eng.tail_slab.data_ptr(), C.byref(rows),
        eng.hp(5), C.byref(eng.dec.descs[-2]), bb.desc['Cout'], st)
    if getattr(eng, 'tail_mode', None) is not None:   # the Gaussian tail (blk_planes.hip)
      fn = lambda: lib.odin_gaussian_tail_fwd_bwd(
          h.data_ptr(), eng.dec.w(nd - 2).data_ptr(), eng.dec.b(nd - 2).data_ptr(), eng.dec.w(nd - 1).data_ptr(),
          eng.dec.b(nd - 1).data_ptr(), eng.x.data_ptr(), eng.dec.outs[-1].data_ptr(), eng.dec.gouts[-2].data_ptr(),
          eng.tail_llk_part.data_ptr(), C.byref(npart), eng.tail_slab.data_ptr(), C.byref(rows), eng.hp(5),
          C.byref(eng.dec.descs[-2]), eng.in_shape[-1], eng.tail_mode, st)
    t = timeit(fn)
    fl = conv_flops(a, B) + 3 * conv_flops(bb, B)
    out.append(dict(layer=f'dec{nd - 2}+{nd - 1}:tail', op='fwd+elbo', us=t * 1e6,
                    gflop=fl * 1e-9, tflops=fl / t * 1e-12, path=lib.odin_debug_last_path().decode(),
                    mfma_gflop=conv_flops(a, B) * 1e-9))
  if neck:
    t = timeit(lambda: eng._neck_fwd(None, st))
    out.append(dict(layer='neck', op='fwd', us=t * 1e6, gflop=neck_fl[0] * 1e-9, tflops=neck_fl[0] / t * 1e-12,
                    path='neck_fwd'))
    if neck_bwd:
      A = eng._nk
      A.dy1, A.klw = eng.dec.gouts[1].data_ptr(), eng.hp(6)
      A.dz_extra = A.dloc_x = A.dscale_x = None
      A.dz, A.dp = eng.dz.data_ptr(), eng.dp.data_ptr()
      A.dh4, A.dy3, A.dx = eng.enc.gouts[ne - 1].data_ptr(), eng.enc.gouts[ne - 2].data_ptr(), eng.enc.gouts[ne - 3].data_ptr()
      A.dh4_amax = A.dy3_amax = A.dx_amax = None
      A.slab1, A.slab0, A.slabl = eng.nk_slab1.data_ptr(), eng.nk_slab0.data_ptr(), eng.nk_slabl.data_ptr()
      t = timeit(lambda: lib.odin_neck_bwd(C.byref(A), st))
      out.append(dict(layer='neck', op='bwd', us=t * 1e6, gflop=neck_fl[1] * 1e-9, tflops=neck_fl[1] / t * 1e-12,
                      path='neck_bwd'))
  if head:
    a, bb = eng.dec.recs[-2], eng.dec.recs[-1]
    Cc = eng.in_shape[-1]
    npart = C.c_int(0)
    fn = lambda: lib.odin_gaussian_head_fwd_bwd(
        eng.dec.outs[-2].data_ptr(), eng.dec.w(nd - 1).data_ptr(), eng.dec.b(nd - 1).data_ptr(), eng.x.data_ptr(),
        eng.dec.outs[-1].data_ptr(), None, eng.dec.gouts[-2].data_ptr(), eng.head_llk_part.data_ptr(),
        C.byref(npart), eng.head_slab.data_ptr(), C.byref(rows),
        eng.head_colsum.data_ptr() if eng.head_colsum is not None else None, eng.hp(5), B, eng.n_per // Cc,
        bb.desc['Cin'], Cc, eng.head_mode, ACT[a.act], None, st)
    t = timeit(fn)
    fl = 3 * conv_flops(bb, B)
    out.append(dict(layer=f'dec{nd - 1}:head', op='fwd+elbo+bwd', us=t * 1e6, gflop=fl * 1e-9,
                    tflops=fl / t * 1e-12, path='gaussian_head'))
  return out


def profile_hbm_kernels(eng, reps=48):
  """The HBM-bound kernels on the shape north_star quotes (64x64x3, batch 256) and on this
  workload's own shape: fused Bernoulli ELBO fwd+bwd (12 B/element), Gaussian head (20 B/element),
  flat Adam (28 B/parameter).  HIP events on the launch stream.  Every kernel is timed twice:
    * `achieved` / `frac`: launches ROTATE through enough distinct buffer sets that the footprint
      touched between two uses of a line exceeds the 256 MB Infinity Cache (MI355X_MICROARCH.md,
      Infinity Cache): the bytes really come from and go to HBM;
    * `cache_hot_*`: the same launch repeated on ONE buffer set (working set < 256 MB: served by the
      Infinity Cache / L2) -- what the kernel sees inside the training step when its operands were
      just produced, NOT an HBM figure."""
  import ctypes as C
  lib, dev = eng.lib, eng.device
  st = eng.stream()
  LLC = 256 * 1024 * 1024

  def timeit(fns):
    for f in fns:
      f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(reps // len(fns), 1) * len(fns)
    e0.record()
    for i in range(n):
      fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

  def entry(kernel, shape, nbytes, t_hbm, t_hot, nsets, t_probe=None, t_own=None):
    extra = {}
    if t_own is not None:
      # the ceiling of THIS part for a stream of this length and shape: a hand-written kernel with the same traffic
      # (two arrays read, one written) and no arithmetic, persistent grid, streaming stores -- the fastest of the launch
      # shapes swept by tools/elbo_ceiling.py (odin_debug_stream_probe, profiles/r04_elbo_stream_sweep.txt)
      extra.update(own_stream_gbs=round(nbytes / t_own * 1e-9, 1), own_stream_frac_of_hbm_peak=round(nbytes / t_own * 1e-9 / PEAK_HBM_GBS, 4),
                   frac_of_own_stream=round(t_own / t_hbm, 4),
                   own_stream='odin_debug_stream_probe variant 8 (out = a + b, 512 workgroups grid-stride, non-temporal stores)')
    if t_probe is not None:
      # a plain elementwise launch with the SAME traffic (2 arrays read, 1 written, same sizes, same rotation):
      # what a stream of this length reaches on this box from cold HBM -- short launches pay their ramp
      extra.update(same_traffic_probe_gbs=round(nbytes / t_probe * 1e-9, 1),
                   same_traffic_probe='torch.add(a, b, out=c) over the same rotating buffer sets',
                   frac_of_probe=round(t_probe / t_hbm, 4))
    return dict(extra, bound='hbm', kernel=kernel, shape=shape, achieved=round(nbytes / t_hbm * 1e-9, 1),
                peak=PEAK_HBM_GBS, unit='GB/s', frac=round(nbytes / t_hbm * 1e-9 / PEAK_HBM_GBS, 4),
                us_per_launch=round(t_hbm * 1e6, 2), mbytes_per_launch=round(nbytes * 1e-6, 2),
                buffer_sets=nsets, footprint_mb=round(nsets * nbytes * 1e-6, 1),
                cache_hot_gbs=round(nbytes / t_hot * 1e-9, 1), cache_hot_us=round(t_hot * 1e6, 2),
                cache_hot_frac_of_hbm_peak=round(nbytes / t_hot * 1e-9 / PEAK_HBM_GBS, 4))

  def nsets_for(nbytes):
    return max(2, int(math.ceil(1.25 * LLC / nbytes)))

  out = []
  npart = C.c_int(0)
  sc = eng.hp(5)
  # [0]: the shape north_star quotes (37.75 MB per launch); [1]: the same images at config 4's batch 512 (75.5 MB):
  # a 38 MB launch lasts ~7 us, of which the part's ramp to full bandwidth is a visible share -- the longer launch
  # shows how much; then this workload's own shape
  shapes = [(256, 64 * 64, 3), (512, 64 * 64, 3)]
  own = (eng.B, int(np.prod(eng.in_shape[:-1])), eng.in_shape[-1])
  if own not in shapes:
    shapes.append(own)
  for (B, npix, Cc) in shapes:
    n = npix * Cc
    ns = nsets_for(12.0 * B * n)
    sets = [(torch.randn(B, n, device=dev), torch.rand(B, n, device=dev), torch.empty(B, n, device=dev))
            for _ in range(ns)]
    lib.odin_elbo_bernoulli_fwd_bwd(None, None, None, None, None, B, n, C.byref(npart), None)
    part = torch.empty(B * npart.value, device=dev)
    mk = lambda lg, x, dl: (lambda: lib.odin_elbo_bernoulli_fwd_bwd(
        lg.data_ptr(), x.data_ptr(), part.data_ptr(), dl.data_ptr(), sc, B, n, C.byref(npart), st))
    fns = [mk(*t) for t in sets]
    probe = [(lambda a=a, b=b, c=c: torch.add(a, b, out=c)) for (a, b, c) in sets]
    own = [(lambda a=a, b=b, c=c: lib.odin_debug_stream_probe(a.data_ptr(), b.data_ptr(), c.data_ptr(), B * n, 8, 512, st))
           for (a, b, c) in sets] if (B * n) % 4 == 0 else None
    out.append(entry('elbo_bernoulli_fwd_bwd', [B, npix, Cc], 12.0 * B * n, timeit(fns), timeit(fns[:1]), ns,
                     timeit(probe), timeit(own) if own else None))
    del probe, own
    del sets, fns
    if (B, npix, Cc) == (512, 64 * 64, 3):
      continue   # (the Gaussian head is priced on the other shapes)
    ns = nsets_for(20.0 * B * n)
    sets = [(torch.randn(B, npix, 2 * Cc, device=dev), torch.rand(B, n, device=dev),
             torch.empty(B, npix, 2 * Cc, device=dev)) for _ in range(ns)]
    lib.odin_elbo_gaussian_fwd_bwd(None, None, None, None, None, B, npix, Cc, 1, C.byref(npart), None)
    part = torch.empty(B * npart.value, device=dev)
    mk = lambda h, x, dh: (lambda: lib.odin_elbo_gaussian_fwd_bwd(
        h.data_ptr(), x.data_ptr(), part.data_ptr(), dh.data_ptr(), sc, B, npix, Cc, 1, C.byref(npart), st))
    fns = [mk(*t) for t in sets]
    out.append(entry('elbo_gaussian_fwd_bwd(softplus1)', [B, npix, Cc], 20.0 * B * n, timeit(fns),
                     timeit(fns[:1]), ns))
    del sets, fns
  for n in (eng.params.numel(), 4012004):  # this model; the FactorVAE discriminator (a18)
    ns = nsets_for(28.0 * n)
    sets = [(torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3, torch.zeros(n, device=dev),
             torch.zeros(n, device=dev)) for _ in range(ns)]
    mk = lambda th, g, m, v: (lambda: lib.odin_adam_step_flat(th.data_ptr(), g.data_ptr(), m.data_ptr(),
                                                              v.data_ptr(), n, eng.hp(0), None, 0.0, None, st))
    fns = [mk(*t) for t in sets]
    out.append(entry('adam_step_flat', [n], 28.0 * n, timeit(fns), timeit(fns[:1]), ns))
    del sets, fns
  torch.cuda.empty_cache()
  return out


def dominant_rooflines(ops):
  """(roofline of the dominant kernel, priced on the pipe it executes on; conv/dense stack summary)."""
  is_split = lambda o: o.get('path', '').endswith('(f16x2)')
  dom = max(ops, key=lambda o: o['us'])
  if is_split(dom):
    bf16_gflop = 3.0 * dom.get('mfma_gflop', dom['gflop'])
    roof = dict(bound='mfma', kernel=f"{dom['layer']}:{dom['op']}", path=dom.get('path'),
                achieved=round(bf16_gflop / dom['us'] * 1e3, 3), peak=PEAK_MFMA_BF16_TFLOPS,
                unit='TFLOP/s (f16 FLOPs executed)',
                frac=round(bf16_gflop / dom['us'] * 1e3 / PEAK_MFMA_BF16_TFLOPS, 4),
                fp32_equivalent_frac=round(dom['tflops'] / PEAK_MFMA_F32_TFLOPS, 4),
                us_per_launch=round(dom['us'], 2))
  else:
    roof = dict(bound='mfma', kernel=f"{dom['layer']}:{dom['op']}", path=dom.get('path'),
                achieved=round(dom['tflops'], 3), peak=PEAK_MFMA_F32_TFLOPS, unit='TFLOP/s',
                frac=round(dom['tflops'] / PEAK_MFMA_F32_TFLOPS, 4), us_per_launch=round(dom['us'], 2))
  conv_us = sum(o['us'] for o in ops)
  conv_gf = sum(o['gflop'] for o in ops)
  stack = dict(us=round(conv_us, 1), gflop=round(conv_gf, 3), tflops=round(conv_gf / conv_us * 1e3, 3),
               frac=round(conv_gf / conv_us * 1e3 / PEAK_MFMA_F32_TFLOPS, 4))
  return roof, stack


def fit_throughput(device, iters=600):
  """The API north_star names: `BetaVAE(**get_networks('dsprites')).fit(...)` end to end (Python loop, per-step
  hyper-parameter copy, graph replay, NaN polling, metrics) on (a) the on-device uint8 pipeline
  `DeviceImageDataset(out=vae.input_buffer(256))` and (b) a plain float32 tensor resident in HBM
  (odin/networks/base_networks.py:642-812, odin/training/trainer.py:536-738)."""
  from odin_ai_amd.data import DeviceImageDataset
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import BetaVAE
  B = 256
  g = torch.Generator(device='cpu').manual_seed(7)
  imgs = (torch.rand(8192, 64, 64, 1, generator=g) < 0.05).to(torch.uint8)
  out = {}
  for tag in ('device_dataset', 'float_tensor'):
    vae = BetaVAE(beta=4.0, device=device, seed=3, **get_networks('dsprites'))
    init_params_(vae._engine(B), seed=3)
    if tag == 'device_dataset':
      train = DeviceImageDataset(imgs, batch_size=B, normalize='probs', premul=255.0, device=device,
                                 out=vae.input_buffer(B))
    else:
      train = imgs.to(device).float().clamp_(1e-6, 1 - 1e-6)
    kw = dict(batch_size=B, learning_rate=1e-3, global_clipnorm=100.0, compile_graph=True)
    vae.fit(train, max_iter=60, **kw)   # graph capture + warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vae.fit(train, max_iter=iters, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert not vae.nan_flag
    out[tag] = dict(images_per_sec=round(B * iters / dt, 1), ms_per_step=round(dt / iters * 1e3, 4), iters=iters)
    del vae, train
  torch.cuda.empty_cache()
  return out


def north_star_3ch(device, steps=50):
  """The shape north_star states its roofline targets on -- 64x64x3 beta-VAE (beta = 4) at batch 256 =
  the Shapes3D networks (image_networks.py:560-597) -- measured in the SAME process as the headline
  workload: step time (HIP-graph replay), conv/dense stack, dominant kernel.  (The ELBO kernel on this
  shape is `elbo_kernel` / `hbm_kernels[0]` of the main line.)"""
  from odin_ai_amd.engine import VAEEngine
  from odin_ai_amd.networks import get_networks
  nets = get_networks('shapes3d')
  enc, dec = nets['encoder'].layers, nets['decoder'].layers
  in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
  B = 256
  eng = VAEEngine(enc, dec, in_shape, zdim, B, device, observation=nets['observation'].posterior, seed=3)
  init_params_(eng, seed=3)
  xb = eng.input_buffer()
  xb.copy_(synthetic_batch('shapes3d', B, in_shape, device, seed=103))
  step = lambda: eng.train_step(xb, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=True)
  for _ in range(30):
    step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    out = step()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  assert math.isfinite(out[0].item()) and eng.flag.item() == 0
  roof, stack = dominant_rooflines(profile_ops(eng))
  res = dict(workload='shapes3d_betavae_b256 (64x64x3, beta=4, batch 256)', images_per_sec=round(B * steps / dt, 1),
             ms_per_step=round(dt / steps * 1e3, 4), steps=steps, conv_stack=stack, roofline=roof,
             in_step_conv_frac=round(stack['gflop'] / (dt / steps) * 1e-3 / PEAK_MFMA_F32_TFLOPS, 4))
  del eng
  torch.cuda.empty_cache()
  return res


def exact_fp32_step(device, workload='dsprites_betavae_b256', steps=50):
  """The same step with EVERY layer on the exact fp32 matrix-core kernels (ODIN_EXACT_FP32=1: no f16-plane kernel; the
  arithmetic of the reference's fp32 Keras layers, image_networks.py:460-513), timed in the same process: what the
  two-plane substitution of the headline line is worth.  The switch is read when a launch is issued, so it is set
  while this engine is built, captured and replayed, and removed afterwards."""
  from odin_ai_amd.engine import VAEEngine
  from odin_ai_amd.networks import get_networks
  ds, kw, B, beta, kind = WORKLOADS[workload]
  nets = get_networks(ds, **kw)
  enc, dec = nets['encoder'].layers, nets['decoder'].layers
  in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
  had = os.environ.get('ODIN_EXACT_FP32')
  os.environ['ODIN_EXACT_FP32'] = '1'
  try:
    eng = VAEEngine(enc, dec, in_shape, zdim, B, device, observation=nets['observation'].posterior, seed=5)
    init_params_(eng, seed=5)
    xb = eng.input_buffer()
    xb.copy_(synthetic_batch(workload, B, in_shape, device, seed=105))
    step = lambda: eng.train_step(xb, None, lr=1e-3, beta=beta, global_clipnorm=100.0, use_graph=True)
    for _ in range(30):
      step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert math.isfinite(out[0].item()) and eng.flag.item() == 0
    ops = profile_ops(eng)
    paths = sorted({o.get('path', '') for o in ops})
    assert not any(p.endswith('(f16x2)') for p in paths), paths
    roof, stack = dominant_rooflines(ops)
  finally:
    if had is None:
      del os.environ['ODIN_EXACT_FP32']
    else:
      os.environ['ODIN_EXACT_FP32'] = had
  res = dict(workload=workload, ms_per_step=round(dt / steps * 1e3, 4), images_per_sec=round(B * steps / dt, 1),
             steps=steps, kernel_families=paths, conv_stack=stack, roofline=roof,
             in_step_conv_frac=round(stack['gflop'] / (dt / steps) * 1e-3 / PEAK_MFMA_F32_TFLOPS, 4),
             note='ODIN_EXACT_FP32=1: v_mfma_f32_32x32x2_f32 / 16x16x4_f32 everywhere (IEEE fp32 products); roofline = its '
                  'dominant launch against the dense fp32 MFMA peak (157.3 TFLOP/s), conv_stack = stand-alone sum of its '
                  'conv / dense launches, in_step_conv_frac = the same FLOPs over the step time')
  del eng
  torch.cuda.empty_cache()
  return res


def _free_port():
  import socket
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def launch_ranks(n, argv):
  """`python bench.py --gpus N` without a launcher: start N fresh children (one process per
  GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun would set them) BEFORE this
  process touches the GPU.  The parent never initialises HIP; rank 0's stdout (the one JSON
  line) passes through; exit status is non-zero if any rank fails."""
  import subprocess
  env0 = dict(os.environ)
  env0.setdefault('MASTER_ADDR', '127.0.0.1')
  env0.setdefault('MASTER_PORT', str(_free_port()))
  env0.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env0['WORLD_SIZE'] = str(n)
  env0['LOCAL_WORLD_SIZE'] = str(n)
  procs = []
  for r in range(n):
    env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                  stdout=json_out() if r == 0 else subprocess.DEVNULL))
  rc = 0
  pending = list(procs)
  while pending:
    for p in list(pending):
      r = p.poll()
      if r is None:
        continue
      pending.remove(p)
      if r != 0 and rc == 0:
        rc = r
        for q in pending:  # one rank died: the others would hang in the next collective
          q.terminate()
    time.sleep(0.05)
  return rc


def dry_run(args, world, rank):
  """Launcher / rendezvous self-test without a GPU: gloo process group, one all-reduce, the
  JSON line of rank 0 names the ranks that answered.  No kernel runs; the metric name says so."""
  import torch.distributed as dist
  seen = 1
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    t = torch.ones(1)
    dist.all_reduce(t)
    seen = int(t.item())
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    print(json.dumps(dict(metric='launcher dry run (no kernels executed)', value=0.0,
                          unit='images/sec', n_gpus=world, steps=args.steps, warmup=args.warmup,
                          rccl=dict(ranks_seen=seen, backend='gloo'))), file=json_out(), flush=True)


def time_allreduce(eng, reps=20):
  """Gradient-bucket all-reduce alone (RCCL, on the stream the step uses), HIP events."""
  import torch.distributed as dist
  for _ in range(3):
    eng.allreduce()
  torch.cuda.synchronize()
  dist.barrier()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    eng.allreduce()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


_JSON_OUT = None


def json_out():
  """The process's ORIGINAL stdout, for the one JSON line.  File descriptor 1 itself is pointed at
  stderr for the rest of the run: RCCL prints a version banner to fd 1 from C, which would otherwise
  land next to the JSON line the driver parses."""
  global _JSON_OUT
  if _JSON_OUT is None:
    sys.stdout.flush()
    _JSON_OUT = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
  return _JSON_OUT


def main():
  json_out()
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--workload', default='dsprites_betavae_b256', choices=sorted(WORKLOADS))
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-steps', type=int, default=25)
  ap.add_argument('--no-graph', action='store_true')
  ap.add_argument('--no-north-star-3ch', action='store_true',
                  help='skip the 64x64x3 beta-VAE (Shapes3D networks) measurement added to the default line')
  ap.add_argument('--no-fit', action='store_true', help='skip the fit()-level throughput measurement')
  ap.add_argument('--no-exact-fp32', action='store_true',
                  help='skip the second engine that times the step on the exact fp32 kernels (ODIN_EXACT_FP32=1)')
  ap.add_argument('--profile-ops', action='store_true', help='print a per-kernel timing table')
  ap.add_argument('--dry-run', action='store_true',
                  help='launcher / rendezvous self-test on CPU (gloo): no kernels, no GPU')
  ap.add_argument('--dp-buckets', type=int, default=0,
                  help='gradient buckets of the data-parallel step (2: the decoder bucket is all-reduced on a '
                  'side stream beside the encoder backward; default: 2 from 4 ranks up when the bucket is >= 8 MB, else 1)')
  ap.add_argument('--engine-opt', action='append', default=[], metavar='KEY=VALUE',
                  help='a VAEEngine keyword argument of the main engine (A/B runs: hyper_ring=False, act_words=False, '
                  'fuse_norm=False, overlap_wgrad=small, early_reduce=True, defer_wgrad=True, neck=False ...)')
  ap.add_argument('--overlap-disc', action='store_true',
                  help="A/B (FactorVAE): the discriminator's pass over z on a side stream beside the decoder instead of "
                  'behind the VAE forward pass on the same stream (measured slower: 0.755 vs 0.735 ms)')
  ap.add_argument('--no-blk', action='store_true',
                  help='A/B: the 4x4 / stride-2 layers that fit no row-window plane kernel on igemm_h.hip (round 5) instead '
                  'of the block-window kernels of blk_planes.hip (odin_debug_blk_planes(0); the audio VAE)')
  ap.add_argument('--no-fused-disc', action='store_true',
                  help="A/B (FactorVAE): the discriminator head's forward, loss, weight and data gradient and permute_dims as "
                  'launches of their own (round 5) instead of odin_disc_head_fwd_bwd / odin_random_permute_dims')
  ap.add_argument('--no-mel-r16', action='store_true',
                  help='A/B (speech): the n_fft = 512 front-end on the general radix-4 kernel (round 5) instead of the register '
                  'radix-16 one (mel.hip: stft_mel512_kernel)')
  ap.add_argument('--no-smallc-planes', action='store_true',
                  help='A/B: the RGB first layer forward on fp32 MFMAs (round 5) instead of two f16 planes (smallc_conv.hip)')
  ap.add_argument('--no-dense-hw', action='store_true',
                  help='A/B: Dense weight / data gradients with both widths >= 256 on the 32 x 32 tiles straight from L2 (round 4) '
                  'instead of the LDS-staged 64 x 64 tiles (dense_h.hip: dense_hw, dense_hd; FactorVAE, CelebA)')
  ap.add_argument('--force-dist', action='store_true',
                  help='initialise the RCCL process group even at world size 1, so that the '
                  'data-parallel step (graph A, RCCL all-reduce, graph B) runs on a 1-GPU box')
  args = ap.parse_args()

  import ast
  eopts = {}
  if args.no_blk:
    from odin_ai_amd import _lib as _l
    _l.load().odin_debug_blk_planes(0)
  if args.no_mel_r16:
    from odin_ai_amd import _lib as _l
    _l.load().odin_debug_mel_r16(0)
  if args.no_smallc_planes:
    from odin_ai_amd import _lib as _l
    _l.load().odin_debug_smallc_planes(0)
  if args.no_dense_hw:
    from odin_ai_amd import _lib as _l
    _l.load().odin_debug_dense_hw_min_tiles(1 << 30)
  for kv in args.engine_opt:
    k, _, v = kv.partition('=')
    try:
      eopts[k] = ast.literal_eval(v)
    except (ValueError, SyntaxError):
      eopts[k] = v
  if args.dp_buckets:
    eopts['dp_buckets'] = args.dp_buckets
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    # no launcher around us: become one.  Nothing in this process has touched the GPU yet.
    sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line '
          f'whose n_gpus differs from what was asked for', file=sys.stderr)
    sys.exit(2)
  if args.dry_run:
    return dry_run(args, world, rank)
  use_dist = world > 1 or args.force_dist
  assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
  ndev = torch.cuda.device_count()
  if world > ndev:
    print(f'bench.py: {world} ranks but {ndev} GPU(s) visible (one process per GPU)',
          file=sys.stderr)
    sys.exit(2)
  dev_index = local_rank
  torch.cuda.set_device(dev_index)
  device = torch.device('cuda', dev_index)
  if use_dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(_free_port()))
    dist.init_process_group(os.environ.get('ODIN_DIST_BACKEND', 'nccl'), rank=rank,
                            world_size=world, device_id=device)

  from odin_ai_amd.engine import VAEEngine
  from odin_ai_amd.networks import get_networks
  if os.environ.get('ODIN_BENCH_IH_MIN_GF'):
    # diagnostics: the launch size from which convolutions run on the two-plane implicit-GEMM kernels (igemm_h.hip)
    from odin_ai_amd import _lib as _L
    _L.load().odin_debug_igemm_h_min_flop(float(os.environ['ODIN_BENCH_IH_MIN_GF']) * 1e9)
  ds, kw, B, beta, kind = WORKLOADS[args.workload]
  nets = get_networks(ds, **kw)
  enc, dec = nets['encoder'].layers, nets['decoder'].layers
  in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
  lr = 1e-3
  use_graph = not args.no_graph
  fv = None
  if kind == 'factor':
    # FactorVAE through the model API (vae.py): x split 128 + 128, VAE step with the
    # discriminator's TC estimate, discriminator step with its own Adam; one graph per iteration
    from odin_ai_amd.vae import FactorVAE
    fv = FactorVAE(device=device, seed=1 + rank, **nets)
    fv.force_dp = use_dist
    fv.engine_options = dict(eopts)
    fv.overlap_disc = bool(args.overlap_disc)
    fv.fuse_discriminator = not args.no_fused_disc
    eng = fv._engine(B // 2)
    fv._discriminator(B // 2)
    beta = 1.0
  else:
    eng = VAEEngine(enc, dec, in_shape, zdim, B, device, observation=nets['observation'].posterior,
                    tc=kind if kind == 'betatc' else None, world_size=world, seed=1 + rank,
                    force_dp=use_dist, **eopts)
  init_params_(eng, seed=1 + 1000 * rank)  # rank 0's weights win: broadcast below
  rccl = None
  if use_dist:
    from odin_ai_amd.dist import broadcast_parameters
    broadcast_parameters(eng.params, src=0, force=True)
    chk = eng.params.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert lo.item() == hi.item(), 'parameter broadcast failed: replicas differ'
    seen = torch.ones(1, device=device)
    dist.all_reduce(seen)
    rccl = dict(ranks_seen=int(seen.item()), backend=dist.get_backend(),
                bucket_bytes=eng.grads.numel() * 4)
  if kind == 'speech':
    # raw audio resident in HBM; every step runs the front-end (pre-emphasis, STFT, mel, dB) and
    # hands the first 96 frames to the VAE step
    from odin_ai_amd.mel import MelsSpecExtractor
    g = torch.Generator(device='cpu').manual_seed(100 + rank)
    t = torch.arange(8000) / 8000.0
    y = (0.1 * torch.randn(B, 8000, generator=g) +
         0.5 * torch.sin(2 * math.pi * (200.0 + 1500.0 * t[None] * torch.rand(B, 1, generator=g)) * t[None])
         ).to(device)
    ex = MelsSpecExtractor(device=device, unit_range=True)
    xb = eng.input_buffer()
    T = in_shape[0]

    def step():
      ex(y, out=xb)   # the first T frames straight into the buffer the step graph reads
      return eng.train_step(xb, None, lr=lr, beta=beta, global_clipnorm=100.0, use_graph=use_graph)
  elif fv is not None:
    x = synthetic_batch(args.workload, B, in_shape, device, seed=100 + rank)
    if use_graph:
      # (resident in the buffer the iteration graph reads, as the plain-VAE workloads: no per-iteration copy)
      xb = fv.input_buffer(B)
      xb.copy_(x)
      x = xb

    def step():
      loss, _ = fv.optimize(x, learning_rate=lr, global_clipnorm=100.0, use_graph=use_graph, snapshot=False)
      return eng.out4
  else:
    x = synthetic_batch(args.workload, B, in_shape, device, seed=100 + rank)
    if use_graph:
      # the batch is resident in HBM in the buffer the step graph reads (what an on-device input
      # pipeline would fill), so no per-step device-to-device copy sits in the timed region
      xb = eng.input_buffer()
      xb.copy_(x)
      x = xb

    def step():
      return eng.train_step(x, None, lr=lr, beta=beta, global_clipnorm=100.0, use_graph=use_graph)

  # Steady-state conditioning before the W warm-up steps: on a freshly started process the first
  # ~100 ms of replays can run 20-30 % slow (clock / power ramp from idle; measured as one slow
  # first block of 100 steps in about 1 process out of 8, steady afterwards), so the GPU is kept
  # busy for ~0.6 s first.  Untimed, like the graph capture itself.
  step()
  torch.cuda.synchronize()
  if use_dist:
    # every step contains a collective: all ranks must run the SAME number of steps
    for _ in range(400):
      step()
    torch.cuda.synchronize()
  else:
    t_pw = time.perf_counter()
    blocks = 0
    while time.perf_counter() - t_pw < 0.6:
      for _ in range(20):
        step()
      torch.cuda.synchronize()
      blocks += 1
      if os.environ.get('ODIN_BENCH_TRACE_FLAG') and eng.flag.item() != 0:
        print(f'bench.py: non-finite gradients first seen in conditioning block {blocks} (20 steps each)',
              file=sys.stderr)
        describe_nonfinite(eng)
        break
  for _ in range(args.warmup):
    step()
  torch.cuda.synchronize()
  if use_dist:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    out = step()
  torch.cuda.synchronize()
  dt_local = time.perf_counter() - t0   # this rank's own K steps, before the closing barrier
  if use_dist:
    dist.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if use_dist:
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = tt.item()
    # spread of the ranks' own clocks (a slow GPU or a late rank shows here, not in the max-over-ranks value)
    tl = torch.tensor([dt_local], device=device, dtype=torch.float64)
    tlo, thi = tl.clone(), tl.clone()
    dist.all_reduce(tlo, op=dist.ReduceOp.MIN)
    dist.all_reduce(thi, op=dist.ReduceOp.MAX)
    rccl['rank_ms_per_step'] = dict(min=round(tlo.item() / args.steps * 1e3, 4), max=round(thi.item() / args.steps * 1e3, 4))
    # replicas must still agree after the timed steps (same all-reduced gradients, same Adam)
    chk = eng.params.double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    rccl['replicas_in_sync'] = bool(lo.item() == hi.item())
    rccl['allreduce_us'] = round(time_allreduce(eng), 2)
    rccl['transport'] = ('RCCL through the C ABI (odin_allreduce_flat, own communicator)'
                         if eng._comm().native else 'torch.distributed ' + dist.get_backend())
    rccl['dp_buckets'] = eng.dp_buckets
    rccl['library'] = eng.lib.odin_comm_library().decode() if eng._comm().native else 'torch.distributed'
    sgs = list(getattr(eng, '_graphs', {}).values()) + [v[0] for v in getattr(fv, '_fgraphs', {}).values()] \
        if use_graph else []
    if sgs:
      rccl['step_segments'] = ''.join('G' if k == 'k' else 'c' for k, _ in sgs[-1].segs)  # G = HIP graph, c = collective
  loss = out[0].item()
  assert math.isfinite(loss), 'training diverged'
  if eng.flag.item() != 0:
    describe_nonfinite(eng)
  assert eng.flag.item() == 0, 'non-finite gradients were skipped during the timed region'

  if rank != 0:
    dist.destroy_process_group()
    return

  # ---- roofline of the dominant kernel (per-launch, HIP events on the launch stream) ----
  # Kernels whose path ends in "(bf16x3)" carry fp32 operands through the bf16 matrix pipe as three
  # exact planes (6 bf16 MFMAs per 16 k-values): they are priced against the bf16 peak with the bf16
  # FLOPs they execute (6 x the fp32 FLOPs of the convolution) in `roofline_split`;
  # `roofline` is the dominant kernel among those that compute in fp32 MFMAs.
  ops = profile_ops(eng)
  is_split = lambda o: o.get('path', '').endswith('(f16x2)')
  fp32_ops = [o for o in ops if not is_split(o)] or ops
  dom = max(fp32_ops, key=lambda o: o['us'])
  roofline = dict(bound='mfma', kernel=f"{dom['layer']}:{dom['op']}", path=dom.get('path'),
                  achieved=round(dom['tflops'], 3), peak=PEAK_MFMA_F32_TFLOPS, unit='TFLOP/s',
                  frac=round(dom['tflops'] / PEAK_MFMA_F32_TFLOPS, 4), traffic=None,
                  us_per_launch=round(dom['us'], 2), gflop_per_launch=round(dom['gflop'], 4),
                  note='fp32 FLOPs / dense fp32 MFMA peak (v_mfma_f32_16x16x4_f32 / 32x32x2_f32)')
  pmc = {}
  try:
    import json as _json
    import glob as _glob
    # the newest round's PMC table (tools/pmc_traffic.py <tag> after tools/pmc.sh <tag>)
    _cand = sorted(_glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                           'r*_pmc_traffic.json')))
    pmc_file = _cand[-1]
    pmc = _json.load(open(pmc_file))
  except (OSError, ValueError, IndexError):
    pmc_file = None

  def attach_traffic(obj):
    # HBM traffic from the committed PMC passes (FETCH_SIZE / WRITE_SIZE cannot be collected inside
    # this process); null when no measurement for this kernel on this workload is on file
    ent = pmc.get(obj['kernel'])
    if ent is not None and args.workload == 'dsprites_betavae_b256':
      obj['traffic'] = ent['traffic_bytes']
      obj['traffic_unit'] = 'bytes'
      obj['algorithmic_bytes'] = ent.get('algorithmic_bytes')
      obj['traffic_source'] = 'profiles/' + os.path.basename(pmc_file)
      obj['traffic_measured_in_this_run'] = False  # (PMC counters cannot be collected inside this process)
      if ent.get('us_in_graph'):
        # what the STEP pays for this launch: its average duration inside the captured graph under rocprofv3
        # (profiles/<round>_kernel_stats.csv), a few us longer than the stand-alone HIP-event time of this run
        obj['us_in_graph'] = ent['us_in_graph']
        obj['us_in_graph_source'] = ent.get('us_in_graph_source')
        if ent.get('algorithmic_bytes'):
          obj['frac_in_graph'] = round(ent['algorithmic_bytes'] / (ent['us_in_graph'] * 1e-6) * 1e-9 / PEAK_HBM_GBS, 4)

  attach_traffic(roofline)
  roofline_split = None
  split_ops = [o for o in ops if is_split(o)]
  if split_ops:
    # A plane kernel executes 3 f16 MFMAs per 16 k-values (fp32 operands as two f16 planes): its matrix-pipe floor
    # is executed FLOPs / the dense f16 peak, its memory floor algorithmic bytes / 8 TB/s.  Since the two-plane form
    # the MEMORY floor is the higher one for the big layers (arithmetic intensity ~146 executed FLOP per byte against
    # a machine balance of 314): the kernel is priced against the roof that binds it, the other leg is kept beside.
    so = max(split_ops, key=lambda o: o['us'])
    f16_gflop = 3.0 * so.get('mfma_gflop', so['gflop'])
    mfma_leg = dict(achieved=round(f16_gflop / so['us'] * 1e3, 3), peak=PEAK_MFMA_BF16_TFLOPS,
                    unit='TFLOP/s (f16 FLOPs executed)', frac=round(f16_gflop / so['us'] * 1e3 / PEAK_MFMA_BF16_TFLOPS, 4),
                    gflop_per_launch=round(f16_gflop, 4),
                    fp32_equivalent_tflops=round(so['tflops'], 3),
                    fp32_equivalent_frac_of_fp32_mfma_peak=round(so['tflops'] / PEAK_MFMA_F32_TFLOPS, 4))
    roofline_split = dict(bound='mfma', kernel=f"{so['layer']}:{so['op']}", path=so.get('path'), traffic=None,
                          us_per_launch=round(so['us'], 2), **mfma_leg,
                          note='fp32 operands as 2 f16 planes: 3 v_mfma_f32_32x32x16_f16 per 16 '
                               'k-values, priced against the dense f16 MFMA peak')
    attach_traffic(roofline_split)
    alg = roofline_split.get('algorithmic_bytes')
    if alg:
      t_hbm, t_mfma = alg / (PEAK_HBM_GBS * 1e9), f16_gflop * 1e9 / (PEAK_MFMA_BF16_TFLOPS * 1e12)
      hbm_leg = dict(achieved=round(alg / (so['us'] * 1e-6) * 1e-9, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                     frac=round(alg / (so['us'] * 1e-6) * 1e-9 / PEAK_HBM_GBS, 4))
      roofline_split['floors_us'] = dict(hbm=round(t_hbm * 1e6, 2), mfma=round(t_mfma * 1e6, 2))
      if t_hbm >= t_mfma:
        roofline_split.update(bound='hbm', mfma_leg=mfma_leg, **hbm_leg)
        roofline_split['note'] = ('memory-bound by the roofline model: algorithmic bytes / 8 TB/s exceeds executed f16 '
                                  'FLOPs / dense f16 peak (fp32 operands as 2 f16 planes, 3 MFMAs per 16 k-values); '
                                  'the matrix-pipe leg is in mfma_leg')
      else:
        roofline_split['hbm_leg'] = hbm_leg
  # `roofline` is THE dominant kernel of the step, priced on the pipe it executes on; when that is a
  # bf16-plane kernel the dominant fp32-MFMA kernel is kept beside it as `roofline_fp32`
  roofline_fp32 = None
  if roofline_split is not None and roofline_split['us_per_launch'] > roofline['us_per_launch']:
    roofline_fp32, roofline = roofline, roofline_split
  conv_us = sum(o['us'] for o in ops)
  conv_gf = sum(o['gflop'] for o in ops)
  stack = dict(us=round(conv_us, 1), gflop=round(conv_gf, 3),
               tflops=round(conv_gf / conv_us * 1e3, 3),
               frac=round(conv_gf / conv_us * 1e3 / PEAK_MFMA_F32_TFLOPS, 4))
  hbm = profile_hbm_kernels(eng) if eng.in_shape[-1] in (1, 3) else [None]
  mel_kernel = None
  if kind == 'speech':
    # the front-end launch alone (stft_mel_f64_kernel: float64 FFT + mel + dB, one workgroup per utterance):
    # algorithmic bytes = the samples read once + the [B, T, n_mels] patch written once
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
      ex(y, out=xb)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
      ex(y, out=xb)
    e1.record()
    torch.cuda.synchronize()
    t_mel = e0.elapsed_time(e1) / 50 * 1e-3
    nbytes = 4.0 * y.numel() + 4.0 * xb.numel()
    mel_kernel = dict(bound='hbm', kernel='stft_mel_db_frames (float64 STFT -> mel -> dB, unit range)',
                      shape=[B, int(y.shape[1])], achieved=round(nbytes / t_mel * 1e-9, 1), peak=PEAK_HBM_GBS,
                      unit='GB/s', frac=round(nbytes / t_mel * 1e-9 / PEAK_HBM_GBS, 4),
                      us_per_launch=round(t_mel * 1e6, 2), mbytes_per_launch=round(nbytes * 1e-6, 2),
                      note='compute-bound in float64 (the precision of the reference): B workgroups of 256 threads, '
                           'n_fft 512: two radix-16 passes in registers, 16 lanes per frame (mel.hip); priced against HBM as north_star asks')
  if args.profile_ops:
    for o in ops:
      print(f"# {o['layer']:14s} {o['op']:6s} {o['us']:9.1f} us {o['gflop']:8.3f} GF "
            f"{o['tflops']:7.2f} TF/s  {o.get('path', '')}", file=sys.stderr)
    print(f"# conv/dense stack: {stack}", file=sys.stderr)

  # ---- CPU baseline: the same step as a torch-CPU fp32 port, all host cores --------------
  cpu = None
  if not args.no_cpu_baseline and world == 1:
    from oracle.torch_ref import TorchFactorTrainer, TorchTrainer, TorchVAE
    # torch-CPU scales to ~32 threads on this step (measured 8..128 on the GPU box's host:
    # 32 is the fastest), so the baseline uses min(32, cores) threads and reports that count
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    model = TorchVAE(enc, dec, in_shape, zdim, observation=nets['observation'].posterior,
                     beta=beta, tc_beta=beta if kind == 'betatc' else None, dtype=torch.float32)
    P = {k: v.detach().cpu().numpy() for k, v in eng.param_views().items()}
    n_cpu = args.cpu_steps
    if kind == 'factor':
      # both optimisers of one FactorVAE iteration (VAE step on 128 images with the discriminator's TC term,
      # discriminator step on the other 128 + the permuted codes)
      dp = fv._discriminator(B // 2).disc
      DP = {k: v.detach().cpu().numpy() for k, v in dp.layout.views(dp.params).items()}
      dlayers = [('dense', int(s2[1]), 'relu' if i + 1 < len(dp.recs) else 'linear')
                 for i, (_, s2, _) in enumerate(e for e in dp.layout.entries if e[0][-1] == 'w')]
      tr = TorchFactorTrainer(model, P, dlayers, DP, lr=lr, tc_coef=fv.tc_coef, threads=cores)
      xc = x.cpu()
      e1, e2 = torch.randn(B // 2, zdim), torch.randn(B // 2, zdim)
      perm = torch.stack([torch.randperm(B // 2) for _ in range(zdim)], 1)
      one = lambda: tr.step(xc, e1, e2, perm)
      what = 'FactorVAE iterations (VAE step + discriminator step)'
    elif kind == 'speech':
      # the front-end on the host (numpy float64 restatement of odin/preprocessing/signal.py, the reference's
      # own arithmetic) + the conv-VAE step
      from oracle import mel_oracle as mo
      tr = TorchTrainer(model, P, lr=lr, threads=cores)
      yc = y.cpu().numpy().astype(np.float64)
      ec = torch.randn(B, zdim)
      T = in_shape[0]

      def one():
        # dB relative to the utterance's maximum, top_db = 80 -> [0, 1] (the mode-2 output of odin_stft_mel_db)
        mel = np.stack([mo.mel_frontend(u)[:T] for u in yc])
        mel = (mel - mel.max(axis=(1, 2), keepdims=True)) / 80.0 + 1.0
        return tr.step(torch.tensor(mel.reshape(B, T, in_shape[1], 1), dtype=torch.float32), ec)
      what = 'speech steps (numpy STFT -> mel -> dB front-end + conv-VAE step)'
      n_cpu = max(2, args.cpu_steps // 8)
    else:
      tr = TorchTrainer(model, P, lr=lr, threads=cores)
      xc, ec = x.cpu(), torch.randn(B, zdim)
      one = lambda: tr.step(xc, ec)
      what = 'training steps'
    one()  # warm-up
    c0 = time.perf_counter()
    for _ in range(n_cpu):
      one()
    cdt = time.perf_counter() - c0
    cpu = dict(value=round(B * n_cpu / cdt, 1), unit='images/sec', cores=cores,
               kind='port', sample=f'{n_cpu} {what} of the same workload '
               f'(batch {B}), torch-CPU fp32 port of the reference step')

  ns3 = None
  if world == 1 and args.workload == 'dsprites_betavae_b256' and not args.no_north_star_3ch:
    ns3 = north_star_3ch(device)
  fit_tp = None
  if world == 1 and args.workload == 'dsprites_betavae_b256' and not args.no_fit:
    fit_tp = fit_throughput(device)
    for v in fit_tp.values():
      v['frac_of_step_replay'] = round(v['images_per_sec'] / (B * args.steps / dt), 4)
  exact = None
  if world == 1 and args.workload == 'dsprites_betavae_b256' and not args.no_exact_fp32 and use_graph:
    exact = exact_fp32_step(device)
  # step-level HBM roofline: bytes of ALL launches of one step (PMC passes, committed table) / step time / 8 TB/s
  hbm_step = None
  ent = pmc.get('step')
  if ent is not None and args.workload == 'dsprites_betavae_b256' and world == 1:
    bps = ent['traffic_bytes']
    hbm_step = dict(bytes_per_step=bps, launches=ent.get('launches'),
                    achieved=round(bps / (dt / args.steps) * 1e-9, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                    frac=round(bps / (dt / args.steps) * 1e-9 / PEAK_HBM_GBS, 4),
                    floor_ms=round(bps / (PEAK_HBM_GBS * 1e9) * 1e3, 4),
                    source='profiles/' + os.path.basename(pmc_file), measured_in_this_run=False)
  res = dict(metric='VAE train images/sec', value=round(B * world * args.steps / dt, 1),
             unit='images/sec', n_gpus=world, steps=args.steps, warmup=args.warmup,
             ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True, scaling='weak',
             vs_baseline=None, dtype='f32', data='synthetic',
             config=dict(workload=args.workload, global_batch=B * world, per_gpu_batch=B,
                         beta=beta, parallelism=f'dp{world}', graph=bool(use_graph),
                         arithmetic='fp32 results; the 4x4/s2 32-/64-channel layers carry their fp32 operands through '
                                    'the f16 matrix pipe as 2 planes (x = h + 2^-11 l; 3 MFMAs per 16 k-values, fp32 '
                                    'accumulation, <= 3*2^-22 per product, gradient tensors scaled per tensor by an '
                                    'exact power of two), fp32 MFMA / VALU elsewhere (ODIN_EXACT_FP32=1: everywhere)',
                         final_loss=round(loss, 4)),
             roofline=roofline, roofline_split=roofline_split, roofline_fp32=roofline_fp32, cpu_baseline=cpu,
             conv_stack=stack,
             in_step_conv_frac=round(stack['gflop'] / (dt / args.steps) * 1e-3 / PEAK_MFMA_F32_TFLOPS, 4),
             elbo_kernel=hbm[0],
             elbo_kernel_b512=next((h for h in hbm if h and h['kernel'] == 'elbo_bernoulli_fwd_bwd' and
                                    h['shape'] == [512, 64 * 64, 3]), None),
             hbm_kernels=hbm,
             north_star_3ch=ns3)
  if exact is not None:
    exact['headline_over_exact'] = round(exact['ms_per_step'] / (dt / args.steps * 1e3), 3)
    res['exact_fp32'] = exact
  if hbm_step is not None:
    res['hbm'] = hbm_step
  if fit_tp is not None:
    res['fit_images_per_sec'] = fit_tp['device_dataset']['images_per_sec']
    res['fit'] = fit_tp
  if mel_kernel is not None:
    res['mel_kernel'] = mel_kernel
  if rccl is not None:
    res['rccl'] = rccl
  print(json.dumps(res), file=json_out(), flush=True)
  if use_dist:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
