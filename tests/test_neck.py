"""odin_neck_fwd / odin_neck_bwd (neck.hip: conv3 -> projection -> latent block -> decoder projection -> deconv1 as
one launch per direction) through the C ABI against an independent float64 restatement (torch autograd over
torch.nn.functional: the same oracle style as oracle/torch_ref.py), on both backends (fixture `bk`).

Reference semantics: image_networks.py:466-471, 494-502 (the layers), dense_distribution.py:339-380 +
continuous.py:443-483 (DistributionDense / MVNDiag), helpers.py:236-286 (KL forms)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from odin_ai_amd import _lib

ACTF = {0: lambda t: t, 1: F.elu, 2: F.relu}


def _act_grad_from_output(act, y):
  if act == 1:
    return torch.where(y > 0, torch.ones_like(y), y + 1)
  if act == 2:
    return (y > 0).to(y.dtype)
  return torch.ones_like(y)


CASES = [
    # B, P, D, C0, acts (below, conv3, proj, dec proj, deconv1), analytic, free_bits, extras, scale of x / dy1
    (3, 128, 5, 8, (1, 1, 0, 0, 1), 0, None, False, 1.0, 1.0),
    (2, 256, 6, 16, (1, 1, 0, 0, 1), 1, 0.4, True, 1.0, 1.0),
    (5, 128, 10, 8, (2, 2, 1, 1, 2), 2, None, False, 1.0, 1.0),          # relu stack, reverse KL, odd batch
    (4, 256, 7, 16, (1, 1, 0, 0, 1), 0, None, True, 3e4, 1e-9),          # activations beyond the f16 window, tiny gradients
]


@pytest.mark.parametrize('B,P,D,C0,acts,analytic,free_bits,extras,xs,gs', CASES)
def test_neck_fwd_bwd(bk, B, P, D, C0, acts, analytic, free_bits, extras, xs, gs):
  L = bk.L
  rng = np.random.default_rng(B * 1000 + P + D)
  N0, J = 16 * C0, 2 * D
  f = lambda *s: (rng.standard_normal(s) * 0.3)
  x_pre = f(B, 8, 8, 64) * xs
  xin = ACTF[acts[0]](torch.tensor(x_pre)).numpy()   # the OUTPUT of the layer below (its activation applied)
  N = dict(x=xin, w3=f(4, 4, 64, 64) * 0.1 / xs, b3=f(64), w4=f(1024, P) * 0.1, b4=f(P), wl=f(P, J) * 0.3, bl=f(J),
           eps=f(B, D), w0=f(D, N0), b0=f(N0), w1=f(4, 4, 64, C0), b1=f(64), dy1=f(B, 8, 8, 64) * gs,
           dzx=f(B, D) * gs, dlx=f(B, D) * gs, dsx=f(B, D) * gs)
  T = {k: bk.T(v) for k, v in N.items()}
  shapes = dict(y3=(B, 4, 4, 64), y4=(B, P), p=(B, J), z=(B, D), kl=(B,), fbmask=(B,), y0=(B, N0), y1=(B, 8, 8, 64),
                dz=(B, D), dp=(B, J), dh4=(B, P), dy3=(B, 4, 4, 64), dx=(B, 8, 8, 64))
  O = {k: bk.zeros(s) for k, s in shapes.items()}
  rows = L.odin_neck_rows(B, P, D, C0)
  assert rows == (B + 1) // 2
  S1, S0, Sl = bk.zeros(rows, 16 * 64 * C0), bk.zeros(rows, D * N0 + N0), bk.zeros(rows, P * J + J)
  klw_v = 0.7 * gs
  klw, step = bk.T([klw_v]), bk.zeros(1, dtype=torch.int32)
  words = bk.zeros(5 * 2048, dtype=torch.int32)
  wp = lambda i: words.data_ptr() + 4 * 2048 * i
  L.odin_absmax(T['x'].data_ptr(), T['x'].numel(), wp(0), None)
  A = _lib.NeckArgs()
  A.B, A.P, A.D, A.C0 = B, P, D, C0
  A.act2, A.act3, A.act4, A.act0, A.act1 = acts
  A.analytic, A.free_bits, A.seed = analytic, -1.0 if free_bits is None else free_bits, 1
  A.step_dev = step.data_ptr()
  for k in ('x', 'w3', 'b3', 'w4', 'b4', 'wl', 'bl', 'w0', 'b0', 'w1', 'b1', 'dy1'):
    setattr(A, k, T[k].data_ptr())
  A.eps_in = A.eps = T['eps'].data_ptr()
  for k in O:
    setattr(A, k, O[k].data_ptr())
  A.klw = klw.data_ptr()
  A.x_amax, A.y1_amax, A.dh4_amax, A.dy3_amax, A.dx_amax = wp(0), wp(1), wp(2), wp(3), wp(4)
  if extras:
    A.dz_extra, A.dloc_x, A.dscale_x = T['dzx'].data_ptr(), T['dlx'].data_ptr(), T['dsx'].data_ptr()
  A.slab1, A.slab0, A.slabl = S1.data_ptr(), S0.data_ptr(), Sl.data_ptr()
  L.odin_neck_fwd(C.byref(A), None)
  L.odin_neck_bwd(C.byref(A), None)
  assert L.odin_debug_last_path().decode() == 'neck_bwd'
  # ---- float64 restatement
  learn = ('x', 'w3', 'b3', 'w4', 'b4', 'wl', 'bl', 'w0', 'b0', 'w1', 'b1')
  d = {k: torch.tensor(v, dtype=torch.float64).requires_grad_(k in learn) for k, v in N.items()}
  y3p = F.conv2d(F.pad(d['x'].permute(0, 3, 1, 2), (1, 1, 1, 1)), d['w3'].permute(3, 2, 0, 1), d['b3'], stride=2)
  y3 = ACTF[acts[1]](y3p).permute(0, 2, 3, 1)
  y4p = y3.reshape(B, 1024) @ d['w4'] + d['b4']
  y4 = ACTF[acts[2]](y4p)
  p = y4 @ d['wl'] + d['bl']
  loc, sc = p[:, :D], F.softplus(p[:, D:])
  z = loc + sc * d['eps']
  if analytic == 1:
    klt = 0.5 * (sc * sc + loc * loc - 1) - torch.log(sc)
  elif analytic == 2:
    klt = torch.log(sc) + 0.5 * (1 + loc * loc) / (sc * sc) - 0.5
  else:
    klt = 0.5 * (z * z - d['eps'] ** 2) - torch.log(sc)
  kl = klt.sum(1)
  mask = torch.ones(B, dtype=torch.float64)
  if free_bits is not None:
    thr = free_bits * D
    mask = (kl > thr).double()
    kl_out = torch.where(kl > thr, kl, torch.full_like(kl, thr))
  else:
    kl_out = kl
  y0p = z @ d['w0'] + d['b0']
  y0 = ACTF[acts[3]](y0p)
  y1p = F.conv_transpose2d(y0.reshape(B, 4, 4, C0).permute(0, 3, 1, 2), d['w1'].permute(3, 2, 0, 1), d['b1'], stride=2,
                           padding=1)
  y1 = ACTF[acts[4]](y1p).permute(0, 2, 3, 1)
  ref_f = dict(y3=y3, y4=y4, p=p, z=z, kl=kl_out, fbmask=mask, y0=y0, y1=y1)
  for k, v in ref_f.items():
    got, v = O[k].cpu().double(), v.detach()
    assert float((got - v).abs().max()) <= 2e-5 * max(1.0, float(v.abs().max())), ('fwd', k)
  # loss whose gradients are what the step back-propagates through the neck
  loss = (d['dy1'].permute(0, 3, 1, 2) * y1p).sum() + klw_v * (kl * mask).sum()
  if extras:
    loss = loss + (d['dzx'] * z).sum() + (d['dlx'] * loc).sum() + (d['dsx'] * sc).sum()
  for t in (y3p, y4p, p, z, y0p):
    t.retain_grad()
  loss.backward()
  ref_b = dict(dy3=y3p.grad.permute(0, 2, 3, 1), dh4=y4p.grad, dp=p.grad,
               dx=d['x'].grad * _act_grad_from_output(acts[0], d['x'].detach()))
  for k, v in ref_b.items():
    got = O[k].cpu().double()
    assert float((got - v).abs().max()) <= 1e-4 * float(v.abs().max()), ('bwd', k, float((got - v).abs().max()), float(v.abs().max()))
  # dz as odin_latent_bwd defines it: the decoder-side term only (g0 W0^T)
  dz_dec = (y0p.grad @ d['w0'].detach().T)
  assert float((O['dz'].cpu().double() - dz_dec).abs().max()) <= 1e-4 * float(dz_dec.abs().max())
  for slab, ref in ((S1, d['w1'].grad.reshape(-1)), (S0, torch.cat([d['w0'].grad.reshape(-1), d['b0'].grad])),
                    (Sl, torch.cat([d['wl'].grad.reshape(-1), d['bl'].grad]))):
    got = slab.cpu().double().sum(0)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
  # range words: valid bounds of what was written
  wv = words.cpu().view(torch.float32).view(5, 2048).max(1).values
  for i, k in ((1, 'y1'), (2, 'dh4'), (3, 'dy3'), (4, 'dx')):
    assert float(wv[i]) >= float(O[k].abs().max()) > 0, (k, float(wv[i]))


def test_neck_rows_regime(bk):
  L = bk.L
  assert L.odin_neck_rows(256, 128, 10, 8) == 128 and L.odin_neck_rows(257, 256, 6, 16) == 129
  assert L.odin_neck_rows(256, 64, 10, 8) == 0 and L.odin_neck_rows(256, 128, 10, 4) == 0
  assert L.odin_neck_rows(256, 128, 40, 8) == 0 and L.odin_neck_rows(2048, 128, 10, 8) == 0
