"""Pointwise / reduction entry points of the C ABI vs the float64 oracle, on the CPU simulator
build (`-m "not gpu"`) and on the hipcc build on an MI355X (`-m gpu`); fixture `bk`."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from odin_ai_amd import _lib
from oracle import vae_oracle as vo


def close(a, b, tol=2e-5):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), np.abs(a - b).max()


@pytest.mark.parametrize('analytic,fb', [(0, -1.0), (1, -1.0), (0, 0.4)])
def test_latent_fwd_bwd(bk, analytic, fb):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(0)
  B, D = 37, 10
  p = rng.standard_normal((B, 2 * D))
  eps = rng.standard_normal((B, D))
  loc, sc = vo.mvn_diag_params(p, D)
  z_ref = loc + sc * eps
  klr = vo.kl_analytic(loc, sc) if analytic else vo.kl_mc(loc, sc, z_ref)
  kl_ref, m_ref = vo.free_bits_clamp(klr, None if fb < 0 else fb, D)
  tp, te = T(p), T(eps)
  z, kl, m = bk.zeros(B, D), bk.zeros(B), bk.zeros(B)
  L.odin_latent_fwd(tp.data_ptr(), te.data_ptr(), z.data_ptr(), kl.data_ptr(), m.data_ptr(), B, D,
                    analytic, fb, None, None)
  close(z.cpu().numpy(), z_ref)
  close(kl.cpu().numpy(), kl_ref)
  assert (m.cpu().numpy() == m_ref).all()
  dz = rng.standard_normal((B, D))
  klw = 4.0 / B
  w = klw * m_ref[:, None]
  if analytic:
    dloc, dsc = w * loc, w * (sc - 1 / sc)
  else:
    dloc, dsc = w * z_ref, w * (z_ref * eps - 1 / sc)
  dloc, dsc = dloc + dz, dsc + dz * eps
  dp_ref = np.concatenate([dloc, dsc * vo.sigmoid(p[:, D:])], -1)
  tdz, tk, dp = T(dz), T([klw]), bk.zeros(B, 2 * D)
  L.odin_latent_bwd(tp.data_ptr(), te.data_ptr(), z.data_ptr(), tdz.data_ptr(), None, m.data_ptr(),
                    tk.data_ptr(), None, None, dp.data_ptr(), B, D, analytic, None)
  close(dp.cpu().numpy(), dp_ref)


@pytest.mark.parametrize('shape', [(3, 8, 8, 1), (2, 64, 64, 3), (5, 7, 3, 1), (3, 64, 64, 1),
                                   (2, 32, 32, 1), (3, 16, 16, 2),
                                   # >= 2^20 elements: the persistent grid-stride form (one partial per 256 elements)
                                   (86, 64, 64, 3), (300, 64, 64, 1)])
def test_elbo_bernoulli_and_finalize(bk, shape):
  if bk.name == 'sim' and int(np.prod(shape)) > 200000:
    pytest.skip('large shapes run on the GPU backend only (fiber simulator: minutes)')
  L, T = bk.L, bk.T
  rng = np.random.default_rng(1)
  B = shape[0]
  N = int(np.prod(shape[1:]))
  lg = rng.standard_normal(shape) * 3
  x = np.clip(rng.random(shape), 1e-6, 1 - 1e-6)
  llk_ref = vo.bernoulli_log_prob(lg, x)
  npart = C.c_int(0)
  tl, tx, sc = T(lg), T(x), T([1.0 / B])
  L.odin_elbo_bernoulli_fwd_bwd(None, None, None, None, None, B, N, C.byref(npart), None)  # dry run
  part = bk.zeros(B * npart.value)
  dl = bk.zeros(shape)
  L.odin_elbo_bernoulli_fwd_bwd(tl.data_ptr(), tx.data_ptr(), part.data_ptr(), dl.data_ptr(),
                                sc.data_ptr(), B, N, C.byref(npart), None)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref, 1e-5)
  close(dl.cpu().numpy(), -(vo.bernoulli_log_prob_grad(lg, x)) / B)
  kl = rng.random(B) * 5
  tkl, hyper, ttc = T(kl), T([4.0, 0.5]), T([0.5])
  llk, out = bk.zeros(B), bk.zeros(4)
  L.odin_elbo_finalize(part.data_ptr(), npart.value, tkl.data_ptr(), hyper.data_ptr(),
                       ttc.data_ptr(), llk.data_ptr(), out.data_ptr(), B, None)
  close(llk.cpu().numpy(), llk_ref, 1e-5)
  loss = -(llk_ref.mean() - 4.0 * kl.mean() - 0.25)
  close(out.cpu().numpy(), [loss, llk_ref.mean(), 4.0 * kl.mean(), 0.25], 1e-5)


@pytest.mark.parametrize('B,n_part', [(5, 1), (70, 33), (256, 240), (9, 100)])
def test_elbo_finalize_partial_counts(bk, B, n_part):
  """odin_elbo_finalize over few (one thread per sample) and many (> 32: one wave per sample) partials per sample"""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(21)
  part = rng.standard_normal((B, n_part)) * 30
  kl = rng.random(B) * 5
  tp, tkl, hyper = T(part), T(kl), T([2.0, 0.0])
  llk, out = bk.zeros(B), bk.zeros(4)
  L.odin_elbo_finalize(tp.data_ptr(), n_part, tkl.data_ptr(), hyper.data_ptr(), None, llk.data_ptr(), out.data_ptr(),
                       B, None)
  ref = part.astype(np.float32).astype(np.float64).sum(1)
  close(llk.cpu().numpy(), ref, 1e-5)
  close(out.cpu().numpy(), [-(ref.mean() - 2.0 * kl.mean()), ref.mean(), 2.0 * kl.mean(), 0.0], 1e-5)


@pytest.mark.parametrize('npix,Cc', [(50, 3), (1024, 3), (512, 1), (256, 2)])
@pytest.mark.parametrize('sp1', [0, 1])
def test_elbo_gaussian(bk, sp1, npix, Cc):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(2)
  B = 3
  h = rng.standard_normal((B, npix, 2 * Cc))
  if not sp1:
    h[..., Cc:] = 0.5 + rng.random((B, npix, Cc))
  x = rng.random((B, npix, Cc))
  loc, raw = h[..., :Cc], h[..., Cc:]
  sd = vo.softplus1(raw) if sp1 else raw
  llk_ref = vo.gaussian_log_prob(loc, sd, x)
  d = (x - loc) / sd
  dsd = vo.sigmoid(raw + vo.SOFTPLUS_INV_1) if sp1 else 1.0
  dh_ref = -np.concatenate([d / sd, (d * d - 1) / sd * dsd], -1) / B
  th, tx, sc = T(h), T(x), T([1.0 / B])
  npart = C.c_int(0)
  L.odin_elbo_gaussian_fwd_bwd(None, None, None, None, None, B, npix, Cc, sp1, C.byref(npart), None)
  part, dh = bk.zeros(B * npart.value), bk.zeros(B, npix, 2 * Cc)
  L.odin_elbo_gaussian_fwd_bwd(th.data_ptr(), tx.data_ptr(), part.data_ptr(), dh.data_ptr(),
                               sc.data_ptr(), B, npix, Cc, sp1, C.byref(npart), None)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref, 1e-5)
  close(dh.cpu().numpy(), dh_ref)


@pytest.mark.parametrize('B,npix,Cin,Cc,sp1,hact', [(3, 300, 32, 1, 1, 'elu'), (2, 1000, 32, 3, 0, 'elu'),
                                                     (5, 64, 16, 1, 0, 'relu'), (2, 530, 8, 3, 1, 'linear'),
                                                     (7, 7680, 32, 1, 1, 'elu')])
def test_gaussian_head(bk, B, npix, Cin, Cc, sp1, hact):
  """odin_gaussian_head_fwd_bwd = Conv2D 1x1 -> Normal log-prob -> its backward pass in one launch, against the
  separate formulas in float64: logits, llk, dlogits, dh = (dlogits w1^T) act'(h), (dW1 | db1), column sums of
  dh, the range word of dh."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(12)
  CO = 2 * Cc
  pre = rng.standard_normal((B, npix, Cin))
  h = {'elu': vo.elu, 'relu': lambda a: np.maximum(a, 0.0), 'linear': lambda a: a}[hact](pre)
  h = h.astype(np.float32).astype(np.float64)
  w1 = rng.standard_normal((Cin, CO)) * 0.2
  b1 = rng.standard_normal(CO) * 0.1
  if not sp1:
    b1[Cc:] += 3.0  # raw scales must be positive
  w1[:, Cc:] *= 0.2
  x = rng.random((B, npix, Cc))
  lg = h @ w1 + b1
  loc, raw = lg[..., :Cc], lg[..., Cc:]
  sd = vo.softplus1(raw) if sp1 else raw
  assert sd.min() > 0.05
  llk_ref = vo.gaussian_log_prob(loc, sd, x)
  d = (x - loc) / sd
  dsd = vo.sigmoid(raw + vo.SOFTPLUS_INV_1) if sp1 else 1.0
  dl_ref = -np.concatenate([d / sd, (d * d - 1) / sd * dsd], -1) / B
  hgrad = {'linear': np.ones_like(h), 'relu': (h > 0).astype(np.float64), 'elu': vo.elu_grad_from_output(h)}[hact]
  dh_ref = (dl_ref @ w1.T) * hgrad
  dW_ref = np.einsum('bpc,bpo->co', h, dl_ref)
  db_ref = dl_ref.sum((0, 1))
  th, tw, tb, tx, sc = T(h), T(w1), T(b1), T(x), T([1.0 / B])
  npart, rows = C.c_int(0), C.c_int(0)
  L.odin_gaussian_head_fwd_bwd(None, None, None, None, None, None, None, None, C.byref(npart), None, C.byref(rows),
                               None, None, B, npix, Cin, Cc, sp1, 0, None, None)
  assert npart.value >= 1 and 1 <= rows.value <= 512
  logits, dl, dh = bk.zeros(B, npix, CO), bk.zeros(B, npix, CO), bk.full((B, npix, Cin), float('nan'))
  part = bk.full((B * npart.value,), float('nan'))
  slab = bk.full((rows.value, Cin * CO + CO), float('nan'))
  cs = bk.full((rows.value, Cin), float('nan'))
  word = bk.zeros(2048, dtype=torch.int32)
  act = _lib.ACT[hact]
  L.odin_gaussian_head_fwd_bwd(th.data_ptr(), tw.data_ptr(), tb.data_ptr(), tx.data_ptr(), logits.data_ptr(),
                               dl.data_ptr(), dh.data_ptr(), part.data_ptr(), C.byref(npart), slab.data_ptr(),
                               C.byref(rows), cs.data_ptr(), sc.data_ptr(), B, npix, Cin, Cc, sp1, act,
                               word.data_ptr(), None)
  close(logits.cpu().numpy(), lg)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref, 1e-5)
  close(dl.cpu().numpy(), dl_ref)
  close(dh.cpu().numpy(), dh_ref)
  g = slab.cpu().numpy().astype(np.float64).sum(0)
  close(g[:Cin * CO].reshape(Cin, CO), dW_ref, 1e-4)
  close(g[Cin * CO:], db_ref, 1e-4)
  close(cs.cpu().numpy().astype(np.float64).sum(0), dh_ref.sum((0, 1)), 1e-4)
  assert float(word.view(torch.float32).max()) == float(dh.abs().max())
  # without the optional outputs
  dh2 = bk.full((B, npix, Cin), float('nan'))
  L.odin_gaussian_head_fwd_bwd(th.data_ptr(), tw.data_ptr(), tb.data_ptr(), tx.data_ptr(), logits.data_ptr(), None,
                               dh2.data_ptr(), part.data_ptr(), C.byref(npart), slab.data_ptr(), C.byref(rows),
                               None, sc.data_ptr(), B, npix, Cin, Cc, sp1, act, None, None)
  assert torch.equal(dh2, dh)


@pytest.mark.parametrize('B,npix,Cin,Cc,hact', [(3, 300, 32, 1, 'elu'), (2, 784, 32, 1, 'relu'), (2, 530, 16, 3, 'elu')])
def test_bernoulli_head(bk, B, npix, Cin, Cc, hact):
  """odin_gaussian_head_fwd_bwd with softplus1 = 3: Conv2D 1x1 -> Bernoulli(logits) log-prob -> backward, one launch"""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(14)
  pre = rng.standard_normal((B, npix, Cin))
  h = {'elu': vo.elu, 'relu': lambda a: np.maximum(a, 0.0)}[hact](pre).astype(np.float32).astype(np.float64)
  w1 = rng.standard_normal((Cin, Cc)) * 0.3
  b1 = rng.standard_normal(Cc) * 0.1
  x = (rng.random((B, npix, Cc)) < 0.3).astype(np.float64).clip(1e-6, 1 - 1e-6)
  lg = h @ w1 + b1
  llk_ref = vo.bernoulli_log_prob(lg.reshape(B, -1), x.reshape(B, -1))
  dl_ref = -vo.bernoulli_log_prob_grad(lg, x) / B
  hgrad = {'relu': (h > 0).astype(np.float64), 'elu': vo.elu_grad_from_output(h)}[hact]
  dh_ref = (dl_ref @ w1.T) * hgrad
  th, tw, tb, tx, sc = T(h), T(w1), T(b1), T(x), T([1.0 / B])
  npart, rows = C.c_int(0), C.c_int(0)
  L.odin_gaussian_head_fwd_bwd(None, None, None, None, None, None, None, None, C.byref(npart), None, C.byref(rows),
                               None, None, B, npix, Cin, Cc, 3, 0, None, None)
  logits, dl, dh = bk.zeros(B, npix, Cc), bk.zeros(B, npix, Cc), bk.full((B, npix, Cin), float('nan'))
  part = bk.full((B * npart.value,), float('nan'))
  slab = bk.full((rows.value, Cin * Cc + Cc), float('nan'))
  cs = bk.full((rows.value, Cin), float('nan'))
  L.odin_gaussian_head_fwd_bwd(th.data_ptr(), tw.data_ptr(), tb.data_ptr(), tx.data_ptr(), logits.data_ptr(),
                               dl.data_ptr(), dh.data_ptr(), part.data_ptr(), C.byref(npart), slab.data_ptr(),
                               C.byref(rows), cs.data_ptr(), sc.data_ptr(), B, npix, Cin, Cc, 3, _lib.ACT[hact],
                               None, None)
  close(logits.cpu().numpy(), lg)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref, 1e-5)
  close(dl.cpu().numpy(), dl_ref)
  close(dh.cpu().numpy(), dh_ref)
  g = slab.cpu().numpy().astype(np.float64).sum(0)
  close(g[:Cin * Cc].reshape(Cin, Cc), np.einsum('bpc,bpo->co', h, dl_ref), 1e-4)
  close(g[Cin * Cc:], dl_ref.sum((0, 1)), 1e-4)
  close(cs.cpu().numpy().astype(np.float64).sum(0), dh_ref.sum((0, 1)), 1e-4)


@pytest.mark.parametrize('B,K,N,nmid,ns,rows', [(256, 6, 1000, 4100, 1001, 32), (37, 10, 260, 12, 5, 3), (64, 32, 64, 0, 64, 1)])
def test_adam_with_folded_gradient_pieces(bk, B, K, N, nmid, ns, rows):
  """odin_adam_step_fold: the update of FactorVAE's discriminator (Adam(1e-5, .5, .9), factor_vae.py:168-176) with the first
  layer's weight gradient (x^T dy | column sums, base_networks.py:1002-1014 under the tape) and the sum of the head's slab
  rows formed inside the launch, and the range words cleared by it: parameters, moments and the gradient buffer against
  the oracle and against the separate launches."""
  from odin_ai_amd._lib import AdamFold
  L, T = bk.L, bk.T
  rng = np.random.default_rng(B + K)
  nA = (K + 1) * N
  offS = nA + nmid
  n = ((offS + ns + 3) // 4) * 4
  x, dy = rng.standard_normal((B, K)), rng.standard_normal((B, N)) * 1e-2
  slab = rng.standard_normal((rows, ns + 3)) * 1e-2
  th, gmid = rng.standard_normal(n), rng.standard_normal(n) * 1e-2
  m, v = rng.standard_normal(n) * 0.01, rng.random(n) * 1e-3
  f32 = lambda a_: np.asarray(a_, np.float32).astype(np.float64)
  g_ref = f32(gmid).copy()
  g_ref[:K * N] = (f32(x).T @ f32(dy)).reshape(-1)
  g_ref[K * N:nA] = f32(dy).sum(0)
  g_ref[offS:offS + ns] = f32(slab)[:, :ns].sum(0)
  t, lr = 3, 1e-5
  th_ref, m_ref, v_ref = vo.adam_keras(f32(th), g_ref, f32(m), f32(v), t, lr, b1=0.5, b2=0.9)
  a = lr * math.sqrt(1 - 0.9 ** t) / (1 - 0.5 ** t)
  hy = T([a, 0.5, 0.9, 1e-7, 1.0])
  tx, tdy, tslab = T(x), T(dy), T(slab)
  tth, tg, tm, tv = T(th), T(gmid), T(m), T(v)
  words = bk.zeros(300, dtype=torch.int32) + 7
  fo = AdamFold(tx.data_ptr(), tdy.data_ptr(), B, K, N, 0, tslab.data_ptr(), rows, ns + 3, ns, offS, words.data_ptr(), 299)
  L.odin_adam_step_fold(tth.data_ptr(), tg.data_ptr(), tm.data_ptr(), tv.data_ptr(), n, hy.data_ptr(), C.byref(fo), None)
  assert int(words[:299].abs().sum()) == 0 and int(words[299]) == 7
  close(tg.cpu().numpy(), g_ref, 2e-6)
  close(tm.cpu().numpy(), m_ref, 1e-6)
  close(tv.cpu().numpy(), v_ref, 1e-6)
  close(tth.cpu().numpy(), th_ref, 1e-6)
  # the separate launch on the gradient buffer the fold left: the same update everywhere (two kernels: the compiler
  # contracts their multiply-adds differently, so to rounding); the untouched gradients bit for bit
  tth2, tg2, tm2, tv2 = T(th), tg.clone(), T(m), T(v)
  L.odin_adam_step_flat(tth2.data_ptr(), tg2.data_ptr(), tm2.data_ptr(), tv2.data_ptr(), n, hy.data_ptr(), None, 0.0, None, None)
  for got, want in ((tth, tth2), (tm, tm2), (tv, tv2)):
    close(got.cpu().numpy(), want.cpu().numpy(), 1e-6)
  assert torch.equal(tg[nA:offS], T(gmid)[nA:offS])


def test_adam_and_sumsq(bk):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(3)
  n = 1027
  th, g = rng.standard_normal(n), rng.standard_normal(n)
  m, v = rng.standard_normal(n) * 0.1, rng.random(n) * 0.1
  t, lr = 7, 1e-3
  th_ref, m_ref, v_ref = vo.adam_keras(th, g, m, v, t, lr)
  a = lr * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
  tth, tg, tm, tv = T(th), T(g), T(m), T(v)
  hy = T([a, 0.9, 0.999, 1e-7, 1.0])
  L.odin_adam_step_flat(tth.data_ptr(), tg.data_ptr(), tm.data_ptr(), tv.data_ptr(), n,
                        hy.data_ptr(), None, 0.0, None, None)
  close(tth.cpu().numpy(), th_ref, 1e-6)
  close(tm.cpu().numpy(), m_ref, 1e-6)
  close(tv.cpu().numpy(), v_ref, 1e-6)
  ws, out = bk.zeros(1024), bk.zeros(1)
  L.odin_sumsq_flat(tg.data_ptr(), n, ws.data_ptr(), out.data_ptr(), None)
  close(out.cpu().numpy(), [(g.astype(np.float32).astype(np.float64) ** 2).sum()], 1e-6)
  # clipping by global norm + NaN guard
  tth2, tm2, tv2 = T(th), T(m), T(v)
  clip = 3.0
  gc, _ = vo.clip_by_global_norm([g], clip)
  th2_ref, _, _ = vo.adam_keras(th, gc[0], m, v, t, lr)
  L.odin_adam_step_flat(tth2.data_ptr(), tg.data_ptr(), tm2.data_ptr(), tv2.data_ptr(), n,
                        hy.data_ptr(), out.data_ptr(), clip, None, None)
  close(tth2.cpu().numpy(), th2_ref, 1e-6)
  nan, flag = T([float('nan')]), bk.zeros(1, dtype=torch.int32)
  before = tth2.clone()
  L.odin_adam_step_flat(tth2.data_ptr(), tg.data_ptr(), tm2.data_ptr(), tv2.data_ptr(), n,
                        hy.data_ptr(), nan.data_ptr(), clip, flag.data_ptr(), None)
  assert flag.item() == 1 and torch.equal(before, tth2)
  # fused norm + update (two launches): bit-identical to sumsq_flat followed by adam_step_flat
  tth3, tm3, tv3 = T(th), T(m), T(v)
  tth4, tm4, tv4 = T(th), T(m), T(v)
  ws3, out3, flag3 = bk.zeros(1024), bk.zeros(1), bk.zeros(1, dtype=torch.int32)
  L.odin_sumsq_adam_flat(tth3.data_ptr(), tg.data_ptr(), tm3.data_ptr(), tv3.data_ptr(), n,
                         hy.data_ptr(), ws3.data_ptr(), out3.data_ptr(), clip, flag3.data_ptr(), None)
  L.odin_adam_step_flat(tth4.data_ptr(), tg.data_ptr(), tm4.data_ptr(), tv4.data_ptr(), n,
                        hy.data_ptr(), out.data_ptr(), clip, None, None)
  assert torch.equal(out3, out) and flag3.item() == 0
  assert torch.equal(tth3, tth4) and torch.equal(tm3, tm4) and torch.equal(tv3, tv4)
  gbad = T(np.where(np.arange(n) == 5, np.inf, g))
  before = tth3.clone()
  L.odin_sumsq_adam_flat(tth3.data_ptr(), gbad.data_ptr(), tm3.data_ptr(), tv3.data_ptr(), n,
                         hy.data_ptr(), ws3.data_ptr(), out3.data_ptr(), clip, flag3.data_ptr(), None)
  assert flag3.item() == 1 and torch.equal(before, tth3)


def test_total_correlation(bk):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(4)
  B, D = 70, 6
  z = rng.standard_normal((B, D))
  p = rng.standard_normal((B, 2 * D))
  loc, sc = vo.mvn_diag_params(p, D)
  tc_ref = vo.total_correlation(z, loc, sc)
  gz, gl, gs = vo.total_correlation_bwd(z, loc, sc)
  coef = 3.0
  tz, tp, tcf = T(z), T(p), T([coef])
  ws = bk.zeros(L.odin_total_correlation_workspace(B, B, D))
  dz, dl, ds = bk.zeros(B, D), bk.zeros(B, D), bk.zeros(B, D)
  L.odin_total_correlation_fwd_bwd(tz.data_ptr(), tp.data_ptr(), ws.data_ptr(), dz.data_ptr(),
                                   dl.data_ptr(), ds.data_ptr(), tcf.data_ptr(), B, D, None)
  close([ws[0].item()], [tc_ref], 1e-5)
  close(dz.cpu().numpy(), coef * gz, 1e-4)
  close(dl.cpu().numpy(), coef * gl, 1e-4)
  close(ds.cpu().numpy(), coef * gs, 1e-4)


@pytest.mark.parametrize('spread', [1.0, 0.3])
def test_total_correlation_on_posterior_samples(bk, spread):
  """The regime the training step is in: z_j is a SAMPLE OF ITS OWN posterior (z = loc + scale*eps),
  so row j's softmax over i peaks sharply at i = j and the gradient weights w_joint - w_latent
  cancel; CelebA size (B=512, D=45) on the GPU build."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(6)
  B, D = (512, 45) if bk.name == 'hip' else (96, 45)
  p = rng.standard_normal((B, 2 * D)) * spread
  loc, sc = vo.mvn_diag_params(p.astype(np.float32).astype(np.float64), D)
  z = (loc + sc * rng.standard_normal((B, D))).astype(np.float32).astype(np.float64)
  tc_ref = vo.total_correlation(z, loc, sc)
  gz, gl, gs = vo.total_correlation_bwd(z, loc, sc)
  tz, tp, tcf = T(z), T(p), T([3.0])
  ws = bk.zeros(L.odin_total_correlation_workspace(B, B, D))
  dz, dl, ds = bk.zeros(B, D), bk.zeros(B, D), bk.zeros(B, D)
  L.odin_total_correlation_fwd_bwd(tz.data_ptr(), tp.data_ptr(), ws.data_ptr(), dz.data_ptr(),
                                   dl.data_ptr(), ds.data_ptr(), tcf.data_ptr(), B, D, None)
  close([ws[0].item()], [tc_ref], 1e-5)
  for got, ref, nm in ((dz, gz, 'dz'), (dl, gl, 'dloc'), (ds, gs, 'dscale')):
    e = np.abs(got.cpu().numpy() - 3.0 * ref).max() / np.abs(3.0 * ref).max()
    assert e <= 1e-4, (nm, e)


def test_permute_and_dtc_and_rng(bk):
  L, T = bk.L, bk.T
  rng = np.random.default_rng(5)
  B, D = 128, 6
  # reference's own test (tests/bayesian/test_vae.py:112-125): portable properties
  z = rng.standard_normal((B, D))
  perm = bk.zeros(B, D, dtype=torch.int32)
  step = bk.T([3], torch.int32)
  L.odin_random_perm(perm.data_ptr(), B, D, 1234, step.data_ptr(), None)
  pn = perm.cpu().numpy()
  for l in range(D):
    assert sorted(pn[:, l].tolist()) == list(range(B))
  assert not all((pn[:, l] == np.arange(B)).all() for l in range(D))
  assert any((pn[:, 0] != pn[:, l]).any() for l in range(1, D))
  tz, out = T(z), bk.zeros(B, D)
  L.odin_permute_dims(tz.data_ptr(), perm.data_ptr(), out.data_ptr(), B, D, None)
  ref = vo.permute_dims(z, pn.astype(np.int64))
  close(out.cpu().numpy(), ref, 1e-7)
  assert (out.cpu().numpy() != z.astype(np.float32)).any()
  assert np.allclose(np.sort(out.cpu().numpy(), 0), np.sort(z.astype(np.float32), 0))
  # the two as one launch: the same permutation, the same rows
  perm2, out2 = bk.zeros(B, D, dtype=torch.int32), bk.zeros(B, D)
  L.odin_random_permute_dims(perm2.data_ptr(), tz.data_ptr(), out2.data_ptr(), B, D, 1234, step.data_ptr(), None)
  assert torch.equal(perm2, perm) and torch.equal(out2, out)
  # dtc loss
  lz, lp = rng.standard_normal(B) * 2, rng.standard_normal(B) * 2
  tlz, tlp = T(lz), T(lp)
  o, dlz, dlp = bk.zeros(1), bk.zeros(B), bk.zeros(B)
  L.odin_dtc_loss_fwd_bwd(tlz.data_ptr(), tlp.data_ptr(), o.data_ptr(), dlz.data_ptr(),
                          dlp.data_ptr(), B, None)
  close(o.cpu().numpy(), [vo.dtc_loss(lz, lp)], 1e-5)
  a, b = vo.dtc_loss_bwd(lz, lp)
  close(dlz.cpu().numpy(), a, 1e-5)
  close(dlp.cpu().numpy(), b, 1e-5)
  # rng: moments + determinism + step dependence
  n = 200001
  r1, r2, r3 = bk.zeros(n), bk.zeros(n), bk.zeros(n)
  s2 = bk.T([4], torch.int32)
  L.odin_rng_normal(r1.data_ptr(), n, 99, step.data_ptr(), None)
  L.odin_rng_normal(r2.data_ptr(), n, 99, step.data_ptr(), None)
  L.odin_rng_normal(r3.data_ptr(), n, 99, s2.data_ptr(), None)
  assert torch.equal(r1, r2) and not torch.equal(r1, r3)
  assert abs(r1.mean().item()) < 0.01 and abs(r1.std().item() - 1) < 0.01
  assert abs((r1 ** 3).mean().item()) < 0.05 and abs((r1 ** 4).mean().item() - 3) < 0.1


def test_gradient_policies(bk):
  """skip_update_threshold / per-variable clipnorm / clipvalue after global-norm scaling
  (base_networks.py:549-596) on a flat buffer of 4 variables."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(8)
  sizes = [37, 1200, 5, 4100]
  offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
  n = int(offs[-1])
  g = rng.standard_normal(n) * np.repeat([0.1, 3.0, 10.0, 0.01], sizes)
  parts = lambda a: [a[offs[i]:offs[i + 1]] for i in range(4)]
  seg = bk.T(offs, torch.int64)
  # per-variable clip_by_norm
  tg = T(g)
  L.odin_clip_by_norm_segments(tg.data_ptr(), seg.data_ptr(), 4, 2.5, None)
  ref, _ = vo.gradient_policies(parts(g), clipnorm=2.5)
  close(tg.cpu().numpy(), np.concatenate(ref), 1e-6)
  # clip_by_value alone, and after clip_by_global_norm
  tg = T(g)
  L.odin_clip_by_value(tg.data_ptr(), n, 0.7, None, 0.0, None)
  close(tg.cpu().numpy(), np.clip(g, -0.7, 0.7), 1e-7)
  tg, ws, n2 = T(g), bk.zeros(1024), bk.zeros(1)
  L.odin_sumsq_flat(tg.data_ptr(), n, ws.data_ptr(), n2.data_ptr(), None)
  L.odin_clip_by_value(tg.data_ptr(), n, 0.05, n2.data_ptr(), 20.0, None)
  ref, _ = vo.gradient_policies(parts(g), global_clipnorm=20.0, clipvalue=0.05)
  close(tg.cpu().numpy(), np.concatenate(ref), 1e-6)
  # skip_update_threshold
  hit, cnt = bk.zeros(1, dtype=torch.int32), bk.zeros(1, dtype=torch.int32)
  on, off = bk.T([1], torch.int32), bk.T([0], torch.int32)
  tg = T(g)
  L.odin_grad_skip_threshold(tg.data_ptr(), n, float(g.max()) + 1.0, on.data_ptr(), hit.data_ptr(),
                             cnt.data_ptr(), None)
  assert hit.item() == 0 and cnt.item() == 0 and np.array_equal(tg.cpu().numpy(), g.astype(np.float32))
  L.odin_grad_skip_threshold(tg.data_ptr(), n, float(g.max()) - 1e-3, off.data_ptr(), hit.data_ptr(),
                             cnt.data_ptr(), None)  # threshold reached but step < when_skip_update
  assert hit.item() == 1 and cnt.item() == 0 and np.array_equal(tg.cpu().numpy(), g.astype(np.float32))
  L.odin_grad_skip_threshold(tg.data_ptr(), n, float(g.max()) - 1e-3, on.data_ptr(), hit.data_ptr(),
                             cnt.data_ptr(), None)
  assert hit.item() == 1 and cnt.item() == 1 and not tg.cpu().numpy().any()


@pytest.mark.parametrize('npix,Cc', [(50, 3), (1024, 3), (512, 1)])
def test_elbo_quantized_logistic(bk, npix, Cc):
  """QuantizedLogistic head (image_networks.py:55-71; quantized.py:50-204): log-prob and its
  gradient wrt (loc, raw), including the edge bins (x*255 at 0 / 255), values that sit exactly
  on a pixel level, wide and very narrow scales."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(12)
  B = 3
  h = rng.standard_normal((B, npix, 2 * Cc))
  h[..., Cc:] *= 3.0                                     # scales from ~1e-3 to ~4 (x 127.5 pixels)
  x = rng.random((B, npix, Cc)).astype(np.float32)
  x[0, :8] = np.array([1e-6, 1 - 1e-6, 0.0, 1.0, 0.5, 200 / 255, 3 / 255, 254 / 255],
                      np.float32)[:, None]
  x = x.astype(np.float64)
  loc, raw = h[..., :Cc], h[..., Cc:]
  loc32, raw32 = loc.astype(np.float32).astype(np.float64), raw.astype(np.float32).astype(np.float64)
  llk_ref = vo.qlogistic_log_prob(loc32, raw32, x)
  gl, gr = vo.qlogistic_log_prob_grad(loc32, raw32, x)
  dh_ref = -np.concatenate([gl, gr], -1) / B
  th, tx, sc = T(h), T(x), T([1.0 / B])
  npart = C.c_int(0)
  L.odin_elbo_gaussian_fwd_bwd(None, None, None, None, None, B, npix, Cc, 2, C.byref(npart), None)
  part, dh = bk.zeros(B * npart.value), bk.zeros(B, npix, 2 * Cc)
  L.odin_elbo_gaussian_fwd_bwd(th.data_ptr(), tx.data_ptr(), part.data_ptr(), dh.data_ptr(),
                               sc.data_ptr(), B, npix, Cc, 2, C.byref(npart), None)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref, 2e-5)
  close(dh.cpu().numpy(), dh_ref, 1e-4)


@pytest.mark.parametrize('npix,Cc', [(784, 1), (12 * 20, 3), (300, 3)])
def test_elbo_mixture_quantized_logistic(bk, npix, Cc):
  """MixtureQuantizedLogistic head (quantized.py:206-349; image_networks.py:72-85): 10 components of
  (logit | loc | raw | channel coefficients), log-prob and gradient wrt every parameter map, ragged
  pixel counts, edge bins, one dominant component and near-uniform mixtures."""
  L, T = bk.L, bk.T
  rng = np.random.default_rng(21)
  B, K = 3, 10
  no = vo.mixql_n_out(Cc)
  h = rng.standard_normal((B, npix, K, no))
  h[..., 1 + Cc:1 + 2 * Cc] *= 2.0          # scales from ~1e-2 to ~4 (x 127.5 pixels)
  h[..., 1 + 2 * Cc:] *= 0.5                # channel coefficients
  h[0, :16, :, 0] *= 8.0                    # sharply peaked mixtures
  h[1, :16, :, 0] *= 0.01                   # near-uniform mixtures
  h = h.reshape(B, npix, K * no)
  x = rng.random((B, npix, Cc)).astype(np.float32)
  x[0, :8] = np.array([1e-6, 1 - 1e-6, 0.0, 1.0, 0.5, 200 / 255, 3 / 255, 254 / 255], np.float32)[:, None]
  x = x.astype(np.float64)
  h32 = h.astype(np.float32).astype(np.float64)
  llk_ref = vo.mixql_log_prob(h32, x, Cc, K)
  dh_ref = -vo.mixql_log_prob_grad(h32, x, Cc, K) / B
  th, tx, sc = T(h), T(x), T([1.0 / B])
  npart = C.c_int(0)
  L.odin_elbo_mixqlogistic_fwd_bwd(None, None, None, None, None, B, npix, Cc, K, C.byref(npart), None)
  part, dh = bk.zeros(B * npart.value), bk.full((B, npix, K * no), float('nan'))
  L.odin_elbo_mixqlogistic_fwd_bwd(th.data_ptr(), tx.data_ptr(), part.data_ptr(), dh.data_ptr(),
                                   sc.data_ptr(), B, npix, Cc, K, C.byref(npart), None)
  close(part.reshape(B, -1).sum(1).cpu().numpy(), llk_ref, 2e-5)
  close(dh.cpu().numpy(), dh_ref, 1e-4)
  with pytest.raises(_lib.OdinError):
    L.odin_elbo_mixqlogistic_fwd_bwd(None, None, None, None, None, B, npix, 2, K, C.byref(npart), None)


@pytest.mark.parametrize('B,P,D,N0,analytic,fb,act0,hact,draw', [
    (37, 128, 10, 128, 0, -1.0, 'linear', 'linear', False),   # dSprites bottleneck, ragged last workgroup
    (16, 256, 6, 256, 1, 0.5, 'linear', 'linear', False),     # Shapes3D, analytic KL + free bits
    (9, 96, 32, 60, 0, -1.0, 'elu', 'relu', True),            # zdim 32 (speech / MNIST), activations, device noise
    (5, 40, 3, 17, 2, -1.0, 'relu', 'elu', False),            # reverse KL, odd widths
    (4, 512, 16, 512, 0, -1.0, 'relu', 'relu', False),        # dense default nets: 512-wide layers, ~100 KB of LDS
])
def test_latent_block_fwd_bwd(bk, B, P, D, N0, analytic, fb, act0, hact, draw):
  """odin_latent_block_fwd / _bwd (the bottleneck as one launch per direction) against the oracle's
  DistributionDense -> reparameterise -> KL -> Dense formulas and their hand-written backward."""
  from odin_ai_amd._lib import ACT
  L, T = bk.L, bk.T
  rows = L.odin_latent_block_rows(B, P, D, N0)
  assert rows > 0
  rng = np.random.default_rng(5)
  h = rng.standard_normal((B, P))
  if hact == 'relu':
    h = np.maximum(h, 0)
  wl, bl = rng.standard_normal((P, 2 * D)) * 0.1, rng.standard_normal(2 * D) * 0.1
  w0, b0 = rng.standard_normal((D, N0)) * 0.3, rng.standard_normal(N0) * 0.1
  th, twl, tbl, tw0, tb0 = T(h), T(wl), T(bl), T(w0), T(b0)
  step = bk.T(np.array([7]), torch.int32)
  teps = bk.zeros(B, D)
  if draw:
    L.odin_rng_normal(teps.data_ptr(), B * D, 1234, step.data_ptr(), None)
    eps = teps.cpu().numpy().astype(np.float64)
    eps_out = bk.full((B, D), float('nan'))
  else:
    eps = rng.standard_normal((B, D))
    teps = T(eps)
    eps_out = teps
  p, z, kl, m, y0 = bk.zeros(B, 2 * D), bk.zeros(B, D), bk.zeros(B), bk.zeros(B), bk.zeros(B, N0)
  L.odin_latent_block_fwd(th.data_ptr(), twl.data_ptr(), tbl.data_ptr(), None if draw else teps.data_ptr(),
                          eps_out.data_ptr(), 1234, step.data_ptr(), p.data_ptr(), z.data_ptr(), kl.data_ptr(),
                          m.data_ptr(), tw0.data_ptr(), tb0.data_ptr(), y0.data_ptr(), B, P, D, N0, ACT[act0],
                          analytic, fb, None, None)
  if draw:  # the noise is the stream odin_rng_normal writes, bit for bit
    assert np.array_equal(eps_out.cpu().numpy(), teps.cpu().numpy())
  p_ref = h @ wl + bl
  loc, sc = vo.mvn_diag_params(p_ref, D)
  z_ref = loc + sc * eps
  if analytic == 2:
    klr = (np.log(sc) + 0.5 * (1 + loc ** 2) / sc ** 2 - 0.5).sum(-1)
  else:
    klr = vo.kl_analytic(loc, sc) if analytic else vo.kl_mc(loc, sc, z_ref)
  kl_ref, m_ref = vo.free_bits_clamp(klr, None if fb < 0 else fb, D)
  close(p.cpu().numpy(), p_ref)
  close(z.cpu().numpy(), z_ref)
  close(kl.cpu().numpy(), kl_ref)
  assert (m.cpu().numpy() == m_ref).all()
  y0_ref = vo._ACT[act0](z_ref @ w0 + b0)
  close(y0.cpu().numpy(), y0_ref)
  # ---- backward ----
  g0 = rng.standard_normal((B, N0))
  dz_extra = rng.standard_normal((B, D)) * 0.1
  klw = 4.0 / B
  dz_ref = g0 @ w0.T
  w = klw * m_ref[:, None]
  if analytic == 2:
    dloc, dsc = w * loc / sc ** 2, w * (1 / sc - (1 + loc ** 2) / sc ** 3)
  elif analytic:
    dloc, dsc = w * loc, w * (sc - 1 / sc)
  else:
    dloc, dsc = w * z_ref, w * (z_ref * eps - 1 / sc)
  gz = dz_ref + dz_extra
  dloc, dsc = dloc + gz, dsc + gz * eps
  dp_ref = np.concatenate([dloc, dsc * vo.sigmoid(p_ref[:, D:])], -1)
  hf = h.astype(np.float32).astype(np.float64)
  hgrad = {'linear': np.ones_like(hf), 'relu': (hf > 0).astype(np.float64),
           'elu': vo.elu_grad_from_output(hf)}[hact]
  dh_ref = (dp_ref @ wl.T) * hgrad
  tg0, tx, tk = T(g0), T(dz_extra), T([klw])
  dz, dp, dh = bk.zeros(B, D), bk.zeros(B, 2 * D), bk.zeros(B, P)
  s0 = bk.full((rows, D * N0 + N0), float('nan'))
  sl = bk.full((rows, P * 2 * D + 2 * D), float('nan'))
  word = bk.zeros(2048, dtype=torch.int32)
  L.odin_latent_block_bwd(tg0.data_ptr(), tw0.data_ptr(), z.data_ptr(), p.data_ptr(), eps_out.data_ptr(),
                          m.data_ptr(), tk.data_ptr(), tx.data_ptr(), None, None, twl.data_ptr(), th.data_ptr(),
                          ACT[hact], dz.data_ptr(), dp.data_ptr(), dh.data_ptr(), s0.data_ptr(), sl.data_ptr(),
                          B, P, D, N0, analytic, word.data_ptr(), None)
  assert float(word.view(torch.float32).max()) == float(dh.abs().max())  # (the range word of dh)
  close(dz.cpu().numpy(), dz_ref)
  close(dp.cpu().numpy(), dp_ref)
  close(dh.cpu().numpy(), dh_ref)
  g = s0.cpu().numpy().astype(np.float64).sum(0)
  close(g[:D * N0].reshape(D, N0), z_ref.T @ g0, 1e-4)
  close(g[D * N0:], g0.sum(0), 1e-4)
  g = sl.cpu().numpy().astype(np.float64).sum(0)
  close(g[:P * 2 * D].reshape(P, 2 * D), h.T @ dp_ref, 1e-4)
  close(g[P * 2 * D:], dp_ref.sum(0), 1e-4)
