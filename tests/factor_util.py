"""FactorVAE iteration (both optimisers) vs the oracle: shared by the simulator test (CPU) and
the GPU parity test.  Everything the reference's two TrainSteps produce is compared: loss, TC
estimate, the VAE gradients INCLUDING the term that flows through the discriminator into z,
the discriminator's gradients, dtc_loss, and both post-Adam parameter sets."""
import numpy as np
import torch

from oracle import vae_oracle as vo
from tests.parity_util import relerr


def check_factor_vae_iteration(fv, nets, units, B1, x, eps1, eps2, perm, lr=1e-3, tc_coef=7.0,
                               start_step=999, tol=1e-4, clip=None):
  D = fv.zdim
  in_shape = tuple(nets['encoder'].input_shape)
  fv._step = start_step  # AnnealingVAE: beta ~ 0.5 at step 1000, so that the KL term carries weight
  eng, disc = fv._engine(B1), fv._discriminator(B1)
  P = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.param_views().items()}
  DP = {(k[1], k[2]): v.detach().cpu().numpy().astype(np.float64)
        for k, v in disc.layout.views(disc.params).items()}
  t = start_step + 1
  beta = vo.interp_linear(t)
  model = vo.OracleVAE(nets['encoder'].layers, nets['decoder'].layers, in_shape, D, beta=beta)
  zero = lambda d: {k: np.zeros_like(v) for k, v in d.items()}
  dl = vo.disc_layers(units)
  ref = vo.factor_vae_iteration(model, P, zero(P), zero(P), t, dl, DP, zero(DP), zero(DP), 1,
                                x.astype(np.float64), eps1.astype(np.float64),
                                eps2.astype(np.float64), perm, lr, tc_coef=tc_coef,
                                global_clipnorm=clip)
  loss, metrics = fv.optimize(x, training=True, learning_rate=lr, eps=eps1, eps2=eps2, perm=perm,
                              global_clipnorm=clip)
  rep = {}
  rep['loss'] = abs(float(loss) - ref['loss']) / max(1.0, abs(ref['loss']))
  rep['tc'] = abs(float(metrics['elbo/tc']) - ref['tc']) / max(1.0, abs(ref['tc']))
  rep['dtc_loss'] = abs(float(metrics['disc/dtc_loss']) - ref['dtc_loss'])
  rep['extra_dz'] = relerr(disc.dz.cpu().numpy(), ref['extra_dz'])
  gv = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
  for k, g in ref['G'].items():
    rep['grad' + str(k)] = relerr(gv[k], g)
  dgv = {(k[1], k[2]): v.cpu().numpy() for k, v in disc.layout.views(disc.grads).items()}
  for k, g in ref['DG'].items():
    rep['dgrad' + str(k)] = relerr(dgv[k], g)
  # bit-exact index work: the permuted codes are a gather of the engine's own z'
  z2 = fv._engine_x2(B1).z.cpu().numpy()
  assert np.array_equal(disc.zperm.cpu().numpy(), np.take_along_axis(z2, perm.astype(np.int64), 0))
  rep['z2'] = np.abs(z2 - ref['z2']).max()
  for k, v in rep.items():
    assert v <= tol, (k, v, rep)
  # post-Adam parameters: (a) both Adam kernels on the engine's own gradients (tight), (b) end
  # to end against the float64 trajectory where the gradient is well conditioned
  pv = {k: v.cpu().numpy() for k, v in eng.param_views().items()}
  keys = list(ref['G'].keys())
  gs = 1.0
  if clip is not None:
    gs = clip / max(vo.global_norm([gv[k] for k in keys]), clip)
  for k in keys:
    pk, _, _ = vo.adam_keras(P[k], gv[k].astype(np.float64) * gs, 0.0, 0.0, t, lr)
    assert np.abs(pv[k] - pk).max() <= 2e-6 * max(1.0, np.abs(pk).max()), ('adam-kernel', k)
    d = np.abs(pv[k] - ref['P'][k])
    good = np.abs(ref['G'][k]) > 1e-3 * np.abs(ref['G'][k]).max()
    assert d[good].max() <= tol, ('param', k, d[good].max())
    assert d.mean() <= 5e-3 * lr, ('param-mean', k, d.mean())
  dpv = {(k[1], k[2]): v.cpu().numpy() for k, v in disc.layout.views(disc.params).items()}
  dlr = 1e-5
  for k in ref['DG']:
    pk, _, _ = vo.adam_keras(DP[k], dgv[k].astype(np.float64), 0.0, 0.0, 1, dlr, b1=0.5, b2=0.9)
    assert np.abs(dpv[k] - pk).max() <= 2e-6 * max(1.0, np.abs(pk).max()), ('disc-adam-kernel', k)
    d = np.abs(dpv[k] - ref['DP'][k])
    good = np.abs(ref['DG'][k]) > 1e-3 * np.abs(ref['DG'][k]).max()
    assert d[good].max() <= tol * 1e-2, ('disc-param', k, d[good].max())  # lr 1e-5: 100x tighter
    assert d.mean() <= 5e-3 * dlr, ('disc-param-mean', k, d.mean())
  rep['unmasked_param_max'] = max(np.abs(pv[k] - ref['P'][k]).max() for k in keys)
  return rep


def check_factor_vae_full_size(fv, nets, units, B1, x, eps1, eps2, perm, lr=1e-3, tc_coef=7.0,
                               start_step=999, tol=1e-4, clip=None, threads=None):
  """The same iteration at BENCHMARK size (BASELINE config 3: 128 + 128 samples, 5 x 1000 discriminator) against
  an independent float64 restatement by torch autograd on the host (oracle/torch_ref.py) -- the numpy oracle's
  hand-written conv loops are too slow there.  Compared: loss, TC estimate, dtc_loss, the gradient that flows
  through D into z, EVERY VAE gradient tensor and EVERY discriminator gradient tensor (the latter from z' of
  the already-updated encoder, factor_vae.py:279-287, evaluated in float64 from the engine's own updated
  parameters)."""
  import torch.nn.functional as F
  from oracle.torch_ref import TorchVAE, t_seq
  if threads:
    torch.set_num_threads(threads)
  D = fv.zdim
  in_shape = tuple(nets['encoder'].input_shape)
  fv._step = start_step
  eng, disc = fv._engine(B1), fv._discriminator(B1)
  P = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.param_views().items()}
  DP = {(k[1], k[2]): v.detach().cpu().numpy().astype(np.float64)
        for k, v in disc.layout.views(disc.params).items()}
  t = start_step + 1
  beta = vo.interp_linear(t)
  dl = vo.disc_layers(units)
  model = TorchVAE(nets['encoder'].layers, nets['decoder'].layers, in_shape, D, beta=beta)
  f64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
  x1, x2 = f64(x[:B1]), f64(x[B1:])
  # ---- step 1 (reference): VAE loss + tc_coef * mean(D(z)), discriminator parameters held fixed
  T = model.tensors(P)
  Dfix = {k: f64(v) for k, v in DP.items()}
  keep = {}

  def extra(o):
    o['z'].retain_grad()
    keep['z'] = o['z']
    keep['tc'] = tc_coef * t_seq(dl, Dfix, o['z'])[:, 0].mean()
    return keep['tc']

  out = model.forward(T, x1, f64(eps1), extra_loss_fn=extra)
  # the D -> z term alone (disc.dz): d tc / d z
  extra_dz, = torch.autograd.grad(keep['tc'], keep['z'], retain_graph=True)
  out['loss'].backward()
  G = {k: v.grad.detach().numpy() for k, v in T.items()}
  # ---- the iteration on the device
  loss, metrics = fv.optimize(x, training=True, learning_rate=lr, eps=eps1, eps2=eps2, perm=perm,
                              global_clipnorm=clip)
  rep = {}
  ref_loss = float(out['loss'].detach())
  rep['loss'] = abs(float(loss) - ref_loss) / max(1.0, abs(ref_loss))
  rep['tc'] = abs(float(metrics['elbo/tc']) - float(keep['tc'].detach())) / max(1.0, abs(float(keep['tc'].detach())))
  rep['extra_dz'] = relerr(disc.dz.cpu().numpy(), extra_dz.numpy())
  gv = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
  for k, g in G.items():
    rep['grad' + str(k)] = relerr(gv[k], g)
  # ---- step 2 (reference): z1 stale, z' from the engine's UPDATED parameters, permuted, dtc_loss
  P2 = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.param_views().items()}
  with torch.no_grad():
    z1 = out['z'].detach()
    z2 = model.forward(model.tensors(P2, requires_grad=False), x2, f64(eps2))['z']
    zp = torch.gather(z2, 0, torch.tensor(perm.astype(np.int64)))
  Dt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in DP.items()}
  # (the engine's own relu branches at pre-activations that are zero to rounding: oracle/torch_ref.py t_seq)
  acts = [o.detach().cpu().to(torch.float64) for o in disc.prog2.outs]
  ties1 = {i: a[:B1] for i, a in enumerate(acts)}
  ties2 = {i: a[B1:] for i, a in enumerate(acts)}
  l1 = t_seq(dl, Dt, z1, relu_ties=ties1, tie_tol=5e-5)[:, 0]
  l2 = t_seq(dl, Dt, zp, relu_ties=ties2, tie_tol=5e-5)[:, 0]
  dloss = 0.5 * (F.softplus(-l1).mean() + F.softplus(l2).mean())
  dloss.backward()
  rep['dtc_loss'] = abs(float(metrics['disc/dtc_loss']) - float(dloss.detach()))
  rep['z2'] = np.abs(fv._engine_x2(B1).z.cpu().numpy() - z2.numpy()).max()
  dgv = {(k[1], k[2]): v.cpu().numpy() for k, v in disc.layout.views(disc.grads).items()}
  for k, v in Dt.items():
    rep['dgrad' + str(k)] = relerr(dgv[k], v.grad.numpy())
  for k, v in rep.items():
    assert v <= tol, (k, v, rep)
  return rep
