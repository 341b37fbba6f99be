"""Whole VAE step (engine host code + kernel sources on the CPU simulator) vs oracle."""
import numpy as np
import pytest
import torch

from odin_ai_amd.engine import VAEEngine
from oracle import vae_oracle as vo
from tests.parity_util import check_engine_vs_oracle, make_case
from tests.simutil import sim_lib


@pytest.fixture(scope='module')
def L():
  return sim_lib()


def tiny_conv_spec(C=1, zdim=5):
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',),
         ('dense', 24, 'linear')]
  dec = [('dense', 32, 'linear'), ('reshape', (2, 2, 8)), ('deconv', 16, 4, 2, 'elu'),
         ('deconv', 8, 4, 2, 'elu'), ('conv', C, 1, 1, 'linear')]
  return enc, dec, (8, 8, C), zdim


def tiny16_spec(C=1, zdim=5):
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',),
         ('dense', 24, 'linear')]
  dec = [('dense', 128, 'linear'), ('reshape', (4, 4, 8)), ('deconv', 16, 4, 2, 'elu'),
         ('deconv', 8, 4, 2, 'elu'), ('conv', C, 1, 1, 'linear')]
  return enc, dec, (16, 16, C), zdim


CASES = [
    ('tiny', dict(beta=4.0), 'bernoulli', 1),
    ('tiny16_fused_tail', dict(beta=4.0), 'bernoulli', 1),
    ('tiny16_fused_tail', dict(beta=2.0), 'bernoulli', 3),
    ('tiny', dict(beta=1.0, analytic=True, free_bits=0.3), 'bernoulli', 3),
    ('tiny_tc', dict(beta=3.0, tc_beta=3.0), 'bernoulli', 3),
    ('tiny_gauss', dict(beta=2.0), 'gaussian_softplus1', 3),
    ('tiny_gauss', dict(beta=1.0), 'gaussian_softplus1', 1),
    ('tiny_gauss', dict(beta=2.0), 'qlogistic', 3),
    ('tiny', dict(beta=2.0, analytic=True, reverse=False), 'bernoulli', 1),
    ('tiny_mixql', dict(beta=2.0), 'mixqlogistic', 1),
    ('tiny_mixql', dict(beta=1.0), 'mixqlogistic', 3),
    ('mnist_dense', dict(), 'bernoulli', 1),
]


@pytest.mark.parametrize('name,kw,obs,C', CASES)
def test_engine_step_matches_oracle(L, name, kw, obs, C):
  B = 6
  if name == 'mnist_dense':
    spec = vo.mnist_dense_spec(4)
    spec = ([('flatten',), ('dense', 40, 'relu'), ('dense', 24, 'relu')],
            [('dense', 24, 'relu'), ('dense', 784, 'linear'), ('reshape', (28, 28, 1))],
            (28, 28, 1), 4)
  elif name == 'tiny_gauss':
    e, d, s, z = tiny_conv_spec(C)
    d = d[:-1] + [('conv', 2 * C, 1, 1, 'linear')]
    spec = (e, d, s, z)
  elif name == 'tiny_mixql':
    e, d, s, z = tiny_conv_spec(C)
    d = d[:-1] + [('conv', 10 * vo.mixql_n_out(C), 1, 1, 'linear')]
    spec = (e, d, s, z)
  elif name.startswith('tiny16'):
    spec = tiny16_spec(C)
  else:
    spec = tiny_conv_spec(C)
  enc, dec, in_shape, zdim, x, eps = make_case(spec, obs, B)
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation=obs, **kw)
  P = model.init_params(seed=11)
  eng = VAEEngine(enc, dec, in_shape, zdim, B, 'cpu', observation=obs,
                  analytic=kw.get('analytic', False), free_bits=kw.get('free_bits'),
                  tc='betatc' if 'tc_beta' in kw else None, lib=L, reverse=kw.get('reverse', True))
  assert eng.fused_tail == name.startswith('tiny16')
  # one pass over the activation below the 1x1 head: Gaussian observations, and Bernoulli ones without a fused tail
  conv_dec = any(l[0] in ('conv', 'deconv') for l in dec)
  assert eng.gauss_head == (conv_dec and (obs == 'gaussian_softplus1' or (obs == 'bernoulli' and not eng.fused_tail)))
  check_engine_vs_oracle(eng, model, P, x, eps, beta=kw.get('beta', 1.0), steps=2, clip=100.0)


# decoders outside the benchmark shapes: the predicate-vs-dispatch class of bug (a Conv2D k x k below a
# Conv2DTranspose wants a column-sum slab; with more than 16384 pixels per batch its data gradient leaves the implicit
# GEMM for the generic kernel, which did not keep the range word it was assumed to keep -> a zero word, 2^115, inf)
CUSTOM = [
    # (name, encoder, decoder, input shape, B)
    ('deconv_conv4_slab',
     [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',), ('dense', 24, 'linear')],
     [('dense', 16 * 16 * 8, 'linear'), ('reshape', (16, 16, 8)), ('deconv', 8, 4, 2, 'elu'),
      ('conv', 8, 4, 1, 'elu'), ('conv', 1, 1, 1, 'linear')], (32, 32, 1), 18),
    ('deconv_conv4_small',
     [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',), ('dense', 24, 'linear')],
     [('dense', 8 * 8 * 8, 'linear'), ('reshape', (8, 8, 8)), ('deconv', 8, 4, 2, 'elu'),
      ('conv', 8, 4, 1, 'elu'), ('conv', 1, 1, 1, 'linear')], (16, 16, 1), 5),
    ('five_by_five',
     [('center',), ('conv', 8, 5, 1, 'elu'), ('conv', 8, 5, 2, 'elu'), ('flatten',), ('dense', 20, 'linear')],
     [('dense', 7 * 7 * 4, 'linear'), ('reshape', (7, 7, 4)), ('deconv', 8, 5, 2, 'elu'), ('conv', 8, 5, 1, 'elu'),
      ('conv', 1, 1, 1, 'linear')], (14, 14, 1), 3),
    ('five_by_five',
     [('center',), ('conv', 8, 5, 1, 'elu'), ('conv', 8, 5, 2, 'elu'), ('flatten',), ('dense', 20, 'linear')],
     [('dense', 7 * 7 * 4, 'linear'), ('reshape', (7, 7, 4)), ('deconv', 8, 5, 2, 'elu'), ('conv', 8, 5, 1, 'elu'),
      ('conv', 1, 1, 1, 'linear')], (14, 14, 1), 11),
]


@pytest.mark.parametrize('zdim', [4, 5, 7])
def test_first_deconv_with_any_latent_width(L, zdim):
  """the flat parameter buffer packs tensors without padding, so an odd latent width leaves every
  decoder weight at an 8-byte offset; the decoders' first Conv2DTranspose (Cin 8 -> 64, k4 s2: smalldeconv.hip) used to
  refuse such weights with rc = -2 instead of reading them with scalar loads"""
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',), ('dense', 24, 'linear')]
  dec = [('dense', 128, 'linear'), ('reshape', (4, 4, 8)), ('deconv', 64, 4, 2, 'elu'), ('deconv', 8, 4, 2, 'elu'),
         ('conv', 1, 1, 1, 'linear')]
  B = 5
  enc, dec, in_shape, zd, x, eps = make_case((enc, dec, (16, 16, 1), zdim), 'bernoulli', B)
  model = vo.OracleVAE(enc, dec, in_shape, zd, observation='bernoulli', beta=2.0)
  P = model.init_params(seed=4)
  eng = VAEEngine(enc, dec, in_shape, zd, B, 'cpu', observation='bernoulli', lib=L)
  w_off = [r for r in eng.dec_recs if r.kind == 'deconv'][0].w_off
  assert (w_off % 4 != 0) == (zdim % 2 == 1)   # (the odd widths do exercise the misaligned path)
  check_engine_vs_oracle(eng, model, P, x, eps, beta=2.0, steps=2, clip=100.0)


def neck_spec(C=1, zdim=5, proj=128):
  """the neck of the dSprites / Shapes3D stacks (image_networks.py:466-471, 494-502) under a shortened encoder / decoder:
  ... -> [8, 8, 64] -> Conv2D(64, 4, 2) -> Flatten -> Dense(proj) | Dense(proj) -> (4, 4, proj / 16) -> deconv 64 -> ..."""
  enc = [('center',), ('conv', 64, 4, 2, 'elu'), ('conv', 64, 4, 2, 'elu'), ('flatten',), ('dense', proj, 'linear')]
  dec = [('dense', proj, 'linear'), ('reshape', (4, 4, proj // 16)), ('deconv', 64, 4, 2, 'elu'),
         ('deconv', 8, 4, 2, 'elu'), ('conv', C, 1, 1, 'linear')]
  return enc, dec, (16, 16, C), zdim


@pytest.mark.parametrize('B,zdim,proj,kw', [(3, 5, 128, dict(beta=2.0)), (2, 6, 256, dict(beta=1.0, analytic=True, free_bits=0.3))])
def test_neck_step_matches_oracle(L, B, zdim, proj, kw):
  """conv3 .. deconv1 as one launch per direction (neck.hip) inside a whole training step, odd batch (a workgroup with
  one sample) and odd latent width (misaligned decoder weights) included"""
  enc, dec, in_shape, zd, x, eps = make_case(neck_spec(1, zdim, proj), 'bernoulli', B)
  model = vo.OracleVAE(enc, dec, in_shape, zd, observation='bernoulli', **kw)
  P = model.init_params(seed=6)
  eng = VAEEngine(enc, dec, in_shape, zd, B, 'cpu', observation='bernoulli', lib=L,
                  analytic=kw.get('analytic', False), free_bits=kw.get('free_bits'))
  assert eng.neck
  eng.debug_check_ranges = True
  # (the 256-wide projection keeps round 5's backward launches by default -- measured faster -- here both directions run)
  eng._neck_bwd_opt = True
  check_engine_vs_oracle(eng, model, P, x, eps, beta=kw['beta'], steps=2 if proj == 128 else 1, clip=100.0)
  if proj == 256:
    # the default policy for this width: neck forward, round 5's backward launches on the tensors it left
    eng1 = VAEEngine(enc, dec, in_shape, zd, B, 'cpu', observation='bernoulli', lib=L,
                     analytic=kw.get('analytic', False), free_bits=kw.get('free_bits'))
    assert eng1.neck and eng1._neck_bwd_opt is None
    check_engine_vs_oracle(eng1, model, P, x, eps, beta=kw['beta'], steps=1, clip=100.0)
    assert not eng1._bwd_neck()


@pytest.mark.parametrize('name,enc,dec,in_shape,B', CUSTOM)
def test_custom_decoders_keep_valid_range_words(L, name, enc, dec, in_shape, B):
  """every range word the engine hands to a consumer bounds its tensor (checked before the words are cleared), the
  gradients are finite and match the oracle"""
  spec = (enc, dec, in_shape, 4)
  enc, dec, in_shape, zdim, x, eps = make_case(spec, 'bernoulli', B)
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation='bernoulli', beta=2.0)
  P = model.init_params(seed=3)
  eng = VAEEngine(enc, dec, in_shape, zdim, B, 'cpu', observation='bernoulli', lib=L)
  eng.debug_check_ranges = True
  rep = check_engine_vs_oracle(eng, model, P, x, eps, beta=2.0, steps=1, clip=100.0)
  for k, v in eng.grad_views().items():
    assert torch.isfinite(v).all(), k
  assert all(np.isfinite(v) for v in rep.values())


def _run_steps(L, ring, lrs, betas, schedule=None, n=None, clip=100.0, rows=128, fuse_norm=True, jump=None):
  """`jump` = (after step i, set step_count to v): what vae.py does when another engine ran steps in between"""
  enc, dec, in_shape, zdim = tiny_conv_spec(1)
  B = 2   # (what is compared here is the scalar plumbing: the smallest batch keeps the simulated steps short)
  eng = VAEEngine(enc, dec, in_shape, zdim, B, 'cpu', lib=L, hyper_ring=ring, hyper_ring_rows=rows,
                  fuse_norm=fuse_norm)
  assert eng.use_hyper_ring == ring
  g = torch.Generator().manual_seed(3)
  eng.params.copy_(torch.randn(eng.params.numel(), generator=g) * 0.05)
  x = torch.rand(B, *in_shape, generator=g).clamp_(1e-6, 1 - 1e-6)
  copies = 0
  orig = eng.set_hyper
  def counting(*a, **k):
    nonlocal copies
    copies += 1
    return orig(*a, **k)
  eng.set_hyper = counting
  outs = []
  for i in range(n or len(lrs)):
    if jump is not None and i == jump[0]:
      eng.step_count = jump[1]
    out = eng.train_step(x, None, lr=lrs[i % len(lrs)], beta=betas[i % len(betas)], global_clipnorm=clip,
                         schedule=schedule)
    outs.append(out.clone())
  return eng.params.clone(), torch.stack(outs), copies, eng


def test_hyper_ring_matches_per_step_copies(L):
  """The device-resident schedule (engine._ring_step, odin_sumsq_adam_ring): same parameters and losses, bit for bit,
  as the per-step host copy -- with constant hyper-parameters (no copy after the first step), across several ring
  refills, with a learning rate that changes in the middle (one more copy), with values that change every step (a copy
  per step, as before) and with a schedule known in advance (no copy after the first step)."""
  R = dict(rows=8)   # (refilled 2-4 rows at a time: 11 steps wrap it)
  n = 11
  p0, o0, c0, _ = _run_steps(L, False, [1e-3], [4.0], n=n, **R)
  p1, o1, c1, eng = _run_steps(L, True, [1e-3], [4.0], n=n, **R)
  assert torch.equal(p0, p1) and torch.equal(o0, o1)
  assert c0 == n and c1 == 1
  assert int(eng.hyper[16:17].view(torch.int32)) == n + 1   # the last Adam loaded the row of the next step
  lrs = [1e-3] * 2 + [5e-4] * 4
  p0, o0, c0, _ = _run_steps(L, False, lrs, [2.0], **R)
  p1, o1, c1, _ = _run_steps(L, True, lrs, [2.0], **R)
  assert torch.equal(p0, p1) and torch.equal(o0, o1) and c1 == 3   # first step, the change, the step that confirms it
  betas = [1.0 + 0.01 * i for i in range(4)]
  p0, o0, c0, _ = _run_steps(L, False, [1e-3], betas, n=4, **R)
  p1, o1, c1, _ = _run_steps(L, True, [1e-3], betas, n=4, **R)
  assert torch.equal(p0, p1) and torch.equal(o0, o1) and c1 == 4
  sched = lambda u: dict(beta=1.0 + 0.01 * (u - 1))
  p2, o2, c2, _ = _run_steps(L, True, [1e-3], betas, schedule=sched, n=4, **R)
  assert torch.equal(p0, p2) and torch.equal(o0, o2) and c2 == 1
  # a schedule rules the step it is asked about too: explicit arguments that contradict it do not cause a copy per step
  p3, o3, c3, _ = _run_steps(L, True, [1e-3], [9.0], schedule=sched, n=4, **R)
  assert torch.equal(p0, p3) and torch.equal(o0, o3) and c3 == 1


def test_hyper_ring_survives_a_jump_of_the_step_counter(L):
  """vae.py overwrites eng.step_count from the model's step on every call; when another batch-size
  engine (an epoch's last partial batch) or a checkpoint restore ran steps in between, the device row `hyper` still
  holds the row AFTER this engine's last step -- the mirror matches with constant hyper-parameters, so the old hit test
  ran the step with a stale row (Adam bias correction, RNG counter and when_skip_update gating of an earlier step).
  A hit now needs the previous ring-advanced step to be t - 1."""
  jump = (3, 7)   # steps 1..3, then the counter says 7: the next step is 8
  p0, o0, c0, e0 = _run_steps(L, False, [1e-3], [4.0], n=6, jump=jump)
  p1, o1, c1, e1 = _run_steps(L, True, [1e-3], [4.0], n=6, jump=jump, rows=8)
  assert e0.step_count == e1.step_count == 10
  assert torch.equal(p0, p1) and torch.equal(o0, o1)
  assert c1 == 2   # the first step and the step after the jump
  assert int(e1.hyper[16:17].view(torch.int32)) == 11


def test_fused_norm_matches_separate_launch(L):
  """The gradient norm's stage-1 launch riding in the slab reduction (odin_slab_reduce_sumsq + odin_adam_ring_parts:
  two launches where odin_slab_reduce + odin_sumsq_adam_ring are three): the same squared norm to rounding (other
  partial sums), hence bit-identical parameters while the clip does not bind and parameters equal to rounding when it
  does; the ELBO outputs (finalised from the STAGED hyper-parameter row) are bit-identical either way."""
  n = 4
  for clip, exact in ((100.0, True), (0.05, False)):
    p0, o0, _, e0 = _run_steps(L, True, [1e-3], [4.0], n=n, clip=clip, fuse_norm=False)
    assert not e0.fuse_norm
    p1, o1, c1, e1 = _run_steps(L, True, [1e-3], [4.0], n=n, clip=clip, fuse_norm=True)
    assert e1.fuse_norm and c1 == 1
    n0, n1 = float(e0.gnorm2), float(e1.gnorm2)
    assert abs(n0 - n1) <= 2e-6 * n0 and n0 > 0
    if exact:
      assert n0 < clip * clip
      assert torch.equal(p0, p1) and torch.equal(o0, o1)
    else:
      assert n0 > clip * clip
      assert float((p0 - p1).abs().max()) <= 1e-6 and float((o0 - o1).abs().max()) <= 1e-4 * float(o0.abs().max())
    assert int(e1.hyper[16:17].view(torch.int32)) == n + 1


@pytest.mark.parametrize('obs,B', [('gaussian_softplus1', 3), ('gaussian_softplus1', 2)])
def test_audio_stack_on_block_window_kernels(L, request, obs, B):
  """the audio VAE's layer shapes in miniature (rows of 20 / 10 / 5 pixels; examples/vae/vae_audio.py:84-110) with the
  block-window plane kernels taking every 32-channel 4x4 / stride-2 layer and the fused Gaussian tail (blk_planes.hip):
  decoder4 -> 1x1 head -> Normal log-prob and its backward as one launch inside a whole training step"""
  request.addfinalizer(lambda old=L.odin_debug_blk_min_flop(0.0): L.odin_debug_blk_min_flop(old))
  enc = [('conv', 32, 4, 2, 'elu'), ('conv', 32, 4, 2, 'elu'), ('flatten',), ('dense', 24, 'linear')]
  dec = [('dense', 3 * 5 * 8, 'linear'), ('reshape', (3, 5, 8)), ('deconv', 32, 4, 2, 'elu'), ('deconv', 32, 4, 2, 'elu'),
         ('conv', 2, 1, 1, 'linear')]
  enc, dec, in_shape, zd, x, eps = make_case((enc, dec, (12, 20, 1), 5), obs, B)
  model = vo.OracleVAE(enc, dec, in_shape, zd, observation=obs, beta=1.0)
  P = model.init_params(seed=9)
  eng = VAEEngine(enc, dec, in_shape, zd, B, 'cpu', observation=obs, lib=L)
  assert eng.fused_tail and eng.tail_mode == (1 if obs == 'gaussian_softplus1' else 0) and not eng.gauss_head
  check_engine_vs_oracle(eng, model, P, x, eps, beta=1.0, steps=2, clip=100.0)
  assert eng._used_fused
