"""GPU parity tests proper: the hipcc-built library on an MI355X vs the float64 oracle.

Tolerance (BASELINE.json north_star: 1e-4 fp32): absolute 1e-4 on per-latent / per-pixel
quantities (loc, raw scale, z, reconstruction), relative 1e-4 of the tensor's max magnitude
on summed quantities (llk[B], loss) and on gradients; post-Adam parameters absolute 1e-4.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import vae_oracle as vo
from tests.parity_util import check_engine_vs_oracle, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
  assert torch.cuda.is_available()
  return torch.device('cuda:0')


@pytest.fixture(scope='module')
def L():
  from odin_ai_amd import _lib
  return _lib.load()


CASES = [
    # name, spec, observation, B, model kwargs, steps
    ('dsprites_beta4', lambda: vo.dsprites_spec(1), 'bernoulli', 8, dict(beta=4.0), 2),
    ('dsprites_analytic_fb', lambda: vo.dsprites_spec(1), 'bernoulli', 5,
     dict(beta=1.0, analytic=True, free_bits=0.5), 1),
    ('shapes3d', lambda: vo.dsprites_spec(3), 'bernoulli', 6, dict(beta=1.0), 1),
    ('celeba_betatc', lambda: vo.celeba_spec(45, 3), 'bernoulli', 8, dict(beta=4.0, tc_beta=4.0), 1),
    ('celeba_gauss', lambda: vo.celeba_spec(45, 6), 'gaussian_softplus1', 4, dict(beta=2.0), 1),
    ('celeba_qlogistic', lambda: vo.celeba_spec(45, 6), 'qlogistic', 4, dict(beta=2.0), 1),
    ('dsprites_mixql', lambda: vo.dsprites_spec(1, n_out_params=30), 'mixqlogistic', 4, dict(beta=2.0), 1),
    ('shapes3d_mixql', lambda: (lambda e, d, s, z: (e, d[:-1] + [('conv', 100, 1, 1, 'linear')], s, z))(*vo.dsprites_spec(3)),
     'mixqlogistic', 3, dict(beta=1.0), 1),
    ('dsprites_reverse_kl', lambda: vo.dsprites_spec(1), 'bernoulli', 5,
     dict(beta=2.0, analytic=True, reverse=False), 1),
    ('mnist_conv', lambda: vo.mnist_conv_spec(), 'bernoulli', 6, dict(), 1),
    ('mnist_dense', lambda: vo.mnist_dense_spec(), 'bernoulli', 16, dict(), 2),
]


@pytest.mark.parametrize('name,spec,obs,B,kw,steps', CASES, ids=[c[0] for c in CASES])
def test_train_step_parity(dev, L, name, spec, obs, B, kw, steps):
  from odin_ai_amd.engine import VAEEngine
  enc, dec, in_shape, zdim, x, eps = make_case(spec(), obs, B, binary=name.startswith('mnist'))
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation=obs, **kw)
  P = model.init_params(seed=3)
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation=obs,
                  analytic=kw.get('analytic', False), free_bits=kw.get('free_bits'),
                  tc='betatc' if 'tc_beta' in kw else None, lib=L, reverse=kw.get('reverse', True))
  rep = check_engine_vs_oracle(eng, model, P, x, eps, beta=kw.get('beta', 1.0), steps=steps,
                               clip=100.0)
  print(name, {k: f'{v:.2e}' for k, v in rep.items() if not k.startswith('grad')})


def test_full_batch_config2_forward_properties(dev, L):
  """BASELINE config 2 at full size (B=256): size-independent properties.
  (1) zero weights -> logits 0 -> llk = -H*W*C*ln2 for every sample (closed form);
  (2) the HIP-graph replay of a step is bit-identical to the eager launch sequence;
  (3) loss decreases over a few Adam steps on a fixed batch."""
  from odin_ai_amd.engine import VAEEngine
  enc, dec, in_shape, zdim = vo.dsprites_spec(1)
  B = 256
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, lib=L)
  x = torch.rand(B, *in_shape, device=dev).clamp_(1e-6, 1 - 1e-6)
  eng.step_count = 1
  eng.set_hyper(lr=1e-3, beta=4.0)
  eng.forward(x)
  torch.cuda.synchronize()
  assert torch.allclose(eng.llk, torch.full_like(eng.llk, -64 * 64 * np.log(2.0)), rtol=1e-6)
  # random init, eager vs graph
  g = torch.Generator(device='cpu').manual_seed(0)
  for (k, shp, off) in eng.layout.entries:
    n = int(np.prod(shp))
    fan = max(1, n // shp[-1])
    eng.params[off:off + n] = (torch.randn(n, generator=g) * (2.0 / fan) ** 0.5 * 0.5).to(dev)
  p0 = eng.params.clone()
  eng.step_count = 0
  eps = torch.randn(B, zdim, device=dev)
  losses = []
  for t in range(4):
    out = eng.train_step(x, eps, lr=1e-3, beta=4.0)
    losses.append(out[0].item())
  p_eager = eng.params.clone()
  eng2 = VAEEngine(enc, dec, in_shape, zdim, B, dev, lib=L)
  eng2.params.copy_(p0)
  for t in range(4):
    eng2.train_step(x, eps, lr=1e-3, beta=4.0, use_graph=True)
  torch.cuda.synchronize()
  assert torch.equal(p_eager, eng2.params), (p_eager - eng2.params).abs().max()
  assert losses[-1] < losses[0], losses


def test_missing_library_fails_loudly(tmp_path):
  from odin_ai_amd import _lib
  with pytest.raises(_lib.OdinError):
    _lib.Lib(str(tmp_path / 'nope.so'))


def test_mel_frontend_matches_reference_golden(dev, L):
  """Speech front-end (BASELINE config 5 front half) vs arrays produced by the reference's
  own signal.py (tests/golden/mel_golden.npz).  The kernel computes in float64 like the
  reference and stores float32: tolerance 2e-5 dB absolute on values of magnitude up to ~100 dB
  (2e-7 relative; north_star's 1e-4 relative would allow 1e-2 dB)."""
  import os
  from odin_ai_amd.mel import MelsSpecExtractor
  G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mel_golden.npz'))
  ex = MelsSpecExtractor(device=dev, lib=L)
  out = ex(G['y']).cpu().numpy()
  assert out.shape == (3, 98, 80)
  for i in range(3):
    assert np.abs(out[i] - G[f'mel_db_{i}']).max() < 2e-5
  # batch 256 (BASELINE size): every utterance of a repeated batch gives the same answer
  yb = torch.tensor(G['y'][:1]).repeat(256, 1).to(dev)
  ob = ex(yb)
  assert torch.equal(ob[0], ob[255]) and np.abs(ob[0].cpu().numpy() - G['mel_db_0']).max() < 2e-5


def test_model_api_on_gpu(dev, L):
  """get_networks -> BetaVAE.fit (HIP-graph replay) lowers the loss; FactorVAE iteration
  runs both optimisers; encode/decode round trip shapes."""
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import BetaVAE, FactorVAE
  vae = BetaVAE(beta=4.0, device=dev, lib=L, **get_networks('dsprites'))
  x = (torch.rand(64, 64, 64, 1, device=dev) < 0.1).float().clamp(1e-6, 1 - 1e-6)
  l0, _ = vae.optimize(x[:32], training=False)
  vae.fit(x, max_iter=30, batch_size=32, learning_rate=1e-3, compile_graph=True)
  l1, _ = vae.optimize(x[:32], training=False)
  assert vae.step == 30 and float(l1) < float(l0)
  px, qz = vae(x[:8])
  assert px.mean().shape == (8, 64, 64, 1) and qz.sample().shape == (8, 10)
  fv = FactorVAE(device=dev, lib=L, **get_networks('shapes3d'))
  xs = torch.rand(16, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
  loss, m = fv.optimize(xs, learning_rate=1e-4)
  assert fv.step == 1 and set(m) >= {'elbo/tc', 'disc/dtc_loss'}
  assert np.isfinite(float(loss)) and abs(float(m['disc/dtc_loss']) - np.log(2.0)) < 0.5


def test_full_size_configs_3_and_4(dev, L):
  """BASELINE configs 3 (FactorVAE, Shapes3D, batch 256 = 128 + 128) and 4 (beta-TCVAE,
  CelebA stack, batch 512) at full size: finite, both optimisers move, loss decreases on a
  fixed batch, TC of the beta-TC estimator is finite."""
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import BetaTCVAE, FactorVAE
  torch.manual_seed(0)
  fv = FactorVAE(device=dev, lib=L, **get_networks('shapes3d'))
  x = torch.rand(256, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
  losses = []
  for _ in range(6):
    loss, m = fv.optimize(x, learning_rate=2e-4, global_clipnorm=100.0)
    losses.append(float(loss))
  assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
  assert np.isfinite(float(m['disc/dtc_loss'])) and np.isfinite(float(m['elbo/tc']))
  tcv = BetaTCVAE(beta=4.0, device=dev, lib=L, **get_networks('celeba'))
  xc = torch.rand(512, 64, 64, 3, device=dev).clamp(1e-6, 1 - 1e-6)
  l0 = None
  for _ in range(5):
    loss, m = tcv.optimize(xc, learning_rate=2e-4, global_clipnorm=100.0)
    l0 = float(loss) if l0 is None else l0
  assert np.isfinite(float(loss)) and float(loss) < l0
  assert np.isfinite(float(m['tc_latents']))


def test_device_input_pipeline(dev, L):
  """uint8 dataset resident in HBM -> batch gather + ImageDataset.normalize on the
  device, bit-exact against the numpy restatement at dSprites size; the batches land in the
  tensor the step graph reads and fit() trains from them."""
  from odin_ai_amd.data import DeviceImageDataset
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import BetaVAE
  from oracle import data_oracle as do
  rng = np.random.default_rng(0)
  imgs = (rng.random((1024, 64, 64, 1)) < 0.15).astype(np.uint8)  # dSprites-like 0/1 pixels
  ds = DeviceImageDataset(imgs, batch_size=256, normalize='probs', premul=255.0, device=dev, lib=L)
  idx = rng.permutation(1024)[:256].astype(np.int32)
  got = ds.gather(torch.from_numpy(idx)).cpu().numpy()
  assert np.array_equal(got, do.gather_normalize(imgs, idx, 'probs', 255.0))
  g8 = rng.integers(0, 256, size=(64, 64, 64, 3), dtype=np.uint8)
  for mode in ('probs', 'tanh', 'raster'):
    d2 = DeviceImageDataset(g8, batch_size=8, normalize=mode, device=dev, lib=L)
    i2 = np.arange(8, dtype=np.int32)[::-1].copy()
    assert np.array_equal(d2.gather(torch.from_numpy(i2)).cpu().numpy(), do.gather_normalize(g8, i2, mode))
  vae = BetaVAE(beta=4.0, device=dev, lib=L, **get_networks('dsprites'))
  ds = DeviceImageDataset(imgs, batch_size=64, normalize='probs', premul=255.0, device=dev, lib=L,
                          out=vae.input_buffer(64))
  x0 = ds.gather(torch.arange(64, dtype=torch.int32)).clone()
  l0, _ = vae.optimize(x0, training=False)
  vae.fit(ds, max_iter=40, batch_size=64, learning_rate=1e-3, compile_graph=True)
  l1, _ = vae.optimize(x0, training=False)
  assert vae.step == 40 and float(l1) < float(l0)


# ------------------------------------------------------------------------------------------
# round 2: FactorVAE backward, raw-scale Gaussian, NaN policy on device, full-size gradients
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('units', [(1000,) * 5, (128, 128)], ids=['5x1000', '2x128'])
def test_factor_vae_iteration_parity(dev, L, units):
  """BASELINE config 3's iteration (factor_vae.py:239-287) on the Shapes3D stack: VAE gradients
  with the term that flows through D into z, discriminator gradients, dtc_loss, permute_dims
  bit-exact, and BOTH post-Adam parameter sets (Adam(lr) and Adam(1e-5,.5,.9)) vs the oracle.
  (128,128) ends in Dense(128 -> 1): the narrow-head tiny-Dense path."""
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import FactorVAE
  from tests.factor_util import check_factor_vae_iteration
  nets = get_networks('shapes3d')
  B1, D = 4, 6
  fv = FactorVAE(discriminator_units=units, tc_coef=7.0, device=dev, lib=L, **nets)
  rng = np.random.default_rng(21)
  x = np.clip(rng.random((2 * B1, 64, 64, 3)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  rep = check_factor_vae_iteration(fv, nets, units, B1, x, eps, eps2, perm, lr=1e-3, clip=100.0)
  print('factor_vae', units, {k: f'{v:.2e}' for k, v in rep.items() if 'grad' not in k})


def test_factor_vae_full_size_gradients(dev, L):
  """BASELINE config 3 at its own size (batch 256 = 128 + 128, discriminator 5 x 1000: factor_vae.py:150-176,
  239-287): the igemm instances of the [128 | 256] x 1000 x 1000 Dense layers and the 128-sample conv stack,
  every VAE gradient (with the term that flows through D into z) and every discriminator gradient against
  float64 torch autograd."""
  import os
  from odin_ai_amd.networks import get_networks
  from odin_ai_amd.vae import FactorVAE
  from tests.factor_util import check_factor_vae_full_size
  nets = get_networks('shapes3d')
  B1, D, units = 128, 6, (1000,) * 5
  fv = FactorVAE(discriminator_units=units, tc_coef=7.0, device=dev, lib=L, **nets)
  rng = np.random.default_rng(33)
  x = np.clip(rng.random((2 * B1, 64, 64, 3)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  rep = check_factor_vae_full_size(fv, nets, units, B1, x, eps, eps2, perm, lr=1e-3, clip=100.0,
                                   threads=min(32, os.cpu_count() or 1))
  worst = max(((k, v) for k, v in rep.items() if 'grad' in k), key=lambda kv: kv[1])
  print('factor_vae_b256', {k: f'{v:.2e}' for k, v in rep.items() if 'grad' not in k}, 'worst grad', worst)


def test_gaussian_raw_scale_parity(dev, L):
  """a11's primary branch (image_networks.py:95-102): Normal(loc, scale) with the scale taken RAW
  from the decoder (no positivity transform).  The scale maps' bias is set positive so that the
  log-prob is finite, as a trained model's would be."""
  from odin_ai_amd.engine import VAEEngine
  enc, dec, in_shape, zdim, x, eps = make_case(vo.celeba_spec(45, 6), 'gaussian', 4)
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation='gaussian', beta=2.0)
  P = model.init_params(seed=3)
  last = max(k[1] for k in P if k[0] == 'dec')
  P[('dec', last, 'w')] = P[('dec', last, 'w')] * 0.05
  P[('dec', last, 'b')][3:] = 1.5
  f = model.forward(P, x, eps)
  assert f['h_d'][..., 3:].min() > 0.5, 'test premise: raw scales must be positive'
  eng = VAEEngine(enc, dec, in_shape, zdim, 4, dev, observation='gaussian', lib=L)
  rep = check_engine_vs_oracle(eng, model, P, x, eps, beta=2.0, clip=100.0)
  print('gaussian_raw', {k: f'{v:.2e}' for k, v in rep.items() if not k.startswith('grad')})


def test_nan_gradients_skip_the_update_on_device(dev, L):
  """Networks.optimize NaN policy (base_networks.py:519-547): a non-finite gradient must leave
  parameters, m and v untouched and raise the device flag -- eager and graph-replayed."""
  from odin_ai_amd.engine import VAEEngine
  enc, dec, in_shape, zdim = vo.dsprites_spec(1)
  for use_graph in (False, True):
    eng = VAEEngine(enc, dec, in_shape, zdim, 8, dev, lib=L)
    g = torch.Generator(device='cpu').manual_seed(0)
    eng.params.copy_((torch.randn(eng.params.numel(), generator=g) * 0.05).to(dev))
    x = torch.rand(8, *in_shape, device=dev).clamp_(1e-6, 1 - 1e-6)
    eng.train_step(x, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=use_graph)
    torch.cuda.synchronize()
    assert eng.flag.item() == 0
    p, m, v = eng.params.clone(), eng.m.clone(), eng.v.clone()
    xb = x.clone()
    xb[3, 10, 10, 0] = float('nan')
    eng.train_step(xb, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=use_graph)
    torch.cuda.synchronize()
    assert eng.flag.item() == 1
    assert torch.equal(p, eng.params) and torch.equal(m, eng.m) and torch.equal(v, eng.v)
    eng.flag.zero_()
    eng.train_step(x, None, lr=1e-3, beta=4.0, global_clipnorm=100.0, use_graph=use_graph)
    torch.cuda.synchronize()
    assert eng.flag.item() == 0 and not torch.equal(p, eng.params)


FULL = [
    # name, spec, B, kwargs, env
    ('dsprites_b256_default', lambda: vo.dsprites_spec(1), 256, dict(beta=4.0), {}),
    ('dsprites_b256_fp32_mfma_only', lambda: vo.dsprites_spec(1), 256, dict(beta=4.0),
     {'ODIN_EXACT_FP32': '1'}),
    # the engine's opt-in launch orders (all OFF by default, engine.py): weight gradients of the small layers on side
    # streams + the decoder's slabs reduced early; every weight gradient on side streams; the plane weight
    # gradients of the step as one deferred launch -- same arithmetic, other launch order
    # (engine keyword arguments since round 6: the 'ENGINE' entry is passed to VAEEngine, everything else is environment)
    ('dsprites_b256_overlap_small_early_reduce', lambda: vo.dsprites_spec(1), 256, dict(beta=4.0),
     {'ENGINE': dict(overlap_wgrad='small', early_reduce=True)}),
    ('dsprites_b256_defer_wgrad', lambda: vo.dsprites_spec(1), 256, dict(beta=4.0), {'ENGINE': dict(defer_wgrad=True)}),
    ('shapes3d_b128', lambda: vo.dsprites_spec(3), 128, dict(beta=1.0), {}),
    ('celeba_b512', lambda: vo.celeba_spec(45, 3), 512, dict(beta=4.0), {}),
    ('celeba_betatc_b512', lambda: vo.celeba_spec(45, 3), 512, dict(beta=4.0, tc_beta=4.0), {}),
    # BASELINE config 1 at its batch size (dense default nets and the mnist conv stack) and config 5's conv VAE
    # on [96, 80, 1] mel patches with the Gaussian (softplus1) observation at batch 256
    ('mnist_dense_b128', lambda: vo.mnist_dense_spec(), 128, dict(beta=1.0), {}),
    ('mnist_conv_b128', lambda: vo.mnist_conv_spec(), 128, dict(beta=1.0), {}),
    ('speech_b256', 'speech', 256, dict(beta=1.0, observation='gaussian_softplus1'), {}),
]


@pytest.mark.parametrize('name,spec,B,kw,env', FULL, ids=[c[0] for c in FULL])
def test_full_batch_gradients_vs_f64_autograd(dev, L, monkeypatch, name, spec, B, kw, env):
  """the kernel variants picked at BASELINE batch sizes (producer/consumer wgrad, tile
  splitting, intra-workgroup split-K, side-stream early reduce, bf16-split instances) compared
  with an independent float64 restatement (torch autograd on the host, oracle/torch_ref.py):
  loss, llk[B], kl[B] and EVERY gradient tensor within 1e-4 of the tensor's maximum."""
  import os
  from odin_ai_amd.engine import VAEEngine
  from oracle.torch_ref import TorchVAE
  env = dict(env)
  eopts = env.pop('ENGINE', {})
  for k, v in env.items():
    monkeypatch.setenv(k, v)
    os.putenv(k, v)
  if spec == 'speech':
    from odin_ai_amd.networks import get_networks
    nets = get_networks('speech', n_frames=96, n_mels=80)
    enc, dec = nets['encoder'].layers, nets['decoder'].layers
    in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
  else:
    enc, dec, in_shape, zdim = spec()
  obs = kw.get('observation', 'bernoulli')
  rng = np.random.default_rng(5)
  x = np.clip(rng.random((B,) + tuple(in_shape)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, zdim)).astype(np.float32)
  ref = TorchVAE(enc, dec, in_shape, zdim, **kw)
  om = vo.OracleVAE(enc, dec, in_shape, zdim, **kw)
  P = om.init_params(seed=9)
  P = {k: v.astype(np.float32).astype(np.float64) for k, v in P.items()}
  torch.set_num_threads(min(32, os.cpu_count() or 1))
  out, G = ref.loss_and_grads(P, x.astype(np.float64), eps.astype(np.float64))
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, tc='betatc' if 'tc_beta' in kw else None,
                  observation=obs, lib=L, **eopts)
  eng.load_params(P)
  eng.step_count = 1
  eng.set_hyper(lr=1e-3, beta=kw['beta'])
  eng.forward(torch.tensor(x, device=dev), torch.tensor(eps, device=dev))
  eng.backward()
  torch.cuda.synchronize()
  for k in env:
    os.unsetenv(k)
  from tests.parity_util import relerr
  rep = dict(loss=abs(eng.out4[0].item() - float(out['loss'])) / abs(float(out['loss'])),
             llk=relerr(eng.llk.cpu().numpy(), out['llk']),
             kl=relerr(eng.kl.cpu().numpy() * kw['beta'], out['kl']))
  gv = {k: v.cpu().numpy() for k, v in eng.grad_views().items()}
  worst = ('', 0.0)
  for k in G:
    e = relerr(gv[k], G[k])
    rep['grad' + str(k)] = e
    if e > worst[1]:
      worst = (k, e)
  print(name, f"loss {rep['loss']:.2e} llk {rep['llk']:.2e} kl {rep['kl']:.2e} worst grad {worst}")
  gtol = 1e-4
  if 'tc_beta' in kw:
    # The minibatch TC estimator is ILL-CONDITIONED in (loc, scale, z) at random initialisation
    # (posterior scales down to 4e-3 beside locations of order 1-7): perturbing the float64
    # oracle's own inputs by the fp32 rounding of p (<= 1e-5 absolute) moves its TC gradients by
    # 2-3e-4 of their maximum (measured: profiles/r02_tc_conditioning.txt).  Any fp32 evaluation
    # -- the TF reference included -- sits that far from float64.  So (a) the TC kernels are
    # held to 1e-5 against the oracle evaluated on the ENGINE's own (p, z), and (b) the
    # end-to-end gradients, which inherit the conditioning, to 1e-3.
    pz = eng.p.cpu().numpy().astype(np.float64), eng.z.cpu().numpy().astype(np.float64)
    loc, sc = vo.mvn_diag_params(pz[0], zdim)
    c = kw['tc_beta'] - 1.0
    for got, want in zip((eng.tc_dz, eng.tc_dloc, eng.tc_dscale),
                         vo.total_correlation_bwd(pz[1], loc, sc)):
      assert relerr(got.cpu().numpy(), c * want) <= 1e-5
    assert abs(eng.tc_ws[0].item() - vo.total_correlation(pz[1], loc, sc)) <= 1e-5 * 300
    gtol = 1e-3
  for k, v in rep.items():
    assert v <= (gtol if k.startswith('grad') else 1e-4), (k, v)


def test_speech_vae_config5_pipeline(dev, L):
  """BASELINE config 5 end to end: raw audio [B, 8000] -> pre-emphasis / STFT / 80 Slaney mels /
  dB (HIP, float64 like the reference) -> first 96 frames -> conv VAE with the Gaussian
  (softplus1) observation of examples/vae/vae_audio.py:84-90 -> one training step.  The
  spectrogram is held to the reference-pinned mel oracle, the step to the VAE oracle fed with
  the ORACLE's spectrogram."""
  from odin_ai_amd.engine import VAEEngine
  from odin_ai_amd.mel import MelsSpecExtractor, spectrogram_batch
  from odin_ai_amd.networks import get_networks
  from oracle import mel_oracle as mo
  B, T = 4, 96
  rng = np.random.default_rng(1)
  t = np.arange(8000) / 8000.0
  y = (0.1 * rng.standard_normal((B, 8000)) +
       0.5 * np.sin(2 * np.pi * (200.0 + 1500.0 * t[None] * (1 + np.arange(B)[:, None])) * t[None])
       ).astype(np.float32)
  ex = MelsSpecExtractor(device=dev, lib=L)
  ref_db = np.stack([mo.mel_frontend(y[i]) for i in range(B)])
  assert np.abs(ex(y).cpu().numpy() - ref_db).max() < 2e-5
  # the conv stack has no batch normalisation (the reference's audio example has one after every
  # conv, vae_audio.py:96-110), so the spectrogram enters in unit range: (dB - max)/top_db + 1
  ex = MelsSpecExtractor(device=dev, lib=L, unit_range=True)
  mel = ex(y)
  assert tuple(mel.shape) == (B, 98, 80)
  ref = (ref_db - ref_db.max(axis=(1, 2), keepdims=True)) / 80.0 + 1.0
  assert np.abs(mel.cpu().numpy() - ref).max() < 1e-6
  x = spectrogram_batch(mel, T)
  nets = get_networks('speech', n_frames=T, n_mels=80)
  enc, dec = nets['encoder'].layers, nets['decoder'].layers
  in_shape, zdim = nets['encoder'].input_shape, nets['latents'].event_shape[0]
  assert in_shape == (96, 80, 1) and zdim == 32
  model = vo.OracleVAE(enc, dec, in_shape, zdim, observation='gaussian_softplus1', beta=1.0)
  P = model.init_params(seed=4)
  eps = rng.standard_normal((B, zdim))
  eng = VAEEngine(enc, dec, in_shape, zdim, B, dev, observation='gaussian_softplus1', lib=L)
  xr = np.ascontiguousarray(ref[:, :T]).reshape(B, T, 80, 1)
  # engine runs on ITS OWN spectrogram, the oracle on the float64 one
  eng.load_params(P)
  rep = check_engine_vs_oracle(eng, model, P, xr, eps, beta=1.0, clip=100.0)
  eng.load_params(model.init_params(seed=4))
  eng.step_count = 1
  eng.set_hyper(lr=1e-3, beta=1.0)
  eng.forward(x.contiguous(), torch.tensor(eps, dtype=torch.float32, device=dev))
  f = model.forward(model.init_params(seed=4), xr, eps)
  assert abs(eng.out4[0].item() - f['loss']) <= 1e-4 * abs(f['loss'])
  print('speech', {k: f'{v:.2e}' for k, v in rep.items() if not k.startswith('grad')})
  # and the model API trains on it
  from odin_ai_amd.vae import VariationalAutoencoder
  vae = VariationalAutoencoder(device=dev, lib=L, **nets)
  xb = x.repeat(8, 1, 1, 1)
  l0, _ = vae.optimize(xb, training=False)
  vae.fit(xb, max_iter=20, batch_size=32, learning_rate=1e-3, global_clipnorm=100.0)
  l1, _ = vae.optimize(xb, training=False)
  assert float(l1) < float(l0)


def test_conv_forward_beyond_the_fill_table_bound(dev, L):
  """A batch whose tiles per workgroup exceed the LDS fill tables of fconv_planes (168 tiles at 32-pixel output rows:
  more than 1344 images of 64x64x32 on 256 CUs) runs on more workgroups than CUs: the result must be the one of the
  two half batches, bit for bit (a tile's arithmetic does not depend on the workgroup that owns it)."""
  from odin_ai_amd import _lib
  B, H, W, Ci, Co = 1400, 64, 64, 32, 32
  g = torch.Generator(device='cpu').manual_seed(5)
  x = torch.randn(B, H, W, Ci, generator=g).to(dev)
  w = (torch.randn(4, 4, Ci, Co, generator=g) * 0.1).to(dev)
  b = (torch.randn(Co, generator=g) * 0.1).to(dev)

  def fwd(xs):
    n = xs.shape[0]
    d = _lib.conv_desc(n, H, W, Ci, 32, 32, Co, 4, 2, 1, 1, 'elu')
    y = torch.empty(n, 32, 32, Co, device=dev)
    L.odin_conv2d_fwd(xs.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), C.byref(d), None)
    return y, L.odin_debug_last_path().decode()

  y, path = fwd(x)
  assert path.startswith('fconv_planes'), path
  y0, _ = fwd(x[:700].contiguous())
  y1, _ = fwd(x[700:].contiguous())
  torch.cuda.synchronize()
  assert torch.equal(y[:700], y0) and torch.equal(y[700:], y1)
  # and against a float64 reference on a few images across the batch (first, seam of the halves, last)
  idx = [0, 699, 700, 1399]
  ref = torch.nn.functional.elu(torch.nn.functional.conv2d(
      x[idx].double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), b.double(), stride=2, padding=1))
  err = (y[idx].double() - ref.permute(0, 2, 3, 1)).abs().max().item()
  assert err < 1e-4 * ref.abs().max().item(), err
