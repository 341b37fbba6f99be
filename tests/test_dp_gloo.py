"""N>1 path on CPU: two gloo ranks (kernel sources on the simulator) == one rank on the
full batch.  Covers the flat-bucket all-reduce, the 1/B_global loss scaling and the
replicated Adam step of `VAEEngine`."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spec():
  enc = [('center',), ('conv', 8, 4, 2, 'elu'), ('conv', 16, 4, 2, 'elu'), ('flatten',),
         ('dense', 24, 'linear')]
  dec = [('dense', 32, 'linear'), ('reshape', (2, 2, 8)), ('deconv', 16, 4, 2, 'elu'),
         ('deconv', 8, 4, 2, 'elu'), ('conv', 1, 1, 1, 'linear')]
  return enc, dec, (8, 8, 1), 4


def _data(B):
  rng = np.random.default_rng(5)
  x = np.clip(rng.random((B, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps = rng.standard_normal((B, 4)).astype(np.float32)
  return torch.tensor(x), torch.tensor(eps)


def _init(eng):
  g = torch.Generator().manual_seed(0)
  eng.params.copy_(torch.randn(eng.params.numel(), generator=g) * 0.1)


def _worker(rank, world, port, out_path, tc=None, buckets=None, segmented=False):
  sys.path.insert(0, ROOT)
  from odin_ai_amd import _lib
  from odin_ai_amd.dist import shard_batch
  from odin_ai_amd.engine import VAEEngine
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  L = _lib.Lib(os.path.join(ROOT, 'tests', 'sim', 'libodin_sim.so'))
  enc, dec, shp, D = _spec()
  B = 8
  x, eps = _data(B)
  eng = VAEEngine(enc, dec, shp, D, B // world, 'cpu', lib=L, world_size=world, tc=tc, dp_buckets=buckets)
  _init(eng)
  xs, es = shard_batch(x, rank, world), shard_batch(eps, rank, world)
  kinds = None
  for _ in range(3):
    if segmented:
      # the graph-replayed step's own machinery (dist.SegmentedGraph over step_program), with the kernel
      # segments launched instead of replayed (no HIP graphs on the CPU)
      from odin_ai_amd.dist import SegmentedGraph
      eng.step_count += 1
      eng.set_hyper(lr=1e-3, beta=4.0)
      sg = SegmentedGraph('cpu', eng.step_program(xs, es, (100.0, None, None, None, True)))
      kinds = ''.join(k for k, _ in sg.segs)
      sg.run_eager()
    else:
      eng.train_step(xs, es, lr=1e-3, beta=4.0, global_clipnorm=100.0)
  if rank == 0:
    torch.save(dict(params=eng.params.clone(), out4=eng.out4.clone(), kinds=kinds), out_path)
  dist.destroy_process_group()


def test_two_gloo_ranks_match_single_rank(tmp_path):
  from odin_ai_amd.engine import VAEEngine
  from tests.simutil import sim_lib
  L = sim_lib()
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  out = str(tmp_path / 'p2.pt')
  mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
  p2 = torch.load(out)['params']
  enc, dec, shp, D = _spec()
  x, eps = _data(8)
  eng = VAEEngine(enc, dec, shp, D, 8, 'cpu', lib=L, world_size=1)
  _init(eng)
  for _ in range(3):
    eng.train_step(x, eps, lr=1e-3, beta=4.0, global_clipnorm=100.0)
  d = (eng.params - p2).abs()
  # identical maths, different fp32 summation order (per-rank partial sums): Adam's
  # normalisation bounds the per-step drift by a fraction of lr
  assert d.max().item() < 2e-4 and d.mean().item() < 2e-6, (d.max().item(), d.mean().item())


def test_beta_tc_two_ranks_match_single_rank(tmp_path):
  """the total-correlation estimator couples every pair of the batch, so under data
  parallelism it is evaluated on the GLOBAL batch (all-gather of (p, z), this rank's rows against
  all posteriors, reduce-scatter of the posterior-side gradients): 2 ranks x 4 samples must
  reproduce 1 rank x 8 samples -- TC value, loss and parameters after 3 Adam steps."""
  from odin_ai_amd.engine import VAEEngine
  from tests.simutil import sim_lib
  L = sim_lib()
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  out = str(tmp_path / 'tc2.pt')
  mp.spawn(_worker, args=(2, port, out, 'betatc'), nprocs=2, join=True)
  r2 = torch.load(out)
  enc, dec, shp, D = _spec()
  x, eps = _data(8)
  eng = VAEEngine(enc, dec, shp, D, 8, 'cpu', lib=L, world_size=1, tc='betatc')
  _init(eng)
  for _ in range(3):
    eng.train_step(x, eps, lr=1e-3, beta=4.0, global_clipnorm=100.0)
  # out4 = [loss, mean llk, mean beta*kl, tc]: llk / kl are per-rank means (ranks differ), the TC
  # term is global and must agree
  assert abs(eng.out4[3].item() - r2['out4'][3].item()) < 1e-4 * max(1.0, abs(eng.out4[3].item()))
  d = (eng.params - r2['params']).abs()
  assert d.max().item() < 2e-4 and d.mean().item() < 2e-6, (d.max().item(), d.mean().item())


@pytest.mark.parametrize('tc,buckets,kinds', [(None, 2, 'kckck'), ('betatc', 1, 'kckckck'), ('betatc', 2, 'kckckckck')])
def test_segmented_program_two_ranks_match_single_rank(tmp_path, tc, buckets, kinds):
  """the data-parallel step as kernel segments around its collectives
  (A | all-gather | B | reduce-scatter | C | all-reduce | D; two gradient buckets: decoder bucket reduced
  before the encoder's backward pass ends) gives the single-rank result."""
  from odin_ai_amd.engine import VAEEngine
  from tests.simutil import sim_lib
  L = sim_lib()
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  out = str(tmp_path / 'seg.pt')
  mp.spawn(_worker, args=(2, port, out, tc, buckets, True), nprocs=2, join=True)
  r2 = torch.load(out)
  assert r2['kinds'] == kinds, r2['kinds']
  enc, dec, shp, D = _spec()
  x, eps = _data(8)
  eng = VAEEngine(enc, dec, shp, D, 8, 'cpu', lib=L, world_size=1, tc=tc)
  _init(eng)
  for _ in range(3):
    eng.train_step(x, eps, lr=1e-3, beta=4.0, global_clipnorm=100.0)
  d = (eng.params - r2['params']).abs()
  assert d.max().item() < 2e-4 and d.mean().item() < 2e-6, (d.max().item(), d.mean().item())


def _factor_nets():
  from odin_ai_amd.networks import RVconf, SequentialNetwork
  enc, dec, shp, D = _spec()
  return dict(encoder=SequentialNetwork(enc, 'Encoder', shp), decoder=SequentialNetwork(dec, 'Decoder', (D,)),
              observation=RVconf(shp, 'bernoulli', projection=False, name='image'),
              latents=RVconf((D,), 'mvndiag', projection=True, name='latents'))


def _factor_data():
  rng = np.random.default_rng(6)
  B1, D = 4, 4
  x = np.clip(rng.random((2 * B1, 8, 8, 1)), 1e-6, 1 - 1e-6).astype(np.float32)
  eps, eps2 = (rng.standard_normal((B1, D)).astype(np.float32) for _ in range(2))
  perm = np.stack([rng.permutation(B1) for _ in range(D)], 1).astype(np.int32)
  return B1, x, eps, eps2, perm


def _factor_run(fv, x, eps, eps2, perm):
  fv._step = 999
  for _ in range(2):
    loss, m = fv.optimize(x, learning_rate=1e-3, eps=eps, eps2=eps2, perm=perm, global_clipnorm=100.0)
  return fv._params.clone(), fv.discriminator.params.clone(), float(m['disc/dtc_loss'])


def _factor_worker(rank, world, port, out_path):
  sys.path.insert(0, ROOT)
  from odin_ai_amd import _lib
  from odin_ai_amd.vae import FactorVAE
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  L = _lib.Lib(os.path.join(ROOT, 'tests', 'sim', 'libodin_sim.so'))
  B1, x, eps, eps2, perm = _factor_data()
  h = B1 // world
  sl = slice(rank * h, (rank + 1) * h)
  # this rank's shard: its rows of x1 followed by its rows of x2; the permutation stays GLOBAL
  xs = np.concatenate([x[:B1][sl], x[B1:][sl]])
  fv = FactorVAE(discriminator_units=(16, 16), device='cpu', lib=L, **_factor_nets())
  p, dpar, dtc = _factor_run(fv, xs, eps[sl], eps2[sl], perm)
  if rank == 0:
    torch.save(dict(p=p, d=dpar), out_path)
  dist.destroy_process_group()


def test_factor_vae_two_ranks_match_single_rank(tmp_path):
  """FactorVAE under data parallelism: global permute_dims (all-gather of z'), both gradient
  buckets all-reduced, replicas start from rank 0's weights -- 2 ranks x (2+2) == 1 rank x (4+4)."""
  from odin_ai_amd.vae import FactorVAE
  from tests.simutil import sim_lib
  L = sim_lib()
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  out = str(tmp_path / 'f2.pt')
  mp.spawn(_factor_worker, args=(2, port, out), nprocs=2, join=True)
  r2 = torch.load(out)
  B1, x, eps, eps2, perm = _factor_data()
  fv = FactorVAE(discriminator_units=(16, 16), device='cpu', lib=L, **_factor_nets())
  p, dpar, _ = _factor_run(fv, x, eps, eps2, perm)
  d1, d2 = (p - r2['p']).abs(), (dpar - r2['d']).abs()
  assert d1.max().item() < 2e-4 and d1.mean().item() < 2e-6, (d1.max().item(), d1.mean().item())
  assert d2.max().item() < 2e-6, d2.max().item()   # Adam(1e-5): steps are 100x smaller


def _run_bench(*argv, env=None):
  import json
  import subprocess
  e = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
    e.pop(k, None)
  e.update(env or {})
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e,
                     capture_output=True, text=True, timeout=300)
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_flag_starts_that_many_ranks():
  """`python bench.py --gpus 2` (no torchrun around it) must start 2 ranks that find each other
  and report n_gpus=2: the launcher, the rendezvous and one collective, on CPU (gloo)."""
  rc, line, err = _run_bench('--gpus', '2', '--dry-run')
  assert rc == 0, err[-500:]
  assert line['n_gpus'] == 2 and line['rccl']['ranks_seen'] == 2


def test_bench_refuses_a_world_that_differs_from_gpus():
  rc, line, err = _run_bench('--gpus', '2', '--dry-run', env={'WORLD_SIZE': '1', 'RANK': '0'})
  assert rc != 0 and line is None and 'refusing' in err


def test_bench_launcher_propagates_a_failing_rank():
  """No GPU here: the real (non-dry) children must fail loudly (no CPU fallback), and the
  launcher's exit status must be non-zero with no JSON line printed."""
  if torch.cuda.is_available():
    pytest.skip('needs a machine without a GPU')
  rc, line, err = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0')
  assert rc != 0 and line is None
