#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the separate rocprofv3 --pmc passes (tools/pmc.sh <tag>; usage:
tools/pmc_traffic.py <tag>, default r04).  Exits non-zero when a kernel it prices no longer exists in the
passes -- a kernel was renamed or rerouted and the table must be updated, not silently left stale.
HBM traffic per launch of the three largest kernels of the dsprites_betavae_b256 step, corrected as
MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts 128-byte reads at 64 bytes: doubled;
WRITE_SIZE exact), next to the algorithmic bytes of the launch."""
import json, re, sys
B = 256
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r06'


COUNTS = {}  # file -> {(kernel, grid): dispatches in that pass} (the passes run different numbers of steps)


def read(fn):
  out = {}
  cnt = COUNTS.setdefault(fn, {})
  for ln in open(fn):
    f = [x.strip() for x in ln.split('|')]
    if len(f) >= 4 and f[0] != 'kernel':
      out[(f[0].replace('void ', ''), f[1])] = float(f[3])
      cnt[(f[0].replace('void ', ''), f[1])] = int(f[2])
  return out


fetch, write = read(f'gpurun_out/{TAG}_pmc_FETCH_SIZE.txt'), read(f'gpurun_out/{TAG}_pmc_WRITE_SIZE.txt')
f4 = 4
kernels = {
    # bench op name: (kernel-name prefix, grid, description, algorithmic bytes)
    'dec4:deconv:bwd': ('bwd_planes_pc_kernel<32>', '196608',
                        'weight + data gradient of the last Conv2DTranspose in one launch (bwd_planes: dy fetched and split once), dSprites B=256',
                        # dY [B,64,64,32] + x [B,32,32,32] read (aux IS x in the step: one tensor), dx [B,32,32,32] written.
                        # (The 256 slab rows of the weight gradient -- 16.8 MB -- are an implementation artefact, not
                        # algorithmic traffic: priced out since round 6)
                        (B * 64 * 64 * 32 + 2 * B * 32 * 32 * 32) * f4),
    'dec4+5:tail:fwd+elbo': ('tconv_planes_kernel<3, 1, 32, 0, false, 2', '131072',
                             'fused decoder tail (tconv_planes, fp32 operands as 2 f16 planes)',
                             # x [B,32,32,32] + target [B,64,64,1] read; logits + d(pre-activation) [B,64,64,32] written
                             (B * 32 * 32 * 32 + 2 * B * 64 * 64 + B * 64 * 64 * 32) * f4),
    'neck:fwd': ('neck_fwd_kernel<8>', '131072',
                 'conv3 + projection + latent block + decoder projection + deconv1 in one launch (neck.hip)',
                 # x [B,8,8,64] read; y3 [B,4,4,64], y4 [B,128], p / z / y0, y1 [B,8,8,64] written; the weights once
                 (2 * B * 8 * 8 * 64 + B * 4 * 4 * 64 + B * (128 + 20 + 10 + 128) + 16 * 64 * 64 + 1024 * 128 +
                  128 * 20 + 10 * 128 + 16 * 64 * 8) * f4),
    'dec2:deconv:bwd': ('bwd_planes2_kernel<8, 2', '131072',
                        'weight + data gradient of decoder2 (64 -> 64 channels, 8x8 -> 16x16): both 32-channel passes in one launch (bwd_planes)',
                        # dY [B,16,16,64] + x [B,8,8,64] read (twice: once per pass; aux IS x); partial sums written and read back,
                        # dx written: [B,8,8,64] each (slab rows priced out, see above)
                        (B * 16 * 16 * 64 + 2 * B * 8 * 8 * 64) * f4),
    'dec3:deconv:fwd': ('tconv_planes2_kernel<1, 16, 2', '262144',
                        'decoder3 forward (64 -> 32 channels, 16x16 -> 32x32): both reduction passes in one launch, the first '
                        "pass's partial sums in LDS (round 6)",
                        # x [B,16,16,64] read, y [B,32,32,32] written
                        (B * 16 * 16 * 64 + B * 32 * 32 * 32) * f4),
}


def find(tab, prefix, grid):
  for (k, key), v in tab.items():
    if k.startswith(prefix) and key.split('/')[0] == grid:
      return k, v
  return None, None


def stats_us(prefix):
  """average in-graph duration (us) of the kernel from the rocprofv3 --stats table of the same run"""
  import csv, glob
  for fn in glob.glob(f'gpurun_out/{TAG}_final_prof_kernel_stats.csv') + glob.glob(f'profiles/{TAG}_kernel_stats.csv'):
    for r in csv.DictReader(open(fn)):
      nm = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '')
      if nm.startswith(prefix):
        return round(float(r['AverageNs']) / 1e3, 2), fn
  return None, None


res = {}
missing = 0
for op, (k, key, desc, alg) in kernels.items():
  kn, fk = find(fetch, k, key)
  _, wk = find(write, k, key)
  if fk is None or wk is None:
    print('missing', op, k, key, file=sys.stderr)
    missing += 1
    continue
  k = kn
  res[op] = dict(kernel=f'{k} ({desc})', FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk,
                 traffic_bytes=int((2 * fk + wk) * 1024), algorithmic_bytes=int(alg),
                 source=f'profiles/{TAG}_pmc_FETCH_SIZE.txt + {TAG}_pmc_WRITE_SIZE.txt (separate --pmc passes of '
                        '`bench.py --steps 20 --warmup 5`); traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 '
                        'reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM section)')
  us, src = stats_us(k.split('(')[0] if '(' in k else k)
  if us is not None:
    res[op]['us_in_graph'] = us
    res[op]['us_in_graph_source'] = f'profiles/{TAG}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py: average over the replayed graphs)'
# the whole step: every kernel of the pass that ran once (or k times) per step -- the first-layer convolution runs
# exactly once per step and gives the step count of the pass; stand-alone probes and torch kernels have other counts
def step_launches(fn):
  """[(kernel name, grid)] of one step from the rocprofv3 timeline tools/profile.sh wrote (tools/timeline.py)"""
  out = []
  for ln in open(fn):
    m = re.match(r'\s*[\d.]+\s+[\d.]+\s+\S+\s+(.*) g=(\d+)', ln)
    if m and not m.group(1).startswith('__amd_rocclr'):
      out.append((m.group(1).strip(), m.group(2)))
  return out


def lookup(tab, name, grid):
  for (k, key), v in tab.items():
    if key.split('/')[0] == grid and k[:60] == name[:60]:
      return v
  # (timelines written before round 5 name the x extent of the grid only)
  hits = [v for (k, key), v in tab.items() if k[:60] == name[:60] and int(key.split('/')[0]) % int(grid) == 0]
  return hits[0] if len(hits) == 1 else None


TL = sys.argv[2] if len(sys.argv) > 2 else f'gpurun_out/{TAG}_final_prof_timeline.txt'
try:
  launches = step_launches(TL)
except OSError:
  launches = []
n_step = len(launches)
if n_step:
  tot_f = tot_w = 0.0
  per = []
  for name, grid in launches:
    fk, wk = lookup(fetch, name, grid), lookup(write, name, grid)
    if fk is None or wk is None:
      print('step: no counters for', name, grid, file=sys.stderr)
      missing += 1
      continue
    tot_f += fk
    tot_w += wk
    per.append((int((2 * fk + wk) * 1024), 1, name[:60], grid))
  per.sort(reverse=True)
  res['step'] = dict(kernel='every launch of one dsprites_betavae_b256 step',
                     FETCH_SIZE_KB=round(tot_f, 1), WRITE_SIZE_KB=round(tot_w, 1),
                     traffic_bytes=int((2 * tot_f + tot_w) * 1024), launches=sum(m for _, m, _, _ in per),
                     largest=[dict(bytes=b, launches=m, kernel=k, grid=g) for b, m, k, g in per[:8]],
                     source=f'sum of the mean per-launch counters (profiles/{TAG}_pmc_FETCH_SIZE.txt / _WRITE_SIZE.txt) over the launches of '
                            f'one step as the rocprofv3 kernel trace lists them (profiles/{TAG}_step_timeline.txt)')
else:
  print('no step timeline:', TL, file=sys.stderr)
  missing += 1
json.dump(res, open(f'profiles/{TAG}_pmc_traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
sys.exit(1 if missing else 0)
