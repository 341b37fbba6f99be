#!/usr/bin/env python3
"""One training step as a timeline from a rocprofv3 --kernel-trace CSV: start offset, duration,
stream/queue, kernel (short).  usage: tools/timeline.py <dir> [step_index [first_layer_launches_per_step]]
(FactorVAE runs the encoder twice per iteration -- both half batches -- so its iteration spans 2 first-layer launches)"""
import csv, glob, re, sys
root = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 200
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = []


def grid_of(r):
  """total work-items of the dispatch (x * y * z), as the --pmc tables name a grid"""
  if 'Grid_Size_X' in r:
    return str(int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1))
  return r.get('Grid_Size', '?')


for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r.get('Stream_Id', '?'),
                 re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']), grid_of(r)))
rows.sort()
# a step starts at its first kernel: the noise (round 2) or the first-layer convolution (the noise is drawn inside
# latent_block_fwd since round 3)
starts = [i for i, r in enumerate(rows) if r[4].startswith('rng_normal_kernel')]
if len(starts) < 10:
  starts = [i for i, r in enumerate(rows) if 'smallc_fwd' in r[4]]
which = min(which, (len(starts) - 1) // per - 1)
i0, i1 = starts[which * per], starts[(which + 1) * per]
t0 = rows[i0][0]
busy_end = t0
gaps = 0
print(f'{"start_us":>9s} {"dur_us":>8s} {"q":>3s} kernel')
for r in rows[i0:i1]:
  name = re.sub(r'^void ', '', r[4])
  name = re.sub(r'\(.*', '', name)[:90]
  ov = '' if r[0] >= busy_end else f'  (overlaps {min(busy_end, r[1]) - r[0]:.0f} ns)'
  if r[0] > busy_end:
    gaps += r[0] - busy_end
  busy_end = max(busy_end, r[1])
  print(f'{(r[0]-t0)/1e3:9.1f} {(r[1]-r[0])/1e3:8.1f} {r[2]:>3s} {name} g={r[5]}{ov}')
print(f'step span {(rows[i1][0]-t0)/1e3:.1f} us, idle gaps inside {gaps/1e3:.1f} us')
