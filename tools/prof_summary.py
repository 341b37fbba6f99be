#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid, LDS) count / avg / total us."""
import csv, glob, sys, collections, re
root = sys.argv[1]
files = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
rows = collections.defaultdict(list)
for f in files:
  for r in csv.DictReader(open(f)):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name)
    key = (name, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('LDS_Block_Size', '?'),
           r.get('VGPR_Count', '?'), r.get('Accum_VGPR_Count', '?'), r.get('SGPR_Count', '?'), r.get('Scratch_Size', '?'))
    rows[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print(f'{"kernel":100s} {"grid":>8s} {"lds":>7s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"scr":>5s} {"n":>6s} {"avg_us":>9s} {"total_us":>10s} {"%":>5s}')
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
  print(f'{k[0][:100]:100s} {k[1]:>8s} {k[2]:>7s} {k[3]:>5s} {k[4]:>5s} {k[5]:>5s} {k[6]:>5s} {len(v):6d} {sum(v)/len(v):9.2f} {sum(v):10.1f} {100*sum(v)/tot:5.1f}')
