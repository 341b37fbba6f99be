#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs: mean counter value per (kernel, grid) dispatch."""
import csv, glob, sys, collections, re
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name)[:70]
    key = (name, r.get('Grid_Size', '?') + '/lds' + r.get('LDS_Block_Size', '?'))
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in acc.values() for c in v})
print('kernel | grid | n | ' + ' | '.join(names))
for k, v in sorted(acc.items(), key=lambda kv: -max(len(x) for x in kv[1].values())):
  n = max(len(x) for x in v.values())
  print(f'{k[0]} | {k[1]} | {n} | ' + ' | '.join(f'{sum(v[c])/len(v[c]):.4g}' if c in v else '-' for c in names))
