#!/usr/bin/env python3
"""phase times of the speech front-end launch (mel.hip) from the wall-clock stamps of workgroup (0, 0)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from odin_ai_amd.mel import MelsSpecExtractor
dev = torch.device('cuda:0')
ex = MelsSpecExtractor(device=dev, unit_range=True)
y = torch.randn(256, 8000, device=dev) * 0.1
buf = torch.zeros(64, dtype=torch.int64, device=dev)
for it in range(4):
  if it == 2: ex.lib.odin_debug_set_mel_stamps(buf.data_ptr())
  buf.zero_()
  out = ex(y)
  torch.cuda.synchronize()
s = [v for v in buf.cpu().tolist() if v]
print('stamps:', len(s))
d = [(s[i + 1] - s[i]) / 100.0 for i in range(len(s) - 1)]
print(' '.join(f'{v:.2f}' for v in d), 'total', (s[-1] - s[0]) / 100.0)
ex.lib.odin_debug_set_mel_stamps(None)
