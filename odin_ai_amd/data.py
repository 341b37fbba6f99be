"""On-device input pipeline: the uint8 image dataset lives in HBM and a
batch is gathered, normalised and written straight into the tensor the training-step graph
reads -- the host (the reference's tf.data map/shuffle/batch, fuel/image_data/_base.py:338-395)
is out of the loop.  Normalisation follows `ImageDataset.normalize` (_base.py:130-147)."""
from typing import Iterator, Optional

import ctypes as C
import numpy as np
import torch

from . import _lib

_MODES = {'probs': 0, 'tanh': 1, 'raster': 2, 'binarized': 3}


class DeviceImageDataset:
  """images: uint8 [N, H, W, C] (numpy or torch).  `premul`: multiplier applied before the
  normalisation (dSprites stores 0/1 pixels and uses 255, fuel/image_data/shapes.py:69-80).

  Iterating yields float32 [B, H, W, C] batches (drop_remainder, reshuffled every epoch with an
  on-device permutation).  With `out=` (e.g. `VAEEngine.input_buffer()`) every batch is written
  into that tensor and the same tensor is yielded, so `fit()` needs no per-step copy."""

  def __init__(self, images, batch_size: int, normalize: str = 'probs', premul: float = 1.0,
               shuffle: bool = True, seed: int = 1, device='cuda:0', lib=None,
               out: Optional[torch.Tensor] = None):
    if normalize not in _MODES:
      raise ValueError(f'normalize must be one of {sorted(_MODES)}')
    self.lib = lib if lib is not None else _lib.load()
    self.device = torch.device(device)
    data = torch.as_tensor(np.ascontiguousarray(images) if isinstance(images, np.ndarray) else images)
    if data.dtype != torch.uint8 or data.dim() != 4:
      raise ValueError('images must be uint8 [N, H, W, C]')
    self.data = data.contiguous().to(self.device)
    self.shape = tuple(self.data.shape[1:])
    self.n_per = int(np.prod(self.shape))
    if self.n_per % 16 != 0:
      raise ValueError('pixels per image must be a multiple of 16')
    self.N, self.B = int(self.data.shape[0]), int(batch_size)
    if self.B > self.N:
      raise ValueError('batch_size larger than the dataset')
    self.mode, self.premul = _MODES[normalize], float(premul)
    self.shuffle, self.seed, self.epoch = bool(shuffle), int(seed), 0
    self.out = out
    if out is not None and (tuple(out.shape) != (self.B,) + self.shape or out.dtype != torch.float32):
      raise ValueError('out must be float32 [batch_size, H, W, C]')

  def __len__(self) -> int:
    return self.N // self.B

  def _stream(self):
    if self.device.type != 'cuda':
      return None
    return torch.cuda.current_stream(self.device).cuda_stream

  def gather(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b] = normalize(images[idx[b]]); idx int32 [B] on the device."""
    idx = idx.to(device=self.device, dtype=torch.int32).contiguous()
    B = idx.numel()
    if out is None:
      out = torch.empty((B,) + self.shape, dtype=torch.float32, device=self.device)
    self.lib.odin_gather_normalize_u8(self.data.data_ptr(), idx.data_ptr(), out.data_ptr(), B,
                                      self.n_per, self.premul, self.mode, self._stream())
    return out

  def __iter__(self) -> Iterator[torch.Tensor]:
    g = torch.Generator(device=self.device).manual_seed(self.seed + self.epoch)
    self.epoch += 1
    order = (torch.randperm(self.N, generator=g, device=self.device) if self.shuffle
             else torch.arange(self.N, device=self.device)).to(torch.int32)
    for i in range(0, self.N - self.B + 1, self.B):
      yield self.gather(order[i:i + self.B], self.out)
