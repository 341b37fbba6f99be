"""TEST INFRASTRUCTURE ONLY (see oracle/vae_oracle.py header): numpy restatement of the
reference's image normalisation, `ImageDataset.normalize` / `normalize255`
(odin/fuel/image_data/_base.py:130-147) and the dSprites pre-scaling by 255
(odin/fuel/image_data/shapes.py:69-72,80).  Parity unpinned against TensorFlow (the reference
function is a tf.Tensor op; the arithmetic is one division and two clips, done here in float32
exactly as tf.clip_by_value / tf.divide do on float32 tensors)."""
import numpy as np

MODES = {'probs': 0, 'tanh': 1, 'raster': 2, 'binarized': 3}


def normalize(images_u8: np.ndarray, normalize: str = 'probs', premul: float = 1.0) -> np.ndarray:
  """images_u8 [..., H, W, C] uint8 -> float32 (_base.py:133-147)."""
  x = images_u8.astype(np.float32) * np.float32(premul)
  if normalize == 'binarized':
    return x
  x = np.clip(x, np.float32(0.0), np.float32(255.0))
  if normalize == 'probs':
    return np.clip(x / np.float32(255.0), np.float32(1e-6), np.float32(1.0) - np.float32(1e-6))
  if normalize == 'tanh':
    return np.clip(x / np.float32(255.0) * np.float32(2.0) - np.float32(1.0),
                   np.float32(-1.0) + np.float32(1e-6), np.float32(1.0) - np.float32(1e-6))
  if normalize == 'raster':
    return x
  raise ValueError(normalize)


def gather_normalize(images_u8: np.ndarray, idx: np.ndarray, normalize_mode: str = 'probs',
                     premul: float = 1.0) -> np.ndarray:
  return normalize(images_u8[idx], normalize_mode, premul)
