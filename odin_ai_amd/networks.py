"""Per-dataset encoder / decoder stacks, mirroring ``odin.networks.get_networks``.

The reference builds Keras layer objects (odin/networks/image_networks.py:907-933); here
a network is a plain description (`SequentialNetwork` holding layer tuples) that the HIP
engine compiles into kernel launches.  Same dataset names, same `zdim` / `qz` / semi-
supervised / hierarchical arguments, same returned dictionary keys
(``encoder, decoder, observation, latents``).

Layer tuples: ('center',) | ('conv', filters, k, stride, act) | ('deconv', ...) |
('flatten',) | ('dense', units, act) | ('reshape', (h, w, c)).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple


@dataclass
class SequentialNetwork:
  """Stand-in for odin.networks.SequentialNetwork (base_networks.py:883-959)."""
  layers: List[tuple]
  name: str = 'Sequential'
  input_shape: Optional[Tuple[int, ...]] = None
  # Keras layer names of the trainable layers, in order (image_networks.py:248-268,463-511,
  # 678-703): they are what a TF checkpoint of the reference model calls its variables
  layer_names: Optional[List[str]] = None

  def __iter__(self):
    return iter(self.layers)

  def __len__(self):
    return len(self.layers)


def layer_names(net: 'SequentialNetwork', prefix: str) -> Dict[int, str]:
  """{index into net.layers: Keras layer name} for the trainable layers."""
  idx = [i for i, L in enumerate(net.layers) if L[0] in ('conv', 'deconv', 'dense')]
  if net.layer_names is not None and len(net.layer_names) == len(idx):
    return dict(zip(idx, net.layer_names))
  return {i: f'{prefix}{j}' for j, i in enumerate(idx)}


@dataclass
class RVconf:
  """The subset of odin.bay.random_variable.RVconf (random_variable.py:175) the VAE path
  needs: event shape, posterior alias, whether a Dense projection is added, name."""
  event_shape: Tuple[int, ...]
  posterior: str = 'mvndiag'
  projection: bool = True
  name: str = 'latents'
  kwargs: dict = field(default_factory=dict)

  @property
  def event_size(self) -> int:
    n = 1
    for i in self.event_shape:
      n *= int(i)
    return n


_OBS_PARAMS = {'bernoulli': 1, 'gaussian': 2, 'gaus': 2, 'normal': 2, 'gaussian_softplus1': 2,
               'qlogistic': 2, 'mixqlogistic': None}


def mixqlogistic_params_size(n_components: int, n_channels: int) -> int:
  """MixtureQuantizedLogistic.params_size (quantized.py:288-292)."""
  return int((n_channels * 2 + n_channels * (n_channels - 1) // 2 + 1) * n_components)


def _observation(input_shape, distribution: str, n_components: int = 10) -> Tuple[int, RVconf]:
  """_parse_distribution (image_networks.py:46-102): number of maps the decoder's last 1x1 conv
  emits IN TOTAL and the observation description (projection=False: the logits / (loc, scale) /
  mixture parameters are the decoder output itself).

  'mixqlogistic': the reference sizes the head as n_channels * (params_size // n_channels)
  (:73-75,:261 `filters=n_channels * n_params`), which equals params_size for 1-channel images
  (30 maps) but gives 99 maps for RGB where MixtureQuantizedLogistic asserts 100 (quantized.py:262-265)
  -- the RGB model cannot be built as shipped; the head here emits params_size maps, the intended
  model."""
  distribution = str(distribution).lower()
  if distribution not in _OBS_PARAMS:
    raise ValueError(
        f"observation {distribution!r} is outside this build's scope "
        f"(supported: bernoulli, gaussian, gaussian_softplus1, qlogistic, mixqlogistic)")
  name = {'gaus': 'gaussian', 'normal': 'gaussian'}.get(distribution, distribution)
  C = int(input_shape[-1])
  if distribution == 'mixqlogistic':
    if C not in (1, 3) or int(n_components) != 10:
      raise ValueError('mixqlogistic: 1 or 3 channels, 10 components (the HIP kernel instances)')
    return mixqlogistic_params_size(n_components, C), RVconf(
        tuple(input_shape), name, projection=False, name='image', kwargs=dict(n_components=int(n_components)))
  return C * _OBS_PARAMS[distribution], RVconf(tuple(input_shape), name, projection=False, name='image')


def dsprites_networks(qz='mvndiag', zdim=None, activation='elu', is_semi_supervised=False,
                      is_hierarchical=False, centerize_image=True, n_channels=1, proj_dim=None,
                      distribution='bernoulli', **kwargs) -> Dict[str, object]:
  """image_networks.py:436-534 (dSprites; Shapes3D via n_channels=3, :560-597)."""
  if is_hierarchical or is_semi_supervised:
    raise NotImplementedError('hierarchical / semi-supervised stacks are out of scope (SURVEY 8)')
  if zdim is None:
    zdim = 10
  if proj_dim is None:
    proj_dim = 128 if n_channels == 1 else 256
  input_shape = (64, 64, int(n_channels))
  n_params, observation = _observation(input_shape, distribution, kwargs.get('n_components', 10))
  a = activation
  enc = ([('center',)] if centerize_image else []) + [
      ('conv', 32, 4, 2, a), ('conv', 32, 4, 2, a), ('conv', 64, 4, 2, a), ('conv', 64, 4, 2, a),
      ('flatten',), ('dense', proj_dim, 'linear')]
  dec = [('dense', proj_dim, 'linear'), ('reshape', (4, 4, proj_dim // 16)),
         ('deconv', 64, 4, 2, a), ('deconv', 64, 4, 2, a), ('deconv', 32, 4, 2, a),
         ('deconv', 32, 4, 2, a), ('conv', n_params, 1, 1, 'linear')]
  enc_names = ['encoder0', 'encoder1', 'encoder2', 'encoder3', 'encoder_proj']
  dec_names = ['decoder_proj', 'decoder1', 'decoder2', 'decoder3', 'decoder4', 'decoder6']
  return dict(encoder=SequentialNetwork(enc, 'Encoder', input_shape, enc_names),
              decoder=SequentialNetwork(dec, 'Decoder', (zdim,), dec_names),
              observation=observation,
              latents=RVconf((zdim,), qz, projection=True, name='latents'))


def shapes3d_networks(qz='mvndiag', zdim=None, **kwargs):
  """image_networks.py:560-597: dsprites stack with 3 channels, zdim 6, proj 256."""
  if zdim is None:
    zdim = 6
  kwargs.setdefault('n_channels', 3)
  return dsprites_networks(qz=qz, zdim=zdim, **kwargs)


def celeba_networks(qz='mvndiag', zdim=None, activation='elu', is_semi_supervised=False,
                    is_hierarchical=False, centerize_image=True, distribution='bernoulli',
                    n_channels=3, **kwargs):
  """image_networks.py:661-725.  The reference means `distribution='qlogistic'` here (decoder5
  emits 2*C maps, :697; `_parse_distribution(input_shape, 'qlogistic')`, :714) but stores the
  3-tuple that call returns as the observation, so the model cannot be built as shipped (SURVEY
  a12).  `distribution='qlogistic'` gives the intended model; Bernoulli (3 maps) and Gaussian
  (6 maps) heads are offered too."""
  if is_hierarchical or is_semi_supervised:
    raise NotImplementedError('hierarchical / semi-supervised stacks are out of scope (SURVEY 8)')
  if zdim is None:
    zdim = 45
  input_shape = (64, 64, n_channels)
  n_params, observation = _observation(input_shape, distribution, kwargs.get('n_components', 10))
  a = activation
  enc = ([('center',)] if centerize_image else []) + [
      ('conv', 32, 4, 2, a), ('conv', 32, 4, 2, a), ('conv', 64, 4, 2, a), ('conv', 64, 4, 1, a),
      ('flatten',), ('dense', 512, 'linear')]
  dec = [('dense', 512, 'linear'), ('reshape', (8, 8, 8)), ('deconv', 64, 4, 1, a),
         ('deconv', 64, 4, 2, a), ('deconv', 32, 4, 2, a), ('deconv', 32, 4, 2, a),
         ('conv', n_params, 1, 1, 'linear')]
  enc_names = ['encoder0', 'encoder1', 'encoder2', 'encoder3', 'encoder_proj']
  dec_names = ['decoder_proj', 'decoder1', 'decoder2', 'decoder3', 'decoder4', 'decoder5']
  return dict(encoder=SequentialNetwork(enc, 'Encoder', input_shape, enc_names),
              decoder=SequentialNetwork(dec, 'Decoder', (zdim,), dec_names),
              observation=observation,
              latents=RVconf((zdim,), qz, projection=True, name='latents'))


def mnist_networks(qz='mvndiag', zdim=None, activation='elu', is_semi_supervised=False,
                   is_hierarchical=False, centerize_image=True, distribution='bernoulli',
                   n_channels=1, **kwargs):
  """image_networks.py:223-292."""
  if is_hierarchical or is_semi_supervised:
    raise NotImplementedError('hierarchical / semi-supervised stacks are out of scope (SURVEY 8)')
  if zdim is None:
    zdim = 32
  input_shape = (28, 28, n_channels)
  n_params, observation = _observation(input_shape, distribution, kwargs.get('n_components', 10))
  a = activation
  enc = ([('center',)] if centerize_image else []) + [
      ('conv', 32, 5, 1, a), ('conv', 32, 5, 2, a), ('conv', 64, 5, 1, a), ('conv', 64, 5, 2, a),
      ('flatten',), ('dense', 196, 'linear')]
  dec = [('dense', 196, 'linear'), ('reshape', (7, 7, 4)), ('deconv', 64, 5, 2, a),
         ('conv', 64, 5, 1, a), ('deconv', 32, 5, 2, a), ('conv', 32, 5, 1, a),
         ('conv', n_params, 1, 1, 'linear')]
  enc_names = ['encoder0', 'encoder1', 'encoder2', 'encoder3', 'encoder_proj']
  dec_names = ['decoder_proj', 'decoder2', 'decoder3', 'decoder4', 'decoder5', 'decoder6']
  return dict(encoder=SequentialNetwork(enc, 'Encoder', input_shape, enc_names),
              decoder=SequentialNetwork(dec, 'Decoder', (zdim,), dec_names),
              observation=observation,
              latents=RVconf((zdim,), qz, projection=True, name='latents'))


def dense_networks(input_shape=(28, 28, 1), zdim=16, units=(512, 512), activation='relu',
                   distribution='bernoulli', **kwargs):
  """VariationalAutoencoder defaults (variational_autoencoder.py:181-185): NetConf((512,512),
  flatten_inputs=True) encoder/decoder (dense_network, base_networks.py:965-1022),
  RVconf(16,'mvndiag',projection=True) latents, RVconf(shape,'bernoulli',projection=True)
  observation (its Dense projection is the decoder's last layer here)."""
  if zdim is None:
    zdim = 16
  n = 1
  for i in input_shape:
    n *= i
  n_params, observation = _observation(input_shape, distribution, kwargs.get('n_components', 10))
  observation.projection = True
  enc = [('flatten',)] + [('dense', u, activation) for u in units]
  dec = [('dense', u, activation) for u in units] + [
      ('dense', (n // input_shape[-1]) * n_params, 'linear'),
      ('reshape', tuple(input_shape[:-1]) + (n_params,))]
  return dict(encoder=SequentialNetwork(enc, 'Encoder', tuple(input_shape)),
              decoder=SequentialNetwork(dec, 'Decoder', (zdim,)), observation=observation,
              latents=RVconf((zdim,), 'mvndiag', projection=True, name='latents'))


def speech_networks(qz='mvndiag', zdim=None, activation='elu', n_frames: int = 96, n_mels: int = 80,
                    distribution='gaus', proj_dim: Optional[int] = None, **kwargs):
  """Speech VAE of BASELINE config 5: the log-mel spectrogram [T, n_mels, 1] produced by the
  odin.preprocessing front-end (`odin_ai_amd.mel.MelsSpecExtractor`) through the image conv
  stack of `dsprites_networks` (image_networks.py:460-520: 4 x conv k4 s2 32/32/64/64 -> Dense;
  Dense -> 4 x deconv k4 s2 64/64/32/32 -> conv 1x1), with the observation of the reference's
  audio example: RVconf(event_shape, 'gaus', projection=False) = GaussianLayer with a
  softplus1 scale (examples/vae/vae_audio.py:84-90; layers/continuous.py:196-260), zdim 32
  (:44).  T and n_mels must be multiples of 16 (four stride-2 stages)."""
  if zdim is None:
    zdim = 32
  if n_frames % 16 or n_mels % 16:
    raise ValueError(f'n_frames={n_frames}, n_mels={n_mels}: both must be multiples of 16')
  h, w = n_frames // 16, n_mels // 16
  if proj_dim is None:
    proj_dim = 8 * h * w
  if proj_dim % (h * w):
    raise ValueError(f'proj_dim={proj_dim} must be a multiple of {h * w}')
  input_shape = (int(n_frames), int(n_mels), 1)
  name = {'gaus': 'gaussian_softplus1', 'gaussian_softplus1': 'gaussian_softplus1',
          'gaussian': 'gaussian', 'normal': 'gaussian'}.get(str(distribution).lower())
  if name is None:
    raise ValueError(f'speech_networks: observation {distribution!r} (gaus | gaussian)')
  observation = RVconf(input_shape, name, projection=False, name='Spectrogram')
  a = activation
  enc = [('conv', 32, 4, 2, a), ('conv', 32, 4, 2, a), ('conv', 64, 4, 2, a), ('conv', 64, 4, 2, a),
         ('flatten',), ('dense', proj_dim, 'linear')]
  dec = [('dense', proj_dim, 'linear'), ('reshape', (h, w, proj_dim // (h * w))),
         ('deconv', 64, 4, 2, a), ('deconv', 64, 4, 2, a), ('deconv', 32, 4, 2, a),
         ('deconv', 32, 4, 2, a), ('conv', 2, 1, 1, 'linear')]
  enc_names = ['encoder0', 'encoder1', 'encoder2', 'encoder3', 'encoder_proj']
  dec_names = ['decoder_proj', 'decoder1', 'decoder2', 'decoder3', 'decoder4', 'decoder6']
  return dict(encoder=SequentialNetwork(enc, 'Encoder', input_shape, enc_names),
              decoder=SequentialNetwork(dec, 'Decoder', (zdim,), dec_names),
              observation=observation,
              latents=RVconf((zdim,), qz, projection=True, name='Latents'))


_DATASETS = {
    'mnist': mnist_networks, 'binarizedmnist': mnist_networks, 'fashionmnist': mnist_networks,
    'dsprites': dsprites_networks, 'dspritesc': dsprites_networks,
    'shapes3d': shapes3d_networks, 'shapes3dsmall': shapes3d_networks,
    'celeba': celeba_networks, 'celebasmall': celeba_networks,
    'dense': dense_networks,
    'fsdd': speech_networks, 'speech': speech_networks,
}


def get_networks(dataset_name: str, *, is_semi_supervised: bool = False,
                 is_hierarchical: bool = False, qz: str = 'mvndiag', zdim: Optional[int] = None,
                 **kwargs) -> Dict[str, object]:
  """odin.networks.get_networks (image_networks.py:907-933)."""
  name = str(dataset_name).lower().strip().replace('_', '')
  if zdim is not None and zdim <= 0:
    zdim = None
  if name not in _DATASETS:
    raise ValueError(f"Cannot find pre-implemented network for dataset with name='{dataset_name}'")
  return _DATASETS[name](qz=qz, zdim=zdim, is_semi_supervised=is_semi_supervised,
                         is_hierarchical=is_hierarchical, **kwargs)
