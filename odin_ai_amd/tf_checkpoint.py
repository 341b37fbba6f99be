"""TensorFlow-format weight files without TensorFlow.

The reference saves with `keras.Model.save_weights(filepath, save_format='tf')`
(odin/networks/base_networks.py:373-390) and loads with `load_weights(filepath)` (:338-371):
a TensorBundle -- `<prefix>.index` (a LevelDB-style table: key -> BundleEntryProto) and
`<prefix>.data-00000-of-00001` (raw tensor bytes) -- plus the object graph that maps Keras
variables to keys.  This module reads and writes that container (formats: tensorflow==2.5.0
core/util/tensor_bundle, core/lib/io/table, core/protobuf/trackable_object_graph.proto --
third-party, restated from their published layouts), so that

  * weights trained by the reference on a machine that has TensorFlow can be evaluated by the
    HIP backend (`VariationalAutoencoder.load_weights` finds variables by their Keras names,
    `encoder0/kernel` ... `latents/bias`, inside the checkpoint's object graph), and
  * weights trained here can be read back with `tf.train.load_checkpoint(prefix)`.

Not verified against a TensorFlow installation in this repository's test environment (there is
none): the tests cover write -> read round trips, the table / record checksums, the published
known answer of CRC-32C, and a reader test on a fixture assembled INDEPENDENTLY of this module in the
shape Keras writes (tests/golden/gen_keras_ckpt.py: nested object graph `encoder/layer_with_weights-N/kernel`,
`save_counter`, optimizer hyper-parameters and slot variables, two data shards, multi-block index).  Loading through `keras.Model.load_weights` additionally needs the
reference model's exact Python attribute paths and is therefore left to `tf.train.load_checkpoint`
+ assignment on that side.
"""
from __future__ import annotations

import os
import struct
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_STRING, DT_INT64 = 1, 2, 3, 7, 9
_NP2DT = {np.dtype('float32'): DT_FLOAT, np.dtype('float64'): DT_DOUBLE, np.dtype('int32'): DT_INT32,
          np.dtype('int64'): DT_INT64}
_DT2NP = {v: k for k, v in _NP2DT.items()}
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8


# ---------------------------------------------------------------- checksums ----------
def _crc32c_py(crc: int, data: bytes) -> int:
  tab = _crc32c_py.tab
  if tab is None:
    tab = []
    for i in range(256):
      c = i
      for _ in range(8):
        c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
      tab.append(c)
    _crc32c_py.tab = tab
  c = crc ^ 0xFFFFFFFF
  for b in data:
    c = (c >> 8) ^ tab[(c ^ b) & 0xFF]
  return c ^ 0xFFFFFFFF


_crc32c_py.tab = None


def crc32c(data: bytes, crc: int = 0, lib=None) -> int:
  """CRC-32C; through libodin_hip.so's host routine when a library handle is given."""
  if lib is not None and len(data) > 64:
    import ctypes as C
    buf = (C.c_char * len(data)).from_buffer_copy(data)
    return int(lib.c.odin_crc32c(C.c_uint32(crc), buf, len(data)))
  return _crc32c_py(crc, data)


def mask(crc: int) -> int:
  return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask(m: int) -> int:
  rot = (m - _MASK_DELTA) & 0xFFFFFFFF
  return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------- protobuf wire ------
def _varint(n: int) -> bytes:
  n &= (1 << 64) - 1
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  shift = val = 0
  while True:
    b = buf[pos]
    pos += 1
    val |= (b & 0x7F) << shift
    if not b & 0x80:
      return val, pos
    shift += 7


def _field(num: int, wire: int, payload: bytes) -> bytes:
  return _varint((num << 3) | wire) + payload


def _ld(num: int, payload: bytes) -> bytes:  # length-delimited
  return _field(num, 2, _varint(len(payload)) + payload)


def _parse(buf: bytes) -> List[Tuple[int, int, object]]:
  """[(field number, wire type, value)]; value: int (varint / fixed) or bytes."""
  out, pos = [], 0
  while pos < len(buf):
    tag, pos = _read_varint(buf, pos)
    num, wire = tag >> 3, tag & 7
    if wire == 0:
      v, pos = _read_varint(buf, pos)
    elif wire == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]
      pos += 8
    elif wire == 2:
      n, pos = _read_varint(buf, pos)
      v = bytes(buf[pos:pos + n])
      pos += n
    elif wire == 5:
      v = struct.unpack_from('<I', buf, pos)[0]
      pos += 4
    else:
      raise ValueError(f'unsupported wire type {wire}')
    out.append((num, wire, v))
  return out


# ---------------------------------------------------------------- table (SSTable) ----
def _block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
  out, restarts, last = bytearray(), [], b''
  for i, (k, v) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(k), len(last)) and k[shared] == last[shared]:
        shared += 1
    out += _varint(shared) + _varint(len(k) - shared) + _varint(len(v)) + k[shared:] + v
    last = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _emit_block(f, contents: bytes, lib=None) -> Tuple[int, int]:
  off = f.tell()
  f.write(contents)
  trailer = b'\x00'  # no compression
  f.write(trailer + struct.pack('<I', mask(crc32c(contents + trailer, lib=lib))))
  return off, len(contents)


def write_table(path: str, items: List[Tuple[bytes, bytes]], block_bytes: int = 4096, lib=None):
  items = sorted(items)
  with open(path, 'wb') as f:
    index, cur, size = [], [], 0
    for k, v in items:
      cur.append((k, v))
      size += len(k) + len(v) + 8
      if size >= block_bytes:
        off, n = _emit_block(f, _block(cur), lib)
        index.append((cur[-1][0], _varint(off) + _varint(n)))
        cur, size = [], 0
    if cur or not index:
      off, n = _emit_block(f, _block(cur), lib)
      index.append((cur[-1][0] if cur else b'', _varint(off) + _varint(n)))
    moff, mn = _emit_block(f, _block([]), lib)
    ioff, inn = _emit_block(f, _block(index, restart_interval=1), lib)
    footer = _varint(moff) + _varint(mn) + _varint(ioff) + _varint(inn)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    f.write(footer)


def _read_block(buf: bytes, off: int, n: int, verify: bool = True, lib=None) -> List[Tuple[bytes, bytes]]:
  contents, trailer = buf[off:off + n], buf[off + n:off + n + 5]
  if trailer[0] != 0:
    kind = {1: 'snappy'}.get(trailer[0], f'type {trailer[0]}')
    raise ValueError(f'{kind}-compressed table block: only uncompressed index files are supported '
                     '(TensorFlow writes checkpoint indices uncompressed; re-save the checkpoint)')
  if verify and unmask(struct.unpack('<I', trailer[1:])[0]) != crc32c(contents + trailer[:1], lib=lib):
    raise ValueError('table block checksum mismatch')
  nrest = struct.unpack_from('<I', contents, len(contents) - 4)[0]
  end = len(contents) - 4 - 4 * nrest
  out, pos, last = [], 0, b''
  while pos < end:
    shared, pos = _read_varint(contents, pos)
    non_shared, pos = _read_varint(contents, pos)
    vlen, pos = _read_varint(contents, pos)
    k = last[:shared] + contents[pos:pos + non_shared]
    pos += non_shared
    out.append((bytes(k), bytes(contents[pos:pos + vlen])))
    pos += vlen
    last = k
  return out


def read_table(path: str, lib=None) -> List[Tuple[bytes, bytes]]:
  buf = open(path, 'rb').read()
  if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != TABLE_MAGIC:
    raise ValueError(f'{path}: not a TensorFlow table file')
  foot = buf[-48:]
  _, p = _read_varint(foot, 0)
  _, p = _read_varint(foot, p)
  ioff, p = _read_varint(foot, p)
  inn, p = _read_varint(foot, p)
  out = []
  for _, handle in _read_block(buf, ioff, inn, lib=lib):
    off, q = _read_varint(handle, 0)
    n, _ = _read_varint(handle, q)
    out += _read_block(buf, off, n, lib=lib)
  return out


# ---------------------------------------------------------------- TensorBundle -------
def _entry_proto(dtype: int, shape, offset: int, size: int, crc: int) -> bytes:
  dims = b''.join(_ld(2, _field(1, 0, _varint(int(d)))) for d in shape)
  out = _field(1, 0, _varint(dtype)) + _ld(2, dims)
  if offset:
    out += _field(4, 0, _varint(offset))
  out += _field(5, 0, _varint(size)) + _field(6, 5, struct.pack('<I', crc))
  return out


def _string_tensor_bytes(s: bytes, lib=None) -> Tuple[bytes, int]:
  """tensor_bundle.cc WriteStringTensor for ONE string: [varint len][masked crc of len][bytes]."""
  c = crc32c(struct.pack('<I', len(s)), lib=None)
  length_ck = struct.pack('<I', mask(c))
  c = crc32c(length_ck, c)
  c = crc32c(s, c, lib=lib)
  return _varint(len(s)) + length_ck + s, c


def _object_graph(names: List[str]) -> bytes:
  """TrackableObjectGraph: root node with one child per variable; each variable node carries the
  attribute VARIABLE_VALUE with its Keras full_name and its checkpoint key."""
  def ref(node_id, local):
    return _ld(1, _field(1, 0, _varint(node_id)) + _ld(2, local.encode()))
  root = b''.join(ref(i + 1, n) for i, n in enumerate(names))
  nodes = [_ld(1, root)]
  for n in names:
    attr = _ld(1, b'VARIABLE_VALUE') + _ld(2, n.encode()) + _ld(3, checkpoint_key(n).encode())
    nodes.append(_ld(1, _ld(2, attr)))
  return b''.join(nodes)


def checkpoint_key(full_name: str) -> str:
  """object-based key of a variable hung directly under the root as child `full_name` (local names are
  escaped as tf.train.Checkpoint does: '.' doubled first, then '/' -> '.S')."""
  return full_name.replace('.', '..').replace('/', '.S') + '/.ATTRIBUTES/VARIABLE_VALUE'


def save_checkpoint(prefix: str, variables: Dict[str, np.ndarray], lib=None):
  """Writes <prefix>.index and <prefix>.data-00000-of-00001.  `variables`: Keras variable name
  (`encoder0/kernel`, ..., `Step`) -> array."""
  names = sorted(variables)
  items, data, off = [], bytearray(), 0
  payload = {checkpoint_key(n): np.asarray(variables[n], order='C') for n in names}
  og, ogc = _string_tensor_bytes(_object_graph(names), lib)
  blobs = {OBJECT_GRAPH_KEY: (DT_STRING, (), og, ogc)}
  for k, a in payload.items():
    if a.dtype not in _NP2DT:
      raise TypeError(f'{k}: dtype {a.dtype} not supported')
    b = a.astype(a.dtype.newbyteorder('<')).tobytes()
    blobs[k] = (_NP2DT[a.dtype], a.shape, b, crc32c(b, lib=lib))
  for k in sorted(blobs):
    dt, shp, b, c = blobs[k]
    items.append((k.encode(), _entry_proto(dt, shp, off, len(b), mask(c))))
    data += b
    off += len(b)
  header = _field(1, 0, _varint(1)) + _ld(3, _field(1, 0, _varint(1)))  # num_shards=1, version.producer=1
  items.append((b'', header))
  d = os.path.dirname(prefix)
  if d:
    os.makedirs(d, exist_ok=True)
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(bytes(data))
  write_table(prefix + '.index', items, lib=lib)


def load_checkpoint(prefix: str, verify: bool = True, lib=None) -> Dict[str, np.ndarray]:
  """Reads a TensorBundle; returns Keras variable full_name -> array for every variable the
  object graph lists (falls back to the raw keys when there is no object graph)."""
  if not os.path.exists(prefix + '.index'):
    raise FileNotFoundError(prefix + '.index')
  entries = {}
  for k, v in read_table(prefix + '.index', lib=lib):
    if k == b'':
      continue
    e = dict(dtype=0, shape=[], shard=0, offset=0, size=0, crc=None)
    for num, wire, val in _parse(v):
      if num == 1:
        e['dtype'] = val
      elif num == 2:
        e['shape'] = [dict((n2, v2) for n2, _, v2 in _parse(d)).get(1, 0)
                      for n1, _, d in _parse(val) if n1 == 2]
      elif num == 3:
        e['shard'] = val
      elif num == 4:
        e['offset'] = val
      elif num == 5:
        e['size'] = val
      elif num == 6:
        e['crc'] = val
    entries[k.decode()] = e
  shards = {}

  def shard(i):
    if i not in shards:
      import glob
      cand = sorted(glob.glob(f'{prefix}.data-{i:05d}-of-*'))
      if not cand:
        raise FileNotFoundError(f'{prefix}.data-{i:05d}-of-*')
      shards[i] = open(cand[0], 'rb').read()
    return shards[i]

  def raw(e):
    return shard(e['shard'])[e['offset']:e['offset'] + e['size']]

  tensors = {}
  for k, e in entries.items():
    if e['dtype'] == DT_STRING:
      continue
    if e['dtype'] not in _DT2NP:
      continue
    b = raw(e)
    if verify and e['crc'] is not None and unmask(e['crc']) != crc32c(b, lib=lib):
      raise ValueError(f'{k}: tensor checksum mismatch')
    tensors[k] = np.frombuffer(b, dtype=_DT2NP[e['dtype']].newbyteorder('<')).reshape(e['shape']).copy()
  if OBJECT_GRAPH_KEY not in entries:
    return tensors
  b = raw(entries[OBJECT_GRAPH_KEY])
  n, pos = _read_varint(b, 0)
  graph = b[pos + 4:pos + 4 + n]
  out = {}
  for num, _, node in _parse(graph):
    if num != 1:
      continue
    for n2, _, attr in _parse(node):
      if n2 != 2:
        continue
      a = {n3: v3 for n3, _, v3 in _parse(attr)}
      if a.get(1) == b'VARIABLE_VALUE' and 3 in a and a[3].decode() in tensors:
        out[a.get(2, a[3]).decode()] = tensors[a[3].decode()]
  return out if out else tensors


# ---------------------------------------------------------------- event files --------
class ScalarEventWriter:
  """Minimal TensorBoard event-file writer for the scalars `Trainer` logs
  (odin/training/trainer.py:52-71: tf.summary.scalar(f'{prefix}loss' | metric names)):
  TFRecord framing [len u64][masked crc(len)][data][masked crc(data)] around Event protos."""

  def __init__(self, logdir: str, lib=None):
    os.makedirs(logdir, exist_ok=True)
    self.path = os.path.join(logdir, f'events.out.tfevents.{int(time.time())}.odin_ai_amd')
    self.lib = lib
    self.f = open(self.path, 'ab')
    self._record(_field(1, 1, struct.pack('<d', time.time())) + _ld(3, b'brain.Event:2'))

  def _record(self, data: bytes):
    hdr = struct.pack('<Q', len(data))
    self.f.write(hdr + struct.pack('<I', mask(crc32c(hdr))) + data +
                 struct.pack('<I', mask(crc32c(data, lib=self.lib))))

  def scalar(self, tag: str, value: float, step: int):
    val = _ld(1, tag.encode()) + _field(2, 5, struct.pack('<f', float(value)))
    ev = (_field(1, 1, struct.pack('<d', time.time())) + _field(2, 0, _varint(int(step))) +
          _ld(5, _ld(1, val)))
    self._record(ev)

  def flush(self):
    self.f.flush()

  def close(self):
    self.f.close()


def read_scalar_events(path: str) -> List[Tuple[int, str, float]]:
  """[(step, tag, value)] of a scalar event file (checksums verified)."""
  buf, pos, out = open(path, 'rb').read(), 0, []
  while pos < len(buf):
    n = struct.unpack_from('<Q', buf, pos)[0]
    if unmask(struct.unpack_from('<I', buf, pos + 8)[0]) != crc32c(buf[pos:pos + 8]):
      raise ValueError('event length checksum mismatch')
    data = buf[pos + 12:pos + 12 + n]
    if unmask(struct.unpack_from('<I', buf, pos + 12 + n)[0]) != crc32c(data):
      raise ValueError('event data checksum mismatch')
    pos += 16 + n
    ev = {num: v for num, _, v in _parse(data)}
    if 5 in ev:
      for n1, _, val in _parse(ev[5]):
        v = {n2: x for n2, _, x in _parse(val)}
        out.append((ev.get(2, 0), v[1].decode(), struct.unpack('<f', struct.pack('<I', v[2]))[0]))
  return out
