"""Reader / writer for the reference's checkpoint FILES: TensorFlow tensor bundles as written by `tf.train.Saver().save(...)`
(reference vae/models.py:154,172-186 and ppo.py:184,202-216):

    <prefix>.index                 LevelDB-style table: key "" -> BundleHeaderProto, key <variable name> -> BundleEntryProto
    <prefix>.data-00000-of-00001   the raw little-endian tensors, back to back, in key order

so that weights trained with the reference load into this implementation (and a checkpoint written here is a valid bundle).
TensorFlow is not required: the two protobuf messages and the table format are small enough to handle directly.

  BundleHeaderProto { int32 num_shards = 1; Endianness endianness = 2; VersionDef version = 3 { int32 producer = 1; } }
  BundleEntryProto  { DataType dtype = 1; TensorShapeProto shape = 2 { repeated Dim dim = 2 { int64 size = 1; } }
                      int32 shard_id = 3; int64 offset = 4; int64 size = 5; fixed32 crc32c = 6; }
  table            = data blocks | metaindex block | index block | 48-byte footer (two block handles, padding, magic);
                     block = prefix-compressed entries, restart offsets, restart count; trailer = type byte (0) + masked crc32c.

Checksums are CRC-32C through the C ABI (`mi_crc32c`); tensors with a wrong checksum are rejected on read.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_DT_TO_NP = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 4: np.uint8, 6: np.int8, 5: np.int16}
_NP_TO_DT = {np.dtype(v): k for k, v in _DT_TO_NP.items()}
BLOCK_SIZE = 262144                                      # tensorflow/core/lib/io/table_options.h: Options::block_size default (the bundle writer keeps it)
RESTART_INTERVAL = 16


def _crc(data, crc=0):
    from . import lib as milib
    b = bytes(data) if not isinstance(data, (bytes, bytearray)) else data
    return int(milib.get().mi_crc32c(crc, b, len(b))) & 0xffffffff


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _unmask(m):
    r = (m - 0xa282ead8) & 0xffffffff
    return ((r >> 17) | (r << 15)) & 0xffffffff


# ---------------------------------------------------------------- protobuf (schema-less, only what the two messages need)
def _varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _enc_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, val


def _tag(fno, wt):
    return _enc_varint((fno << 3) | wt)


def _entry_proto(dtype, shape, offset, size, crc_masked):
    dims = b"".join(_tag(2, 2) + _enc_varint(len(d)) + d for d in (_tag(1, 0) + _enc_varint(int(n)) for n in shape))
    out = _tag(1, 0) + _enc_varint(dtype) + _tag(2, 2) + _enc_varint(len(dims)) + dims
    # shard_id 0 is the proto default (omitted, as TF does); offset 0 likewise
    if offset:
        out += _tag(4, 0) + _enc_varint(offset)
    out += _tag(5, 0) + _enc_varint(size) + _tag(6, 5) + struct.pack("<I", crc_masked)
    return out


def _header_proto(num_shards=1):
    version = _tag(1, 0) + _enc_varint(1)                       # VersionDef.producer = 1 (kTensorBundleVersion)
    return _tag(1, 0) + _enc_varint(num_shards) + _tag(3, 2) + _enc_varint(len(version)) + version   # endianness LITTLE = 0 (default)


def _parse_entry(val):
    e = {"dtype": None, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for fno, wt, v in _fields(val):
        if fno == 1:
            e["dtype"] = v
        elif fno == 2:
            for f2, _, dim in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, s in _fields(dim):
                        if f3 == 1:
                            size = s
                    e["shape"].append(size)
        elif fno == 3:
            e["shard_id"] = v
        elif fno == 4:
            e["offset"] = v
        elif fno == 5:
            e["size"] = v
        elif fno == 6:
            e["crc32c"] = struct.unpack("<I", v)[0]
        elif fno == 7:
            e["sliced"] = True
    return e


# ---------------------------------------------------------------- table (LevelDB format, uncompressed blocks)
def _read_block(data, offset, size, verify=True):
    blk = data[offset:offset + size]
    if len(blk) != size or len(data) < offset + size + 5:
        raise ValueError("truncated table block")
    if data[offset + size] != 0:
        raise ValueError("compressed table blocks are not supported (TF writes bundle indices uncompressed)")
    if verify:
        want = _unmask(struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0])
        if _crc(data[offset:offset + size + 1]) != want:
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


class _BlockBuilder:
    def __init__(self, restart_interval=RESTART_INTERVAL):
        self.buf, self.restarts, self.count, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, val):
        shared = 0
        if self.count % self.ri == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(val)) + key[shared:] + val
        self.last, self.count = key, self.count + 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _shortest_separator(start, limit):
    """LevelDB BytewiseComparator::FindShortestSeparator: a short key k with start <= k < limit (index-block keys)."""
    n = min(len(start), len(limit))
    i = 0
    while i < n and start[i] == limit[i]:
        i += 1
    if i < n and start[i] < 0xff and start[i] + 1 < limit[i]:
        return start[:i] + bytes([start[i] + 1])
    return start


def _short_successor(key):
    """LevelDB FindShortSuccessor: a short key >= key (index entry of the last data block)."""
    for i, b in enumerate(key):
        if b != 0xff:
            return key[:i] + bytes([b + 1])
    return key


def _write_table(path, items):
    """items: sorted [(key bytes, value bytes)].  Same block layout, separators and trailers as TensorFlow's table builder: rebuilding
    the reference's shipped .index files from their parsed entries reproduces them byte for byte (tests/test_host_logic.py)."""
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", _mask(_crc(block + b"\x00"))))
        return _enc_varint(off) + _enc_varint(len(block))

    index, bb = _BlockBuilder(1), _BlockBuilder()         # the index block restarts at every entry (LevelDB)
    pending = None                                        # (last key, handle) of a finished block: its index key needs the NEXT key
    for key, val in items:
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        bb.add(key, val)
        if bb.size() >= BLOCK_SIZE:                       # LevelDB flushes once the estimated size reaches block_size
            pending = (bb.last, emit(bb.finish()))
            bb = _BlockBuilder()
    if bb.count:
        pending = (bb.last, emit(bb.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta_handle = emit(_BlockBuilder().finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(out)
    os.replace(tmp, path)


def read_index(index_path, verify=True):
    """-> {variable name: entry dict} (+ key "" -> header fields) of a `<prefix>.index` file."""
    data = open(index_path, "rb").read()
    if len(data) < 48 or data[-8:] != struct.pack("<Q", TABLE_MAGIC):
        raise ValueError("%s is not a TensorFlow bundle index (bad table magic)" % index_path)
    footer = data[-48:]
    _, p = _varint(footer, 0)
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)
    isz, p = _varint(footer, p)
    entries, header = {}, None
    for _, handle in _read_block(data, ioff, isz, verify):
        boff, q = _varint(handle, 0)
        bsz, q = _varint(handle, q)
        for key, val in _read_block(data, boff, bsz, verify):
            if key == b"":
                header = {f: v for f, _, v in _fields(val)}
            else:
                entries[key.decode()] = _parse_entry(val)
    if header is None:
        raise ValueError("%s: no bundle header entry" % index_path)
    if header.get(2, 0) != 0:
        raise ValueError("%s: big-endian bundles are not supported" % index_path)
    return entries, int(header.get(1, 1))


def read_bundle(prefix, verify=True):
    """tf.train.load_checkpoint(prefix) equivalent: {variable name: numpy array} of every (non-string, unsliced) variable."""
    entries, num_shards = read_index(prefix + ".index", verify)
    shards = {}
    out = {}
    for name, e in entries.items():
        if e["sliced"] or e["dtype"] not in _DT_TO_NP:
            raise ValueError("variable %s: dtype %s / partitioned variables are not supported" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, num_shards)
            if not os.path.exists(path):
                raise FileNotFoundError("%s (the .index is present but its data shard is missing)" % path)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        dt = np.dtype(_DT_TO_NP[e["dtype"]])
        n = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if raw.size != e["size"] or n * dt.itemsize != e["size"]:
            raise ValueError("variable %s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
        buf = np.ascontiguousarray(raw)
        if verify and e["crc32c"] is not None and _crc(buf.tobytes()) != _unmask(e["crc32c"]):
            raise ValueError("variable %s: data checksum mismatch" % name)
        out[name] = buf.view(dt).reshape(e["shape"]).copy()
    return out


def write_bundle(prefix, variables):
    """tf.train.Saver-compatible files for {name: array}: `<prefix>.index` + `<prefix>.data-00000-of-00001`."""
    names = sorted(variables, key=lambda s: s.encode())
    items, offset = [(b"", _header_proto(1))], 0
    data_path = prefix + ".data-00000-of-00001"
    tmp = data_path + ".tmp"
    with open(tmp, "wb") as f:
        for name in names:
            a = np.asarray(variables[name])
            shape = a.shape                                  # (np.ascontiguousarray would turn a scalar into shape (1,))
            a = np.ascontiguousarray(a)
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            if a.dtype not in _NP_TO_DT:
                raise ValueError("variable %s: dtype %s has no TensorFlow bundle encoding here" % (name, a.dtype))
            raw = a.tobytes()
            f.write(raw)
            items.append((name.encode(), _entry_proto(_NP_TO_DT[a.dtype], shape, offset, len(raw), _mask(_crc(raw)))))
            offset += len(raw)
    os.replace(tmp, data_path)
    _write_table(prefix + ".index", items)
    return prefix
