"""TensorBoard event files, written the way tf.summary.FileWriter writes them (reference vae/models.py:145-151,169-170 and
ppo.py:150-181,262-273): `<log_dir>/events.out.tfevents.<unix time>.<host>` = TFRecord-framed Event protos

    record  = uint64 length | uint32 masked_crc32c(length bytes) | payload | uint32 masked_crc32c(payload)
    Event   { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    Summary { repeated Value value = 1 { string tag = 1; float simple_value = 2; SummaryMetadata metadata = 9; TensorProto tensor = 8; } }

so `tensorboard --logdir <model_dir>/logs` shows the same scalar charts as for the reference.  TensorFlow is not needed (the two messages
are encoded directly; CRC-32C through the C ABI's mi_crc32c).  Never on a timed path.  `read_events` parses such a file back and checks
both checksums of every record: it reads the reference's own event files (tests/golden/ref_events_head.bin) and ours.
"""
import os
import socket
import struct
import time


def _crc_masked(data):
    from . import lib as milib
    crc = int(milib.get().mi_crc32c(0, bytes(data), len(data))) & 0xffffffff
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _varint(v):
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


def _ld(fno, payload):                                       # length-delimited field
    return _varint((fno << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time, step=None, file_version=None, summary=None):
    out = _varint((1 << 3) | 1) + struct.pack("<d", wall_time)
    if step is not None:
        out += _varint((2 << 3) | 0) + _varint(int(step))
    if file_version is not None:
        out += _ld(3, file_version.encode())
    if summary is not None:
        out += _ld(5, summary)
    return out


class SummaryWriter:
    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, "ab")
        self._record(_event(time.time(), file_version="brain.Event:2"))

    def _record(self, payload):
        hdr = struct.pack("<Q", len(payload))
        self._f.write(hdr + struct.pack("<I", _crc_masked(hdr)) + payload + struct.pack("<I", _crc_masked(payload)))

    def add_scalar(self, tag, value, step):
        val = _ld(1, str(tag).encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
        self._record(_event(time.time(), step=step, summary=_ld(1, val)))

    def add_text(self, tag, text, step=0):
        """tf.summary.text: a DT_STRING tensor value carrying the "text" plugin tag (TensorBoard's Text dashboard; markdown)."""
        if isinstance(text, dict):
            text = "\n".join("|%s|%s|" % (k, v) for k, v in [("key", "value"), ("---", "---")] + list(text.items()))
        meta = _ld(1, _ld(1, b"text"))                                   # SummaryMetadata.plugin_data.plugin_name
        shape = _ld(2, _varint((1 << 3) | 0) + _varint(1))               # TensorShapeProto.dim { size: 1 }
        tensor = _varint((1 << 3) | 0) + _varint(7) + _ld(2, shape) + _ld(8, str(text).encode())   # dtype DT_STRING, string_val
        val = _ld(1, str(tag).encode()) + _ld(9, meta) + _ld(8, tensor)
        self._record(_event(time.time(), step=step, summary=_ld(1, val)))

    def flush(self):
        self._f.flush()

    def close(self):
        try:
            self._f.close()
        except Exception:
            pass


# ---------------------------------------------------------------- reader (scalars and the file_version record)
def _read_varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fno, wt, val


def read_events(path, verify=True):
    """-> (file_version or None, {tag: [(step, wall_time, value), ...]}) of one events.out.tfevents file; checks both CRCs of every record."""
    data = open(path, "rb").read()
    pos, series, version = 0, {}, None
    while pos + 12 <= len(data):
        hdr = data[pos:pos + 8]
        ln = struct.unpack("<Q", hdr)[0]
        payload = data[pos + 12:pos + 12 + ln]
        if len(payload) != ln or pos + 16 + ln > len(data):
            break                                                        # truncated tail (a writer that is still running)
        if verify:
            if struct.unpack("<I", data[pos + 8:pos + 12])[0] != _crc_masked(hdr) or struct.unpack("<I", data[pos + 12 + ln:pos + 16 + ln])[0] != _crc_masked(payload):
                raise ValueError("%s: record checksum mismatch at byte %d" % (path, pos))
        pos += 16 + ln
        step, wall = 0, None
        for fno, wt, v in _fields(payload):
            if fno == 1 and wt == 1:
                wall = struct.unpack("<d", v)[0]
            elif fno == 2 and wt == 0:
                step = v
            elif fno == 3 and wt == 2:
                version = v.decode()
            elif fno == 5 and wt == 2:
                for f2, _, value in _fields(v):
                    if f2 != 1:
                        continue
                    tag, sv = None, None
                    for f3, w3, x in _fields(value):
                        if f3 == 1:
                            tag = x.decode()
                        elif f3 == 2 and w3 == 5:
                            sv = struct.unpack("<f", x)[0]
                    if tag is not None and sv is not None:
                        series.setdefault(tag, []).append((step, wall, sv))
    return version, series
