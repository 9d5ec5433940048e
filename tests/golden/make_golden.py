#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the READ-ONLY reference checkout.

Run in the authoring container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_golden.py [/root/reference]

Outputs (all small, committed):
  ref_variables.json   name -> {shape, dtype} for every variable in the reference's shipped TF checkpoints
                       (parsed from the `.index` SSTables: rgb VAE, seg VAE, PPO agent). Pins the
                       variable names/layouts/param counts our state-dict must reproduce.
  ref_event_scalars.json  first/last/min scalar values logged by the reference's own VAE training runs
                       (parsed from the TensorBoard event files). Coarse known-answers: the untrained
                       validation reconstruction loss must be ~= n_pixels*ln2.
  ref_events_head.bin  the first small records of the reference's rgb VAE training event file, verbatim (golden bytes for mi355/summary.py)
  ref_index/*.index    the three shipped TF bundle index files, verbatim (golden bytes for mi355/tf_bundle.py)
  real_frames_u8.npy   16 real CARLA frames (uint8 [16,80,160,3]) from vae/data/rgb/{0..15}.png and
  real_seg_u8.npy      the matching segmentation class-id maps (uint8 [16,80,160,1], values 0..12).

TensorFlow is not installed here, so the two container formats are parsed by hand:
  * `.index`  = LevelDB-style table (blocks of prefix-compressed key/value entries + 48-byte footer),
                values are BundleEntryProto {1:dtype, 2:TensorShapeProto{2:dim{1:size}}, 3:shard, 4:offset, 5:size}.
  * events    = TFRecord framing (u64 len, u32 crc, payload, u32 crc) of Event protos
                {1:wall_time f64, 2:step, 5:Summary{1:Value{1:tag, 2:simple_value f32}}}.
"""
import json
import os
import struct
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

TF_DTYPES = {1: "float32", 2: "float64", 3: "int32", 9: "int64", 7: "string", 10: "bool"}


def varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def proto_fields(buf):
    """Yield (field_no, wire_type, value) for one protobuf message (no schema)."""
    pos = 0
    while pos < len(buf):
        key, pos = varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fno, wt, val


def read_block(data, offset, size):
    """Decode one LevelDB table block -> list of (key, value)."""
    blk = data[offset:offset + size]
    assert data[offset + size] == 0, "compressed index blocks are not expected"
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = varint(blk, pos)
        non_shared, pos = varint(blk, pos)
        vlen, pos = varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        out.append((key, blk[pos:pos + vlen]))
        pos += vlen
    return out


def parse_index(path):
    data = open(path, "rb").read()
    footer = data[-48:]
    assert footer[-8:] == struct.pack("<Q", 0xdb4775248b80fb57), "bad table magic"
    _, p = varint(footer, 0)          # metaindex handle (offset, size)
    _, p = varint(footer, p)
    ioff, p = varint(footer, p)       # index handle
    isz, p = varint(footer, p)
    variables = {}
    for _, handle in read_block(data, ioff, isz):
        boff, q = varint(handle, 0)
        bsz, q = varint(handle, q)
        for key, val in read_block(data, boff, bsz):
            if key == b"":
                continue                  # BundleHeaderProto
            dtype, shape = None, []
            for fno, wt, v in proto_fields(val):
                if fno == 1:
                    dtype = TF_DTYPES.get(v, str(v))
                elif fno == 2:
                    for f2, _, dim in proto_fields(v):
                        if f2 == 2:
                            size = 0
                            for f3, _, s in proto_fields(dim):
                                if f3 == 1:
                                    size = s
                            shape.append(size)
            variables[key.decode()] = {"dtype": dtype, "shape": shape}
    return variables


def parse_events(path):
    data = open(path, "rb").read()
    pos, series = 0, {}
    while pos + 12 <= len(data):
        ln = struct.unpack("<Q", data[pos:pos + 8])[0]
        payload = data[pos + 12:pos + 12 + ln]
        pos += 12 + ln + 4
        step, wall = 0, None
        vals = []
        for fno, wt, v in proto_fields(payload):
            if fno == 1 and wt == 1:
                wall = struct.unpack("<d", v)[0]
            elif fno == 2 and wt == 0:
                step = v
            elif fno == 5 and wt == 2:
                for f2, _, value in proto_fields(v):
                    if f2 != 1:
                        continue
                    tag, sv = None, None
                    for f3, w3, x in proto_fields(value):
                        if f3 == 1:
                            tag = x.decode()
                        elif f3 == 2 and w3 == 5:
                            sv = struct.unpack("<f", x)[0]
                    if tag is not None and sv is not None:
                        vals.append((tag, sv))
        for tag, sv in vals:
            series.setdefault(tag, []).append((step, wall, sv))
    return series


def summarize(series):
    out = {}
    for tag, pts in series.items():
        vals = [p[2] for p in pts]
        imin = int(np.argmin(vals))
        out[tag] = {"n": len(pts), "first_step": pts[0][0], "first": vals[0], "last_step": pts[-1][0],
                    "last": vals[-1], "min": vals[imin], "min_step": pts[imin][0],
                    "wall_span_s": pts[-1][1] - pts[0][1]}
    return out


def main():
    from PIL import Image
    ckpts = {
        "vae_rgb": "vae/models/rgb_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/checkpoints/model.ckpt-232.index",
        "vae_seg": "vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/checkpoints/model.ckpt-255.index",
        "ppo_agent": "models/pretrained_agent/checkpoints/model.ckpt-705.index",
    }
    variables = {k: parse_index(os.path.join(REF, p)) for k, p in ckpts.items()}
    # the three shipped bundle INDEX files themselves (2.7 KB each; their .data shards are not part of the reference checkout):
    # byte-level golden vectors for the table writer / reader of mi355/tf_bundle.py
    import shutil
    os.makedirs(os.path.join(OUT, "ref_index"), exist_ok=True)
    for k, p in ckpts.items():
        dst = os.path.join(OUT, "ref_index", k + ".index")
        shutil.copyfile(os.path.join(REF, p), dst)
        os.chmod(dst, 0o644)
    meta = {"_source": ckpts, "_generator": "tests/golden/make_golden.py"}
    with open(os.path.join(OUT, "ref_variables.json"), "w") as f:
        json.dump({**meta, **variables}, f, indent=1, sort_keys=True)

    events = {
        "vae_rgb/train": "vae/models/rgb_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/logs/train",
        "vae_rgb/val": "vae/models/rgb_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/logs/val",
        "vae_seg/train": "vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/logs/train",
        "vae_seg/val": "vae/models/seg_bce_cnn_zdim64_beta1_kl_tolerance0.0_data/logs/val",
    }
    scal = {"_generator": "tests/golden/make_golden.py"}
    for k, d in events.items():
        d = os.path.join(REF, d)
        fn = sorted(os.listdir(d))[0]
        scal[k] = {"_source": os.path.join(os.path.relpath(d, REF), fn), **summarize(parse_events(os.path.join(d, fn)))}
    with open(os.path.join(OUT, "ref_event_scalars.json"), "w") as f:
        json.dump(scal, f, indent=1, sort_keys=True)
    # the first small TFRecord records of the rgb VAE training log, verbatim (file_version + the first scalar events; the graph_def
    # record between them is skipped -- every record carries its own length and checksums): golden bytes for mi355/summary.py
    d = os.path.join(REF, events["vae_rgb/train"])
    data = open(os.path.join(d, sorted(os.listdir(d))[0]), "rb").read()
    pos, recs = 0, []
    while pos + 12 <= len(data) and len(recs) < 12:
        ln = struct.unpack("<Q", data[pos:pos + 8])[0]
        recs.append(data[pos:pos + 16 + ln])
        pos += 16 + ln
    with open(os.path.join(OUT, "ref_events_head.bin"), "wb") as f:
        f.write(b"".join([r for r in recs if len(r) < 400][:7]))

    rgb = np.stack([np.asarray(Image.open(os.path.join(REF, "vae/data/rgb/%d.png" % i)))[:, :, :3] for i in range(16)])
    seg = np.stack([np.asarray(Image.open(os.path.join(REF, "vae/data/segmentation/%d.png" % i)))[:, :, :1] for i in range(16)])
    assert rgb.shape == (16, 80, 160, 3) and rgb.dtype == np.uint8, rgb.shape
    assert seg.shape == (16, 80, 160, 1) and seg.max() <= 12, (seg.shape, seg.max())
    np.save(os.path.join(OUT, "real_frames_u8.npy"), rgb)
    np.save(os.path.join(OUT, "real_seg_u8.npy"), seg)
    for k, v in variables.items():
        n_train = sum(int(np.prod(e["shape"])) for n, e in v.items()
                      if e["dtype"] == "float32" and "Adam" not in n and "_power" not in n and "policy_old" not in n)
        print(k, len(v), "variables; trainable-ish float params:", n_train)
    for k, v in scal.items():
        if k.startswith("_"):
            continue
        print(k, {t: (round(s["first"], 3), round(s["min"], 3), s["min_step"]) for t, s in v.items() if not t.startswith("_")})


if __name__ == "__main__":
    main()
