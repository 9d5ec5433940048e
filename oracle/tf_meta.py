"""Decoder for TensorFlow MetaGraphDef (`.meta`) files -> plain node lists for oracle/tf_graph.py.

TEST INFRASTRUCTURE (see oracle/__init__.py).  TensorFlow is not needed: a GraphDef is plain protobuf, decoded here from the wire format
with the field numbers of tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape}.proto and protobuf/meta_graph.proto:
  MetaGraphDef{2: GraphDef}   GraphDef{1: NodeDef}   NodeDef{1: name, 2: op, 3: input*, 4: device, 5: map<string, AttrValue>}
  AttrValue{1: list, 2: s, 3: i, 4: f, 5: b, 6: type, 7: shape, 8: tensor, 10: func}   ListValue{2: s*, 3: i*, 4: f*, 5: b*, 6: type*, 7: shape*}
  TensorProto{1: dtype, 2: shape, 4: tensor_content, 5: float_val*, 6: double_val*, 7: int_val*, 8: string_val*, 10: int64_val*, 11: bool_val*}
  TensorShapeProto{2: dim{1: size}, 3: unknown_rank}
Used by tests/golden/make_graph_fixture.py (the committed fixtures of the reference's shipped graphs) and tools/check_meta.py (the same
comparison for any `.meta` a user of the reference has next to their own checkpoints).
"""
import re
import struct

import numpy as np

DT = {1: "float32", 2: "float64", 3: "int32", 7: "string", 9: "int64", 10: "bool"}


def varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fno, wt, val


def sint(v):                                         # varint -> signed int64
    return v - (1 << 64) if v >= (1 << 63) else v


def packed_varints(wt, v):
    if wt == 0:
        return [sint(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = varint(v, pos)
        out.append(sint(x))
    return out


def packed_fixed(wt, v, fmt, size):
    if wt != 2:
        return [struct.unpack("<" + fmt, v)[0]]
    return list(struct.unpack("<%d%s" % (len(v) // size, fmt), v))


def shape_proto(buf):
    dims, unknown = [], False
    for fno, wt, v in fields(buf):
        if fno == 2:
            size = 0
            for f2, _, x in fields(v):
                if f2 == 1:
                    size = sint(x)
            dims.append(size)
        elif fno == 3:
            unknown = bool(v)
    return None if unknown else dims


def tensor_proto(buf):
    dtype, shape, content, vals = 0, [], None, []
    for fno, wt, v in fields(buf):
        if fno == 1:
            dtype = v
        elif fno == 2:
            shape = shape_proto(v)
        elif fno == 4:
            content = v
        elif fno == 5:
            vals += packed_fixed(wt, v, "f", 4)
        elif fno == 6:
            vals += packed_fixed(wt, v, "d", 8)
        elif fno in (7, 10):
            vals += packed_varints(wt, v)
        elif fno == 11:
            vals += [bool(x) for x in packed_varints(wt, v)]
        elif fno == 8:
            vals.append(v.decode("latin1"))
    name = DT.get(dtype, "dtype%d" % dtype)
    if content is not None and name != "string":
        vals = np.frombuffer(content, dtype=np.dtype(name)).tolist()
    return {"dtype": name, "shape": shape, "values": vals}


def attr_value(buf):
    for fno, wt, v in fields(buf):
        if fno == 2:
            return {"s": v.decode("latin1")}
        if fno == 3:
            return {"i": sint(v)}
        if fno == 4:
            return {"f": struct.unpack("<f", v)[0]}
        if fno == 5:
            return {"b": bool(v)}
        if fno == 6:
            return {"type": DT.get(v, "dtype%d" % v)}
        if fno == 7:
            return {"shape": shape_proto(v)}
        if fno == 8:
            return {"tensor": tensor_proto(v)}
        if fno == 1:
            lst = {}
            for f2, w2, x in fields(v):
                if f2 == 2:
                    lst.setdefault("s", []).append(x.decode("latin1"))
                elif f2 == 3:
                    lst.setdefault("i", []).extend(packed_varints(w2, x))
                elif f2 == 4:
                    lst.setdefault("f", []).extend(packed_fixed(w2, x, "f", 4))
                elif f2 == 5:
                    lst.setdefault("b", []).extend(bool(y) for y in packed_varints(w2, x))
                elif f2 == 6:
                    lst.setdefault("type", []).extend(DT.get(y, "dtype%d" % y) for y in packed_varints(w2, x))
                elif f2 == 7:
                    lst.setdefault("shape", []).append(shape_proto(x))
            return {"list": lst}
    return {"list": {}}                                           # an empty AttrValue is an empty list


def graph_nodes(meta_path):
    data = open(meta_path, "rb").read()
    nodes = []
    for fno, wt, v in fields(data):
        if fno != 2 or wt != 2:
            continue
        for f2, w2, nd in fields(v):
            if f2 != 1 or w2 != 2:
                continue
            n = {"name": None, "op": None, "input": [], "attr": {}}
            for f3, _, x in fields(nd):
                if f3 == 1:
                    n["name"] = x.decode()
                elif f3 == 2:
                    n["op"] = x.decode()
                elif f3 == 3:
                    n["input"].append(x.decode())
                elif f3 == 5:
                    key, val = None, b""
                    for f4, _, y in fields(x):
                        if f4 == 1:
                            key = y.decode()
                        elif f4 == 2:
                            val = y
                    if key not in ("_class", "_output_shapes"):
                        n["attr"][key] = attr_value(val)
            nodes.append(n)
    return nodes


DROP_OPS = ("ScalarSummary", "MergeSummary", "SaveV2", "RestoreV2", "HistogramSummary", "ImageSummary", "TensorSummaryV2", "TensorSummary")


def is_saver(name):
    return any(part == "save" or part.startswith("save_") for part in name.split("/"))


def prune(nodes):
    """Saver and summary plumbing out (and whatever consumes only that); everything that computes stays: forward, losses, the gradients/
    sub-graph, ApplyAdam and its constants, the initializers."""
    dropped = {n["name"] for n in nodes if is_saver(n["name"]) or n["op"] in DROP_OPS or
               (n["op"] == "NoOp" and re.fullmatch(r"(.*/)?init(_\d+)?", n["name"]))}      # one tf.global_variables_initializer() group per call
    while True:
        more = {n["name"] for n in nodes if n["name"] not in dropped and any(i.lstrip("^").split(":")[0] in dropped for i in n["input"])}
        if not more:
            break
        dropped |= more
    return [n for n in nodes if n["name"] not in dropped]
