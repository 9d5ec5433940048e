"""Executor for the reference's own serialized TensorFlow graphs (tests/golden/ref_graph_*.json.gz).

TEST INFRASTRUCTURE (see oracle/__init__.py): imported by tests/ only, never by the product path.

The reference ships the MetaGraphDef tf.train.Saver wrote next to each checkpoint (vae/models/*/checkpoints/model.ckpt-N.meta,
models/pretrained_agent/checkpoints/model.ckpt-N.meta): the node list of the graph it trained with -- forward pass, losses, the
`gradients/` sub-graph tf.gradients generated, every ApplyAdam node with its hyper-parameter constants.  TensorFlow 1.13 cannot be
installed here, so tests/golden/make_graph_fixture.py decodes those protobufs into JSON and this module runs them: each op below
follows the op's published definition (tensorflow/core/ops/*.cc docs, api_def) in plain numpy -- convolutions are written out from
their defining sums, not delegated to another framework's conv -- in float64 by default, so the oracle restatements
(oracle/vae_oracle.py, oracle/ppo_oracle.py) can be compared with the reference GRAPH to ~1e-10 instead of with a reading of the
reference's Python.  What this pins: which ops, in what order, with which attributes and constants (loss formulas, reductions, clip
ranges, Adam beta/epsilon/learning-rate wiring, gradient flow incl. stop-gradients).  What it cannot pin: the float32 rounding of
TensorFlow's kernels and its random streams (noise is fed in).

Execution model: Graph.run(fetches, feed) evaluates the fetched nodes recursively (data and control inputs); variable writes
(Assign / AssignAdd / ApplyAdam) are staged and committed when run() returns, i.e. every read in one run() sees the values from before
it -- the synchronous semantics everyone assumes of sess.run(train_op).  Switch/Merge (the assert guards tf.distributions and
vae/models.py:24-30 verify_range put in the graph) are executed with dead-branch propagation.
"""
import gzip
import json
import sys

import numpy as np


class _Dead:
    def __repr__(self):
        return "<dead>"


DEAD = _Dead()


def load_fixture(path):
    with gzip.open(path, "rb") as f:
        return json.load(f)


def _ref(s):
    """'name:1' -> ('name', 1, False);  '^name' -> ('name', 0, True)."""
    if s.startswith("^"):
        return s[1:], 0, True
    name, _, idx = s.partition(":")
    return name, int(idx) if idx else 0, False


class Graph:
    def __init__(self, nodes, float_dtype=np.float64):
        self.nodes = {n["name"]: n for n in nodes}
        self.order = [n["name"] for n in nodes]
        self.fd = np.dtype(float_dtype)
        self.vars = {}

    # ------------------------------------------------------------------ helpers
    def attr(self, node, key, kind=None, default=None):
        a = node["attr"].get(key)
        if a is None:
            return default
        if kind is None:
            (kind, val), = a.items()
            return val
        return a[kind]

    def _np_dtype(self, name):
        if name in ("float32", "float64"):
            return self.fd
        return np.dtype({"int32": np.int32, "int64": np.int64, "bool": np.bool_}[name])

    def _flt(self, x):
        x = np.asarray(x)
        return x.astype(self.fd) if x.dtype.kind == "f" else x

    def variables(self):
        return [n for n in self.order if self.nodes[n]["op"] == "VariableV2"]

    def variable_shape(self, name):
        return tuple(self.attr(self.nodes[name], "shape", "shape"))

    def set_variable(self, name, value):
        node = self.nodes[name]
        if node["op"] != "VariableV2":
            raise KeyError("%s is not a variable" % name)
        dt = self._np_dtype(self.attr(node, "dtype", "type"))
        v = np.array(value, dtype=dt)
        if tuple(v.shape) != self.variable_shape(name):
            raise ValueError("%s: shape %s, graph says %s" % (name, v.shape, self.variable_shape(name)))
        self.vars[name] = v

    def const(self, name):
        return self._const(self.nodes[name])

    def _const(self, node):
        t = self.attr(node, "value", "tensor")
        shape = tuple(t["shape"] or [])
        if t["dtype"] == "string":
            return np.array(t["values"], dtype=object).reshape(shape) if t["values"] else np.array("", dtype=object)
        dt = self._np_dtype(t["dtype"])
        vals = np.array(t["values"], dtype=dt)
        n = int(np.prod(shape)) if shape else 1
        if vals.size == n:
            return vals.reshape(shape)
        if vals.size == 0:
            return np.zeros(shape, dt)
        out = np.empty(n, dt)                     # TensorProto: fewer values than elements -> the last one repeats
        out[:vals.size] = vals
        out[vals.size:] = vals[-1]
        return out.reshape(shape)

    # ------------------------------------------------------------------ evaluation
    def run(self, fetches, feed=None):
        single = isinstance(fetches, str)
        names = [fetches] if single else list(fetches)
        self._cache, self._staged = {}, {}
        self._feed = {k: self._flt(v) for k, v in (feed or {}).items()}
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 20000))
        try:
            out = []
            for f in names:
                name, idx, ctrl = _ref(f)
                res = self._eval(name)
                out.append(None if (ctrl or len(res) == 0) else res[idx])
        finally:
            sys.setrecursionlimit(old)
        for k, v in self._staged.items():
            self.vars[k] = v
        self._cache = self._staged = self._feed = None
        return out[0] if single else out

    def _eval(self, name):
        if name in self._cache:
            return self._cache[name]
        if name in self._feed:
            res = (self._feed[name],)
            self._cache[name] = res
            return res
        node = self.nodes[name]
        op = node["op"]
        data, dead = [], False
        if op == "Merge":
            alive = None
            for i, s in enumerate(node["input"]):
                n, idx, ctrl = _ref(s)
                r = self._eval(n)
                if ctrl or r is DEAD or r[idx] is DEAD:
                    continue
                if alive is None:
                    alive = (r[idx], np.int32(i))
            res = DEAD if alive is None else alive
            self._cache[name] = res
            return res
        for s in node["input"]:
            n, idx, ctrl = _ref(s)
            r = self._eval(n)
            if r is DEAD:
                dead = True
            elif not ctrl:
                v = r[idx]
                if v is DEAD:
                    dead = True
                data.append(v)
        if dead:
            res = DEAD
        else:
            fn = getattr(self, "op_" + op, None)
            if fn is None:
                raise NotImplementedError("op %s (node %s)" % (op, name))
            res = fn(node, *data)
            if not isinstance(res, tuple):
                res = (res,)
        self._cache[name] = res
        return res

    def _var_of(self, node):
        """Variable node behind input 0 of a stateful op (through Identity chains)."""
        n = _ref(node["input"][0])[0]
        while self.nodes[n]["op"] == "Identity":
            n = _ref(self.nodes[n]["input"][0])[0]
        if self.nodes[n]["op"] != "VariableV2":
            raise ValueError("%s: input 0 is not a variable" % node["name"])
        return n

    # ------------------------------------------------------------------ sources, state
    def op_NoOp(self, node):
        return ()

    def op_Const(self, node):
        return self._const(node)

    def op_Placeholder(self, node):
        raise KeyError("placeholder %s must be fed" % node["name"])

    def op_PlaceholderWithDefault(self, node, x):
        return x

    def _random(self, node, *a):
        raise KeyError("random op %s must be fed (TensorFlow's random streams are not reproduced)" % node["name"])

    op_RandomStandardNormal = op_RandomUniform = op_TruncatedNormal = _random

    def op_VariableV2(self, node):
        if node["name"] not in self.vars:
            raise KeyError("variable %s is uninitialised" % node["name"])
        return self.vars[node["name"]]

    def op_Identity(self, node, x):
        return x

    op_StopGradient = op_PreventGradient = op_Identity

    def op_Assign(self, node, ref, value):
        self._staged[self._var_of(node)] = np.array(value, dtype=ref.dtype if isinstance(ref, np.ndarray) else None)
        return value

    def op_AssignAdd(self, node, ref, value):
        new = ref + value
        self._staged[self._var_of(node)] = new
        return new

    def op_ApplyAdam(self, node, var, m, v, b1p, b2p, lr, b1, b2, eps, grad):
        """training_ops.cc ApplyAdam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2); var -= lr_t*m/(sqrt(v)+eps)."""
        if self.attr(node, "use_nesterov", "b", False):
            raise NotImplementedError("nesterov Adam")
        names = [_ref(s)[0] for s in node["input"][:3]]
        alpha = lr * np.sqrt(1 - b2p) / (1 - b1p)
        m2 = m + (grad - m) * (1 - b1)
        v2 = v + (grad * grad - v) * (1 - b2)
        var2 = var - (m2 * alpha) / (np.sqrt(v2) + eps)
        for n, val in zip(names, (var2, m2, v2)):
            if self.nodes[n]["op"] != "VariableV2":
                raise ValueError("ApplyAdam %s: %s is not a variable" % (node["name"], n))
            self._staged[n] = val
        return var2

    # ------------------------------------------------------------------ control flow
    def op_Switch(self, node, data, pred):
        return (DEAD, data) if bool(pred) else (data, DEAD)

    def op_Assert(self, node, cond, *data):
        if not bool(cond):
            raise AssertionError("tf.Assert %s failed: %s" % (node["name"], [np.asarray(d).tolist() if not isinstance(d, str) else d for d in data]))
        return ()

    # ------------------------------------------------------------------ shapes
    def op_Shape(self, node, x):
        return np.array(np.shape(x), dtype=self._np_dtype(self.attr(node, "out_type", "type", "int32")))

    def op_Size(self, node, x):
        return np.array(np.size(x), dtype=self._np_dtype(self.attr(node, "out_type", "type", "int32")))

    def op_ShapeN(self, node, *xs):
        return tuple(np.array(np.shape(x), dtype=np.int32) for x in xs)

    def op_Reshape(self, node, x, shape):
        return np.reshape(x, [int(s) for s in shape])

    def op_Squeeze(self, node, x):
        dims = self.attr(node, "squeeze_dims", "list", {}).get("i", [])
        return np.squeeze(x, axis=tuple(int(d) for d in dims)) if dims else np.squeeze(x)

    def op_ExpandDims(self, node, x, axis):
        return np.expand_dims(x, int(axis))

    def op_Pack(self, node, *xs):
        return np.stack(xs, axis=self.attr(node, "axis", "i", 0))

    def op_ConcatV2(self, node, *xs):
        return np.concatenate([np.atleast_1d(x) for x in xs[:-1]], axis=int(xs[-1]))

    def op_Fill(self, node, dims, value):
        return np.full([int(d) for d in dims], value, dtype=np.asarray(value).dtype)

    def op_ZerosLike(self, node, x):
        return np.zeros_like(x)

    def op_OnesLike(self, node, x):
        return np.ones_like(x)

    def op_Range(self, node, start, limit, delta):
        return np.arange(start, limit, delta, dtype=np.asarray(start).dtype)

    def op_Tile(self, node, x, multiples):
        return np.tile(x, [int(m) for m in multiples])

    def op_Cast(self, node, x):
        return np.asarray(x).astype(self._np_dtype(self.attr(node, "DstT", "type")))

    def op_StridedSlice(self, node, x, begin, end, strides):
        bm, em = self.attr(node, "begin_mask", "i", 0), self.attr(node, "end_mask", "i", 0)
        sm = self.attr(node, "shrink_axis_mask", "i", 0)
        if self.attr(node, "ellipsis_mask", "i", 0) or self.attr(node, "new_axis_mask", "i", 0):
            raise NotImplementedError("StridedSlice ellipsis/new_axis masks")
        idx = []
        for d in range(len(begin)):
            b, e, s = int(begin[d]), int(end[d]), int(strides[d])
            if sm & (1 << d):
                idx.append(b)
                continue
            idx.append(slice(None if bm & (1 << d) else b, None if em & (1 << d) else e, s))
        return np.asarray(x)[tuple(idx)]

    def op_BroadcastArgs(self, node, s0, s1):
        return np.array(np.broadcast_shapes(tuple(int(a) for a in s0), tuple(int(a) for a in s1)), dtype=np.int32)

    def op_BroadcastGradientArgs(self, node, s0, s1):
        """Axes each operand's gradient is summed over to undo numpy-style broadcasting of shapes s0, s1."""
        s0, s1 = [int(a) for a in s0], [int(a) for a in s1]
        n = max(len(s0), len(s1))
        p0, p1 = [1] * (n - len(s0)) + s0, [1] * (n - len(s1)) + s1
        r0 = [i for i in range(n) if p0[i] == 1 and (p1[i] != 1 or i < n - len(s0))]
        r1 = [i for i in range(n) if p1[i] == 1 and (p0[i] != 1 or i < n - len(s1))]
        return np.array(r0, np.int32), np.array(r1, np.int32)

    def op_DynamicStitch(self, node, *args):
        n = self.attr(node, "N", "i")
        idx, data = args[:n], args[n:]
        size = max(int(np.max(i)) for i in idx if np.size(i)) + 1
        first = np.asarray(data[0])
        out = np.zeros((size,) + first.shape[np.ndim(idx[0]):], first.dtype)
        for i, d in zip(idx, data):
            out[np.asarray(i)] = d
        return out

    # ------------------------------------------------------------------ reductions
    def _reduce(self, fn, node, x, axes):
        axes = tuple(int(a) % max(np.ndim(x), 1) for a in np.atleast_1d(axes)) if np.ndim(x) else ()
        return fn(x, axis=axes, keepdims=bool(self.attr(node, "keep_dims", "b", False))) if np.ndim(x) else np.asarray(x)

    def op_Sum(self, node, x, axes):
        return self._reduce(np.sum, node, x, axes)

    def op_Mean(self, node, x, axes):
        return self._reduce(np.mean, node, x, axes)

    def op_Prod(self, node, x, axes):
        return self._reduce(np.prod, node, x, axes).astype(np.asarray(x).dtype)

    def op_Min(self, node, x, axes):
        return self._reduce(np.min, node, x, axes)

    def op_Max(self, node, x, axes):
        return self._reduce(np.max, node, x, axes)

    def op_All(self, node, x, axes):
        return self._reduce(np.all, node, x, axes)

    def op_AddN(self, node, *xs):
        out = xs[0]
        for x in xs[1:]:
            out = out + x
        return out

    # ------------------------------------------------------------------ element-wise
    def op_Add(self, node, a, b):
        return a + b

    op_AddV2 = op_Add

    def op_Sub(self, node, a, b):
        return a - b

    def op_Mul(self, node, a, b):
        return a * b

    def op_RealDiv(self, node, a, b):
        return a / b

    def op_DivNoNan(self, node, a, b):
        b = np.asarray(b)
        return np.where(b == 0, np.zeros_like(a * b), a / np.where(b == 0, np.ones_like(b), b))

    def op_FloorDiv(self, node, a, b):
        return np.floor_divide(a, b)

    def op_FloorMod(self, node, a, b):
        return np.mod(a, b)

    def op_Maximum(self, node, a, b):
        return np.maximum(a, b)

    def op_Minimum(self, node, a, b):
        return np.minimum(a, b)

    def op_Pow(self, node, a, b):
        return np.power(a, b)

    def op_SquaredDifference(self, node, a, b):
        return (a - b) * (a - b)

    def op_GreaterEqual(self, node, a, b):
        return a >= b

    def op_Greater(self, node, a, b):
        return a > b

    def op_LessEqual(self, node, a, b):
        return a <= b

    def op_Less(self, node, a, b):
        return a < b

    def op_Equal(self, node, a, b):
        return a == b

    def op_LogicalAnd(self, node, a, b):
        return np.logical_and(a, b)

    def op_Select(self, node, c, t, e):
        c = np.asarray(c)
        if c.ndim == 1 and np.ndim(t) > 1:              # Select's "condition is a vector over the first dimension" form
            c = c.reshape((-1,) + (1,) * (np.ndim(t) - 1))
        return np.where(c, t, e)

    def op_Neg(self, node, x):
        return -x

    def op_Exp(self, node, x):
        return np.exp(x)

    def op_Log(self, node, x):
        return np.log(x)

    def op_Log1p(self, node, x):
        return np.log1p(x)

    def op_Square(self, node, x):
        return x * x

    def op_Sqrt(self, node, x):
        return np.sqrt(x)

    def op_Floor(self, node, x):
        return np.floor(x)

    def op_Reciprocal(self, node, x):
        return 1.0 / x

    def op_Sigmoid(self, node, x):
        return 1.0 / (1.0 + np.exp(-x))

    def op_Tanh(self, node, x):
        return np.tanh(x)

    def op_Relu(self, node, x):
        return np.maximum(x, 0)

    def op_ReluGrad(self, node, g, features):
        return np.where(features > 0, g, np.zeros_like(g))

    def op_TanhGrad(self, node, y, dy):
        return dy * (1.0 - y * y)

    def op_SigmoidGrad(self, node, y, dy):
        return dy * y * (1.0 - y)

    def op_ReciprocalGrad(self, node, y, dy):
        return -dy * y * y

    def op_SqrtGrad(self, node, y, dy):
        return dy * 0.5 / y

    # ------------------------------------------------------------------ dense / conv layers (NHWC activations, HWIO filters)
    def _nhwc(self, node):
        if self.attr(node, "data_format", "s", "NHWC") != "NHWC":
            raise NotImplementedError("data_format %s" % self.attr(node, "data_format", "s"))

    def op_BiasAdd(self, node, x, b):
        self._nhwc(node)
        return x + b

    def op_BiasAddGrad(self, node, g):
        self._nhwc(node)
        return np.sum(g, axis=tuple(range(np.ndim(g) - 1)))

    def op_MatMul(self, node, a, b):
        if self.attr(node, "transpose_a", "b", False):
            a = a.T
        if self.attr(node, "transpose_b", "b", False):
            b = b.T
        return a @ b

    def _conv_geometry(self, node, in_hw, k_hw):
        """-> (stride_h, stride_w, pad_top, pad_left, out_h, out_w) for the op's strides/padding attrs (dilation 1 only)."""
        self._nhwc(node)
        st = self.attr(node, "strides", "list")["i"]
        dil = (self.attr(node, "dilations", "list") or {}).get("i", [1, 1, 1, 1])
        if st[0] != 1 or st[3] != 1 or any(d != 1 for d in dil):
            raise NotImplementedError("strides %s dilations %s" % (st, dil))
        sh, sw = int(st[1]), int(st[2])
        pad = self.attr(node, "padding", "s")
        (ih, iw), (kh, kw) = in_hw, k_hw
        if pad == "VALID":
            return sh, sw, 0, 0, (ih - kh) // sh + 1, (iw - kw) // sw + 1
        if pad == "SAME":
            oh, ow = -(-ih // sh), -(-iw // sw)
            ph, pw = max((oh - 1) * sh + kh - ih, 0), max((ow - 1) * sw + kw - iw, 0)
            return sh, sw, ph // 2, pw // 2, oh, ow
        raise NotImplementedError("padding %s" % pad)

    def _pad_input(self, x, pt, pl, sh, sw, kh, kw, oh, ow):
        need_h, need_w = (oh - 1) * sh + kh, (ow - 1) * sw + kw
        pb, pr = max(need_h - x.shape[1] - pt, 0), max(need_w - x.shape[2] - pl, 0)
        return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0))) if (pt or pl or pb or pr) else x

    def op_Conv2D(self, node, x, w):
        """out[b,i,j,k] = sum_{di,dj,q} x[b, s*i+di, s*j+dj, q] * w[di,dj,q,k]   (cross-correlation, tf.nn.conv2d's definition)."""
        kh, kw = w.shape[:2]
        sh, sw, pt, pl, oh, ow = self._conv_geometry(node, x.shape[1:3], (kh, kw))
        xp = self._pad_input(x, pt, pl, sh, sw, kh, kw, oh, ow)
        out = np.zeros((x.shape[0], oh, ow, w.shape[3]), x.dtype)
        for di in range(kh):
            for dj in range(kw):
                out += xp[:, di:di + sh * (oh - 1) + 1:sh, dj:dj + sw * (ow - 1) + 1:sw, :] @ w[di, dj]
        return out

    def op_Conv2DBackpropInput(self, node, input_sizes, w, dy):
        """Gradient of Conv2D w.r.t. its input (= tf.nn.conv2d_transpose): dx[b, s*i+di, s*j+dj, q] += dy[b,i,j,k] * w[di,dj,q,k]."""
        n, ih, iw, c = [int(v) for v in input_sizes]
        kh, kw = w.shape[:2]
        sh, sw, pt, pl, oh, ow = self._conv_geometry(node, (ih, iw), (kh, kw))
        if dy.shape[1:3] != (oh, ow):
            raise ValueError("%s: out_backprop %s does not match the forward geometry %s" % (node["name"], dy.shape, (oh, ow)))
        ph, pw = max((oh - 1) * sh + kh, ih + pt), max((ow - 1) * sw + kw, iw + pl)
        dx = np.zeros((n, ph, pw, c), dy.dtype)
        for di in range(kh):
            for dj in range(kw):
                dx[:, di:di + sh * (oh - 1) + 1:sh, dj:dj + sw * (ow - 1) + 1:sw, :] += dy @ w[di, dj].T
        return dx[:, pt:pt + ih, pl:pl + iw, :]

    def op_Conv2DBackpropFilter(self, node, x, filter_sizes, dy):
        """Gradient of Conv2D w.r.t. its filter: dw[di,dj,q,k] = sum_{b,i,j} x[b, s*i+di, s*j+dj, q] * dy[b,i,j,k]."""
        kh, kw, cin, cout = [int(v) for v in filter_sizes]
        sh, sw, pt, pl, oh, ow = self._conv_geometry(node, x.shape[1:3], (kh, kw))
        xp = self._pad_input(x, pt, pl, sh, sw, kh, kw, oh, ow)
        dw = np.zeros((kh, kw, cin, cout), x.dtype)
        g = dy.reshape(-1, cout)
        for di in range(kh):
            for dj in range(kw):
                dw[di, dj] = xp[:, di:di + sh * (oh - 1) + 1:sh, dj:dj + sw * (ow - 1) + 1:sw, :].reshape(-1, cin).T @ g
        return dw
