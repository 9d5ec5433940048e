"""Unit checks of the graph executor's own ops (oracle/tf_graph.py) against independent statements of the same definitions:
torch's convolutions for Conv2D / Conv2DBackpropInput / Conv2DBackpropFilter (VALID and SAME, strides 1 and 2), numpy for the shape
ops whose TensorFlow semantics are easy to get wrong (StridedSlice masks, BroadcastGradientArgs, DynamicStitch, Select with a vector
condition, DivNoNan, Switch / Merge dead-branch propagation, staged variable writes)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf_graph as tg


def node(name, op, inputs=(), **attr):
    return {"name": name, "op": op, "input": list(inputs), "attr": attr}


def const(name, value, dtype=None):
    a = np.asarray(value)
    dt = dtype or {"f": "float32", "i": "int32", "b": "bool"}[a.dtype.kind]
    return node(name, "Const", value={"tensor": {"dtype": dt, "shape": list(a.shape), "values": a.reshape(-1).tolist()}})


def run(nodes, fetch, feed=None):
    return tg.Graph(nodes).run(fetch, feed)


@pytest.mark.parametrize("padding,stride,k,hw", [("VALID", 2, 4, (11, 14)), ("VALID", 2, 5, (9, 12)), ("SAME", 1, 3, (6, 7)), ("SAME", 2, 3, (7, 8)), ("SAME", 2, 4, (8, 9))])
def test_conv_ops_match_torch(padding, stride, k, hw):
    rng = np.random.RandomState(k + stride)
    x = rng.standard_normal((2,) + hw + (3,))
    w = rng.standard_normal((k, k, 3, 5))
    attrs = dict(strides={"list": {"i": [1, stride, stride, 1]}}, padding={"s": padding}, data_format={"s": "NHWC"})
    g = tg.Graph([node("x", "Placeholder"), node("w", "Placeholder"), node("y", "Conv2D", ["x", "w"], **attrs)])
    y = g.run("y", {"x": x, "w": w})
    # independent statement: torch cross-correlation on NCHW with TensorFlow's SAME rule (extra padding goes to the bottom / right)
    xt, wt = torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(3, 2, 0, 1)
    if padding == "SAME":
        oh, ow = -(-hw[0] // stride), -(-hw[1] // stride)
        ph, pw = max((oh - 1) * stride + k - hw[0], 0), max((ow - 1) * stride + k - hw[1], 0)
        xt = F.pad(xt, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    xt = xt.clone().requires_grad_(True)
    wt = wt.clone().requires_grad_(True)
    yt = F.conv2d(xt, wt, stride=stride)
    assert np.allclose(y, yt.detach().permute(0, 2, 3, 1).numpy(), atol=1e-12)
    dy = rng.standard_normal(y.shape)
    yt.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    gi = tg.Graph([node("s", "Placeholder"), node("w", "Placeholder"), node("dy", "Placeholder"), node("dx", "Conv2DBackpropInput", ["s", "w", "dy"], **attrs),
                   node("x", "Placeholder"), node("fs", "Placeholder"), node("dw", "Conv2DBackpropFilter", ["x", "fs", "dy"], **attrs)])
    dx, dw = gi.run(["dx", "dw"], {"s": np.array(x.shape, np.int32), "w": w, "dy": dy, "x": x, "fs": np.array(w.shape, np.int32)})
    dxt = xt.grad.permute(0, 2, 3, 1).numpy()
    if padding == "SAME":                                          # crop torch's gradient of the padded input back to the input
        dxt = dxt[:, ph // 2:ph // 2 + hw[0], pw // 2:pw // 2 + hw[1], :]
    assert np.allclose(dx, dxt, atol=1e-12)
    assert np.allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), atol=1e-11)


def test_strided_slice_masks():
    x = np.arange(24).reshape(2, 3, 4)
    base = [node("x", "Placeholder"), const("b", [1, 0, 1]), const("e", [2, 2, 3]), const("s", [1, 1, 2])]
    def ss(**masks):
        attrs = {k: {"i": v} for k, v in dict(dict(begin_mask=0, end_mask=0, shrink_axis_mask=0, ellipsis_mask=0, new_axis_mask=0), **masks).items()}
        return run(base + [node("y", "StridedSlice", ["x", "b", "e", "s"], **attrs)], "y", {"x": x})
    assert np.array_equal(ss(), x[1:2, 0:2, 1:3:2])
    assert np.array_equal(ss(begin_mask=0b010, end_mask=0b100), x[1:2, :2, 1::2])
    assert np.array_equal(ss(shrink_axis_mask=0b001), x[1, 0:2, 1:3:2])
    shape = run([node("x", "Placeholder"), node("sh", "Shape", ["x"]), const("b", [0]), const("e", [1]), const("s", [1]),
                 node("n", "StridedSlice", ["sh", "b", "e", "s"], begin_mask={"i": 0}, end_mask={"i": 0}, shrink_axis_mask={"i": 1}, ellipsis_mask={"i": 0}, new_axis_mask={"i": 0})],
                "n", {"x": x})
    assert shape == 2 and np.ndim(shape) == 0                       # the `batch = tf.shape(x)[0]` idiom all over tf.layers


def test_broadcast_gradient_args_and_reduction_roundtrip():
    cases = [((3, 4), (4,)), ((3, 4), ()), ((2, 1, 5), (3, 1)), ((5,), (5,)), ((1,), (7, 1)), ((32, 1), (32, 1))]
    for s0, s1 in cases:
        r0, r1 = run([const("a", list(s0)) if s0 else const("a", np.zeros(0, np.int32)), const("b", list(s1)) if s1 else const("b", np.zeros(0, np.int32)),
                      node("r", "BroadcastGradientArgs", ["a", "b"])], ["r:0", "r:1"])
        out = np.broadcast_shapes(s0, s1)
        g = np.random.RandomState(0).standard_normal(out)
        # summing the upstream gradient over the returned axes and reshaping gives the gradient of each broadcast operand (d(a+b))
        for shape, red in ((s0, r0), (s1, r1)):
            want = g
            lead = len(out) - len(shape)
            want = want.sum(axis=tuple(range(lead))) if lead else want
            for ax, n in enumerate(shape):
                if n == 1 and want.shape[ax] != 1:
                    want = want.sum(axis=ax, keepdims=True)
            got = g.sum(axis=tuple(int(a) for a in red)).reshape(shape) if len(red) else g.reshape(shape)
            assert np.allclose(got, want), (s0, s1, red)


def test_dynamic_stitch_select_divnonan_fill_tile():
    y = run([const("i0", [0, 2]), const("i1", [1, 3]), const("d0", [10, 30]), const("d1", [20, 40]), node("y", "DynamicStitch", ["i0", "i1", "d0", "d1"], N={"i": 2})], "y")
    assert y.tolist() == [10, 20, 30, 40]
    c, t, e = np.array([True, False]), np.ones((2, 3)), np.zeros((2, 3))
    y = run([node("c", "Placeholder"), node("t", "Placeholder"), node("e", "Placeholder"), node("y", "Select", ["c", "t", "e"])], "y", {"c": c, "t": t, "e": e})
    assert y.tolist() == [[1, 1, 1], [0, 0, 0]]                     # vector condition selects ROWS
    y = run([const("a", [1.0, 2.0, 3.0]), const("b", [2.0, 0.0, 4.0]), node("y", "DivNoNan", ["a", "b"])], "y")
    assert y.tolist() == [0.5, 0.0, 0.75]
    y = run([const("d", [2, 3]), const("v", 7.0), node("f", "Fill", ["d", "v"]), const("m", [2, 1]), node("y", "Tile", ["f", "m"])], "y")
    assert y.shape == (4, 3) and (y == 7).all()
    y = run([const("x", [[1.0, 2.0], [3.0, 4.0]]), const("ax", [0, 1]), node("m", "Mean", ["x", "ax"], keep_dims={"b": False}), const("a1", -1),
             node("s", "Sum", ["x", "a1"], keep_dims={"b": True})], ["m", "s"])
    assert float(y[0]) == 2.5 and y[1].tolist() == [[3.0], [7.0]]


def test_const_fill_semantics_and_float_width():
    g32 = tg.Graph([node("c", "Const", value={"tensor": {"dtype": "float32", "shape": [2, 2], "values": [0.1]}})], np.float32)
    g64 = tg.Graph(g32.nodes.values(), np.float64)
    assert g32.run("c").dtype == np.float32 and g32.run("c").tolist() == [[np.float32(0.1)] * 2] * 2        # one value fills the shape
    assert g64.run("c").dtype == np.float64
    c = tg.Graph([node("c", "Const", value={"tensor": {"dtype": "int32", "shape": [4], "values": [1, 2]}})]).run("c")
    assert c.tolist() == [1, 2, 2, 2]                                                                    # the last value repeats


def test_switch_merge_assert_and_staged_writes():
    def guard(pred):
        nodes = [const("p", pred), node("sw", "Switch", ["p", "p"]), node("t", "Identity", ["sw:1"]), node("f", "Identity", ["sw"]),
                 node("noop", "NoOp", ["^t"]), node("ct", "Identity", ["t", "^noop"]), const("msg", 0), node("as", "Assert", ["f", "msg"]),
                 node("cf", "Identity", ["f", "^as"]), node("mg", "Merge", ["cf", "ct"]), const("x", 5.0), node("y", "Identity", ["x", "^mg"])]
        return run(nodes, "y")
    assert guard(True) == 5.0                                       # true branch: the Assert is dead and never runs
    with pytest.raises(AssertionError):
        guard(False)
    nodes = [node("v", "VariableV2", shape={"shape": []}, dtype={"type": "float32"}), node("r", "Identity", ["v"]), const("one", 1.0),
             node("a", "Add", ["r", "one"]), node("w", "Assign", ["v", "a"]), node("twice", "Mul", ["r", "a", "^w"])]
    g = tg.Graph(nodes)
    g.set_variable("v", 2.0)
    assert g.run("twice") == 6.0 and g.vars["v"] == 3.0             # reads in one run() see the value from before it; the write lands after
    assert g.run("twice") == 12.0 and g.vars["v"] == 4.0
    with pytest.raises(KeyError):
        tg.Graph([node("ph", "Placeholder")]).run("ph")
    with pytest.raises(KeyError):
        tg.Graph([const("s", [2]), node("n", "RandomStandardNormal", ["s"])]).run("n")


def test_apply_adam_formula():
    rng = np.random.RandomState(3)
    var, m, v, grad = rng.standard_normal(7), 0.1 * rng.standard_normal(7), np.abs(rng.standard_normal(7)), rng.standard_normal(7)
    vd = lambda n: node(n, "VariableV2", shape={"shape": [7]}, dtype={"type": "float32"})
    nodes = [vd("var"), vd("m"), vd("v"), const("b1p", 0.81), const("b2p", 0.998), const("lr", 1e-3), const("b1", 0.9), const("b2", 0.999), const("eps", 1e-8),
             node("g", "Placeholder"), node("adam", "ApplyAdam", ["var", "m", "v", "b1p", "b2p", "lr", "b1", "b2", "eps", "g"], use_nesterov={"b": False})]
    g = tg.Graph(nodes)
    for n, x in (("var", var), ("m", m), ("v", v)):
        g.set_variable(n, x)
    g.run("adam", {"g": grad})
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3                         # (the helper stores python floats: no float32 rounding of the constants here)
    alpha = lr * np.sqrt(1 - 0.998) / (1 - 0.81)
    m2 = b1 * m + (1 - b1) * grad
    v2 = b2 * v + (1 - b2) * grad * grad
    assert np.allclose(g.vars["m"], m2, rtol=1e-12) and np.allclose(g.vars["v"], v2, rtol=1e-12)
    assert np.allclose(g.vars["var"], var - alpha * m2 / (np.sqrt(v2) + eps), rtol=1e-12)
