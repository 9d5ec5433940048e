"""Shared by the reference-graph tests: load a tests/golden/ref_graph_*.json.gz fixture into oracle.tf_graph.Graph and give its variables
values (trainable ones from a {reference variable name: array} dict, Adam slots zero, beta powers / counters at their initial values)."""
import os

import numpy as np

from oracle import tf_graph as tg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

VAE_EPS = "vae/Normal/sample/random_normal/RandomStandardNormal"          # fed [1, B, z]: Normal.sample() draws sample_shape [1] + batch shape
PPO_EPS = "policy/Normal/sample/random_normal/RandomStandardNormal"       # fed [1, M, A]


def load_graph(which, float_dtype=np.float64):
    doc = tg.load_fixture(os.path.join(GOLDEN, "ref_graph_%s.json.gz" % which))
    return tg.Graph(doc["nodes"], float_dtype), doc


def adam_nodes(g):
    """[(ApplyAdam node name, variable name, m slot, v slot, gradient tensor)] in graph order."""
    out = []
    for n in g.order:
        node = g.nodes[n]
        if node["op"] == "ApplyAdam":
            out.append((n, node["input"][0], node["input"][1], node["input"][2], node["input"][9]))
    return out


def init_variables(g, named):
    """Trainable variables <- named; every other variable <- what its own initializer Assign feeds it when that is a constant / zeros."""
    for k, v in named.items():
        g.set_variable(k, v)
    for _, var, m, v, _ in adam_nodes(g):
        g.set_variable(m, np.zeros(g.variable_shape(m)))
        g.set_variable(v, np.zeros(g.variable_shape(v)))
    for name in g.variables():
        if name in g.vars:
            continue
        init = name + "/initial_value"
        if init in g.nodes and g.nodes[init]["op"] == "Const":
            g.set_variable(name, g.const(init))
        elif name + "/Initializer/zeros" in g.nodes:
            g.set_variable(name, np.zeros(g.variable_shape(name)))
    return g
