"""The InceptionV3 graph this repository executes == the graph the reference repository lists (SURVEY 8 a5 / c).

`tests/golden/inception_v3_topology.json` is what the reference's own TF-slim listing builds
(`image_realism/IS/bird/inception/slim/inception_model.py:48-299` over `slim/ops.py`, `slim/scopes.py`), recorded by
`tests/golden/make_golden_topology.py` running those files under a stub tensorflow.  Here the CPU oracle
(`oracle/inception_oracle.py`) and the product's module tree (`tise_toolbox_amd.inception.InceptionV3`) are traced
op by op (conv / batch-norm / relu / pools / concat with their dataflow) and compared node for node with the
pool3 ancestors of that listing: 94 convolutions, kernel / stride / padding / channels / spatial sizes, BatchNorm
epsilon, pool kinds and windows, concat order and widths, 5.7112 GMAC per image.

Known, deliberate differences between the torchvision model the FID path uses (`FID/inception.py:57`) and the TF-slim
listing -- the comparison normalises exactly these and nothing else:
  1. 3x3 stride-1 SAME average pools: TensorFlow divides by the number of in-image taps, torchvision's
     `F.avg_pool2d(x, 3, 1, 1)` by 9 (count_include_pad=True).
  2. BatchNorm: slim's has no gamma (scale=False), torchvision's is affine (weight + bias).
  3. torchvision names paddings explicitly ((k-1)/2 per side); slim says SAME at stride 1.
  4. The auxiliary head (2 convs) exists in both and is on neither pool3 path.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inception_v3_topology.json")


def reference_pool3_graph():
    """Canonical op list of the reference listing restricted to the ancestors of the global average pool."""
    t = json.load(open(GOLDEN))
    nodes = {n["id"]: n for n in t["nodes"]}
    final = [n for n in t["nodes"] if n["op"] == "avg_pool" and n["scope"].endswith("logits/pool")]
    assert len(final) == 1
    keep, stack = set(), [final[0]["id"]]
    while stack:
        i = stack.pop()
        if i not in keep:
            keep.add(i)
            stack.extend(nodes[i]["inputs"])
    canon, where = [], {}                     # where: reference node id -> index in canon

    def pad_of(n):
        if n["padding"] == "VALID":
            return [0, 0]
        assert n["stride"] == [1, 1]          # SAME only at stride 1 in this graph: (k-1)/2 per side
        return [n["kernel"][0] // 2, n["kernel"][1] // 2]
    for n in t["nodes"]:
        if n["id"] not in keep:
            continue
        ins = [where[i] for i in n["inputs"]]
        if n["op"] == "input":
            canon.append(("input", (), ()))
        elif n["op"] == "conv2d":
            canon.append(("conv", (n["cin"], n["cout"], *n["kernel"], *n["stride"], *pad_of(n), *n["in_hw"], *n["out_shape"][:2]), tuple(ins)))
        elif n["op"] == "batch_norm":
            assert n["uses_moving_statistics"] and n["has_beta"]
            canon.append(("bn", (n["epsilon"],), tuple(ins)))
        elif n["op"] == "relu":
            canon.append(("relu", (), tuple(ins)))
        elif n["op"] in ("max_pool", "avg_pool"):
            if n["id"] == final[0]["id"]:
                assert n["kernel"] == n["in_hw"] == [8, 8] and n["padding"] == "VALID"
                canon.append(("global_avg", (), tuple(ins)))
            else:
                canon.append((n["op"], (*n["kernel"], *n["stride"], *pad_of(n), *n["in_hw"], *n["out_shape"][:2]), tuple(ins)))
        elif n["op"] == "concat":
            canon.append(("concat", tuple(n["widths"]), tuple(ins)))
        else:
            raise AssertionError(n["op"])
        where[n["id"]] = len(canon) - 1
    return canon


class Tracer:
    """Records conv / bn / relu / pool / concat calls of a PyTorch forward (NCHW) with their dataflow."""

    def __init__(self):
        self.ops = []
        self.orig = {}

    def _tag(self, t, entry):
        self.ops.append(entry)
        t._node = len(self.ops) - 1
        return t

    def __enter__(self):
        tr, o = self, self.orig
        for name in ("conv2d", "batch_norm", "relu", "max_pool2d", "avg_pool2d", "adaptive_avg_pool2d"):
            o[name] = getattr(F, name)
        o["cat"] = torch.cat

        def pair(v):
            return [v, v] if isinstance(v, int) else list(v)

        def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
            y = o["conv2d"](x, w, b, stride, padding, *a, **k)
            assert b is None
            return tr._tag(y, ("conv", (w.shape[1], w.shape[0], w.shape[2], w.shape[3], *pair(stride), *pair(padding),
                                        x.shape[2], x.shape[3], y.shape[2], y.shape[3]), (x._node,)))

        def batch_norm(x, rm, rv, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
            assert not training and rm is not None and weight is not None and bias is not None
            src = x._node
            return tr._tag(o["batch_norm"](x, rm, rv, weight, bias, training, momentum, eps), ("bn", (eps,), (src,)))

        def relu(x, inplace=False):
            src = x._node
            return tr._tag(o["relu"](x, inplace), ("relu", (), (src,)))

        def pool(kind):
            def f(x, kernel_size, stride=None, padding=0, *a, **k):
                # only torchvision's defaults may be passed on: dilation 1, ceil_mode False, count_include_pad True
                assert all(v in (1, False, None) for v in a) and all(v in (1, False, None) for v in k.values()) and \
                    "count_include_pad" not in k and "divisor_override" not in k
                y = o[kind + "2d"](x, kernel_size, stride, padding)
                return tr._tag(y, (kind, (*pair(kernel_size), *pair(stride if stride is not None else kernel_size), *pair(padding),
                                          x.shape[2], x.shape[3], y.shape[2], y.shape[3]), (x._node,)))
            return f

        def adaptive(x, size):
            assert tuple(pair(size)) == (1, 1)
            return tr._tag(o["adaptive_avg_pool2d"](x, size), ("global_avg", (), (x._node,)))

        def cat(ts, dim=0):
            assert dim == 1
            return tr._tag(o["cat"](ts, dim), ("concat", tuple(t.shape[1] for t in ts), tuple(t._node for t in ts)))
        F.conv2d, F.batch_norm, F.relu = conv2d, batch_norm, relu
        F.max_pool2d, F.avg_pool2d, F.adaptive_avg_pool2d = pool("max_pool"), pool("avg_pool"), adaptive
        torch.cat = cat
        return self

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(torch if name == "cat" else F, name, fn)

    def input(self, x):
        return self._tag(x, ("input", (), ()))


@pytest.fixture(scope="module")
def ref_graph():
    return reference_pool3_graph()


def test_reference_listing_numbers(ref_graph):
    convs = [op for op in ref_graph if op[0] == "conv"]
    assert len(convs) == 94
    macs = sum(c[1][0] * c[1][1] * c[1][2] * c[1][3] * c[1][10] * c[1][11] for c in convs)
    assert macs == 5_711_168_096 and round(macs / 1e9, 4) == 5.7112       # 5.7112 GMAC per image
    assert sum(1 for op in ref_graph if op[0] == "max_pool") == 4 and sum(1 for op in ref_graph if op[0] == "avg_pool") == 9
    assert sum(1 for op in ref_graph if op[0] == "concat") == 15 and all(op[1] == (0.001,) for op in ref_graph if op[0] == "bn")
    params = sum(c[1][0] * c[1][1] * c[1][2] * c[1][3] for c in convs)
    assert params == 21_751_136                                          # conv weights on the pool3 path


def test_oracle_executes_the_reference_graph(ref_graph):
    from oracle import inception_oracle
    from tise_toolbox_amd.inception import Inception3
    sd = {k: v.float() for k, v in Inception3().state_dict().items()}
    counter = inception_oracle.MacCounter()
    with Tracer() as tr, torch.no_grad():
        x = tr.input(torch.rand(1, 3, 299, 299))
        inception_oracle.inception_forward(sd, x, resize_input=False, normalize_input=False, counter=counter)
    assert tr.ops == ref_graph
    assert counter.convs == 94 and counter.macs == 5_711_168_096


def test_product_module_executes_the_reference_graph(ref_graph):
    from tise_toolbox_amd.inception import Inception3, InceptionV3
    import tise_toolbox_amd.inception as inc
    old = inc.build_inception3
    inc.build_inception3 = lambda *a, **k: Inception3(num_classes=1000).eval()      # topology only: skip the calibration
    try:
        m = InceptionV3([3]).eval()
    finally:
        inc.build_inception3 = old
    with Tracer() as tr, torch.no_grad():
        x = tr.input(torch.rand(1, 3, 299, 299))
        m(x, prenormalized=True)
    assert tr.ops == ref_graph
