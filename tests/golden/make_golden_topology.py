#!/usr/bin/env python3
"""InceptionV3 topology as the reference repository itself lists it -> tests/golden/inception_v3_topology.json.

The arithmetic of the FID trunk is third-party (torchvision 0.9.1, absent here), but the reference ships a listing of
the very same graph: `image_realism/IS/bird/inception/slim/inception_model.py:48-299` on top of
`slim/ops.py` (conv2d = tf.nn.conv2d + batch_norm(eps) + relu, pools) and `slim/scopes.py` (arg_scope defaults).
This script EXECUTES those three reference files by path -- their own code decides every kernel size, stride, padding
mode, channel count, BatchNorm epsilon and concat order -- under stub third-party modules:

* stub `tensorflow`: symbolic tensors that carry a static shape; every `tf.nn.*` / `tf.concat` call appends a node
  (op, input node ids, attributes, output shape) to a list; variable / name scopes only keep names.
* stub `inception.slim.variables` / `losses` (variable creation: shape bookkeeping only).

Stored: numbers and op names only (the node list), no reference source text.

    python tests/golden/make_golden_topology.py        (needs /root/reference; run in the build container)
"""
import contextlib
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
SLIM = "/root/reference/image_realism/IS/bird/inception/slim"
sys.dont_write_bytecode = True

NODES = []
SCOPE = []


class TensorShape:
    def __init__(self, dims):
        self.dims = list(dims)

    def __getitem__(self, i):
        r = self.dims[i]
        return TensorShape(r) if isinstance(i, slice) else r

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def num_elements(self):
        n = 1
        for d in self.dims:
            n *= d
        return n

    def as_list(self):
        return list(self.dims)


class Tensor:
    def __init__(self, shape, node=None):
        self.shape = list(shape)
        self.node = node

    def get_shape(self):
        return TensorShape(self.shape)

    def set_shape(self, s):
        pass


def _emit(op, inputs, shape, **attrs):
    NODES.append({"id": len(NODES), "op": op, "scope": "/".join(SCOPE), "inputs": [t.node for t in inputs],
                  "out_shape": list(shape[1:]), **attrs})
    return Tensor(shape, len(NODES) - 1)


def _out_hw(h, w, k, s, padding):
    if padding == "SAME":                                      # TensorFlow: ceil(in / stride)
        return -(-h // s[0]), -(-w // s[1])
    assert padding == "VALID"
    return (h - k[0]) // s[0] + 1, (w - k[1]) // s[1] + 1


def _conv2d(x, w, strides, padding):
    kh, kw, cin, cout = w.shape
    assert strides[0] == strides[3] == 1 and x.shape[3] == cin
    oh, ow = _out_hw(x.shape[1], x.shape[2], (kh, kw), strides[1:3], padding)
    return _emit("conv2d", [x], [x.shape[0], oh, ow, cout], kernel=[kh, kw], stride=list(strides[1:3]), padding=padding,
                 cin=cin, cout=cout, in_hw=x.shape[1:3])


def _pool(kind):
    def f(x, ksize, strides, padding):
        assert ksize[0] == ksize[3] == 1 and strides[0] == strides[3] == 1
        oh, ow = _out_hw(x.shape[1], x.shape[2], ksize[1:3], strides[1:3], padding)
        return _emit(kind, [x], [x.shape[0], oh, ow, x.shape[3]], kernel=list(ksize[1:3]), stride=list(strides[1:3]),
                     padding=padding, in_hw=x.shape[1:3])
    return f


def _batch_normalization(x, mean, variance, offset, scale, variance_epsilon):
    return _emit("batch_norm", [x], x.shape, epsilon=variance_epsilon, has_beta=offset is not None, has_gamma=scale is not None,
                 uses_moving_statistics=bool(getattr(mean, "moving", False)))


def _relu(x):
    return _emit("relu", [x], x.shape)


def _concat(values, axis):
    assert axis == 3
    return _emit("concat", values, values[0].shape[:3] + [sum(v.shape[3] for v in values)], widths=[v.shape[3] for v in values])


@contextlib.contextmanager
def _scope(name, default_name=None, values=None, reuse=None):
    SCOPE.append(name or default_name)
    try:
        yield
    finally:
        SCOPE.pop()


def _variable(name, shape=None, initializer=None, **kw):
    v = Tensor(list(shape))
    v.moving = name.startswith("moving_")
    return v


def build_stubs():
    tf = types.ModuleType("tensorflow")
    tf.TensorShape = TensorShape
    tf.nn = types.SimpleNamespace(
        conv2d=_conv2d, max_pool=_pool("max_pool"), avg_pool=_pool("avg_pool"), relu=_relu,
        batch_normalization=_batch_normalization,
        softmax=lambda x, name=None: _emit("softmax", [x], x.shape),
        xw_plus_b=lambda x, w, b: _emit("fc", [x], [x.shape[0], w.shape[1]], cin=w.shape[0], cout=w.shape[1]),
        dropout=lambda x, keep_prob: x, bias_add=lambda x, b: x)
    tf.variable_scope = _scope
    tf.name_scope = _scope
    tf.concat = _concat
    tf.identity = lambda x: x
    tf.reshape = lambda x, s: _emit("flatten", [x], [x.shape[0], s[1]])
    for n in ("truncated_normal_initializer", "constant_initializer", "zeros_initializer", "ones_initializer"):
        setattr(tf, n, lambda *a, **k: None)
    tf.GraphKeys = types.SimpleNamespace(MOVING_AVERAGE_VARIABLES="moving_average_variables")
    tf.add_to_collection = lambda *a: None
    coll = {}
    fw_ops = types.ModuleType("tensorflow.python.framework.ops")
    fw_ops.get_collection = lambda k: coll.get(k, [])
    fw_ops.add_to_collection = lambda k, v: coll.setdefault(k, []).append(v)
    mods = {"tensorflow": tf, "tensorflow.python": types.ModuleType("tensorflow.python"),
            "tensorflow.python.framework": types.ModuleType("tensorflow.python.framework"),
            "tensorflow.python.framework.ops": fw_ops,
            "tensorflow.python.training": types.ModuleType("tensorflow.python.training"),
            "tensorflow.python.training.moving_averages": types.ModuleType("tensorflow.python.training.moving_averages"),
            "inception": types.ModuleType("inception"), "inception.slim": types.ModuleType("inception.slim"),
            "inception.slim.variables": types.ModuleType("inception.slim.variables"),
            "inception.slim.losses": types.ModuleType("inception.slim.losses")}
    mods["tensorflow.python.framework"].ops = fw_ops
    mods["tensorflow.python.training"].moving_averages = mods["tensorflow.python.training.moving_averages"]
    mods["inception.slim.variables"].variable = _variable
    mods["inception.slim.losses"].l2_regularizer = lambda wd: None
    mods["inception.slim"].variables = mods["inception.slim.variables"]
    mods["inception.slim"].losses = mods["inception.slim.losses"]
    mods["inception"].slim = mods["inception.slim"]
    sys.modules.update(mods)
    return mods


def load_reference(name):
    """Execute a reference slim file by path as module inception.slim.<name>."""
    spec = importlib.util.spec_from_file_location(f"inception.slim.{name}", os.path.join(SLIM, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"inception.slim.{name}"] = mod
    setattr(sys.modules["inception.slim"], name, mod)
    spec.loader.exec_module(mod)
    return mod


def main():
    build_stubs()
    scopes = load_reference("scopes")
    ops = load_reference("ops")
    model = load_reference("inception_model")
    images = Tensor([1, 299, 299, 3])
    NODES.append({"id": 0, "op": "input", "scope": "", "inputs": [], "out_shape": [299, 299, 3]})
    images.node = 0
    # the defaults the reference evaluates with (inception_score_star_bird.py builds the graph through
    # inception_model.inception_v3 inside slim's parameter scope: BatchNorm on every conv, epsilon 0.001)
    params = model.inception_v3_parameters
    ctx = params() if hasattr(params(), "__enter__") else contextlib.contextmanager(params)()
    with ctx:
        logits, end_points = model.inception_v3(images, dropout_keep_prob=1.0, num_classes=1000, is_training=False)
    out = {"source": "image_realism/IS/bird/inception/slim/{inception_model,ops,scopes}.py executed under stub tensorflow "
                     "(tests/golden/make_golden_topology.py)",
           "end_points": {k: v.node for k, v in end_points.items()}, "logits": logits.node, "nodes": NODES}
    path = os.path.join(HERE, "inception_v3_topology.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    convs = [n for n in NODES if n["op"] == "conv2d"]
    print(f"{len(NODES)} nodes, {len(convs)} convs (incl. the auxiliary head) -> {path}")


if __name__ == "__main__":
    main()
