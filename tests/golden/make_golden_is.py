#!/usr/bin/env python3
"""Golden vectors for the IS* family (SURVEY.md section 8 a8 / a8' / a8''), produced by the reference scripts.

The three reference modules cannot be imported as they are: two need TensorFlow 1.x (+ a network download,
+ `inception.slim`), the third needs torchvision + a weight file and runs the whole evaluation at import.
Everything AROUND the network forward -- the file walk, the batch loop, the temperature constant, the class
slice, the split rule, the KL / entropy reduction, mean / std and the result-file text -- is the reference's own
code, so it is executed for real, by path (`runpy.run_path`), under stub third-party modules:

* stub `tensorflow`: a tiny symbolic graph (`matmul`/`slice`/`div`/`softmax` nodes record their constants, so the
  TEMPERATURE and the SLICE come from the reference's source, not from this file); `Session.run` evaluates the
  node graph in float32 on "logits" that are a fixed seeded function of the fed image batch.
* stub `scipy.misc.imread/imresize` (Pillow, as scipy 1.1.0 implemented them), stub `inception.slim`.
* stub `torchvision` (`models.inception_v3` -> a callable that returns seeded logits; `transforms` -> Pillow).

The fixtures store the raw logits the stub produced (in the reference's processing order) and what the
reference computed from them (full-precision floats + the text it wrote).  Numbers only; no reference text.

    python tests/golden/make_golden_is.py        (needs /root/reference; run in the build container)
"""
import contextlib
import io
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


# ------------------------------------------------------------------------------------------------ seeded "network"
class SeededLogits:
    """logits = fixed random projection of 96 fixed sample positions of the image batch (float32)."""
    def __init__(self, n_classes, seed, scale):
        rng = np.random.default_rng(seed)
        self.w = (rng.standard_normal((96, n_classes)) * scale).astype(np.float32)
        self.pos = rng.integers(0, 299 * 299 * 3, size=96)
        self.log = []

    def __call__(self, batch_nhwc_or_nchw):
        x = np.asarray(batch_nhwc_or_nchw, dtype=np.float32)
        flat = x.reshape(x.shape[0], -1)[:, self.pos]
        flat = (flat - flat.mean(axis=1, keepdims=True)) / (flat.std(axis=1, keepdims=True) + 1e-6)
        out = (flat @ self.w).astype(np.float32)
        self.log.append(out.copy())
        return out


# ------------------------------------------------------------------------------------------------ stub tensorflow
class Node:
    def __init__(self, op, inputs=(), **attrs):
        self.op, self.inputs, self.attrs = op, inputs, attrs
        self.graph = self
        self.outputs = []

    def get_operations(self):
        return []


def evaluate(node, feed):
    """float32 evaluation of the recorded graph (tf.div on float32 tensors, tf.nn.softmax)."""
    if node.op == "logits":
        return node.attrs["net"](feed)
    if node.op == "const":
        return np.float32(node.attrs["value"])
    if node.op == "slice":
        x = evaluate(node.inputs[0], feed)
        b, s = node.attrs["begin"], node.attrs["size"]
        return x[b[0]:b[0] + s[0], b[1]:b[1] + s[1]]
    if node.op == "div":
        return (evaluate(node.inputs[0], feed) / evaluate(node.inputs[1], feed)).astype(np.float32)
    if node.op == "softmax":
        z = evaluate(node.inputs[0], feed)
        z = z - z.max(axis=1, keepdims=True)
        e = np.exp(z)
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    raise ValueError(node.op)


class Ctx:
    def __init__(self, *a, **k):
        self.gpu_options = types.SimpleNamespace(allow_growth=False)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def as_default(self):
        return self


def make_tensorflow(net, record, flag_values):
    tf = types.ModuleType("tensorflow")

    class Session(Ctx):
        def __init__(self, config=None):
            super().__init__()
            g = types.SimpleNamespace()
            g.get_tensor_by_name = lambda name: Node("pool3")
            g.get_operation_by_name = lambda name: types.SimpleNamespace(inputs=[None, Node("w")])
            self.graph = g

        def run(self, node, feed):
            (key, value), = feed.items()
            record["feed_keys"].add(key)
            return evaluate(node, value)

    def matmul(a, b):
        return Node("logits", net=net)

    def div(a, b):
        record["temperature"] = float(b.attrs["value"])
        return Node("div", (a, b))

    def slice_(x, begin, size):
        record["slice_begin"], record["slice_size"] = list(begin), list(size)
        return Node("slice", (x,), begin=begin, size=size)

    tf.ConfigProto = Ctx
    tf.Session = Session
    tf.Graph = Ctx
    tf.device = Ctx
    tf.GraphDef = lambda: types.SimpleNamespace(ParseFromString=lambda b: None)
    tf.import_graph_def = lambda *a, **k: None
    tf.TensorShape = lambda s: s
    tf.squeeze = lambda x, axes: x
    tf.matmul = matmul
    tf.div = div
    tf.slice = slice_
    tf.constant = lambda v: Node("const", value=v)
    tf.float32 = np.float32
    tf.placeholder = lambda dtype, shape, name=None: Node("placeholder")
    tf.nn = types.SimpleNamespace(softmax=lambda x: Node("softmax", (x,)), relu=None)

    class FastGFile(Ctx):
        def read(self):
            return b""
    tf.gfile = types.SimpleNamespace(FastGFile=FastGFile)

    # tf.app.flags / tf.app.run (bird script)
    flags_obj = types.SimpleNamespace()

    def define(name, default, doc=""):
        setattr(flags_obj, name, flag_values.get(name, default))
    flags = types.SimpleNamespace(FLAGS=flags_obj, DEFINE_string=define, DEFINE_integer=define)
    tf.app = types.SimpleNamespace(flags=flags, run=lambda: sys.modules["__main__"].main())
    ema = types.SimpleNamespace(variables_to_restore=lambda: {})
    tf.train = types.SimpleNamespace(ExponentialMovingAverage=lambda d: ema,
                                     Saver=lambda v: types.SimpleNamespace(restore=lambda s, p: None))
    return tf


def install_scipy_misc():
    """scipy.misc.imread / imresize as scipy 1.1.0 had them (thin Pillow wrappers)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import scipy.misc as misc

    def imread(name):
        return np.array(Image.open(name))

    def imresize(arr, size, interp="bilinear"):
        assert interp == "bilinear"
        return np.array(Image.fromarray(arr).resize((size[1], size[0]), Image.BILINEAR))

    misc.imread, misc.imresize = imread, imresize


def write_images(folder, n, seed, gray_every=0, nested=True):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(folder, "sub"), exist_ok=True)
    for i in range(n):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        if gray_every and i % gray_every == 0:
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        sub = "sub" if (nested and i % 3 == 0) else ""
        Image.fromarray(a).save(os.path.join(folder, sub, f"{i:04d}.png"))
    open(os.path.join(folder, "notes.txt"), "w").write("not an image")


@contextlib.contextmanager
def patched(obj, **attrs):
    old = {k: getattr(obj, k, None) for k in attrs}
    for k, v in attrs.items():
        setattr(obj, k, v)
    try:
        yield
    finally:
        for k, v in old.items():
            setattr(obj, k, v)


@contextlib.contextmanager
def argv(*a):
    old = sys.argv
    sys.argv = list(a)
    try:
        yield
    finally:
        sys.argv = old


# ------------------------------------------------------------------------------------------------ coco
def run_coco(n, seed, out_name):
    script = os.path.join(REF, "image_realism", "IS", "coco", "inception_score_star_coco.py")
    net = SeededLogits(1008, seed, 0.9)
    record = {"feed_keys": set()}
    sys.modules["tensorflow"] = make_tensorflow(net, record, {})
    install_scipy_misc()
    import tarfile
    real_exists = os.path.exists
    with tempfile.TemporaryDirectory() as tmp:
        folder = os.path.join(tmp, "imgs")
        write_images(folder, n, seed, gray_every=7)
        saved = os.path.join(tmp, "out.txt")
        # _init_inception (:64-112) runs at import: keep it away from the network and from /tmp/imagenet
        with patched(os.path, exists=lambda p: True if str(p).startswith("/tmp/imagenet") else real_exists(p)), \
                patched(tarfile, open=lambda *a, **k: types.SimpleNamespace(extractall=lambda d: None)), \
                argv("inception_score_star_coco.py", "--image_folder", folder, "--saved_file", saved), \
                contextlib.redirect_stdout(io.StringIO()):
            g = runpy.run_path(script, run_name="__main__")
            text = open(saved).read()
            logits_main = np.concatenate(net.log, 0)
            net.log.clear()
            # the same functions again for the un-rounded floats
            images = g["load_data"](folder)
            mean, std = g["get_inception_score"](images)
    logits = np.concatenate(net.log, 0)
    assert np.array_equal(logits, logits_main) and logits.shape == (n, 1008)
    assert record["feed_keys"] == {"ExpandDims:0"}
    np.savez_compressed(os.path.join(HERE, out_name), logits=logits, temperature=record["temperature"],
                        mean=mean, std=std, expected_text=text, splits=10, rule="coco", drop_first=False)
    print(out_name, record["temperature"], mean, std, text)


# ------------------------------------------------------------------------------------------------ bird
def run_bird(n, batch, seed, out_name):
    script = os.path.join(REF, "image_realism", "IS", "bird", "inception_score_star_bird.py")
    net = SeededLogits(51, seed, 0.7)
    record = {"feed_keys": set()}
    with tempfile.TemporaryDirectory() as tmp:
        folder = os.path.join(tmp, "imgs")
        write_images(folder, n, seed)
        saved = os.path.join(tmp, "out.txt")
        sys.modules["tensorflow"] = make_tensorflow(net, record, {"image_folder": folder, "saved_file": saved,
                                                                  "batch_size": batch})
        install_scipy_misc()
        inception = types.ModuleType("inception")
        slim_pkg = types.ModuleType("inception.slim")
        slim = types.SimpleNamespace(
            arg_scope=lambda *a, **k: Ctx(), ops=types.SimpleNamespace(conv2d=None, fc=None),
            inception=types.SimpleNamespace(
                inception_v3=lambda images, **k: (Node("logits", net=net), {"aux_logits": None})))
        slim_pkg.slim = slim
        inception.slim = slim_pkg
        sys.modules["inception"], sys.modules["inception.slim"] = inception, slim_pkg
        np.random.seed(seed)
        order = list(np.arange(n)); np.random.shuffle(order)          # the shuffle the script is about to draw
        np.random.seed(seed)
        with argv("inception_score_star_bird.py"), contextlib.redirect_stdout(io.StringIO()):
            runpy.run_path(script, run_name="__main__")
        text = open(saved).read()
    logits = np.concatenate(net.log, 0)
    assert logits.shape == ((n // batch) * batch, 51)
    assert record["slice_begin"] == [0, 1] and record["slice_size"] == [batch, 50]
    mean, std = (float(t) for t in text.replace("IS = ", "").split("  +-  "))
    np.savez_compressed(os.path.join(HERE, out_name), logits=logits, temperature=record["temperature"],
                        mean=mean, std=std, expected_text=text, splits=10, rule="coco", drop_first=True,
                        batch_size=batch, n_files=n, shuffle=np.array(order))
    print(out_name, record["temperature"], text)


# ------------------------------------------------------------------------------------------------ O-IS
def run_ois(n, seed, out_name):
    script = os.path.join(REF, "object_fidelity", "O-IS", "object_centric_inception_score.py")
    net = SeededLogits(80, seed, 1.6)

    class Model:
        def __init__(self):
            self.AuxLogits = types.SimpleNamespace(fc=None)
            self.fc = None

        def load_state_dict(self, sd):
            pass

        def type(self, t):
            return self

        def eval(self):
            return self

        def __call__(self, x):
            return torch.from_numpy(net(x.numpy()))

    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    transforms = types.ModuleType("torchvision.transforms")
    models.inception_v3 = lambda pretrained, transform_input: Model()

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x
    transforms.Compose = Compose
    transforms.Resize = lambda size: (lambda img: img.resize((size[1], size[0]), Image.BILINEAR))
    transforms.ToTensor = lambda: (lambda img: torch.from_numpy(
        np.asarray(img, dtype=np.uint8).transpose(2, 0, 1).copy()).float().div(255))
    transforms.Normalize = lambda m, s: (lambda t: (t - torch.tensor(m).view(3, 1, 1)) / torch.tensor(s).view(3, 1, 1))
    tv.models, tv.transforms = models, transforms
    sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.transforms": transforms})
    with tempfile.TemporaryDirectory() as tmp:
        folder = os.path.join(tmp, "crops")
        write_images(folder, n, seed, nested=False)
        os.remove(os.path.join(folder, "notes.txt")); os.rmdir(os.path.join(folder, "sub"))   # os.listdir takes all
        saved = os.path.join(tmp, "out.txt")
        with patched(torch, load=lambda p: {}), \
                patched(torch.cuda, FloatTensor=torch.FloatTensor, set_device=lambda i: None), \
                argv("object_centric_inception_score.py", "--image_dir", folder, "--saved_file", saved), \
                contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            runpy.run_path(script, run_name="ref_ois")
        text = open(saved).read()
    logits = np.concatenate(net.log, 0)
    assert logits.shape == (n, 80)
    mean, std = (float(t) for t in text.replace("O-IS: ", "").split(" +-  "))
    np.savez_compressed(os.path.join(HERE, out_name), logits=logits, mean=mean, std=std, expected_text=text,
                        splits=10, rule="ois", drop_first=False, batch_size=32)
    print(out_name, text)


if __name__ == "__main__":
    run_coco(57, 11, "is_ref_coco_57.npz")          # 57: split borders i*57//10 are uneven
    run_coco(130, 12, "is_ref_coco_130.npz")
    run_bird(150, 64, 13, "is_ref_bird_150.npz")    # floor(150/64) = 2 batches: 22 files never scored
    run_ois(97, 14, "is_ref_ois_97.npz")            # 97 // 10 = 9 per split, 7 rows dropped
