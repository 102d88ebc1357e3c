"""InceptionV3 feature extractor on PyTorch-ROCm (mirror of the reference wrapper).

Reference interface: ``image_realism/FID/inception.py:6-134`` (class ``InceptionV3``:
``BLOCK_INDEX_BY_DIM``, ``output_blocks``, ``resize_input``, ``normalize_input``,
``requires_grad``; ``forward(inp) -> list[Tensor]``).  The reference cuts its four
blocks out of ``torchvision.models.inception_v3(pretrained=True)`` (:57); torchvision
is not a dependency here, so ``Inception3`` below is an own definition of that
published topology whose ``state_dict`` keys and shapes are torchvision's, so the
reference's weight files (``inception_v3_google-1a9a5a14.pth``, and the 80-class
fine-tune of ``object_fidelity/O-FID/inception.py:58-64``) load unchanged.

``north_star`` assigns the conv stack to PyTorch-ROCm (MIOpen), fp32.  What is new
here for MI355X: eval-mode BatchNorm is folded into the conv (one kernel per
layer, no separate normalisation pass), the whole trunk runs channels-last so the
HIP resize kernel can emit the input layout directly, and the input affine of
``inception.py:120-124`` is fused into that resize kernel's lookup table.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv2d(nn.Module):
    """conv(bias=False) -> BatchNorm(eps=1e-3) -> ReLU; torchvision key names ``conv``/``bn``."""

    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        self._folded = None

    def forward(self, x):
        if self._folded is not None:
            w, b = self._folded
            return F.relu_(F.conv2d(x, w, b, self.conv.stride, self.conv.padding))
        return F.relu(self.bn(self.conv(x)), inplace=True)

    @torch.no_grad()
    def fold(self, memory_format=torch.contiguous_format):
        """Fold eval-mode BN into the conv: w' = w*g/sqrt(v+eps), b' = beta - mean*g/sqrt(v+eps)."""
        bn = self.bn
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (self.conv.weight * scale.view(-1, 1, 1, 1)).contiguous(memory_format=memory_format)
        b = (bn.bias - bn.running_mean * scale).contiguous()
        self._folded = (w, b)

    def unfold(self):
        self._folded = None


class InceptionA(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch5x5_1 = BasicConv2d(cin, 48, kernel_size=1)
        self.branch5x5_2 = BasicConv2d(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, padding=1)
        self.branch_pool = BasicConv2d(cin, pool_features, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b5 = self.branch5x5_2(self.branch5x5_1(x))
        b3 = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b5, b3, bp], 1)


class InceptionB(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = BasicConv2d(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3(x)
        bd = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([b3, bd, bp], 1)


class InceptionC(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7_2 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_1(x)
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(bd))))
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b7, bd, bp], 1)


class InceptionD(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch3x3_2 = BasicConv2d(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        bp = F.max_pool2d(x, kernel_size=3, stride=2)
        return torch.cat([b3, b7, bp], 1)


class InceptionE(nn.Module):
    def __init__(self, cin):
        super().__init__()
        self.branch1x1 = BasicConv2d(cin, 320, kernel_size=1)
        self.branch3x3_1 = BasicConv2d(cin, 384, kernel_size=1)
        self.branch3x3_2a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b3 = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(b3), self.branch3x3_2b(b3)], 1)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(bd), self.branch3x3dbl_3b(bd)], 1)
        bp = self.branch_pool(F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
        return torch.cat([b1, b3, bd, bp], 1)


class InceptionAux(nn.Module):
    """Present only so torchvision state_dicts load with strict=True; never run."""

    def __init__(self, cin, num_classes):
        super().__init__()
        self.conv0 = BasicConv2d(cin, 128, kernel_size=1)
        self.conv1 = BasicConv2d(128, 768, kernel_size=5)
        self.fc = nn.Linear(768, num_classes)


class Inception3(nn.Module):
    """The torchvision ``Inception3`` module tree (names, shapes) without its forward."""

    def __init__(self, num_classes=1000, aux_logits=True):
        super().__init__()
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, kernel_size=3)
        self.Mixed_5b = InceptionA(192, pool_features=32)
        self.Mixed_5c = InceptionA(256, pool_features=64)
        self.Mixed_5d = InceptionA(288, pool_features=64)
        self.Mixed_6a = InceptionB(288)
        self.Mixed_6b = InceptionC(768, c7=128)
        self.Mixed_6c = InceptionC(768, c7=160)
        self.Mixed_6d = InceptionC(768, c7=160)
        self.Mixed_6e = InceptionC(768, c7=192)
        if aux_logits:
            self.AuxLogits = InceptionAux(768, num_classes)
        self.Mixed_7a = InceptionD(768)
        self.Mixed_7b = InceptionE(1280)
        self.Mixed_7c = InceptionE(2048)
        self.fc = nn.Linear(2048, num_classes)


def _trunk_forward(net, x):
    """pool3 features of an ``Inception3`` module tree (used only to calibrate stand-in weights)."""
    x = net.Conv2d_2b_3x3(net.Conv2d_2a_3x3(net.Conv2d_1a_3x3(x)))
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    x = net.Conv2d_4a_3x3(net.Conv2d_3b_1x1(x))
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e",
                 "Mixed_7a", "Mixed_7b", "Mixed_7c"):
        x = getattr(net, name)(x)
    return F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)


_SEEDED_CACHE = {}
# white-noise fraction of each image of the BatchNorm calibration batch (seeded_init_)
CALIBRATION_NOISE_FRACTIONS = (0.0, 0.0, 0.0, 0.03, 0.05, 0.05, 0.1, 0.2, 0.35, 0.5, 0.75, 1.0)


@torch.no_grad()
def seeded_init_(net, seed=0, calibration="fid"):
    """Deterministic stand-in weights (no pretrained file exists offline).

    He-normal conv weights, then a data-dependent BatchNorm calibration: one train-mode pass over a
    fixed synthetic batch sets every BN's running mean/variance to the statistics of its own input,
    as in a trained network.  Without it a 94-layer random ReLU stack maps every image to almost
    the same feature vector (FID ~ 0, IS ~ 1) and the parity tests would test nothing; with it the
    features are image dependent.  The classifier bias is centred on the calibration batch so the
    logits are image dependent too.  Everything is drawn and computed on the CPU from one
    generator, so CPU and GPU runs see identical parameters.  Throughput is weight-independent;
    scores obtained with these weights are only comparable between paths run on them.
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        # data-parallel run: rank 0 calibrates and the engine broadcasts its parameters to every rank
        # (engine.RealismEngine -> dist.broadcast_module_), so the other ranks skip the CPU convolutions
        return net
    key = (seed, net.fc.out_features, calibration)
    if key in _SEEDED_CACHE:
        # (the module tree may be a skeleton without storage -- build_inception3 -- so the tensors are ASSIGNED; clones, because
        # the in-memory copy serves every later model of the process)
        net.load_state_dict({k: v.clone() for k, v in _SEEDED_CACHE[key].items()}, assign=True)
        return net
    cached = _standin_cache_load(key)
    if cached is not None:
        _SEEDED_CACHE[key] = cached
        net.load_state_dict({k: v.clone() for k, v in cached.items()}, assign=True)
        return net
    _materialize_(net)
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name, m in net.named_modules():
        if isinstance(m, nn.Conv2d):
            fan_in = m.in_channels * m.kernel_size[0] * m.kernel_size[1]
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
            m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
            m.running_mean.zero_()
            m.running_var.fill_(1.0)
        elif isinstance(m, nn.Linear):
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.3)
            m.bias.zero_()
    # calibration batch: smooth random fields in [0,1], already through the inception.py:120-124 affine
    # ... blended with white noise at fractions from 0 (smooth, photo-like) to 1 (i.i.d. uniform pixels, BASELINE
    # configs[0]'s images): a stack calibrated on smooth fields alone amplifies white noise layer by layer (pool3 mean
    # 51 instead of 0.25, FID ~1.3e3 for configs[0]: outside the published range the |dFID| <= 1e-3 budget is meant for)
    alpha = torch.tensor(CALIBRATION_NOISE_FRACTIONS).view(-1, 1, 1, 1)
    n_cal = alpha.shape[0]
    x = torch.rand((n_cal, 3, 10, 10), generator=g)
    x = F.interpolate(x, size=(299, 299), mode="bicubic", align_corners=False).clamp_(0.0, 1.0)
    x = (1.0 - alpha) * x + alpha * torch.rand((n_cal, 3, 299, 299), generator=g)
    if calibration == "pm1":             # O-IS convention: Normalize((.5,.5,.5),(.5,.5,.5)) -> [-1, 1]
        x = (x - 0.5) / 0.5
    else:                                # FID wrapper convention: inception.py:120-124 on [0, 1] pixels
        x[:, 0] = x[:, 0] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
        x[:, 1] = x[:, 1] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
        x[:, 2] = x[:, 2] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
    bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0                     # running stats := statistics of this one batch
    was_training = net.training
    # ONE host thread for the two calibration passes: the rounding of oneDNN's convolutions depends on how the work is
    # split over threads, and torchrun sets OMP_NUM_THREADS=1 -- without this a 1-process and an N-process run of the
    # same job would get stand-in weights that differ in the last bits (FID differing by ~3e-6)
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    net.train()
    _trunk_forward(net, x)
    # bring pool3 to the magnitude of real InceptionV3 features (mean ~0.25) so that FID values -- and
    # with them the absolute |dFID| <= 1e-3 budget -- sit in the range the reference publishes (2..200)
    for name in ("branch1x1", "branch3x3_2a", "branch3x3_2b", "branch3x3dbl_3a", "branch3x3dbl_3b", "branch_pool"):
        bn = getattr(net.Mixed_7c, name).bn
        bn.weight.mul_(0.7)
        bn.bias.mul_(0.7)
    net.eval()
    feats = _trunk_forward(net, x)
    for m, mom in zip(bns, old):
        m.momentum = mom
    net.fc.bias.copy_(-(net.fc.weight @ feats.mean(0)))
    torch.set_num_threads(n_threads)
    if was_training:
        net.train()
    _SEEDED_CACHE[key] = {k: v.clone() for k, v in net.state_dict().items()}
    _standin_cache_store(key, _SEEDED_CACHE[key])
    return net


# ---- disk cache of the calibrated stand-ins -------------------------------------------------------------------------
# The two one-thread calibration passes cost ~50 s per process on the GPU boxes' hosts (a rank of a multi-process test, every
# bench run, every CLI invocation with --synthetic-weights paid them).  The result is a pure function of (seed, classes,
# calibration convention) and of the code above, so it is kept as a file: $TISE_STANDIN_CACHE, else a per-user 0700
# directory under the temp dir.  The file holds exactly the state_dict the calibration produced -- loading it gives the
# same bits -- and is written to a temporary name and renamed into place.  TISE_STANDIN_CACHE=off disables it.
_STANDIN_VERSION = 3


def _standin_cache_path(key):
    import hashlib
    import os
    import tempfile
    root = os.environ.get("TISE_STANDIN_CACHE")
    if root == "off":
        return None
    if not root:
        uid = os.getuid() if hasattr(os, "getuid") else 0
        root = os.path.join(tempfile.gettempdir(), f"tise_toolbox_amd_standin_{uid}")
    tag = hashlib.sha256(repr((_STANDIN_VERSION, key, CALIBRATION_NOISE_FRACTIONS, torch.__version__)).encode()).hexdigest()[:20]
    return os.path.join(root, f"inception3_standin_{key[0]}_{key[1]}_{key[2]}_{tag}.pt")


def _standin_cache_load(key):
    import os
    path = _standin_cache_path(key)
    if path is None or not os.path.exists(path):
        return None
    try:
        if hasattr(os, "getuid") and os.stat(path).st_uid != os.getuid():
            return None
        sd = torch.load(path, map_location="cpu", weights_only=True)
        return sd if isinstance(sd, dict) and "fc.bias" in sd else None
    except Exception:                                                     # a torn or foreign file: calibrate again
        return None


def _standin_cache_store(key, sd):
    import os
    path = _standin_cache_path(key)
    if path is None:
        return
    try:
        os.makedirs(os.path.dirname(path), mode=0o700, exist_ok=True)
        tmp = f"{path}.{os.getpid()}.tmp"
        torch.save(sd, tmp)
        os.replace(tmp, path)
    except Exception:                                                     # read-only temp dir: no cache, no error
        pass


import contextlib  # noqa: E402


@contextlib.contextmanager
def _no_default_init():
    """Module constructors without their default parameter initialisation: on the ``meta`` device every ``kaiming_uniform_``
    still runs through torch's Python decompositions (0.17 s for the 96 convolutions, profiles/r06d_startup.txt) to
    initialise storage that does not exist."""
    saved = [(cls, cls.reset_parameters) for cls in (nn.Conv2d, nn.Linear, nn.BatchNorm2d)]
    try:
        for cls, _ in saved:
            cls.reset_parameters = lambda self: None
        yield
    finally:
        for cls, fn in saved:
            cls.reset_parameters = fn


def _materialize_(net):
    """Give a skeleton built on the ``meta`` device real (uninitialised) CPU storage; counters start at zero."""
    if any(p.is_meta for p in net.parameters()):
        net.to_empty(device="cpu")
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None:
                m.num_batches_tracked.zero_()
    return net


def build_inception3(weights=None, num_classes=1000, seed=0, calibration="fid"):
    """Construct ``Inception3`` and load ``weights`` (a torchvision-format state_dict
    path) or, when ``weights`` is None, the seeded stand-in parameters.

    The module tree is built WITHOUT storage (``meta`` device) and the loaded tensors are assigned to it
    (``load_state_dict(assign=True)``): torch's default initialisation of 96 convolutions (kaiming_uniform_ over 24 M
    values) and the copy of every tensor into it were 0.11 s of a CLI process's start-up for values that are overwritten
    at once (tools/startup_probe.py, profiles/r06d_startup.txt).  Only when stand-in weights have to be computed from
    scratch (no cache file yet) does the skeleton get real storage first."""
    with torch.device("meta"), _no_default_init():
        net = Inception3(num_classes=num_classes, aux_logits=True)
    if weights is not None:
        sd = torch.load(weights, map_location="cpu")
        if isinstance(sd, dict) and "state_dict" in sd:
            sd = sd["state_dict"]
        net.load_state_dict(sd, strict=True, assign=True)
    else:
        seeded_init_(net, seed, calibration)
        if any(p.is_meta for p in net.parameters()):        # a rank other than 0 of a multi-process run: rank 0 broadcasts
            _materialize_(net)
            for prm in net.parameters():
                prm.detach().zero_()
    return net.eval()


class InceptionV3(nn.Module):
    """Drop-in for the reference wrapper (``image_realism/FID/inception.py:6``).

    Extra keyword arguments (all optional, defaults reproduce the reference call
    ``InceptionV3([block_idx])``): ``weights`` (state_dict path; the reference
    downloads the torchvision file, there is no network here), ``num_classes``
    (80 for the O-FID/O-IS fine-tune, ``O-FID/inception.py:58-64``), ``seed``.
    """

    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}   # inception.py:14-19

    def __init__(self, output_blocks=[DEFAULT_BLOCK_INDEX], resize_input=True, normalize_input=True,
                 requires_grad=False, weights=None, num_classes=1000, seed=0, calibration="fid"):
        super().__init__()
        self.resize_input = resize_input
        self.normalize_input = normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, "Last possible output block index is 3"   # inception.py:53

        inception = build_inception3(weights, num_classes, seed, calibration)
        self.blocks = nn.ModuleList()
        self.blocks.append(nn.Sequential(                                   # inception.py:59-66
            inception.Conv2d_1a_3x3, inception.Conv2d_2a_3x3, inception.Conv2d_2b_3x3,
            nn.MaxPool2d(kernel_size=3, stride=2)))
        if self.last_needed_block >= 1:                                     # :68-71
            self.blocks.append(nn.Sequential(
                inception.Conv2d_3b_1x1, inception.Conv2d_4a_3x3, nn.MaxPool2d(kernel_size=3, stride=2)))
        if self.last_needed_block >= 2:                                     # :73-85
            self.blocks.append(nn.Sequential(
                inception.Mixed_5b, inception.Mixed_5c, inception.Mixed_5d, inception.Mixed_6a,
                inception.Mixed_6b, inception.Mixed_6c, inception.Mixed_6d, inception.Mixed_6e))
        if self.last_needed_block >= 3:                                     # :87-95
            self.blocks.append(nn.Sequential(
                inception.Mixed_7a, inception.Mixed_7b, inception.Mixed_7c,
                nn.AdaptiveAvgPool2d(output_size=(1, 1))))
        # classifier head: not part of the reference FID wrapper; IS* needs the logits
        # (north_star: PyTorch InceptionV3 logits with the IS* temperature).
        self.fc = inception.fc
        for p in self.parameters():                                         # :97-98
            p.requires_grad = requires_grad
        self._input_prenormalized = False

    def fold_bn(self, memory_format=torch.contiguous_format):
        for m in self.modules():
            if isinstance(m, BasicConv2d):
                m.fold(memory_format)
        return self

    def unfold_bn(self):
        for m in self.modules():
            if isinstance(m, BasicConv2d):
                m.unfold()
        return self

    def forward(self, inp, prenormalized=False):
        """inception.py:100-134.  ``prenormalized=True`` is the fused device path: the
        input is already 299x299 with the :120-124 affine applied by the resize kernel."""
        outp = []
        x = inp if prenormalized else self.preprocess(inp)
        for idx, block in enumerate(self.blocks):                           # :126-132
            x = block(x)
            if idx in self.output_blocks:
                outp.append(x)
            if idx == self.last_needed_block:
                break
        return outp

    def preprocess(self, x):
        """inception.py:117-124: optional align_corners bilinear resize to 299x299, then the input affine."""
        if self.resize_input and tuple(x.shape[-2:]) != (299, 299):
            # :117-118; for 299x299 input align_corners bilinear is the identity map
            x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=True)
        if self.normalize_input:                                            # :120-124
            x = x.clone()
            x[:, 0] = x[:, 0] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
            x[:, 1] = x[:, 1] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
            x[:, 2] = x[:, 2] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
        return x

    def logits(self, pool3, bias=True):
        """fc head on pool3 features (B,2048[,1,1]) -> (B, num_classes).

        ``bias=False`` is the head of the reference IS* for COCO: image_realism/IS/coco/inception_score_star_coco.py:104-105
        takes ONLY the weight matrix of the graph's last layer (``w = ...get_operation_by_name("softmax/logits/MatMul").inputs[1];
        logits = tf.matmul(tf.squeeze(pool3, [1, 2]), w)``) -- the graph's BiasAdd is deliberately left out.  The O-IS
        script runs the whole torch model (object_centric_inception_score.py:41-60) and the bird script the slim
        ``logits`` end point (inception_score_star_bird.py:189): both WITH the bias.  ``fc_bias_for_rule`` maps the rules."""
        x = pool3.flatten(1)
        return self.fc(x) if bias else F.linear(x, self.fc.weight)


@torch.no_grad()
def to_device_flat(module, device):
    """``module.to(device)`` with ONE host->device copy per dtype instead of one per tensor: the 566 parameters and buffers
    of InceptionV3 are gathered into a flat host buffer, copied once, and every parameter / buffer becomes a view of the
    device buffer (0.135 s -> ~0.02 s of a CLI's start-up; profiles/r06d_startup.txt).  Values, names, dtypes and
    state_dict() are what ``.to(device)`` gives.  Tensors already on the device are left alone."""
    device = torch.device(device)
    groups = {}
    for mod in module.modules():
        for store in (mod._parameters, mod._buffers):
            for name, t in store.items():
                if t is not None and t.device != device and not t.is_meta:
                    groups.setdefault(t.dtype, []).append((store, name, t))
    for dtype, items in groups.items():
        # distinct tensors only (a parameter shared by two modules must stay shared)
        uniq, seen = [], {}
        for store, name, t in items:
            if id(t) not in seen:
                seen[id(t)] = len(uniq)
                uniq.append(t)
        sizes = [t.numel() for t in uniq]
        flat = torch.empty(sum(sizes), dtype=dtype)
        off = 0
        for t, n in zip(uniq, sizes):
            flat[off:off + n] = t.detach().reshape(-1)
            off += n
        dflat = flat.to(device)
        views, off = [], 0
        for t, n in zip(uniq, sizes):
            views.append(dflat[off:off + n].view(t.shape))
            off += n
        for store, name, t in items:
            v = views[seen[id(t)]]
            if isinstance(t, nn.Parameter):
                store[name] = nn.Parameter(v, requires_grad=t.requires_grad)
            else:
                store[name] = v
    return module


def fc_bias_for_rule(rule, fc_bias="auto"):
    """Whether the classifier bias enters the IS* logits: ``fc_bias`` "on" / "off" force it, "auto" follows the
    reference script of the rule -- coco: no bias (inception_score_star_coco.py:104-105), bird / ois: bias
    (inception_score_star_bird.py:189, object_centric_inception_score.py:41-60)."""
    if isinstance(fc_bias, bool):
        return fc_bias
    if fc_bias in ("on", "off"):
        return fc_bias == "on"
    if fc_bias not in (None, "auto"):
        raise ValueError(f"fc_bias must be auto, on or off, not {fc_bias!r}")
    return rule != "coco"
