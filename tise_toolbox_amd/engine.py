"""Device pipeline of the image-realism hot path: uint8 batch -> resize -> InceptionV3 -> statistics.

One ``RealismEngine`` per process / GPU.  A step takes a batch of decoded uint8 images that is
already in HBM and
  1. resizes it to 299x299 and applies ToTensor + the input affine      (csrc/resize.hip)
  2. runs the InceptionV3 trunk to pool3 (+ the fc head for IS*)        (csrc/conv_split.hip, conv_pipe.hip,
                                                                          trunk_ops.hip: hand-written split-fp16
                                                                          MFMA convolutions; fc GEMM via torch)
  3. folds the pool3 rows into fp64 {n, sum x, sum x x^T}               (csrc/stats.hip, fp64 MFMA)
  4. folds the logits into the per-split IS* sums                        (csrc/is_score.hip)
Nothing returns to the host until the end: ``reduce()`` all-reduces the sufficient statistics over
RCCL (world size > 1), ``statistics()`` finalises (mu, sigma) and the Frechet distance is evaluated
on the device (csrc/frechet.hip).

Reference path replaced: the loop of ``get_activations`` (image_realism/FID/fid_score.py:99-113:
batch.cuda(); model(batch); pred.cpu().numpy() per batch), ``np.mean``/``np.cov`` (:194-195),
``calculate_frechet_distance`` (:121-171) and the IS* loop
(image_realism/IS/coco/inception_score_star_coco.py:44-60).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib, device, dist as tdist
from .inception import InceptionV3, fc_bias_for_rule

# MIOpen's find mode also times its reference "naive" direct convolution (~120 ms per call at batch
# 500, ~40 s per process); it can never win, so keep it out of the search.
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")

T_COCO = 0.9091363549232483    # image_realism/IS/coco/inception_score_star_coco.py:107
T_BIRD = 0.5980541706085205    # image_realism/IS/bird/inception_score_star_bird.py:192
T_OIS = 2.1737587451934814     # object_fidelity/O-IS/object_centric_inception_score.py:55


def require_gpu():
    from .hostinfo import limit_torch_threads
    limit_torch_threads()                                   # 256 visible hardware threads, 16 CPUs of cgroup quota: see hostinfo
    _lib.load()
    if not torch.cuda.is_available():
        raise _lib.TiseLibraryError(
            "no HIP device visible: tise_toolbox_amd is the MI355X path of the TISE image-realism metrics and "
            "has no CPU fallback (the reference's --gpu '' CPU mode is not provided)")


def run_with_exact_fallback(fn, what="the evaluation"):
    """``fn()``; when the split-fp16 trunk's range guard fires (FloatingPointError from check_numerics: an activation beyond
    the fp16 range of the operand format -- weights / inputs far from InceptionV3's) the job is run AGAIN, in this process,
    on the exact-fp32 convolution path (``TISE_CONV=miopen``: a new engine, no exec) instead of telling the user to rerun
    by hand.  Under torchrun every rank raises together (check_numerics is collective), so every rank
    reruns.  ``--conv exact`` / TISE_CONV=miopen start there directly."""
    import sys
    try:
        return fn()
    except FloatingPointError as e:
        if os.environ.get("TISE_CONV", "split") != "split":
            raise
        print(f"[tise] {e}\n[tise] running {what} again on the exact-fp32 convolution path (TISE_CONV=miopen)", file=sys.stderr, flush=True)
        os.environ["TISE_CONV"] = "miopen"
        return fn()


class RealismEngine:
    def __init__(self, dims=2048, device_index=None, weights=None, num_classes=1000, seed=0,
                 channels_last=None, fold_bn=True, with_logits=False, model=None, normalize_input=True,
                 lut=None, fused=None, fc_bias="auto"):
        require_gpu()
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        # MIOpen find mode (time every applicable solver once per conv shape, keep the fastest);
        # TISE_MIOPEN_FIND=0 falls back to MIOpen's immediate-mode heuristic (faster start-up)
        torch.backends.cudnn.benchmark = os.environ.get("TISE_MIOPEN_FIND", "1") != "0"
        self.dims = dims
        self.with_logits = with_logits
        # classifier bias in the IS* logits: "auto" follows the rule given to begin() -- coco: NO bias, as
        # inception_score_star_coco.py:104-105 forms its logits from the weight matrix alone; bird / ois: bias
        self._fc_bias_mode = fc_bias
        self.fc_bias = fc_bias_for_rule("coco", fc_bias)
        if channels_last is None:
            channels_last = os.environ.get("TISE_CHANNELS_LAST", "1") != "0"
        self.channels_last = channels_last
        if model is None:
            block = InceptionV3.BLOCK_INDEX_BY_DIM[dims]
            model = InceptionV3([block], weights=weights, num_classes=num_classes, seed=seed,
                                normalize_input=normalize_input)
        from .inception import to_device_flat
        self.model = to_device_flat(model, self.device).eval()              # model.to(device), one copy per dtype
        # every rank must run the SAME parameters: the stand-in weights are calibrated with CPU convolutions
        # whose rounding depends on the host thread count, so rank 0's copy is broadcast (no-op for 1 process)
        tdist.broadcast_module_(self.model)
        if channels_last:
            self.model = self.model.to(memory_format=torch.channels_last)
        if fold_bn and hasattr(self.model, "fold_bn"):
            self.model.fold_bn(torch.channels_last if channels_last else torch.contiguous_format)
        if lut is None:
            lut = device.make_lut(normalize_input=getattr(self.model, "normalize_input", True))
        self.lut = lut
        # MIOpen convs + hand-written HIP epilogues (trunk.py); TISE_FUSED_TRUNK=0 runs the plain module graph
        self.fused = None
        if fused is None:
            fused = os.environ.get("TISE_FUSED_TRUNK", "1") != "0"
        if fused and channels_last and isinstance(self.model, InceptionV3):
            from .trunk import FusedTrunk, SplitTrunk
            # TISE_CONV=split (default, every --dims): hand-written split-precision fp16-MFMA convolutions;
            # TISE_CONV=miopen: MIOpen fp32 convolutions + HIP epilogues
            use_split = os.environ.get("TISE_CONV", "split") == "split"
            self.fused = (SplitTrunk if use_split else FusedTrunk)(self.model, self.device)
        self.stats = None
        self.is_acc = None
        # Optional hipGraph replay of resize + trunk (~85 launches per batch, all on one stream), TISE_GRAPH=1.
        # Measured no faster than eager launching here (25.7 vs 25.6 ms per 500-image step: the host keeps the queue
        # full and the GPU's launch-to-launch gap is the same either way), so it is off by default; it pays when the
        # host is the slower side (small batches).  Only for the all-HIP trunk (MIOpen picks solvers at run time).
        from .trunk import SplitTrunk as _ST
        self._graph_ok = isinstance(self.fused, _ST) and os.environ.get("TISE_GRAPH", "0") == "1"
        self._u8_stem = isinstance(self.fused, _ST) and os.environ.get("TISE_U8_STEM", "1") != "0"
        self.lut_dev = torch.from_numpy(np.ascontiguousarray(self.lut, dtype=np.float32).reshape(-1)).to(self.device)
        self._graphs = {}

    # ---- per-batch device work -----------------------------------------------------------------
    def _features_from_u8_eager(self, batch_u8):
        if self._u8_stem:
            # all-HIP trunk: the resize kernel writes the Pillow-exact uint8 image only and the stem convolution
            # applies the input table (134 MB instead of 536 MB of network input per 500 images; same features)
            return self._trunk_u8(device.resize_u8_only(batch_u8, (299, 299)))
        x = device.resize_bilinear_u8(batch_u8, (299, 299), self.lut, channels_last=self.channels_last)
        return self._trunk(x, prenormalized=True)

    def _trunk_u8(self, u8):
        pred = self.fused.forward_u8(u8, self.lut_dev)
        feats = pred.reshape(pred.shape[0], -1)
        return feats, self._logits(feats)

    def _logits(self, feats):
        """IS* logits of the pool3 rows just computed: the all-HIP trunk runs the classifier layer as a 1x1 split-precision
        convolution on them (SplitTrunk.fc_logits); the MIOpen trunk and user-supplied models use the module's own fc."""
        if not self.with_logits:
            return None
        if getattr(self.fused, "sfc", None) is not None and getattr(self.fused, "_feat_split", None) is not None \
                and self.fused._feat_split.shape[0] == feats.shape[0]:
            return self.fused.fc_logits(feats.shape[0], bias=self.fc_bias)
        return self.model.logits(feats, bias=self.fc_bias)

    @torch.no_grad()
    def features_from_u8(self, batch_u8):
        """(B,H,W,3) uint8 on the device -> pool3 (B,dims) fp32 [and logits (B,C)]."""
        if not self._graph_ok:
            return self._features_from_u8_eager(batch_u8)
        # the captured launches bake in the classifier-bias decision and whether logits are formed at all (begin(rule=...) flips
        # fc_bias: coco no bias, bird / ois bias): both are part of the key, or a coco graph would be replayed for an ois pass
        key = tuple(batch_u8.shape) + (bool(self.fc_bias), bool(self.with_logits))
        entry = self._graphs.get(key)
        if entry is None:
            # the first two batches of a shape run eagerly (library caches: resize plans, lookup table, hipBLASLt
            # workspace); the third is captured
            n = self._graphs.get(("seen", key), 0) + 1
            self._graphs[("seen", key)] = n
            if n <= 2:
                return self._features_from_u8_eager(batch_u8)
            try:
                static_in = batch_u8.clone()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    feats, logits = self._features_from_u8_eager(static_in)
                entry = (graph, static_in, feats, logits)
                self._graphs[key] = entry
            except Exception as e:                                    # capture not possible: stay eager, say so once
                print(f"[tise] hipGraph capture disabled: {type(e).__name__}: {e}", flush=True)
                self._graph_ok = False
                torch.cuda.synchronize()
                return self._features_from_u8_eager(batch_u8)
        graph, static_in, feats, logits = entry
        static_in.copy_(batch_u8)
        graph.replay()
        # the graph's output buffers are overwritten by the next replay: hand out copies (4 MB per 500 images)
        return feats.clone(), (logits.clone() if logits is not None else None)

    @torch.no_grad()
    def features_from_u8_list(self, crops):
        """Ragged batch: list of (H_i,W_i,3) uint8 tensors (object crops of different sizes, O-FID / O-IS) ->
        pool3 (B,dims) [and logits].  Each crop is resized (Pillow-exact) into its row of ONE (B,299,299,3) uint8
        buffer -- one small launch per crop, resize plans cached per size -- and the trunk runs ONCE on the batch."""
        if len(crops) == 0:
            raise ValueError("empty batch")
        u8 = torch.empty((len(crops), 299, 299, 3), dtype=torch.uint8, device=self.device)
        for i, c in enumerate(crops):
            if c.dim() == 4:
                c = c[0]
            device.resize_u8_only(c.to(self.device, non_blocking=True).unsqueeze(0), (299, 299), out=u8[i:i + 1])
        if self._u8_stem:
            return self._trunk_u8(u8)
        lut = self.lut_dev.view(3, 256)
        x = torch.stack([lut[c][u8[..., c].long()] for c in range(3)], dim=-1)              # byte -> input value, NHWC
        return self._trunk(x.permute(0, 3, 1, 2), prenormalized=True)                       # NCHW view, channels_last

    @torch.no_grad()
    def features_from_float(self, batch):
        """(B,3,H,W) fp32 in [0,1] (the reference's DataLoader output) -> pool3 (B,dims) fp32."""
        x = batch.to(self.device, non_blocking=True).float()
        if hasattr(self.model, "preprocess"):
            x = self.model.preprocess(x)
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        return self._trunk(x, prenormalized=True)

    def _trunk(self, x, prenormalized):
        if self.fused is not None and prenormalized:
            pred = self.fused(x)
        elif hasattr(self.model, "preprocess"):
            pred = self.model(x, prenormalized=prenormalized)[0]
        else:
            pred = self.model(x)[0]
        if pred.shape[2] != 1 or pred.shape[3] != 1:                 # fid_score.py:110-111
            pred = F.adaptive_avg_pool2d(pred, output_size=(1, 1))
        feats = pred.reshape(pred.shape[0], -1)
        if feats.stride(1) != 1:
            feats = feats.contiguous()
        used_fused = self.fused is not None and prenormalized
        return feats, (self._logits(feats) if used_fused else (self.model.logits(feats, bias=self.fc_bias) if self.with_logits else None))

    # ---- memory -----------------------------------------------------------------------------------
    ACT_BYTES_PER_IMAGE = int(14.5 * 2**20)     # peak of a trunk pass per 256 x 256 image (tools/activation_memory_probe.py: 14.4 MiB at 250-1000, 13.5 at 3000)

    def reserve_activations(self, images):
        """ONE allocation of the trunk's peak footprint for a pass of ``images`` images, released at once into torch's caching
        allocator -- the passes then split that block instead of asking the driver again and again.

        Round 6 (tools/cli_child_probe.py, profiles/r06z_cli_copy_trace.txt): the device batches of a fed image set GROW (50, 100,
        250 ... 3000 images), every larger pass needs larger blocks than the cache holds, and a fresh process ended with 54 GiB
        reserved for a 39.5 GiB peak.  That is harmless when the driver hands out clean VRAM (1 ms per call) and ruinous when it
        has to clear it first -- memory a previous process freed moments ago is cleared at ~45 GB/s: 0.97 s for 44 GB -- the
        README recipe then took 4.2-4.4 s instead of 3.1-3.3 with the main thread inside hipMalloc while the feed piled up (what
        round 6 first read as a "slow-copy mode").  With one allocation up front the worst case is one clearing of the peak.
        No-op for the MIOpen trunk, for a size already reserved, and when the request exceeds 60 % of the free memory."""
        from .trunk import SplitTrunk
        images = int(images)
        if images <= getattr(self, "_reserved_images", 0) or not isinstance(self.fused, SplitTrunk) or os.environ.get("TISE_RESERVE", "1") == "0":
            return 0
        nbytes = images * self.ACT_BYTES_PER_IMAGE
        free, _total = torch.cuda.mem_get_info(self.device)
        have = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)      # cached, free blocks
        if nbytes <= have or nbytes > 0.6 * free:
            self._reserved_images = images
            return 0
        x = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        del x
        self._reserved_images = images
        return nbytes

    # ---- accumulation ---------------------------------------------------------------------------
    def begin(self, n_total=None, temperature=T_COCO, splits=10, rule="coco", drop_first_class=False, fc_bias=None):
        """New image set.  ``rule`` also decides (unless ``fc_bias`` / the constructor's ``fc_bias`` force it) whether the
        classifier bias enters the logits of the following steps: inception.fc_bias_for_rule."""
        self.fc_bias = fc_bias_for_rule(rule, self._fc_bias_mode if fc_bias is None else fc_bias)
        self.stats = device.StatsAccumulator(self.dims, self.device)
        self.is_acc = None
        if self.with_logits:
            if n_total is None:
                raise ValueError("IS* needs the global image count (split membership is by global index)")
            c = self.model.fc.out_features
            self.is_acc = device.InceptionScoreAccumulator(c, n_total, temperature, splits, rule,
                                                           drop_first_class, self.device)

    def step_u8(self, batch_u8, idx_base=0):
        feats, logits = self.features_from_u8(batch_u8)
        self.stats.update(feats)
        if self.is_acc is not None:
            self.is_acc.update(logits, idx_base)
        return feats

    def step_u8_list(self, crops, idx_base=0):
        feats, logits = self.features_from_u8_list(crops)
        self.stats.update(feats)
        if self.is_acc is not None:
            self.is_acc.update(logits, idx_base)
        return feats

    def step_float(self, batch, idx_base=0):
        feats, logits = self.features_from_float(batch)
        self.stats.update(feats)
        if self.is_acc is not None:
            self.is_acc.update(logits, idx_base)
        return feats

    def reduce(self):
        """One all-reduce(SUM) per accumulator over RCCL (no-op for a single process)."""
        tdist.all_reduce_sum_(self.stats.buffer())
        if self.is_acc is not None:
            tdist.all_reduce_sum_(self.is_acc.acc)

    def check_numerics(self, collective=True):
        """Raise if the split-fp16 trunk met a value outside its range since the last check (device.check_split_overflow).
        ``collective``: under torchrun all ranks agree on the flag first (call it from every rank)."""
        from .trunk import SplitTrunk
        if isinstance(self.fused, SplitTrunk):
            flag = device.read_split_overflow()
            if collective and tdist.world_size() > 1:                   # every rank must raise, or the others hang in the next collective
                t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=self.device)
                flag = bool(tdist.all_reduce_sum_(t).item() > 0)
            device.check_split_overflow(flag=flag)

    def statistics(self):
        """(mu, sigma) fp64 CUDA tensors of everything accumulated (after reduce())."""
        self.check_numerics()
        return self.stats.finalize()

    def inception_score(self):
        self.check_numerics()
        mean, std, _ = self.is_acc.finalize()
        return mean, std


# ---- device batch: decoupled from the loader's --batch-size ------------------------------------------------
DEVICE_BATCH_DEFAULT = 5000      # images per trunk pass of a job whose pixels are RESIDENT (bench.py's `value`: a rank's share is cut into equal
                                 # batches of at most this).  Two alternating runs per size on one box, images/s of the driver's command:
                                 # round 4 (tools/batch_sweep_r04.sh) 1000 / 1500 / 2000 / 3000 -> 25.39 / 25.51 / 25.60 / 25.62 k;
                                 # round 6 (profiles/r06ad_device_batch_sweep.txt) 3000 / 3750 / 5000 / 6000 / 7500 -> 26.05 / 26.13 / 26.23 / 26.17 / 26.18 k
                                 # (the pooled-epilogue kernels walk whole images per workgroup, the tile tails of the 8 x 8 layers shrink,
                                 # 85 launch boundaries per pass); activations of a 5000-image pass: 71 GiB of the 288 GB
FEED_DEVICE_BATCH_DEFAULT = 1000 # images per trunk pass of a FED image set (the CLIs, bench.py's host_feed / png_feed / cli_process legs).
                                 # Round 5 used 3000 here too.  Round 6: a fresh process pays for its device memory -- the driver clears
                                 # VRAM that another process freed moments ago at ~45 GB/s when it hands it out again -- so the README
                                 # recipe met 0 .. 1 s of hipMalloc for the 3000-image footprint (reserve_activations) and 0 .. 0.3 s for
                                 # the 1000-image one (14.5 GiB), against 0.02-0.03 s more image loop per 30 000 images; and the feeds
                                 # lose nothing: png_feed 0.948 / 0.951 of resident at 1000 / 1500 (0.94 at 3000), host_feed 0.986 / 0.991
                                 # (0.98) -- finer pipelining of copies and passes (profiles/r06y_feed_device_batch.txt)
STAGING_BYTES_CAP = 1 << 30      # uint8 pixels of one staging buffer / one pending ragged batch


def device_batch_images(batch_size, image_bytes=256 * 256 * 3):
    """Images per trunk pass for a loader that delivers ``batch_size`` images at a time: whole loader batches up to
    TISE_DEVICE_BATCH images (default FEED_DEVICE_BATCH_DEFAULT = 1000: the trunk's launches -- 85 per pass -- and tile tails are
    amortised over 1000 images instead of the README recipe's 50, README.md:214-219) and at most 1 GiB of uint8 pixels.  The
    reference's ``--batch-size`` keeps its one semantic role, the drop-last rule (fid_score.py:90-96,215-217);
    features do not depend on how images are batched (tests/test_gpu_kernels.py: batch invariance, bit for bit)."""
    target = int(os.environ.get("TISE_DEVICE_BATCH", str(FEED_DEVICE_BATCH_DEFAULT)))
    cap = max(1, STAGING_BYTES_CAP // max(1, int(image_bytes)))
    target = max(1, min(target, cap))
    return max(1, target // max(1, int(batch_size))) * int(batch_size)


RAMP_HEAD = (64, 128, 256, 512, 1024, 1024, 1536, 2048)     # first device batches of a fed image set (images; rounded to loader batches)
RAMP_TAIL = (1024, 512)                                      # last ones: see item_schedule


def item_schedule(n_rows, batch_size, limit):
    """Device-batch sizes of an image set of ``n_rows`` images that arrives from a feed (PNG files, a DataLoader), in order;
    every entry is a whole number of loader batches, none exceeds ``limit``, the entries sum to n_rows.

    A fixed device batch of 3000 images makes the trunk wait for the first 3000 decoded images (~0.135 s of a 0.46 s job
    from 12 000 files) and, when decode is the slower side, start its last 3000 only after the last file.
    Measured in round 6 (tools/png_feed_probe.py, profiles/r06c_png_feed_timeline.txt): sixteen inflate-only workers deliver
    ~36 k images/s against the trunk's 25.7 k, so after a start at [500, 1000] the jump to 3000 still left the device idle
    for 22 ms, and the first 500 images cost 29 ms before anything ran.  So the set opens with a geometric ramp from ~64
    images (a starved device loses nothing on a small pass), grows by at most ~1.5 x per batch towards ``limit`` (what a
    decoder 1.3 x faster than the trunk can keep up with), and closes with two short batches (the device's last pass is
    what remains after the last file is decoded).  TISE_RAMP=0: equal batches.  Features are bit for bit independent of
    the batching (tests/test_gpu_kernels.py: batch invariance); the fp64 sums depend on it in the last bits, which is why
    the schedule is a pure function of (n_rows, batch_size, limit) and EVERY feed of the CLIs uses it: same files -> the
    same FID to the last bit."""
    bs = max(1, int(batch_size))
    limit = max(bs, (int(limit) // bs) * bs)
    n_rows = int(n_rows)
    if n_rows <= 0:
        return []

    def equal(n):                                           # n images (whole loader batches) in the fewest near-equal batches <= limit
        if n <= 0:
            return []
        nb, k = n // bs, -(-n // limit)
        q, r = divmod(nb, k)
        out = [(q + (1 if i < r else 0)) * bs for i in range(k)]
        if n % bs:
            out[-1] += n % bs                                # (callers pass whole batches; kept total for any n)
        return out

    if os.environ.get("TISE_RAMP", "1") == "0":
        return equal(n_rows)
    rnd = lambda r: max(bs, (r // bs) * bs)                 # noqa: E731
    tail = [rnd(r) for r in RAMP_TAIL if rnd(r) < limit]
    if n_rows < sum(tail) + rnd(RAMP_HEAD[0]) + bs:
        tail = []
    rem = n_rows - sum(tail)
    head = []
    for r in RAMP_HEAD:
        r = rnd(r)
        if r >= limit or rem - r < r:                       # the ramp has reached the full batch, or what is left is no bigger than the step
            break
        head.append(r)
        rem -= r
    return head + equal(rem) + tail


def coalesce_u8(batches, dev, limit, schedule=None):
    """Generator: consecutive equal-shaped uint8 (B, H, W, 3) batches of ``batches`` (host -- pinned or not -- or device
    tensors) gathered into device batches of up to ``limit`` images, in order; anything else (ragged crop lists, float
    tensors, a batch of another image size) passes through unchanged after what was gathered before it.
    Two staging buffers: the copies of the NEXT device batch run on a side stream while the trunk works on the current
    one; a buffer is refilled only after the consumer's stream has passed the point where it handed it back.
    ``schedule`` = (n_rows, batch_size) of the image set: the device batches follow item_schedule(n_rows, batch_size, limit)
    (computed once the image size -- and with it the clamped limit -- is known) instead of ``limit`` every time; it applies
    as long as dense batches of one shape arrive (anything else falls back to ``limit``)."""
    dev = torch.device(dev)
    sched_args, schedule = (tuple(schedule) if schedule else None), None
    sched_i = 0

    def cur_limit():
        if schedule is not None and sched_i < len(schedule):
            return min(limit, schedule[sched_i])
        return limit
    side = device.feed_stream(dev)                         # one high-priority stream per device
    bufs, freed, keep = [None, None], [None, None], []
    cur, fill = 0, 0
    limit0, shape0 = limit, None

    def flush():
        nonlocal cur, fill
        done = torch.cuda.Event()
        done.record(side)
        torch.cuda.current_stream(dev).wait_event(done)
        out = bufs[cur][:fill]
        return out

    def handed_back():
        nonlocal cur, fill, sched_i
        sched_i += 1
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        freed[cur] = ev
        cur ^= 1
        fill = 0
        keep.clear()                                          # (the side stream finished with the host batches: `done` was waited on)

    for b in batches:
        dense = isinstance(b, torch.Tensor) and b.dtype == torch.uint8 and b.dim() == 4 and b.shape[3] == 3
        if dense and tuple(b.shape[1:]) != shape0:
            # the callers size ``limit`` for 256 x 256 images; the staging buffers are sized from what actually arrives: whole
            # loader batches, at most STAGING_BYTES_CAP of pixels each (1024 x 1024 PNGs: 341 images, not 3000 x 3 MB twice)
            if fill:
                yield flush()
                handed_back()
            shape0 = tuple(b.shape[1:])
            per = max(1, int(b[0].numel()))
            nb = max(1, int(b.shape[0]))
            limit = min(limit0, max(nb, STAGING_BYTES_CAP // per // nb * nb))
            if sched_args is not None and sched_i == 0:
                schedule = item_schedule(sched_args[0], sched_args[1], limit)
            sched_args = None
        if not dense or b.shape[0] >= limit:
            if fill:
                yield flush()
                handed_back()
            schedule = None                                    # a foreign item: the schedule no longer describes what follows
            yield b
            continue
        if fill and (tuple(bufs[cur].shape[1:]) != tuple(b.shape[1:]) or fill + b.shape[0] > cur_limit()):
            yield flush()
            handed_back()
        if bufs[cur] is None or tuple(bufs[cur].shape[1:]) != tuple(b.shape[1:]):
            bufs[cur] = torch.empty((limit,) + tuple(b.shape[1:]), dtype=torch.uint8, device=dev)
            # the caching allocator hands out blocks in the order of the CONSUMER's stream: this one may be the memory of
            # activations that kernels already queued there still use -- the side stream must not write before them
            side.wait_stream(torch.cuda.current_stream(dev))
            bufs[cur].record_stream(side)                      # ... and the allocator must not recycle it under a copy still in flight there
        if fill == 0 and freed[cur] is not None:
            side.wait_event(freed[cur])
        if b.is_cuda:                                          # produced on the consumer's stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            side.wait_event(ev)
        with torch.cuda.stream(side):
            bufs[cur][fill:fill + b.shape[0]].copy_(b, non_blocking=True)
        keep.append(b)                                         # a pinned host batch must outlive its asynchronous copy
        fill += b.shape[0]
    if fill:
        yield flush()
        handed_back()
    side.synchronize()


def coalesce_batches(loader, dev, limit):
    """Device batches for crop directories: equal-sized uint8 batches are gathered by coalesce_u8; ragged batches
    (lists of crops of different sizes, img_data.collate_u8) are concatenated up to ``limit`` crops, so that the trunk
    runs once per ~1000 crops whatever --batch-size is (order preserved)."""
    pending, pending_bytes = [], 0

    def ragged(it):
        nonlocal pending, pending_bytes
        for b in it:
            if isinstance(b, (list, tuple)):
                nbytes = sum(int(c.numel()) for c in b)
                if pending and (len(pending) + len(b) > limit or pending_bytes + nbytes > STAGING_BYTES_CAP):
                    yield pending
                    pending, pending_bytes = [], 0
                pending = pending + list(b)
                pending_bytes += nbytes
            else:
                if pending:
                    yield pending
                    pending, pending_bytes = [], 0
                yield b
        if pending:
            yield pending
            pending, pending_bytes = [], 0
    return coalesce_u8(ragged(loader), dev, limit)


_SOLVERS = {}


def frechet_solver(dims, dev):
    key = (int(dims), str(dev))
    if key not in _SOLVERS:
        _SOLVERS[key] = device.FrechetSolver(dims, dev)
    return _SOLVERS[key]
