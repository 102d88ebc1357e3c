#!/usr/bin/env python3
"""FID on MI355X -- drop-in for the reference ``image_realism/FID/fid_score.py``.

Same module-level functions (names, argument meaning, return types, error behaviour) and the
same CLI flags / result-file text as the reference, so a caller switches by changing the import:

    get_activations                 fid_score.py:67      calculate_activation_statistics  :174
    calculate_frechet_distance      :121                 _compute_statistics_of_path      :199
    calculate_fid_given_paths       :223                 CLI                              :51-64,:241-254

What runs where: PNG decode on DataLoader workers (as the reference, :215-217); resize + input
affine, InceptionV3, fp64 mean/covariance accumulation and the Frechet distance on the GPU
(``engine.py`` -> ``libtise_hip.so`` + PyTorch-ROCm).  With ``torchrun --nproc-per-node N`` the file
list is sharded over N GPUs and the sufficient statistics are combined with one RCCL all-reduce.

Deviations from the reference, all deliberate (SURVEY.md notes N3-N5):
  * ``--gpu ""`` (CPU mode) is refused: this package is the MI355X path and has no CPU fallback.
  * ``--path1/--path2`` are required (upstream defaults them to the int 64 by a copy-paste slip).
  * extra flags: ``--weights`` (torchvision-format state_dict; there is no network to download
    it), ``--seed`` (stand-in weights when no file is given), ``--save-stats`` (write mu/sigma .npz,
    the format upstream only reads), ``--label`` ("FID" | "O-FID", object_fidelity/O-FID/fid_score.py:216-222),
    ``--num-classes`` (80 for the O-FID fine-tune, O-FID/inception.py:58-64).
"""
import os
import sys
import warnings
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

import numpy as np
import torch
import torch.utils.data

from . import _lib, device, dist as tdist, img_data
from .engine import RealismEngine, frechet_solver, require_gpu
from .inception import InceptionV3

warnings.filterwarnings("ignore")          # fid_score.py:49


def _build_parser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter)
    parser.add_argument("--batch-size", type=int, default=64, help="Batch size to use")
    parser.add_argument("--dims", type=int, default=2048, choices=list(InceptionV3.BLOCK_INDEX_BY_DIM),
                        help=("Dimensionality of Inception features to use. " "By default, uses pool3 features"))
    parser.add_argument("-c", "--gpu", default="0", type=str, help="GPU to use (CPU mode is not provided)")
    parser.add_argument("--path1", type=str, required=True)
    parser.add_argument("--path2", type=str, required=True)
    parser.add_argument("--saved_file", type=str, default="")
    parser.add_argument("--weights", type=str, default=None, help="torchvision-format InceptionV3 state_dict (.pth)")
    parser.add_argument("--num-classes", type=int, default=1000)
    parser.add_argument("--seed", type=int, default=0, help="seed of the stand-in weights when --weights is absent")
    parser.add_argument("--save-stats", type=str, default="", help="write mu/sigma of --path2 to this .npz")
    parser.add_argument("--label", type=str, default="FID", choices=["FID", "O-FID"])
    parser.add_argument("--num-workers", type=int, default=min(32, os.cpu_count() or 8),
                        help="PNG-decoding DataLoader workers (the reference hard-codes 8, fid_score.py:216)")
    return parser


def _engine_for(model, dims):
    """Wrap a user-supplied reference-style model, or build the default one."""
    if isinstance(model, RealismEngine):
        return model
    eng = getattr(model, "_tise_engine", None)
    if eng is None:
        eng = RealismEngine(dims=dims, model=model, fold_bn=isinstance(model, InceptionV3))
        try:
            model._tise_engine = eng
        except Exception:
            pass
    return eng


def _forward_batch(engine, model, batch):
    """One batch -> (B, dims) fp32 features on the device, for either input convention."""
    if isinstance(batch, (list, tuple)):                 # ragged uint8 crops: one resize launch per image
        feats = [engine.features_from_u8(b.unsqueeze(0).to(engine.device, non_blocking=True))[0] for b in batch]
        return torch.cat(feats, 0)
    if batch.dtype == torch.uint8:                       # (B,H,W,3) decoded images: fused device resize
        return engine.features_from_u8(batch.to(engine.device, non_blocking=True))[0]
    if isinstance(model, InceptionV3):
        return engine.features_from_float(batch)[0]
    # arbitrary nn.Module following the reference contract model(batch)[0] -> (B, dims, h, w)
    with torch.no_grad():
        pred = model(batch.to(engine.device))[0]
        if pred.shape[2] != 1 or pred.shape[3] != 1:     # fid_score.py:110-111
            pred = torch.nn.functional.adaptive_avg_pool2d(pred, output_size=(1, 1))
        return pred.reshape(pred.shape[0], -1).float().contiguous()


def _check_cuda(cuda):
    if not cuda:
        raise _lib.TiseLibraryError(
            "cuda=False / --gpu '' requests the reference's CPU path; tise_toolbox_amd is MI355X-only "
            "and has no CPU fallback")
    require_gpu()


def get_activations(images, model, batch_size=64, dims=2048, cuda=True, verbose=True):
    """Activations of the pool_3 layer for all images (fid_score.py:67-118).

    ``images``: sized iterable of batches (``len`` = number of batches, :90) of either
    float (B,3,H,W) tensors in [0,1] (reference convention) or uint8 (B,H,W,3) tensors.
    Returns a float64 numpy array (n_used, dims) -- one device->host copy at the end instead of
    one per batch (:113).
    """
    _check_cuda(cuda)
    model.eval()                                          # :86
    d0 = images.__len__() * batch_size                    # :90
    if batch_size > d0:                                   # :91-93
        print(("Warning: batch size is bigger than the data size. " "Setting batch size to data size"))
        batch_size = d0
    n_batches = d0 // batch_size                          # :95  (ZeroDivisionError for an empty loader, as upstream)
    n_used_imgs = n_batches * batch_size                  # :96
    engine = _engine_for(model, dims)
    pred_dev = torch.empty((n_used_imgs, dims), dtype=torch.float32, device=engine.device)
    for i, batch in enumerate(images):                    # :99
        start = i * batch_size
        end = start + batch_size
        pred_dev[start:end] = _forward_batch(engine, model, batch).reshape(batch_size, -1)   # :113
    if verbose:
        print(" done")                                    # :116
    return pred_dev.cpu().numpy().astype(np.float64)      # :98 pred_arr is float64


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """Frechet distance d^2 = ||mu_1 - mu_2||^2 + Tr(C_1 + C_2 - 2*sqrt(C_1*C_2))  (fid_score.py:121-171).

    Accepts numpy arrays or CUDA tensors; evaluated on the device in fp64 (csrc/frechet.hip).
    Returns np.float64 like the reference.  The reference's rescue branch for a singular
    product (:156-160) is kept: when the device reports non-finite values the message is printed
    and the computation repeated with ``eps`` added to both diagonals.  The complex-residue
    ``ValueError`` (:163-167) cannot occur here: eigenvalues of a symmetric matrix are real.
    """
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())

    def prep(x, nd):
        if isinstance(x, torch.Tensor):
            x = x.to(dev, torch.float64)
            return torch.atleast_1d(x) if nd == 1 else torch.atleast_2d(x)
        x = np.atleast_1d(x) if nd == 1 else np.atleast_2d(x)             # :143-147
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)

    mu1, mu2 = prep(mu1, 1), prep(mu2, 1)
    sigma1, sigma2 = prep(sigma1, 2), prep(sigma2, 2)
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"            # :149
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"    # :150
    d = mu1.shape[0]
    solver = frechet_solver(d, dev)
    res = solver.distance(mu1, sigma1, mu2, sigma2, 0.0)
    if res["flags"] & _lib.TISE_FLAG_NONFINITE:                                                        # :156-160
        msg = ("fid calculation produces singular product; " "adding %s to diagonal of cov estimates") % eps
        print(msg)
        res = solver.distance(mu1, sigma1, mu2, sigma2, float(eps))
    calculate_frechet_distance.last_result = res
    return np.float64(res["fid"])


def calculate_activation_statistics(images, model, batch_size=64, dims=2048, cuda=True, verbose=True,
                                    return_device=False):
    """mu = mean(act), sigma = cov(act) of the pool_3 activations (fid_score.py:174-196).

    Unlike the reference no (N, dims) float64 array is ever materialised: every batch is folded
    into fp64 {n, sum x, sum x x^T} on the device and (mu, sigma) are finalised there.  Under
    torchrun each rank passes ITS shard of the batches; the sums are all-reduced over RCCL.
    Returns numpy float64 arrays (or CUDA tensors with ``return_device``).
    """
    _check_cuda(cuda)
    model.eval()
    d0 = images.__len__() * batch_size                    # fid_score.py:90-96 bookkeeping
    if batch_size > d0:
        print(("Warning: batch size is bigger than the data size. " "Setting batch size to data size"))
        batch_size = d0
    n_batches = d0 // batch_size
    engine = _engine_for(model, dims)
    stats = device.StatsAccumulator(dims, engine.device)
    for i, batch in enumerate(images):
        if i >= n_batches:
            break
        stats.update(_forward_batch(engine, model, batch))
    tdist.all_reduce_sum_(stats.buffer())
    mu, sigma = stats.finalize()
    if verbose:
        print(" done")
    if return_device:
        return mu, sigma
    return mu.cpu().numpy(), sigma.cpu().numpy()


def _compute_statistics_of_path(path, model, batch_size, dims, cuda, num_workers=8):
    """fid_score.py:199-220: an .npz holds (mu, sigma); a directory is walked and pushed through the net."""
    if path.endswith(".npz"):
        f = np.load(path, allow_pickle=True)              # :201-203
        m, s = f["mu"][:], f["sigma"][:]
        f.close()
        return m, s
    files = img_data.get_filenames(path)                  # os.walk order (img_data.py:27-35)
    rank, world, _ = tdist.env_world()
    shard, _ = tdist.shard_files(files, batch_size, rank, world)       # drop_last=True (:215-217), whole batches
    dataset = img_data.Dataset(path, transform=None, file_names=shard)
    dataloader = torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False, drop_last=True,
                                             num_workers=num_workers, collate_fn=img_data.collate_u8,
                                             pin_memory=True)
    return calculate_activation_statistics(dataloader, model, batch_size, dims, cuda)


def calculate_fid_given_paths(paths, batch_size, cuda, dims, weights=None, num_classes=1000, seed=0,
                              save_stats="", num_workers=8):
    """Calculates the FID of two paths (fid_score.py:223-238)."""
    for p in paths:
        if not os.path.exists(p):
            raise RuntimeError("Invalid path: %s" % p)    # :225-227
    _check_cuda(cuda)
    block_idx = InceptionV3.BLOCK_INDEX_BY_DIM[dims]
    model = InceptionV3([block_idx], weights=weights, num_classes=num_classes, seed=seed)
    model.cuda()                                          # :232-233
    m1, s1 = _compute_statistics_of_path(paths[0], model, batch_size, dims, cuda, num_workers)
    m2, s2 = _compute_statistics_of_path(paths[1], model, batch_size, dims, cuda, num_workers)
    if save_stats and tdist.is_main():
        np.savez(save_stats, mu=np.asarray(m2), sigma=np.asarray(s2))
    fid_value = calculate_frechet_distance(m1, s1, m2, s2)
    return fid_value


def main(argv=None):
    args = _build_parser().parse_args(argv)
    if args.gpu == "":
        _check_cuda(False)
    rank, world, local_rank = tdist.init_from_env()
    if world == 1:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)        # reference: CUDA_VISIBLE_DEVICES = args.gpu (:243)
    paths = [args.path1, args.path2]
    if tdist.is_main():
        print(paths)                                                   # :247
    fid_value = calculate_fid_given_paths(paths, args.batch_size, args.gpu, args.dims, args.weights,
                                          args.num_classes, args.seed, args.save_stats, args.num_workers).item()
    if tdist.is_main():
        if args.saved_file:
            with open(args.saved_file, "w") as f:
                f.write(f"{args.label}: {fid_value}")                  # :251-252 (no trailing newline)
        print(f"{args.label}: {fid_value}")                            # :254
    return fid_value


if __name__ == "__main__":
    main()
