"""Build recipe for libtise_hip.so (hipcc, gfx950 only, in-tree output).

``python -m tise_toolbox_amd.build`` or ``__graft_entry__.build()``.  The library is
written next to this file so that it travels with the source tree (it is git-ignored).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtise_hip.so")
SOURCES = ["capi.hip", "stats.hip", "resize.hip", "is_score.hip", "frechet.hip", "trunk_ops.hip", "conv_split.hip", "conv_pipe.hip", "retrieval.hip", "clip_ops.hip", "png_unfilter.hip"]
HEADERS = ["common.h", "gemm_tile.h", "conv_epilogue.h", os.path.join("..", "..", "include", "tise_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(p) > t for p in deps if os.path.exists(p))


PNG_LIB = os.path.join(HERE, "libtise_png.so")
PNG_SRC = os.path.join(CSRC, "png_decode.c")


PNG_WORKER = os.path.join(HERE, "tise_png_worker")
PNG_WORKER_SRC = os.path.join(CSRC, "png_worker.c")


def build_png(force=False, verbose=True):
    """gcc over csrc/png_decode.c -> tise_toolbox_amd/libtise_png.so: the host-side PNG decoder of the image feed (plain C,
    links zlib, dlopens libdeflate when present; no HIP, so a worker never loads the GPU runtime); and over
    csrc/png_worker.c + png_decode.c -> tise_toolbox_amd/tise_png_worker: the native decode process of the feed."""
    cc = os.environ.get("CC", "gcc")
    newest = max(os.path.getmtime(PNG_SRC), os.path.getmtime(PNG_WORKER_SRC))
    if force or not os.path.exists(PNG_LIB) or os.path.getmtime(PNG_LIB) < os.path.getmtime(PNG_SRC):
        cmd = [cc, "-O3", "-mssse3", "-msse4.1", "-fPIC", "-shared", "-o", PNG_LIB, PNG_SRC, "-lz", "-ldl"]
        if verbose:
            print("[tise build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    if force or not os.path.exists(PNG_WORKER) or os.path.getmtime(PNG_WORKER) < newest:
        cmd = [cc, "-O3", "-mssse3", "-msse4.1", "-o", PNG_WORKER, PNG_WORKER_SRC, PNG_SRC, "-lz", "-ldl"]
        if verbose:
            print("[tise build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return PNG_LIB


OBJ_DIR = os.path.join(CSRC, "_obj")       # per-source objects (git-ignored: *.o): a change to one kernel file recompiles that file only


def _deps(src):
    """Headers a source includes (one level of #include "..." is all csrc/ uses) + the public header."""
    out = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "tise_hip.h")]
    with open(os.path.join(CSRC, src)) as f:
        for line in f:
            if line.startswith('#include "'):
                out.append(os.path.join(CSRC, line.split('"')[1]))
    return [p for p in out if os.path.exists(p)]


def build(force=False, verbose=True, jobs=None):
    """Compile every HIP source for gfx950 into tise_toolbox_amd/libtise_hip.so (and the host PNG decoder, build_png).
    Each source becomes csrc/_obj/<name>.o (compiled in parallel, recompiled only when it or a header it includes is
    newer), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    build_png(force, verbose)
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    todo, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src)):
            todo.append((src, obj))

    def compile_one(item):
        src, obj = item
        cmd = [_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[tise build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)

    if jobs is None:
        jobs = max(1, min(len(todo) or 1, (os.cpu_count() or 4), int(os.environ.get("TISE_BUILD_JOBS", "8"))))
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(compile_one, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print("[tise build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
