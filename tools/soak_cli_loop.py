"""Soak of the drop-in CLIs inside ONE process: 16 FID runs over two directories of 400 random PNGs (alternating sides) with an IS\* run
every fourth -- file descriptors, resident memory, child processes and device memory must not grow, the two FID values must repeat to the bit.
    python tools/soak_cli_loop.py"""
import os, sys, time, tempfile, resource
os.environ.setdefault("TISE_RELEASE_MODEL", "1")      # fid_score._own_model: release a call's model on return (opt-in)
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from PIL import Image
from tise_toolbox_amd import fid_score, inception_score
root = tempfile.mkdtemp(prefix="tise_soak_")
rng = np.random.default_rng(0)
for name in ("a", "b"):
    os.makedirs(os.path.join(root, name))
    for i in range(400):
        Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(os.path.join(root, name, f"{i:05d}.png"), compress_level=1)
def nfd(): return len(os.listdir("/proc/self/fd"))
def rss(): return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20
def nchild():
    me = os.getpid(); n = 0
    for p in os.listdir("/proc"):
        if p.isdigit():
            try:
                if int(open(f"/proc/{p}/stat").read().rsplit(")", 1)[1].split()[1]) == me: n += 1
            except Exception: pass
    return n
vals = []
for it in range(16):
    a, b = ("a", "b") if it % 2 == 0 else ("b", "a")
    v = fid_score.main(["--batch-size", "50", "--path1", os.path.join(root, a), "--path2", os.path.join(root, b), "--synthetic-weights"])
    vals.append(v)
    if it % 4 == 3:
        inception_score.main(["--image_folder", os.path.join(root, "a"), "--batch-size", "50", "--synthetic-weights"])
    print(f"iter {it}: fid {v:.9f} fds {nfd()} rss {rss():.0f} MiB children {nchild()} torch alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB", flush=True)
assert len(set(round(v, 9) for v in vals[0::2])) == 1 and len(set(round(v, 9) for v in vals[1::2])) == 1, vals
assert nfd() <= 16 and nchild() == 0, (nfd(), nchild())
assert torch.cuda.memory_allocated() / 2**20 < 3000, torch.cuda.memory_allocated()          # one call's worth: nothing accumulates
print("soak ok")

# ---- second part: the ragged-crop per-class path (DataLoader workers forked from this process), six runs ----
crops = {}
for side in ("gen", "ref"):
    d = os.path.join(root, "crops_" + side)
    os.makedirs(d)
    k = 0
    for c in ("person", "dog", "traffic light", "cup", "zebra", "pizza"):
        for j in range(24):
            hh, ww = 40 + (7 * j + k) % 60, 36 + (11 * j + 3 * k) % 70
            Image.fromarray(rng.integers(0, 256, (hh, ww, 3), dtype=np.uint8)).save(os.path.join(d, f"im{k}_{c}_{k}.png"), compress_level=1)
            k += 1
    crops[side] = d
pc = []
for it in range(6):
    per = fid_score.main(["--batch-size", "16", "--path1", crops["ref"], "--path2", crops["gen"], "--label", "O-FID", "--num-classes", "80",
                          "--per-class", "--synthetic-weights", "--num-workers", "4"])
    pc.append(tuple(round(float(v), 9) for v in per.values()))
    print(f"per-class iter {it}: {len(per)} classes, fds {nfd()} rss {rss():.0f} MiB children {nchild()} torch alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB", flush=True)
assert len(set(pc)) == 1, pc
assert nfd() <= 20 and nchild() == 0, (nfd(), nchild())
print("per-class soak ok")

# ---- third part: the RP-COCO CLI (CLIP stand-in towers), six runs ----
import pickle  # noqa: E402
from tise_toolbox_amd import RP_coco  # noqa: E402
words = ["a", "red", "bus", "dog", "on", "the", "grass", "two", "people", "near", "table"]
img_dir = os.path.join(root, "rp_images")
os.makedirs(img_dir)
items = []
pool = [" ".join(rng.choice(words, 5)) + f" x{k}" for k in range(30)]
for i in range(64):
    items.append({"caption_id": 100 + i, "caption": " ".join(rng.choice(words, 5)) + f" {i}",
                  "mismatched_captions": [pool[(i * 3 + 5 * j) % 30] for j in range(6)]})
    Image.fromarray(rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)).save(os.path.join(img_dir, f"{100 + i}.png"))
pkl = os.path.join(root, "rp.pkl")
pickle.dump(items, open(pkl, "wb"))
rp = []
for it in range(6):
    mean, std = RP_coco.main(["--image_dir", img_dir, "--rp_input_file", pkl, "--saved_file_path", os.path.join(root, "rp.txt"),
                              "--synthetic-weights", "--seed", "7"])
    rp.append((round(float(mean), 9), round(float(std), 9)))
    print(f"rp iter {it}: {rp[-1]} fds {nfd()} rss {rss():.0f} MiB children {nchild()} torch alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB", flush=True)
assert len(set(rp)) == 1, rp
assert nfd() <= 20 and nchild() == 0, (nfd(), nchild())
print("rp soak ok")

# ---- fourth part: the O-IS CLI on the crop directory (DataLoader workers), the FID CLI through --u8-cache, four runs each ----
from tise_toolbox_amd import object_centric_inception_score as ois  # noqa: E402
ov, cv = [], []
for it in range(4):
    m, sd = ois.main(["--image_dir", crops["gen"], "--synthetic-weights"])
    ov.append((round(float(m), 9), round(float(sd), 9)))
    v = fid_score.main(["--batch-size", "50", "--path1", os.path.join(root, "a"), "--path2", os.path.join(root, "b"), "--synthetic-weights", "--u8-cache"])
    cv.append(round(float(v), 9))
    print(f"o-is / u8-cache iter {it}: {ov[-1]} {cv[-1]} fds {nfd()} rss {rss():.0f} MiB children {nchild()} torch alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB", flush=True)
assert len(set(ov)) == 1 and len(set(cv)) == 1 and cv[0] == round(vals[1], 9), (ov, cv, vals[1])
assert nfd() <= 24 and nchild() == 0, (nfd(), nchild())
print("o-is / u8-cache soak ok")
