import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, bench
from tise_toolbox_amd.engine import RealismEngine
dev = torch.device("cuda", 0)
eng = RealismEngine(dims=2048, device_index=0, seed=0, with_logits=False)
for n in (250, 1000, 2000, 3000):
    data = bench.synth_images_device(0, n, dev, seed=0)
    torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    eng.features_from_u8(data)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(n, "images: peak allocated above base", (torch.cuda.max_memory_allocated() - base) / 2**30, "GiB =", (torch.cuda.max_memory_allocated() - base) / n / 2**20, "MiB/image; reserved", torch.cuda.memory_reserved() / 2**30, "GiB; first pass", round(dt, 3), "s")
    t0 = time.perf_counter(); eng.features_from_u8(data); torch.cuda.synchronize(); print("   second pass", round(time.perf_counter() - t0, 3), "s")
