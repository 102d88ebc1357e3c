"""Where the time before the first device batch of the PNG feed goes: each setup step of png_ring.PngRingLoader timed alone."""
import mmap
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tise_toolbox_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)
_lib.load()
for rep in range(3):
    t = [time.perf_counter()]
    size = 36 * 8 * 197056
    fd = os.memfd_create("probe")
    os.ftruncate(fd, size)
    m = mmap.mmap(fd, size)
    t.append(time.perf_counter())
    addr = np.frombuffer(m, dtype=np.uint8).ctypes.data
    _lib.call("tise_host_register", addr, size)
    t.append(time.perf_counter())
    side = torch.cuda.Stream(device=dev)
    t.append(time.perf_counter())
    bufs = [torch.empty((2800, 256, 256, 3), dtype=torch.uint8, device=dev) for _ in range(3)]
    raw = torch.empty((2800, 197056), dtype=torch.uint8, device=dev)
    t.append(time.perf_counter())
    ev = [torch.cuda.Event() for _ in range(6)]
    t.append(time.perf_counter())
    _lib.call("tise_memcpy_h2d_async", raw.data_ptr(), addr, 8 * 197056, side.cuda_stream)
    side.synchronize()
    t.append(time.perf_counter())
    _lib.call("tise_host_unregister", addr)
    t.append(time.perf_counter())
    names = ["memfd+mmap", "hipHostRegister 57 MB", "stream", "4 device buffers", "events", "first 1.6 MB copy + sync", "unregister"]
    print(" | ".join(f"{n} {1e3 * (b - a):.2f} ms" for n, a, b in zip(names, t, t[1:])))
    del bufs, raw
    m.close()
    os.close(fd)
