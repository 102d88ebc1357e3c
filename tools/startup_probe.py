"""Where a CLI process's start-up goes: imports and the model / engine build under cProfile (top cumulative entries)."""
import cProfile
import io
import os
import pstats
import sys
import time

t0 = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
t1 = time.perf_counter()
from tise_toolbox_amd import fid_score  # noqa: E402
t2 = time.perf_counter()
print(f"import torch {t1 - t0:.2f} s, import tise_toolbox_amd.fid_score {t2 - t1:.2f} s")
torch.cuda.init()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t3 = time.perf_counter()
print(f"HIP context + first allocation {t3 - t2:.2f} s")
# first build untimed: it creates the stand-in cache file on a fresh box (50 s of calibration) and loads the code objects
fid_score._engine_for(fid_score._build_model(2048, None, 1000, 0), 2048)
torch.cuda.synchronize()
from tise_toolbox_amd import inception as _inc  # noqa: E402
_inc._SEEDED_CACHE.clear()                       # the second build reads the cache FILE again, like a fresh CLI process
t3 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
model = fid_score._build_model(2048, None, 1000, 0)
t4 = time.perf_counter()
eng = fid_score._engine_for(model, 2048)
torch.cuda.synchronize()
t5 = time.perf_counter()
pr.disable()
print(f"_build_model {t4 - t3:.2f} s, engine (fold, pack, code objects) {t5 - t4:.2f} s")
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
