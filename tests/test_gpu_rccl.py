"""RCCL executes: a one-rank "nccl" group forced through the product's dist.init_from_env on the single GPU of the test
box (BASELINE configs[2]'s collective).  Every leg is a FRESH child process (tools/rccl_probe.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(*legs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_probe.py"), *legs], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout)


def test_rccl_one_rank_group_runs_the_jobs_collectives(cuda_device):
    """init (backend nccl) -> all-reduce of the 33.57 MB [S|s|n] buffer and the IS* sums -> reduce to the owner -> any_rank
    -> barrier -> destroy, through tise_toolbox_amd.dist and RealismEngine.reduce(); with HSA_ENABLE_IPC_MODE_LEGACY=0 and
    with the variable unset (a one-rank group opens no IPC handle, so both must work)."""
    res = _probe("world1")["world1"]
    print(json.dumps(res, indent=1))
    for name, leg in res.items():
        assert leg["returncode"] == 0, (name, leg.get("stderr_tail"))
        assert leg["backend"] == "nccl" and leg["world_size"] == 1
        assert leg["stats_buffer_bytes"] >= 8 * (2048 * 2048 + 2048 + 1)
        assert leg["engine_reduce_ok"] and leg["sigma_err_vs_npcov"] <= 1e-12
        assert 0 < leg["allreduce_stats_ms"] < 50


def test_device_memory_ipc_between_processes(cuda_device):
    """hipIpcGetMemHandle / OpenMemHandle between two processes on the GPU -- what HSA_ENABLE_IPC_MODE_LEGACY governs and
    what RCCL's intra-node P2P transport needs between ranks.  It must work in the configuration bench._self_launch starts
    its ranks with (variable = 0).  With the variable unset the child never gets the tensor (`python tools/rccl_probe.py ipc`,
    profiles/r05a_rccl_world1.txt) -- which is why _self_launch keeps it."""
    # (the leg with the variable UNSET hangs in hipIpcOpenMemHandle until the probe's 120 s timeout -- profiles/r05a_rccl_world1.txt
    # records it; the suite runs the configuration the product uses)
    res = _probe("ipc", "--with-var-only")["ipc"]
    print(json.dumps(res, indent=1))
    assert res["with HSA_ENABLE_IPC_MODE_LEGACY=0"].get("ok") is True, res
