"""MI355X-native image-realism hot path of the TISE toolbox (IS* / FID)."""
import os as _os

__version__ = "0.1.0"

# A rank of a multi-process job (torchrun exports WORLD_SIZE): RCCL's intra-node transport needs dmabuf IPC on this driver
# (dist._ipc_default), and ROCr reads the switch once, when HIP is first initialised -- so the default goes in at import,
# before any torch.cuda call of the CLI.  TISE_DIST_BACKEND=gloo (CPU tests) leaves the environment alone.
if int(_os.environ.get("WORLD_SIZE", "1") or 1) > 1 and _os.environ.get("TISE_DIST_BACKEND") != "gloo":
    _os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
