"""MI355X-native drop-in for the path-tracing path of vk-raytracing-engine.

The product is the C-ABI shared library built from csrc/ (include/vkrt.h) plus the C++ host
mirror of the reference's `HelloVulkan` interface in host/.  This Python package is harness
plumbing only: ctypes declarations, flat-scene containers, torch device buffers / streams
and the torch.distributed gather for multi-GPU strips.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libvkrt.so")
if os.environ.get("VKRT_LIB"):  # test hook: A/B of two builds of the library from one tree (tools/probe_variants.sh)
    LIB_PATH = os.path.abspath(os.environ["VKRT_LIB"])
HOST_LIB_PATH = os.path.join(PKG_DIR, "libvkrt_host.so")

from . import abi  # noqa: E402,F401
from . import flat_scene  # noqa: E402,F401


def source_hash():
    """sha256 (first 16 hex digits) over the kernel / ABI sources: ties committed PMC summaries (profiles/pmc_*.json) to
    the code they were measured on, so bench.py can refuse stale ones."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.h")) + glob.glob(os.path.join(PKG_DIR, "csrc", "*.hip"))
                   + glob.glob(os.path.join(PKG_DIR, "csrc", "*.cpp")) + glob.glob(os.path.join(REPO_ROOT, "include", "*.h"))
                   + [os.path.join(PKG_DIR, "csrc", "Makefile")])  # (the Makefile: compiler flags change the kernels too)
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
