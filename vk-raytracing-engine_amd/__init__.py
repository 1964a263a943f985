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
HOST_LIB_PATH = os.path.join(PKG_DIR, "libvkrt_host.so")

from . import abi  # noqa: E402,F401
from . import flat_scene  # noqa: E402,F401
