"""swarmkit_amd — MI355X-native batch task-placement engine behind swarmkit's scheduler seam.

The product is libswp.so (swarmkit_amd/lib/, built from swarmkit_amd/csrc/ with hipcc for gfx950)
and its C ABI include/swp.h. This package is the thin Python binding used by tests and bench.py;
there is no CPU implementation of the placement path in here.
"""
from .abi import Engine, SwpError, load_library, build_library  # noqa: F401
