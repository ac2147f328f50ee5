"""ractip_amd -- MI355X-native probability-matrix engine for RactIP's hot path.

Holds only what the path needs: csrc/ (HIP kernels + the C ABI of
include/ractip_hot.h), host/ (C++ mirror of the RactIP::contrafold /
contraduplex / rnaduplex call surface), hot.py (ctypes binding used by tests and
bench.py) and data/ (the CONTRAfold weights).
"""
from .hot import Context, RhError, load_library, tri_offset, tri_size  # noqa: F401
