"""goldrush_amd — MI355X-native GoldRush-Path hot path.

Spaced-seed ntHash -> multi-index Bloom filter (miBF) fill / ID insert / tile
query as hand-written gfx950 HIP kernels behind the C ABI of include/grpath.h.
This Python package is only the ctypes plumbing used by the tests and
bench.py; the product host is the C++ `goldrush-path` CLI (goldrush_amd/csrc).
"""
from . import native  # noqa: F401

__all__ = ["native"]
