"""defuse_amd — MI355X-native split-read alignment path of deFuse (dosplitalign's DP hot path).

The product is the C-ABI library built from defuse_amd/csrc (include/defuse_dsa.h) and the tool
binaries on top of it; this Python package is only the ctypes plumbing used by tests and bench.py.
"""
__version__ = "0.1"
