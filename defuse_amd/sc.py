"""ctypes binding of include/defuse_sc.h (greedy set cover on the GPU); test/bench plumbing only."""
import ctypes

import numpy as np

from .dsa import load_library


class ScTiming(ctypes.Structure):
    _fields_ = [("build_ms", ctypes.c_float), ("components_ms", ctypes.c_float), ("greedy_ms", ctypes.c_float),
                ("total_ms", ctypes.c_float), ("n_components", ctypes.c_int32), ("n_large", ctypes.c_int32),
                ("cc_iterations", ctypes.c_int32), ("pad_", ctypes.c_int32)]


def cover(clusters, device=0):
    """clusters: list of lists of fragment indices.  Returns (solution lists, timing)."""
    lib = load_library()
    lib.sc_cover.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                             ctypes.c_void_p, ctypes.POINTER(ScTiming)]
    lib.sc_last_error.restype = ctypes.c_char_p
    off = np.zeros(len(clusters) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(c) for c in clusters])
    el = np.array([e for c in clusters for e in c], dtype=np.int32)
    max_el = int(el.max()) if len(el) else -1
    owner = np.full(max_el + 1, -1, dtype=np.int32)
    t = ScTiming()
    rc = lib.sc_cover(device, off.ctypes.data, el.ctypes.data if len(el) else None, len(clusters), max_el,
                      owner.ctypes.data if len(owner) else None, ctypes.byref(t))
    if rc != 0:
        raise RuntimeError("sc_cover failed (%d): %s" % (rc, lib.sc_last_error().decode()))
    sol = [[] for _ in clusters]
    for c, lst in enumerate(clusters):
        for e in lst:
            if owner[e] == c and e not in sol[c]:
                sol[c].append(e)
    return sol, t
