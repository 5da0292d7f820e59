"""ctypes binding of include/defuse_mpe.h (mate-pair EM clustering on the GPU); test/bench plumbing only."""
import ctypes

import numpy as np

from .dsa import load_library


class MpeParams(ctypes.Structure):
    _fields_ = [("fragment_mean", ctypes.c_double), ("fragment_stddev", ctypes.c_double), ("min_probability", ctypes.c_double),
                ("min_cluster_size", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class MpeTiming(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_float), ("n_problems", ctypes.c_int32), ("n_mate_pairs", ctypes.c_int64),
                ("em_iterations", ctypes.c_int64), ("n_failed", ctypes.c_int32), ("n_wave_problems", ctypes.c_int32)]


def cluster_batch(mean, sd, min_prob, min_size, prob_off, x, y, u, to_xo, to_yo, device=0):
    """Returns (n_clusters per problem, member bit masks per mate pair, status per problem, timing)."""
    lib = load_library()
    lib.mpe_cluster_batch.argtypes = [ctypes.c_int, ctypes.POINTER(MpeParams), ctypes.c_void_p, ctypes.c_int32] + \
        [ctypes.c_void_p] * 8 + [ctypes.POINTER(MpeTiming)]
    lib.mpe_last_error.restype = ctypes.c_char_p
    prob_off = np.ascontiguousarray(prob_off, dtype=np.int64)
    n = len(prob_off) - 1
    x, y, u = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, u))
    to_xo, to_yo = (np.ascontiguousarray(v, dtype=np.int32) for v in (to_xo, to_yo))
    total = int(prob_off[-1])
    n_clusters = np.zeros(n, dtype=np.int32)
    member = np.zeros(max(total, 1), dtype=np.uint16)
    status = np.zeros(n, dtype=np.int32)
    prm = MpeParams(mean, sd, min_prob, int(min_size), 0)
    t = MpeTiming()
    rc = lib.mpe_cluster_batch(device, ctypes.byref(prm), prob_off.ctypes.data, n, x.ctypes.data, y.ctypes.data, u.ctypes.data,
                               to_xo.ctypes.data, to_yo.ctypes.data, n_clusters.ctypes.data, member.ctypes.data, status.ctypes.data,
                               ctypes.byref(t))
    if rc != 0:
        raise RuntimeError("mpe_cluster_batch failed (%d): %s" % (rc, lib.mpe_last_error().decode()))
    return n_clusters, member[:total], status, t
