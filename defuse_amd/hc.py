"""ctypes binding of include/defuse_hc.h (batched average-linkage clustering on the GPU); test/bench plumbing only."""
import ctypes

import numpy as np

from .dsa import load_library


class HcTiming(ctypes.Structure):
    _fields_ = [("upload_ms", ctypes.c_float), ("kernel_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("n_merges", ctypes.c_int32)]


def cluster_batch(tables, thresholds, device=0):
    """tables: list of n×n arrays; thresholds: one per table.  Returns (list of cluster lists per table, timing)."""
    lib = load_library()
    vp = ctypes.c_void_p
    lib.hc_cluster_batch.argtypes = [ctypes.c_int, ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(HcTiming)]
    lib.hc_last_error.restype = ctypes.c_char_p
    mats = [np.ascontiguousarray(t, dtype=np.float64).reshape(len(t), len(t)) if len(t) else np.zeros((0, 0)) for t in tables]
    n_items = np.array([len(m) for m in mats], dtype=np.int32)
    dist_off = np.zeros(len(mats), dtype=np.int64)
    if len(mats):
        dist_off[1:] = np.cumsum(n_items.astype(np.int64) ** 2)[:-1]
    flat = np.concatenate([m.ravel() for m in mats]) if len(mats) else np.zeros(0)
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    total = int(n_items.sum())
    members = np.zeros(max(total, 1), dtype=np.int32)
    cluster_of = np.zeros(max(total, 1), dtype=np.int32)
    n_clusters = np.zeros(max(len(mats), 1), dtype=np.int32)
    t = HcTiming()
    rc = lib.hc_cluster_batch(device, len(mats), n_items.ctypes.data, dist_off.ctypes.data, flat.ctypes.data if flat.size else None,
                              thr.ctypes.data, members.ctypes.data, cluster_of.ctypes.data, n_clusters.ctypes.data, ctypes.byref(t))
    if rc != 0:
        raise RuntimeError("hc_cluster_batch failed (%d): %s" % (rc, lib.hc_last_error().decode()))
    out, base = [], 0
    for p, n in enumerate(n_items):
        cl = [[] for _ in range(n_clusters[p])]
        for k in range(n):
            cl[cluster_of[base + k]].append(int(members[base + k]))
        out.append(cl)
        base += n
    return out, t
