"""ctypes binding of include/defuse_la.h (batched SimpleAligner scores on the GPU); test/bench plumbing only."""
import ctypes

import numpy as np

from .dsa import load_library

LA_ITEM = np.dtype([("ref_off", np.int64), ("seq_off", np.int64), ("ref_len", np.int32), ("seq_len", np.int32)])


class LaTiming(ctypes.Structure):
    _fields_ = [("pack_ms", ctypes.c_float), ("kernel_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("n_packed16", ctypes.c_int32), ("n_int32", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("cells", ctypes.c_int64)]


def align_batch(pairs, match, mismatch, gap, device=0, min_score=None):
    """pairs: list of (reference bytes, sequence bytes).  Returns (int32 scores, timing).  With min_score (one
    int per pair) a score below its minimum is only guaranteed to be below it (la_align_batch_min)."""
    lib = load_library()
    lib.la_align_batch_min.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(LaTiming)]
    lib.la_last_error.restype = ctypes.c_char_p
    items = np.zeros(len(pairs), dtype=LA_ITEM)
    chunks, off = [], 0
    for k, (r, s) in enumerate(pairs):
        items[k] = (off, off + len(r), len(r), len(s))
        chunks += [bytes(r), bytes(s)]
        off += len(r) + len(s)
    pool = np.frombuffer(b"".join(chunks) + b"\0", dtype=np.uint8)
    scores = np.zeros(len(pairs), dtype=np.int32)
    t = LaTiming()
    need = None if min_score is None else np.ascontiguousarray(min_score, dtype=np.int32)
    rc = lib.la_align_batch_min(device, match, mismatch, gap, pool.ctypes.data, off, items.ctypes.data if len(pairs) else None,
                                len(pairs), need.ctypes.data if need is not None and len(pairs) else None,
                                scores.ctypes.data if len(pairs) else None, ctypes.byref(t))
    if rc != 0:
        raise RuntimeError("la_align_batch_min failed (%d): %s" % (rc, lib.la_last_error().decode()))
    return scores, t
