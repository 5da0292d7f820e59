"""ctypes binding of include/defuse_dsa.h (libdefuse_dsa.so, HIP/gfx950).

No CPU fallback: if the library is missing or there is no GPU, construction raises.
The numpy structured dtypes mirror the C structs byte for byte.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEFUSE_DSA_LIB") or os.path.join(_HERE, "libdefuse_dsa.so")

FUSION_DTYPE = np.dtype([("fusion_id", "<i4"), ("ref0_off", "<i4"), ("ref0_len", "<i4"),
                         ("ref1_off", "<i4"), ("ref1_len", "<i4")])
PAIR_DTYPE = np.dtype([("fusion_idx", "<i4"), ("read_off", "<i4"), ("read_len", "<i4"), ("frag", "<i4"),
                       ("read_end", "u1"), ("revcomp", "u1"), ("pad_", "u1", (2,))])
RECORD_DTYPE = np.dtype([(n, "<i4") for n in ("fusion_id", "frag", "read_end", "revcomp", "ref_first",
                                             "ref_second", "read_first", "read_second", "score", "pair_idx")])
assert FUSION_DTYPE.itemsize == 20 and PAIR_DTYPE.itemsize == 20 and RECORD_DTYPE.itemsize == 40

EXPORTS = ["dsa_create", "dsa_destroy", "dsa_get_limits", "dsa_last_error", "dsa_version", "dsa_build_flags", "dsa_device_count", "dsa_pick_device", "dsa_pick_device_among", "dsa_set_plan_options", "dsa_set_scratch_budget", "dsa_share_scratch", "dsa_align_batch",
           "dsa_upload", "dsa_plan", "dsa_run", "dsa_download", "dsa_copy_records_device", "dsa_get_timing", "dsa_set_stream", "dsa_synchronize",
           "dsa_stream_create", "dsa_stream_destroy", "dsa_stream_submit", "dsa_stream_collect", "dsa_stream_recollect", "dsa_stream_last_error",
           "dsa_host_alloc", "dsa_host_free", "dsa_host_register", "dsa_host_unregister"]

PLAN_NO_REORDER, PLAN_NO_RANK, PLAN_NO_TIGHTEN, PLAN_NO_LPT = 1, 2, 4, 8
DSA_E_CAPACITY = -1
DSA_E_ARG = -3
DSA_E_BUSY = -5


class Limits(ctypes.Structure):
    _fields_ = [("max_read_len", ctypes.c_int32), ("max_ref_len", ctypes.c_int32), ("tile_cols", ctypes.c_int32)]


class Timing(ctypes.Structure):
    _fields_ = [("pack_ms", ctypes.c_float), ("fill_ms", ctypes.c_float), ("finish_ms", ctypes.c_float),
                ("total_ms", ctypes.c_float), ("fill_launches", ctypes.c_int32), ("n_generic_tasks", ctypes.c_int32),
                ("cells", ctypes.c_int64), ("n_records", ctypes.c_int64), ("n_replay_tasks", ctypes.c_int64),
                ("plan_ms", ctypes.c_float), ("pad_", ctypes.c_float)]


class DsaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dsa error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """Loads libdefuse_dsa.so; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run `python -m defuse_amd.build`" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        lib.dsa_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
        lib.dsa_destroy.argtypes = [vp]
        lib.dsa_destroy.restype = None
        lib.dsa_get_limits.argtypes = [vp, ctypes.POINTER(Limits)]
        lib.dsa_last_error.argtypes = [vp]
        lib.dsa_last_error.restype = ctypes.c_char_p
        lib.dsa_version.restype = ctypes.c_char_p
        lib.dsa_build_flags.restype = ctypes.c_char_p
        batch = [vp, vp, i64, vp, i32, vp, i64, vp, i64]
        lib.dsa_align_batch.argtypes = batch + [vp, i64, ctypes.POINTER(i64)]
        lib.dsa_upload.argtypes = batch
        lib.dsa_plan.argtypes = [vp]
        lib.dsa_run.argtypes = [vp, ctypes.POINTER(i64)]
        lib.dsa_download.argtypes = [vp, vp, i64, ctypes.POINTER(i64)]
        lib.dsa_copy_records_device.argtypes = [vp, vp, i64, ctypes.POINTER(i64)]
        lib.dsa_get_timing.argtypes = [vp, ctypes.POINTER(Timing)]
        lib.dsa_set_stream.argtypes = [vp, vp]
        lib.dsa_synchronize.argtypes = [vp]
        lib.dsa_set_scratch_budget.argtypes = [vp, i64]
        lib.dsa_set_plan_options.argtypes = [vp, ctypes.c_uint]
        lib.dsa_share_scratch.argtypes = [vp, vp]
        lib.dsa_pick_device_among.argtypes = [ctypes.c_int]
        lib.dsa_stream_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int]
        lib.dsa_stream_destroy.argtypes = [vp]
        lib.dsa_stream_destroy.restype = None
        lib.dsa_stream_submit.argtypes = [vp] + batch[1:] + [vp, i64]
        lib.dsa_stream_collect.argtypes = [vp, ctypes.POINTER(i64)]
        lib.dsa_stream_recollect.argtypes = [vp, vp, i64, ctypes.POINTER(i64)]
        lib.dsa_stream_last_error.argtypes = [vp]
        lib.dsa_stream_last_error.restype = ctypes.c_char_p
        lib.dsa_host_alloc.argtypes = [ctypes.c_size_t]
        lib.dsa_host_alloc.restype = vp
        lib.dsa_host_free.argtypes = [vp]
        lib.dsa_host_free.restype = None
        lib.dsa_host_register.argtypes = [vp, ctypes.c_size_t]
        lib.dsa_host_unregister.argtypes = [vp]
        _lib = lib
    return _lib


def _check_arrays(ref_bytes, fusions, read_bytes, pairs):
    ref_bytes = np.ascontiguousarray(ref_bytes, dtype=np.uint8)
    read_bytes = np.ascontiguousarray(read_bytes, dtype=np.uint8)
    fusions = np.ascontiguousarray(fusions, dtype=FUSION_DTYPE)
    pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    return ref_bytes, fusions, read_bytes, pairs


class Context:
    """One dsa_ctx (one device).  Mirrors the staged C API: upload -> run -> download."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = ctypes.c_void_p()
        self.device = int(device)
        rc = self.lib.dsa_create(ctypes.byref(self.h), int(device))
        if rc != 0:
            raise DsaError(rc, "dsa_create failed (no HIP device %d?)" % device)
        self._keep = None

    def close(self):
        if self.h:
            self.lib.dsa_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        raise DsaError(rc, self.lib.dsa_last_error(self.h).decode())

    def limits(self):
        lim = Limits()
        self.lib.dsa_get_limits(self.h, ctypes.byref(lim))
        return lim

    def set_plan_options(self, flags):
        """dsa_set_plan_options: PLAN_NO_* bits; effective from the next upload() / plan()."""
        rc = self.lib.dsa_set_plan_options(self.h, int(flags))
        if rc != 0:
            self._err(rc)

    def set_scratch_budget(self, nbytes):
        rc = self.lib.dsa_set_scratch_budget(self.h, int(nbytes))
        if rc != 0:
            self._err(rc)

    def share_scratch(self, donor):
        rc = self.lib.dsa_share_scratch(self.h, donor.h)
        if rc != 0:
            self._err(rc)

    def set_stream(self, stream_ptr):
        self.lib.dsa_set_stream(self.h, ctypes.c_void_p(stream_ptr))

    def upload(self, ref_bytes, fusions, read_bytes, pairs):
        ref_bytes, fusions, read_bytes, pairs = _check_arrays(ref_bytes, fusions, read_bytes, pairs)
        rc = self.lib.dsa_upload(self.h, ref_bytes.ctypes.data, ref_bytes.size, fusions.ctypes.data, len(fusions),
                                 read_bytes.ctypes.data, read_bytes.size, pairs.ctypes.data, len(pairs))
        if rc != 0:
            self._err(rc)

    def plan(self):
        """dsa_plan: the sweep planning of the resident upload once more (bench.py times plan + run per step)."""
        rc = self.lib.dsa_plan(self.h)
        if rc != 0:
            self._err(rc)

    def run(self):
        n = ctypes.c_int64(0)
        rc = self.lib.dsa_run(self.h, ctypes.byref(n))
        if rc != 0:
            self._err(rc)
        return n.value

    def download(self):
        n = ctypes.c_int64(0)
        rc = self.lib.dsa_download(self.h, None, 0, ctypes.byref(n))
        if rc not in (0, DSA_E_CAPACITY):
            self._err(rc)
        out = np.zeros(n.value, dtype=RECORD_DTYPE)
        if n.value:
            rc = self.lib.dsa_download(self.h, out.ctypes.data, n.value, ctypes.byref(n))
            if rc != 0:
                self._err(rc)
        return out

    def records_to_device(self, device_ptr, capacity):
        """Copies the records of the last run into device memory the caller owns (e.g. a torch tensor's
        data_ptr()); returns the record count.  Raises DsaError(DSA_E_CAPACITY) if it does not fit."""
        n = ctypes.c_int64(0)
        rc = self.lib.dsa_copy_records_device(self.h, ctypes.c_void_p(device_ptr), int(capacity), ctypes.byref(n))
        if rc != 0:
            self._err(rc)
        return n.value

    def timing(self):
        t = Timing()
        self.lib.dsa_get_timing(self.h, ctypes.byref(t))
        return t

    def align_batch_into(self, ref_bytes, fusions, read_bytes, pairs, out):
        """dsa_align_batch on the caller's arrays as they are (no copies: pinned buffers stay pinned); the records go to
        `out` (RECORD_DTYPE).  Returns the record count; raises DsaError(DSA_E_CAPACITY) if `out` is too small."""
        n = ctypes.c_int64(0)
        rc = self.lib.dsa_align_batch(self.h, ref_bytes.ctypes.data, ref_bytes.size, fusions.ctypes.data, len(fusions),
                                      read_bytes.ctypes.data, read_bytes.size, pairs.ctypes.data, len(pairs),
                                      out.ctypes.data, len(out), ctypes.byref(n))
        if rc != 0:
            self._err(rc)
        return n.value

    def align_batch(self, ref_bytes, fusions, read_bytes, pairs):
        """dsa_align_batch: host arrays in, numpy record array out."""
        ref_bytes, fusions, read_bytes, pairs = _check_arrays(ref_bytes, fusions, read_bytes, pairs)
        cap = max(1024, 2 * len(pairs))
        while True:
            out = np.zeros(cap, dtype=RECORD_DTYPE)
            n = ctypes.c_int64(0)
            rc = self.lib.dsa_align_batch(self.h, ref_bytes.ctypes.data, ref_bytes.size, fusions.ctypes.data,
                                          len(fusions), read_bytes.ctypes.data, read_bytes.size, pairs.ctypes.data,
                                          len(pairs), out.ctypes.data, cap, ctypes.byref(n))
            if rc == 0:
                return out[:n.value].copy()
            if rc == DSA_E_CAPACITY:
                cap = int(n.value)
                continue
            self._err(rc)


class PinnedArray:
    """A numpy array in pinned host memory (dsa_host_alloc): the buffers of a Stream copy asynchronously from / to it."""

    def __init__(self, shape, dtype):
        self.lib = load_library()
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        self.ptr = self.lib.dsa_host_alloc(max(n, 1))
        if not self.ptr:
            raise MemoryError("dsa_host_alloc(%d) failed" % n)
        buf = (ctypes.c_uint8 * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.uint8, count=n).view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.lib.dsa_host_free(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pinned_copy(a):
    """A pinned copy of a numpy array (keep the returned PinnedArray alive while its .array is in use)."""
    a = np.ascontiguousarray(a)
    p = PinnedArray(a.shape, a.dtype)
    p.array[...] = a
    return p


class Stream:
    """dsa_stream: batches submitted one after the other, up to `depth` in flight, collected in order.  The arrays given to
    submit() (inputs and `out`) are used as they are - no copies - and must stay alive and untouched until collect()."""

    def __init__(self, device=0, depth=3):
        self.lib = load_library()
        self.h = ctypes.c_void_p()
        rc = self.lib.dsa_stream_create(ctypes.byref(self.h), int(device), int(depth))
        if rc != 0:
            raise DsaError(rc, "dsa_stream_create failed (no HIP device %d?)" % device)
        self._inflight = []

    def close(self):
        if self.h:
            self.lib.dsa_stream_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, rc):
        raise DsaError(rc, self.lib.dsa_stream_last_error(self.h).decode())

    def submit(self, ref_bytes, fusions, read_bytes, pairs, out):
        for a, dt in ((ref_bytes, np.uint8), (fusions, FUSION_DTYPE), (read_bytes, np.uint8), (pairs, PAIR_DTYPE), (out, RECORD_DTYPE)):
            if a.dtype != dt or not a.flags["C_CONTIGUOUS"]:
                raise ValueError("Stream.submit takes contiguous arrays of the C-ABI dtypes (they are not copied)")
        rc = self.lib.dsa_stream_submit(self.h, ref_bytes.ctypes.data, ref_bytes.size, fusions.ctypes.data, len(fusions),
                                        read_bytes.ctypes.data, read_bytes.size, pairs.ctypes.data, len(pairs), out.ctypes.data, len(out))
        if rc != 0:
            self._err(rc)
        self._inflight.append((ref_bytes, fusions, read_bytes, pairs, out))

    def collect(self):
        """Records of the oldest batch: a view of the `out` array of its submit (a fresh array if they did not fit).
        The C side has consumed the oldest batch on every return code except DSA_E_CAPACITY (it stays the oldest until
        recollect took it) and the DSA_E_ARG of "nothing submitted": this side drops its entry in step, so that a caller
        who catches a DsaError and carries on gets the following batches' own arrays."""
        n = ctypes.c_int64(0)
        rc = self.lib.dsa_stream_collect(self.h, ctypes.byref(n))
        if rc == DSA_E_CAPACITY:
            big = np.zeros(n.value, dtype=RECORD_DTYPE)
            rc = self.lib.dsa_stream_recollect(self.h, big.ctypes.data, len(big), ctypes.byref(n))
            if rc == DSA_E_CAPACITY:          # (cannot happen with a buffer of the size just reported; the batch stays)
                self._err(rc)
            self._inflight.pop(0)
            if rc != 0:
                self._err(rc)
            return big
        if rc != 0:
            if self._inflight and rc != DSA_E_ARG:
                self._inflight.pop(0)
            self._err(rc)
        out = self._inflight.pop(0)[4]
        return out[:n.value]
