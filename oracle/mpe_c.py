"""ctypes bindings of oracle/libmpe_oracle.so (the C restatement of MatePairEM + AS 136 + AS 241) and, when it
has been built, of oracle/_ref/libasa_ref.so (the reference's own asa136.C / asa241.C compiled as they lie).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmpe_oracle.so")
REF = os.path.join(HERE, "_ref", "libasa_ref.so")


class Params(C.Structure):
    _fields_ = [("fragment_mean", C.c_double), ("fragment_stddev", C.c_double), ("min_probability", C.c_double),
                ("min_cluster_size", C.c_int32), ("pad_", C.c_int32)]


class Diag(C.Structure):
    _fields_ = [("min_prob_margin", C.c_double), ("min_tol_margin", C.c_double), ("min_bic_gap", C.c_double),
                ("min_deriv_margin", C.c_double), ("min_merge_margin", C.c_double),
                ("nk_zero", C.c_int64), ("nk_zero_first_iter", C.c_int64), ("ll_underflow", C.c_int64), ("kkz_fail", C.c_int64),
                ("em_iterations", C.c_int64), ("merge_equal", C.c_int64), ("all_k_failed", C.c_int64),
                ("min_deriv_margin_first", C.c_double), ("min_merge_margin_first", C.c_double), ("deriv_zero", C.c_int64),
                ("iters_by_k", C.c_int64 * 12), ("ll_by_k", C.c_double * 12)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n in ("iters_by_k", "ll_by_k") else getattr(self, n)) for n, _ in self._fields_}


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("oracle/libmpe_oracle.so is not built (make -C oracle)")
        _lib = C.CDLL(LIB)
        _lib.ora_cdf_inverse.restype = C.c_double
        _lib.ora_cdf_inverse.argtypes = [C.c_double]
        _lib.ora_normalpdf.restype = C.c_double
        _lib.ora_normalpdf.argtypes = [C.c_double] * 3
        _lib.ora_min_probability.restype = C.c_double
        _lib.ora_min_probability.argtypes = [C.c_double] * 2
        _lib.ora_kmns.restype = None
        _lib.ora_kmns.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ora_mpe_cluster_batch.restype = C.c_int
        _lib.ora_mpe_cluster_batch.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int32] + [C.c_void_p] * 5 + [C.c_void_p] * 3 + \
            [C.POINTER(Diag), C.c_void_p]
    return _lib


def ref():
    """The reference's own AS 136 / AS 241 objects, or None when oracle/_ref was not built."""
    global _ref
    if _ref is None and os.path.exists(REF):
        _ref = C.CDLL(REF)
        _ref._Z4kmnsPdiiS_iPiS0_iS_S0_.restype = None          # kmns(double*,int,int,double*,int,int*,int*,int,double*,int*)
        _ref._Z4kmnsPdiiS_iPiS0_iS_S0_.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_void_p, C.c_void_p]
        _ref._Z24r8_normal_01_cdf_inversed.restype = C.c_double   # r8_normal_01_cdf_inverse(double)
        _ref._Z24r8_normal_01_cdf_inversed.argtypes = [C.c_double]
        _ref._Z20normal_01_cdf_valuesPiPdS0_.restype = None       # normal_01_cdf_values(int*,double*,double*)
        _ref._Z20normal_01_cdf_valuesPiPdS0_.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return _ref


def _kmns(fn, a, m, n, c, k, iters):
    a = np.ascontiguousarray(a, dtype=np.float64).copy()
    c = np.ascontiguousarray(c, dtype=np.float64).copy()
    ic1 = np.zeros(m, dtype=np.int32)
    nc = np.zeros(max(k, 1), dtype=np.int32)
    wss = np.zeros(max(k, 1), dtype=np.float64)
    ifault = C.c_int(0)
    fn(a.ctypes.data, m, n, c.ctypes.data, k, ic1.ctypes.data, nc.ctypes.data, iters, wss.ctypes.data, C.byref(ifault))
    return ic1, nc, wss, c, ifault.value


def kmns(a, m, n, c, k, iters=1000):
    return _kmns(lib().ora_kmns, a, m, n, c, k, iters)


def ref_kmns(a, m, n, c, k, iters=1000):
    return _kmns(ref()._Z4kmnsPdiiS_iPiS0_iS_S0_, a, m, n, c, k, iters)


def ranks_desc(v):
    """Rank of every element when sorted by value descending, ties by index ascending (SURVEY 8(c) canonical order)."""
    order = np.lexsort((np.arange(len(v)), -np.asarray(v, dtype=np.float64)))
    r = np.empty(len(v), dtype=np.int32)
    r[order] = np.arange(len(v), dtype=np.int32)
    return r


def cluster_batch(mean, sd, min_prob, min_size, prob_off, x, y, u, to_xo, to_yo, per_problem_diag=False):
    prob_off = np.ascontiguousarray(prob_off, dtype=np.int64)
    n = len(prob_off) - 1
    x, y, u = (np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, u))
    to_xo, to_yo = (np.ascontiguousarray(v, dtype=np.int32) for v in (to_xo, to_yo))
    total = int(prob_off[-1])
    n_clusters = np.zeros(n, dtype=np.int32)
    member = np.zeros(max(total, 1), dtype=np.uint16)
    status = np.zeros(n, dtype=np.int32)
    prm = Params(mean, sd, min_prob, int(min_size), 0)
    dg = Diag()
    pd = (Diag * n)() if per_problem_diag else None
    rc = lib().ora_mpe_cluster_batch(C.byref(prm), prob_off.ctypes.data, n, x.ctypes.data, y.ctypes.data, u.ctypes.data,
                                     to_xo.ctypes.data, to_yo.ctypes.data, n_clusters.ctypes.data, member.ctypes.data,
                                     status.ctypes.data, C.byref(dg), C.addressof(pd) if pd is not None else None)
    assert rc == 0
    return n_clusters, member[:total], status, dg, pd


def do_clustering(mean, sd, min_prob, min_size, mate_pairs):
    """One problem, same result shape as MatePairEM.do_clustering of clustermatepairs_oracle.py: list of member lists.
    mate_pairs: list of ((s1, e1), (s2, e2)), strand-remapped."""
    n = len(mate_pairs)
    if n < float(min_size):
        return []
    x = np.array([mp[0][1] for mp in mate_pairs], dtype=np.float64)
    y = np.array([mp[1][1] for mp in mate_pairs], dtype=np.float64)
    u = np.array([mean - (mp[0][1] - mp[0][0] + 1) - (mp[1][1] - mp[1][0] + 1) for mp in mate_pairs], dtype=np.float64)
    ncl, member, status, _, _ = cluster_batch(mean, sd, min_prob, min_size, [0, n], x, y, u, ranks_desc(x), ranks_desc(y))
    if status[0]:
        raise SystemExit("Error: a DebugCheck of the reference fired")
    return [[i for i in range(n) if member[i] >> j & 1] for j in range(int(ncl[0]))]
