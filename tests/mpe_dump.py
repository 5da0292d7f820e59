"""Reads the file `DEFUSE_CMP_DUMP_EM=path bin/clustermatepairs ...` writes: exactly the arrays the tool hands to
mpe_cluster_batch (include/defuse_mpe.h).  Test infrastructure."""
import numpy as np


def read_em_dump(path):
    raw = np.fromfile(path, dtype=np.uint8)
    n_prob, n_mp = (int(v) for v in raw[:16].view(np.int64))
    prm = raw[16:48]
    mean, sd, min_prob = (float(v) for v in prm[:24].view(np.float64))
    min_size = int(prm[24:28].view(np.int32)[0])
    o = 48
    def take(dtype, n):
        nonlocal o
        nb = np.dtype(dtype).itemsize * n
        v = raw[o:o + nb].view(dtype).copy()
        o += nb
        return v
    prob_off = take(np.int64, n_prob + 1)
    x, y, u = take(np.float64, n_mp), take(np.float64, n_mp), take(np.float64, n_mp)
    to_xo, to_yo = take(np.int32, n_mp), take(np.int32, n_mp)
    assert o == len(raw) and int(prob_off[-1]) == n_mp
    return dict(mean=mean, sd=sd, min_prob=min_prob, min_size=min_size, prob_off=prob_off, x=x, y=y, u=u, to_xo=to_xo, to_yo=to_yo)
