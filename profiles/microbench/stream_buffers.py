"""Per-batch time of a dsa_stream against the kind of host memory its buffers live in: pinned (dsa_host_alloc), ordinary
pageable (numpy), a shared anonymous mapping (what bin/dosplitalign shares with its worker process), the same registered with
dsa_host_register.  Usage: python profiles/microbench/stream_buffers.py [pairs per batch]"""
import ctypes
import mmap
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from defuse_amd import dsa, synth

npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
batch = dsa._check_arrays(*synth.make_batch(npairs // 100, 100, lq=76, lr=389, seed=3))
lib = dsa.load_library()
st = dsa.Stream(0, depth=3)
cap = 2 * len(batch[3]) + 1024
keep = []


def shared(a, register):
    m = mmap.mmap(-1, max(a.nbytes, 1))                       # MAP_SHARED | MAP_ANONYMOUS
    keep.append(m)
    v = np.frombuffer(m, dtype=np.uint8, count=a.nbytes).view(a.dtype).reshape(a.shape)
    v[...] = a
    if register:
        rc = lib.dsa_host_register(ctypes.c_void_p(v.ctypes.data), a.nbytes)
        assert rc == 0, rc
    return v


kinds = {
    "pinned (dsa_host_alloc)": lambda a: keep.append(dsa.pinned_copy(a)) or keep[-1].array,
    "pageable (numpy)": lambda a: a.copy(),
    "shared anonymous mapping": lambda a: shared(a, False),
    "shared anonymous mapping, dsa_host_register": lambda a: shared(a, True),
}
for name, make in kinds.items():
    slots = [[make(a) for a in batch] + [make(np.zeros(cap, dtype=dsa.RECORD_DTYPE))] for _ in range(3)]
    for rnd in range(3):
        t0 = time.perf_counter()
        sub = 0
        for k in range(9):
            while sub < 9 and sub - k < 3:
                st.submit(*slots[sub % 3])
                sub += 1
            n = len(st.collect())
        dt = (time.perf_counter() - t0) / 9
        if rnd:
            print("%-46s round %d: %.2f ms per batch of %d pairs (%d records)" % (name, rnd, dt * 1e3, len(batch[3]), n))
# one batch at a time (nothing overlapped): what a batch costs from submit to records
for name in ("pinned (dsa_host_alloc)", "shared anonymous mapping"):
    slot = [kinds[name](a) for a in batch] + [kinds[name](np.zeros(cap, dtype=dsa.RECORD_DTYPE))]
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        st.submit(*slot)
        st.collect()
        ts.append(time.perf_counter() - t0)
    print("%-46s one at a time: %s ms" % (name, " ".join("%.2f" % (t * 1e3) for t in ts)))
