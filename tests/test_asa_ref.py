"""AS 136 (k-means) and AS 241 (normal quantile): the C restatement (oracle/mpe_oracle.c) and the Python one
(oracle/clustermatepairs_oracle.py) against the REFERENCE'S OWN asa136.C / asa241.C, compiled as they lie into
oracle/_ref/libasa_ref.so by oracle/Makefile (they include nothing beyond the C++ standard library, so they build
here although the four tools do not).  This is the one part of the clustermatepairs arithmetic that is pinned by
the reference itself.  The file travels prebuilt to the GPU box; without it the tests skip."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def asa(built):
    from oracle import mpe_c
    if mpe_c.ref() is None:
        pytest.skip("oracle/_ref/libasa_ref.so not built (reference sources absent)")
    return mpe_c


def kmns_cases():
    rng = np.random.default_rng(136)
    for trial in range(400):
        m = int(rng.integers(3, 120))
        k = int(rng.integers(2, min(10, m - 1) + 1))
        kind = trial % 5
        if kind == 0:                                    # blobs
            cen = rng.normal(0, 500, size=(k, 2))
            pts = cen[rng.integers(0, k, size=m)] + rng.normal(0, 30, size=(m, 2))
        elif kind == 1:                                  # integer coordinates with duplicates (what the tool feeds)
            pts = rng.integers(-50, 50, size=(m, 2)).astype(np.float64) + np.array([37000.0, -29000.0])
        elif kind == 2:                                  # a line: many equal distances
            pts = np.stack([np.arange(m) * 10.0, np.zeros(m)], axis=1)
        elif kind == 3:                                  # two far groups, k larger than the groups
            pts = np.concatenate([rng.normal(0, 5, size=(m // 2, 2)), rng.normal(10000, 5, size=(m - m // 2, 2))])
        else:
            pts = rng.random(size=(m, 2))
        seeds = pts[rng.choice(m, size=k, replace=False)] if trial % 7 else pts[:k]
        yield m, k, pts, seeds


def test_kmns_restatements_equal_the_reference(asa):
    from oracle import clustermatepairs_oracle as o
    n_checked = n_fault1 = 0
    for m, k, pts, seeds in kmns_cases():
        a = np.concatenate([pts[:, 0], pts[:, 1]])       # column-major a[i + j*m]
        c = np.concatenate([seeds[:, 0], seeds[:, 1]])
        r_ic1, r_nc, r_wss, r_c, r_fault = asa.ref_kmns(a, m, 2, c, k)
        o_ic1, o_nc, o_wss, o_c, o_fault = asa.kmns(a, m, 2, c, k)
        assert o_fault == r_fault
        if r_fault == 1:                                 # a seed attracted no point: outputs are not defined beyond ic1 / nc
            n_fault1 += 1
            assert (o_ic1 == r_ic1).all() and (o_nc == r_nc).all()
            continue
        assert (o_ic1 == r_ic1).all() and (o_nc == r_nc).all()
        assert o_c.tobytes() == r_c.tobytes() and o_wss.tobytes() == r_wss.tobytes()      # bit for bit
        if m <= 40:                                      # the Python restatement is slow
            cc = list(c)
            p_ic1, p_nc, p_wss, p_fault = o.kmns(list(a), m, 2, cc, k, 1000)
            assert p_fault == r_fault and p_ic1 == list(r_ic1) and p_nc == list(r_nc)
            assert np.array(cc).tobytes() == r_c.tobytes() and np.array(p_wss).tobytes() == r_wss.tobytes()
        n_checked += 1
    assert n_checked > 300
    # k <= 1 and m <= k: ifault 3
    assert asa.ref_kmns(np.zeros(4), 2, 2, np.zeros(4), 2)[4] == 3 and asa.kmns(np.zeros(4), 2, 2, np.zeros(4), 2)[4] == 3
    assert asa.ref_kmns(np.zeros(8), 4, 2, np.zeros(2), 1)[4] == 3 and asa.kmns(np.zeros(8), 4, 2, np.zeros(2), 1)[4] == 3


def test_cdf_inverse_equals_the_reference(asa):
    from oracle import clustermatepairs_oracle as o
    ref = asa.ref()._Z24r8_normal_01_cdf_inversed
    rng = np.random.default_rng(241)
    ps = list(rng.random(20000)) + [0.025, 0.005, 0.0005, 1e-10, 1e-300, 1 - 1e-12, 0.075, 0.925, 0.5, 0.0, 1.0, -1.0, 2.0,
                                   (1 - 0.95) / 2, (1 - 0.99) / 2, (1 - 0.9) / 2]
    for p in ps:
        r = ref(p)
        assert asa.lib().ora_cdf_inverse(p) == r
        assert o.normal_01_cdf_inverse(p) == r


def test_cdf_inverse_on_the_reference_table(asa):
    """asa241.C carries its own known-answer table (normal_01_cdf_values, :13-113): x and CDF(x) to 16 digits.  The inverse
    must bring every tabulated CDF value back to its x."""
    import ctypes as C
    n, x, fx = C.c_int(0), C.c_double(0), C.c_double(0)
    seen = 0
    while True:
        asa.ref()._Z20normal_01_cdf_valuesPiPdS0_(C.byref(n), C.byref(x), C.byref(fx))
        if n.value == 0:
            break
        seen += 1
        assert abs(asa.lib().ora_cdf_inverse(fx.value) - x.value) < 1e-13 * max(1.0, abs(x.value)) + 2e-14 / max(1e-3, fx.value * (1 - fx.value))
    assert seen >= 10


def test_min_probability_of_the_tool():
    """mMinProbability (tools/MatePairEM.cpp:49-50) as the tool computes it on the host, against both restatements."""
    from oracle import clustermatepairs_oracle as o, mpe_c
    for sd, prec in ((30.0, 0.95), (45.0, 0.95), (30.0, 0.99), (12.5, 0.5)):
        assert mpe_c.lib().ora_min_probability(sd, prec) == o.MatePairEM(300.0, sd, prec, 5).min_prob
