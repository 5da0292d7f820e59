"""Average-linkage clusterer (tools/HierarchicalClusterer.cpp:46-140): the oracle against
hand-computed cases on the CPU, the HIP path against the oracle on the GPU (clusters, member order, result order)."""
import ctypes

import numpy as np
import pytest

from oracle import hierarchical_oracle as ora


def _points_table(xs):
    xs = np.asarray(xs, dtype=np.float64)
    return np.abs(xs[:, None] - xs[None, :])


def test_oracle_hand_cases():
    assert ora.do_clustering([], 1.0) == []
    assert ora.do_clustering([[0.0]], 1.0) == [[0]]
    # two tight groups on a line: 0,1 | 10,11 ; threshold 5 keeps them apart
    t = _points_table([0, 1, 10, 11])
    assert ora.do_clustering(t, 5.0) == [[0, 1], [2, 3]]
    # a threshold above the average linkage between the groups (10) merges everything; the merged clusters are
    # index 4 = {0,1} and 5 = {2,3}, the last merge lists cluster 4 before cluster 5
    assert ora.do_clustering(t, 10.5) == [[0, 1, 2, 3]]
    assert ora.do_clustering(t, 10.0) == [[0, 1], [2, 3]]          # strict '<' (:69)
    # an untouched item stays in front of the merged clusters (index list order, :124-139)
    t = _points_table([0, 50, 1])
    assert ora.do_clustering(t, 5.0) == [[1], [0, 2]]


def test_oracle_size_weighting():
    # items at 0, 2, 5: merge (0,2) first (d=2); distance of {0,2} to 5 is (1*5 + 1*3)/2 = 4
    t = _points_table([0, 2, 5])
    assert ora.do_clustering(t, 4.0) == [[2], [0, 1]]
    assert ora.do_clustering(t, 4.000001) == [[2, 0, 1]]        # SortedPair(2, 3): item 2 first
    # items at 0, 2, 5, 10: {0,1} is cluster 4; (2,4) merges next at 4 with members of the smaller index (2) first;
    # its distance to item 3 is (1*5 + 2*9)/3 = 7.67 with d({0,1},3) = (10+8)/2 = 9
    t = _points_table([0, 2, 5, 10])
    assert ora.do_clustering(t, 7.6) == [[3], [2, 0, 1]]
    assert ora.do_clustering(t, 7.7) == [[3, 2, 0, 1]]
    # items at 0, 2, 5, 9: d({0,1},2) = 4 ties with d(2,3) = 4, which entered the table first
    t = _points_table([0, 2, 5, 9])
    assert ora.do_clustering(t, 4.5) == [[0, 1], [2, 3]]


def test_oracle_tie_is_first_entered():
    # all distances equal: (0,1) entered first and merges first (a last-entered rule would give [[0, 1, 2]])
    t = np.ones((3, 3))
    assert ora.do_clustering(t, 1.5) == [[2, 0, 1]]
    # with four items (2,3) entered before the distances of the merged cluster 4: clusters 4 and 5 merge last
    t = np.ones((4, 4))
    assert ora.do_clustering(t, 1.5) == [[0, 1, 2, 3]]
    # only the upper triangle is read
    t = np.array([[0, 1, 9], [7, 0, 9], [7, 7, 0]], dtype=float)
    assert ora.do_clustering(t, 2.0) == [[2], [0, 1]]


def test_abi_exports(built):
    from defuse_amd import dsa
    lib = dsa.load_library()
    for sym in ("hc_cluster_batch", "hc_last_error"):
        assert hasattr(lib, sym), sym


def _random_tables(seed, count, nmax, quantised):
    rng = np.random.default_rng(seed)
    tabs, thr = [], []
    for _ in range(count):
        n = int(rng.integers(0, nmax + 1))
        if quantised:        # many exact ties: small integer distances
            t = rng.integers(1, 6, size=(n, n)).astype(np.float64)
        else:                # clustered points in the plane plus noise in the lower triangle (must be ignored)
            k = max(1, n // 6)
            c = rng.uniform(0, 1000, size=(k, 2))
            pts = c[rng.integers(0, k, size=n)] + rng.normal(0, 15, size=(n, 2))
            t = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
            t = np.triu(t) + np.tril(rng.uniform(0, 5, size=(n, n)), -1)
        tabs.append(t)
        thr.append(float(rng.choice([0.0, 2.5, 3.0, 60.0, 200.0, 1e9])))
    return tabs, thr


@pytest.mark.gpu
@pytest.mark.parametrize("seed,count,nmax,quantised", [(1, 40, 40, False), (2, 40, 33, True), (3, 6, 300, False),
                                                       (4, 4, 150, True)])
def test_gpu_matches_oracle(built, seed, count, nmax, quantised):
    from defuse_amd import hc
    tabs, thr = _random_tables(seed, count, nmax, quantised)
    got, timing = hc.cluster_batch(tabs, thr)
    merges = 0
    for p, (t, th) in enumerate(zip(tabs, thr)):
        exp = ora.do_clustering(t, th)
        assert got[p] == exp, "table %d (n=%d, threshold %g)" % (p, len(t), th)
        merges += len(t) - len(exp)
    assert timing.n_merges == merges


@pytest.mark.gpu
def test_gpu_hand_cases(built):
    from defuse_amd import hc
    got, _ = hc.cluster_batch([np.zeros((0, 0)), np.zeros((1, 1)), _points_table([0, 50, 1]), np.ones((4, 4))],
                              [1.0, 1.0, 5.0, 1.5])
    assert got == [[], [[0]], [[1], [0, 2]], [[0, 1, 2, 3]]]
    assert hc.cluster_batch([], [])[0] == []
