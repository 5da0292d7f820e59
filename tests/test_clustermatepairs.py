"""clustermatepairs: oracle behaviour (CPU) and the drop-in binary against the oracle (GPU)."""
import os
import subprocess

import pytest

from tests import cmp_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "bin", "clustermatepairs")


def test_oracle_two_loci_shape():
    """SURVEY.md Appendix A's mini check: two loci (12 + 8 fragments) give two clusters, and cluster 1's
    end 0 is the chrB+ side because RefBinPacked ids order strand above the reference index."""
    from oracle import clustermatepairs_oracle as o
    txt, n = o.clustermatepairs(cmp_cases.two_loci(), 300, 30, 0.95, 5)
    assert n == 2
    lines = [l.split("\t") for l in txt.splitlines()]
    assert {l[0] for l in lines} == {"0", "1"} and len(lines) % 2 == 0 and len(lines) >= 34
    first_c1 = [l for l in lines if l[0] == "1" and l[1] == "0"][0]
    assert first_c1[4:6] == ["chrB", "+"]
    # the read-end column is inverted relative to the input (AlignmentStream.cpp:181), SURVEY a-8
    assert [l for l in lines if l[0] == "0" and l[1] == "0"][0][3] == "1"


def test_oracle_kmns_and_cdf_inverse():
    from oracle import clustermatepairs_oracle as o
    assert abs(o.normal_01_cdf_inverse(0.025) + 1.959963984540054) < 1e-12
    a = [0.0, 0.1, 0.2, 10.0, 10.1, 10.2] + [5.0, 5.1, 4.9, 1.0, 1.1, 0.9]      # column-major, n = 2
    ic1, nc, wss, ifault = o.kmns(a, 6, 2, [0.0, 10.0, 5.0, 1.0], 2, 100)
    assert ifault == 0 and ic1 == [1, 1, 1, 2, 2, 2] and nc == [3, 3]


def test_oracle_concordant_and_small_clusters_dropped():
    from oracle import clustermatepairs_oracle as o
    lines = ["0\t0\tchr1\t+\t1000\t1049\n", "0\t1\tchr1\t-\t1200\t1249\n"] * 1
    assert o.clustermatepairs(lines, 300, 30, 0.95, 5) == ("", 0)
    txt, n = o.clustermatepairs(cmp_cases.two_loci(), 300, 30, 0.95, 9)     # the 8-fragment locus is below -m 9
    assert n == 1


def test_cli(built):
    from defuse_amd import build
    build.build_tools()
    r = subprocess.run([TOOL, "-a", "x"], capture_output=True, text=True)
    assert r.returncode == 1 and "One or more required arguments missing!" in r.stderr
    r = subprocess.run([TOOL, "--help"], capture_output=True, text=True)
    assert "Mate Pair Clustering Tool" in r.stdout


def run_tool(lines, tmp_path, m=5, stdin=False, env=None):
    p = tmp_path / "spanning.txt"
    p.write_text("".join(lines))
    out = tmp_path / "clusters.txt"
    args = [TOOL, "-a", "-" if stdin else str(p), "-c", str(out), "-u", "300", "-s", "30", "-p", "0.95", "-m", str(m)]
    r = subprocess.run(args, capture_output=True, text=True, input="".join(lines) if stdin else None,
                       env=dict(os.environ, **env) if env else None)
    return r, out.read_text() if out.exists() else None


@pytest.mark.gpu
def test_tool_two_loci(built, tmp_path):
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    lines = cmp_cases.two_loci()
    r, txt = run_tool(lines, tmp_path, stdin=True)                    # the pipeline pipes `cat files |` into -a -
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5)
    assert txt == exp
    assert r.stdout == ("Finding pairs of reference sequences connected by pairs of alignments\nInitializing clusterer\n"
                        "Creating clusters\nCreated %d clusters\n" % n)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4, 5])
def test_tool_matches_oracle(built, tmp_path, seed):
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    lines = cmp_cases.many_loci(seed)
    r, txt = run_tool(lines, tmp_path)
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5)
    assert n >= 8
    assert txt == exp
    # the same run with the EM workspaces cut into many chunks
    r, txt = run_tool(lines, tmp_path, env={"DEFUSE_MPE_SCRATCH_MB": "1"})
    assert r.returncode == 0, r.stderr
    assert txt == exp
    # split by problem size / every fit in one lane (the default gives every fit a wave of its own)
    for wave_min in ("40", "1000000000"):
        r, txt = run_tool(lines, tmp_path, env={"DEFUSE_MPE_WAVE_MIN": wave_min})
        assert r.returncode == 0, r.stderr
        assert txt == exp, "DEFUSE_MPE_WAVE_MIN=" + wave_min


@pytest.mark.gpu
def test_tool_large_bin_pairs(built, tmp_path):
    """Bigger problems (hundreds of mate pairs per bin pair, several mixtures) and -m 3."""
    import numpy as np
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    rng = np.random.default_rng(11)
    lines = []
    frag = 0
    for (n, ba, bb) in ((220, 40000, 90000), (150, 40120, 90060), (90, 41000, 90500), (60, 200000, 300000)):
        lines += cmp_cases.locus_fragments(rng, frag, n, "chr1", "+", ba, "chr2", "-", bb)
        frag += n
    r, txt = run_tool(lines, tmp_path, m=3)
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 3)
    assert n >= 4 and txt == exp
    for wave_min in ("100", "1000000000"):
        r, txt = run_tool(lines, tmp_path, m=3, env={"DEFUSE_MPE_WAVE_MIN": wave_min})
        assert r.returncode == 0, r.stderr
        assert txt == exp, "DEFUSE_MPE_WAVE_MIN=" + wave_min


@pytest.mark.parametrize("seed", [3, 6])
def test_host_stages_same_for_every_thread_count(built, tmp_path, seed):
    """No GPU needed: DEFUSE_CMP_DUMP_PROBLEMS writes what the host stages hand to the device (bin pairs in canonical order,
    mate pair coordinates, sort ranks, alignment tables) and stops — here with the host transcription of the bin-pair
    bucketing (DEFUSE_CMP_HOST_BINNING=1, the cross-check of the device path; test_device_bin_pairs_* hold the two against
    each other on a GPU).  Pieces cut at fragment boundaries, parsed and binned side by side, must give the bytes a single
    reader gives — also when there are more threads than fragments."""
    from defuse_amd import build
    build.build_tools()
    lines = cmp_cases.many_loci(seed)
    dumps = []
    for threads in ("1", "2", "3", "7", "16", "100"):
        dump = tmp_path / ("dump." + threads)
        r, _ = run_tool(lines, tmp_path, env={"DEFUSE_THREADS": threads, "DEFUSE_CMP_DUMP_PROBLEMS": str(dump), "DEFUSE_CMP_HOST_BINNING": "1"})
        assert r.returncode == 0, r.stderr
        dumps.append(dump.read_bytes())
    assert len(dumps[0]) > 1000
    assert all(d == dumps[0] for d in dumps[1:])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,threads", [(3, "1"), (6, "7"), (12, "16")])
def test_device_bin_pairs_give_the_host_cross_checks_problems(built, tmp_path, seed, threads):
    """The bin pairs built on the GPU (the default: include/defuse_cmp.h — concordance filter, AddBinPairs, a stable radix
    sort by bin-pair key) against the host transcription of tools/clustermatepairs.cpp:211-290 (DEFUSE_CMP_HOST_BINNING=1):
    everything downstream of them — problems, mate pair coordinates, ranks, alignment tables — dumped by
    DEFUSE_CMP_DUMP_PROBLEMS, byte for byte; on loci with multi-mapping ends and on the 20 000-fragment config-3 sample."""
    from defuse_amd import build
    build.build_tools()
    for name, lines in (("loci", cmp_cases.many_loci(seed)), ("config3", None)):
        if lines is None:
            path = tmp_path / "c3.txt"
            cmp_cases.config3_write(20000, str(path))
            lines = open(path).read().splitlines(True)
        dumps = {}
        for mode in ("device", "host"):
            dump = tmp_path / ("dump.%s.%s" % (name, mode))
            env = {"DEFUSE_THREADS": threads, "DEFUSE_CMP_DUMP_PROBLEMS": str(dump)}
            if mode == "host":
                env["DEFUSE_CMP_HOST_BINNING"] = "1"
            r, _ = run_tool(lines, tmp_path, env=env)
            assert r.returncode == 0, r.stderr
            dumps[mode] = dump.read_bytes()
        assert len(dumps["host"]) > 1000 and dumps["device"] == dumps["host"], name


@pytest.mark.gpu
def test_device_bin_pairs_through_the_c_abi(built):
    """cmp_bin_* (ctypes) on fragments with dozens of alignments per end, both ends in the same bins, alignments over several
    bins and concordant pairs: keys, offsets and both lists of every bin pair equal the oracle's map (CheckConcordant +
    AddBinPairs, oracle/clustermatepairs_oracle.py:add_fragment), list by list in the reference's order of appends; the
    records uploaded in three pieces; a stop condition comes back as the first offending alignment in file order."""
    import ctypes
    import numpy as np
    from defuse_amd import build
    from tests import test_cmp_bins as tb
    lib = ctypes.CDLL(build.build_lib())
    vp, i64 = ctypes.c_void_p, ctypes.c_int64

    class Stats(ctypes.Structure):
        _fields_ = [("n_fragments", i64), ("n_concordant", i64), ("n_keys", i64), ("n_first", i64), ("n_second", i64), ("err_record", i64),
                    ("err_kind", ctypes.c_int32), ("err_value", ctypes.c_int32), ("device_ms", ctypes.c_float), ("pad_", ctypes.c_float)]
    lib.cmp_bin_create.argtypes = [ctypes.POINTER(vp), ctypes.c_int]
    lib.cmp_bin_destroy.argtypes = [vp]
    lib.cmp_bin_reserve.argtypes = [vp, i64, i64]
    lib.cmp_bin_upload_records.argtypes = [vp, vp, i64, i64]
    lib.cmp_bin_upload_fragments.argtypes = [vp, vp, i64, i64]
    lib.cmp_bin_run.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(Stats)]
    lib.cmp_bin_fetch.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.cmp_last_error.restype = ctypes.c_char_p
    h = vp()
    assert lib.cmp_bin_create(ctypes.byref(h), 0) == 0

    def run(frs, mfr):
        recs, starts = tb.to_records(frs)
        assert lib.cmp_bin_reserve(h, len(recs), len(starts) - 1) == 0
        cut = [0, len(recs) // 3, 2 * len(recs) // 3, len(recs)]
        for a, b in zip(cut, cut[1:]):
            part = np.ascontiguousarray(recs[a:b])
            assert lib.cmp_bin_upload_records(h, part.ctypes.data, len(part), a) == 0
        assert lib.cmp_bin_upload_fragments(h, starts.ctypes.data, len(starts), 0) == 0
        st = Stats()
        assert lib.cmp_bin_run(h, mfr, ctypes.byref(st)) == 0, lib.cmp_last_error()
        return st, starts

    for seed, mfr in ((1, 600), (5, 1500), (7, 250)):
        frs = tb.random_fragments(seed, 3000)
        st, _ = run(frs, mfr)
        exp, conc = tb.oracle_bin_pairs(frs, mfr)
        assert st.err_record == -1 and st.n_concordant == conc and st.n_keys == len(exp)
        keys = np.zeros(st.n_keys, dtype=np.uint64)
        off1, off2 = np.zeros(st.n_keys + 1, dtype=np.int64), np.zeros(st.n_keys + 1, dtype=np.int64)
        p1, p2 = np.zeros(st.n_first, dtype=tb.PACKED), np.zeros(st.n_second, dtype=tb.PACKED)
        assert lib.cmp_bin_fetch(h, keys.ctypes.data, off1.ctypes.data, off2.ctypes.data, p1.ctypes.data, p2.ctypes.data) == 0
        assert list(keys) == sorted(keys) and len(set(keys.tolist())) == len(keys)
        got = tb.as_lists(keys, off1, off2, p1.tolist(), p2.tolist())
        assert got == {k: (v[0], v[1]) for k, v in exp.items()}
    frs = tb.random_fragments(9, 50)
    frs[20][0]["ref"] = (1 << 18) + 5
    frs[30][0]["region"] = ((1 << 13) * 32768 + 10, (1 << 13) * 32768 + 80)
    st, starts = run(frs, 600)
    from oracle import clustermatepairs_oracle as ora
    first = None
    for k, als in enumerate(frs):
        try:
            ora.add_fragment(als, 600, {})
        except SystemExit as e:
            first = (k, 2 if "too many reference" in str(e) else 3)
            break
    assert first and st.err_kind == first[1] and int(starts[first[0]]) <= st.err_record < int(starts[first[0] + 1])
    # an empty input
    assert lib.cmp_bin_reserve(h, 0, 0) == 0
    z = np.zeros(1, dtype=np.uint32)
    assert lib.cmp_bin_upload_fragments(h, z.ctypes.data, 1, 0) == 0
    st = Stats()
    assert lib.cmp_bin_run(h, 600, ctypes.byref(st)) == 0 and st.n_keys == 0
    lib.cmp_bin_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["1", "5"])
def test_tool_matches_oracle_threads(built, tmp_path, threads):
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    lines = cmp_cases.many_loci(8)
    r, txt = run_tool(lines, tmp_path, env={"DEFUSE_THREADS": threads})
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5)
    assert n >= 5 and txt == exp


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", ["0,0,0", "2", "all"])
def test_tool_shares_bin_pairs_over_devices(built, tmp_path, gpus):
    """DEFUSE_GPUS: the bin pairs go to several devices in contiguous shares (here every share lands on the box's one GPU,
    from host threads of their own); the cluster file must not depend on the shares."""
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    lines = cmp_cases.many_loci(9)
    r, txt = run_tool(lines, tmp_path, env={"DEFUSE_GPUS": gpus, "DEFUSE_TIMING": "1"})
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5)
    assert n >= 5 and txt == exp
    shares = {"0,0,0": "3 device share(s)", "2": "2 device share(s)", "all": "device share(s)"}[gpus]
    assert shares in r.stderr


@pytest.mark.gpu
def test_tool_degenerate_bin_pairs(built, tmp_path):
    """Bin pairs whose mate pairs coincide: the KKZ seeding runs out of distinct points (SelectKKZ returns false for every K
    beyond the number of distinct points, tools/MatePairEM.cpp:327-386), so only some of the K fits have a likelihood."""
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    lines = []
    frag = 0
    for (n, a, b) in ((9, 5000, 9000), (7, 40000, 52000)):                   # all mate pairs identical
        for _ in range(n):
            lines.append("%d\t0\tchr1\t+\t%d\t%d\n" % (frag, a, a + 49))
            lines.append("%d\t1\tchr2\t-\t%d\t%d\n" % (frag, b, b + 49))
            frag += 1
    for k in range(12):                                                       # two distinct mate pairs, six times each
        a, b = (80000, 91000) if k % 2 else (80030, 91010)
        lines.append("%d\t0\tchr1\t+\t%d\t%d\n" % (frag, a, a + 49))
        lines.append("%d\t1\tchr2\t-\t%d\t%d\n" % (frag, b, b + 49))
        frag += 1
    for k in range(10):                                                       # three distinct ones
        a, b = [(120000, 131000), (120040, 131020), (120090, 131070)][k % 3]
        lines.append("%d\t0\tchr1\t-\t%d\t%d\n" % (frag, a, a + 49))
        lines.append("%d\t1\tchr2\t+\t%d\t%d\n" % (frag, b, b + 49))
        frag += 1
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5)
    assert n >= 3
    for wave_min in (None, "1000000000"):
        r, txt = run_tool(lines, tmp_path, env={"DEFUSE_MPE_WAVE_MIN": wave_min} if wave_min else None)
        assert r.returncode == 0, r.stderr
        assert txt == exp, "DEFUSE_MPE_WAVE_MIN=%s" % wave_min


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] (2x100 bp, Zipf support 1..500, 5 % multi-mapping, 10 % decoys)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [3, 4, 5])
def test_c_em_equals_python_em(built, seed):
    """The two restatements of MatePairEM (oracle/mpe_oracle.c, oracle/clustermatepairs_oracle.py) give the same file."""
    from oracle import clustermatepairs_oracle as o
    lines = cmp_cases.many_loci(seed)
    assert o.clustermatepairs(lines, 300, 30, 0.95, 5, em="c") == o.clustermatepairs(lines, 300, 30, 0.95, 5)


def test_c_em_equals_python_em_on_a_config3_sample(built):
    from oracle import clustermatepairs_oracle as o
    lines = cmp_cases.config3_lines(2500)
    a = o.clustermatepairs(lines, 300, 30, 0.95, 5, em="c")
    assert a == o.clustermatepairs(lines, 300, 30, 0.95, 5) and a[1] >= 20


@pytest.mark.gpu
def test_config3_sample_through_clustermatepairs_and_setcover(built, tmp_path):
    """configs[2] at 40 000 fragments (about 490 loci, bin pairs of up to 500 mate pairs, K up to 10, multi-mappers, decoys):
    bin/clustermatepairs then bin/setcover, both files byte for byte against the oracles."""
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o, setcover_oracle as so
    build.build_tools()
    lines = cmp_cases.config3_lines(40000)
    r, txt = run_tool(lines, tmp_path, env={"DEFUSE_TIMING": "1"})
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5, em="c")
    assert n > 500 and txt == exp
    assert "Created %d clusters" % n in r.stdout
    sizes = {}
    for l in exp.splitlines():
        f = l.split("\t", 2)
        if f[1] == "0":
            sizes[f[0]] = sizes.get(f[0], 0) + 1
    assert max(sizes.values()) >= 150                                   # the large loci are there (EM splits them into components)
    cl, sc = tmp_path / "clusters.txt", tmp_path / "clusters.sc"
    r = subprocess.run([os.path.join(ROOT, "bin", "setcover"), "-c", str(cl), "-m", "5", "-o", str(sc)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cover = so.setcover(str(cl), 5)
    assert sc.read_text() == cover and len(cover.splitlines()) > 10000


@pytest.mark.gpu
def test_config3_one_million_fragments_em_against_the_c_oracle(built, tmp_path):
    """configs[2] at 1 M fragments: the host stages of bin/clustermatepairs write the arrays they would hand to the device
    (DEFUSE_CMP_DUMP_EM), then mpe_cluster_batch (HIP, through the C ABI) and the C restatement run on exactly those arrays:
    about 9 000 bin pairs, 1 M mate pairs, 2.5 M EM iterations; every membership bit must agree.  The restatement also
    reports how close the run came to the decisions a last-ulp difference between ocml's and glibc's exp/log could flip."""
    from defuse_amd import build, mpe
    from oracle import mpe_c
    from tests.mpe_dump import read_em_dump
    build.build_tools()
    span, dump = tmp_path / "spanning.txt", tmp_path / "em.bin"
    cmp_cases.config3_write(1_000_000, str(span))
    r = subprocess.run([TOOL, "-a", str(span), "-c", str(tmp_path / "unused"), "-u", "300", "-s", "30", "-p", "0.95", "-m", "5"],
                       capture_output=True, text=True, env=dict(os.environ, DEFUSE_CMP_DUMP_EM=str(dump)))
    assert r.returncode == 0, r.stderr
    d = read_em_dump(str(dump))
    assert len(d["prob_off"]) - 1 > 5000 and len(d["x"]) > 500000
    args = (d["mean"], d["sd"], d["min_prob"], d["min_size"], d["prob_off"], d["x"], d["y"], d["u"], d["to_xo"], d["to_yo"])
    its = tmp_path / "iters.bin"
    os.environ["DEFUSE_MPE_DUMP_ITERS"] = str(its)
    try:
        g_ncl, g_member, g_status, t = mpe.cluster_batch(*args)
    finally:
        del os.environ["DEFUSE_MPE_DUMP_ITERS"]
    o_ncl, o_member, o_status, dg, pd = mpe_c.cluster_batch(*args, per_problem_diag=True)
    assert not o_status.any() and not g_status.any()
    assert (g_ncl == o_ncl).all()
    assert g_member.tobytes() == o_member.tobytes()
    print("config-3 1M EM margins:", {k: v for k, v in dg.as_dict().items() if not k.endswith("_by_k")}, "kernel %.1f ms, device EM iterations %d, "
          "restatement %d" % (t.kernel_ms, t.em_iterations, dg.em_iterations))
    # The iteration counts, fit by fit (round-2 verdict: 2 542 480 against 2 542 492 was "not analysed").  Both count every
    # likelihood evaluation of every fit plus the refit's.  They differ in a handful of fits, every one of them a fit with
    # MORE components than the K the problem ends up with: there the surplus components share mate pairs with
    # responsibilities at rounding level, the M step's exact comparisons (the two inner knife edges) fall
    # the other way under ocml's exp / log than under glibc's, and the fit walks a slightly different way along a flat
    # likelihood ridge until |dLL| < 0.001 stops it - a few iterations earlier or later, at a log-likelihood that differs in the
    # fourth digit.  Such a fit loses the BIC comparison by its 2 ln N per surplus component either way: the chosen K, the
    # iterations of its fit and of its refit, and every membership bit are the same.
    import numpy as np
    n = len(d["prob_off"]) - 1
    raw = np.fromfile(str(its), dtype=np.int64)
    dev, dev_ll = raw[:n * 12].reshape(n, 12), raw[n * 12:].view(np.float64).reshape(n, 12)
    ora = np.array([list(pd[p].iters_by_k) for p in range(n)], dtype=np.int64)
    ora_ll = np.array([list(pd[p].ll_by_k) for p in range(n)], dtype=np.float64)
    assert int(dev[:, 1:].sum()) == t.em_iterations and int(ora[:, 1:].sum()) == dg.em_iterations
    assert (dev[:, 0] == ora[:, 0]).all()                                       # the chosen K
    assert (dev[:, 11] == ora[:, 11]).all()                                     # the refit
    differ = [(p, k) for p in range(n) for k in range(1, 11) if dev[p, k] != ora[p, k]]
    assert all(k > dev[p, 0] for p, k in differ), differ                        # only fits with surplus components
    assert sum(int(dev[p, k] - ora[p, k]) for p, k in differ) == t.em_iterations - dg.em_iterations    # ... explain the whole difference
    assert len({p for p, _ in differ}) <= n // 200                              # a handful of 9 000 problems
    same = np.ones((n, 12), dtype=bool)
    for p, k in differ:
        same[p, k] = False
    both = same & (dev_ll != 0) & (ora_ll != 0)
    both[:, 0] = both[:, 11] = False
    rel_same = (np.abs(dev_ll - ora_ll)[both] / np.abs(ora_ll[both])).max()
    rel_diff = max((abs(dev_ll[p, k] - ora_ll[p, k]) / abs(ora_ll[p, k]) for p, k in differ if dev_ll[p, k] and ora_ll[p, k]), default=0.0)
    print("fits whose iteration counts differ: %d in %d problems (device - restatement = %d iterations); log-likelihood at the end, largest "
          "relative difference: %.2e over those fits, %.2e over all the others" % (len(differ), len({p for p, _ in differ}),
                                                                                t.em_iterations - dg.em_iterations, rel_diff, rel_same))
    assert rel_diff < 1e-2
    # the knife edges: all decisions are far from a last-ulp flip on this workload
    assert dg.nk_zero_first_iter == 0 and dg.all_k_failed == 0
    assert dg.min_prob_margin > 1e-9 and dg.min_tol_margin > 1e-9 and dg.min_bic_gap > 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", ["1", "3", "50"])
def test_tool_chunks_give_the_same_file(built, tmp_path, chunks):
    """DEFUSE_CMP_CHUNKS: the bin pairs go through build -> GPU -> write in chunks whose stages overlap; cluster ids are a
    running count over the chunks, so any number of chunks (more than bin pairs included) gives the oracle's file."""
    from defuse_amd import build
    from oracle import clustermatepairs_oracle as o
    build.build_tools()
    lines = cmp_cases.many_loci(10)
    r, txt = run_tool(lines, tmp_path, env={"DEFUSE_CMP_CHUNKS": chunks, "DEFUSE_TIMING": "1"})
    assert r.returncode == 0, r.stderr
    exp, n = o.clustermatepairs(lines, 300, 30, 0.95, 5)
    assert n >= 5 and txt == exp
    assert "chunk(s), stages overlapped" in r.stderr


def test_c_oracle_reaches_the_rare_paths_on_the_adversarial_batch(built):
    """What tests/cmp_cases.adversarial_em_batch is for: components that lose every responsibility, likelihoods that
    underflow, seedings that run out of distinct points.  The stale-state reads the reference's members would allow
    (a component without responsibility in the FIRST M step of a fit, every K failing) never occur: k-means leaves no
    cluster empty, and K = 1 cannot fail."""
    from oracle import mpe_c
    off, x, y, u, to_xo, to_yo = cmp_cases.adversarial_em_batch(1, 2000)
    mp = mpe_c.lib().ora_min_probability(30.0, 0.95)
    ncl, member, status, dg, _ = mpe_c.cluster_batch(300.0, 30.0, mp, 5, off, x, y, u, to_xo, to_yo)
    assert dg.nk_zero > 10 and dg.ll_underflow > 0 and dg.kkz_fail > 1000
    assert dg.nk_zero_first_iter == 0 and dg.all_k_failed == 0 and not status.any()
    assert ncl.sum() > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("wave_min", [None, "1000000000"])
def test_em_rare_paths_wave_lane_and_oracle_agree(built, wave_min):
    """The adversarial batch through mpe_cluster_batch — one wave per bin pair (default) and one lane per fit
    (DEFUSE_MPE_WAVE_MIN) — against the C oracle: every membership bit."""
    from defuse_amd import mpe
    from oracle import mpe_c
    mp = mpe_c.lib().ora_min_probability(30.0, 0.95)
    old = os.environ.get("DEFUSE_MPE_WAVE_MIN")
    try:
        if wave_min:
            os.environ["DEFUSE_MPE_WAVE_MIN"] = wave_min
        for seed in (1, 2):
            args = (300.0, 30.0, mp, 5) + cmp_cases.adversarial_em_batch(seed, 4000)
            o_ncl, o_member, o_status, dg, _ = mpe_c.cluster_batch(*args)
            g_ncl, g_member, g_status, t = mpe.cluster_batch(*args)
            assert dg.nk_zero > 20 and dg.ll_underflow > 0 and dg.kkz_fail > 1000
            assert not g_status.any() and (g_ncl == o_ncl).all()
            assert g_member.tobytes() == o_member.tobytes()
            assert (t.n_wave_problems == 0) == bool(wave_min)
    finally:
        if old is None:
            os.environ.pop("DEFUSE_MPE_WAVE_MIN", None)
        else:
            os.environ["DEFUSE_MPE_WAVE_MIN"] = old


@pytest.mark.gpu
def test_em_breakpoint_search_on_runs_and_ties(built):
    """The M step finds the first breakpoint with a positive derivative by searching (hinted by the EM iteration before, else
    by bisection over the runs of equal coordinates) and walks only the last few; with DEFUSE_MPE_NO_JUMP it walks from the
    first one as the reference does.  On problems full of runs and tied prefix sums: both give the C oracle's memberships,
    bit for bit, and the same number of EM iterations as each other (any other breakpoint changes a likelihood somewhere)."""
    from defuse_amd import mpe
    from oracle import mpe_c
    mp = mpe_c.lib().ora_min_probability(30.0, 0.95)
    old = os.environ.get("DEFUSE_MPE_NO_JUMP")
    try:
        for seed in (7, 8):
            args = (300.0, 30.0, mp, 5) + cmp_cases.tie_heavy_em_batch(seed, 300)
            o_ncl, o_member, o_status, dg, _ = mpe_c.cluster_batch(*args)
            assert not o_status.any() and o_ncl.sum() > 1000
            runs = {}
            for no_jump in (False, True):
                if no_jump:
                    os.environ["DEFUSE_MPE_NO_JUMP"] = "1"
                else:
                    os.environ.pop("DEFUSE_MPE_NO_JUMP", None)
                g_ncl, g_member, g_status, t = mpe.cluster_batch(*args)
                assert not g_status.any() and (g_ncl == o_ncl).all()
                assert g_member.tobytes() == o_member.tobytes(), (seed, no_jump)
                runs[no_jump] = t.em_iterations
            assert runs[False] == runs[True] > 10000
    finally:
        if old is None:
            os.environ.pop("DEFUSE_MPE_NO_JUMP", None)
        else:
            os.environ["DEFUSE_MPE_NO_JUMP"] = old


@pytest.mark.gpu
def test_em_shares_on_streams_give_the_same_memberships(built):
    """mpe_cluster_batch runs the sorted bin pairs in shares on streams of their own (k-means start-ups, then EM, each; the small
    problems' EM under the tail of the large problems' k-means).  Whatever the cut points (DEFUSE_MPE_SHARES; 0 = one share):
    the C oracle's memberships bit for bit, and the same EM iteration count as every other cut."""
    from defuse_amd import mpe
    from oracle import mpe_c
    mp = mpe_c.lib().ora_min_probability(30.0, 0.95)
    old = os.environ.get("DEFUSE_MPE_SHARES")
    try:
        for make, seed, n in ((cmp_cases.adversarial_em_batch, 3, 4000), (cmp_cases.tie_heavy_em_batch, 9, 1100)):
            args = (300.0, 30.0, mp, 5) + make(seed, n)
            o_ncl, o_member, o_status, dg, _ = mpe_c.cluster_batch(*args)
            iters = set()
            for shares in ("0", "0.3", "0.5", "0.05,0.2,0.6", "0.9"):
                os.environ["DEFUSE_MPE_SHARES"] = shares
                g_ncl, g_member, g_status, t = mpe.cluster_batch(*args)
                assert (g_status != 0).tolist() == (o_status != 0).tolist() and (g_ncl == o_ncl).all(), shares
                assert g_member.tobytes() == o_member.tobytes(), (make.__name__, shares)
                iters.add(int(t.em_iterations))
            assert len(iters) == 1
    finally:
        if old is None:
            os.environ.pop("DEFUSE_MPE_SHARES", None)
        else:
            os.environ["DEFUSE_MPE_SHARES"] = old
