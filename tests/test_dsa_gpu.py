"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMOKE = os.path.join(ROOT, "tests", "golden", "smoke")


@pytest.fixture(scope="module")
def ora(built):
    from oracle import dosplitalign_oracle as o
    return o


def as_tuples(recs):
    return [tuple(int(x) for x in r) for r in recs]


def check_batch(gpu_ctx, ora, batch):
    got = gpu_ctx.align_batch(*batch)
    exp = ora.align_batch(*batch)
    assert len(got) == len(exp), (len(got), len(exp))
    assert got.tobytes() == exp.tobytes()
    return got


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_parity_mixed(gpu_ctx, ora, seed):
    got = check_batch(gpu_ctx, ora, cases.mixed_batch(seed))
    assert len(got) > 50


def test_parity_long_reads_many_tiles(gpu_ctx, ora):
    # 2x150-like geometry: Lq=150, Lr ~ 585 -> 10 tiles; also reads longer than one tile
    got = check_batch(gpu_ctx, ora, cases.mixed_batch(11, n_fusions=4, reads_per_fusion=30, lq=150, lr=(500, 600)))
    assert len(got) > 20


def test_parity_many_tiles_paths(gpu_ctx, ora):
    """References of 17-20 tiles (beyond the 16-tile winning-tile masks: the combine step falls back to
    the per-tile maxima) and of more than 64 tiles (every tile of a kept pair is replayed)."""
    got = check_batch(gpu_ctx, ora, cases.mixed_batch(31, n_fusions=3, reads_per_fusion=24, lq=60, lr=(1050, 1250)))
    assert len(got) > 10
    got = check_batch(gpu_ctx, ora, cases.mixed_batch(32, n_fusions=2, reads_per_fusion=10, lq=40, lr=(4150, 4300)))
    assert len(got) > 3


def test_parity_single_fusion_fast_path(gpu_ctx, ora):
    """One fusion, many reads over {A,C,G,T,N}: every workgroup takes the LDS-table kernels."""
    import numpy as np
    rng = np.random.default_rng(77)
    bb = cases.BatchBuilder()
    ref0, ref1 = cases.rnd(rng, 389), cases.rnd(rng, 389, b"ACGTN")
    f = bb.add_fusion(ref0, ref1)
    for r in range(700):
        read = cases.mutate(rng, cases.split_read(rng, ref0, ref1, 76), 0.01)
        if r % 50 == 0:
            b = bytearray(read)
            b[int(rng.integers(0, len(b)))] = ord("N")
            read = bytes(b)
        bb.add_read(f, read)
    got = check_batch(gpu_ctx, ora, bb.arrays())
    assert len(got) > 300


@pytest.mark.parametrize("lq,lr,n_reads", [(7600, 9000, 1), (300, 16320, 2), (1000, 3000, 4), (5, 40, 8), (8, 64, 8), (9, 65, 8)])
def test_parity_limits(gpu_ctx, ora, lq, lr, n_reads):
    """The documented limits (reads of 7600 bases, windows of 255 tiles) and the smallest inputs that can still
    produce an alignment (the anchor is 8: a read of 9 can split 8 + 1... none of these may differ from the oracle)."""
    import numpy as np
    rng = np.random.default_rng(lq + lr)
    bb = cases.BatchBuilder()
    ref0, ref1 = cases.rnd(rng, lr), cases.rnd(rng, lr)
    f = bb.add_fusion(ref0, ref1)
    for _ in range(n_reads):
        bb.add_read(f, cases.mutate(rng, cases.split_read(rng, ref0, ref1, lq), 0.01))
    check_batch(gpu_ctx, ora, bb.arrays())


def test_parity_beyond_the_16_bit_kernels(built, ora):
    """Reads longer than 7600 bases and windows longer than 16320 (the reference's matrix grows on demand,
    tools/Matrix.h:98-107): such pairs are swept in 32 bits by kernels of their own, in the same batch as ordinary pairs, and
    their records come out in the caller's pair order like everyone else's - also across slices, also when every pair of the
    batch is a long one, also with tied columns (a window that holds the junction twice)."""
    import numpy as np
    from defuse_amd import dsa
    rng = np.random.default_rng(2024)
    bb = cases.BatchBuilder()
    ref0, ref1 = cases.rnd(rng, 12000), cases.rnd(rng, 11000)
    f_long = bb.add_fusion(ref0, ref1)                              # windows fit, the read does not
    s0, s1 = cases.rnd(rng, 330), cases.rnd(rng, 350)
    f_short = bb.add_fusion(s0, s1)
    w0 = cases.rnd(rng, 9000)
    seg = w0[4000:4060]
    w0 = w0[:7000] + seg + w0[7060:]                                # the same 60 bases twice in a window of 20 000: tied columns
    wide0, wide1 = w0 + cases.rnd(rng, 11000), cases.rnd(rng, 17000)
    f_wide = bb.add_fusion(wide0, wide1)                            # short reads, over-long windows
    for k in range(30):
        bb.add_read(f_short, cases.mutate(rng, cases.split_read(rng, s0, s1, 76), 0.01))
    bb.add_read(f_long, cases.mutate(rng, cases.split_read(rng, ref0, ref1, 10000), 0.01))
    for k in range(30):
        bb.add_read(f_short, cases.mutate(rng, cases.split_read(rng, s0, s1, 76), 0.01))
    bb.add_read(f_wide, wide0[4030:4060] + wide1[100:150])          # junction after the repeated segment
    bb.add_read(f_wide, cases.mutate(rng, cases.split_read(rng, wide0, wide1, 120), 0.02))
    bb.add_read(f_long, cases.mutate(rng, cases.split_read(rng, ref0, ref1, 7601), 0.005))
    for k in range(300):
        bb.add_read(f_short, cases.mutate(rng, cases.split_read(rng, s0, s1, 76), 0.01))
    batch = bb.arrays()
    exp = ora.align_batch(*batch)
    ctx = dsa.Context(0)
    lim = ctx.limits()
    assert lim.max_read_len >= 1 << 20 and lim.max_ref_len >= 1 << 24
    got = ctx.align_batch(*batch)
    assert got.tobytes() == exp.tobytes()
    long_pairs = [30, 61, 62, 63]
    assert all(np.any(exp["pair_idx"] == p) for p in long_pairs)          # every long pair really aligns
    assert np.sum(exp["pair_idx"] == 61) >= 2                             # the repeated segment: tied columns
    assert ctx.timing().cells > 2 * 10001 * 12001
    ctx.close()
    os.environ["DEFUSE_DSA_SLICE_PAIRS"] = "256"                          # two slices, long pairs in the first
    try:
        ctx = dsa.Context(0)
        assert ctx.align_batch(*batch).tobytes() == exp.tobytes()
        assert ctx.timing().fill_launches == 2
        # a batch of long pairs only
        only = cases.BatchBuilder()
        f = only.add_fusion(wide0, wide1)
        only.add_read(f, wide0[4030:4060] + wide1[100:150])
        only.add_read(f, cases.mutate(rng, cases.split_read(rng, wide0, wide1, 90), 0.02))
        b2 = only.arrays()
        assert ctx.align_batch(*b2).tobytes() == ora.align_batch(*b2).tobytes()
        ctx.close()
    finally:
        del os.environ["DEFUSE_DSA_SLICE_PAIRS"]


def test_parity_pruning_diagonal_entry(gpu_ctx, ora):
    """A borderline alignment (one mismatch + one inserted read base: exactly minScore) that leaves tile 0 through
    its last column at row 27 and enters tile 1 diagonally at row 28, alone in its wave: everything of tile 1 up
    to row 27 is dead, so the pruning may only stop after the row that the live boundary value still feeds."""
    import numpy as np
    hits = 0
    for seed in range(6):
        rng = np.random.default_rng(500 + seed)
        ref0, ref1 = bytearray(cases.rnd(rng, 130)), cases.rnd(rng, 90)
        seg = bytearray(ref0[38:66])                     # columns 39..66: rows 27 / 28 sit on columns 64 / 65
        seg[9] = ord("A") if seg[9] != ord("A") else ord("C")          # one mismatch
        other = ord("G") if seg[5] != ord("G") else ord("T")
        prefix = bytes(seg[:5]) + bytes([other]) + bytes(seg[5:])      # one inserted read base: 29 rows, 28 columns
        read = prefix + ref1[:4]                                        # Lq = 33, minScore 59 = 51 + 8
        bb = cases.BatchBuilder()
        f = bb.add_fusion(bytes(ref0), ref1)
        bb.add_read(f, read)
        batch = bb.arrays()
        exp = ora.align_batch(*batch)
        hits += int(any(int(r["read_first"]) == 29 for r in exp))
        check_batch(gpu_ctx, ora, batch)
    assert hits >= 3          # the constructed split is really what wins in most draws


def test_parity_sweep_plans(gpu_ctx, ora, monkeypatch):
    """The sweep plan is a speed heuristic: the same records come out with the device-side plan (pairs grouped by
    fusion), with the caller's order kept (DEFUSE_DSA_NO_REORDER) and for pairs that are not grouped by fusion at
    all (no plan possible), always in the caller's pair order."""
    import numpy as np
    ref, fus, reads, pairs = cases.mixed_batch(41, n_fusions=30, reads_per_fusion=35, lq=76, lr=(300, 420))
    base = check_batch(gpu_ctx, ora, (ref, fus, reads, pairs))
    monkeypatch.setenv("DEFUSE_DSA_NO_REORDER", "1")
    same = check_batch(gpu_ctx, ora, (ref, fus, reads, pairs))
    assert same.tobytes() == base.tobytes()
    monkeypatch.delenv("DEFUSE_DSA_NO_REORDER")
    rng = np.random.default_rng(3)
    shuffled = pairs[rng.permutation(len(pairs))].copy()              # fusions interleaved: several runs per fusion
    check_batch(gpu_ctx, ora, (ref, fus, reads, shuffled))
    assert len(base) > 500


def test_parity_split_table_tier(gpu_ctx, ora):
    """Workgroups of five to ten fusions (30 reads each, reads over {A,C,G,T,N}) take the fill kernel with two
    small LDS tables per fusion; twelve reads per fusion push workgroups past ten fusions onto the generic one."""
    import numpy as np
    rng = np.random.default_rng(79)
    for reads_per_fusion, n_fusions in ((30, 40), (12, 60)):
        bb = cases.BatchBuilder()
        for k in range(n_fusions):
            ref0, ref1 = cases.rnd(rng, int(rng.integers(200, 420))), cases.rnd(rng, int(rng.integers(200, 420)), b"ACGTN")
            f = bb.add_fusion(ref0, ref1)
            for r in range(reads_per_fusion):
                read = cases.mutate(rng, cases.split_read(rng, ref0, ref1, int(rng.integers(40, 77))), 0.02)
                if r % 9 == 0:
                    b = bytearray(read)
                    b[int(rng.integers(0, len(b)))] = ord("N")
                    read = bytes(b)
                bb.add_read(f, read)
        got = check_batch(gpu_ctx, ora, bb.arrays())
        assert len(got) > 200


def test_parity_exotic_reads_hand_workgroups_over(gpu_ctx, ora):
    """Several workgroups of two fusions; a lowercase read every 300 pairs makes the fast kernel hand just
    those workgroups to the generic one (decided on the device while packing the rows)."""
    import numpy as np
    rng = np.random.default_rng(78)
    bb = cases.BatchBuilder()
    for _ in range(2):
        ref0, ref1 = cases.rnd(rng, 389), cases.rnd(rng, 300)
        f = bb.add_fusion(ref0, ref1)
        for r in range(900):
            read = cases.mutate(rng, cases.split_read(rng, ref0, ref1, 76), 0.01)
            if r % 300 == 150:
                read = read.lower()
            bb.add_read(f, read)
    got = check_batch(gpu_ctx, ora, bb.arrays())
    assert len(got) > 800


def test_parity_ties(gpu_ctx, ora):
    got = check_batch(gpu_ctx, ora, cases.tie_batch(4))
    # the tie cases must really produce multi-record pairs
    frags, counts = np.unique(got["frag"], return_counts=True)
    assert counts.max() > 10


def test_parity_repeat_rich_full_workgroups(gpu_ctx, ora):
    """Many kept splits in several tiles, in runs of a hundred pairs: the emit paths beyond registers (LDS staging, a whole
    wave per pair, the walk through global memory), long generic replay tasks, and finish buffers that overflow."""
    batch = cases.repeat_batch(11)
    got = check_batch(gpu_ctx, ora, batch)
    per_pair = np.bincount(got["pair_idx"], minlength=len(batch[3]))
    assert per_pair.max() > 40                       # really heavy pairs
    assert (per_pair > 8).sum() > 200                # and runs of them
    t = gpu_ctx.timing()
    assert t.n_generic_tasks > 500


def test_parity_edges(gpu_ctx, ora):
    check_batch(gpu_ctx, ora, cases.edge_batch())


def test_empty_batch(gpu_ctx):
    from defuse_amd import dsa
    out = gpu_ctx.align_batch(np.zeros(0, np.uint8), np.zeros(0, dsa.FUSION_DTYPE), np.zeros(0, np.uint8),
                              np.zeros(0, dsa.PAIR_DTYPE))
    assert len(out) == 0


def test_bad_arguments(gpu_ctx):
    from defuse_amd import dsa
    ref, fus, reads, pairs = cases.edge_batch()
    bad = pairs.copy()
    bad["fusion_idx"][0] = len(fus)
    with pytest.raises(dsa.DsaError) as e:
        gpu_ctx.align_batch(ref, fus, reads, bad)
    assert e.value.code == -3
    bad = pairs.copy()
    bad["read_off"][3] = reads.size
    with pytest.raises(dsa.DsaError):
        gpu_ctx.align_batch(ref, fus, reads, bad)


def test_smoke_vector_through_gpu(gpu_ctx, ora):
    """The reference's known-answer vector, candidates enumerated by the host-logic restatement,
    DP on the GPU."""
    from defuse_amd import dsa
    d = SMOKE + "/"
    tasks = ora.create_tasks(d + "ref.fa", d + "exons.txt", 300, 30, 50, 50, ora.read_align_region_pairs(d + "regions.txt"))
    reads = {}
    ora.read_fastq(d + "reads.1.fastq", reads)
    ora.read_fastq(d + "reads.2.fastq", reads)
    bb = cases.BatchBuilder()
    fidx = {}
    for t, frag, rend, revcomp, seq in ora.enumerate_candidates(tasks, reads, d + "improper.sam"):
        if t.fusion_id not in fidx:
            fidx[t.fusion_id] = bb.add_fusion(t.seq[0], t.seq[1], fusion_id=t.fusion_id)
        bb.add_read(fidx[t.fusion_id], seq, frag=frag, read_end=rend, revcomp=revcomp)
    got = gpu_ctx.align_batch(*bb.arrays())
    exp = [tuple(int(x) for x in l.split()) for l in open(d + "expected.split.align.txt")]
    assert [t[:9] for t in as_tuples(got)] == exp


def test_slicing_gives_same_records(built, ora):
    """A tiny scratch budget forces several slices; the concatenation must not change anything."""
    from defuse_amd import dsa
    batch = cases.mixed_batch(21, n_fusions=10, reads_per_fusion=150, lq=40, lr=(80, 200))
    os.environ["DEFUSE_DSA_SCRATCH_MB"] = "1"
    try:
        ctx = dsa.Context(0)
    finally:
        del os.environ["DEFUSE_DSA_SCRATCH_MB"]
    got = ctx.align_batch(*batch)
    assert ctx.timing().fill_launches > 1
    ctx.close()
    exp = ora.align_batch(*batch)
    assert got.tobytes() == exp.tobytes()


def test_slices_of_heavy_pairs_run_twice(built, ora):
    """Several slices of the repeat-rich batch (finish buffers overflow in every slice, only the first one writes its records
    before the host has the total), and the batch is run a second time in the same context (descriptors resident, buffers
    grown, the listed kernels' grids taken from the first run)."""
    from defuse_amd import dsa
    batch = cases.repeat_batch(12, n_fusions=15)
    exp = ora.align_batch(*batch)
    os.environ["DEFUSE_DSA_SCRATCH_MB"] = "3"
    try:
        ctx = dsa.Context(0)
    finally:
        del os.environ["DEFUSE_DSA_SCRATCH_MB"]
    ctx.upload(*batch)
    for _ in range(2):
        ctx.run()
        got = ctx.download()
        assert ctx.timing().fill_launches > 1
        assert got.tobytes() == exp.tobytes()
    ctx.close()


def test_plan_again_and_kernels_a_lane_did_not_expect(built, ora):
    """dsa_plan on a resident upload (what bench.py times per step) changes nothing, and a context whose lanes last ran a
    batch of large fusions only (so the next slice launches k_fill_fast<0> alone) runs a batch that needs the split-table
    tiers and the generic kernel: the slice reports what it needed and is run again with every kernel."""
    from defuse_amd import dsa
    rng = np.random.default_rng(91)
    big = cases.mixed_batch(51, n_fusions=6, reads_per_fusion=130, lq=60, lr=(200, 330))
    bb = cases.BatchBuilder()
    for k in range(70):                       # 20 / 9 / 3 reads per fusion: both split-table tiers and the generic kernel
        ref0, ref1 = cases.rnd(rng, int(rng.integers(150, 400))), cases.rnd(rng, int(rng.integers(150, 400)))
        f = bb.add_fusion(ref0, ref1)
        for r in range(20 if k < 25 else 9 if k < 45 else 3):
            read = cases.mutate(rng, cases.split_read(rng, ref0, ref1, int(rng.integers(40, 77))), 0.02)
            bb.add_read(f, read.lower() if (k == 10 and r == 3) else read)
    small = bb.arrays()
    ctx = dsa.Context(0)
    exp_big, exp_small = ora.align_batch(*big), ora.align_batch(*small)
    ctx.upload(*big)
    ctx.run()
    assert ctx.download().tobytes() == exp_big.tobytes()
    ctx.plan()
    ctx.plan()
    ctx.run()
    assert ctx.download().tobytes() == exp_big.tobytes()
    assert ctx.timing().plan_ms > 0
    ctx.upload(*small)
    ctx.run()
    assert ctx.download().tobytes() == exp_small.tobytes()
    ctx.plan()
    ctx.run()
    assert ctx.download().tobytes() == exp_small.tobytes()
    ctx.upload(*big)
    ctx.run()
    assert ctx.download().tobytes() == exp_big.tobytes()
    ctx.close()


def test_run_after_a_failed_run_on_shared_lanes(built, ora):
    """A run that stops half way leaves a slice in flight on its pipeline lane.  The lanes may be shared with other contexts
    (dsa_share_scratch): the next run, here of a context with FEWER slices than the stale slice index, starts from clean
    lanes and gives the right records, and so does a retry of the run that failed.  (Runs in a child process: the failure
    is injected by an environment switch that counts the runs of the process.)"""
    import subprocess
    import sys
    code = '''
import os, sys
sys.path.insert(0, %r)
from defuse_amd import dsa
from oracle import dosplitalign_oracle as ora
from tests import cases
os.environ["DEFUSE_DSA_SCRATCH_MB"] = "1"
big = cases.mixed_batch(21, n_fusions=10, reads_per_fusion=150, lq=40, lr=(80, 200))      # several slices at 1 MB
small = cases.mixed_batch(22, n_fusions=2, reads_per_fusion=20, lq=40, lr=(80, 120))      # one slice
a, b = dsa.Context(0), dsa.Context(0)
b.share_scratch(a)
a.upload(*big)
b.upload(*small)
a.run()
assert a.timing().fill_launches > 2
try:
    a.run()                      # the second run of the process: injected failure, last slice left in flight
    raise SystemExit("the injected failure did not happen")
except dsa.DsaError as e:
    assert "injected" in str(e), e
b.run()
assert b.download().tobytes() == ora.align_batch(*small).tobytes()
a.run()
assert a.download().tobytes() == ora.align_batch(*big).tobytes()
print("ok")
''' % ROOT
    env = dict(os.environ, DEFUSE_DSA_TEST_FAIL_RUN="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout, r.stderr)


@pytest.mark.parametrize("mix", [dict(decoy_frac=0.5), dict(inside_frac=0.5), dict(inside_frac=1.0), dict(inside_frac=0.4, decoy_frac=0.2, lq=150, lr=590)],
                         ids=["decoys-50", "inside-50", "inside-100", "2x150-mixed"])
def test_workload_mixes_equal_the_oracle(built, mix):
    """The mixes bench.py reports beside the headline (what DoAlignment really enumerates, tools/SplitAlignment.cpp:266-303:
    mates that do not cross the junction, mates that do not align at all), 100 000 pairs each, every record against the CPU
    oracle — with the planner's bounds (whole-read-on-one-side bound included) and without them."""
    import bench
    from defuse_amd import dsa, synth
    kw = dict(mix)
    lq, lr = kw.pop("lq", 76), kw.pop("lr", 389)
    batch = synth.make_batch(1000, 100, lq=lq, lr=lr, seed=11, **kw)
    exp = bench.oracle_records(batch, len(batch[3]))
    ctx = dsa.Context(0)
    for flags in (0, dsa.PLAN_NO_TIGHTEN):
        ctx.set_plan_options(flags)
        got = ctx.align_batch(*batch)
        assert len(got) == len(exp) and got.tobytes() == exp.tobytes(), (mix, flags)
    ctx.close()


@pytest.mark.parametrize("shape", [(33334, 3, 76, 389), (20000, 5, 150, 590), (100000, 1, 76, 389)], ids=["3-per-fusion", "5-per-fusion-2x150", "1-per-fusion"])
def test_many_fusions_with_few_reads_equal_the_oracle(built, shape):
    """The shape of a pipeline chunk (a million reads against every fusion: a handful of candidates per fusion and batch,
    profiles/r04/reads_per_fusion.txt): workgroups hold more fusions than the table kernels take and run in k_fill_generic, the
    planner works on fusions of one to five reads.  100 000 pairs, every record against the CPU oracle."""
    import bench
    from defuse_amd import dsa, synth
    fusions, reads, lq, lr = shape
    batch = synth.make_batch(fusions, reads, lq=lq, lr=lr, seed=17)
    exp = bench.oracle_records(batch, len(batch[3]))
    ctx = dsa.Context(0)
    got = ctx.align_batch(*batch)
    assert len(got) == len(exp) and got.tobytes() == exp.tobytes(), shape
    assert ctx.timing().n_generic_tasks > 0
    ctx.close()


def test_timing_cells_is_exact_on_every_planning_path(built):
    """dsa_timing.cells = sum over pairs of (Lref0 + 1 + Lref1 + 1) * (Lread + 1), whatever the planning did: the planned
    sweep, the caller's order (PLAN_NO_REORDER), a fusion whose pairs are not contiguous (the identity path of the slice),
    reads beyond the 16-bit kernels (their blanked copies must not count twice), several slices."""
    from defuse_amd import dsa

    def expected(b):
        ref, fus, reads, pairs = b
        f = fus[pairs["fusion_idx"]]
        return int(((f["ref0_len"].astype(np.int64) + 1 + f["ref1_len"] + 1) * (pairs["read_len"].astype(np.int64) + 1)).sum())
    ctx = dsa.Context(0)
    b = cases.mixed_batch(31, n_fusions=9, reads_per_fusion=70, lq=60, lr=(150, 400))
    for flags in (0, dsa.PLAN_NO_REORDER, dsa.PLAN_NO_TIGHTEN | dsa.PLAN_NO_RANK):
        ctx.set_plan_options(flags)
        ctx.align_batch(*b)
        assert ctx.timing().cells == expected(b), flags
    ctx.set_plan_options(0)
    ref, fus, reads, pairs = b
    shuffled = pairs[np.random.default_rng(3).permutation(len(pairs))]           # fusions in several runs
    ctx.align_batch(ref, fus, reads, shuffled)
    assert ctx.timing().cells == expected((ref, fus, reads, shuffled))
    ctx.set_scratch_budget(1 << 20)                                              # several slices
    ctx.align_batch(*b)
    assert ctx.timing().fill_launches > 1 and ctx.timing().cells == expected(b)
    ctx.close()
    ctx = dsa.Context(0)
    long_b = cases.mixed_batch(32, n_fusions=2, reads_per_fusion=6, lq=60, lr=(150, 300))
    ref, fus, reads, pairs = long_b
    big = np.frombuffer(cases.rnd(np.random.default_rng(5), 8000), dtype=np.uint8)      # one read of 8000 bases: the 32-bit path
    reads2 = np.concatenate([reads, big])
    pairs2 = np.concatenate([pairs, pairs[:1]])
    pairs2[-1]["read_off"], pairs2[-1]["read_len"] = len(reads), len(big)
    ctx.align_batch(ref, fus, reads2, pairs2)
    assert ctx.timing().cells == expected((ref, fus, reads2, pairs2))
    ctx.close()


def test_stream_carries_on_after_a_failed_batch(built, ora):
    """A batch of a stream that ends with a device / run error is consumed on the C side; the Python wrapper drops its entry
    in step, so a caller that catches the DsaError gets the FOLLOWING batches' own records (round-3 advice: it used to
    return views of the previous batch's array)."""
    import subprocess
    import sys
    code = '''
import sys
sys.path.insert(0, %r)
import numpy as np
from defuse_amd import dsa
from oracle import dosplitalign_oracle as ora
from tests import cases
batches = [cases.mixed_batch(70 + k, n_fusions=3 + k, reads_per_fusion=30 + 10 * k, lq=40 + 5 * k, lr=(90, 200)) for k in range(4)]
st = dsa.Stream(0, depth=2)
outs = [np.zeros(4 * len(b[3]) + 64, dtype=dsa.RECORD_DTYPE) for b in batches]
got, k_sub = [], 0
for k in range(4):
    while k_sub < 4 and k_sub - k < 2:
        st.submit(*dsa._check_arrays(*batches[k_sub]), outs[k_sub])
        k_sub += 1
    try:
        got.append(st.collect().copy())
    except dsa.DsaError as e:
        assert "injected" in str(e), e
        got.append(None)
assert got[1] is None and got[0] is not None, [g is None for g in got]     # the second run of the process fails
for k in (0, 2, 3):
    assert got[k].tobytes() == ora.align_batch(*batches[k]).tobytes(), k
assert not st._inflight
st.close()
print("ok")
''' % ROOT
    env = dict(os.environ, DEFUSE_DSA_TEST_FAIL_RUN="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout, r.stderr)


def test_stream_of_batches_equals_the_oracle(built, ora):
    """The streaming entry points (what a tool that cuts its candidates into batches calls): seven different batches through
    a stream of depth 3 - pinned and ordinary host buffers, one batch whose records do not fit its buffer (collected
    again into a larger one), a submit beyond the depth (refused), an empty batch - every batch equals the oracle."""
    from defuse_amd import dsa
    batches = [cases.mixed_batch(60 + k, n_fusions=4 + 3 * k, reads_per_fusion=20 + 25 * (k % 3), lq=40 + 9 * k, lr=(90 + 30 * k, 160 + 40 * k))
               for k in range(6)]
    batches.insert(3, cases.repeat_batch(13, n_fusions=3))            # many records per pair
    exp = [ora.align_batch(*b) for b in batches]
    st = dsa.Stream(0, depth=3)
    keep, outs = [], []
    for k, b in enumerate(batches):
        if k % 2 == 0:                                                # pinned inputs and output
            pins = [dsa.pinned_copy(a) for a in b]
            keep.append(pins)
            arrs = [p.array for p in pins]
        else:
            arrs = list(dsa._check_arrays(*b))
        cap = 16 if k == 3 else len(exp[k]) + 5                       # batch 3: far too small
        if k % 2 == 0:
            po = dsa.PinnedArray((cap,), dsa.RECORD_DTYPE)
            keep.append(po)
            out = po.array
        else:
            out = np.zeros(cap, dtype=dsa.RECORD_DTYPE)
        outs.append((arrs, out))
    got = []
    submitted = 0
    for k in range(len(batches)):
        while submitted < len(batches) and submitted - k < 3:
            st.submit(*outs[submitted][0], outs[submitted][1])
            submitted += 1
        if k == 0:
            with pytest.raises(dsa.DsaError) as e:                    # depth batches in flight
                st.submit(*outs[0][0], outs[0][1])
            assert e.value.code == dsa.DSA_E_BUSY
        got.append(st.collect().copy())
    for k in range(len(batches)):
        assert got[k].tobytes() == exp[k].tobytes(), k
    assert len(exp[3]) > 16
    # an empty batch, and the stream is reusable after everything was collected
    empty = [np.zeros(0, np.uint8), np.zeros(0, dsa.FUSION_DTYPE), np.zeros(0, np.uint8), np.zeros(0, dsa.PAIR_DTYPE)]
    st.submit(*empty, np.zeros(4, dsa.RECORD_DTYPE))
    st.submit(*outs[1][0], outs[1][1])
    assert len(st.collect()) == 0
    assert st.collect().tobytes() == exp[1].tobytes()
    with pytest.raises(dsa.DsaError):
        st.collect()                                                  # nothing submitted
    st.close()


def test_full_size_properties(gpu_ctx, ora):
    """BASELINE config 2 at full size (10k fusions x 100 reads, 2x76): size-independent properties
    plus an oracle check on a random sample of fusions."""
    from defuse_amd import synth
    lq, lr, F, P = 76, 389, 10000, 100
    ref, fus, reads, pairs = synth.make_batch(F, P, lq=lq, lr=lr, seed=2)
    gpu_ctx.upload(ref, fus, reads, pairs)
    n = gpu_ctx.run()
    got = gpu_ctx.download()
    assert n == len(got) and n > 0.9 * F * P
    t = gpu_ctx.timing()
    assert t.cells == F * P * synth.cells_per_align(lq, lr)
    # structural invariants of SplitReadAligner::GetAlignments output
    assert (got["read_first"] + got["read_second"] == lq).all()
    assert (got["score"] >= 8).all() and (got["score"] <= 2 * np.maximum(got["read_first"], got["read_second"])).all()
    assert (got["ref_first"] >= 1).all() and (got["ref_first"] <= lr).all()
    assert (got["ref_second"] >= -1).all() and (got["ref_second"] < lr - 1).all()
    assert (np.diff(got["frag"]) >= 0).all()                       # ordered by pair
    key = np.stack([got["frag"], got["ref_first"], got["ref_second"]], axis=1)
    assert len(np.unique(key, axis=0)) == len(key)                 # refSplit de-duplicated per pair
    # idempotence: a second run over the resident batch gives identical bytes
    assert gpu_ctx.run() == n
    assert gpu_ctx.download().tobytes() == got.tobytes()
    # oracle on a sample of whole fusions
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(F, size=12, replace=False))
    sel = np.isin(pairs["fusion_idx"], sample)
    exp = ora.align_batch(ref, fus, reads, pairs[sel])
    sub = got[np.isin(got["fusion_id"], sample)]
    for name in got.dtype.names:
        if name != "pair_idx":                                     # the sample renumbers the pairs
            assert (sub[name] == exp[name]).all(), name
    assert len(sub) == len(exp)
    # order independence: the same fusions presented in reverse order give the same records per pair
    order = np.argsort(-pairs["fusion_idx"][sel], kind="stable")
    rev = gpu_ctx.align_batch(ref, fus, reads, pairs[sel][order])
    assert sorted(t[:9] for t in as_tuples(rev)) == sorted(t[:9] for t in as_tuples(exp))


@pytest.mark.parametrize("F,P,lq,lr", [(10000, 100, 76, 389), (5000, 200, 100, 390)])
def test_full_size_every_record_against_oracle(gpu_ctx, ora, F, P, lq, lr):
    """BASELINE config 2 at full size (and one GPU's worth of config 4's shape: 200 reads per fusion, 2x100, Lref 390),
    every one of its ~1.7 M records against the CPU oracle (oracle/dsa_oracle.c runs
    without the GIL: one thread per host core, each on a contiguous share of the pairs, about ten seconds on the box's
    16 cores).  Set DEFUSE_TEST_FULL_ORACLE=0 to skip it on a machine with few cores."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from defuse_amd import synth
    if os.environ.get("DEFUSE_TEST_FULL_ORACLE", "1") == "0":
        pytest.skip("DEFUSE_TEST_FULL_ORACLE=0")
    ref, fus, reads, pairs = synth.make_batch(F, P, lq=lq, lr=lr, seed=2)
    got = gpu_ctx.align_batch(ref, fus, reads, pairs)
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    bounds = np.linspace(0, len(pairs), cores + 1).astype(np.int64)
    with ThreadPoolExecutor(max_workers=cores) as ex:
        parts = list(ex.map(lambda k: ora.align_batch(ref, fus, reads, pairs[bounds[k]:bounds[k + 1]]), range(cores)))
    for k, part in enumerate(parts):                               # a share numbers its pairs from zero
        part["pair_idx"] += int(bounds[k])
    exp = np.concatenate(parts)
    assert len(got) == len(exp) and len(exp) > 1_500_000
    assert got.tobytes() == exp.tobytes()


def test_bench_share_runs_several_uploads_like_one(gpu_ctx):
    """bench.py's multi-upload loop (one rank's share of BASELINE configs[3] kept resident as several uploads, one dsa_ctx
    each, run one after the other): every record of every upload against the oracle, pair numbers job-wide."""
    import bench
    from defuse_amd import dsa, synth
    from oracle import dosplitalign_oracle as ora
    workload = dict(fusions=700, reads=40, lq=100, lr=390)
    share = bench.Share(dsa, synth, 0, workload, 100, 400, seed_base=77, on_device=True, log=lambda m: None, upload_fusions=128,
                        keep_batches=True, gen_device="cpu")   # torch's own HIP runtime cannot start after the library's has (one process)
    try:
        assert len(share.ctxs) == 3 and share.total_pairs == 300 * 40
        assert share.pair_base == [100 * 40, 200 * 40, 300 * 40]
        n1, _ = share.run()
        n2, ts = share.run()                                    # a second step over the resident uploads gives the same
        assert n1 == n2 == sum(t.n_records for t in ts)
        for ctx, batch in zip(share.ctxs, share.batches):
            got = ctx.download()
            exp = ora.align_batch(*batch)
            assert len(exp) > 1000 and got.tobytes() == exp.tobytes()
            assert batch[1]["fusion_id"][0] in (100, 200, 300)   # fusion ids are the job's, not the upload's
    finally:
        share.close()
