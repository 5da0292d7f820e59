"""bench.py's N > 1 path as far as one card allows: two ranks (gloo instead of RCCL, both on GPU 0) run the strong-scaling
job of BASELINE configs[3] at a small size — contiguous fusion ranges, no data-path collective, one final gather of the
records on rank 0.  The 8-GPU run over RCCL is the driver's; this covers main()'s multi-rank code on real kernels: the gather
is verified, the gathered records equal the oracle on a sample of every rank's share, and a rank that dies makes the job
exit non-zero instead of hanging."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUSIONS, READS, LQ, LR = 3000, 200, 100, 390           # bench.py's config4 shape, 600 k aligns in all


def run_bench(tmp_path, extra_env=None, timeout=900):
    env = dict(os.environ, DEFUSE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    dump = str(tmp_path / "records.npy")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fusions", str(FUSIONS), "--steps", "3",
                           "--warmup", "1", "--dump-records", dump], capture_output=True, text=True, env=env, timeout=timeout), dump


def test_two_ranks_share_one_card_and_gather(built, tmp_path):
    r, dump = run_bench(tmp_path)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["aligns_per_step_this_gpu"] == FUSIONS // 2 * READS
    g = line["gather"]
    assert g["verified"] is True and g["backend"] == "gloo" and g["records"] == line["config"]["job_records"] > FUSIONS * READS
    assert line["steps"] >= 3 and line["ms_per_step"] * line["steps"] >= 900          # the timed region is at least a second
    # the gathered records against the oracle, on the first fusions of either rank's share (a process of its own: torch's HIP
    # runtime has to come up before the library's, and this pytest process already holds the library)
    code = '''
import sys
sys.path.insert(0, %r)
import numpy as np
import torch
from defuse_amd import dsa, synth
from oracle import dosplitalign_oracle as ora
recs = np.load(%r).view(dsa.RECORD_DTYPE).reshape(-1)
assert np.all(np.diff(recs["pair_idx"].astype(np.int64)) >= 0)          # rank order = pair order of the job
F, P, LQ, LR, SAMPLE = %d, %d, %d, %d, 40
for rank in (0, 1):
    lo, hi = F * rank // 2, F * (rank + 1) // 2
    ref, fus, reads, pairs = synth.make_batch_device(hi - lo, P, LQ, LR, 1000 + lo, "cuda:0", fusion_id_base=lo)
    exp = ora.align_batch(ref, fus, reads, pairs[:SAMPLE * P])
    exp["pair_idx"] += lo * P
    got = recs[(recs["pair_idx"] >= lo * P) & (recs["pair_idx"] < (lo + SAMPLE) * P)]
    assert len(exp) > SAMPLE * P and got.tobytes() == exp.tobytes(), rank
print("ok")
''' % (ROOT, dump, FUSIONS, READS, LQ, LR)
    v = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert v.returncode == 0 and v.stdout.strip().endswith("ok"), (v.stdout, v.stderr[-3000:])


def test_a_lost_rank_fails_the_job(built, tmp_path):
    r, _ = run_bench(tmp_path, {"DEFUSE_BENCH_TEST_EXIT_RANK": "1"}, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]      # no result line from a job that lost a rank
