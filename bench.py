#!/usr/bin/env python3
"""bench.py — split-read DP aligns/s on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path (pack -> DP fill -> combine / replay / emit) over the whole job's synthetic
candidates, which are resident in HBM before the timed region starts.

  N = 1   BASELINE.json configs[1]: 10k synthetic fusions x 100 reads, 2x76 bp (Lref 389), 1 M aligns, one upload.
  N > 1   BASELINE.json configs[3], STRONG scaling: 1 M fusions x 200 reads, 2x100 bp (Lref 390), 200 M aligns in all;
          rank r takes the contiguous fusion range [r F/N, (r+1) F/N) and keeps it resident as several uploads (one
          dsa_ctx each: an upload addresses its read bytes with 32-bit offsets, include/defuse_dsa.h), which a step runs
          one after the other.  Fusions are independent (tools/SplitAlignment.cpp:292), so the timed region holds no
          collective beyond the barrier / max-reduction of the clock.  After it the path's one exchange step — the
          gather of all result records on rank 0 — runs once over RCCL, device to device, and is
          reported by itself ("gather"); if it fails the run fails.  Rank 0 then repeats its own share alone
          ("one_gpu_same_share") so that the line shows what a GPU does on that share with the other ranks idle.
  --workload config2|config4 and --fusions/--reads/--lq/--lr override the defaults (--fusions is the whole job).

    python bench.py                       # N = 1
    python bench.py --gpus 2              # starts its own 2 ranks (children, before this process touches a GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIN_TIMED_S = 3.0          # the timed region is at least this long whatever --steps says ("steps" reports what ran)
PROFILE_DIR = os.path.join(ROOT, "profiles", "r04")
WORKLOADS = {"config2": dict(fusions=10000, reads=100, lq=76, lr=389),
             "config4": dict(fusions=1000000, reads=200, lq=100, lr=390)}
UPLOAD_FUSIONS = 50000     # fusions per upload of a multi-upload share: 10 M aligns, 1 GB of read bytes at config 4
UPLOAD_SCRATCH = 24 << 30  # scratch planes per pipeline lane; the uploads of a share run one after the other and share the lanes


def library_hash():
    from defuse_amd import dsa
    lib = dsa.load_library()
    return lib.dsa_version().decode().split()[-1]


def library_build_flags():
    from defuse_amd import dsa
    return dsa.load_library().dsa_build_flags().decode()


def hip_runtime_mapped():
    """The libamdhip64 this process has mapped.  PyTorch-ROCm brings its own copy under the same SONAME as the system one the
    library links to; whichever is loaded first serves both, and exactly one may be mapped — two copies would mean two HIP
    runtimes with two sets of streams and contexts behind one process's kernels."""
    paths = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                paths.add(line.split()[-1])
    if len(paths) != 1:
        raise SystemExit("bench.py: expected exactly one libamdhip64 in this process, found %s" % (sorted(paths) or "none"))
    return paths.pop()


def profile_block(name, workload, lib_hash, same_size=True):
    """A committed rocprofv3 counter summary (profiles/r04/<name>), only if it was taken on this workload AND on the
    kernels that are running now (same hash of defuse_amd/csrc + flags as the loaded library reports).  same_size=False
    accepts counters of the same shape (reads per fusion, read and window length) taken on another number of fusions: the
    caller scales per-launch figures by the aligns per launch (the kernels' work per align does not depend on the count)."""
    path = os.path.join(PROFILE_DIR, name)
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, "no %s" % os.path.relpath(path, ROOT)
    w = d.get("workload", {})
    keys = ("fusions", "reads", "lq", "lr") if same_size else ("reads", "lq", "lr")
    if workload is not None and tuple(w.get(k) for k in keys) != tuple(workload[k] for k in keys):
        return None, "%s was taken on another workload" % os.path.relpath(path, ROOT)
    if d.get("source_hash") != lib_hash:
        return None, "%s was taken on other kernels (source hash %s, library %s): re-profile" % (
            os.path.relpath(path, ROOT), d.get("source_hash"), lib_hash)
    return d, os.path.relpath(path, ROOT)


def valu_issue(workload, lib_hash, launch_ms, suffix="", aligns_per_launch=None):
    """The unit that binds this integer kernel (not HBM, not MFMA) is VALU issue.  Wave instructions per launch
    by kind come from the SQ counter passes of this build (pmc_sq.json); the duration is this run's HIP-event time; the
    peak prices every kind at its measured issue rate (profiles/microbench/valu_rate*.hip: 2 cycles per wave for
    v_add_u32 / v_sub / v_xor and the other plain VOP2 integer ops, 4 for VOP3P and max3), per SIMD, at the nominal 2.4 GHz."""
    d, src = profile_block("pmc_sq%s.json" % suffix, workload, lib_hash, same_size=not suffix)
    if d is None:
        return None, src
    n = d["launches"]
    scale = 1.0
    if suffix and aligns_per_launch and d.get("aligns_per_launch"):
        scale = aligns_per_launch / d["aligns_per_launch"]              # counters of the same shape at another launch size
    instr = scale * d["pass1"]["SQ_INSTS_VALU"] / n
    busy_quads = scale * d["pass1"].get("SQ_ACTIVE_INST_VALU", 0) / n   # quad-cycles in which a SIMD issued VALU
    simds = 256 * 4
    kernel_cycles = launch_ms * 1e-3 * 2.4e9
    out = {"valu_instructions_per_launch": instr,
           "issue_cycles_flat4": 4.0 * instr / simds, "kernel_cycles_nominal": kernel_cycles,
           "frac_flat4": 4.0 * instr / simds / kernel_cycles,
           "measured_busy_frac": (4.0 * busy_quads / simds / kernel_cycles) if busy_quads else None,
           "unit": "cycles per SIMD at 2.4 GHz nominal", "source": src}
    md, _ = profile_block("fill_mix.json", None, lib_hash)      # static mix of the row sweep (profiles/microbench/fill_mix.py)
    mix = md.get("mix") if md else None
    if mix:
        per = 2.0 * mix["two_cycle"] + 4.0 * mix["four_cycle"]
        out["issue_cycles_priced"] = per * instr / simds
        out["frac_priced"] = per * instr / simds / kernel_cycles
        out["mix"] = mix
    return out, src


def cpu_baseline(ref, fus, reads, pairs, budget_s=12.0):
    """Times the CPU oracle (a literal port of the reference's algorithm, oracle/dsa_oracle.c) on a bounded sample of the
    same workload: one thread per host core this process may use (the way deFuse itself scales: one dosplitalign per read
    chunk), each on its own contiguous share of the sample.  Returns the block and the records of the first share."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import dosplitalign_oracle as ora
    n = 256
    t0 = time.perf_counter()
    ora.align_batch(ref, fus, reads, pairs[:n])
    per = (time.perf_counter() - t0) / n                      # one thread, seconds per align
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # a one-GPU box's CPU share
    share = int(max(n, min(len(pairs) // cores, budget_s / max(per, 1e-9))))
    chunks = [pairs[k * share:(k + 1) * share] for k in range(cores)]
    chunks = [c for c in chunks if len(c)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(chunks)) as ex:      # the C oracle runs without the GIL
        res = list(ex.map(lambda c: ora.align_batch(ref, fus, reads, c), chunks))
    dt = time.perf_counter() - t0
    total = sum(len(c) for c in chunks)
    return {"value": total / dt, "unit": "aligns/s", "cores": len(chunks), "kind": "port",
            "sample": "first %d aligns of the same batch in %d shares, oracle/dsa_oracle.c, one thread per share, %.1f s "
                      "(one thread alone: %.0f aligns/s)" % (total, len(chunks), dt, 1.0 / per)}, res[0], len(chunks[0])


def oracle_records(batch, n_pairs, threads=16):
    """The CPU oracle's records of the first n_pairs pairs of a batch (the checker of the sensitivity legs), on host threads."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import dosplitalign_oracle as ora
    ref, fus, reads, pairs = batch
    per = -(-n_pairs // threads)
    chunks = [(k, pairs[k:min(n_pairs, k + per)]) for k in range(0, n_pairs, per)]
    with ThreadPoolExecutor(max_workers=len(chunks)) as ex:
        res = list(ex.map(lambda c: ora.align_batch(ref, fus, reads, c[1]), chunks))
    for (k, _), r in zip(chunks, res):
        r["pair_idx"] += k
    return np.concatenate(res)


def sensitivity(dsa, synth, main_ctx, workload, main_batch, min_s=0.5, check_pairs=20000):
    """Secondary legs beside the headline, never the value: the headline batch makes EVERY read cross the junction, which is
    the friendliest case for the exact pruning (DoAlignment really enumerates every mate whose partner falls in a mate region,
    tools/SplitAlignment.cpp:266-303).  Each leg: plan + run per step on a resident batch of the same sizes, its records
    compared with the CPU oracle on the first pairs."""
    import numpy as np
    legs = {}
    F, P, lq, lr = workload["fusions"], workload["reads"], workload["lq"], workload["lr"]
    specs = [("decoys_50pct", dict(decoy_frac=0.5), 0, "half of the reads replaced by random sequence (candidates that do not align)"),
             ("inside_one_window_50pct", dict(inside_frac=0.5), 0, "half of the reads lie wholly inside one window (one matrix scores 2 Lq, the other side stays zero)"),
             ("no_per_pair_bound", None, dsa.PLAN_NO_TIGHTEN, "the headline batch without the per-pair score bounds (pruning against minScore only)")]
    for name, kw, flags, what in specs:
        batch = main_batch if kw is None else synth.make_batch(F, P, lq=lq, lr=lr, seed=2, **kw)
        ctx = dsa.Context(main_ctx.device)
        ctx.share_scratch(main_ctx)
        ctx.set_plan_options(flags)
        ctx.upload(*batch)
        n_rec = ctx.run()
        t = ctx.timing()
        got = ctx.download()
        nchk = min(check_pairs, len(batch[3]))
        exp = oracle_records(batch, nchk)
        if got[got["pair_idx"] < nchk].tobytes() != exp.tobytes():
            raise SystemExit("bench.py: sensitivity leg %s: GPU records differ from the oracle on the sample" % name)
        k, dt = 0, 0.0
        t0 = time.perf_counter()
        while dt < min_s:
            ctx.plan()
            ctx.run()
            k += 1
            dt = time.perf_counter() - t0
        t = ctx.timing()
        legs[name] = {"aligns_per_s": len(batch[3]) * k / dt, "ms_per_step": dt / k * 1e3, "steps": k, "fill_ms": t.fill_ms, "plan_ms": t.plan_ms,
                      "finish_ms": t.finish_ms, "records_per_align": round(n_rec / len(batch[3]), 4), "oracle_checked_pairs": nchk, "what": what}
        ctx.close()
    return legs


def one_shot(ctx, batch, reps=5, stream_batches=12, depth=3):
    """The path as a caller without anything resident sees it, PCIe included (reported beside the value, never as it):
    "single" = one dsa_align_batch (upload, validation, planning, run, records out), "streamed" = the same batch submitted
    over and over through a dsa_stream of `depth` batches in flight (copies in, plan / run and copies out overlap), time
    per batch in the steady state.  Host buffers are pinned (dsa_host_alloc)."""
    import numpy as np
    from defuse_amd import dsa
    pins = [dsa.pinned_copy(a) for a in dsa._check_arrays(*batch)]
    arrs = [p.array for p in pins]
    n_pairs = len(arrs[3])
    cap = max(1024, 3 * n_pairs)
    outs = [dsa.PinnedArray((cap,), dsa.RECORD_DTYPE) for _ in range(depth)]
    n = ctx.align_batch_into(*arrs, outs[0].array)          # warm: buffers of the context grow here
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        n = ctx.align_batch_into(*arrs, outs[0].array)
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    single = {"ms_per_batch": best * 1e3, "median_ms": sorted(ts)[len(ts) // 2] * 1e3, "reps": reps}
    st = dsa.Stream(ctx_device(ctx), depth)
    for k in range(depth):                                   # warm: every slot grows its buffers
        st.submit(*arrs, outs[k].array)
    for k in range(depth):
        st.collect()
    t0 = time.perf_counter()
    sub = 0
    n_rec = 0
    for k in range(stream_batches):
        while sub < stream_batches and sub - k < depth:
            st.submit(*arrs, outs[sub % depth].array)
            sub += 1
        n_rec = len(st.collect())
    dt = (time.perf_counter() - t0) / stream_batches
    st.close()
    assert n_rec == n
    for p in pins + outs:
        p.free()
    return {"ms_per_1M_aligns": dt * 1e3 * 1e6 / n_pairs, "ms_per_batch": dt * 1e3, "aligns_per_s": n_pairs / dt, "batches": stream_batches,
            "depth": depth, "records_per_batch": int(n), "single_call": single,
            "note": "dsa_stream_submit / dsa_stream_collect, %d batches of %d aligns through %d slots, pinned host buffers in and out; "
                    "single_call = one dsa_align_batch, nothing overlapped" % (stream_batches, n_pairs, depth)}


def ctx_device(ctx):
    return getattr(ctx, "device", 0)


class Share:
    """One rank's part of the job: its fusion range, resident as one or several uploads."""

    def __init__(self, dsa, synth, device_index, workload, f_lo, f_hi, seed_base, on_device, log, upload_fusions=UPLOAD_FUSIONS, keep_batches=False, gen_device=None):
        import numpy as np
        self.ctxs, self.n_pairs, self.pair_base = [], [], []
        self.first_batch = None
        self.batches = [] if keep_batches else None
        n_up = 1 if not on_device else max(1, -(-(f_hi - f_lo) // upload_fusions))
        per = -(-(f_hi - f_lo) // n_up)
        lo = f_lo
        while lo < f_hi:
            hi = min(f_hi, lo + per)
            t0 = time.perf_counter()
            if on_device:
                b = synth.make_batch_device(hi - lo, workload["reads"], workload["lq"], workload["lr"], seed_base + lo,
                                            gen_device or "cuda:%d" % device_index, fusion_id_base=lo)
            else:
                b = synth.make_batch(hi - lo, workload["reads"], lq=workload["lq"], lr=workload["lr"], seed=seed_base)
            ctx = dsa.Context(device_index)
            if n_up > 1:
                if self.ctxs:
                    ctx.share_scratch(self.ctxs[0])
                else:
                    ctx.set_scratch_budget(UPLOAD_SCRATCH)
            ctx.upload(*b)
            self.ctxs.append(ctx)
            self.n_pairs.append(len(b[3]))
            self.pair_base.append(lo * workload["reads"])
            if self.first_batch is None:
                self.first_batch = b
            if keep_batches:
                self.batches.append(b)
            log("fusions [%d, %d): %d aligns generated and uploaded in %.1f s" % (lo, hi, len(b[3]), time.perf_counter() - t0))
            lo = hi
        self.total_pairs = int(np.sum(self.n_pairs))

    def run(self, plan=True):
        """One step over the share: per upload the sweep planning (dsa_plan: order of the fusions and of the pairs, per-pair
        score bounds) and the run; returns (records, per-upload timings).  plan=False runs an already planned resident
        batch again (reported as "resident_rerun", never as the value)."""
        n, ts = 0, []
        for ctx in self.ctxs:
            if plan:
                ctx.plan()
            n += ctx.run()                     # synchronous: returns after the upload's last kernel finished
            ts.append(ctx.timing())
        return n, ts

    def close(self):
        for ctx in self.ctxs:
            ctx.close()


def gather_job(share, rank, world, backend, barrier, dump=None):
    """The final gather of the job's records on rank 0 (defuse_amd/shard.py:gather_records), once, after the timed steps:
    every upload's records are copied device -> device into one tensor per rank (pair numbers made job-wide), then one
    all_gather of the counts and one group of exact-size isend/irecv.  Any failure propagates (non-zero exit)."""
    import torch
    from defuse_amd import shard
    n_rec = [int(c.timing().n_records) for c in share.ctxs]
    buf = torch.empty((max(sum(n_rec), 1), shard.RECORD_WORDS), dtype=torch.int32, device="cuda")
    at = 0
    for ctx, n, base in zip(share.ctxs, n_rec, share.pair_base):
        got = ctx.records_to_device(buf[at:].data_ptr(), buf.shape[0] - at)
        assert got == n
        buf[at:at + n, shard.RECORD_WORDS - 1] += base
        at += n
    rows = buf[:at] if backend == "nccl" else buf[:at].cpu()      # gloo rehearsal: host tensors
    pair_lo, pair_hi = share.pair_base[0], share.pair_base[-1] + share.n_pairs[-1]
    barrier()
    t0 = time.perf_counter()
    out, counts = shard.gather_records(rows)
    barrier()
    ms = (time.perf_counter() - t0) * 1e3
    bounds = torch.tensor([pair_lo, pair_hi], dtype=torch.int64, device=rows.device)
    allb = torch.zeros(2 * world, dtype=torch.int64, device=rows.device)
    torch.distributed.all_gather_into_tensor(allb, bounds)
    if rank != 0:
        return None
    total = sum(counts)
    if out.shape[0] != total:
        raise RuntimeError("gather: %d records arrived, %d announced" % (out.shape[0], total))
    lo = 0
    for r, c in enumerate(counts):             # every rank's records arrived in its slot with its own pair numbers
        if c:
            col = out[lo:lo + c, shard.RECORD_WORDS - 1]
            if int(col.min()) < int(allb[2 * r]) or int(col.max()) >= int(allb[2 * r + 1]):
                raise RuntimeError("gather: records of rank %d carry pair numbers outside its share" % r)
        lo += c
    nbytes = (total - counts[0]) * shard.RECORD_WORDS * 4
    if dump:
        import numpy as np
        np.save(dump, out.cpu().numpy())
    return {"ms": ms, "records": total, "bytes_moved": nbytes, "GB/s": nbytes / (ms * 1e-3) / 1e9, "verified": True,
            "backend": "rccl" if backend == "nccl" else backend, "included_in_value": False, "times_per_job": 1,
            "pattern": "all_gather of counts + grouped isend/irecv, exact sizes, peers -> rank 0"}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as children (before this process touches a GPU) with
    the environment torch.distributed.run would give them, relay rank 0's line, fail if any rank fails."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a rank that fails takes the job down: the others are given a few seconds to notice (a collective that lost its peer
    # raises) and are then ended, so that a lost rank is a non-zero exit and never a hang
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and rc == 0:
                rc = p.returncode
                deadline = time.monotonic() + 20.0
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    for p in procs:
        rc = rc or p.returncode
    sys.exit(rc if rc else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 100 at config 2, 20 at config 4")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None, help="default: config2 at N = 1, config4 (strong scaling) at N > 1")
    ap.add_argument("--fusions", type=int, default=None, help="fusions of the WHOLE job")
    ap.add_argument("--reads", type=int, default=None)
    ap.add_argument("--lq", type=int, default=None)
    ap.add_argument("--lr", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the secondary workload-mix legs (N = 1)")
    ap.add_argument("--dump-records", default=None, help="N > 1: rank 0 writes the gathered records of the job (numpy .npy) here")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3 passes: exactly --steps timed steps and nothing else (no minimum duration, no resident re-run, "
                         "no one-shot leg, no CPU baseline)")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; start it as\n  python bench.py --gpus %d\nor\n  python -m torch.distributed.run "
                         "--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d\n"
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
        sys.exit(2)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    from defuse_amd import dsa, synth

    name = args.workload or ("config2" if world == 1 else "config4")
    workload = dict(WORKLOADS[name])
    for k in ("fusions", "reads", "lq", "lr"):
        if getattr(args, k) is not None:
            workload[k] = getattr(args, k)
    custom = workload != WORKLOADS[name]
    steps = args.steps if args.steps is not None else (100 if workload["fusions"] * workload["reads"] <= 4_000_000 else 20)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one process per GPU; DEFUSE_BENCH_BACKEND=gloo lets several ranks share one card for a rehearsal
    backend = os.environ.get("DEFUSE_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def log(msg):
        sys.stderr.write("[bench rank %d] %s\n" % (rank, msg))
        sys.stderr.flush()

    # this rank's contiguous fusion range of the job
    F = workload["fusions"]
    f_lo, f_hi = F * rank // world, F * (rank + 1) // world
    multi = world > 1 or (f_hi - f_lo) * workload["reads"] * workload["lq"] >= 2 ** 31 - 2 ** 20 or (f_hi - f_lo) > 2 * UPLOAD_FUSIONS
    share = Share(dsa, synth, local_rank, workload, f_lo, f_hi, seed_base=2 if not multi else 1000, on_device=multi, log=log)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, plan=True):
        """k whole steps between two barriers; the maximum of the ranks' clocks."""
        nonlocal n_rec
        fill_ms, pack_ms, finish_ms, plan_ms, launches = [], [], [], [], []
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            n_rec, ts = share.run(plan)
            fill_ms.append(sum(t.fill_ms for t in ts))
            launches.append(sum(t.fill_launches for t in ts))
            pack_ms.append(sum(t.pack_ms for t in ts))
            finish_ms.append(sum(t.finish_ms for t in ts))
            plan_ms.append(sum(t.plan_ms for t in ts))
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, (fill_ms, pack_ms, finish_ms, plan_ms, launches)

    n_rec = 0
    for _ in range(args.warmup):
        n_rec, _ = share.run()
    hip_path = hip_runtime_mapped()       # after the first kernels of both torch and the library: one runtime serves them
    # The CPU baseline runs BEFORE the timed GPU region (rank 0, N = 1): ten seconds of host threads at the end of the run
    # would leave the GPU idle in every utilisation sample an observer takes there.
    base = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not args.profile_run:
        if not args.warmup:
            n_rec, _ = share.run()
        ref, fus, reads, pairs = share.first_batch
        base, ora_recs, n_checked = cpu_baseline(ref, fus, reads, pairs)
        got = share.ctxs[0].download()
        got = got[got["pair_idx"] < n_checked]
        base["gpu_records_equal_on_first_share"] = bool(got.tobytes() == ora_recs.tobytes())
        if not base["gpu_records_equal_on_first_share"]:
            raise SystemExit("bench.py: GPU records differ from the oracle on the sample")
    steps_requested = steps
    elapsed, stage = timed(steps)
    if elapsed < MIN_TIMED_S and not args.profile_run:
        # a timed region shorter than MIN_TIMED_S does not stand: the same loop again with as many steps as fill it
        # (every rank derives the same count from the reduced clock)
        steps = int(steps * MIN_TIMED_S * 1.15 / max(elapsed, 1e-6)) + 1
        elapsed, stage = timed(steps)
    fill_ms, pack_ms, finish_ms, plan_ms, launches = stage
    # the same resident, already planned batch run again without its planning: round 2's number, for comparison only
    k_rerun = max(3, steps // 4) if not args.profile_run else 1
    rerun_s, _ = timed(k_rerun, plan=False)
    if world > 1:
        nn = torch.tensor([share.total_pairs * steps, n_rec], dtype=torch.int64, device=red_dev)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_aligns, job_records = int(nn[0].item()), int(nn[1].item())
    else:
        total_aligns, job_records = share.total_pairs * steps, n_rec

    gather = None
    alone = None
    if world > 1:
        if os.environ.get("DEFUSE_BENCH_GATHER", "1") != "0":
            if os.environ.get("DEFUSE_BENCH_TEST_EXIT_RANK") == str(rank):  # tests: a rank that dies before the exchange
                os._exit(3)
            gather = gather_job(share, rank, world, backend, barrier, args.dump_records)       # raises on failure: the run fails
        # rank 0 alone on its share, the other ranks idle at the barrier
        if rank == 0:
            k = max(3, steps // 4)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                share.run()                   # planning included, as in the timed steps
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            rate = share.total_pairs * k / dt
            alone = {"aligns_per_s": rate, "steps": k, "aligns_per_step": share.total_pairs,
                     "efficiency_vs_this": (total_aligns / elapsed) / (world * rate),
                     "note": "rank 0 repeats its own share with the other ranks idle; efficiency = value / (n_gpus x this)"}
        barrier()

    if rank == 0:
        lib_hash = library_hash()
        t = share.ctxs[0].timing()
        rec_per_align = n_rec / share.total_pairs
        bytes_per_align = synth.algorithmic_bytes_per_align(workload["lq"], workload["lr"], workload["reads"], rec_per_align)
        n_launch = float(np.mean(launches))
        launch_ms = float(np.mean(fill_ms)) / max(1.0, n_launch)
        aligns_per_launch = share.total_pairs / max(1.0, n_launch)
        achieved = bytes_per_align * aligns_per_launch / (launch_ms * 1e-3) / 1e9
        cells = synth.cells_per_align(workload["lq"], workload["lr"])
        # counters are per launch: a launch is one upload (the whole batch at N = 1, UPLOAD_FUSIONS fusions of the configs[3] shape else)
        per_upload = dict(workload, fusions=share.n_pairs[0] // workload["reads"])
        suffix = "" if name == "config2" and not multi else "_config4"
        tr, tr_src = profile_block("pmc_traffic%s.json" % suffix, per_upload, lib_hash, same_size=not suffix)
        traffic = tr["hbm_bytes_per_launch"] if tr else None
        if tr and suffix and tr.get("aligns_per_launch"):
            traffic *= aligns_per_launch / tr["aligns_per_launch"]
        vi, vi_src = valu_issue(per_upload, lib_hash, launch_ms, suffix, aligns_per_launch)
        hbm = {"hbm_achieved": achieved, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": achieved / HBM_PEAK_GBS,
               "hbm_traffic_GBps": (traffic / (launch_ms * 1e-3) / 1e9) if traffic else None}
        if vi:
            # the unit that binds this integer kernel: VALU issue slots (4 cycles per wave instruction and SIMD, flat)
            roof = {"bound": "valu_issue", "achieved": vi["issue_cycles_flat4"], "peak": vi["kernel_cycles_nominal"], "unit": "issue cycles per SIMD and launch (2.4 GHz nominal)",
                    "frac": vi["frac_flat4"]}
        else:
            # no counters of this build at hand: the line can only answer the metric's HBM question
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
        roof.update(hbm)
        roof.update({"traffic": traffic, "traffic_source": tr_src,
                     "traffic_over_algorithmic": (traffic / (bytes_per_align * aligns_per_launch)) if traffic else None,
                     "kernel": "k_fill_fast", "kernel_ms": launch_ms, "launches_per_step": n_launch,
                     "algorithmic_bytes_per_align": round(bytes_per_align, 2),
                     "gcups_kernel": cells * aligns_per_launch / (launch_ms * 1e-3) / 1e9,
                     "valu_issue": vi, "valu_issue_source": vi_src, "library_source_hash": lib_hash,
                     "note": "integer DP: the binding unit is VALU issue, not HBM or MFMA (SURVEY 8(d)) - frac prices every VALU wave instruction at "
                             "4 issue cycles; the hbm_* fields answer the metric's HBM question (algorithmic bytes / kernel time against 8 TB/s); traffic >> "
                             "algorithmic bytes because tile checkpoints and tile maxima (needed for exact tie enumeration) stream through HBM, see "
                             "DESIGN.md 5; traffic / valu_issue are null unless the committed counters carry this library's source hash"})
        out = {
            "metric": "split-read DP aligns/sec", "value": total_aligns / elapsed, "unit": "aligns/s",
            "n_gpus": world, "steps": steps, "steps_requested": steps_requested, "warmup": args.warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "scaling_note": "N > 1 splits ONE job (BASELINE configs[3], 200 M aligns) over the GPUs; the default N = 1 line runs BASELINE configs[1] "
                            "(1 M aligns of another shape): the N = 1 point of the configs[3] curve is `bench.py --gpus 1 --workload config4`, and every "
                            "N > 1 line carries it per GPU as strong_scaling.n1_aligns_per_s",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic", "sched": library_build_flags().split()[0].split("=", 1)[-1],
            "hip_runtime": hip_path,
            "config": {"workload": "BASELINE %s%s: %s synthetic candidate fusions x %d reads, 2x%d bp (Lref %d), split-read DP + split search, "
                                   "bit-exact; whole job = %d aligns per step" % (
                                       "configs[1]" if name == "config2" else "configs[3]", " (custom sizes)" if custom else "",
                                       ("%dk" % (F // 1000)) if F < 1000000 else ("%gM" % (F / 1e6)), workload["reads"], workload["lq"],
                                       workload["lr"], F * workload["reads"]),
                       "aligns_per_step_this_gpu": share.total_pairs, "uploads_this_gpu": len(share.ctxs), "cells_per_align": cells,
                       "records_per_align": round(rec_per_align, 4), "job_records": job_records,
                       "parallelism": "contiguous fusion ranges over %d GPU(s), no data-path collective" % world},
            "roofline": roof,
            "step": "dsa_plan + dsa_run per upload: sweep planning (fusion order, in-fusion order, per-pair score bounds), reference "
                    "packing, DP fill with combine / replay in its tail, left-over replay, count, scan, emit; inputs resident in HBM",
            "stage_ms": {"plan": float(np.mean(plan_ms)), "pack": float(np.mean(pack_ms)), "fill": float(np.mean(fill_ms)),
                         "finish": float(np.mean(finish_ms))},
            "resident_rerun": {"value": (total_aligns // steps) * k_rerun / rerun_s,
                               "ms_per_step": rerun_s / k_rerun * 1e3, "steps": k_rerun,
                               "note": "dsa_run alone on the already planned resident batch (what round 2 reported as value)"},
            "replay_tiles_per_align": round(t.n_replay_tasks / max(1, share.n_pairs[0]), 4),
            "generic_replay_tiles_per_align": round(t.n_generic_tasks / max(1, share.n_pairs[0]), 4),
        }
        if gather is not None:
            out["gather"] = gather
            out["job_aligns_per_s_with_one_gather"] = total_aligns / (elapsed + steps * gather["ms"] * 1e-3)
        if alone is not None:
            out["one_gpu_same_share"] = alone
            out["strong_scaling"] = {"n_gpus": world, "aligns_per_s": total_aligns / elapsed, "n1_aligns_per_s": alone["aligns_per_s"],
                                     "efficiency": alone["efficiency_vs_this"],
                                     "n1_measured_on": "the same workload in this run: rank 0's share (1/%d of the job, same shape and upload size) repeated with the "
                                                       "other ranks idle; a GPU runs its uploads one after the other, so the whole job on one GPU runs at this rate "
                                                       "(`bench.py --gpus 1 --workload config4` measures exactly that)" % world}
        if base is not None:
            out["cpu_baseline"] = base
        if world == 1 and len(share.ctxs) == 1 and not args.profile_run and not multi and not args.no_sensitivity:
            out["sensitivity"] = sensitivity(dsa, synth, share.ctxs[0], workload, share.first_batch)
        if world == 1 and len(share.ctxs) == 1 and not args.profile_run:
            out["one_shot"] = one_shot(share.ctxs[0], share.first_batch)
        print(json.dumps(out), flush=True)
    share.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
