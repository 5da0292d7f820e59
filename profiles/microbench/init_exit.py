"""Start and end of a GPU process against its number of HIP streams and GPU_MAX_HW_QUEUES (hip_init_exit.hip); three runs each.
Usage: python profiles/microbench/init_exit.py"""
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "hip_init_exit")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value", "-o", exe, os.path.join(HERE, "hip_init_exit.hip")])
for q in (None, "1", "2"):
    for ns, extra in ((0, []), (1, []), (2, []), (4, []), (8, []), (4, ["reset"])):
        for rep in range(3):
            env = dict(os.environ)
            if q:
                env["GPU_MAX_HW_QUEUES"] = q
            t0 = time.time()
            p = subprocess.run([exe, str(ns)] + extra, capture_output=True, text=True, env=env)
            t1 = time.time()
            lines = p.stdout.strip().splitlines()
            end = float(lines[-1].split()[-1]) if lines and lines[-1].startswith("main ends") else t1
            print("GPU_MAX_HW_QUEUES=%s streams=%d%s: wall %.0f ms, exit %.0f ms | %s" % (q or "unset", ns, " +reset" if extra else "", 1e3 * (t1 - t0), 1e3 * (t1 - end), lines[0] if lines else p.stderr[-200:]))
# a process that never touches the GPU, for the spawn cost of this harness
t0 = time.time(); subprocess.run(["/bin/true"]); print("spawn of /bin/true: %.1f ms" % (1e3 * (time.time() - t0)))
