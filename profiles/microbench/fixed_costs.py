"""Fixed costs of a short-lived GPU tool process on this box (profiles/microbench/hip_fixed_costs.hip): runtime start, first
kernel, allocations, and the time between the last line of main() and the moment the parent sees the process gone.
Usage: python profiles/microbench/fixed_costs.py"""
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "hip_fixed_costs")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value", "-o", exe, os.path.join(HERE, "hip_fixed_costs.hip")])
for label, args, env in (("4 GiB of device memory freed before exit", ["4096", "64"], {}),
                         ("4 GiB of device memory left to the system, _exit", ["4096", "64", "nofree"], {"PROBE_UNDERSCORE_EXIT": "1"}),
                         ("1 GiB of device memory left to the system, _exit", ["1024", "64", "nofree"], {"PROBE_UNDERSCORE_EXIT": "1"}),
                         ("16 GiB of device memory left to the system, _exit", ["16384", "64", "nofree"], {"PROBE_UNDERSCORE_EXIT": "1"})):
    t0 = time.time()
    p = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, PROBE_T0=repr(t0), **env))
    t1 = time.time()
    print("== %s: rc %d, %.3f s wall" % (label, p.returncode, t1 - t0))
    print(p.stdout.rstrip())
    last = [l for l in p.stdout.splitlines() if l.startswith("main ends at wall")]
    if last:
        print("main's last line -> process gone: %.1f ms" % (1e3 * (t1 - float(last[0].split()[-1]))))
    print(p.stderr[-500:])
