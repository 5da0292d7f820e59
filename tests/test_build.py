"""The build's bookkeeping (defuse_amd/build.py): the source hash in dsa_version() covers the flags that were ACTUALLY used,
so counters taken on a library built with the scheduler flag are never accepted for one built without it."""
import os
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(hipcc):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from defuse_amd import build\n"
            "print(build.sched_name(), build.source_hash(), ' '.join(build.effective_dsa_flags()))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPCC=hipcc), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout.strip().splitlines()[-1].split(None, 2), r.stdout


def test_a_compiler_without_the_scheduler_flag_gives_another_hash(tmp_path):
    """HIPCC pointing at a wrapper that rejects -amdgpu-sched-strategy (as an older compiler would): the build falls back to
    the default scheduler, says so, and hashes the flags it really used."""
    wrapper = tmp_path / "hipcc_without_the_flag"
    wrapper.write_text("#!/bin/sh\nfor a in \"$@\"; do case \"$a\" in -amdgpu-sched-strategy=*) echo \"clang: Unknown command line argument '$a'\" >&2; exit 1;; esac; done\n"
                       "exec /opt/rocm/bin/hipcc \"$@\"\n")
    wrapper.chmod(wrapper.stat().st_mode | stat.S_IEXEC)
    (sched, h, *flags), _ = _probe("/opt/rocm/bin/hipcc")
    (sched_fb, h_fb, *flags_fb), out_fb = _probe(str(wrapper))
    assert sched == "iterative-ilp" and flags and "iterative-ilp" in flags[0]
    assert sched_fb == "default" and not flags_fb and "rejects" in out_fb
    assert h != h_fb and len(h) == len(h_fb) == 12


def test_the_library_says_what_it_was_built_with():
    """dsa_build_flags() of the built library names the scheduler, and dsa_version() carries the hash build.py computes for
    these sources and flags (both are what bench.py prints and checks committed counters against)."""
    import ctypes
    sys.path.insert(0, ROOT)
    from defuse_amd import build
    lib = ctypes.CDLL(build.build_lib())
    lib.dsa_build_flags.restype = ctypes.c_char_p
    lib.dsa_version.restype = ctypes.c_char_p
    flags = lib.dsa_build_flags().decode()
    assert flags.startswith("sched=" + build.sched_name()) and "--offload-arch=gfx950" in flags, flags
    assert lib.dsa_version().decode().split()[-1] == build.source_hash()
