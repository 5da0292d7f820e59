"""Builds the in-tree native artefacts: the HIP C-ABI library (gfx950) and the tool binaries.

    python -m defuse_amd.build            # everything
    python -m defuse_amd.build --lib      # only libdefuse_dsa.so

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with gpurun.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdefuse_dsa.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


LIB_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared"]
# The split-read kernels (dsa_api.hip) only: LLVM's iterative ILP scheduler.  The sweep of a tile row is 64 dependent maxima
# beside independent additions, table reads and row-maximum updates in one straight block; the default scheduler orders it for
# register pressure, this one interleaves the independent work into the chain's issue gaps: fill kernel 3.05 -> 2.88 ms on one
# box (profiles/r03/compiler_flags_ab.txt; max-ilp, max-memory-clause, iterative-minreg and -maxocc are slower or equal).
DSA_FLAGS = ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]


def lib_sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + \
           [os.path.join(ROOT, "include", h) for h in ("defuse_dsa.h", "defuse_sc.h", "defuse_mpe.h", "defuse_la.h", "defuse_hc.h", "defuse_cov.h", "defuse_cmp.h")]


_dsa_flags_probe = {}


def effective_dsa_flags():
    """DSA_FLAGS if this hipcc accepts them, else [] (the default scheduler: a slower fill kernel).  Probed once per
    compiler with an empty translation unit, so that a genuine compile error in the sources is never taken for a
    rejected flag, and so that the hash and dsa_build_flags() say what was really used."""
    if HIPCC not in _dsa_flags_probe:
        ok = True
        if DSA_FLAGS:
            r = subprocess.run([HIPCC, "--offload-arch=" + ARCH] + DSA_FLAGS + ["-x", "hip", "-c", os.devnull, "-o", os.devnull],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            ok = r.returncode == 0
            if not ok:
                print("build: this hipcc rejects %s; dsa_api.hip is compiled with the default scheduler (slower fill kernel)\n%s" %
                      (" ".join(DSA_FLAGS), r.stdout[-400:]), flush=True)
        _dsa_flags_probe[HIPCC] = list(DSA_FLAGS) if ok else []
    return list(_dsa_flags_probe[HIPCC])


def sched_name(flags=None):
    """"iterative-ilp" | "default": the instruction scheduler dsa_api.hip is built with (bench.py prints it)."""
    flags = effective_dsa_flags() if flags is None else flags
    for f in flags:
        if f.startswith("-amdgpu-sched-strategy="):
            return f.split("=", 1)[1]
    return "default"


def source_hash(extra_flags=(), dsa_flags=None):
    """12 hex digits over the library's sources and the compile flags ACTUALLY used (a build that fell back to the default
    scheduler has another hash than one with DSA_FLAGS).  Compiled into the library (dsa_version()) and written into every
    profile JSON (profiles/microbench/*.sh), so that bench.py can tell whether committed counters were taken on the kernels
    it is running."""
    import hashlib
    if dsa_flags is None:
        dsa_flags = effective_dsa_flags()
    h = hashlib.sha256()
    for p in lib_sources():
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    h.update(" ".join(LIB_FLAGS + list(dsa_flags) + list(extra_flags)).encode())
    return h.hexdigest()[:12]


def build_lib(force=False):
    srcs = lib_sources()
    if force or _newer(LIB, srcs + [os.path.abspath(__file__)]):      # the flags live in this file
        compile_lib(LIB)
    return LIB


def compile_lib(out, extra_flags=()):
    """dsa_api.hip into an object of its own (with the scheduler flag where the compiler has it), then the library from it
    and the other sources.  Object and library are written under names of this process and renamed into place: several
    builders (ranks that all call build_lib) never read each other's half-written files."""
    extra = list(extra_flags)
    dsa_flags = effective_dsa_flags()
    define = ["-DDSA_BUILD_HASH=\"%s\"" % source_hash(extra, dsa_flags),
              "-DDSA_BUILD_FLAGS=\"sched=%s %s\"" % (sched_name(dsa_flags), " ".join(f for f in LIB_FLAGS + extra if f not in ("-fPIC", "-shared")))]
    obj_dir = os.path.join(HERE, "_build")
    os.makedirs(obj_dir, exist_ok=True)
    tag = ".%d.tmp" % os.getpid()
    obj = os.path.join(obj_dir, os.path.basename(out) + ".dsa_api.o" + tag)
    tmp_out = out + tag
    try:
        _run([HIPCC] + [f for f in LIB_FLAGS if f != "-shared"] + dsa_flags + extra + define + ["-c", "-o", obj, os.path.join(CSRC, "dsa_api.hip")])
        _run([HIPCC] + LIB_FLAGS + extra + define + ["-o", tmp_out, obj] +
             [os.path.join(CSRC, f) for f in ("sc_api.hip", "mpe_api.hip", "la_api.hip", "hc_api.hip", "cov_api.hip", "cmp_api.hip")])
        os.replace(tmp_out, out)
    finally:
        for f in (obj, tmp_out):
            if os.path.exists(f):
                os.unlink(f)
    return out


TOOLS = ["dosplitalign", "evalsplitalign", "setcover", "clustermatepairs", "localalign", "defuse_glue", "calccov"]


def build_tools(force=False):
    """The drop-in tool binaries (C++17 host code on the C ABI) -> bin/."""
    bindir = os.path.join(ROOT, "bin")
    os.makedirs(bindir, exist_ok=True)
    lib = build_lib(force)
    outs = []
    for t in TOOLS:
        src = os.path.join(ROOT, "tools_src", t + ".cpp")
        out = os.path.join(bindir, t)
        deps = [src, os.path.join(ROOT, "tools_src", "defuse_host.hpp"), os.path.join(ROOT, "tools_src", "evaluate.hpp"), os.path.join(ROOT, "include", "defuse_dsa.h"),
                os.path.join(ROOT, "include", "defuse_sc.h"), os.path.join(ROOT, "include", "defuse_mpe.h"),
                os.path.join(ROOT, "include", "defuse_la.h"), os.path.join(ROOT, "include", "defuse_cov.h"),
                os.path.join(ROOT, "include", "defuse_cmp.h"), lib]
        if force or _newer(out, deps):
            if t == "dosplitalign":          # opens the C-ABI library at run time (on a helper thread), see tools_src/dosplitalign.cpp
                _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-pthread", "-o", out, src, "-ldl"])
            else:
                _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-pthread", "-o", out, src, lib,
                      "-Wl,-rpath,$ORIGIN/../defuse_amd"])
        outs.append(out)
    return outs


def build_sanitized(kind, force=False):
    """The host code of the tools under a sanitizer (the CPU build is where sanitizers run; the GPU pool
    has none): kind "asan" = -fsanitize=address,undefined, "tsan" = -fsanitize=thread, binaries in bin/<kind>/.  They link
    the same C-ABI library; tests/test_sanitizers.py runs their host stages (threaded parsers, binners, glue) with them."""
    flags = {"asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], "tsan": ["-fsanitize=thread"]}[kind]
    bindir = os.path.join(ROOT, "bin", kind)
    os.makedirs(bindir, exist_ok=True)
    lib = build_lib()
    outs = {}
    for t in TOOLS:
        src = os.path.join(ROOT, "tools_src", t + ".cpp")
        out = os.path.join(bindir, t)
        if force or _newer(out, [src, os.path.join(ROOT, "tools_src", "defuse_host.hpp"), os.path.join(ROOT, "tools_src", "evaluate.hpp"), lib]):
            cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread"] + flags + ["-o", out, src]
            cmd += ["-ldl"] if t == "dosplitalign" else [lib, "-Wl,-rpath,$ORIGIN/../../defuse_amd"]
            _run(cmd)
        outs[t] = out
    return outs


def build_oracle(force=False):
    """The checkers: oracle/libdsa_oracle.so, oracle/libmpe_oracle.so and - only where /root/reference is present, i.e. in
    the build container - oracle/_ref/libasa_ref.so from the reference's own asa136.C / asa241.C (oracle/Makefile)."""
    out = os.path.join(ROOT, "oracle", "libdsa_oracle.so")
    out2 = os.path.join(ROOT, "oracle", "libmpe_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", "dsa_oracle.c"), os.path.join(ROOT, "include", "defuse_dsa.h")]
    srcs2 = [os.path.join(ROOT, "oracle", "mpe_oracle.c")]
    ref_missing = os.path.exists("/root/reference/tools/asa136.C") and not (
        os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libasa_ref.so")) and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libfaidx_ref.so")))
    if force or _newer(out, srcs) or _newer(out2, srcs2) or ref_missing:
        _run(["make", "-C", os.path.join(ROOT, "oracle")] + (["-B"] if force else []))
    return out


def main(argv):
    force = "--force" in argv
    build_lib(force)
    if "--lib" not in argv:
        build_tools(force)
        build_oracle(force)


if __name__ == "__main__":
    main(sys.argv[1:])
