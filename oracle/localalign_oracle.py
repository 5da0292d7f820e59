"""CPU restatement of the reference `localalign` tool (tools/localalign.cpp:31-92) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(bin/localalign -> libdefuse_dsa.so -> HIP kernels) never does.

Parity unpinned: the reference holds no test or golden vector for localalign, and neither the tool nor
SimpleAligner.cpp can be compiled here (Common.h pulls in Boost headers the image lacks).  The matrix
recursion is restated in oracle/dsa_oracle.c (ora_simple_align, SimpleAligner.cpp:24-64); this module
restates the text protocol around it:

  * stdin lines `id \\t reference \\t sequence` (split on tabs, >= 3 fields, extra fields ignored,
    localalign.cpp:62-82); an empty line or a short line stops the run with exit status 1 *after* the
    lines before it were answered (the reference prints as it goes);
  * `score = Align(reference, sequence)`, `maxScore = len(sequence) * match` (int), `percent =
    (double) score / (double) maxScore`; lines with `percent < threshold` are dropped (:84-93);
  * output `id \\t score \\t percent` with the default ostream formatting of a double (%g, 6 digits).
"""
import ctypes
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _oracle():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(_HERE, "libdsa_oracle.so"))
        _lib.ora_simple_align.restype = ctypes.c_int
        _lib.ora_simple_align.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                          ctypes.c_char_p, ctypes.c_int]
    return _lib


def simple_align(match, mismatch, gap, reference, sequence):
    """SimpleAligner::Align on two byte strings."""
    reference, sequence = bytes(reference), bytes(sequence)
    return _oracle().ora_simple_align(match, mismatch, gap, reference, len(reference), sequence, len(sequence))


def simple_align_py(match, mismatch, gap, reference, sequence):
    """The same score written independently (row by row, two rolling rows) — a cross-check of the C restatement."""
    reference, sequence = bytes(reference), bytes(sequence)
    prev = [0] * (len(reference) + 1)          # j = 0
    best = 0
    for j in range(1, len(sequence) + 1):
        cur = [j * gap] + [0] * len(reference)
        q = sequence[j - 1]
        for i in range(1, len(reference) + 1):
            v = max(prev[i - 1] + (match if reference[i - 1] == q else mismatch), cur[i - 1] + gap, prev[i] + gap)
            cur[i] = v
            if v > best:
                best = v
        prev = cur
    return best


def format_double(x):
    """operator<<(ostream&, double) with default flags: %g with 6 significant digits."""
    if math.isnan(x):
        return "-nan" if math.copysign(1.0, x) < 0 else "nan"
    return "%g" % x


def run(lines, match, mismatch, gap, threshold=0.0):
    """Returns (stdout_text, stderr_text, exit_status) of `localalign -m -x -g [-t]` fed with `lines`."""
    out, err = [], []
    for n, line in enumerate(lines, 1):
        line = line.rstrip("\n")
        if len(line) == 0:
            err.append("Error: Empty line %d\n" % n)
            return "".join(out), "".join(err), 1
        f = line.split("\t")
        if len(f) < 3:
            err.append("Error: Format error for line %d\n" % n)
            return "".join(out), "".join(err), 1
        score = simple_align(match, mismatch, gap, f[1].encode(), f[2].encode())
        max_score = int(np.int32(np.uint64(len(f[2]) * match & 0xFFFFFFFFFFFFFFFF).astype(np.uint32)))  # size_t product -> int
        if max_score == 0:
            percent = -math.nan if score == 0 else math.copysign(math.inf, score)     # x86: 0.0/0.0 is -nan
        else:
            percent = float(score) / float(max_score)
        if percent < threshold:
            continue
        out.append("%s\t%d\t%s\n" % (f[0], score, format_double(percent)))
    return "".join(out), "".join(err), 0
