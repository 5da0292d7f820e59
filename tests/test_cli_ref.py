"""Command lines of the drop-in tools against the REFERENCE'S OWN command-line library: the header-only TCLAP vendored under
include/tclap, driven by oracle/tclap_ref.cpp (built by oracle/Makefile into oracle/_ref/tclap_ref where /root/reference is
present).  The specifications below are the argument definitions of the reference's mains (file:line); for every tool and a list
of command lines — help, version, missing / unknown / repeated / valueless / malformed arguments, `--` — the drop-in binary must
print what TCLAP prints, byte for byte on stdout and stderr, and exit with its code.  Parsed values are compared too."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "tclap_ref")

# (flag, name, description, value type[:label shown in the usage text], required) in the order the reference defines them
SPECS = {
    "dosplitalign": ("Fusion sequence prediction by split reads", [                     # tools/dosplitalign.cpp:43-56
        ("f", "fasta", "Reference Fasta", "string", 1), ("e", "exons", "Exon Regions Filename", "string", 1),
        ("u", "ufrag", "Fragment Length Mean", "float", 1), ("s", "sfrag", "Fragment Length Standard Deviation", "float", 1),
        ("n", "minread", "Minimum Read Length", "int:integer", 1), ("x", "maxread", "Maximum Read Length", "int:integer", 1),
        ("r", "regions", "Fusion Regions Filename", "string", 1), ("i", "improper", "Improper Alignments Sam Filename", "string", 1),
        ("1", "seq1", "End 1 Sequences", "string", 1), ("2", "seq2", "End 2 Sequences", "string", 1),
        ("a", "align", "Split Alignments Filename", "string", 1)]),
    "clustermatepairs": ("Mate Pair Clustering Tool", [                                  # tools/clustermatepairs.cpp:400-406
        ("a", "align", "Alignments Filename", "string", 1), ("c", "clusters", "Output Clusters Filename", "string", 1),
        ("u", "fragmentmean", "Fragment Length Mean", "float:integer", 1),
        ("s", "fragmentstddev", "Fragment Length Standard Deviation", "float:integer", 1),
        ("p", "precision", "Precision", "float:double", 1), ("m", "minclustersize", "Minimum Cluster Size", "int:integer", 1)]),
    "setcover": ("Set cover for maximum parsimony", [                                    # tools/setcover.cpp:120-123
        ("c", "clusters", "Clusters Filename", "string", 1), ("m", "minclustersize", "Minimum Cluster Size", "int:integer", 1),
        ("o", "outclust", "Output Clusters Filename", "string", 1)]),
    "evalsplitalign": ("Fusion sequence prediction by split reads", [                   # tools/evalsplitalign.cpp:43-56
        ("f", "fasta", "Reference Fasta", "string", 1), ("e", "exons", "Exon Regions Filename", "string", 1),
        ("u", "ufrag", "Fragment Length Mean", "float", 1), ("s", "sfrag", "Fragment Length Standard Deviation", "float", 1),
        ("n", "minread", "Minimum Read Length", "int:integer", 1), ("x", "maxread", "Maximum Read Length", "int:integer", 1),
        ("r", "regions", "Fusion Regions Filename", "string", 1), ("a", "align", "Split Alignments Filename", "string", 1),
        ("q", "seq", "Sequences Filename", "string", 1), ("b", "break", "Break Positions Filename", "string", 1),
        ("p", "predalign", "Prediction Split Alignments Filename", "string", 1)]),
    "localalign": ("Local realignment tool", [                                           # tools/localalign.cpp:34-38
        ("m", "match", "Match Score", "int:int", 1), ("x", "mismatch", "Mismatch Score", "int:int", 1), ("g", "gap", "Gap Score", "int:int", 1),
        ("t", "threshold", "Percent Perfect Threshold", "float", 0)]),
    "calccov": ("Calculate covariance stats from concordant alignments", [              # tools/calccov.cpp:80-89
        ("c", "conc", "Concordant Sam Filename", "string", 1), ("g", "genetran", "Gene Transcripts Filename", "string", 1),
        ("l", "len", "Spanning Length Samples Filename", "string", 1), ("p", "pos", "Split Position Samples Filename", "string", 1),
        ("m", "min", "Split Minimum Samples Filename", "string", 1), ("d", "density", "Covariance Sampling Density", "float", 1),
        ("a", "anchor", "Gene Transcripts Filename", "int:integer", 1), ("t", "trim", "Trim Length for Spanning Alignments", "int:integer", 1),
        ("", "multiexon", "Use Multi-Exon Transcripts", "switch", 0)]),
}


@pytest.fixture(scope="module")
def driver(built):
    from defuse_amd import build
    build.build_tools()
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/tclap_ref not built (reference headers absent)")
    return DRIVER


def reference(driver, tool, args):
    message, spec = SPECS[tool]
    cmd = [driver, tool, message, str(len(spec))]
    for (flag, name, desc, typ, req) in spec:
        cmd += [flag, name, desc, typ, str(req)]
    return subprocess.run(cmd + ["--"] + args, capture_output=True, text=True)


def failing_lines(tool):
    """Command lines that end inside the parser (so the tool never gets to its work)."""
    _, spec = SPECS[tool]
    first, second = spec[0], spec[1]
    ints = [s for s in spec if s[3].startswith("int")]
    floats = [s for s in spec if s[3].startswith("float")]
    lines = [["--help"], ["-h"], ["--version"], [], ["-" + first[0], "x"], ["--" + first[1], "x", "--" + first[1], "y"],
             ["--bogus", "1"], ["stray"], ["-" + first[0]], ["-" + first[0] + "x"], ["--" + first[1] + "=x"],
             ["-" + first[0], "x", "-" + second[0]], ["--", "-" + first[0], "x"]]
    if ints:
        lines += [["-" + ints[0][0], "3.5"], ["-" + ints[0][0], "abc"], ["--" + ints[0][1], "7x"]]
    if floats:
        lines += [["-" + floats[0][0], "1.5e"], ["-" + floats[0][0], "x"]]
    if tool == "calccov":
        lines += [["--multiexon", "--multiexon"]]
    return lines


@pytest.mark.parametrize("tool", sorted(SPECS))
def test_parser_output_equals_tclap(driver, tool):
    binary = os.path.join(ROOT, "bin", tool)
    for args in failing_lines(tool):
        ref = reference(driver, tool, args)
        got = subprocess.run([binary] + args, capture_output=True, text=True, stdin=subprocess.DEVNULL)
        assert ref.returncode != 0 or args[0] in ("--help", "-h", "--version"), args     # these lines are meant not to reach the tool
        assert (got.returncode, got.stdout, got.stderr) == (ref.returncode, ref.stdout, ref.stderr), (tool, args)


def test_dosplitalign_fused_options_exist_only_on_request(driver):
    """The five options of the fused mode are not part of the command line unless DEFUSE_FUSED=1: without it `--clusters` is an
    unknown argument exactly as it is for the reference's parser, with it the help text lists them."""
    binary = os.path.join(ROOT, "bin", "dosplitalign")
    env = {k: v for k, v in os.environ.items() if k != "DEFUSE_FUSED"}
    for args in (["--clusters", "x"], ["--sorted"], ["-q", "x"]):
        ref = reference(driver, "dosplitalign", args)
        got = subprocess.run([binary] + args, capture_output=True, text=True, env=env)
        assert (got.returncode, got.stdout, got.stderr) == (ref.returncode, ref.stdout, ref.stderr), args
    got = subprocess.run([binary, "--help"], capture_output=True, text=True, env=dict(env, DEFUSE_FUSED="1"))
    assert got.returncode == 0 and "--clusters <string>" in got.stdout and "--sorted" in got.stdout


def test_parsed_values_equal_tclap(driver):
    """Values TCLAP accepts: negative numbers after a flag (the pipeline's `localalign -m 10 -x -5 -g -5 -t 0.8`), exponents."""
    ref = reference(driver, "localalign", ["-m", "10", "-x", "-5", "-g", "-5", "-t", "0.8"])
    assert ref.returncode == 0 and ref.stdout == "match\t10\nmismatch\t-5\ngap\t-5\nthreshold\t0.8\n"
    got = subprocess.run([os.path.join(ROOT, "bin", "localalign"), "-m", "10", "-x", "-5", "-g", "-5", "-t", "0.8"], input="", capture_output=True, text=True)
    assert got.returncode == 0 and got.stdout == "" and got.stderr == ""
    ref = reference(driver, "localalign", ["-m", "10", "-x", "-5", "-g", "-5", "-t", "8e-1"])
    assert ref.returncode == 0 and ref.stdout.endswith("threshold\t0.8\n")
