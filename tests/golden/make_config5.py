"""Writes tests/golden/config5/: the outputs of the REFERENCE'S OWN Perl glue (scripts/merge_clusters.pl,
remove_duplicates.pl, get_align_regions.pl, run from /root/reference in the build container) on the intermediate
files of the config-5-shaped end-to-end case (tests/e2e_case.py), whose tool stages are computed by the oracles.
tests/test_end_to_end.py regenerates the inputs from the same seed and compares every stage of the product chain.

    python tests/golden/make_config5.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
SCRIPTS = "/root/reference/scripts"


def perl(script, args, stdin=None):
    r = subprocess.run(["perl", os.path.join(SCRIPTS, script)] + args, input=stdin, capture_output=True, text=True,
                       env=dict(os.environ, PERL_HASH_SEED="0"))
    assert r.returncode == 0, r.stderr
    return r.stdout


def oracle_chain(case, workdir, glue):
    """The whole chain with the oracles for the four tools and `glue(step, args, stdin)` for the three Perl steps.
    Returns a dict of stage name -> text."""
    from oracle import clustermatepairs_oracle as co, setcover_oracle as so, dosplitalign_oracle as do
    st = {}
    cl_paths = []
    for sp in case["spanning"]:
        txt, n = co.clustermatepairs(open(sp).readlines(), case["ufrag"], case["sfrag"], 0.95, 5, em="c")
        name = "clusters." + os.path.basename(sp).split(".", 1)[1]
        st[name] = txt
        path = os.path.join(workdir, name)
        open(path, "w").write(txt)
        cl_paths.append(path)
    st["clusters.all"] = glue("merge_clusters", cl_paths, None)
    p_all = os.path.join(workdir, "clusters.all")
    open(p_all, "w").write(st["clusters.all"])
    st["clusters.sc.all"] = so.setcover(p_all, 5)
    st["clusters.sc"] = glue("remove_duplicates", ["5"], st["clusters.sc.all"])
    st["clusters.sc.regions"] = glue("get_align_regions", [], st["clusters.sc"])
    p_reg = os.path.join(workdir, "clusters.sc.regions")
    open(p_reg, "w").write(st["clusters.sc.regions"])
    common = (case["fasta"], case["exons"], case["ufrag"], case["sfrag"], case["minread"], case["maxread"], p_reg)
    st["splitreads.alignments"] = do.dosplitalign(*common, case["improper"], case["seq1"], case["seq2"])
    rows = sorted(st["splitreads.alignments"].splitlines(True), key=lambda l: int(l.split("\t")[0]))     # sort -n -k 1 (stable)
    st["splitreads.alignments.sorted"] = "".join(rows)
    p_al = os.path.join(workdir, "splitreads.alignments.sorted")
    open(p_al, "w").write(st["splitreads.alignments.sorted"])
    seq, brk, pred = do.evalsplitalign(*common, p_al)
    st["splitreads.seq"], st["splitreads.break"], st["splitreads.predalign"] = seq, brk, pred
    return st


def main():
    from tests import e2e_case
    out = os.path.join(HERE, "config5")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        case = e2e_case.build(os.path.join(tmp, "case"))
        st = oracle_chain(case, tmp, lambda step, args, stdin: perl(step + ".pl", args, stdin))
        for name in ("clusters.all", "clusters.sc", "clusters.sc.regions"):
            open(os.path.join(out, name + ".perl.txt"), "w").write(st[name])
        digest = {}
        for k in ("fasta", "exons", "improper", "seq1", "seq2"):
            digest[os.path.basename(case[k])] = hashlib.md5(open(case[k], "rb").read()).hexdigest()
        for sp in case["spanning"]:
            digest[os.path.basename(sp)] = hashlib.md5(open(sp, "rb").read()).hexdigest()
        json.dump(digest, open(os.path.join(out, "inputs.md5.json"), "w"), indent=1, sort_keys=True)
        print({k: len(v.splitlines()) for k, v in st.items()})
        print(st["splitreads.break"])


if __name__ == "__main__":
    main()
