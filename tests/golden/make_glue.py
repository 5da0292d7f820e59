"""Generates tests/golden/glue/: inputs for the five glue steps and the outputs of the reference's own Perl scripts on them
(scripts/merge_clusters.pl, get_align_regions.pl, remove_duplicates.pl, filter_unmatched.pl, divide_sam_chr_pairs.pl).
Run in the build container only (needs /root/reference and perl); the fixtures it writes are what travels.

    python tests/golden/make_glue.py

Where a script's output order is Perl's hash order (get_align_regions, remove_duplicates, the per-fragment file order of
divide_sam_chr_pairs) the recorded output is what perl printed (PERL_HASH_SEED=0); the tests compare in the order-free way
tests/test_glue.py states per script."""
import os
import random
import shutil
import subprocess

REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "glue")
ENV = dict(os.environ, PERL_HASH_SEED="0", PERL_PERTURB_KEYS="0")


def perl(script, args=(), stdin=None, cwd=None):
    r = subprocess.run(["perl", os.path.join(REF, script)] + list(args), input=stdin, capture_output=True, text=True, env=ENV, cwd=cwd)
    assert r.returncode == 0, (script, r.stderr)
    return r.stdout


def cluster_lines(rng, ids, dup=False):
    lines = []
    frag = rng.randrange(1000)
    for cid in ids:
        ca, cb = rng.choice(["chr1", "chr2", "ENSG01|ENST07"]), rng.choice(["chr3", "chrX"])
        sa, sb = rng.choice("+-"), rng.choice("+-")
        n = rng.randrange(2, 9)
        base_a, base_b = rng.randrange(1000, 90000), rng.randrange(1000, 90000)
        pos = []
        for k in range(n):
            if dup and pos and rng.random() < 0.4:
                a, b = rng.choice(pos)                      # a PCR duplicate: same pair of positions
            else:
                a, b = base_a + rng.randrange(0, 200), base_b + rng.randrange(0, 200)
            pos.append((a, b))
            frag += rng.randrange(1, 4)
            lines.append("%d\t0\t%d\t%d\t%s\t%s\t%d\t%d\n" % (cid, frag, rng.randrange(2), ca, sa, a, a + 49))
            lines.append("%d\t1\t%d\t%d\t%s\t%s\t%d\t%d\n" % (cid, frag, rng.randrange(2), cb, sb, b, b + 49))
    return lines


def sam_lines(rng):
    refs = ["chr1", "chr2", "chr3", "ENSG01|ENST07", "ENSG02|ENST09"]
    lines = ["@HD\tVN:1.0\n", "@SQ\tSN:chr1\tLN:100000\n"]
    for frag in range(0, 120, rng.choice([1, 1, 2])):
        ends = [1, 2] if rng.random() < 0.8 else [rng.choice([1, 2])]
        for e in ends:
            for _ in range(rng.choice([1, 1, 1, 2, 3])):
                ref = rng.choice(refs)
                flag = rng.choice([0, 16, 64, 80])
                seq = "".join(rng.choice("ACGT") for _ in range(rng.choice([36, 50, 76])))
                lines.append("%d/%d\t%d\t%s\t%d\t255\t%dM\t*\t0\t0\t%s\t%s\n" % (frag, e, flag, ref, rng.randrange(1, 90000), len(seq), seq, "I" * len(seq)))
    return lines


def main():
    rng = random.Random(11)
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)

    def put(name, text):
        with open(os.path.join(OUT, name), "w") as f:
            f.write(text)

    # merge_clusters: three files, ids that repeat across files and jump inside a file
    files = []
    for k, ids in enumerate(([0, 1, 5], [0, 2], [7])):
        name = "merge_in%d.txt" % k
        put(name, "".join(cluster_lines(rng, ids)))
        files.append(name)
    put("merge_out.txt", perl("merge_clusters.pl", files, cwd=OUT))

    # get_align_regions
    text = "".join(cluster_lines(rng, [0, 1, 2, 3, 10, 11]))
    put("regions_in.txt", text)
    put("regions_out.perl.txt", perl("get_align_regions.pl", stdin=text))

    # remove_duplicates, threshold 3
    text = "".join(cluster_lines(rng, list(range(12)), dup=True))
    put("dups_in.txt", text)
    put("dups_out.perl.txt", perl("remove_duplicates.pl", ["3"], stdin=text))

    # filter_unmatched | divide_sam_chr_pairs (alignjob.pl:330)
    sam = "".join(sam_lines(rng))
    put("improper.sam", sam)
    body = "".join(l for l in sam.splitlines(True) if not l.startswith("@"))     # the script has no header handling
    matched = perl("filter_unmatched.pl", stdin=body)
    put("matched.sam", matched)
    put("trans_chr.txt", "ENSG01\tENST07\tchr2\t+\t100\t200\t\nENSG02\tENST09\tchr9\t-\t5\t50\t\n")
    os.makedirs(os.path.join(OUT, "div"))
    listing = perl("divide_sam_chr_pairs.pl", ["-t", "trans_chr.txt", "-p", "div/"], stdin=matched, cwd=OUT)
    put("div_list.txt", listing)
    print("wrote", sorted(os.listdir(OUT)), sorted(os.listdir(os.path.join(OUT, "div"))))


if __name__ == "__main__":
    main()
