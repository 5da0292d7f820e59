"""Regenerates the inputs of the known-answer smoke vector.

The EXPECTED outputs in tests/golden/smoke/expected.* are NOT produced by this script: they are the
outputs of the reference's own sources on these inputs, recorded in SURVEY.md Appendix A at survey
time.  This script only rebuilds the input files from the documented recipe (python random, seed 7).
"""
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def main(out=os.path.join(HERE, "smoke")):
    os.makedirs(out, exist_ok=True)
    random.seed(7)
    rnd = lambda n: "".join(random.choice("ACGT") for _ in range(n))
    A = rnd(3000)
    B = rnd(3000)
    with open(os.path.join(out, "ref.fa"), "w") as f:
        for name, s in (("chrA", A), ("chrB", B)):
            f.write(">%s\n" % name)
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")
    with open(os.path.join(out, "exons.txt"), "w") as f:
        f.write("geneA\ttxA\tchrA\t+\t100\t2900\t\n")
        f.write("geneB\ttxB\tchrB\t+\t100\t2900\t\n")
    with open(os.path.join(out, "regions.txt"), "w") as f:
        f.write("0\t0\tchrA\t+\t500\t600\n")
        f.write("0\t1\tchrB\t-\t1100\t1200\n")
    fused = A[:650] + B[999:]
    f1 = open(os.path.join(out, "reads.1.fastq"), "w")
    f2 = open(os.path.join(out, "reads.2.fastq"), "w")
    sam = open(os.path.join(out, "improper.sam"), "w")
    for fi, off in enumerate(range(605, 650, 3)):
        r1 = fused[off:off + 50]
        mate = B[1149 + fi:1199 + fi]
        f1.write("@%d/1\n%s\n+\n%s\n" % (fi, r1, "I" * 50))
        f2.write("@%d/2\n%s\n+\n%s\n" % (fi, rc(mate), "I" * 50))
        sam.write("%d/2\t16\tchrB\t%d\t255\t50M\t*\t0\t0\t%s\t%s\n" % (fi, 1150 + fi, mate, "I" * 50))
    f1.close(); f2.close(); sam.close()


if __name__ == "__main__":
    main()
