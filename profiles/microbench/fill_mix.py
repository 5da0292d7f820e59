"""Static instruction mix of the hot loop of k_fill_fast<0> (the row sweep over one 64-column tile): hipcc --save-temps on
defuse_amd/csrc/dsa_api.hip (cross-compiles here, no GPU needed), the three basic blocks with the most v_pk_maximum3_f16,
VALU opcodes sorted into the two issue classes measured by profiles/microbench/valu_rate*.hip (2 cycles per wave:
v_add_u32 / v_sub / v_xor / v_mov / v_cndmask / v_cmp / shifts / v_or / v_and; 4 cycles: VOP3P packed ops, v_max3, v_perm).
Writes profiles/r04/fill_mix.json with the library's source hash; bench.py prices the VALU issue peak with it.

    python profiles/microbench/fill_mix.py"""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from defuse_amd import build

FOUR = ("v_pk_", "v_max3", "v_min3", "v_perm", "v_mad", "v_mul", "v_med3", "v_lshl_add", "v_add3", "v_and_or", "v_or3", "v_bfe", "v_alignbit")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([build.HIPCC] + build.LIB_FLAGS[:-2] + build.effective_dsa_flags() + ["-c", "--save-temps", "-o", os.path.join(tmp, "dsa.o"),
                               os.path.join(build.CSRC, "dsa_api.hip")], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        src = open(os.path.join(tmp, "dsa_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    start = next(i for i, l in enumerate(src) if l.startswith("_ZN3dsa11k_fill_fastILi0ELb0EE"))
    end = next(i for i in range(start, len(src)) if src[i].strip().startswith(".Lfunc_end"))
    blocks, cur = [], []
    for l in src[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append(cur)
            cur = []
        else:
            t = l.strip()
            if t and not t.startswith(";") and not t.startswith("."):
                cur.append(t.split()[0])
    blocks.append(cur)
    hot = sorted(blocks, key=lambda b: -sum(1 for x in b if x.startswith("v_pk_maximum3")))[:3]
    c = collections.Counter(x for b in hot for x in b)
    valu = {k: v for k, v in c.items() if k.startswith("v_")}
    four = sum(v for k, v in valu.items() if k.startswith(FOUR))
    total = sum(valu.values())
    out = {"source_hash": build.source_hash(), "kernel": "k_fill_fast<0>", "scope": "static, the three row-sweep blocks (one row of 64 columns each)",
           "valu_per_row": total / 3.0, "valu_per_column_step": total / 3.0 / 64.0,
           "mix": {"two_cycle": (total - four) / total, "four_cycle": four / total},
           "priced_issue_cycles_per_column_step": (2.0 * (total - four) + 4.0 * four) / 3.0 / 64.0,
           "opcodes": dict(sorted(valu.items(), key=lambda kv: -kv[1]))}
    path = os.path.join(ROOT, "profiles", "r04", "fill_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("valu_per_column_step", "mix", "priced_issue_cycles_per_column_step", "source_hash")}))


if __name__ == "__main__":
    main()
