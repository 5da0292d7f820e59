"""The bin-pair builder of clustermatepairs (defuse_amd/csrc/cmp_api.hip, include/defuse_cmp.h).

Without a GPU: the per-fragment logic of the kernels — concordance, the reference's stop conditions, the entries a fragment
adds and the order it writes them in — is __host__ __device__ code; a host program compiled from the same source walks the
fragments in file order, sorts stably by key as the device does, and must give the oracle's map of bin pairs, list by list
(fragments with many alignments per end, both ends in the same bins, alignments over several bins, concordant pairs).
With a GPU (-m gpu): the C ABI itself against the oracle on the same inputs, and the tool with device binning against the
tool with the host cross-check path, dump for dump."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = np.dtype([("fragment", "<i4"), ("start", "<i4"), ("end", "<i4"), ("meta", "<u4")])
PACKED = np.dtype([("fragment", "<i4"), ("read_end", "<i4"), ("rel_start", "<u2"), ("rel_end", "<u2")])


def random_fragments(seed, n_fragments, heavy_every=17):
    """Fragments as lists of alignment dicts (oracle form) and as cmp_record rows + fragment starts."""
    rng = np.random.default_rng(seed)
    frs = []
    for f in range(n_fragments):
        als = []
        heavy = f % heavy_every == 0
        n0 = int(rng.integers(1, 40 if heavy else 3))
        n1 = int(rng.integers(0 if f % 29 == 0 else 1, 40 if heavy else 3))
        # a few loci shared by all fragments: lists with entries of many fragments.  An end-0 alignment lies 6-12 kb below a
        # locus and an end-1 alignment 6-12 kb above one — often in one 32 kb bin (the symmetric combinations, when both ends have
        # alignments at both loci) and never close enough to make the fragment concordant; some alignments lie anywhere.
        loci = [(k % 5, 50000 + 90000 * k) for k in range(12)]
        anchor = [loci[int(rng.integers(0, len(loci)))] for _ in range(2)]
        for end, n in ((0, n0), (1, n1)):
            for _ in range(n):
                ref, pos = anchor[int(rng.integers(0, 2))]
                near = pos + (1 if end else -1) * int(rng.integers(6000, 12000))
                start = max(1, near if rng.random() < 0.85 else int(rng.integers(1, 600000)))
                if rng.random() < 0.15:
                    start = (start // 32768) * 32768 + int(rng.integers(-700, 700))      # near a bin edge: two or three bins
                    start = max(1, start)
                length = int(rng.integers(30, 151))
                als.append(dict(frag=1000 + f, readEnd=end, ref=ref, strand=int(rng.integers(0, 2)), region=(start, start + length - 1)))
        order = rng.permutation(len(als))                       # the ends interleaved, as an aligner's output may be
        frs.append([als[k] for k in order])
    return frs


def to_records(frs):
    rows, starts = [], [0]
    for als in frs:
        for a in als:
            rows.append((a["frag"], a["region"][0], a["region"][1], a["ref"] | (a["strand"] << 28) | (a["readEnd"] << 29)))
        starts.append(len(rows))
    return np.array(rows, dtype=REC), np.array(starts, dtype=np.uint32)


def oracle_bin_pairs(frs, mfr):
    from oracle import clustermatepairs_oracle as ora
    bp = {}
    conc = 0
    for als in frs:
        before = sum(len(v[0]) + len(v[1]) for v in bp.values())
        c = [set(), set()]
        for a in als:
            for b in ora.get_bins(a["region"], mfr, mfr):
                c[a["readEnd"]].add((a["ref"], b))
        conc += 1 if c[0] & c[1] else 0
        ora.add_fragment(als, mfr, bp)
    return bp, conc


def as_lists(keys, off1, off2, p1, p2):
    out = {}
    for k, key in enumerate(keys):
        key = int(key)
        f = [tuple(int(x) for x in r) for r in p1[off1[k]:off1[k + 1]]]
        s = [tuple(int(x) for x in r) for r in p2[off2[k]:off2[k + 1]]]
        out[(key >> 32, key & 0xFFFFFFFF)] = (f, s)
    return out


@pytest.fixture(scope="module")
def host_walk(tmp_path_factory):
    d = tmp_path_factory.mktemp("cmpwalk")
    src = d / "walk.hip"
    src.write_text(r'''
#include "%s/defuse_amd/csrc/cmp_api.hip"
#include <algorithm>
#include <cstdio>
// the kernels' per-fragment code on the host, fragments in file order, then the device's stable sort by key
int main(int argc, char** argv) {
    FILE* in = fopen(argv[1], "rb");
    long long n, nf; int mfr;
    if (fread(&n, 8, 1, in) != 1 || fread(&nf, 8, 1, in) != 1 || fread(&mfr, 4, 1, in) != 1) return 2;
    std::vector<cmp_record> recs(n); std::vector<uint32_t> fs(nf + 1);
    if (n && fread(recs.data(), sizeof(cmp_record), n, in) != (size_t)n) return 2;
    if (fread(fs.data(), 4, nf + 1, in) != (size_t)(nf + 1)) return 2;
    struct E { unsigned long long key; cmp_packed p; };
    std::vector<E> side[2];
    long long conc = 0;
    for (long long f = 0; f < nf; ++f) {
        const cmp_record* r = recs.data() + fs[f]; const int m = (int)(fs[f + 1] - fs[f]);
        if (fragment_concordant(r, m, mfr)) { ++conc; continue; }
        int bad, value; const int kind = fragment_error(r, m, mfr, bad, value);
        if (kind) { printf("error %%d %%lld %%d\n", kind, (long long)fs[f] + bad, value); return 0; }
        // as k_cmp_count + k_cmp_emit place them: `first` lists end 0 then end 1, `second` lists end 1 then end 0
        std::vector<E> part[2][2];
        fragment_entries(r, m, mfr, [&](int s, int ea, unsigned long long key, int ia, int b) {
            cmp_packed p; p.fragment = r[ia].fragment; p.read_end = rec_end(r[ia]);
            p.rel_start = (uint16_t)(r[ia].start - b * BIN_LENGTH + BIN_LENGTH / 2); p.rel_end = (uint16_t)(r[ia].end - b * BIN_LENGTH + BIN_LENGTH / 2);
            part[s][ea].push_back(E{key, p});
        });
        for (const E& e : part[0][0]) side[0].push_back(e);
        for (const E& e : part[0][1]) side[0].push_back(e);
        for (const E& e : part[1][1]) side[1].push_back(e);
        for (const E& e : part[1][0]) side[1].push_back(e);
    }
    printf("concordant %%lld\n", conc);
    for (int s = 0; s < 2; ++s) {
        std::stable_sort(side[s].begin(), side[s].end(), [](const E& a, const E& b) { return a.key < b.key; });
        for (const E& e : side[s]) printf("%%d %%llu %%d %%d %%d %%d\n", s, e.key, e.p.fragment, e.p.read_end, (int)e.p.rel_start, (int)e.p.rel_end);
    }
    return 0;
}
''' % ROOT)
    exe = d / "walk"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-Wno-unused-result", "-o", str(exe), str(src)])
    return str(exe), d


def write_input(path, recs, starts, mfr):
    with open(path, "wb") as f:
        f.write(struct.pack("<qqi", len(recs), len(starts) - 1, mfr))
        f.write(recs.tobytes())
        f.write(starts.tobytes())


@pytest.mark.parametrize("seed,mfr", [(1, 600), (2, 600), (3, 1500), (4, 250)])
def test_fragment_walk_equals_the_oracle(host_walk, seed, mfr):
    exe, d = host_walk
    frs = random_fragments(seed, 400)
    recs, starts = to_records(frs)
    write_input(d / "in.bin", recs, starts, mfr)
    out = subprocess.run([exe, str(d / "in.bin")], capture_output=True, text=True, check=True).stdout.splitlines()
    exp, conc = oracle_bin_pairs(frs, mfr)
    assert out[0] == "concordant %d" % conc and conc > 5
    got = {}
    for line in out[1:]:
        s, key, fr, e, rs, re_ = (int(x) for x in line.split())
        got.setdefault((key >> 32, key & 0xFFFFFFFF), ([], []))[s].append((fr, e, rs, re_))
    assert set(got) == set(exp) and len(exp) > 300
    multi = 0
    for key in exp:
        assert got[key][0] == exp[key][0] and got[key][1] == exp[key][1], key
        multi += 1 if len(exp[key][0]) > 3 else 0
    assert multi > 20                                             # lists with several entries: their order is what is being tested


def test_stop_conditions_in_file_order(host_walk):
    """The first alignment on which the reference stops — relative position outside 16 bits cannot happen with extend <
    half a bin, so: too many reference sequences, chromosome too large — is reported with its kind and value."""
    exe, d = host_walk
    frs = random_fragments(9, 50)
    frs[20][0]["ref"] = (1 << 18) + 5
    frs[30][0]["region"] = ((1 << 13) * 32768 + 10, (1 << 13) * 32768 + 80)
    recs, starts = to_records(frs)
    write_input(d / "err.bin", recs, starts, 600)
    out = subprocess.run([exe, str(d / "err.bin")], capture_output=True, text=True, check=True).stdout.splitlines()
    from oracle import clustermatepairs_oracle as ora
    # fragment 20 may be concordant (then nothing stops there)
    first = None
    for k, als in enumerate(frs):
        try:
            ora.add_fragment(als, 600, {})
        except SystemExit as e:
            first = (k, str(e))
            break
    assert first is not None
    kind = 2 if "too many reference" in first[1] else 3
    assert out[0].startswith("error %d " % kind), (out[0], first)
    assert int(out[0].split()[2]) in range(int(starts[first[0]]), int(starts[first[0] + 1]))
