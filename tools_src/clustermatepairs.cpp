// clustermatepairs — drop-in replacement of the reference tool (tools/clustermatepairs.cpp:389-589):
// same command line (-a may be "-" for stdin), same compact alignment input, same cluster output
// lines and progress lines.  The input is parsed by host threads; the concordance filter and the 32 kb bin-pair
// bucketing of all fragments (tools/clustermatepairs.cpp:146-290) run on the GPU through include/defuse_cmp.h (one
// thread per fragment, a stable radix sort by bin-pair key in place of the reference's hash map); the per-bin-pair
// filters and the output (:292-375,478-584) are host code; MatePairEM::DoClustering for all bin pairs runs on the
// GPU in batches through include/defuse_mpe.h.  No CPU fallback: DEFUSE_CMP_HOST_BINNING=1 selects the host
// transcription of the bucketing as a cross-check (tests compare the two), it is never taken by itself.
// Iteration orders the reference leaves to boost::unordered_map are the canonical ascending-key
// orders of SURVEY.md 8(c).
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <numeric>

#include "../include/defuse_cmp.h"
#include "../include/defuse_dsa.h"
#include "../include/defuse_mpe.h"
#define DEFUSE_HUGE_NEW          // large blocks from huge-page mappings (defuse_host.hpp)
#include "defuse_host.hpp"

using namespace defuse;

namespace {

struct CompactAlignment { int fragmentIndex, readEnd, referenceIndex, strand; Region region; };
struct AlignmentPacked { int fragmentIndex, readEnd; unsigned short relativeStart, relativeEnd; };

const int binLength = 1 << 15;

// Binning::GetBins (tools/clustermatepairs.cpp:152-162) appears inline below as
//   startBin = (region.start - extend) / length, endBin = (region.end + extend) / length      (C++ int division)
unsigned pack_ref_bin(int ref, int strand, int bin)   // RefBinPacked (:28-65); the caller has checked ref < 2^18 and bin < 2^13
{
    return (unsigned)ref | ((unsigned)strand << 18) | (((unsigned)bin & 0x1FFFu) << 19);
}

// r8_normal_01_cdf_inverse (AS 241, tools/asa241.C:424-563)
double poly(const double* a, double x) { double v = 0.0; for (int i = 7; i >= 0; --i) v = v * x + a[i]; return v; }
double normal_01_cdf_inverse(double p)
{
    static const double a[8] = {3.3871328727963666080, 1.3314166789178437745e+2, 1.9715909503065514427e+3, 1.3731693765509461125e+4,
                                4.5921953931549871457e+4, 6.7265770927008700853e+4, 3.3430575583588128105e+4, 2.5090809287301226727e+3};
    static const double b[8] = {1.0, 4.2313330701600911252e+1, 6.8718700749205790830e+2, 5.3941960214247511077e+3,
                                2.1213794301586595867e+4, 3.9307895800092710610e+4, 2.8729085735721942674e+4, 5.2264952788528545610e+3};
    static const double c[8] = {1.42343711074968357734, 4.63033784615654529590, 5.76949722146069140550, 3.64784832476320460504,
                                1.27045825245236838258, 2.41780725177450611770e-1, 2.27238449892691845833e-2, 7.74545014278341407640e-4};
    static const double d[8] = {1.0, 2.05319162663775882187, 1.67638483018380384940, 6.89767334985100004550e-1,
                                1.48103976427480074590e-1, 1.51986665636164571966e-2, 5.47593808499534494600e-4, 1.05075007164441684324e-9};
    static const double e[8] = {6.65790464350110377720, 5.46378491116411436990, 1.78482653991729133580, 2.96560571828504891230e-1,
                                2.65321895265761230930e-2, 1.24266094738807843860e-3, 2.71155556874348757815e-5, 2.01033439929228813265e-7};
    static const double f[8] = {1.0, 5.99832206555887937690e-1, 1.36929880922735805310e-1, 1.48753612908506148525e-2,
                                7.86869131145613259100e-4, 1.84631831751005468180e-5, 1.42151175831644588870e-7, 2.04426310338993978564e-15};
    if (p <= 0.0) return -1.0e30;
    if (1.0 <= p) return 1.0e30;
    const double q = p - 0.5;
    if (std::fabs(q) <= 0.425) {
        const double r = 0.180625 - q * q;
        return q * poly(a, r) / poly(b, r);
    }
    double r = q < 0.0 ? p : 1.0 - p;
    if (r <= 0.0) std::exit(1);
    r = std::sqrt(-std::log(r));
    double value;
    if (r <= 5.0) { r = r - 1.6; value = poly(c, r) / poly(d, r); }
    else { r = r - 5.0; value = poly(e, r) / poly(f, r); }
    return q < 0.0 ? -value : value;
}
double normalpdf(double x, double mu, double sigma)   // tools/Common.cpp:61-69
{
    const double coeff = 1.0 / (sigma * std::sqrt(2 * M_PI));
    const double dist = (x - mu) / sigma;
    return coeff * std::exp(-0.5 * dist * dist);
}

Region StrandRemap(const Region& r, int strand)   // tools/MatePairEM.cpp:75-83
{
    Region out;
    out.start = strand == PlusStrand ? r.start : -r.end;
    out.end = strand == PlusStrand ? r.end : -r.start;
    return out;
}

static_assert(sizeof(AlignmentPacked) == sizeof(cmp_packed) && sizeof(AlignmentPacked) == 12, "AlignmentPacked is cmp_packed");

// the two lists of a bin pair, wherever they live (the device's sorted arrays, or the host cross-check's vectors)
struct ListView {
    const AlignmentPacked* p = nullptr;
    size_t n = 0;
    size_t size() const { return n; }
    const AlignmentPacked& operator[](size_t k) const { return p[k]; }
};
struct PairView { ListView first, second; };

// one surviving bin pair, ready for clustering and for writing its clusters afterwards
struct Problem {
    std::vector<CompactAlignment> alignments1, alignments2;
    std::vector<std::pair<int, int>> alignPairs;
};

// the alignments of one side of a bin pair grouped by fragment: `idx` holds alignment indices, fragments ascending
// (canonical order), indices ascending inside a fragment; group g owns idx[off[g] .. off[g+1])
struct FragmentGroups {
    std::vector<int> frag, off, idx;
    size_t size() const { return frag.size(); }
};

void GroupByFragment(const std::vector<CompactAlignment>& alignments, FragmentGroups& g)   // GetFragmentAlignments :292-300
{
    g.frag.clear(); g.off.clear(); g.idx.resize(alignments.size());
    std::iota(g.idx.begin(), g.idx.end(), 0);
    std::stable_sort(g.idx.begin(), g.idx.end(), [&](int a, int b) { return alignments[a].fragmentIndex < alignments[b].fragmentIndex; });
    for (size_t k = 0; k < g.idx.size(); ++k)
        if (k == 0 || alignments[g.idx[k]].fragmentIndex != alignments[g.idx[k - 1]].fragmentIndex) {
            g.frag.push_back(alignments[g.idx[k]].fragmentIndex);
            g.off.push_back((int)k);
        }
    g.off.push_back((int)g.idx.size());
}

// FilterUnmatched (:302-314) then FilterOverlapping (:316-358) on one side: fragments absent from the other side go,
// and per fragment an alignment is kept only if none of its minFusionRange bins was taken by an earlier one of the
// same read end and (reference, strand)
void FilterSide(FragmentGroups& g, const FragmentGroups& other, const std::vector<CompactAlignment>& alignments, int minFusionRange)
{
    FragmentGroups out;
    out.idx.reserve(g.idx.size());
    std::vector<std::pair<unsigned, int>> taken[2];
    size_t o = 0;
    for (size_t f = 0; f < g.size(); ++f) {
        while (o < other.size() && other.frag[o] < g.frag[f]) ++o;
        if (o == other.size() || other.frag[o] != g.frag[f]) continue;
        out.frag.push_back(g.frag[f]);
        out.off.push_back((int)out.idx.size());
        taken[0].clear(); taken[1].clear();
        for (int k = g.off[f]; k < g.off[f + 1]; ++k) {
            const CompactAlignment& a = alignments[g.idx[k]];
            const int startBin = a.region.start / minFusionRange, endBin = a.region.end / minFusionRange;   // GetBins, extend 0
            const unsigned refStrandId = (unsigned)a.referenceIndex | ((unsigned)a.strand << 31);
            std::vector<std::pair<unsigned, int>>& t = taken[a.readEnd];
            bool overlapping = false;
            for (int b = startBin; b <= endBin && !overlapping; ++b)
                overlapping = std::find(t.begin(), t.end(), std::make_pair(refStrandId, b)) != t.end();
            if (!overlapping) {
                for (int b = startBin; b <= endBin; ++b) t.push_back(std::make_pair(refStrandId, b));
                out.idx.push_back(g.idx[k]);
            }
        }
    }
    out.off.push_back((int)out.idx.size());
    g = std::move(out);
}

}  // namespace

int main(int argc, char* argv[])
{
    keep_freed_memory();
    CmdLine cmd("Mate Pair Clustering Tool");
    cmd.add("a", "align", "Alignments Filename", "string");
    cmd.add("c", "clusters", "Output Clusters Filename", "string");
    cmd.add("u", "fragmentmean", "Fragment Length Mean", "integer", "float");       // a double the reference labels "integer" (:403-404)
    cmd.add("s", "fragmentstddev", "Fragment Length Standard Deviation", "integer", "float");
    cmd.add("p", "precision", "Precision", "double", "float");
    cmd.add("m", "minclustersize", "Minimum Cluster Size", "integer");
    cmd.parse(argc, argv);
    const double fragmentMean = cmd.real("fragmentmean"), fragmentStdDev = cmd.real("fragmentstddev"), precision = cmd.real("precision");
    const int minClusterSize = cmd.integer("minclustersize");
    const int minFusionRange = (int)(fragmentMean + 10 * fragmentStdDev);

    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_stage = now();
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[clustermatepairs] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t;
    };
    // The HIP runtime takes 0.15-0.25 s to come up: the bin-pair builder — this process's first GPU call — is created
    // on a thread of its own while the input is read and parsed.  An error exit while that thread is inside the runtime must not run
    // exit handlers beside it: die() flushes and leaves at once, as the success path does.
    die_hook() = [] {
        std::cout.flush();
        std::cerr.flush();
        fflush(nullptr);
        _exit(1);
    };
    const bool host_binning = [] { const char* e = std::getenv("DEFUSE_CMP_HOST_BINNING"); return e && std::atoi(e) != 0; }();
    int gpu_device = -1;                                                      // the device of this process's first GPU call
    cmp_binner* binner = nullptr;
    int binner_rc = DSA_OK;
    std::string binner_error;
    std::thread binner_thread;
    if (!host_binning)
        binner_thread = std::thread([&] {
            gpu_device = dsa_pick_device();
            binner_rc = cmp_bin_create(&binner, gpu_device);
            if (binner_rc != DSA_OK) binner_error = cmp_last_error();
        });
    struct ThreadJoiner {                    // (an exception that unwinds main must not meet a joinable thread)
        std::thread& th;
        ~ThreadJoiner() { if (th.joinable()) th.join(); }
    } binner_joiner{binner_thread};
    std::cout << "Finding pairs of reference sequences connected by pairs of alignments" << std::endl;
    // The input is taken into memory whole and cut into one piece per host thread at fragment boundaries.  Every piece is
    // parsed, then (with reference indices in order of first appearance over the whole file, as a serial reader gives
    // them) binned into bin pairs of its own; the pieces' bin pairs are joined in file order, so every list holds its
    // alignments in the order a single reader would have appended them.
    MappedText text;
    text.load(cmd.str("align"), "Error: Unable to open alignment file ");
    stage("  input in memory");
    unsigned nThreads = host_threads();
    if (text.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nThreads = 1;      // small inputs: threads only on request (tests)

    auto line_end = [&](size_t pos) {                       // one past the line's newline (or the end of the text)
        const char* nl = (const char*)memchr(text.data() + pos, '\n', text.size() - pos);
        return nl ? (size_t)(nl - text.data()) + 1 : text.size();
    };
    auto name_of = [&](size_t pos, size_t& len) {           // first field of the line at pos
        const size_t e = line_end(pos);
        size_t stop = (e > pos && text[e - 1] == '\n') ? e - 1 : e;
        const char* tab = (const char*)memchr(text.data() + pos, '\t', stop - pos);
        len = tab ? (size_t)(tab - (text.data() + pos)) : stop - pos;
        return text.data() + pos;
    };
    std::vector<size_t> cut(nThreads + 1, text.size());
    cut[0] = 0;
    for (unsigned t = 1; t < nThreads; ++t) {
        size_t pos = std::max(cut[t - 1], text.size() / nThreads * t);
        if (pos > 0 && pos < text.size() && text[pos - 1] != '\n') pos = line_end(pos);
        while (pos > 0 && pos < text.size()) {                // not inside a fragment: move on while the name repeats
            size_t prev = pos - 1;
            while (prev > 0 && text[prev - 1] != '\n') --prev;
            size_t l0, l1;
            const char* n0 = name_of(prev, l0);
            const char* n1 = name_of(pos, l1);
            if (l0 != l1 || memcmp(n0, n1, l0) != 0) break;
            pos = line_end(pos);
        }
        cut[t] = pos;
    }

    struct Piece {
        std::vector<cmp_record> recs;                        // reference index (in meta): piece-local until remapped
        std::vector<uint32_t> fragStart;                     // first record of every fragment, plus the end
        std::vector<std::string> refNames;
        std::vector<int> remap;                              // piece-local reference index -> index over the whole file
        std::unordered_map<std::string, int> refIndex;
        size_t lines = 0;
        size_t errorLine = 0;                                // 1-based line inside the piece, 0 = none
        std::string error;
        std::vector<std::string> errorStdout;
        FlatMap64 binPairIndex{1 << 16};
        std::vector<uint64_t> binPairKey;
        std::vector<std::pair<std::vector<AlignmentPacked>, std::vector<AlignmentPacked>>> binPairStore;
    };
    std::vector<Piece> pieces(nThreads);
    auto run_threads = [&](const std::function<void(unsigned)>& fn) { defuse::run_threads(nThreads, fn); };

    // CompactAlignmentStream + FragmentAlignmentStream (tools/AlignmentStream.cpp:156-221), one piece
    auto parse_piece = [&](unsigned t) {
        Piece& pc = pieces[t];
        size_t pos = cut[t];
        const size_t stop = cut[t + 1];
        pc.recs.reserve((stop - pos) / 28 + 16);              // a line is at least ~30 bytes; no regrowth while parsing
        pc.fragStart.reserve((stop - pos) / 56 + 16);
        const char* curName = nullptr;
        size_t curLen = 0;
        std::string refKey;
        while (pos < stop) {
            const size_t e = line_end(pos);
            const char* line = text.data() + pos;
            const size_t len = (e > pos && text[e - 1] == '\n') ? e - 1 - pos : e - pos;
            pos = e;
            ++pc.lines;
            auto fail = [&](const std::string& msg) { pc.errorLine = pc.lines; pc.error = msg; };
            if (len == 0) { fail("Error: Empty alignment line "); return; }
            const char* fs[7];
            int nf = 0;
            fs[nf++] = line;
            for (const char* p = line; nf < 7;) {
                const char* tab = (const char*)memchr(p, '\t', (size_t)(line + len - p));
                if (!tab) break;
                fs[nf++] = p = tab + 1;
            }
            if (nf < 6) { fail("Error: Format error for alignment line "); return; }
            if (nf < 7) fs[6] = line + len + 1;                      // field k is [fs[k], fs[k+1] - 1)
            auto flen = [&](int k) { return (size_t)(fs[k + 1] - 1 - fs[k]); };
            if (!curName || flen(0) != curLen || memcmp(fs[0], curName, curLen) != 0) {
                pc.fragStart.push_back((uint32_t)pc.recs.size());
                curName = fs[0];
                curLen = flen(0);
            }
            cmp_record a;
            if (!field_int(fs[0], flen(0), a.fragment)) { fail("Error: bad integer '" + std::string(fs[0], flen(0)) + "' as fragment name on line "); return; }
            const int readEnd = (flen(1) == 1 && fs[1][0] == '1') ? 0 : 1;
            refKey.assign(fs[2], flen(2));
            auto ri = pc.refIndex.find(refKey);
            if (ri == pc.refIndex.end()) {
                ri = pc.refIndex.emplace(refKey, (int)pc.refNames.size()).first;
                pc.refNames.push_back(refKey);
            }
            const int strand = (flen(3) == 1 && fs[3][0] == '-') ? MinusStrand : PlusStrand;
            a.meta = CMP_META(ri->second, strand, readEnd);
            if (!field_int(fs[4], flen(4), a.start) || !field_int(fs[5], flen(5), a.end)) {
                fail("Error: bad integer '" + std::string(fs[4], (size_t)(line + len - fs[4])) + "' on line ");
                return;
            }
            pc.recs.push_back(a);
        }
    };
    run_threads(parse_piece);
    if (timing) { std::cerr << "[clustermatepairs]   lines per piece:"; for (const Piece& pc : pieces) std::cerr << " " << pc.lines; std::cerr << std::endl; }
    stage("  parsed");
    {
        size_t lineBase = 0;
        for (const Piece& pc : pieces) {                      // the first bad line of the file, as a serial reader meets it
            if (pc.errorLine) die(pc.error + std::to_string(lineBase + pc.errorLine));
            lineBase += pc.lines;
        }
    }
    text.release();

    std::vector<std::string> refNames;
    {
        std::unordered_map<std::string, int> refIndex;
        for (Piece& pc : pieces) {
            std::vector<int> remap(pc.refNames.size());
            for (size_t k = 0; k < pc.refNames.size(); ++k) {
                auto ri = refIndex.find(pc.refNames[k]);
                if (ri == refIndex.end()) {
                    ri = refIndex.emplace(pc.refNames[k], (int)refNames.size()).first;
                    refNames.push_back(pc.refNames[k]);
                }
                remap[k] = ri->second;
            }
            pc.remap.swap(remap);
            pc.fragStart.push_back((uint32_t)pc.recs.size());
        }
    }
    if (refNames.size() >= ((size_t)1 << 28)) die("Error: more than 2^28 reference sequences");
    run_threads([&](unsigned t) {
        Piece& pc = pieces[t];
        for (cmp_record& a : pc.recs) a.meta = (a.meta & ~0x0FFFFFFFu) | (uint32_t)pc.remap[a.meta & 0x0FFFFFFFu];
    });

    typedef std::pair<std::vector<AlignmentPacked>, std::vector<AlignmentPacked>> PackedPair;
    // The bin pairs: key (first.id << 32 | second.id) and the two lists of each.  By default they are built on the GPU
    // (include/defuse_cmp.h); DEFUSE_CMP_HOST_BINNING=1 runs the host transcription below instead — a cross-check the tests
    // hold against the device path, never a fallback.
    std::vector<uint64_t> binPairKey;
    std::vector<PairView> binPairStore;
    struct Joined { std::vector<uint64_t> key; std::vector<PackedPair> store; };
    std::vector<Joined> joined(host_binning ? nThreads : 0);                  // owner of the lists (host path)
    std::unique_ptr<AlignmentPacked[]> devFirst, devSecond;                   // owner of the lists (device path)
    auto first_device = [&] {
        if (binner_thread.joinable()) binner_thread.join();
        if (gpu_device < 0) gpu_device = dsa_pick_device();
        return gpu_device;
    };
    if (host_binning) {
        auto bin_piece = [&](unsigned t) {
            Piece& pc = pieces[t];
            auto bin_pair = [&](unsigned lo, unsigned hi) -> PackedPair& {
                const uint64_t key = ((uint64_t)lo << 32) | hi;
                bool added;
                uint32_t& slot = pc.binPairIndex.find_or_add(key, added);
                if (added) {
                    slot = (uint32_t)pc.binPairStore.size();
                    pc.binPairKey.push_back(key);
                    pc.binPairStore.emplace_back();
                }
                return pc.binPairStore[slot];
            };
            std::vector<std::pair<int, int>> conc[2];
            std::vector<std::pair<unsigned, AlignmentPacked>> binned[2];
            std::vector<CompactAlignment> recs(pc.recs.size());
            for (size_t k = 0; k < recs.size(); ++k) {
                const cmp_record& r = pc.recs[k];
                recs[k] = CompactAlignment{r.fragment, (int)((r.meta >> 29) & 1u), (int)(r.meta & 0x0FFFFFFFu), (int)((r.meta >> 28) & 1u), Region{r.start, r.end}};
            }
            for (size_t fr = 0; fr + 1 < pc.fragStart.size(); ++fr) {
                const CompactAlignment* first = recs.data() + pc.fragStart[fr];
                const CompactAlignment* last = recs.data() + pc.fragStart[fr + 1];
                // CheckConcordant (:211-244): a (reference, bin) shared by the two ends
                conc[0].clear(); conc[1].clear();
                for (const CompactAlignment* a = first; a != last; ++a) {
                    const int startBin = (a->region.start - minFusionRange) / minFusionRange, endBin = (a->region.end + minFusionRange) / minFusionRange;
                    for (int b = startBin; b <= endBin; ++b) conc[a->readEnd].push_back(std::make_pair(a->referenceIndex, b));
                }
                bool concordant = false;
                for (const auto& rb : conc[0])
                    if (std::find(conc[1].begin(), conc[1].end(), rb) != conc[1].end()) { concordant = true; break; }
                if (concordant) continue;
                // AddBinPairs (:246-290): per read end the packed alignments by bin id ascending, arrival order inside a bin
                binned[0].clear(); binned[1].clear();
                for (const CompactAlignment* a = first; a != last; ++a) {
                    const int startBin = (a->region.start - minFusionRange) / binLength, endBin = (a->region.end + minFusionRange) / binLength;
                    for (int b = startBin; b <= endBin; ++b) {
                        const int rs = a->region.start - b * binLength + binLength / 2, re = a->region.end - b * binLength + binLength / 2;
                        if (rs < 0 || re < 0 || rs >= (1 << 16) || re >= (1 << 16)) {
                            pc.errorLine = 1;
                            pc.error = "Error: relativeStart >= 0 failed (alignment does not fit its bin)";
                            return;
                        }
                        if (a->referenceIndex >= (1 << 18) || b >= (1 << 13)) {        // RefBinPacked (:28-65)
                            const bool refs = a->referenceIndex >= (1 << 18);
                            pc.errorLine = 1;
                            pc.errorStdout = {std::to_string(refs ? a->referenceIndex : b), std::to_string(refs ? (1 << 18) : (1 << 13))};
                            pc.error = refs ? "Packing failed, too many reference sequences" : "Packing failed, chromosome too large";
                            return;
                        }
                        binned[a->readEnd].push_back(std::make_pair(pack_ref_bin(a->referenceIndex, a->strand, b),
                                                                    AlignmentPacked{a->fragmentIndex, a->readEnd, (unsigned short)rs, (unsigned short)re}));
                    }
                }
                for (int e = 0; e < 2; ++e)
                    std::stable_sort(binned[e].begin(), binned[e].end(), [](const std::pair<unsigned, AlignmentPacked>& x,
                                                                            const std::pair<unsigned, AlignmentPacked>& y) { return x.first < y.first; });
                for (size_t i0 = 0; i0 < binned[0].size();) {
                    size_t i1 = i0;
                    while (i1 < binned[0].size() && binned[0][i1].first == binned[0][i0].first) ++i1;
                    for (size_t j0 = 0; j0 < binned[1].size();) {
                        size_t j1 = j0;
                        while (j1 < binned[1].size() && binned[1][j1].first == binned[1][j0].first) ++j1;
                        const unsigned id1 = binned[0][i0].first, id2 = binned[1][j0].first;
                        const bool fwd = id1 < id2;
                        PackedPair& e = fwd ? bin_pair(id1, id2) : bin_pair(id2, id1);
                        std::vector<AlignmentPacked>& d1 = fwd ? e.first : e.second;
                        std::vector<AlignmentPacked>& d2 = fwd ? e.second : e.first;
                        for (size_t k = i0; k < i1; ++k) d1.push_back(binned[0][k].second);
                        for (size_t k = j0; k < j1; ++k) d2.push_back(binned[1][k].second);
                        j0 = j1;
                    }
                    i0 = i1;
                }
            }
            pc.recs.clear();
            pc.recs.shrink_to_fit();
            pc.binPairIndex.release();
        };
        run_threads(bin_piece);
        stage("  binned");
        for (const Piece& pc : pieces)
            if (pc.errorLine) {
                for (const std::string& l : pc.errorStdout) std::cout << l << std::endl;
                die(pc.error);
            }

        // bin pairs by (first.id, second.id), the pieces joined in file order; visited in ascending key order afterwards (canonical)
        // Thread t joins the keys whose hash falls to it, walking the pieces in file order; the joined lists stay where they are
        // (one table of pointers over all threads' shares instead of a copy), and every thread frees its own piece at the end.
        {
            auto join_keys = [&](unsigned t) {
                Joined& j = joined[t];
                if (nThreads == 1) {
                    j.key.swap(pieces[0].binPairKey);
                    j.store.swap(pieces[0].binPairStore);
                    return;
                }
                size_t mine = 0;
                for (const Piece& pc : pieces)
                    for (uint64_t key : pc.binPairKey) mine += ((key * 0x9E3779B97F4A7C15ULL >> 40) % nThreads == t) ? 1 : 0;
                j.key.reserve(mine);                              // upper bound: no regrowth, the lists are never moved twice
                j.store.reserve(mine);
                FlatMap64 index(1 << 16);
                for (Piece& pc : pieces)
                    for (size_t k = 0; k < pc.binPairKey.size(); ++k) {
                        const uint64_t key = pc.binPairKey[k];
                        if ((key * 0x9E3779B97F4A7C15ULL >> 40) % nThreads != t) continue;
                        bool added;
                        uint32_t& slot = index.find_or_add(key, added);
                        if (added) {
                            slot = (uint32_t)j.store.size();
                            j.key.push_back(key);
                            j.store.push_back(std::move(pc.binPairStore[k]));
                        } else {
                            PackedPair& d = j.store[slot];
                            d.first.insert(d.first.end(), pc.binPairStore[k].first.begin(), pc.binPairStore[k].first.end());
                            d.second.insert(d.second.end(), pc.binPairStore[k].second.begin(), pc.binPairStore[k].second.end());
                        }
                    }
            };
            run_threads(join_keys);
            run_threads([&](unsigned t) { pieces[t] = Piece(); });          // millions of small lists: freed side by side
        }
        size_t total = 0;
        for (const Joined& j : joined) total += j.key.size();
        binPairKey.reserve(total);
        binPairStore.reserve(total);
        for (Joined& j : joined) {
            binPairKey.insert(binPairKey.end(), j.key.begin(), j.key.end());
            for (PackedPair& pp : j.store)
                binPairStore.push_back(PairView{ListView{pp.first.data(), pp.first.size()}, ListView{pp.second.data(), pp.second.size()}});
        }
    } else {
        // records and fragment starts go to the device piece by piece (every piece from its own thread), the device does the
        // rest: CheckConcordant, AddBinPairs, the map of bin pairs as a stable sort by key (defuse_amd/csrc/cmp_api.hip)
        std::vector<size_t> recBase(nThreads + 1, 0), fragBase(nThreads + 1, 0);
        for (unsigned t = 0; t < nThreads; ++t) {
            recBase[t + 1] = recBase[t] + pieces[t].recs.size();
            fragBase[t + 1] = fragBase[t] + pieces[t].fragStart.size() - 1;
        }
        if (binner_thread.joinable()) binner_thread.join();
        if (binner_rc != DSA_OK || !binner) die(std::string("Error: no usable MI355X/HIP device (") + binner_error + ")");
        if (cmp_bin_reserve(binner, (int64_t)recBase[nThreads], (int64_t)fragBase[nThreads]) != DSA_OK) die(std::string("Error: ") + cmp_last_error());
        std::vector<int> rcs(nThreads, DSA_OK);
        run_threads([&](unsigned t) {
            Piece& pc = pieces[t];
            rcs[t] = cmp_bin_upload_records(binner, pc.recs.data(), (int64_t)pc.recs.size(), (int64_t)recBase[t]);
            if (rcs[t] != DSA_OK) return;
            std::vector<uint32_t> fs(pc.fragStart.size() - (t + 1 < nThreads ? 1 : 0));      // the last piece brings the end
            for (size_t k = 0; k < fs.size(); ++k) fs[k] = (uint32_t)(pc.fragStart[k] + recBase[t]);
            rcs[t] = cmp_bin_upload_fragments(binner, fs.data(), (int64_t)fs.size(), (int64_t)fragBase[t]);
        });
        for (int rc : rcs)
            if (rc != DSA_OK) die(std::string("Error: bin pairs on the GPU failed: ") + cmp_last_error());
        stage("  records on the device");
        cmp_stats st{};
        if (cmp_bin_run(binner, minFusionRange, &st) != DSA_OK) die(std::string("Error: bin pairs on the GPU failed: ") + cmp_last_error());
        if (st.err_record >= 0) {
            // the alignment on which the reference stops (:178-192 DebugChecks, :28-65 packing limits): its message and exit
            unsigned t = 0;
            while (t + 1 < nThreads && (size_t)st.err_record >= recBase[t + 1]) ++t;
            const cmp_record& a = pieces[t].recs[(size_t)st.err_record - recBase[t]];
            if (st.err_kind == 1) die("Error: relativeStart >= 0 failed (alignment does not fit its bin)");
            const bool refs = st.err_kind == 2;
            const int startBin = (a.start - minFusionRange) / binLength;
            std::cout << (refs ? (int)(a.meta & 0x0FFFFFFFu) : std::max(startBin, 1 << 13)) << std::endl << (refs ? (1 << 18) : (1 << 13)) << std::endl;
            die(refs ? "Packing failed, too many reference sequences" : "Packing failed, chromosome too large");
        }
        std::vector<int64_t> off1((size_t)st.n_keys + 1), off2((size_t)st.n_keys + 1);
        binPairKey.resize((size_t)st.n_keys);
        devFirst.reset(new AlignmentPacked[(size_t)st.n_first + 1]);              // (not value-initialised: every thread touches its own share first)
        devSecond.reset(new AlignmentPacked[(size_t)st.n_second + 1]);
        if (cmp_bin_fetch(binner, binPairKey.data(), off1.data(), off2.data(), nullptr, nullptr) != DSA_OK)
            die(std::string("Error: bin pairs on the GPU failed: ") + cmp_last_error());
        run_threads([&](unsigned t) {
            const int64_t a0 = st.n_first * t / nThreads, a1 = st.n_first * (t + 1) / nThreads;
            const int64_t b0 = st.n_second * t / nThreads, b1 = st.n_second * (t + 1) / nThreads;
            rcs[t] = cmp_bin_fetch_part(binner, 0, a0, a1 - a0, (cmp_packed*)devFirst.get() + a0);
            if (rcs[t] == DSA_OK) rcs[t] = cmp_bin_fetch_part(binner, 1, b0, b1 - b0, (cmp_packed*)devSecond.get() + b0);
        });
        for (int rc : rcs)
            if (rc != DSA_OK) die(std::string("Error: bin pairs on the GPU failed: ") + cmp_last_error());
        cmp_bin_destroy(binner);
        binPairStore.resize((size_t)st.n_keys);
        for (size_t k = 0; k < (size_t)st.n_keys; ++k)
            binPairStore[k] = PairView{ListView{devFirst.get() + off1[k], (size_t)(off1[k + 1] - off1[k])},
                                       ListView{devSecond.get() + off2[k], (size_t)(off2[k + 1] - off2[k])}};
        if (timing)
            std::cerr << "[clustermatepairs]   bin pairs on the device: " << st.n_fragments << " fragments, " << st.n_concordant << " concordant, " << st.n_keys
                      << " bin pairs, " << st.n_first + st.n_second << " entries, kernels + sorts " << st.device_ms << " ms" << std::endl;
        run_threads([&](unsigned t) { pieces[t] = Piece(); });
    }

    stage("read + bin pairs");
    std::cout << "Initializing clusterer" << std::endl;
    mpe_params prm{};
    prm.fragment_mean = fragmentMean;
    prm.fragment_stddev = fragmentStdDev;
    prm.min_cluster_size = minClusterSize;
    {   // MatePairEM::Initialize (tools/MatePairEM.cpp:43-58)
        const double x = -fragmentStdDev * normal_01_cdf_inverse((1 - precision) / 2);
        prm.min_probability = normalpdf(x, 0, fragmentStdDev);
    }

    OrderedFileWriter out;
    if (!out.open_file(cmd.str("clusters"))) die("Error: unable to write to clusters file");

    std::cout << "Creating clusters" << std::endl;
    // per bin pair: unpack, match fragments, drop overlapping alignments, enumerate alignment pairs (:478-545)
    std::vector<uint32_t> bpOrder(binPairKey.size());
    std::iota(bpOrder.begin(), bpOrder.end(), 0u);
    std::sort(bpOrder.begin(), bpOrder.end(), [&](uint32_t a, uint32_t b) { return binPairKey[a] < binPairKey[b]; });
    // Bin pairs are independent.  They are taken in canonical order in a few CHUNKS (contiguous ranges of about equal
    // alignment count); a chunk goes through three stages — (1) its problems are built, every host thread on a contiguous
    // share of the chunk's bin pairs; (2) the mate pair EM of the chunk runs on the GPU; (3) its cluster lines are formatted
    // and written — and the stages of consecutive chunks overlap: while the GPU clusters chunk c, the host builds chunk
    // c+1 and writes chunk c-1.  Cluster ids are a running count over the chunks in order, so the file does not depend on
    // the cut.  DEFUSE_CMP_CHUNKS sets the number of chunks (default: 1 below 8 M alignments in bin pairs, else 4).
    struct PartOut {
        std::vector<Problem> problems;
        std::vector<int64_t> sizes;
        std::vector<double> X, Y, U;
        std::vector<int32_t> toXO, toYO;
    };
    struct Chunk {
        std::vector<Problem> problems;
        std::vector<int64_t> probOff;
        std::vector<double> X, Y, U;
        std::vector<int32_t> toXO, toYO, nClusters;
        std::vector<uint16_t> member;
    };
    size_t totalAlignments = 0;
    for (const PairView& pp : binPairStore) totalAlignments += pp.first.size() + pp.second.size();
    const bool dumping = std::getenv("DEFUSE_CMP_DUMP_PROBLEMS") || std::getenv("DEFUSE_CMP_DUMP_EM");
    unsigned nChunks = totalAlignments >= ((size_t)8 << 20) ? 4u : 1u;
    if (const char* e = std::getenv("DEFUSE_CMP_CHUNKS")) nChunks = (unsigned)std::max(1, std::atoi(e));
    if (dumping || bpOrder.empty()) nChunks = 1;
    nChunks = (unsigned)std::min<size_t>(nChunks, std::max<size_t>(1, bpOrder.size()));
    auto cut_by_alignments = [&](size_t lo, size_t hi, unsigned parts_n) {     // [lo, hi) of bpOrder into parts of about equal alignment count
        std::vector<size_t> cutAt(parts_n + 1, hi);
        cutAt[0] = lo;
        size_t total = 0;
        for (size_t oi = lo; oi < hi; ++oi) total += binPairStore[bpOrder[oi]].first.size() + binPairStore[bpOrder[oi]].second.size();
        size_t acc = 0;
        unsigned t = 1;
        for (size_t oi = lo; oi < hi && t < parts_n; ++oi) {
            acc += binPairStore[bpOrder[oi]].first.size() + binPairStore[bpOrder[oi]].second.size();
            while (t < parts_n && acc >= total / parts_n * t) cutAt[t++] = oi + 1;
        }
        return cutAt;
    };
    const std::vector<size_t> chunkCut = cut_by_alignments(0, bpOrder.size(), nChunks);

    auto build_chunk = [&](size_t cLo, size_t cHi, Chunk& ck) {
    std::vector<PartOut> parts(nThreads);
    const std::vector<size_t> share = cut_by_alignments(cLo, cHi, nThreads);
    auto build_share = [&](unsigned t) {
        PartOut& o = parts[t];
        const size_t lo = share[t], hi = share[t + 1];
        FragmentGroups fr1, fr2, fr2all;
        for (size_t oi = lo; oi < hi; ++oi) {
        const uint32_t bpi = bpOrder[oi];
        const PairView& pp = binPairStore[bpi];
        if ((int)pp.first.size() < minClusterSize || (int)pp.second.size() < minClusterSize) continue;
        Problem prob;
        auto unpack = [&](unsigned id, const ListView& packed, std::vector<CompactAlignment>& al) {
            const int ref = (int)(id & 0x3FFFFu), strand = (int)((id >> 18) & 1u), bin = (int)(id >> 19);
            al.resize(packed.size());
            for (size_t k = 0; k < packed.size(); ++k) {
                al[k].referenceIndex = ref;
                al[k].strand = strand;
                al[k].fragmentIndex = packed[k].fragmentIndex;
                al[k].readEnd = packed[k].readEnd;
                al[k].region.start = packed[k].relativeStart + bin * binLength - binLength / 2;
                al[k].region.end = packed[k].relativeEnd + bin * binLength - binLength / 2;
            }
        };
        unpack((unsigned)(binPairKey[bpi] >> 32), pp.first, prob.alignments1);
        unpack((unsigned)(binPairKey[bpi] & 0xFFFFFFFFu), pp.second, prob.alignments2);
        GroupByFragment(prob.alignments1, fr1);
        GroupByFragment(prob.alignments2, fr2);
        fr2all = fr2;
        FilterSide(fr2, fr1, prob.alignments2, minFusionRange);      // FilterUnmatched: only the fragment sets matter
        FilterSide(fr1, fr2all, prob.alignments1, minFusionRange);
        if ((int)fr1.size() < minClusterSize || (int)fr2.size() < minClusterSize) continue;
        // GetAlignPairs (:360-375): per fragment the cartesian product; an alignment index belongs to one fragment,
        // so no pair can come twice
        for (size_t f = 0; f < fr1.size(); ++f)
            for (int k1 = fr1.off[f]; k1 < fr1.off[f + 1]; ++k1)
                for (int k2 = fr2.off[f]; k2 < fr2.off[f + 1]; ++k2) prob.alignPairs.push_back(std::make_pair(fr1.idx[k1], fr2.idx[k2]));
        const size_t n = prob.alignPairs.size(), base = o.X.size();
        for (size_t k = 0; k < n; ++k) {
            const CompactAlignment& a1 = prob.alignments1[prob.alignPairs[k].first];
            const CompactAlignment& a2 = prob.alignments2[prob.alignPairs[k].second];
            const Region r1 = StrandRemap(a1.region, a1.strand), r2 = StrandRemap(a2.region, a2.strand);
            o.X.push_back(r1.end);
            o.Y.push_back(r2.end);
            o.U.push_back(fragmentMean - (r1.end - r1.start + 1) - (r2.end - r2.start + 1));   // DoClustering :554-556
        }
        // ranks when sorted by x (y) descending, ties by index ascending (the reference's std::sort leaves ties open)
        std::vector<int> ord(n);
        o.toXO.resize(base + n);
        o.toYO.resize(base + n);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return o.X[base + a] > o.X[base + b]; });
        for (size_t r = 0; r < n; ++r) o.toXO[base + ord[r]] = (int32_t)r;
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return o.Y[base + a] > o.Y[base + b]; });
        for (size_t r = 0; r < n; ++r) o.toYO[base + ord[r]] = (int32_t)r;
        o.sizes.push_back((int64_t)n);
        o.problems.push_back(std::move(prob));
    }
    };
    run_threads(build_share);
    ck.probOff.assign(1, 0);
    for (PartOut& o : parts) {
        for (size_t k = 0; k < o.problems.size(); ++k) {
            ck.problems.push_back(std::move(o.problems[k]));
            ck.probOff.push_back(ck.probOff.back() + o.sizes[k]);
        }
        ck.X.insert(ck.X.end(), o.X.begin(), o.X.end());
        ck.Y.insert(ck.Y.end(), o.Y.begin(), o.Y.end());
        ck.U.insert(ck.U.end(), o.U.begin(), o.U.end());
        ck.toXO.insert(ck.toXO.end(), o.toXO.begin(), o.toXO.end());
        ck.toYO.insert(ck.toYO.end(), o.toYO.begin(), o.toYO.end());
        o = PartOut();
    }
    };

    // DEFUSE_GPUS = "all", a count, or a list of device ordinals: the bin pairs are shared out over those GPUs
    // (bin pairs are independent); otherwise one GPU as for every tool (DEFUSE_GPU / lock files)
    std::vector<int> devices;
    mpe_timing tsum{};
    auto cluster_chunk = [&](Chunk& ck) {
        ck.nClusters.assign(ck.problems.size(), 0);
        ck.member.assign(ck.X.size(), 0);
        if (ck.problems.empty()) return;
        if (devices.empty()) {
            if (const char* g = std::getenv("DEFUSE_GPUS")) {
                const std::string spec = g;
                const int have = dsa_device_count();
                if (spec == "all") for (int d = 0; d < have; ++d) devices.push_back(d);
                else if (spec.find(',') != std::string::npos) for (const std::string& f : split_tabs(spec, ',')) devices.push_back(std::atoi(f.c_str()));
                else for (int d = 0; d < std::atoi(spec.c_str()); ++d) devices.push_back(have > 0 ? d % have : d);
            }
            if (devices.empty()) devices.push_back(first_device());
        }
        std::vector<int32_t> status(ck.problems.size(), 0);
        mpe_timing t;
        const int rc = mpe_cluster_batch_sharded(devices.data(), (int32_t)devices.size(), &prm, ck.probOff.data(), (int32_t)ck.problems.size(),
                                                 ck.X.data(), ck.Y.data(), ck.U.data(), ck.toXO.data(), ck.toYO.data(), ck.nClusters.data(),
                                                 ck.member.data(), status.data(), &t);
        if (rc != 0) die(std::string("Error: mate pair clustering on the GPU failed: ") + mpe_last_error());
        tsum.kernel_ms += t.kernel_ms; tsum.n_problems += t.n_problems; tsum.n_mate_pairs += t.n_mate_pairs;
        tsum.em_iterations += t.em_iterations; tsum.n_wave_problems += t.n_wave_problems;
        for (size_t p = 0; p < ck.problems.size(); ++p)
            if (status[p]) die("Error: a consistency check of the mate pair clusterer failed (DebugCheck in the reference)");
    };

    // output (:549-583): per emitted cluster one alignment pair per distinct fragment, in mate pair order; every host thread
    // formats a contiguous share of the problems (cluster ids from a prefix sum of the emitted clusters), the texts are
    // written in order
    int clusterID = 0;
    auto write_chunk = [&](Chunk& ck) {
        const std::vector<Problem>& problems = ck.problems;
        const std::vector<int64_t>& probOff = ck.probOff;
        std::vector<int> firstCluster(problems.size() + 1, clusterID);
        for (size_t p = 0; p < problems.size(); ++p) firstCluster[p + 1] = firstCluster[p] + ck.nClusters[p];
        clusterID = firstCluster[problems.size()];
        std::vector<size_t> outShare(nThreads + 1, 0);
        std::vector<std::string> texts(nThreads);
        auto format_share = [&](unsigned t) {
            std::string& buf = texts[t];
            std::vector<int> usedFragments;
            auto put_int = [&](long long v) {
                char tmp[16];
                buf.append(tmp, (size_t)(defuse::put_int(tmp, (int)v) - tmp));
            };
            for (size_t p = outShare[t]; p < outShare[t + 1]; ++p) {
                const Problem& prob = problems[p];
                const int64_t base = probOff[p];
                for (int j = 0; j < ck.nClusters[p]; ++j) {
                    const int id = firstCluster[p] + j;
                    usedFragments.clear();
                    for (size_t k = 0; k < prob.alignPairs.size(); ++k) {
                        if (!((ck.member[base + k] >> j) & 1)) continue;
                        const CompactAlignment& a1 = prob.alignments1[prob.alignPairs[k].first];
                        const CompactAlignment& a2 = prob.alignments2[prob.alignPairs[k].second];
                        // mate pairs are listed fragment by fragment, so a fragment seen before is the last one used
                        if (!usedFragments.empty() && usedFragments.back() == a1.fragmentIndex) continue;
                        usedFragments.push_back(a1.fragmentIndex);
                        for (int ce = 0; ce <= 1; ++ce) {
                            const CompactAlignment& a = ce ? a2 : a1;
                            put_int(id); buf += '\t'; put_int(ce); buf += '\t'; put_int(a.fragmentIndex); buf += '\t'; put_int(a.readEnd);
                            buf += '\t'; buf += refNames[a.referenceIndex]; buf += '\t'; buf += (a.strand == PlusStrand ? '+' : '-'); buf += '\t';
                            put_int(a.region.start); buf += '\t'; put_int(a.region.end); buf += '\n';
                        }
                    }
                }
            }
        };
        for (size_t lo = 0; lo < problems.size();) {              // rounds of about four million mate pairs bound the text held in memory
            size_t hi = lo;
            while (hi < problems.size() && probOff[hi] - probOff[lo] < (1 << 22)) ++hi;
            outShare[0] = lo;
            for (unsigned t = 1; t <= nThreads; ++t) {
                const int64_t want = probOff[lo] + (probOff[hi] - probOff[lo]) / nThreads * t;
                size_t at = t == nThreads ? hi : (size_t)(std::lower_bound(probOff.begin() + lo, probOff.begin() + hi, want) - probOff.begin());
                outShare[t] = std::min(std::max(at, outShare[t - 1]), hi);
            }
            run_threads(format_share);
            out.write_round_async(std::move(texts), nThreads);          // copied into the file while the next round is formatted
            texts = std::vector<std::string>(nThreads);
            lo = hi;
        }
    };

    std::vector<Chunk> chunks(nChunks);
    if (dumping) {
        Chunk& ck = chunks[0];
        build_chunk(0, bpOrder.size(), ck);
        stage("problems");
        const std::vector<Problem>& problems = ck.problems;
        const std::vector<int64_t>& probOff = ck.probOff;
        const std::vector<double>&X = ck.X, &Y = ck.Y, &U = ck.U;
        const std::vector<int32_t>&toXO = ck.toXO, &toYO = ck.toYO;
        if (const char* dump = std::getenv("DEFUSE_CMP_DUMP_PROBLEMS")) {      // regression aid: the host stages' result, no GPU needed
            std::ofstream d(dump, std::ios::binary);
            auto put = [&](const void* p, size_t n) { d.write((const char*)p, (std::streamsize)n); };
            put(probOff.data(), probOff.size() * sizeof(int64_t));
            put(X.data(), X.size() * sizeof(double)); put(Y.data(), Y.size() * sizeof(double)); put(U.data(), U.size() * sizeof(double));
            put(toXO.data(), toXO.size() * sizeof(int32_t)); put(toYO.data(), toYO.size() * sizeof(int32_t));
            for (const Problem& pr : problems) {
                put(pr.alignments1.data(), pr.alignments1.size() * sizeof(CompactAlignment));
                put(pr.alignments2.data(), pr.alignments2.size() * sizeof(CompactAlignment));
                put(pr.alignPairs.data(), pr.alignPairs.size() * sizeof(std::pair<int, int>));
            }
            for (const std::string& r : refNames) d << r << "\n";
            return 0;
        }
        const char* dump = std::getenv("DEFUSE_CMP_DUMP_EM");                  // test aid: exactly the arrays mpe_cluster_batch receives, then stop
        std::ofstream d(dump, std::ios::binary);
        const int64_t head[2] = {(int64_t)problems.size(), (int64_t)X.size()};
        auto put = [&](const void* p, size_t n) { d.write((const char*)p, (std::streamsize)n); };
        put(head, sizeof(head));
        put(&prm, sizeof(prm));
        put(probOff.data(), probOff.size() * sizeof(int64_t));
        put(X.data(), X.size() * sizeof(double)); put(Y.data(), Y.size() * sizeof(double)); put(U.data(), U.size() * sizeof(double));
        put(toXO.data(), toXO.size() * sizeof(int32_t)); put(toYO.data(), toYO.size() * sizeof(int32_t));
        return d.good() ? 0 : 1;
    }
    {
        // stage hand-offs: built[c] / clustered[c] are set under one mutex; the GPU thread and the writer wait for them in order
        std::mutex mu;
        std::condition_variable cv;
        std::vector<char> built(nChunks, 0), clustered(nChunks, 0);
        double t_build = 0, t_cluster = 0, t_write = 0;
        auto wait_for = [&](std::vector<char>& flag, unsigned c) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return flag[c] != 0; });
        };
        auto signal = [&](std::vector<char>& flag, unsigned c) {
            { std::lock_guard<std::mutex> lk(mu); flag[c] = 1; }
            cv.notify_all();
        };
        std::thread gpu([&] {
            for (unsigned c = 0; c < nChunks; ++c) {
                wait_for(built, c);
                const double t0 = now();
                cluster_chunk(chunks[c]);
                t_cluster += now() - t0;
                signal(clustered, c);
            }
        });
        std::thread writer([&] {
            for (unsigned c = 0; c < nChunks; ++c) {
                wait_for(clustered, c);
                const double t0 = now();
                write_chunk(chunks[c]);
                chunks[c] = Chunk();
                t_write += now() - t0;
            }
        });
        for (unsigned c = 0; c < nChunks; ++c) {
            const double t0 = now();
            build_chunk(chunkCut[c], chunkCut[c + 1], chunks[c]);
            t_build += now() - t0;
            signal(built, c);
        }
        gpu.join();
        writer.join();
        if (timing) {
            std::cerr << "[clustermatepairs] " << tsum.n_problems << " bin pairs, " << tsum.n_mate_pairs << " mate pairs, " << tsum.em_iterations
                      << " EM iterations, " << tsum.n_wave_problems << " bin pairs with a wave each, " << devices.size() << " device share(s), kernel "
                      << tsum.kernel_ms << " ms" << std::endl;
            std::cerr << "[clustermatepairs] " << nChunks << " chunk(s), stages overlapped: problems " << t_build << " s, clustering " << t_cluster
                      << " s, output " << t_write << " s" << std::endl;
        }
    }
    if (!out.close_file()) die("Error: failed writing the clusters file");
    stage("problems + clustering + output");
    std::cout << "Created " << clusterID << " clusters" << std::endl;
    // The clusters file is complete and closed.  The process ends here without destroying its tables one by one (gigabytes in
    // millions of blocks at full size: more than a second of the round-3 tool's 8.9 s) and without the GPU runtime's own
    // shutdown; the system takes all of it back at once.
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    if (std::getenv("DEFUSE_FULL_EXIT")) exit(0);      // under a profiler that writes its files at exit (rocprofv3): atexit handlers run
    _exit(0);
}
