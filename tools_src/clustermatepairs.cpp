// clustermatepairs — drop-in replacement of the reference tool (tools/clustermatepairs.cpp:389-589):
// same command line (-a may be "-" for stdin), same compact alignment input, same cluster output
// lines and progress lines.  The host part (concordance filter, 32 kb bin-pair bucketing, per-bin-pair
// filters, output) follows tools/clustermatepairs.cpp:146-375,478-584; MatePairEM::DoClustering for
// all bin pairs runs on the GPU in one batch through include/defuse_mpe.h (no CPU fallback).
// Iteration orders the reference leaves to boost::unordered_map are the canonical ascending-key
// orders of SURVEY.md 8(c).
#include <numeric>

#include "../include/defuse_dsa.h"
#include "../include/defuse_mpe.h"
#include "defuse_host.hpp"

using namespace defuse;

namespace {

struct CompactAlignment { int fragmentIndex, readEnd, referenceIndex, strand; Region region; };
struct AlignmentPacked { int fragmentIndex, readEnd; unsigned short relativeStart, relativeEnd; };

const int binLength = 1 << 15;

// Binning::GetBins (tools/clustermatepairs.cpp:152-162): C++ int division
void GetBins(const Region& region, int length, int extend, std::vector<int>& bins)
{
    const int startBin = (region.start - extend) / length, endBin = (region.end + extend) / length;
    for (int b = startBin; b <= endBin; ++b) bins.push_back(b);
}

unsigned pack_ref_bin(int ref, int strand, int bin)   // RefBinPacked (:28-65)
{
    if (ref >= (1 << 18)) { std::cout << ref << std::endl << (1 << 18) << std::endl; die("Packing failed, too many reference sequences"); }
    if (bin >= (1 << 13)) { std::cout << bin << std::endl << (1 << 13) << std::endl; die("Packing failed, chromosome too large"); }
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

// one surviving bin pair, ready for clustering and for writing its clusters afterwards
struct Problem {
    std::vector<CompactAlignment> alignments1, alignments2;
    std::vector<std::pair<int, int>> alignPairs;
};

typedef std::map<int, std::vector<int>> IntegerVecMap;   // canonical: ascending fragment index

void FilterOverlapping(IntegerVecMap& fragments, const std::vector<CompactAlignment>& alignments, int minFusionRange)   // :316-358
{
    for (auto& kv : fragments) {
        std::set<std::pair<unsigned, int>> bins[2];
        std::vector<int> filtered;
        for (int idx : kv.second) {
            const CompactAlignment& a = alignments[idx];
            std::vector<int> rangeBins;
            GetBins(a.region, minFusionRange, 0, rangeBins);
            const unsigned refStrandId = (unsigned)a.referenceIndex | ((unsigned)a.strand << 31);
            bool overlapping = false;
            for (int b : rangeBins) overlapping |= bins[a.readEnd].count(std::make_pair(refStrandId, b)) != 0;
            if (!overlapping) {
                for (int b : rangeBins) bins[a.readEnd].insert(std::make_pair(refStrandId, b));
                filtered.push_back(idx);
            }
        }
        kv.second.swap(filtered);
    }
}

}  // namespace

int main(int argc, char* argv[])
{
    CmdLine cmd("Mate Pair Clustering Tool");
    cmd.add("a", "align", "Alignments Filename", "string");
    cmd.add("c", "clusters", "Output Clusters Filename", "string");
    cmd.add("u", "fragmentmean", "Fragment Length Mean", "float");
    cmd.add("s", "fragmentstddev", "Fragment Length Standard Deviation", "float");
    cmd.add("p", "precision", "Precision", "float");
    cmd.add("m", "minclustersize", "Minimum Cluster Size", "integer");
    cmd.parse(argc, argv);
    const double fragmentMean = cmd.real("fragmentmean"), fragmentStdDev = cmd.real("fragmentstddev"), precision = cmd.real("precision");
    const int minClusterSize = cmd.integer("minclustersize");
    const int minFusionRange = (int)(fragmentMean + 10 * fragmentStdDev);

    std::cout << "Finding pairs of reference sequences connected by pairs of alignments" << std::endl;
    std::ifstream file;
    std::istream* in = &std::cin;
    if (cmd.str("align") != "-") {
        file.open(cmd.str("align").c_str());
        if (!file.good()) die("Error: Unable to open alignment file " + cmd.str("align"));
        in = &file;
    }
    std::vector<std::string> refNames;
    std::unordered_map<std::string, int> refIndex;
    typedef std::pair<std::vector<AlignmentPacked>, std::vector<AlignmentPacked>> PackedPair;
    std::map<std::pair<unsigned, unsigned>, PackedPair> binPairs;   // canonical: ascending (first.id, second.id)

    auto process_fragment = [&](const std::vector<CompactAlignment>& alignments) {
        // CheckConcordant (:211-244)
        std::set<std::pair<int, int>> conc[2];
        for (const CompactAlignment& a : alignments) {
            std::vector<int> bins;
            GetBins(a.region, minFusionRange, minFusionRange, bins);
            for (int b : bins) conc[a.readEnd].insert(std::make_pair(a.referenceIndex, b));
        }
        for (const auto& rb : conc[0])
            if (conc[1].count(rb)) return;
        // AddBinPairs (:246-290)
        std::map<unsigned, std::vector<AlignmentPacked>> binned[2];
        for (const CompactAlignment& a : alignments) {
            std::vector<int> bins;
            GetBins(a.region, binLength, minFusionRange, bins);
            for (int b : bins) {
                const int rs = a.region.start - b * binLength + binLength / 2, re = a.region.end - b * binLength + binLength / 2;
                if (rs < 0 || re < 0 || rs >= (1 << 16) || re >= (1 << 16)) die("Error: relativeStart >= 0 failed (alignment does not fit its bin)");
                binned[a.readEnd][pack_ref_bin(a.referenceIndex, a.strand, b)].push_back(
                    AlignmentPacked{a.fragmentIndex, a.readEnd, (unsigned short)rs, (unsigned short)re});
            }
        }
        for (const auto& b1 : binned[0])
            for (const auto& b2 : binned[1]) {
                if (b1.first < b2.first) {
                    PackedPair& e = binPairs[std::make_pair(b1.first, b2.first)];
                    e.first.insert(e.first.end(), b1.second.begin(), b1.second.end());
                    e.second.insert(e.second.end(), b2.second.begin(), b2.second.end());
                } else {
                    PackedPair& e = binPairs[std::make_pair(b2.first, b1.first)];
                    e.first.insert(e.first.end(), b2.second.begin(), b2.second.end());
                    e.second.insert(e.second.end(), b1.second.begin(), b1.second.end());
                }
            }
    };

    {   // CompactAlignmentStream + FragmentAlignmentStream (tools/AlignmentStream.cpp:156-221)
        std::string line, curName;
        std::vector<CompactAlignment> cur;
        int lineNumber = 0;
        while (std::getline(*in, line)) {
            ++lineNumber;
            if (line.empty()) die("Error: Empty alignment line " + std::to_string(lineNumber));
            std::vector<std::string> f = split_tabs(line);
            if (f.size() < 6) die("Error: Format error for alignment line " + std::to_string(lineNumber));
            if (!cur.empty() && f[0] != curName) {
                process_fragment(cur);
                cur.clear();
            }
            curName = f[0];
            CompactAlignment a;
            a.fragmentIndex = lexical_int_or_die(f[0], "as fragment name on line " + std::to_string(lineNumber));
            a.readEnd = (f[1] == "1") ? 0 : 1;
            auto ri = refIndex.find(f[2]);
            if (ri == refIndex.end()) {
                ri = refIndex.emplace(f[2], (int)refNames.size()).first;
                refNames.push_back(f[2]);
            }
            a.referenceIndex = ri->second;
            a.strand = (f[3] == "-") ? MinusStrand : PlusStrand;
            a.region.start = lexical_int_or_die(f[4], "on line " + std::to_string(lineNumber));
            a.region.end = lexical_int_or_die(f[5], "on line " + std::to_string(lineNumber));
            cur.push_back(a);
        }
        if (!cur.empty()) process_fragment(cur);
    }

    std::cout << "Initializing clusterer" << std::endl;
    mpe_params prm{};
    prm.fragment_mean = fragmentMean;
    prm.fragment_stddev = fragmentStdDev;
    prm.min_cluster_size = minClusterSize;
    {   // MatePairEM::Initialize (tools/MatePairEM.cpp:43-58)
        const double x = -fragmentStdDev * normal_01_cdf_inverse((1 - precision) / 2);
        prm.min_probability = normalpdf(x, 0, fragmentStdDev);
    }

    std::ofstream out(cmd.str("clusters").c_str());
    if (!out) die("Error: unable to write to clusters file");

    std::cout << "Creating clusters" << std::endl;
    // per bin pair: unpack, match fragments, drop overlapping alignments, enumerate alignment pairs (:478-545)
    std::vector<Problem> problems;
    std::vector<int64_t> probOff(1, 0);
    std::vector<double> X, Y, U;
    std::vector<int32_t> toXO, toYO;
    for (const auto& bp : binPairs) {
        const PackedPair& pp = bp.second;
        if ((int)pp.first.size() < minClusterSize || (int)pp.second.size() < minClusterSize) continue;
        Problem prob;
        auto unpack = [&](unsigned id, const std::vector<AlignmentPacked>& packed, std::vector<CompactAlignment>& al) {
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
        unpack(bp.first.first, pp.first, prob.alignments1);
        unpack(bp.first.second, pp.second, prob.alignments2);
        IntegerVecMap fr1, fr2;
        for (size_t k = 0; k < prob.alignments1.size(); ++k) fr1[prob.alignments1[k].fragmentIndex].push_back((int)k);
        for (size_t k = 0; k < prob.alignments2.size(); ++k) fr2[prob.alignments2[k].fragmentIndex].push_back((int)k);
        for (auto it = fr2.begin(); it != fr2.end();) it = fr1.count(it->first) ? std::next(it) : fr2.erase(it);   // FilterUnmatched
        for (auto it = fr1.begin(); it != fr1.end();) it = fr2.count(it->first) ? std::next(it) : fr1.erase(it);
        FilterOverlapping(fr1, prob.alignments1, minFusionRange);
        FilterOverlapping(fr2, prob.alignments2, minFusionRange);
        if ((int)fr1.size() < minClusterSize || (int)fr2.size() < minClusterSize) continue;
        std::set<std::pair<int, int>> seen;                           // PairIndex: first-seen order
        for (const auto& kv : fr1)
            for (int i1 : kv.second)
                for (int i2 : fr2[kv.first])
                    if (seen.insert(std::make_pair(i1, i2)).second) prob.alignPairs.push_back(std::make_pair(i1, i2));
        const size_t n = prob.alignPairs.size(), base = X.size();
        for (size_t k = 0; k < n; ++k) {
            const CompactAlignment& a1 = prob.alignments1[prob.alignPairs[k].first];
            const CompactAlignment& a2 = prob.alignments2[prob.alignPairs[k].second];
            const Region r1 = StrandRemap(a1.region, a1.strand), r2 = StrandRemap(a2.region, a2.strand);
            X.push_back(r1.end);
            Y.push_back(r2.end);
            U.push_back(fragmentMean - (r1.end - r1.start + 1) - (r2.end - r2.start + 1));   // DoClustering :554-556
        }
        // ranks when sorted by x (y) descending, ties by index ascending (the reference's std::sort leaves ties open)
        std::vector<int> ord(n);
        toXO.resize(base + n);
        toYO.resize(base + n);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return X[base + a] > X[base + b]; });
        for (size_t r = 0; r < n; ++r) toXO[base + ord[r]] = (int32_t)r;
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return Y[base + a] > Y[base + b]; });
        for (size_t r = 0; r < n; ++r) toYO[base + ord[r]] = (int32_t)r;
        probOff.push_back((int64_t)X.size());
        problems.push_back(std::move(prob));
    }
    binPairs.clear();

    std::vector<int32_t> nClusters(problems.size(), 0), status(problems.size(), 0);
    std::vector<uint16_t> member(X.size(), 0);
    if (!problems.empty()) {
        mpe_timing t;
        const int rc = mpe_cluster_batch(dsa_pick_device(), &prm, probOff.data(), (int32_t)problems.size(), X.data(), Y.data(),
                                         U.data(), toXO.data(), toYO.data(), nClusters.data(), member.data(), status.data(), &t);
        if (rc != 0) die(std::string("Error: mate pair clustering on the GPU failed: ") + mpe_last_error());
        if (std::getenv("DEFUSE_TIMING"))
            std::cerr << "[clustermatepairs] " << t.n_problems << " bin pairs, " << t.n_mate_pairs << " mate pairs, " << t.em_iterations
                      << " EM iterations, " << t.n_wave_problems << " bin pairs with a wave per fit, kernel " << t.kernel_ms << " ms" << std::endl;
        for (size_t p = 0; p < problems.size(); ++p)
            if (status[p]) die("Error: a consistency check of the mate pair clusterer failed (DebugCheck in the reference)");
    }

    // output (:549-583): per emitted cluster one alignment pair per distinct fragment, in mate pair order
    int clusterID = 0;
    for (size_t p = 0; p < problems.size(); ++p) {
        const Problem& prob = problems[p];
        const int64_t base = probOff[p];
        for (int j = 0; j < nClusters[p]; ++j) {
            std::set<int> usedFragments;
            for (size_t k = 0; k < prob.alignPairs.size(); ++k) {
                if (!((member[base + k] >> j) & 1)) continue;
                const CompactAlignment& a1 = prob.alignments1[prob.alignPairs[k].first];
                const CompactAlignment& a2 = prob.alignments2[prob.alignPairs[k].second];
                if (!usedFragments.insert(a1.fragmentIndex).second) continue;
                for (int ce = 0; ce <= 1; ++ce) {
                    const CompactAlignment& a = ce ? a2 : a1;
                    out << clusterID << "\t" << ce << "\t" << a.fragmentIndex << "\t" << a.readEnd << "\t" << refNames[a.referenceIndex]
                        << "\t" << (a.strand == PlusStrand ? "+" : "-") << "\t" << a.region.start << "\t" << a.region.end << std::endl;
                }
            }
            ++clusterID;
        }
    }
    out.close();
    std::cout << "Created " << clusterID << " clusters" << std::endl;
    return 0;
}
