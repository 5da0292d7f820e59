// calccov — drop-in replacement of the reference tool (tools/calccov.cpp:66-250): same command line, same inputs, the same
// three sample files.  Sample positions are drawn on single-transcript genes (or all genes with --multiexon) with the C
// library's rand() after srand(11), exactly as the reference does (:116, :137) — so, as there, the positions are those of
// the platform's libc (glibc: the TYPE_3 additive-feedback generator; oracle/calccov_oracle.py restates it).  The
// per-fragment search of the concordant alignments for the samples they cover (:155-215) runs on the GPU through
// include/defuse_cov.h.  Orders the reference leaves to boost::unordered_set are the canonical ones of SURVEY.md 8(c):
// genes ascending by name, the samples of a fragment ascending by index.  No CPU fallback: without a GPU the tool exits 1
// as soon as there is a fragment to look at.
#include <chrono>

#include "../include/defuse_cov.h"
#include "../include/defuse_dsa.h"
#include "defuse_host.hpp"

using namespace defuse;

int main(int argc, char* argv[])
{
    CmdLine cmd("Calculate covariance stats from concordant alignments");
    cmd.add("c", "conc", "Concordant Sam Filename", "string");
    cmd.add("g", "genetran", "Gene Transcripts Filename", "string");
    cmd.add("l", "len", "Spanning Length Samples Filename", "string");
    cmd.add("p", "pos", "Split Position Samples Filename", "string");
    cmd.add("m", "min", "Split Minimum Samples Filename", "string");
    cmd.add("d", "density", "Covariance Sampling Density", "float");
    cmd.add("a", "anchor", "Gene Transcripts Filename", "integer");          // the reference's own description text
    cmd.add("t", "trim", "Trim Length for Spanning Alignments", "integer");
    cmd.add_switch("", "multiexon", "Use Multi-Exon Transcripts");
    cmd.parse(argc, argv);
    const double density = cmd.real("density");
    const int anchor = cmd.integer("anchor"), trim = cmd.integer("trim");
    const bool multiexon = cmd.is_set("multiexon");
    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_stage = now();
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[calccov] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t;
    };

    ExonRegions geneTranscripts;
    {
        std::ifstream f(cmd.str("genetran").c_str());
        if (!f.good() || !geneTranscripts.Read(f)) die("Error: Unable to gene transcripts file " + cmd.str("genetran"));
    }

    // :116-150: sample positions, transcript by transcript
    srand(11);
    std::unordered_map<std::string, int> refIndex;          // NameIndex: "gene|transcript" -> index
    std::vector<int64_t> sampleOff(1, 0);
    std::vector<int32_t> samplePos;
    for (const std::string& gene : geneTranscripts.GetGenes()) {
        const std::vector<std::string>& tr = geneTranscripts.GetGeneTranscripts(gene);
        if (tr.size() != 1 && !multiexon) continue;
        const std::string id = gene + "|" + tr[0];
        if (refIndex.emplace(id, (int)sampleOff.size() - 1).second == false) continue;      // (cannot happen: genes are unique)
        const int length = geneTranscripts.GetTranscriptLength(tr[0]);
        const int numMarkers = (int)(length * density);
        for (int k = 0; k < numMarkers; ++k) samplePos.push_back(rand() % length + 1);
        sampleOff.push_back((int64_t)samplePos.size());
    }
    stage("transcripts + sample positions");

    // SamAlignmentStream + FragmentAlignmentStream (tools/AlignmentStream.cpp:39-130, :190-221): fragments are runs of one
    // name; the text is mapped and cut into one piece per host thread at fragment boundaries
    MappedText sam;
    sam.load(cmd.str("conc"), "Error: Unable to open sam file ");
    unsigned nThreads = host_threads();
    if (sam.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nThreads = 1;
    auto name_of = [&](size_t pos, size_t& len) {            // qname of the line at pos without a trailing /1 or /2
        const size_t e = sam.line_end(pos);
        const char* tab = (const char*)memchr(sam.data() + pos, '\t', e - pos);
        len = tab ? (size_t)(tab - (sam.data() + pos)) : e - pos - ((e > pos && sam[e - 1] == '\n') ? 1 : 0);
        const char* q = sam.data() + pos;
        const char* slash = (const char*)memchr(q, '/', len);
        if (slash && !memchr(slash + 1, '/', (size_t)(q + len - slash - 1))) len = (size_t)(slash - q);
        return q;
    };
    std::vector<size_t> cut = sam.cut_lines(0, sam.size(), nThreads);
    for (unsigned t = 1; t < nThreads; ++t) {                // not inside a fragment: move on while the name repeats
        size_t pos = std::max(cut[t], cut[t - 1]);
        while (pos > 0 && pos < sam.size()) {
            if (sam[pos] == '@') break;                        // header lines are skipped by the reader, never part of a run
            size_t prev = pos - 1;
            while (prev > 0 && sam[prev - 1] != '\n') --prev;
            size_t l0, l1;
            const char* n0 = name_of(prev, l0);
            const char* n1 = name_of(pos, l1);
            if (l0 != l1 || memcmp(n0, n1, l0) != 0) break;
            pos = sam.line_end(pos);
        }
        cut[t] = pos;
    }
    struct Piece {
        std::vector<cov_fragment> frags;
        size_t lines = 0, errorLine = 0;
        int errorKind = 0;                                    // ParseSamLine kinds, 7 = not two alignments
        size_t badCount = 0;
        std::string badName;
    };
    std::vector<Piece> pieces(nThreads);
    run_threads(nThreads, [&](unsigned t) {
        Piece& pc = pieces[t];
        SamFields f;
        int readEnd = -1;
        std::string curName, ref0, key;
        Region reg[2];
        size_t count = 0;
        auto close_fragment = [&]() -> bool {                  // GetNextAlignments returned: the checks of :158-170
            if (count == 0) return true;
            if (count != 2) { pc.errorKind = 7; pc.badCount = count; pc.badName = curName; return false; }
            auto ri = refIndex.find(ref0);
            if (ri != refIndex.end()) {
                cov_fragment cf;
                cf.ref = ri->second;
                for (int e = 0; e < 2; ++e) { cf.start[e] = reg[e].start; cf.end[e] = reg[e].end; }
                pc.frags.push_back(cf);
            }
            return true;
        };
        for (size_t pos = cut[t]; pos < cut[t + 1];) {
            const size_t e = sam.line_end(pos);
            const char* line = sam.data() + pos;
            const size_t len = (e > pos && sam[e - 1] == '\n') ? e - 1 - pos : e - pos;
            pos = e;
            ++pc.lines;
            const int kind = ParseSamLine(line, len, f, readEnd);
            if (kind == 1) continue;
            if (kind) { pc.errorLine = pc.lines; pc.errorKind = kind; return; }
            if (count == 0 || f.fragment_len != curName.size() || memcmp(f.fragment, curName.data(), f.fragment_len) != 0) {
                if (!close_fragment()) { pc.errorLine = pc.lines; return; }
                curName.assign(f.fragment, f.fragment_len);
                count = 0;
            }
            if (count == 0) ref0.assign(f.reference, f.reference_len);
            if (count < 2) reg[count] = f.region;
            ++count;
        }
        if (!close_fragment()) pc.errorLine = pc.lines + 1;
    });
    {
        size_t lineBase = 0;
        for (const Piece& pc : pieces) {                      // the first problem of the file, as a serial reader meets it
            if (pc.errorLine) {
                if (pc.errorKind == 7) {
                    std::cerr << "Error: expected 2 alignments per fragment" << std::endl;
                    std::cerr << "retrieved " << pc.badCount << " alignments for " << pc.badName << std::endl;
                    return 1;
                }
                DieSamLine(pc.errorKind, lineBase + pc.errorLine);
            }
            lineBase += pc.lines;
        }
    }
    std::vector<cov_fragment> frags;
    for (Piece& pc : pieces) frags.insert(frags.end(), pc.frags.begin(), pc.frags.end());
    stage("concordant alignments");

    std::vector<int32_t> lenIdx, lenVal, splitIdx;
    std::vector<double> splitPos, splitMin;
    if (!frags.empty()) {
        int64_t nl = 0, ns = 0;
        cov_timing ct;
        const int dev = dsa_pick_device();
        int rc = cov_sample_batch(dev, sampleOff.data(), (int32_t)sampleOff.size() - 1, samplePos.data(), frags.data(), (int64_t)frags.size(),
                                  trim, anchor, nullptr, nullptr, 0, &nl, nullptr, nullptr, nullptr, 0, &ns, &ct);
        if (rc == DSA_E_CAPACITY) {
            lenIdx.resize((size_t)nl); lenVal.resize((size_t)nl);
            splitIdx.resize((size_t)ns); splitPos.resize((size_t)ns); splitMin.resize((size_t)ns);
            rc = cov_sample_batch(dev, sampleOff.data(), (int32_t)sampleOff.size() - 1, samplePos.data(), frags.data(), (int64_t)frags.size(),
                                  trim, anchor, lenIdx.data(), lenVal.data(), nl, &nl, splitIdx.data(), splitPos.data(), splitMin.data(), ns, &ns, &ct);
        }
        if (rc != DSA_OK) die(std::string("Error: sampling on the GPU failed: ") + cov_last_error());
        if (timing) std::cerr << "[calccov] " << frags.size() << " fragments on sampled transcripts, " << nl << " length and " << ns
                              << " split samples, kernel " << ct.kernel_ms << " ms" << std::endl;
    }
    stage("samples");

    // the three files (:217-233): index <tab> value, doubles at the stream's default precision
    auto write_file = [&](const std::string& name, size_t n, const std::function<void(std::string&, size_t)>& put) {
        OrderedFileWriter out;
        if (!out.open_file(name)) die("Error: unable to write to " + name);
        const unsigned nt = n < 65536 ? 1u : nThreads;
        std::vector<std::string> texts(nt);
        run_threads(nt, [&](unsigned t) {
            for (size_t k = n * t / nt; k < n * (t + 1) / nt; ++k) put(texts[t], k);
        });
        out.write_round(texts, nt);
        if (!out.close_file()) die("Error: failed writing " + name);
    };
    auto put_double = [](std::string& buf, double x) {
        char tmp[40];
        buf.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%g", x));
    };
    write_file(cmd.str("len"), lenIdx.size(), [&](std::string& b, size_t k) { append_int(b, lenIdx[k]); b += '\t'; append_int(b, lenVal[k]); b += '\n'; });
    write_file(cmd.str("pos"), splitIdx.size(), [&](std::string& b, size_t k) { append_int(b, splitIdx[k]); b += '\t'; put_double(b, splitPos[k]); b += '\n'; });
    write_file(cmd.str("min"), splitIdx.size(), [&](std::string& b, size_t k) { append_int(b, splitIdx[k]); b += '\t'; put_double(b, splitMin[k]); b += '\n'; });
    stage("output");
    return 0;
}
