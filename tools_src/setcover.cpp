// setcover — drop-in replacement of the reference tool (tools/setcover.cpp:112-147): same command
// line, same cluster file format in and out, same progress lines on stdout.  ReadClusters /
// WriteClusters follow tools/Parsers.cpp:23-170; SetCover itself runs on the GPU through
// include/defuse_sc.h (no CPU fallback: without a HIP device the tool exits 1).
#include "../include/defuse_dsa.h"
#include "../include/defuse_sc.h"
#include "defuse_host.hpp"

#include <chrono>

using namespace defuse;

namespace {

struct ClusterLine { int clusterID, clusterEnd, fragmentIndex; };

}  // namespace

int main(int argc, char* argv[])
{
    CmdLine cmd("Set cover for maximum parsimony");
    cmd.add("c", "clusters", "Clusters Filename", "string");
    cmd.add("m", "minclustersize", "Minimum Cluster Size", "integer");
    cmd.add("o", "outclust", "Output Clusters Filename", "string");
    cmd.parse(argc, argv);
    const std::string inName = cmd.str("clusters"), outName = cmd.str("outclust");
    const int minClusterSize = cmd.integer("minclustersize");

    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_stage = now();
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[setcover] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t;
    };
    std::cout << "Reading clusters" << std::endl;
    // ReadClusters (tools/Parsers.cpp:23-84): cluster-end-0 lines only, clusters[id] in file order.  The file is mapped and
    // parsed in one piece per host thread; the pieces' (cluster, fragment) lists are laid out cluster by cluster in file order.
    MappedText text;
    text.load(inName, "Error: unable to read from clusters file ");
    unsigned nThreads = host_threads();
    if (text.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nThreads = 1;      // small inputs: threads only on request (tests)
    struct Piece {
        std::vector<std::pair<int, int>> members;          // (cluster, fragment) of the end-0 lines, in order
        size_t lines = 0, errorLine = 0;
        int errorKind = 0;                                 // 1 empty line, 2 format, 3 not an integer, 4 negative cluster id
        std::string errorText;
        int maxCluster = -1;
    };
    std::vector<Piece> pieces(nThreads);
    const std::vector<size_t> cut = text.cut_lines(0, text.size(), nThreads);
    auto parse_line = [](const char* line, size_t len, ClusterLine& out) -> int {     // 0 ok, else the error kind
        if (len == 0) return 1;
        const char* end = line + len;
        const char* t1 = (const char*)memchr(line, '\t', len);
        const char* t2 = t1 ? (const char*)memchr(t1 + 1, '\t', (size_t)(end - t1 - 1)) : nullptr;
        if (!t2) return 2;
        const char* t3 = (const char*)memchr(t2 + 1, '\t', (size_t)(end - t2 - 1));
        if (!t3) t3 = end;
        if (!field_int(line, (size_t)(t1 - line), out.clusterID) || !field_int(t1 + 1, (size_t)(t2 - t1 - 1), out.clusterEnd) ||
            !field_int(t2 + 1, (size_t)(t3 - t2 - 1), out.fragmentIndex))
            return 3;
        return 0;
    };
    auto report = [&](int kind, size_t lineNumber, const std::string& lineText, const std::string& filename) {
        if (kind == 1) die("Error: Empty clusters line " + std::to_string(lineNumber) + " of " + filename);
        if (kind == 2) die("Error: Format error for clusters line " + std::to_string(lineNumber) + " of " + filename);
        if (kind == 3) { std::cerr << "Failed to interpret line:" << std::endl << lineText << std::endl; std::exit(1); }
        die("Error: Invalid cluster ID for line " + std::to_string(lineNumber) + " of " + filename);
    };
    run_threads(nThreads, [&](unsigned t) {
        Piece& pc = pieces[t];
        pc.members.reserve((cut[t + 1] - cut[t]) / 60 + 16);
        ClusterLine cl;
        for (size_t pos = cut[t]; pos < cut[t + 1];) {
            const size_t e = text.line_end(pos);
            const char* line = text.data() + pos;
            const size_t len = (e > pos && text[e - 1] == '\n') ? e - 1 - pos : e - pos;
            pos = e;
            ++pc.lines;
            int kind = parse_line(line, len, cl);
            if (!kind && cl.clusterEnd == 0 && cl.clusterID < 0) kind = 4;
            if (kind) { pc.errorLine = pc.lines; pc.errorKind = kind; pc.errorText.assign(line, len); return; }
            if (cl.clusterEnd != 0) continue;
            pc.maxCluster = std::max(pc.maxCluster, cl.clusterID);
            pc.members.push_back(std::make_pair(cl.clusterID, cl.fragmentIndex));
        }
    });
    {
        size_t lineBase = 0;
        for (const Piece& pc : pieces) {                     // the first bad line of the file, as a serial reader meets it
            if (pc.errorLine) report(pc.errorKind, lineBase + pc.errorLine, pc.errorText, inName);
            lineBase += pc.lines;
        }
    }
    int maxCluster = -1;
    for (const Piece& pc : pieces) maxCluster = std::max(maxCluster, pc.maxCluster);
    const size_t nClusters = (size_t)(maxCluster + 1);
    stage("read");

    std::cout << "Calculating set cover solution" << std::endl;
    int maxElement = -1;                                  // FindMaxElement (tools/Common.cpp:71-89)
    std::vector<int64_t> off(nClusters + 1, 0);
    for (const Piece& pc : pieces)
        for (const auto& m : pc.members) ++off[(size_t)m.first + 1];
    for (size_t c = 0; c < nClusters; ++c) off[c + 1] += off[c];
    std::vector<int32_t> elements((size_t)off[nClusters]);
    {
        std::vector<int64_t> at(off.begin(), off.end() - 1);
        for (Piece& pc : pieces) {                            // pieces in file order: every cluster keeps its file order
            for (const auto& m : pc.members) {
                if (m.second < 0) die("Error: negative elements not permitted");
                maxElement = std::max(maxElement, m.second);
                elements[(size_t)at[m.first]++] = m.second;
            }
            std::vector<std::pair<int, int>>().swap(pc.members);
        }
    }
    std::vector<int32_t> owner((size_t)maxElement + 1, -1);
    if (!elements.empty()) {
        sc_timing t;
        const int rc = sc_cover(dsa_pick_device(), off.data(), elements.data(), (int32_t)nClusters, maxElement,
                                owner.data(), &t);
        if (rc != 0) die(std::string("Error: set cover on the GPU failed: ") + sc_last_error());
        if (std::getenv("DEFUSE_TIMING"))
            std::cerr << "[setcover] components " << t.n_components << " (large " << t.n_large << "), build " << t.build_ms
                      << " ms, components " << t.components_ms << " ms, greedy " << t.greedy_ms << " ms" << std::endl;
    }
    stage("set cover");
    std::vector<int64_t> solutionSize(nClusters, 0);
    for (int32_t o : owner)
        if (o >= 0) ++solutionSize[o];

    std::cout << "Writing out clusters" << std::endl;
    // WriteClusters (tools/Parsers.cpp:86-170): copy the input lines (both ends) whose fragment was
    // assigned to the line's cluster, for clusters that kept at least minClusterSize fragments
    OrderedFileWriter out;
    if (!out.open_file(outName)) die("Error: unable to write to clusters file " + outName);
    // the lines are filtered in rounds of 256 MiB of input, each cut into one piece per host thread; texts written in order
    std::vector<std::string> texts(nThreads);
    std::vector<Piece> round(nThreads);
    size_t lineBase = 0;
    for (size_t lo = 0; lo < text.size();) {
        size_t hi = std::min(text.size(), lo + ((size_t)1 << 28));
        if (hi < text.size()) hi = text.line_end(hi - 1);
        const std::vector<size_t> rc = text.cut_lines(lo, hi, nThreads);
        run_threads(nThreads, [&](unsigned t) {
            Piece& pc = round[t];
            pc = Piece();
            std::string& buf = texts[t];
            buf.clear();
            ClusterLine cl;
            for (size_t pos = rc[t]; pos < rc[t + 1];) {
                const size_t e = text.line_end(pos);
                const char* line = text.data() + pos;
                const size_t len = (e > pos && text[e - 1] == '\n') ? e - 1 - pos : e - pos;
                pos = e;
                ++pc.lines;
                int kind = parse_line(line, len, cl);
                if (!kind && cl.clusterID < 0) kind = 4;
                if (kind) { pc.errorLine = pc.lines; pc.errorKind = kind; pc.errorText.assign(line, len); return; }
                if ((size_t)cl.clusterID >= nClusters) continue;            // an id that only occurs with end 1 (UB in the reference)
                if ((int64_t)solutionSize[cl.clusterID] < (int64_t)minClusterSize) continue;
                if (cl.fragmentIndex >= 0 && cl.fragmentIndex <= maxElement && owner[cl.fragmentIndex] == cl.clusterID) {
                    buf.append(line, len);
                    buf += '\n';
                }
            }
        });
        for (unsigned t = 0; t < nThreads; ++t) {
            if (round[t].errorLine) report(round[t].errorKind, lineBase + round[t].errorLine, round[t].errorText, outName);
            lineBase += round[t].lines;
        }
        out.write_round(texts, nThreads);
        lo = hi;
    }
    stage("write");
    return out.close_file() ? 0 : 1;
}
