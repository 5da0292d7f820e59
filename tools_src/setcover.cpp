// setcover — drop-in replacement of the reference tool (tools/setcover.cpp:112-147): same command
// line, same cluster file format in and out, same progress lines on stdout.  ReadClusters /
// WriteClusters follow tools/Parsers.cpp:23-170; SetCover itself runs on the GPU through
// include/defuse_sc.h (no CPU fallback: without a HIP device the tool exits 1).
#include "../include/defuse_dsa.h"
#include "../include/defuse_sc.h"
#include "defuse_host.hpp"

using namespace defuse;

namespace {

struct ClusterLine { int clusterID, clusterEnd, fragmentIndex; };

// ClusterMembership line (tools/Parsers.cpp:36-78): at least three tab-separated fields, the first three integers
void parse_cluster_line(const char* line, size_t len, int lineNumber, const std::string& filename, ClusterLine& out)
{
    if (len == 0) die("Error: Empty clusters line " + std::to_string(lineNumber) + " of " + filename);
    const char* end = line + len;
    const char* t1 = (const char*)memchr(line, '\t', len);
    const char* t2 = t1 ? (const char*)memchr(t1 + 1, '\t', (size_t)(end - t1 - 1)) : nullptr;
    if (!t2) die("Error: Format error for clusters line " + std::to_string(lineNumber) + " of " + filename);
    const char* t3 = (const char*)memchr(t2 + 1, '\t', (size_t)(end - t2 - 1));
    if (!t3) t3 = end;
    if (!field_int(line, (size_t)(t1 - line), out.clusterID) || !field_int(t1 + 1, (size_t)(t2 - t1 - 1), out.clusterEnd) ||
        !field_int(t2 + 1, (size_t)(t3 - t2 - 1), out.fragmentIndex)) {
        std::cerr << "Failed to interpret line:" << std::endl << std::string(line, len) << std::endl;
        std::exit(1);
    }
}

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

    std::cout << "Reading clusters" << std::endl;
    // ReadClusters (tools/Parsers.cpp:23-84): cluster-end-0 lines only, clusters[id] in file order
    std::vector<std::vector<int>> clusters;
    {
        FILE* in = fopen(inName.c_str(), "rb");
        if (!in) die("Error: unable to read from clusters file " + inName);
        LineReader reader(in);
        const char* line;
        size_t len;
        int lineNumber = 0;
        ClusterLine cl;
        while (reader.next(line, len)) {
            parse_cluster_line(line, len, ++lineNumber, inName, cl);
            if (cl.clusterEnd != 0) continue;
            if (cl.clusterID < 0) die("Error: Invalid cluster ID for line " + std::to_string(lineNumber) + " of " + inName);
            if ((int)clusters.size() < cl.clusterID + 1) clusters.resize(cl.clusterID + 1);
            clusters[cl.clusterID].push_back(cl.fragmentIndex);
        }
        fclose(in);
    }

    std::cout << "Calculating set cover solution" << std::endl;
    int maxElement = -1;                                  // FindMaxElement (tools/Common.cpp:71-89)
    std::vector<int64_t> off(clusters.size() + 1, 0);
    std::vector<int32_t> elements;
    for (size_t c = 0; c < clusters.size(); ++c) {
        for (int e : clusters[c]) {
            if (e < 0) die("Error: negative elements not permitted");
            maxElement = std::max(maxElement, e);
            elements.push_back(e);
        }
        off[c + 1] = (int64_t)elements.size();
    }
    std::vector<int32_t> owner((size_t)maxElement + 1, -1);
    if (!elements.empty()) {
        sc_timing t;
        const int rc = sc_cover(dsa_pick_device(), off.data(), elements.data(), (int32_t)clusters.size(), maxElement,
                                owner.data(), &t);
        if (rc != 0) die(std::string("Error: set cover on the GPU failed: ") + sc_last_error());
        if (std::getenv("DEFUSE_TIMING"))
            std::cerr << "[setcover] components " << t.n_components << " (large " << t.n_large << "), build " << t.build_ms
                      << " ms, components " << t.components_ms << " ms, greedy " << t.greedy_ms << " ms" << std::endl;
    }
    std::vector<int64_t> solutionSize(clusters.size(), 0);
    for (int32_t o : owner)
        if (o >= 0) ++solutionSize[o];

    std::cout << "Writing out clusters" << std::endl;
    // WriteClusters (tools/Parsers.cpp:86-170): copy the input lines (both ends) whose fragment was
    // assigned to the line's cluster, for clusters that kept at least minClusterSize fragments
    std::ofstream out(outName.c_str());
    if (!out) die("Error: unable to write to clusters file " + outName);
    FILE* in = fopen(inName.c_str(), "rb");
    if (!in) die("Error: unable to read from clusters file " + inName);
    LineReader reader(in);
    const char* line;
    size_t len;
    int lineNumber = 0;
    ClusterLine cl;
    std::string buf;
    buf.reserve((1u << 22) + 4096);
    while (reader.next(line, len)) {
        parse_cluster_line(line, len, ++lineNumber, outName, cl);
        if (cl.clusterID < 0) die("Error: Invalid cluster ID for line " + std::to_string(lineNumber) + " of " + outName);
        if ((size_t)cl.clusterID >= clusters.size()) continue;            // an id that only occurs with end 1 (UB in the reference)
        if ((int64_t)solutionSize[cl.clusterID] < (int64_t)minClusterSize) continue;
        if (cl.fragmentIndex >= 0 && cl.fragmentIndex <= maxElement && owner[cl.fragmentIndex] == cl.clusterID) {
            buf.append(line, len);
            buf += '\n';
            if (buf.size() > (1u << 22)) { out.write(buf.data(), (std::streamsize)buf.size()); buf.clear(); }
        }
    }
    out.write(buf.data(), (std::streamsize)buf.size());
    fclose(in);
    out.close();
    return out.good() ? 0 : 1;
}
