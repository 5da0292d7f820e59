// setcover — drop-in replacement of the reference tool (tools/setcover.cpp:112-147): same command
// line, same cluster file format in and out, same progress lines on stdout.  ReadClusters /
// WriteClusters follow tools/Parsers.cpp:23-170; SetCover itself runs on the GPU through
// include/defuse_sc.h (no CPU fallback: without a HIP device the tool exits 1).
#include "../include/defuse_dsa.h"
#include "../include/defuse_sc.h"
#include "defuse_host.hpp"

#include <chrono>
#include <mutex>
#include <thread>

using namespace defuse;

namespace {

struct ClusterLine { int clusterID, clusterEnd, fragmentIndex; };

}  // namespace

int main(int argc, char* argv[])
{
    keep_freed_memory();
    CmdLine cmd("Set cover for maximum parsimony");
    cmd.add("c", "clusters", "Clusters Filename", "string");
    cmd.add("m", "minclustersize", "Minimum Cluster Size", "integer");
    cmd.add("o", "outclust", "Output Clusters Filename", "string");
    cmd.parse(argc, argv);
    const std::string inName = cmd.str("clusters"), outName = cmd.str("outclust");
    const int minClusterSize = cmd.integer("minclustersize");

    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_stage = now(), t_part = t_stage;
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[setcover] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t_part = t;
    };
    auto part = [&](const char* name) {                  // a part of the running stage
        const double t = now();
        if (timing) std::cerr << "[setcover]   " << name << " " << (t - t_part) << " s" << std::endl;
        t_part = t;
    };
    // The device comes up beside the parsing (a fresh process spends a few tenths of a second in the runtime's start); what it
    // finds — no device, say — is sc_cover's to report, at the place the tool has always reported it.  An error exit first
    // lets that thread leave the runtime, then ends the process without running exit handlers beside it.
    std::thread warm([] { (void)sc_prepare(dsa_pick_device()); });
    struct Joiner {                                      // (an exception that unwinds main must not meet a joinable thread)
        std::thread& th;
        ~Joiner() { if (th.joinable()) th.join(); }
    } warm_joiner{warm};
    static std::mutex die_mutex;
    die_hook() = [&] {
        die_mutex.lock();
        if (warm.joinable()) warm.join();
        std::cout.flush();
        std::cerr.flush();
        fflush(nullptr);
        _exit(1);
    };
    std::cout << "Reading clusters" << std::endl;
    // ReadClusters (tools/Parsers.cpp:23-84): cluster-end-0 lines only, clusters[id] in file order.  The file is mapped and
    // parsed in one piece per host thread; the pieces' (cluster, fragment) lists are laid out cluster by cluster in file order.
    MappedText text;
    text.load(inName, "Error: unable to read from clusters file ");
    part("file mapped");
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
    // The common line — three plain integers of at most nine digits, tab separated — in one pass; anything else goes through
    // parse_line, which knows the reference's error cases.  Returns the end of the line (one past its newline).
    auto parse_fast = [&](size_t pos, size_t limit, ClusterLine& out, size_t& len, int& kind) -> size_t {
        const char* p = text.data();
        size_t k = pos;
        int v[3];
        bool plain = true;
        for (int f = 0; f < 3 && plain; ++f) {
            bool neg = false;
            if (k < limit && (p[k] == '-' || p[k] == '+')) { neg = p[k] == '-'; ++k; }
            const size_t d0 = k;
            int x = 0;
            while (k < limit && (unsigned)(p[k] - '0') < 10u) x = x * 10 + (p[k++] - '0');
            if (k == d0 || k - d0 > 9) { plain = false; break; }
            v[f] = neg ? -x : x;
            if (f < 2) {
                if (k < limit && p[k] == '\t') ++k;
                else plain = false;
            } else if (k < limit && p[k] != '\t' && p[k] != '\n') plain = false;
        }
        const char* nl = (const char*)memchr(p + (plain ? k : pos), '\n', limit - (plain ? k : pos));
        const size_t e = nl ? (size_t)(nl - p) + 1 : limit;
        len = (e > pos && p[e - 1] == '\n') ? e - 1 - pos : e - pos;
        if (plain) { out.clusterID = v[0]; out.clusterEnd = v[1]; out.fragmentIndex = v[2]; kind = 0; }
        else kind = parse_line(p + pos, len, out);
        return e;
    };
    auto report = [&](int kind, size_t lineNumber, const std::string& lineText, const std::string& filename) {
        if (kind == 1) die("Error: Empty clusters line " + std::to_string(lineNumber) + " of " + filename);
        if (kind == 2) die("Error: Format error for clusters line " + std::to_string(lineNumber) + " of " + filename);
        if (kind == 3) {
            std::cerr << "Failed to interpret line:" << std::endl << lineText << std::endl;
            if (die_hook()) die_hook()();
            std::exit(1);
        }
        die("Error: Invalid cluster ID for line " + std::to_string(lineNumber) + " of " + filename);
    };
    run_threads(nThreads, [&](unsigned t) {
        Piece& pc = pieces[t];
        pc.members.reserve((cut[t + 1] - cut[t]) / 60 + 16);
        ClusterLine cl;
        for (size_t pos = cut[t]; pos < cut[t + 1];) {
            const char* line = text.data() + pos;
            size_t len;
            int kind;
            pos = parse_fast(pos, cut[t + 1], cl, len, kind);
            ++pc.lines;
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
    part("lines parsed");
    int maxCluster = -1;
    for (const Piece& pc : pieces) maxCluster = std::max(maxCluster, pc.maxCluster);
    const size_t nClusters = (size_t)(maxCluster + 1);
    stage("read");

    std::cout << "Calculating set cover solution" << std::endl;
    int maxElement = -1;                                  // FindMaxElement (tools/Common.cpp:71-89)
    // clusters[id] in file order: piece t counts its members per cluster over the range of ids it has seen (a file written
    // cluster by cluster gives the pieces ranges that barely overlap), the counts of the pieces in front of it are where its
    // members of a cluster begin, and every piece places its own.
    std::vector<int64_t> off(nClusters + 1, 0);
    std::vector<int32_t> elements;
    {
        struct Range { int lo = 0, hi = -1; std::vector<int64_t> count; int maxElement = -1; bool negative = false; };
        std::vector<Range> range(nThreads);
        run_threads(nThreads, [&](unsigned t) {
            Range& r = range[t];
            const Piece& pc = pieces[t];
            if (pc.members.empty()) return;
            int lo = pc.members[0].first, hi = lo;
            for (const auto& m : pc.members) { lo = std::min(lo, m.first); hi = std::max(hi, m.first); }
            r.lo = lo; r.hi = hi;
            r.count.assign((size_t)(hi - lo) + 1, 0);
            for (const auto& m : pc.members) {
                ++r.count[(size_t)(m.first - lo)];
                if (m.second < 0) r.negative = true;
                r.maxElement = std::max(r.maxElement, m.second);
            }
        });
        for (const Range& r : range) {
            if (r.negative) die("Error: negative elements not permitted");
            maxElement = std::max(maxElement, r.maxElement);
        }
        for (Range& r : range)                                // count -> members of the cluster in the pieces in front
            for (int c = r.lo; c <= r.hi; ++c) {
                const int64_t mine = r.count[(size_t)(c - r.lo)];
                r.count[(size_t)(c - r.lo)] = off[(size_t)c + 1];
                off[(size_t)c + 1] += mine;
            }
        for (size_t c = 0; c < nClusters; ++c) off[c + 1] += off[c];
        elements.resize((size_t)off[nClusters]);
        run_threads(nThreads, [&](unsigned t) {
            Range& r = range[t];
            Piece& pc = pieces[t];
            for (const auto& m : pc.members) elements[(size_t)(off[m.first] + r.count[(size_t)(m.first - r.lo)]++)] = m.second;
            std::vector<std::pair<int, int>>().swap(pc.members);
        });
    }
    part("clusters laid out");
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
    run_threads(nThreads, [&](unsigned t) {               // every thread a stretch of the fragments; few of them meet at a cluster
        const size_t lo = owner.size() / nThreads * t, hi = t + 1 == nThreads ? owner.size() : owner.size() / nThreads * (t + 1);
        for (size_t e = lo; e < hi; ++e)
            if (owner[e] >= 0) __atomic_fetch_add(&solutionSize[(size_t)owner[e]], (int64_t)1, __ATOMIC_RELAXED);
    });
    part("cluster sizes");

    std::cout << "Writing out clusters" << std::endl;
    // WriteClusters (tools/Parsers.cpp:86-170): copy the input lines (both ends) whose fragment was
    // assigned to the line's cluster, for clusters that kept at least minClusterSize fragments
    OrderedFileWriter out;
    if (!out.open_file(outName)) die("Error: unable to write to clusters file " + outName);
    // the lines are filtered in rounds of 256 MiB of input, each cut into one piece per host thread; texts written in order
    std::vector<std::string> texts(nThreads);
    std::vector<Piece> round(nThreads);
    size_t lineBase = 0;
    double tFilter = 0.0, tWrite = 0.0;
    for (size_t lo = 0; lo < text.size();) {
        size_t hi = std::min(text.size(), lo + ((size_t)1 << 28));
        if (hi < text.size()) hi = text.line_end(hi - 1);
        const std::vector<size_t> rc = text.cut_lines(lo, hi, nThreads);
        const double t0 = now();
        run_threads(nThreads, [&](unsigned t) {
            Piece& pc = round[t];
            pc = Piece();
            std::string& buf = texts[t];
            buf.clear();
            buf.reserve((rc[t + 1] - rc[t]) * 3 / 4 + 4096);
            // lines in batches: the owner of a line's fragment is a random word of a table far larger than the caches, so the
            // batch's words are asked for while its lines are parsed and looked at afterwards
            constexpr int BATCH = 32;
            struct Parsed { const char* line; size_t len; ClusterLine cl; };
            Parsed batch[BATCH];
            for (size_t pos = rc[t]; pos < rc[t + 1];) {
                int nb = 0;
                for (; nb < BATCH && pos < rc[t + 1]; ++nb) {
                    Parsed& pl = batch[nb];
                    pl.line = text.data() + pos;
                    int kind;
                    pos = parse_fast(pos, rc[t + 1], pl.cl, pl.len, kind);
                    ++pc.lines;
                    if (!kind && pl.cl.clusterID < 0) kind = 4;
                    if (kind) { pc.errorLine = pc.lines; pc.errorKind = kind; pc.errorText.assign(pl.line, pl.len); return; }
                    if ((size_t)pl.cl.clusterID < nClusters) __builtin_prefetch(&solutionSize[pl.cl.clusterID]);
                    if (pl.cl.fragmentIndex >= 0 && pl.cl.fragmentIndex <= maxElement) __builtin_prefetch(&owner[pl.cl.fragmentIndex]);
                }
                for (int k = 0; k < nb; ++k) {
                    const ClusterLine& cl = batch[k].cl;
                    if ((size_t)cl.clusterID >= nClusters) continue;            // an id that only occurs with end 1 (UB in the reference)
                    if ((int64_t)solutionSize[cl.clusterID] < (int64_t)minClusterSize) continue;
                    if (cl.fragmentIndex >= 0 && cl.fragmentIndex <= maxElement && owner[cl.fragmentIndex] == cl.clusterID) {
                        buf.append(batch[k].line, batch[k].len);
                        buf += '\n';
                    }
                }
            }
        });
        for (unsigned t = 0; t < nThreads; ++t) {
            if (round[t].errorLine) report(round[t].errorKind, lineBase + round[t].errorLine, round[t].errorText, outName);
            lineBase += round[t].lines;
        }
        const double t1 = now();
        out.write_round_async(std::move(texts), nThreads);              // copied into the file while the next round is filtered
        texts = std::vector<std::string>(nThreads);
        tFilter += t1 - t0;
        tWrite += now() - t1;
        lo = hi;
    }
    {
        const double t1 = now();
        out.wait_async();
        tWrite += now() - t1;
    }
    if (timing) std::cerr << "[setcover]   lines filtered " << tFilter << " s, waited for the writer " << tWrite << " s" << std::endl;
    stage("write");
    warm.join();
    // the process ends here: unmapping the input and the runtime's own shutdown cost tenths of a second that produce nothing
    const int rc = out.close_file() ? 0 : 1;
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    if (std::getenv("DEFUSE_FULL_EXIT")) exit(rc);     // under a profiler that writes its files at exit (rocprofv3): atexit handlers run
    _exit(rc);
}
