// dosplitalign — drop-in replacement of the reference tool (tools/dosplitalign.cpp:25-111): same
// command line, same input formats, same output lines; the per-candidate
// SplitAlignmentTask::Align of tools/SplitAlignment.cpp:294 is replaced by one batched call into the
// MI355X library (include/defuse_dsa.h).  There is no CPU fallback: without a GPU the tool exits 1.
//
// Output order: one SAM record at a time, its overlapping cluster ends in ascending (signed) id order
// — the canonical order of SURVEY.md 8(c) for the reference's unordered_set iteration (the pipeline
// sorts the file by fusion id afterwards, scripts/defuse_run.pl:528).
//
// Environment: DEFUSE_GPU=<ordinal> selects the device (default: pid mod device count, so the processes the
// pipeline runs side by side spread over a node's GPUs; with HIP_VISIBLE_DEVICES the
// ordinal is relative to the visible set).
#include <chrono>
#include <numeric>

#include "../include/defuse_dsa.h"
#include "defuse_host.hpp"

using namespace defuse;

int main(int argc, char* argv[])
{
    CmdLine cmd("Fusion sequence prediction by split reads");
    cmd.add("f", "fasta", "Reference Fasta", "string");
    cmd.add("e", "exons", "Exon Regions Filename", "string");
    cmd.add("u", "ufrag", "Fragment Length Mean", "float");
    cmd.add("s", "sfrag", "Fragment Length Standard Deviation", "float");
    cmd.add("n", "minread", "Minimum Read Length", "integer");
    cmd.add("x", "maxread", "Maximum Read Length", "integer");
    cmd.add("r", "regions", "Fusion Regions Filename", "string");
    cmd.add("i", "improper", "Improper Alignments Sam Filename", "string");
    cmd.add("1", "seq1", "End 1 Sequences", "string");
    cmd.add("2", "seq2", "End 2 Sequences", "string");
    cmd.add("a", "align", "Split Alignments Filename", "string");
    cmd.parse(argc, argv);
    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_stage = now(), t_gpu = 0.0, t_write = 0.0;
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[dosplitalign] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t;
    };

    const std::map<int, std::vector<Location>> regions = ReadAlignRegionPairs(cmd.str("regions"));
    std::map<int, SplitAlignmentTask> tasks = CreateTasks(cmd.str("fasta"), cmd.str("exons"), cmd.real("ufrag"), cmd.real("sfrag"),
                                                         cmd.integer("minread"), cmd.integer("maxread"), regions);

    // SplitReadRealigner::AddTask (tools/SplitAlignment.cpp:236-251): 2000 bp bins over the mate regions
    BinnedLocations binned(2000);
    for (const auto& kv : tasks)
        for (int ce = 0; ce <= 1; ++ce)
            for (const Location& loc : kv.second.mMateRegions[ce]) binned.Add(pack_id(kv.first, ce), loc);

    stage("regions + windows");
    ReadStore reads;
    if (!AddReads(cmd.str("seq1"), reads) || !AddReads(cmd.str("seq2"), reads)) {
        std::cout << "Error: unable to read sequences" << std::endl;
        return 1;
    }

    stage("reads");
    // the GPU batch: one dsa_fusion per task that gets at least one candidate
    std::vector<uint8_t> ref_bytes, read_bytes;
    std::vector<dsa_fusion> fusions;
    std::vector<dsa_pair> cand;                    // candidates in the reference's visiting order
    std::unordered_map<int, int> fusion_index;     // fusion id -> index into fusions
    std::unordered_set<uint64_t> candidate_unique;   // (fusion, read id, revComp) seen (:268, :292)

    std::ofstream out(cmd.str("align").c_str());
    if (!out.good()) die("Error: Unable to open " + cmd.str("align"));

    // Candidates go to the GPU in batches (DEFUSE_DSA_BATCH_PAIRS, default 4 M) and their lines are
    // written in the reference's visiting order, so a run of any size streams through.
    dsa_ctx* ctx = nullptr;
    size_t batch_pairs = (size_t)4 << 20;
    if (const char* e = std::getenv("DEFUSE_DSA_BATCH_PAIRS")) batch_pairs = std::max<size_t>(1, (size_t)std::atoll(e));
    auto flush = [&]() {
        if (cand.empty()) return;
        // group by fusion for the kernels, fusions with many candidates first: the table-driven kernels take
        // workgroups (256 consecutive pairs) of at most four fusions, so the small fusions are kept together
        // at the end instead of dragging their big neighbours onto the generic path (stable: keeps the
        // visiting order inside a fusion)
        std::vector<int64_t> per_fusion(fusions.size(), 0);
        for (const dsa_pair& c : cand) ++per_fusion[c.fusion_idx];
        std::vector<int64_t> order(cand.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) {
            const int fa = cand[a].fusion_idx, fb = cand[b].fusion_idx;
            if (per_fusion[fa] != per_fusion[fb]) return per_fusion[fa] > per_fusion[fb];
            return fa < fb;
        });
        std::vector<dsa_pair> pairs(cand.size());
        for (size_t k = 0; k < order.size(); ++k) pairs[k] = cand[order[k]];

        if (!ctx) {
            if (dsa_create(&ctx, dsa_pick_device()) != DSA_OK) die("Error: no usable MI355X/HIP device (dsa_create failed)");
        }
        std::vector<dsa_record> recs(std::max<size_t>(1024, 2 * pairs.size()));
        int64_t n = 0;
        const double t_g0 = now();
        int rc = dsa_align_batch(ctx, ref_bytes.data(), (int64_t)ref_bytes.size(), fusions.data(), (int32_t)fusions.size(),
                                 read_bytes.data(), (int64_t)read_bytes.size(), pairs.data(), (int64_t)pairs.size(), recs.data(),
                                 (int64_t)recs.size(), &n);
        if (rc == DSA_E_CAPACITY) {
            recs.resize((size_t)n);
            rc = dsa_align_batch(ctx, ref_bytes.data(), (int64_t)ref_bytes.size(), fusions.data(), (int32_t)fusions.size(),
                                 read_bytes.data(), (int64_t)read_bytes.size(), pairs.data(), (int64_t)pairs.size(),
                                 recs.data(), (int64_t)recs.size(), &n);
        }
        if (rc != DSA_OK) die(std::string("Error: split alignment on the GPU failed: ") + dsa_last_error(ctx));
        const double t_g1 = now();
        t_gpu += t_g1 - t_g0;

        // back to the visiting order: records arrive grouped by batch pair index
        std::vector<int64_t> first(pairs.size() + 1, 0);
        for (int64_t k = 0; k < n; ++k) ++first[recs[k].pair_idx + 1];
        for (size_t k = 0; k < pairs.size(); ++k) first[k + 1] += first[k];
        std::vector<int64_t> slot_of(cand.size());
        for (size_t k = 0; k < order.size(); ++k) slot_of[order[k]] = (int64_t)k;
        std::string buf;
        for (size_t c = 0; c < cand.size(); ++c) {
            const int64_t k = slot_of[c];
            for (int64_t r = first[k]; r < first[k + 1]; ++r) {
                const dsa_record& a = recs[r];
                // SplitAlignment::WriteAlignment (tools/SplitAlignment.cpp:305-317): nine fields, each followed by a tab
                for (int v : {a.fusion_id, a.frag, a.read_end, a.revcomp, a.ref_first, a.ref_second, a.read_first, a.read_second, a.score}) {
                    append_int(buf, v);
                    buf += '\t';
                }
                buf += '\n';
            }
            if (buf.size() > (1u << 20)) { out << buf; buf.clear(); }
        }
        out << buf;
        t_write += now() - t_g1;
        ref_bytes.clear();
        read_bytes.clear();
        fusions.clear();
        cand.clear();
        fusion_index.clear();
    };

    // SplitReadRealigner::DoAlignment (tools/SplitAlignment.cpp:266-303).  The SAM text is mapped and taken in rounds of
    // 256 MiB, each cut into one piece per host thread: the pieces parse their records and look up the mate regions they
    // overlap side by side; the candidates are then taken up in file order by one thread (the seen-set, the batches and the
    // output are sequential by nature).
    MappedText sam;
    sam.load(cmd.str("improper"), "Error: Unable to open sam file ");
    unsigned nThreads = host_threads();
    if (sam.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nThreads = 1;
    struct Hit { int frag, readEnd; uint32_t first, count; };           // readEnd -1: inherited from before the piece
    struct SamPiece {
        std::vector<Hit> hits;
        std::vector<int> ids;
        size_t lines = 0, errorLine = 0;
        int errorKind = 0, lastReadEnd = -2;                             // -2: no record of the piece set the read end
        std::string errorText;
    };
    std::vector<SamPiece> pieces(nThreads);
    std::string seq;
    size_t lineBase = 0;
    int carryReadEnd = 0;                                                // the reference's RawAlignment starts with read end 0
    for (size_t lo = 0; lo < sam.size();) {
        size_t hi = std::min(sam.size(), lo + ((size_t)1 << 28));
        if (hi < sam.size()) hi = sam.line_end(hi - 1);
        const std::vector<size_t> cut = sam.cut_lines(lo, hi, nThreads);
        run_threads(nThreads, [&](unsigned t) {
            SamPiece& pc = pieces[t];
            pc = SamPiece();
            std::vector<int> overlapping;
            std::string reference;
            SamFields f;
            int readEnd = -1;
            for (size_t pos = cut[t]; pos < cut[t + 1];) {
                const size_t e = sam.line_end(pos);
                const char* line = sam.data() + pos;
                const size_t len = (e > pos && sam[e - 1] == '\n') ? e - 1 - pos : e - pos;
                pos = e;
                ++pc.lines;
                const int kind = ParseSamLine(line, len, f, readEnd);
                if (kind == 1) continue;
                if (kind) { pc.errorLine = pc.lines; pc.errorKind = kind; return; }
                reference.assign(f.reference, f.reference_len);
                binned.Overlapping(reference, f.strand, f.region, overlapping);   // ascending signed order: end-1 ids (negative) first
                if (overlapping.empty()) continue;
                Hit h;
                if (!field_int(f.fragment, f.fragment_len, h.frag)) {
                    pc.errorLine = pc.lines; pc.errorKind = 6; pc.errorText.assign(f.fragment, f.fragment_len);
                    return;
                }
                h.readEnd = readEnd;
                h.first = (uint32_t)pc.ids.size();
                h.count = (uint32_t)overlapping.size();
                pc.ids.insert(pc.ids.end(), overlapping.begin(), overlapping.end());
                pc.hits.push_back(h);
            }
            if (readEnd >= 0) pc.lastReadEnd = readEnd;
        });
        stage("  sam records + overlaps of a round");
        for (SamPiece& pc : pieces) {
            for (const Hit& h : pc.hits) {
                const int mateReadEnd = h.readEnd < 0 ? carryReadEnd : h.readEnd;
                const int frag = h.frag;
                for (uint32_t k = 0; k < h.count; ++k) {
                    const int cid = pc.ids[h.first + k];
                    const int cluster_end = cid < 0 ? 1 : 0;
                    const int fusion_id = cid & 0x7FFFFFFF;
                    const int read_end = (mateReadEnd == 0) ? 1 : 0;
                    const int revcomp = (cluster_end == 0) ? 1 : 0;
                    const int rid = pack_id(frag, read_end);
                    if (!candidate_unique.insert(((uint64_t)(uint32_t)fusion_id << 33) | ((uint64_t)(uint32_t)rid << 1) | (uint64_t)revcomp).second) continue;
                    const char* rs = nullptr;              // a missing read aligns as the empty string (operator[] in the reference, :286)
                    size_t rn = 0;
                    if (reads.get(frag, read_end, rs, rn)) seq.assign(rs, rn); else seq.clear();
                    if (revcomp) ReverseComplement(seq);
                    auto fi = fusion_index.find(fusion_id);
                    if (fi == fusion_index.end()) {
                        const SplitAlignmentTask& t = tasks[fusion_id];
                        dsa_fusion f;
                        f.fusion_id = fusion_id;
                        f.ref0_off = (int32_t)ref_bytes.size();
                        f.ref0_len = (int32_t)t.mSplitAlignSeq[0].size();
                        ref_bytes.insert(ref_bytes.end(), t.mSplitAlignSeq[0].begin(), t.mSplitAlignSeq[0].end());
                        f.ref1_off = (int32_t)ref_bytes.size();
                        f.ref1_len = (int32_t)t.mSplitAlignSeq[1].size();
                        ref_bytes.insert(ref_bytes.end(), t.mSplitAlignSeq[1].begin(), t.mSplitAlignSeq[1].end());
                        fi = fusion_index.emplace(fusion_id, (int)fusions.size()).first;
                        fusions.push_back(f);
                    }
                    dsa_pair p{};
                    p.fusion_idx = fi->second;
                    p.read_off = (int32_t)read_bytes.size();
                    p.read_len = (int32_t)seq.size();
                    p.frag = frag;
                    p.read_end = (uint8_t)read_end;
                    p.revcomp = (uint8_t)revcomp;
                    read_bytes.insert(read_bytes.end(), seq.begin(), seq.end());
                    cand.push_back(p);
                }
                // between two SAM records: a batch never splits the candidates of one record
                if (cand.size() >= batch_pairs || read_bytes.size() > ((size_t)1 << 30) || ref_bytes.size() > ((size_t)1 << 30)) flush();
            }
            if (pc.errorLine) {                                         // the records before the bad line were taken up, as a serial reader does
                flush();
                if (pc.errorKind == 6) die("Error: bad integer '" + pc.errorText + "' as fragment name");
                DieSamLine(pc.errorKind, lineBase + pc.errorLine);
            }
            lineBase += pc.lines;
            if (pc.lastReadEnd != -2) carryReadEnd = pc.lastReadEnd;
        }
        lo = hi;
    }
    flush();
    stage("candidates + alignment + output");
    if (timing) std::cerr << "[dosplitalign] of which GPU calls " << t_gpu << " s, formatting and writing " << t_write << " s" << std::endl;
    if (ctx) dsa_destroy(ctx);
    out.close();
    if (!out.good()) die("Error: failed writing " + cmd.str("align"));
    return 0;
}
