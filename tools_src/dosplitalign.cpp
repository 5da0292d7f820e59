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
#include <mutex>
#include <numeric>

#include <dlfcn.h>

#include "../include/defuse_dsa.h"
#include "evaluate.hpp"

using namespace defuse;

namespace {

// The C-ABI library is opened at run time (as a JNI / ctypes / cgo binding of the same ABI would) instead of being a link
// dependency: loading the HIP runtime and registering the code objects takes a tenth of a second, which now passes on a
// helper thread while the main thread reads the text inputs.  $DEFUSE_DSA_LIB names the library, default
// <directory of this binary>/../defuse_amd/libdefuse_dsa.so.
struct DsaLib {
    decltype(&dsa_create) create = nullptr;
    decltype(&dsa_destroy) destroy = nullptr;
    decltype(&dsa_pick_device) pick_device = nullptr;
    decltype(&dsa_align_batch) align_batch = nullptr;
    decltype(&dsa_last_error) last_error = nullptr;
    std::string error;
    bool load()
    {
        std::string path;
        if (const char* e = std::getenv("DEFUSE_DSA_LIB")) path = e;
        else {
            char exe[4096];
            const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
            if (n <= 0) { error = "cannot resolve /proc/self/exe"; return false; }
            exe[n] = 0;
            path = exe;
            path = path.substr(0, path.find_last_of('/')) + "/../defuse_amd/libdefuse_dsa.so";
        }
        void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { error = dlerror(); return false; }
        create = (decltype(create))dlsym(h, "dsa_create");
        destroy = (decltype(destroy))dlsym(h, "dsa_destroy");
        pick_device = (decltype(pick_device))dlsym(h, "dsa_pick_device");
        align_batch = (decltype(align_batch))dlsym(h, "dsa_align_batch");
        last_error = (decltype(last_error))dlsym(h, "dsa_last_error");
        if (!create || !destroy || !pick_device || !align_batch || !last_error) { error = "missing symbols in " + path; return false; }
        return true;
    }
};

}  // namespace

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
    // The command line is the reference's, argument for argument (tools/dosplitalign.cpp:43-56): help, usage and error texts
    // are compared with TCLAP's byte for byte (tests/test_cli_ref.py).
    // Fused mode, only with DEFUSE_FUSED=1 in the environment — without it these five options do not
    // exist: one process goes from the set-cover clusters to the breakpoint predictions and still writes every intermediate
    // file the pipeline's separate steps would (scripts/defuse_run.pl:506-533: get_align_regions.pl -> dosplitalign ->
    // sort -n -k 1 -> evalsplitalign).
    const bool fused_cli = [] { const char* e = std::getenv("DEFUSE_FUSED"); return e && std::atoi(e) != 0; }();
    if (fused_cli) {
        cmd.add_optional("c", "clusters", "Fused mode: clusters file (setcover / remove_duplicates output); the regions file named by -r is WRITTEN "
                         "from it by the rule of get_align_regions.pl", "string", "");
        cmd.add_switch("", "sorted", "Fused mode: write the alignments in the order of `LC_ALL=C sort -n -k 1`");
        cmd.add_optional("q", "seq", "Fused mode: Sequence Predictions Filename (evalsplitalign -q)", "string", "");
        cmd.add_optional("b", "break", "Fused mode: Breakpoint Predictions Filename (evalsplitalign -b)", "string", "");
        cmd.add_optional("p", "predalign", "Fused mode: Predicted Alignments Filename (evalsplitalign -p)", "string", "");
    }
    cmd.parse(argc, argv);
    const std::string opt_clusters = fused_cli ? cmd.str("clusters") : std::string(), opt_seq = fused_cli ? cmd.str("seq") : std::string(),
                      opt_break = fused_cli ? cmd.str("break") : std::string(), opt_predalign = fused_cli ? cmd.str("predalign") : std::string();
    const bool fused_eval = !opt_seq.empty() || !opt_break.empty() || !opt_predalign.empty();
    if (fused_eval && (opt_seq.empty() || opt_break.empty() || opt_predalign.empty()))
        die("Error: the fused mode needs all of --seq, --break and --predalign");
    const bool collect = fused_eval || (fused_cli && cmd.is_set("sorted"));
    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_main = now();
    double t_stage = now(), t_gpu = 0.0, t_write = 0.0;
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[dosplitalign] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t;
    };

    // The HIP runtime and the context come up on a helper thread (a tenth of a second that would otherwise sit in front of
    // the first batch) — started when the first SAM record that overlaps a mate region has been seen, i.e. after every cheap
    // check of the inputs (command line, regions, exons, FASTA index, FASTQ files) and only in a run that will have a
    // candidate to align: a run without one never loads the library, let alone touches a GPU.
    dsa_ctx* ctx = nullptr;
    int ctx_rc = DSA_OK;
    DsaLib dsa;
    std::thread ctx_thread;
    std::once_flag ctx_once;
    auto start_ctx = [&] {
        std::call_once(ctx_once, [&] { ctx_thread = std::thread([&] { ctx_rc = dsa.load() ? dsa.create(&ctx, dsa.pick_device()) : DSA_E_DEVICE; }); });
    };
    // An error exit (die(), on whichever thread) first lets the helper finish what it is in the middle of — dlopen, the
    // runtime's start — and then ends the process the way the success path does, without running exit handlers beside
    // threads that are still alive.  Only one thread gets that far; any other that fails meanwhile waits behind it.
    static std::mutex die_mutex;
    die_hook() = [&] {
        die_mutex.lock();
        if (ctx_thread.joinable() && std::this_thread::get_id() != ctx_thread.get_id()) ctx_thread.join();
        std::cout.flush();
        std::cerr.flush();
        fflush(nullptr);
        _exit(1);
    };

    if (!opt_clusters.empty()) {
        // scripts/get_align_regions.pl:14-53 (as bin/defuse_glue get_align_regions): per cluster end the reference, the strand
        // and the span of its alignments; clusters ascending, end 0 then 1
        struct EndInfo { std::string ref, strand; int start = 0, end = 0; bool have = false; };
        std::map<int, std::map<int, EndInfo>> clusters;
        MappedText ctext;
        ctext.load(opt_clusters, "Error: Unable to open clusters file ");
        for (size_t pos = 0; pos < ctext.size();) {
            const size_t e = ctext.line_end(pos);
            const size_t len = (e > pos && ctext[e - 1] == '\n') ? e - 1 - pos : e - pos;
            const std::vector<std::string> f = split_tabs(std::string(ctext.data() + pos, len));
            pos = e;
            if (f.size() < 8) die("Error: cluster line with fewer than 8 fields");
            const int id = lexical_int_or_die(f[0], "as cluster id"), ce = lexical_int_or_die(f[1], "as cluster end");
            const int start = lexical_int_or_die(f[6], "as start"), end = lexical_int_or_die(f[7], "as end");
            EndInfo& ei = clusters[id][ce];
            ei.ref = f[4];
            ei.strand = f[5];
            if (!ei.have) { ei.start = start; ei.end = end; ei.have = true; }
            ei.start = std::min(ei.start, start);
            ei.end = std::max(ei.end, end);
        }
        std::string text;
        for (const auto& c : clusters) {
            if (c.second.size() != 2) die("Error: Did not find 2 ends for cluster " + std::to_string(c.first));
            for (const auto& en : c.second) {
                append_int(text, c.first); text += '\t';
                append_int(text, en.first); text += '\t';
                text += en.second.ref; text += '\t';
                text += en.second.strand; text += '\t';
                append_int(text, en.second.start); text += '\t';
                append_int(text, en.second.end); text += '\n';
            }
        }
        OrderedFileWriter rf;
        if (!rf.open_file(cmd.str("regions"))) die("Error: unable to write " + cmd.str("regions"));
        rf.write_round({text}, 1);
        if (!rf.close_file()) die("Error: failed writing " + cmd.str("regions"));
        stage("clusters -> regions file");
    }
    const std::map<int, std::vector<Location>> regions = ReadAlignRegionPairs(cmd.str("regions"));
    stage("regions");
    // The two FASTQ files are read on threads of their own while the main thread sets the tasks up (FASTA index — built and
    // written when it is missing —, exon table, windows): independent inputs.  Whatever the reads' side has to say is held back
    // until the tasks are done, so that messages and exits come in the reference's order (tasks first, tools/dosplitalign.cpp:
    // 90-100).  A lookup asks the second file's store first, so a read id that both files hold resolves to the later one as
    // `reads[id] = sequence` does (tools/SplitAlignment.cpp:253-264).
    ReadStorePair reads;
    bool reads_ok[2] = {false, false};
    std::ostringstream reads_err[2];
    std::string reads_fatal[2];
    const std::string read_names[2] = {cmd.str("seq1"), cmd.str("seq2")};
    std::thread read_threads[2];
    for (int t = 0; t < 2; ++t)
        read_threads[t] = std::thread([&, t] { reads_ok[t] = AddReads(read_names[t], reads.file[t], reads_err[t], &reads_fatal[t]); });
    std::map<int, SplitAlignmentTask> tasks = CreateTasks(cmd.str("fasta"), cmd.str("exons"), cmd.real("ufrag"), cmd.real("sfrag"),
                                                         cmd.integer("minread"), cmd.integer("maxread"), regions);

    // SplitReadRealigner::AddTask (tools/SplitAlignment.cpp:236-251): 2000 bp bins over the mate regions
    // The ids in the bins are task ORDINALS (position in ascending fusion id order) rather than fusion ids: they sort the
    // same way, and per-batch tables indexed by them replace hash lookups in the take-up loop.
    BinnedLocations binned(2000);
    std::vector<const SplitAlignmentTask*> task_of;
    for (const auto& kv : tasks) {
        if (kv.first < 0) die("Error: negative fusion id " + std::to_string(kv.first));
        for (int ce = 0; ce <= 1; ++ce)
            for (const Location& loc : kv.second.mMateRegions[ce]) binned.Add(pack_id((int)task_of.size(), ce), loc);
        task_of.push_back(&kv.second);
    }
    stage("fasta index + exons + windows");
    for (int t = 0; t < 2; ++t) {
        read_threads[t].join();
        std::cerr << reads_err[t].str();
    }
    for (int t = 0; t < 2; ++t)
        if (!reads_fatal[t].empty()) die(reads_fatal[t]);
    if (!reads_ok[0] || !reads_ok[1]) {
        std::cout << "Error: unable to read sequences" << std::endl;
        std::cout.flush();
        std::cerr.flush();
        _exit(1);                             // (no helper thread exists any more; nothing to unwind that the system does not)
    }

    stage("reads");
    // One GPU batch: the candidates of a run of SAM records, in the reference's visiting order.
    struct Batch {
        std::vector<uint8_t> ref_bytes, read_bytes;
        std::vector<dsa_fusion> fusions;
        std::vector<dsa_pair> cand;
        void clear() { ref_bytes.clear(); read_bytes.clear(); fusions.clear(); cand.clear(); }
    };

    OrderedFileWriter out;
    if (!out.open_file(cmd.str("align"))) die("Error: Unable to open " + cmd.str("align"));

    // Candidates go to the GPU in batches (DEFUSE_DSA_BATCH_PAIRS, default 4 M) and their lines are written in the
    // reference's visiting order, so a run of any size streams through.  A batch is aligned, formatted and written by a
    // helper thread while the main thread enumerates the next one.
    size_t batch_pairs = (size_t)4 << 20;
    if (const char* e = std::getenv("DEFUSE_DSA_BATCH_PAIRS")) batch_pairs = std::max<size_t>(1, (size_t)std::atoll(e));
    const unsigned nThreads = host_threads();
    std::mutex time_mutex;
    std::vector<std::string> collected;
    auto run_batch = [&](Batch& B) {
        std::vector<dsa_pair>& cand = B.cand;
        if (cand.empty()) return;
        const double t_g0 = now();
        // Grouped by fusion for the kernels, fusions with many candidates first: the table-driven kernels take workgroups
        // (256 consecutive pairs) of at most four fusions, so the small fusions are kept together at the end instead of
        // dragging their big neighbours onto the generic path.  A counting sort: stable, so the visiting order inside a
        // fusion is kept.
        const size_t nf = B.fusions.size(), nc = cand.size();
        std::vector<int64_t> per_fusion(nf, 0);
        for (const dsa_pair& c : cand) ++per_fusion[c.fusion_idx];
        std::vector<int32_t> forder(nf);
        std::iota(forder.begin(), forder.end(), 0);
        std::stable_sort(forder.begin(), forder.end(), [&](int32_t x, int32_t y) { return per_fusion[x] > per_fusion[y]; });
        std::vector<int64_t> start(nf + 1, 0);
        {
            int64_t at = 0;
            for (int32_t f : forder) { start[f] = at; at += per_fusion[f]; }
        }
        std::vector<int64_t> slot_of(nc);
        std::vector<dsa_pair> pairs(nc);
        {
            std::vector<int64_t> cur(start.begin(), start.begin() + (std::ptrdiff_t)nf);
            for (size_t c = 0; c < nc; ++c) {
                const int64_t k = cur[cand[c].fusion_idx]++;
                slot_of[c] = k;
                pairs[(size_t)k] = cand[c];
            }
        }
        start_ctx();                              // (already running: a batch has candidates, so a record had overlaps)
        {
            static std::mutex join_mutex;
            std::lock_guard<std::mutex> lk(join_mutex);
            if (ctx_thread.joinable()) ctx_thread.join();
        }
        if (!dsa.error.empty()) die("Error: cannot load the split alignment library: " + dsa.error);
        if (ctx_rc != DSA_OK || !ctx) die("Error: no usable MI355X/HIP device (dsa_create failed)");
        std::vector<dsa_record> recs(std::max<size_t>(1024, 2 * pairs.size()));
        int64_t n = 0;
        int rc = dsa.align_batch(ctx, B.ref_bytes.data(), (int64_t)B.ref_bytes.size(), B.fusions.data(), (int32_t)nf,
                                 B.read_bytes.data(), (int64_t)B.read_bytes.size(), pairs.data(), (int64_t)pairs.size(), recs.data(),
                                 (int64_t)recs.size(), &n);
        if (rc == DSA_E_CAPACITY) {
            recs.resize((size_t)n);
            rc = dsa.align_batch(ctx, B.ref_bytes.data(), (int64_t)B.ref_bytes.size(), B.fusions.data(), (int32_t)nf,
                                 B.read_bytes.data(), (int64_t)B.read_bytes.size(), pairs.data(), (int64_t)pairs.size(),
                                 recs.data(), (int64_t)recs.size(), &n);
        }
        if (rc != DSA_OK) die(std::string("Error: split alignment on the GPU failed: ") + dsa.last_error(ctx));
        const double t_g1 = now();

        // back to the visiting order: records arrive grouped by batch pair index; contiguous shares of the candidates are
        // formatted side by side and written in order
        std::vector<int64_t> first(nc + 1, 0);
        for (int64_t k = 0; k < n; ++k) ++first[(size_t)recs[(size_t)k].pair_idx + 1];
        for (size_t k = 0; k < nc; ++k) first[k + 1] += first[k];
        const unsigned nt = nc < 4096 ? 1u : nThreads;
        std::vector<std::string> texts(nt);
        run_threads(nt, [&](unsigned t) {
            std::string& buf = texts[t];
            const size_t lo = nc * t / nt, hi = nc * (t + 1) / nt;
            for (size_t c = lo; c < hi; ++c) {
                const size_t k = (size_t)slot_of[c];
                for (int64_t r = first[k]; r < first[k + 1]; ++r) {
                    const dsa_record& a = recs[(size_t)r];
                    // SplitAlignment::WriteAlignment (tools/SplitAlignment.cpp:305-317): nine fields, each followed by a tab
                    for (int v : {a.fusion_id, a.frag, a.read_end, a.revcomp, a.ref_first, a.ref_second, a.read_first, a.read_second, a.score}) {
                        append_int(buf, v);
                        buf += '\t';
                    }
                    buf += '\n';
                }
            }
        });
        if (collect)
            for (std::string& tx : texts) collected.push_back(std::move(tx));     // fused mode: sorted and evaluated at the end
        else
            out.write_round(texts, nt);
        std::lock_guard<std::mutex> lk(time_mutex);
        t_gpu += t_g1 - t_g0;
        t_write += now() - t_g1;
    };
    Batch batch[2];
    int cur = 0;
    std::thread worker;
    auto flush = [&]() {                       // hand the current batch to the helper, continue in the other one
        if (worker.joinable()) worker.join();
        if (batch[cur].cand.empty()) return;
        Batch* b = &batch[cur];
        worker = std::thread([&run_batch, b] { run_batch(*b); b->clear(); });
        cur ^= 1;
    };

    // SplitReadRealigner::DoAlignment (tools/SplitAlignment.cpp:266-303).  The SAM text is mapped and taken in rounds of
    // 256 MiB, each cut into one piece per host thread.  Per round: (1) the pieces parse their records and look up the mate
    // regions they overlap, side by side; (2) the de-duplication on (fusion, read, revComp) (:268, :292: first come, first
    // kept) runs with the keys shared out over the threads by hash — every thread walks all candidates of the round in
    // file order and keeps the seen-set of its own keys; (3) one thread takes the kept candidates up in file order
    // (fusion table, offsets, batch cuts: cheap); (4) the reads are copied and reverse-complemented side by side.
    MappedText sam;
    sam.load(cmd.str("improper"), "Error: Unable to open sam file ");
    unsigned nPieces = nThreads;
    if (sam.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nPieces = 1;
    struct Hit { int frag, readEnd; uint32_t first, count; };           // readEnd -1: inherited from before the piece
    struct SamPiece {
        std::vector<Hit> hits;
        std::vector<int> ids;
        std::vector<uint8_t> keep;                                       // per id: first occurrence of its key
        size_t lines = 0, errorLine = 0;
        int errorKind = 0, lastReadEnd = -2, carryIn = 0;                // -2: no record of the piece set the read end
        std::string errorText;
    };
    std::vector<SamPiece> pieces(nPieces);
    std::vector<FlatSet64> seen(nPieces, FlatSet64(1 << 12));            // (fusion, read id, revComp) seen, by key hash
    std::vector<int> fusion_slot(task_of.size(), -1);                    // task ordinal -> index into the current batch's fusions
    std::vector<int> slots_used;
    size_t lineBase = 0;
    int carryReadEnd = 0;                                                // the reference's RawAlignment starts with read end 0
    auto new_batch = [&] {
        for (int o : slots_used) fusion_slot[(size_t)o] = -1;
        slots_used.clear();
    };
    auto key_of = [](int fusion_id, int rid, int revcomp) {
        return ((uint64_t)(uint32_t)fusion_id << 33) | ((uint64_t)(uint32_t)rid << 1) | (uint64_t)revcomp;
    };
    for (size_t lo = 0; lo < sam.size();) {
        size_t hi = std::min(sam.size(), lo + ((size_t)1 << 28));
        if (hi < sam.size()) hi = sam.line_end(hi - 1);
        const std::vector<size_t> cut = sam.cut_lines(lo, hi, nPieces);
        run_threads(nPieces, [&](unsigned t) {
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
        for (const SamPiece& pc : pieces)
            if (!pc.hits.empty()) { start_ctx(); break; }
        stage("  sam records + overlaps of a round");
        for (SamPiece& pc : pieces) {                                    // read end a piece's first records inherit
            pc.carryIn = carryReadEnd;
            if (pc.lastReadEnd != -2) carryReadEnd = pc.lastReadEnd;
            pc.keep.assign(pc.ids.size(), 0);
        }
        run_threads(nPieces, [&](unsigned t) {                           // (2)
            FlatSet64& mine = seen[t];
            for (SamPiece& pc : pieces)
                for (const Hit& h : pc.hits) {
                    const int mateReadEnd = h.readEnd < 0 ? pc.carryIn : h.readEnd;
                    const int rid = pack_id(h.frag, (mateReadEnd == 0) ? 1 : 0);
                    for (uint32_t k = 0; k < h.count; ++k) {
                        const int cid = pc.ids[h.first + k];
                        const uint64_t key = key_of(task_of[(size_t)(cid & 0x7FFFFFFF)]->mFusionID, rid, cid < 0 ? 0 : 1);
                        if (FlatSet64::hash(key ^ 0x9e3779b97f4a7c15ULL) % nPieces != t) continue;
                        if (mine.insert(key)) pc.keep[h.first + k] = 1;
                    }
                }
        });
        stage("  de-duplication of a round");
        // (3) + (4): the kept candidates of the round, batch by batch
        struct Pending { const char* src; size_t len; size_t dst; bool revcomp; };
        std::vector<Pending> copies;
        size_t read_at = batch[cur].read_bytes.size();                   // bytes of the current batch handed out so far
        {
            size_t kept = 0;
            for (const SamPiece& pc : pieces)
                for (uint8_t k : pc.keep) kept += k;
            copies.reserve(kept);
            batch[cur].cand.reserve(batch[cur].cand.size() + std::min(kept, batch_pairs + 4096));
        }
        auto fill_reads = [&](Batch& B) {
            B.read_bytes.resize(read_at);                                  // grown once per call, not per candidate
            if (copies.empty()) return;
            const unsigned nt = copies.size() < 4096 ? 1u : nThreads;
            run_threads(nt, [&](unsigned t) {
                std::string tmp;
                for (size_t k = copies.size() * t / nt; k < copies.size() * (t + 1) / nt; ++k) {
                    const Pending& c = copies[k];
                    if (!c.revcomp) { std::memcpy(B.read_bytes.data() + c.dst, c.src, c.len); continue; }
                    tmp.assign(c.src, c.len);
                    ReverseComplement(tmp);                               // the one definition of the complement (tools/Common.cpp)
                    std::memcpy(B.read_bytes.data() + c.dst, tmp.data(), c.len);
                }
            });
            copies.clear();
        };
        for (SamPiece& pc : pieces) {
            for (const Hit& h : pc.hits) {
                Batch& B = batch[cur];
                const int mateReadEnd = h.readEnd < 0 ? pc.carryIn : h.readEnd;
                const int frag = h.frag;
                const int read_end = (mateReadEnd == 0) ? 1 : 0;
                for (uint32_t k = 0; k < h.count; ++k) {
                    if (!pc.keep[h.first + k]) continue;
                    const int cid = pc.ids[h.first + k];
                    const int cluster_end = cid < 0 ? 1 : 0;
                    const int ordinal = cid & 0x7FFFFFFF;
                    const int revcomp = (cluster_end == 0) ? 1 : 0;
                    const char* rs = nullptr;              // a missing read aligns as the empty string (operator[] in the reference, :286)
                    size_t rn = 0;
                    if (!reads.get(frag, read_end, rs, rn)) rn = 0;
                    if (fusion_slot[(size_t)ordinal] < 0) {
                        const SplitAlignmentTask& t = *task_of[(size_t)ordinal];
                        dsa_fusion f;
                        f.fusion_id = t.mFusionID;
                        f.ref0_off = (int32_t)B.ref_bytes.size();
                        f.ref0_len = (int32_t)t.mSplitAlignSeq[0].size();
                        B.ref_bytes.insert(B.ref_bytes.end(), t.mSplitAlignSeq[0].begin(), t.mSplitAlignSeq[0].end());
                        f.ref1_off = (int32_t)B.ref_bytes.size();
                        f.ref1_len = (int32_t)t.mSplitAlignSeq[1].size();
                        B.ref_bytes.insert(B.ref_bytes.end(), t.mSplitAlignSeq[1].begin(), t.mSplitAlignSeq[1].end());
                        fusion_slot[(size_t)ordinal] = (int)B.fusions.size();
                        slots_used.push_back(ordinal);
                        B.fusions.push_back(f);
                    }
                    dsa_pair p{};
                    p.fusion_idx = fusion_slot[(size_t)ordinal];
                    p.read_off = (int32_t)read_at;
                    p.read_len = (int32_t)rn;
                    p.frag = frag;
                    p.read_end = (uint8_t)read_end;
                    p.revcomp = (uint8_t)revcomp;
                    copies.push_back(Pending{rs, rn, read_at, revcomp != 0});
                    read_at += rn;
                    B.cand.push_back(p);
                }
                // between two SAM records: a batch never splits the candidates of one record
                if (B.cand.size() >= batch_pairs || read_at > ((size_t)1 << 30) || B.ref_bytes.size() > ((size_t)1 << 30)) {
                    fill_reads(B);
                    new_batch();
                    flush();
                    read_at = batch[cur].read_bytes.size();
                }
            }
            if (pc.errorLine) {                                         // the records before the bad line were taken up, as a serial reader does
                fill_reads(batch[cur]);
                new_batch();
                flush();
                if (worker.joinable()) worker.join();
                out.close_file();
                if (pc.errorKind == 6) die("Error: bad integer '" + pc.errorText + "' as fragment name");
                DieSamLine(pc.errorKind, lineBase + pc.errorLine);
            }
            lineBase += pc.lines;
        }
        fill_reads(batch[cur]);            // the round's SAM text stays mapped, but the copies list is per round
        stage("  candidates of a round");
        lo = hi;
    }
    new_batch();
    flush();
    if (worker.joinable()) worker.join();
    if (collect) {
        // `sort -n -k 1` of the pipeline (scripts/defuse_run.pl:528) in the C locale: by fusion id, lines of one fusion in byte
        // order (sort's last-resort comparison).  Lines are indexed, grouped by id with a stable sort, and the groups — which
        // are independent — are ordered, written and (fused evaluation) evaluated by contiguous shares of the groups.
        struct Line { int id; uint32_t len; const char* p; };
        std::vector<Line> lines;
        for (const std::string& tx : collected)
            for (size_t pos = 0; pos < tx.size();) {
                const char* nl = (const char*)memchr(tx.data() + pos, '\n', tx.size() - pos);
                const size_t e = nl ? (size_t)(nl - tx.data()) + 1 : tx.size();
                int id = 0;
                const char* tab = (const char*)memchr(tx.data() + pos, '\t', e - pos);
                field_int(tx.data() + pos, tab ? (size_t)(tab - (tx.data() + pos)) : 0, id);
                lines.push_back(Line{id, (uint32_t)(e - pos), tx.data() + pos});
                pos = e;
            }
        std::stable_sort(lines.begin(), lines.end(), [](const Line& a, const Line& b) { return a.id < b.id; });
        std::vector<size_t> group(1, 0);
        for (size_t k = 1; k < lines.size(); ++k)
            if (lines[k].id != lines[k - 1].id) group.push_back(k);
        if (!lines.empty()) group.push_back(lines.size());
        const size_t ng = group.empty() ? 0 : group.size() - 1;
        const unsigned nt = ng < 64 ? 1u : nThreads;
        std::vector<std::string> sorted_text(nt);
        std::vector<EvalTexts> ev(nt);
        const SplitAlignmentTask emptyTask;
        run_threads(nt, [&](unsigned t) {
            std::vector<SplitAlignment> alignments;
            std::vector<const SplitAlignment*> kept;
            std::map<std::pair<int, int>, int> splitScore;
            for (size_t g = ng * t / nt; g < ng * (t + 1) / nt; ++g) {
                std::sort(lines.begin() + (std::ptrdiff_t)group[g], lines.begin() + (std::ptrdiff_t)group[g + 1], [](const Line& a, const Line& b) {
                    const int c = memcmp(a.p, b.p, std::min(a.len, b.len));
                    return c != 0 ? c < 0 : a.len < b.len;
                });
                alignments.clear();
                for (size_t k = group[g]; k < group[g + 1]; ++k) {
                    sorted_text[t].append(lines[k].p, lines[k].len);
                    if (fused_eval) {
                        SplitAlignment a;
                        bool id_read;
                        const std::string err = parse_line(lines[k].p, lines[k].len - 1, a, id_read);
                        if (!err.empty()) die(err);
                        alignments.push_back(a);
                    }
                }
                if (fused_eval) {
                    auto ti = tasks.find(lines[group[g]].id);
                    EvaluateGroup(ti == tasks.end() ? emptyTask : ti->second, alignments, ev[t], kept, splitScore);
                }
            }
        });
        out.write_round(sorted_text, nt);
        if (fused_eval) {
            const std::string names[3] = {opt_seq, opt_break, opt_predalign};
            for (int f = 0; f < 3; ++f) {
                OrderedFileWriter w;
                if (!w.open_file(names[f])) die("Error: Unable to open " + names[f]);
                std::vector<std::string> parts(nt);
                for (unsigned t = 0; t < nt; ++t) parts[t].swap(f == 0 ? ev[t].seq : f == 1 ? ev[t].brk : ev[t].pred);
                w.write_round(parts, nt);
                if (!w.close_file()) die("Error: failed writing " + names[f]);
            }
        }
        stage("fused: sort + evaluation + files");
    }
    stage("candidates + alignment + output");
    if (timing) std::cerr << "[dosplitalign] of which GPU calls " << t_gpu << " s, formatting and writing " << t_write << " s" << std::endl;
    if (ctx_thread.joinable()) ctx_thread.join();
    if (!out.close_file()) die("Error: failed writing " + cmd.str("align"));
    if (timing) std::cerr << "[dosplitalign] main() " << (now() - t_main) << " s" << std::endl;
    // The output is complete and closed.  The process ends here without unwinding the HIP runtime, the context and the
    // mapped inputs one by one (a tenth of a second of teardown that the operating system does at once anyway).
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    _exit(0);
}
