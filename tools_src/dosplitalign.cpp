// dosplitalign — drop-in replacement of the reference tool (tools/dosplitalign.cpp:25-111): same
// command line, same input formats, same output lines; the per-candidate
// SplitAlignmentTask::Align of tools/SplitAlignment.cpp:294 is replaced by batches streamed through the
// MI355X library (include/defuse_dsa.h: dsa_stream_*).  There is no CPU fallback: without a GPU the tool exits 1.
//
// Output order: one SAM record at a time, its overlapping cluster ends in ascending (signed) id order
// — the canonical order of SURVEY.md 8(c) for the reference's unordered_set iteration (the pipeline
// sorts the file by fusion id afterwards, scripts/defuse_run.pl:528).
//
// How the process is laid out (profiles/r04/tools/: what a short-lived GPU process pays on an MI355X host).  The pipeline
// starts one dosplitalign per chunk of a million reads (scripts/config.txt:112), and for such a chunk the DP is four
// milliseconds of GPU time — while the HIP runtime needs 0.12-0.24 s to start, every one of a process's first four streams
// 8-20 ms, the first batch 40 ms more than a later one, and a process that touched the GPU 50-120 ms to go away at exit.  So:
//   * the GPU side lives in a WORKER PROCESS forked at the start (before any thread exists): it opens the C-ABI library (as a
//     JNI / ctypes / cgo binding of the same ABI would), creates the dsa_stream, runs a tiny batch through it so that the
//     code objects and buffers exist, and then serves batches.  All of that passes while the main process reads its text
//     inputs.  The main process never touches the GPU, so nothing of the runtime's teardown stands between its last output
//     byte and its exit; the worker ends with it (PR_SET_PDEATHSIG) and the system cleans up behind both.
//     DEFUSE_DSA_INPROCESS=1 runs the same worker code on a thread of the one process instead.
//   * main process and worker share DEPTH batch slots (anonymous shared mappings made before the fork; DEFUSE_DSA_PINNED=1
//     makes the worker pin them): while the worker aligns batch k, the main thread and its team build batch k+1 in place
//     and a second thread with a team of its own formats and writes batch k-1.
//   * every host stage runs on all threads of a team: FASTQ parsing (ReadTable), SAM parsing, the de-duplication, the
//     candidate table of a round, the batches (fusion table, windows, counting sort by fusion, read copies), formatting, writing.
//
// Environment: DEFUSE_GPU=<ordinal> selects the device (default: lock files, dsa_pick_device; with HIP_VISIBLE_DEVICES the
// ordinal is relative to the visible set); DEFUSE_THREADS host threads per team (default 8); DEFUSE_DSA_BATCH_PAIRS
// candidates per batch (default 262144: a chunk of a million candidates is four batches, three in flight).
#include <chrono>
#include <climits>
#include <new>
#include <numeric>

#include <dlfcn.h>
#include <semaphore.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>

#include "../include/defuse_dsa.h"
#include "evaluate.hpp"

using namespace defuse;

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The C-ABI library is opened at run time instead of being a link dependency: only the worker ever loads it (and the HIP
// runtime behind it).  $DEFUSE_DSA_LIB names the library, default <directory of this binary>/../defuse_amd/libdefuse_dsa.so.
struct DsaLib {
    decltype(&dsa_pick_device) pick_device = nullptr;
    decltype(&dsa_stream_create) stream_create = nullptr;
    decltype(&dsa_stream_destroy) stream_destroy = nullptr;
    decltype(&dsa_stream_submit) stream_submit = nullptr;
    decltype(&dsa_stream_collect) stream_collect = nullptr;
    decltype(&dsa_stream_recollect) stream_recollect = nullptr;
    decltype(&dsa_stream_last_error) stream_last_error = nullptr;
    decltype(&dsa_host_register) host_register = nullptr;
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
            path = path.substr(0, path.find_last_of('/'));
            // (bin/<sanitizer>/dosplitalign sits one directory deeper)
            path += (access((path + "/../defuse_amd/libdefuse_dsa.so").c_str(), R_OK) == 0) ? "/../defuse_amd/libdefuse_dsa.so" : "/../../defuse_amd/libdefuse_dsa.so";
        }
        void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { error = dlerror(); return false; }
        pick_device = (decltype(pick_device))dlsym(h, "dsa_pick_device");
        stream_create = (decltype(stream_create))dlsym(h, "dsa_stream_create");
        stream_destroy = (decltype(stream_destroy))dlsym(h, "dsa_stream_destroy");
        stream_submit = (decltype(stream_submit))dlsym(h, "dsa_stream_submit");
        stream_collect = (decltype(stream_collect))dlsym(h, "dsa_stream_collect");
        stream_recollect = (decltype(stream_recollect))dlsym(h, "dsa_stream_recollect");
        stream_last_error = (decltype(stream_last_error))dlsym(h, "dsa_stream_last_error");
        host_register = (decltype(host_register))dlsym(h, "dsa_host_register");
        if (!pick_device || !stream_create || !stream_destroy || !stream_submit || !stream_collect || !stream_recollect || !stream_last_error || !host_register) {
            error = "missing symbols in " + path;
            return false;
        }
        return true;
    }
};

constexpr int DEPTH = 3;               // batch slots: one being built, one with the worker, one being written

// What the two sides share besides the slots' buffers.  Lives in a shared mapping; semaphores are process-shared.
struct Channel {
    sem_t submit_sem;                  // main -> worker: one post per submitted batch (and one to make it look at `quit`)
    sem_t collect_sem;                 // worker's submitter -> worker's collector
    sem_t done_sem[DEPTH];             // worker -> main: the records of the slot's batch are in place (or slot.rc says why not)
    std::atomic<int> status;           // 0 starting, 1 ready, -1 failed (error[] says why)
    std::atomic<int> quit;
    char error[1024];
    struct Slot {
        int64_t ref_len, read_len, n_pairs, n_records;
        int32_t n_fusions, rc;
        double t_submit, t_submitted, t_done;       // worker's clock: submit call entered / returned, records in place
    } slot[DEPTH];
    double t_start, t_loaded, t_stream, t_ready;     // worker's clock (same steady clock as the main process: one machine)
};

// The buffers of the slots: shared anonymous mappings made before the fork (same addresses on both sides), reserved at their
// largest useful size without committing memory (MAP_NORESERVE: pages exist once touched).
struct Regions {
    uint8_t* ref[DEPTH];
    uint8_t* reads[DEPTH];
    dsa_fusion* fusions[DEPTH];
    dsa_pair* pairs[DEPTH];
    dsa_record* recs[DEPTH];
    size_t cap_ref = 0, cap_reads = 0, cap_fusions = 0, cap_pairs = 0, cap_recs = 0;      // elements
    size_t hint_pairs = 0, hint_read_bytes = 0;       // what a batch usually holds: that much is made to exist ahead of its use
};

void* map_shared(size_t bytes)
{
    void* p = mmap(nullptr, std::max<size_t>(bytes, 4096), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    (void)madvise(p, bytes, MADV_HUGEPAGE);          // (takes effect where shared memory may use huge pages; harmless elsewhere)
    return p;
}

// Makes the first `bytes` of a slot buffer exist before anybody needs them.  A fresh page of a shared mapping costs a fault of
// 2-3 microseconds when it is first written — 15 ms for the records of a batch if the one thread that receives them pays, per
// slot; here the kernel allocates them in one call, on a helper thread, while the GPU runtime starts and the inputs are parsed.
void prefault(void* p, size_t bytes)
{
    if (!p || !bytes) return;
#ifdef MADV_POPULATE_WRITE
    if (madvise(p, bytes, MADV_POPULATE_WRITE) == 0) return;
#endif
    volatile char* c = (volatile char*)p;
    for (size_t k = 0; k < bytes; k += 4096) c[k] = 0;
}

// ---- the GPU worker ---------------------------------------------------------------------------------------------------
void worker_fail(Channel* ch, const std::string& msg)
{
    std::snprintf(ch->error, sizeof ch->error, "%s", msg.c_str());
    ch->status.store(-1);
    for (uint64_t k = 0;; ++k) {                     // every batch that still comes is answered with the failure
        while (sem_wait(&ch->submit_sem) != 0 && errno == EINTR) {}
        if (ch->quit.load()) return;
        Channel::Slot& sl = ch->slot[k % DEPTH];
        sl.rc = DSA_E_DEVICE;
        sl.n_records = 0;
        sem_post(&ch->done_sem[k % DEPTH]);
    }
}

void worker_main(Channel* ch, const Regions* R)
{
    ch->t_start = now();
    std::thread recs_ahead([R] {                      // the pages the records of the first batches land in, and this side's view of the inputs'
        for (int s = 0; s < DEPTH; ++s) prefault(R->recs[s], std::min(R->cap_recs, 2 * R->hint_pairs + 4096) * sizeof(dsa_record));
        for (int s = 0; s < DEPTH; ++s) {
            prefault(R->pairs[s], R->hint_pairs * sizeof(dsa_pair));
            prefault(R->reads[s], std::min(R->cap_reads, R->hint_read_bytes));
            prefault(R->ref[s], std::min<size_t>(R->cap_ref, (size_t)8 << 20));
        }
    });
    recs_ahead.detach();
    // two hardware queues are all this process uses (one compute lane at a time and the copies): each of a process's first
    // four costs 8-20 ms to create (profiles/r04/tools/init_exit.txt); the user's own setting wins
    setenv("GPU_MAX_HW_QUEUES", "2", 0);
    DsaLib dsa;
    if (!dsa.load()) return worker_fail(ch, "Error: cannot load the split alignment library: " + dsa.error);
    ch->t_loaded = now();
    dsa_stream* st = nullptr;
    if (dsa.stream_create(&st, dsa.pick_device(), DEPTH) != DSA_OK || !st)
        return worker_fail(ch, "Error: no usable MI355X/HIP device (dsa_stream_create failed)");
    ch->t_stream = now();
    {
        // One tiny batch through the stream: the code objects are loaded, every kernel of the path has run once and the small
        // buffers exist — 40 ms that would otherwise sit in front of the first real batch.  Its buffers are the worker's own.
        const int L = 40, W = 96, NP = 64;
        std::vector<uint8_t> ref(2 * W), reads((size_t)NP * L);
        for (int i = 0; i < 2 * W; ++i) ref[(size_t)i] = (uint8_t)"ACGT"[(i * 7 + i / 5) & 3];
        for (int p = 0; p < NP; ++p)
            for (int j = 0; j < L; ++j) reads[(size_t)p * L + j] = j < L / 2 ? ref[(size_t)(p % 8 + j)] : ref[(size_t)(W + 10 + p % 8 + j - L / 2)];
        dsa_fusion fu{1, 0, W, W, W};
        std::vector<dsa_pair> pairs(NP);
        for (int p = 0; p < NP; ++p) {
            pairs[(size_t)p] = dsa_pair{};
            pairs[(size_t)p].read_off = p * L;
            pairs[(size_t)p].read_len = L;
            pairs[(size_t)p].frag = p;
        }
        std::vector<dsa_record> out(4096);
        int64_t n = 0;
        int rc = dsa.stream_submit(st, ref.data(), (int64_t)ref.size(), &fu, 1, reads.data(), (int64_t)reads.size(), pairs.data(), NP, out.data(), (int64_t)out.size());
        if (rc == DSA_OK) rc = dsa.stream_collect(st, &n);
        if (rc != DSA_OK && rc != DSA_E_CAPACITY) return worker_fail(ch, std::string("Error: split alignment on the GPU failed: ") + dsa.stream_last_error(st));
        if (rc == DSA_E_CAPACITY) {                   // (cannot happen with 64 pairs; keep the stream in step all the same)
            std::vector<dsa_record> big((size_t)n);
            (void)dsa.stream_recollect(st, big.data(), n, &n);
        }
    }
    if (const char* e = std::getenv("DEFUSE_DSA_PINNED"))
        if (std::atoi(e) != 0) {
            // pin the part of every slot a batch of the default size uses (whole mappings would commit gigabytes); a batch that
            // needs more is copied from / to the unpinned rest at the runtime's staged rate
            const size_t np = std::min(R->cap_pairs, (size_t)300000);
            for (int s = 0; s < DEPTH; ++s) {
                (void)dsa.host_register(R->pairs[s], np * sizeof(dsa_pair));
                (void)dsa.host_register(R->reads[s], std::min(R->cap_reads, np * 160));
                (void)dsa.host_register(R->recs[s], std::min(R->cap_recs, 2 * np) * sizeof(dsa_record));
            }
        }
    ch->t_ready = now();
    ch->status.store(1);
    std::thread collector([&] {
        for (uint64_t k = 0;; ++k) {
            while (sem_wait(&ch->collect_sem) != 0 && errno == EINTR) {}
            if (ch->quit.load()) return;
            const int s = (int)(k % DEPTH);
            Channel::Slot& sl = ch->slot[s];
            int64_t n = 0;
            int rc = dsa.stream_collect(st, &n);
            if (rc == DSA_E_CAPACITY) rc = dsa.stream_recollect(st, R->recs[s], (int64_t)R->cap_recs, &n);
            if (rc == DSA_E_CAPACITY) std::snprintf(ch->error, sizeof ch->error, "Error: %lld alignments in one batch; lower DEFUSE_DSA_BATCH_PAIRS", (long long)n);
            else if (rc != DSA_OK) std::snprintf(ch->error, sizeof ch->error, "Error: split alignment on the GPU failed: %s", dsa.stream_last_error(st));
            sl.n_records = n;
            sl.rc = rc;
            sl.t_done = now();
            sem_post(&ch->done_sem[s]);
        }
    });
    for (uint64_t k = 0;; ++k) {
        while (sem_wait(&ch->submit_sem) != 0 && errno == EINTR) {}
        if (ch->quit.load()) break;
        const int s = (int)(k % DEPTH);
        Channel::Slot& sl = ch->slot[s];
        sl.t_submit = now();
        // (room for the records a batch of this size usually has; more are fetched by the collector into the whole mapping)
        const int64_t out_cap = (int64_t)std::min<size_t>(R->cap_recs, (size_t)sl.n_pairs * 4 + 4096);
        const int rc = dsa.stream_submit(st, R->ref[s], sl.ref_len, R->fusions[s], sl.n_fusions, R->reads[s], sl.read_len, R->pairs[s], sl.n_pairs, R->recs[s], out_cap);
        sl.t_submitted = now();
        if (rc != DSA_OK) {
            std::snprintf(ch->error, sizeof ch->error, "Error: split alignment on the GPU failed: %s", dsa.stream_last_error(st));
            sl.rc = rc;
            sl.n_records = 0;
            sl.t_done = now();
            sem_post(&ch->done_sem[s]);         // the main process ends on the first failure: nothing later is waited for
            continue;
        }
        sem_post(&ch->collect_sem);
    }
    sem_post(&ch->collect_sem);
    collector.join();
}

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
    // DEFUSE_DSA_SORTED=1: the reference's command line, the alignments written in the order of `LC_ALL=C sort -n -k 1` (the order
    // inside the file is free — the pipeline sorts every chunk's file next, scripts/defuse_run.pl:528 — and GNU sort runs 2.6 times
    // faster over input that is in order already)
    const bool sorted_env = [] { const char* e = std::getenv("DEFUSE_DSA_SORTED"); return e && std::atoi(e) != 0; }();
    const bool collect = fused_eval || (fused_cli && cmd.is_set("sorted")) || sorted_env;
    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    const double t_main = now();
    double t_stage = now();
    auto stage = [&](const char* name) {
        const double t = now();
        if (timing) std::cerr << "[dosplitalign] " << name << " " << (t - t_stage) << " s" << std::endl;
        t_stage = t;
    };

    // An error exit (die(), on whichever thread) ends the process the way the success path does — output flushed, no exit
    // handlers beside threads that are still alive — and takes the worker process with it.  Only one thread gets that far.
    static std::mutex die_mutex;
    static pid_t worker_pid = -1;
    die_hook() = [] {
        die_mutex.lock();
        if (worker_pid > 0) kill(worker_pid, SIGKILL);
        std::cout.flush();
        std::cerr.flush();
        fflush(nullptr);
        _exit(1);
    };

    // ---- the slots and the worker (before any thread of this process exists) ----------------------------------------------
    size_t batch_pairs = (size_t)1 << 18;
    if (const char* e = std::getenv("DEFUSE_DSA_BATCH_PAIRS")) batch_pairs = std::max<size_t>(1, (size_t)std::atoll(e));
    // a batch never splits the candidates of one SAM record, and a record has at most two candidates per fusion
    size_t limit_read_bytes = (size_t)1 << 30, limit_ref_bytes = (size_t)1 << 30;      // offsets into a batch's bytes are int32
    if (const char* e = std::getenv("DEFUSE_DSA_BATCH_READ_BYTES")) limit_read_bytes = std::max<size_t>(1, (size_t)std::atoll(e));
    if (const char* e = std::getenv("DEFUSE_DSA_BATCH_REF_BYTES")) limit_ref_bytes = std::max<size_t>(1, (size_t)std::atoll(e));
    Regions R;
    Channel* ch = nullptr;
    bool have_worker = false, inprocess = false;
    std::thread worker_thread;
    auto start_worker = [&](size_t n_fusions) {          // n_fusions: the regions' count, or a bound of it (it sizes address space only)
        R.cap_pairs = std::min<size_t>(batch_pairs + 2 * n_fusions + 4096, ((size_t)1 << 31) - 1);
        R.cap_fusions = std::min(R.cap_pairs, n_fusions) + 1;
        R.hint_pairs = std::min(batch_pairs, R.cap_pairs);
        R.hint_read_bytes = std::min<size_t>(R.hint_pairs * (size_t)std::max(cmd.integer("maxread"), 32), (size_t)256 << 20);
        ch = (Channel*)map_shared(sizeof(Channel));
        // The slots are reserved at their largest useful size (address space, not memory).  Under a limit on address space
        // (`ulimit -v`, a scheduler's h_vmem) that may be refused: the byte buffers are then reserved at half the size, and half
        // again, and the batches are cut to what was had.
        bool ok = false;
        for (int attempt = 0; attempt < 7 && ch && !ok; ++attempt) {
            R.cap_ref = (((size_t)1 << 31) - 1) >> attempt;
            R.cap_reads = (((size_t)1 << 31) - 1) >> attempt;
            R.cap_recs = std::max<size_t>(8 * R.cap_pairs, ((size_t)64 << 20) >> attempt);
            ok = true;
            for (int s = 0; s < DEPTH && ok; ++s) {
                R.ref[s] = (uint8_t*)map_shared(R.cap_ref);
                R.reads[s] = (uint8_t*)map_shared(R.cap_reads);
                R.fusions[s] = (dsa_fusion*)map_shared(R.cap_fusions * sizeof(dsa_fusion));
                R.pairs[s] = (dsa_pair*)map_shared(R.cap_pairs * sizeof(dsa_pair));
                R.recs[s] = (dsa_record*)map_shared(R.cap_recs * sizeof(dsa_record));
                ok = R.ref[s] && R.reads[s] && R.fusions[s] && R.pairs[s] && R.recs[s];
            }
            if (!ok)
                for (int s = 0; s < DEPTH; ++s) {
                    if (R.ref[s]) munmap(R.ref[s], std::max<size_t>(R.cap_ref, 4096));
                    if (R.reads[s]) munmap(R.reads[s], std::max<size_t>(R.cap_reads, 4096));
                    if (R.fusions[s]) munmap(R.fusions[s], std::max<size_t>(R.cap_fusions * sizeof(dsa_fusion), 4096));
                    if (R.pairs[s]) munmap(R.pairs[s], std::max<size_t>(R.cap_pairs * sizeof(dsa_pair), 4096));
                    if (R.recs[s]) munmap(R.recs[s], std::max<size_t>(R.cap_recs * sizeof(dsa_record), 4096));
                    R.ref[s] = R.reads[s] = nullptr; R.fusions[s] = nullptr; R.pairs[s] = nullptr; R.recs[s] = nullptr;
                }
        }
        if (!ok) die("Error: cannot reserve the batch buffers");
        limit_read_bytes = std::min(limit_read_bytes, R.cap_reads);
        limit_ref_bytes = std::min(limit_ref_bytes, R.cap_ref);
        new (ch) Channel();
        sem_init(&ch->submit_sem, 1, 0);
        sem_init(&ch->collect_sem, 1, 0);
        for (sem_t& s : ch->done_sem) sem_init(&s, 1, 0);
        ch->status.store(0);
        ch->quit.store(0);
        ch->error[0] = 0;
        inprocess = [] { const char* e = std::getenv("DEFUSE_DSA_INPROCESS"); return e && std::atoi(e) != 0; }();
        if (!inprocess) {
            std::cout.flush();
            std::cerr.flush();
            const pid_t parent = getpid();
            const pid_t pid = fork();
            if (pid == 0) {
                // the worker: ends with the main process, holds none of its pipes open (a parent that reads our stdout / stderr to
                // their end must not wait for the worker's teardown), reports through the channel only
                prctl(PR_SET_PDEATHSIG, SIGKILL);
                if (getppid() != parent) _exit(0);
                const int nul = open("/dev/null", O_RDWR);
                if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); dup2(nul, 2); }
                for (int fd = 3; fd < 256; ++fd) close(fd);
                die_hook() = [] { _exit(1); };
                worker_main(ch, &R);
                _exit(0);
            }
            if (pid < 0) inprocess = true;           // no second process to be had: the same worker on a thread
            else worker_pid = pid;
        }
        if (inprocess) worker_thread = std::thread(worker_main, ch, &R);
        have_worker = true;
    };
    // The worker is started as early as the sizes allow: the HIP runtime takes 0.15-0.35 s to come up, as long as this process
    // needs for regions, tasks and bins.  The slots are address space (MAP_NORESERVE): a bound of the number of fusions from the
    // size of the regions (or clusters) file is as good as the count.  An input that is not a plain file, or a very large one,
    // starts the worker after the regions are read, by their count.
    {
        const std::string& sized_by = opt_clusters.empty() ? cmd.str("regions") : opt_clusters;
        struct stat st;
        if (sized_by != "-" && stat(sized_by.c_str(), &st) == 0 && S_ISREG(st.st_mode)) {
            const size_t bound = (size_t)st.st_size / (opt_clusters.empty() ? 12 : 32) + 16;      // two lines per fusion, this short at least
            if (bound <= ((size_t)4 << 20)) start_worker(bound);
        }
    }
    if (!opt_clusters.empty()) {
        // scripts/get_align_regions.pl:14-53 (as bin/defuse_glue get_align_regions): per cluster end the reference, the strand
        // and the span of its alignments; clusters ascending, end 0 then 1
        std::string text;
        try {
            ClusterPieces pieces;                 // (its threads end before the worker is forked)
            pieces.load(opt_clusters);
            text = align_regions_text(pieces);
        } catch (const GlueError& g) {
            die(g.msg);
        }
        OrderedFileWriter rf;
        if (!rf.open_file(cmd.str("regions"))) die("Error: unable to write " + cmd.str("regions"));
        rf.write_round({text}, 1);
        if (!rf.close_file()) die("Error: failed writing " + cmd.str("regions"));
        stage("clusters -> regions file");
    }
    const std::map<int, std::vector<Location>> regions = ReadAlignRegionPairs(cmd.str("regions"));
    stage("regions");

    if (!have_worker && !regions.empty()) start_worker(regions.size());
    std::thread inputs_ahead;
    if (have_worker) {
        inputs_ahead = std::thread([&R] {             // the pages the first batches are built in
            for (int s = 0; s < DEPTH; ++s) {
                prefault(R.pairs[s], R.hint_pairs * sizeof(dsa_pair));
                prefault(R.reads[s], std::min(R.cap_reads, R.hint_read_bytes));
                prefault(R.ref[s], std::min<size_t>(R.cap_ref, (size_t)8 << 20));
                prefault(R.fusions[s], std::min<size_t>(R.cap_fusions, 65536) * sizeof(dsa_fusion));
            }
        });
        inputs_ahead.detach();
    }
    // waits for a slot's records; notices a worker process that is gone
    auto wait_done = [&](int s) {
        for (;;) {
            timespec ts;
            clock_gettime(CLOCK_REALTIME, &ts);
            ts.tv_nsec += 200000000;
            if (ts.tv_nsec >= 1000000000) { ts.tv_nsec -= 1000000000; ++ts.tv_sec; }
            if (sem_timedwait(&ch->done_sem[s], &ts) == 0) return;
            if (errno == EINTR) continue;
            if (worker_pid > 0) {
                int st = 0;
                if (waitpid(worker_pid, &st, WNOHANG) == worker_pid) {
                    worker_pid = -1;
                    die("Error: the GPU worker process ended unexpectedly (status " + std::to_string(st) + ")");
                }
            }
        }
    };

    // threads per team: DEFUSE_THREADS, else up to 16 (the teams replace a thread start per pass by a wake-up, so sixteen pay
    // where eight used to be the limit: ten million candidates in 1.10 s instead of 1.33 s on a 16-core GPU box)
    const unsigned nThreads = std::getenv("DEFUSE_THREADS") ? host_threads() : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    // The two FASTQ files are read by a team of their own while the main thread sets the tasks up (FASTA index — built and
    // written when it is missing —, exon table, windows) and parses the first round of SAM text: independent inputs.  Whatever
    // the reads' side has to say is held back until the tasks are done, so that messages and exits come in the reference's
    // order (tasks first, tools/dosplitalign.cpp:90-100).  A lookup asks the second file's table first, so a read id that both
    // files hold resolves to the later one as `reads[id] = sequence` does (tools/SplitAlignment.cpp:253-264).
    ReadTablePair reads;
    bool reads_ok[2] = {false, false};
    std::ostringstream reads_err[2];
    std::string reads_fatal[2];
    const std::string read_names[2] = {cmd.str("seq1"), cmd.str("seq2")};
    double t_reads_done = 0;
    std::thread reads_thread([&] {
        Team rteam(nThreads);
        for (int t = 0; t < 2; ++t) reads_ok[t] = reads.file[t].load(read_names[t], rteam, reads_err[t], &reads_fatal[t]);
        t_reads_done = now();
    });
    struct Joiner {                          // (an exception that unwinds main must not meet a joinable thread)
        std::thread& th;
        ~Joiner() { if (th.joinable()) th.join(); }
    } reads_joiner{reads_thread};
    std::map<int, SplitAlignmentTask> tasks = CreateTasks(cmd.str("fasta"), cmd.str("exons"), cmd.real("ufrag"), cmd.real("sfrag"),
                                                         cmd.integer("minread"), cmd.integer("maxread"), regions, nThreads);
    stage("  tasks (windows from the FASTA, mate regions)");

    // SplitReadRealigner::AddTask (tools/SplitAlignment.cpp:236-251): 2000 bp bins over the mate regions
    // The ids in the bins are task ORDINALS (position in ascending fusion id order) rather than fusion ids: they sort the
    // same way, and per-batch tables indexed by them replace hash lookups.
    BinnedLocations binned(2000);
    std::vector<const SplitAlignmentTask*> task_of;
    std::vector<int> fusion_id_of;
    for (const auto& kv : tasks) {
        if (kv.first < 0) die("Error: negative fusion id " + std::to_string(kv.first));
        for (int ce = 0; ce <= 1; ++ce)
            for (const Location& loc : kv.second.mMateRegions[ce]) binned.Add(pack_id((int)task_of.size(), ce), loc);
        task_of.push_back(&kv.second);
        fusion_id_of.push_back(kv.second.mFusionID);
    }
    binned.Finish();
    const size_t n_tasks = task_of.size();
    stage("fasta index + exons + windows");
    bool reads_reported = false;
    auto report_reads = [&] {                // once, before the first candidate is taken up
        if (reads_reported) return;
        reads_reported = true;
        reads_thread.join();
        for (int t = 0; t < 2; ++t) std::cerr << reads_err[t].str();
        for (int t = 0; t < 2; ++t)
            if (!reads_fatal[t].empty()) die(reads_fatal[t]);
        if (!reads_ok[0] || !reads_ok[1]) {
            std::cout << "Error: unable to read sequences" << std::endl;
            die_hook()();
        }
        if (timing) std::cerr << "[dosplitalign] reads: in place " << (t_reads_done - t_main) << " s after the start" << std::endl;
    };

    OrderedFileWriter out;
    if (!out.open_file(cmd.str("align"))) die("Error: Unable to open " + cmd.str("align"));

    // ---- the writer: collects the batches in order, formats and writes them ------------------------------------------------
    struct BatchMeta {
        std::vector<int32_t> slot_of;        // candidate (visiting order) -> pair index in the batch (grouped by fusion)
        size_t n = 0;
        double t_built = 0;
    };
    BatchMeta meta[DEPTH];
    sem_t free_sem;                          // slots the builder may fill
    sem_init(&free_sem, 0, DEPTH);
    std::mutex q_mutex;
    std::condition_variable q_cv;
    uint64_t n_built = 0;                    // batches handed to the worker
    bool no_more = false;
    struct Text {                            // a formatted batch share kept for the fused tail: the writer's buffer itself, not a copy
        std::unique_ptr<char[]> p;
        size_t len = 0;
        const char* data() const { return p.get(); }
        size_t size() const { return len; }
    };
    std::vector<Text> collected;
    double t_wait_gpu = 0, t_format = 0, t_service = 0, t_first_wait = 0;
    uint64_t n_batches = 0;
    double svc_ms[8][2] = {};
    int n_svc = 0;
    std::thread writer([&] {
        Team wteam(nThreads);
        std::vector<int32_t> first, count;
        struct Buf { std::unique_ptr<char[]> p; size_t cap = 0, len = 0; };
        std::vector<Buf> bufs(wteam.size());
        for (uint64_t k = 0;; ++k) {
            {
                std::unique_lock<std::mutex> lk(q_mutex);
                q_cv.wait(lk, [&] { return n_built > k || no_more; });
                if (n_built <= k) return;
            }
            const int s = (int)(k % DEPTH);
            const double t0 = now();
            wait_done(s);
            const double t1 = now();
            const Channel::Slot& sl = ch->slot[s];
            if (sl.rc != DSA_OK) die(ch->error[0] ? std::string(ch->error) : std::string("Error: split alignment on the GPU failed"));
            if ((size_t)sl.n_records > R.cap_recs)
                die("Error: " + std::to_string(sl.n_records) + " alignments in one batch; lower DEFUSE_DSA_BATCH_PAIRS");
            t_wait_gpu += t1 - t0;
            if (k == 0) t_first_wait = t1 - t0;
            t_service += sl.t_done - sl.t_submit;
            if (k < 8) { svc_ms[k][0] = 1e3 * (sl.t_submitted - sl.t_submit); svc_ms[k][1] = 1e3 * (sl.t_done - sl.t_submit); n_svc = (int)k + 1; }
            const BatchMeta& M = meta[s];
            const size_t nc = M.n, nr = (size_t)sl.n_records;
            const dsa_record* recs = R.recs[s];
            first.resize(nc);
            count.resize(nc);
            const unsigned nt = wteam.size();
            std::vector<size_t> sizes(nt, 0);
            std::vector<off_t> at;
            wteam.run([&](unsigned t) {
                // records arrive grouped by pair index: where the run of every pair begins and how long it is
                for (size_t c = nc * t / nt; c < nc * (t + 1) / nt; ++c) count[c] = 0;
                wteam.barrier();
                for (size_t r = nr * t / nt; r < nr * (t + 1) / nt; ++r) {
                    const int32_t p = recs[r].pair_idx;
                    if (r > 0 && recs[r - 1].pair_idx == p) continue;
                    size_t e = r + 1;
                    while (e < nr && recs[e].pair_idx == p) ++e;
                    first[(size_t)p] = (int32_t)r;
                    count[(size_t)p] = (int32_t)(e - r);
                }
                wteam.barrier();
                // back to the visiting order: contiguous shares of the candidates are formatted side by side
                const size_t lo = nc * t / nt, hi = nc * (t + 1) / nt;
                size_t mine = 0;
                for (size_t c = lo; c < hi; ++c) mine += (size_t)count[(size_t)M.slot_of[c]];
                Buf& B = bufs[t];
                if (B.cap < mine * 112 + 16) {
                    B.cap = mine * 112 + mine * 14 + 4096;
                    B.p.reset(new char[B.cap]);
                }
                char* w = B.p.get();
                for (size_t c = lo; c < hi; ++c) {
                    const size_t p = (size_t)M.slot_of[c];
                    if (!count[p]) continue;
                    for (int32_t r = first[p], e = first[p] + count[p]; r < e; ++r) {
                        const dsa_record& a = recs[r];
                        // SplitAlignment::WriteAlignment (tools/SplitAlignment.cpp:305-317): nine fields, each followed by a tab
                        for (int v : {a.fusion_id, a.frag, a.read_end, a.revcomp, a.ref_first, a.ref_second, a.read_first, a.read_second, a.score}) {
                            w = put_int(w, v);
                            *w++ = '\t';
                        }
                        *w++ = '\n';
                    }
                }
                B.len = (size_t)(w - B.p.get());
                sizes[t] = B.len;
                if (collect || !out.seekable()) return;
                wteam.barrier();
                if (t == 0) at = out.reserve_parts(sizes);
                wteam.barrier();
                out.write_part(B.p.get(), B.len, at[t]);
            });
            if (collect)
                for (Buf& B : bufs) {                    // fused mode: sorted and evaluated at the end
                    if (!B.len) continue;
                    collected.emplace_back();
                    collected.back().p = std::move(B.p);
                    collected.back().len = B.len;
                    B.cap = B.len = 0;
                }
            else if (!out.seekable())
                for (const Buf& B : bufs) out.append(B.p.get(), B.len);
            t_format += now() - t1;
            ++n_batches;
            sem_post(&free_sem);
        }
    });
    auto finish_writer = [&] {               // everything submitted is written; then the writer ends
        {
            std::lock_guard<std::mutex> lk(q_mutex);
            no_more = true;
        }
        q_cv.notify_all();
        if (writer.joinable()) writer.join();
    };
    Joiner writer_joiner{writer};

    // SplitReadRealigner::DoAlignment (tools/SplitAlignment.cpp:266-303).  The SAM text is mapped and taken in rounds of
    // 256 MiB, each cut into one piece per host thread.  Per round: (1) the pieces parse their records and look up the mate
    // regions they overlap, side by side; (2) the de-duplication on (fusion, read, revComp) (:268, :292: first come, first
    // kept) runs with the keys shared out over the threads by hash — every thread walks all candidates of the round in
    // file order and keeps the seen-set of its own keys; (3) the kept candidates of the round become one table in visiting
    // order (read found, lengths summed), side by side; (4) the table is cut into batches at record boundaries and every
    // batch is built in its slot by the whole team: fusion table and windows, counting sort of the pairs by fusion (the
    // device plans its sweep per fusion and wants a fusion's pairs together), oriented read bytes.
    MappedText sam;
    sam.load(cmd.str("improper"), "Error: Unable to open sam file ", false);
    unsigned nPieces = nThreads;
    if (sam.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nPieces = 1;
    Team team(nPieces);
    struct Hit { int frag, readEnd; uint32_t first, count; };           // readEnd -1: inherited from before the piece
    struct SamPiece {
        std::vector<Hit> hits;
        std::vector<int> ids;
        std::vector<uint8_t> keep;                                       // per id: first occurrence of its key
        size_t lines = 0, errorLine = 0, kept = 0, cand_at = 0;
        int errorKind = 0, lastReadEnd = -2, carryIn = 0;                // -2: no record of the piece set the read end
        std::string errorText;
    };
    struct Cand {                                                        // one kept candidate, in visiting order
        const char* rs;
        uint64_t roff;                                                   // read bytes of the round before this one
        uint32_t rn;
        int32_t cid, frag;
        uint8_t read_end, first_of_record;
    };
    std::vector<SamPiece> pieces(nPieces);
    std::vector<FlatSet64> seen(nPieces, FlatSet64(1 << 12));            // (fusion, read id, revComp) seen, by key hash
    std::vector<Cand> cands;
    std::vector<std::atomic<uint8_t>> used(n_tasks);                     // per task ordinal: in the batch being built
    for (auto& u : used) u.store(0, std::memory_order_relaxed);
    std::vector<int32_t> fusion_slot(n_tasks, -1);                       // task ordinal -> index into the batch's fusions
    std::vector<std::vector<int32_t>> hist(nPieces);                     // per thread: pairs per fusion slot of its share
    std::vector<size_t> part_a(nPieces + 1), part_b(nPieces + 1);
    std::vector<int64_t> slot_start;
    std::vector<int32_t> slot_count;
    size_t lineBase = 0;
    int carryReadEnd = 0;                                                // the reference's RawAlignment starts with read end 0
    double t_build = 0, t_slot_wait = 0;
    uint8_t comp[256];                                                   // ReverseComplement's table (tools/Common.cpp:32-54): ACGT and acgt swap, anything else stays
    {
        std::string all(256, '\0');
        for (int c = 0; c < 256; ++c) all[(size_t)c] = (char)c;
        std::string rc = all;
        ReverseComplement(rc);                                           // the one definition of the complement
        for (int c = 0; c < 256; ++c) comp[(uint8_t)all[(size_t)c]] = (uint8_t)rc[(size_t)(255 - c)];
    }
    auto key_of = [](int fusion_id, int rid, int revcomp) {
        return ((uint64_t)(uint32_t)fusion_id << 33) | ((uint64_t)(uint32_t)rid << 1) | (uint64_t)revcomp;
    };

    // One batch = candidates [c0, c1) of the round's table, built in slot s.  false: its windows do not fit (the caller halves it).
    auto build_batch = [&](size_t c0, size_t c1, int s, bool single_record) -> bool {
        const size_t nc = c1 - c0;
        const unsigned nt = team.size();
        BatchMeta& M = meta[s];
        M.slot_of.resize(nc);
        M.n = nc;
        dsa_pair* const pairs = R.pairs[s];
        dsa_fusion* const fus = R.fusions[s];
        uint8_t* const refb = R.ref[s];
        uint8_t* const readb = R.reads[s];
        const uint64_t rbase = cands[c0].roff;
        size_t n_slots = 0, ref_total = 0;
        bool fits = true;
        team.run([&](unsigned t) {
            const size_t lo = c0 + nc * t / nt, hi = c0 + nc * (t + 1) / nt;
            // (a) the fusions of the batch
            for (size_t c = lo; c < hi; ++c) used[(size_t)(cands[c].cid & 0x7FFFFFFF)].store(1, std::memory_order_relaxed);
            team.barrier();
            // (b) their slots in ascending ordinal order, their windows behind one another
            const size_t olo = n_tasks * t / nt, ohi = n_tasks * (t + 1) / nt;
            size_t cnt = 0, bytes = 0;
            for (size_t o = olo; o < ohi; ++o)
                if (used[o].load(std::memory_order_relaxed)) {
                    ++cnt;
                    bytes += task_of[o]->mSplitAlignSeq[0].size() + task_of[o]->mSplitAlignSeq[1].size();
                }
            part_a[t + 1] = cnt;
            part_b[t + 1] = bytes;
            team.barrier();
            if (t == 0) {
                part_a[0] = part_b[0] = 0;
                for (unsigned u = 0; u < nt; ++u) { part_a[u + 1] += part_a[u]; part_b[u + 1] += part_b[u]; }
                n_slots = part_a[nt];
                ref_total = part_b[nt];
                fits = ref_total <= (single_record ? R.cap_ref : limit_ref_bytes) && n_slots <= R.cap_fusions;       // one record's candidates cannot be split
                slot_start.resize(n_slots + 1);
                slot_count.resize(n_slots + 1);
            }
            team.barrier();
            size_t slot = part_a[t], at = part_b[t];
            for (size_t o = olo; o < ohi; ++o)
                if (used[o].load(std::memory_order_relaxed)) {
                    used[o].store(0, std::memory_order_relaxed);
                    if (!fits) continue;
                    const SplitAlignmentTask& tk = *task_of[o];
                    dsa_fusion f;
                    f.fusion_id = tk.mFusionID;
                    f.ref0_off = (int32_t)at;
                    f.ref0_len = (int32_t)tk.mSplitAlignSeq[0].size();
                    std::memcpy(refb + at, tk.mSplitAlignSeq[0].data(), tk.mSplitAlignSeq[0].size());
                    at += tk.mSplitAlignSeq[0].size();
                    f.ref1_off = (int32_t)at;
                    f.ref1_len = (int32_t)tk.mSplitAlignSeq[1].size();
                    std::memcpy(refb + at, tk.mSplitAlignSeq[1].data(), tk.mSplitAlignSeq[1].size());
                    at += tk.mSplitAlignSeq[1].size();
                    fus[slot] = f;
                    fusion_slot[o] = (int32_t)slot++;
                }
            if (!fits) return;                               // (every thread sees the same verdict: no barrier is left alone)
            // (c) stable counting sort of the candidates by fusion slot: every thread counts its share, ...
            std::vector<int32_t>& h = hist[t];
            if (h.size() < n_slots) h.resize(n_slots);
            std::fill(h.begin(), h.begin() + (std::ptrdiff_t)n_slots, 0);
            team.barrier();                                   // (fusion_slot complete)
            for (size_t c = lo; c < hi; ++c) ++h[(size_t)fusion_slot[(size_t)(cands[c].cid & 0x7FFFFFFF)]];
            team.barrier();
            // ... the counts of a slot become the threads' starts inside it, the slots' totals their starts in the batch, ...
            const size_t slo = n_slots * t / nt, shi = n_slots * (t + 1) / nt;
            size_t total = 0;
            for (size_t k = slo; k < shi; ++k) {
                int32_t run = 0;
                for (unsigned u = 0; u < nt; ++u) {
                    const int32_t v = hist[u][k];
                    hist[u][k] = run;
                    run += v;
                }
                slot_count[k] = run;
                total += (size_t)run;
            }
            part_a[t + 1] = total;
            team.barrier();
            if (t == 0) {
                part_a[0] = 0;
                for (unsigned u = 0; u < nt; ++u) part_a[u + 1] += part_a[u];
            }
            team.barrier();
            {
                int64_t run = (int64_t)part_a[t];
                for (size_t k = slo; k < shi; ++k) {
                    slot_start[k] = run;
                    run += slot_count[k];
                }
            }
            team.barrier();
            // ... and every candidate goes to its place: pair, oriented read bytes
            for (size_t c = lo; c < hi; ++c) {
                const Cand& cd = cands[c];
                const size_t sl = (size_t)fusion_slot[(size_t)(cd.cid & 0x7FFFFFFF)];
                const size_t k = (size_t)slot_start[sl] + (size_t)h[sl]++;
                const int revcomp = cd.cid < 0 ? 0 : 1;                  // cluster end 0 -> reverse complement (tools/SplitAlignment.cpp:283-290)
                dsa_pair p{};
                p.fusion_idx = (int32_t)sl;
                p.read_off = (int32_t)(cd.roff - rbase);
                p.read_len = (int32_t)cd.rn;
                p.frag = cd.frag;
                p.read_end = cd.read_end;
                p.revcomp = (uint8_t)revcomp;
                pairs[k] = p;
                M.slot_of[c - c0] = (int32_t)k;
                uint8_t* dst = readb + (cd.roff - rbase);
                if (!revcomp) std::memcpy(dst, cd.rs, cd.rn);
                else
                    for (uint32_t j = 0; j < cd.rn; ++j) dst[j] = comp[(uint8_t)cd.rs[cd.rn - 1 - j]];
            }
        });
        if (!fits) return false;
        Channel::Slot& sl = ch->slot[s];
        sl.ref_len = (int64_t)ref_total;
        sl.n_fusions = (int32_t)n_slots;
        sl.read_len = (int64_t)((c1 < cands.size() ? cands[c1].roff : cands.back().roff + cands.back().rn) - rbase);
        sl.n_pairs = (int64_t)nc;
        sl.rc = DSA_OK;
        sl.n_records = 0;
        return true;
    };
    // hands the candidates [c0, c1) of the round's table to the worker, in as many batches as their windows need
    std::function<void(size_t, size_t)> submit_range = [&](size_t c0, size_t c1) {
        if (c0 >= c1) return;
        double t0 = now();
        while (sem_wait(&free_sem) != 0 && errno == EINTR) {}
        const double t1 = now();
        t_slot_wait += t1 - t0;
        const int s = (int)(n_built % DEPTH);
        bool single = true;
        for (size_t c = c0 + 1; c < c1 && single; ++c) single = !cands[c].first_of_record;
        if (!build_batch(c0, c1, s, single)) {
            sem_post(&free_sem);
            if (single) die("Error: the reference windows of one alignment's candidates exceed the batch buffer (" + std::to_string(R.cap_ref) + " bytes)");
            size_t mid = c0 + (c1 - c0) / 2;                             // the windows of the batch's fusions exceed the limit: two halves,
            while (mid < c1 && !cands[mid].first_of_record) ++mid;       // cut between two records
            if (mid >= c1) {
                mid = c0 + (c1 - c0) / 2;
                while (mid > c0 && !cands[mid].first_of_record) --mid;
            }
            submit_range(c0, mid);
            submit_range(mid, c1);
            return;
        }
        meta[s].t_built = now();
        t_build += meta[s].t_built - t1;
        {
            std::lock_guard<std::mutex> lk(q_mutex);
            ++n_built;
        }
        sem_post(&ch->submit_sem);
        q_cv.notify_all();
    };

    for (size_t lo = 0; lo < sam.size();) {
        size_t hi = std::min(sam.size(), lo + ((size_t)1 << 28));
        if (hi < sam.size()) hi = sam.line_end(hi - 1);
        const std::vector<size_t> cut = sam.cut_lines(lo, hi, nPieces);
        team.run([&](unsigned t) {                                       // (1)
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
                if (kind) { pc.errorLine = pc.lines; pc.errorKind = kind; break; }
                reference.assign(f.reference, f.reference_len);
                binned.Overlapping(reference, f.strand, f.region, overlapping);   // ascending signed order: end-1 ids (negative) first
                if (overlapping.empty()) continue;
                Hit h;
                if (!field_int(f.fragment, f.fragment_len, h.frag)) {
                    pc.errorLine = pc.lines; pc.errorKind = 6; pc.errorText.assign(f.fragment, f.fragment_len);
                    break;
                }
                h.readEnd = readEnd;
                h.first = (uint32_t)pc.ids.size();
                h.count = (uint32_t)overlapping.size();
                pc.ids.insert(pc.ids.end(), overlapping.begin(), overlapping.end());
                pc.hits.push_back(h);
            }
            if (readEnd >= 0) pc.lastReadEnd = readEnd;
            pc.keep.assign(pc.ids.size(), 0);
        });
        // a piece with a bad line ends the input: what later pieces found is dropped, as a serial reader never saw it
        unsigned last_piece = nPieces;
        for (unsigned t = 0; t < nPieces; ++t)
            if (pieces[t].errorLine) { last_piece = t + 1; break; }
        for (unsigned t = last_piece; t < nPieces; ++t) { pieces[t].hits.clear(); pieces[t].ids.clear(); pieces[t].keep.clear(); }
        stage("  sam records + overlaps of a round");
        size_t ids_total = 0;
        for (SamPiece& pc : pieces) {                                    // read end a piece's first records inherit
            pc.carryIn = carryReadEnd;
            if (pc.lastReadEnd != -2) carryReadEnd = pc.lastReadEnd;
            ids_total += pc.ids.size();
        }
        team.run([&](unsigned t) {                                       // (2)
            FlatSet64& mine = seen[t];
            // room for the round's keys — and, from what the first round found per byte of text, for the whole file's, so that
            // the tables are not rebuilt round after round
            const size_t whole = lo == 0 ? (size_t)((double)ids_total * (double)sam.size() / (double)(hi - lo)) : 0;
            mine.reserve(std::max(mine.size() + ids_total / nPieces + ids_total / (4 * nPieces) + 64, whole / nPieces + whole / (4 * nPieces)));
            // the thread's own keys are inserted a few dozen behind the scan, their slots asked for as they are met: the
            // tables grow to tens of megabytes per thread and every insert would otherwise wait for memory
            constexpr int AHEAD = 32;
            uint64_t pend_key[AHEAD];
            uint8_t* pend_keep[AHEAD];
            int np = 0;
            auto drain = [&] {
                for (int k = 0; k < np; ++k)
                    if (mine.insert(pend_key[k])) *pend_keep[k] = 1;
                np = 0;
            };
            for (SamPiece& pc : pieces)
                for (const Hit& h : pc.hits) {
                    const int mateReadEnd = h.readEnd < 0 ? pc.carryIn : h.readEnd;
                    const int rid = pack_id(h.frag, (mateReadEnd == 0) ? 1 : 0);
                    for (uint32_t k = 0; k < h.count; ++k) {
                        const int cid = pc.ids[h.first + k];
                        const uint64_t key = key_of(fusion_id_of[(size_t)(cid & 0x7FFFFFFF)], rid, cid < 0 ? 0 : 1);
                        if (FlatSet64::hash(key ^ 0x9e3779b97f4a7c15ULL) % nPieces != t) continue;
                        mine.prefetch(key);
                        pend_key[np] = key;
                        pend_keep[np] = &pc.keep[h.first + k];
                        if (++np == AHEAD) drain();
                    }
                }
            drain();
        });
        stage("  de-duplication of a round");
        if (ids_total) report_reads();
        // (3) the kept candidates of the round in visiting order
        team.run([&](unsigned t) {
            SamPiece& pc = pieces[t];
            size_t kept = 0;
            for (uint8_t k : pc.keep) kept += k;
            pc.kept = kept;
            team.barrier();
            if (t == 0) {
                size_t at = 0;
                for (SamPiece& q : pieces) { q.cand_at = at; at += q.kept; }
                cands.resize(at);
            }
            team.barrier();
            size_t at = pc.cand_at;
            uint64_t bytes = 0;
            for (const Hit& h : pc.hits) {
                const int mateReadEnd = h.readEnd < 0 ? pc.carryIn : h.readEnd;
                const int read_end = (mateReadEnd == 0) ? 1 : 0;
                bool first = true;
                for (uint32_t k = 0; k < h.count; ++k) {
                    if (!pc.keep[h.first + k]) continue;
                    Cand cd;
                    cd.cid = pc.ids[h.first + k];
                    cd.frag = h.frag;
                    cd.read_end = (uint8_t)read_end;
                    cd.first_of_record = first ? 1 : 0;
                    first = false;
                    size_t rn = 0;                            // a missing read aligns as the empty string (operator[] in the reference, :286)
                    cd.rs = nullptr;
                    if (!reads.get(h.frag, read_end, cd.rs, rn)) { rn = 0; cd.rs = ""; }
                    cd.rn = (uint32_t)rn;
                    cd.roff = bytes;                          // relative to the piece for now
                    bytes += rn;
                    cands[at++] = cd;
                }
            }
            part_b[t + 1] = (size_t)bytes;
            team.barrier();
            if (t == 0) {
                part_b[0] = 0;
                for (unsigned u = 0; u < nPieces; ++u) part_b[u + 1] += part_b[u];
            }
            team.barrier();
            for (size_t c = pc.cand_at; c < pc.cand_at + pc.kept; ++c) cands[c].roff += part_b[t];
        });
        stage("  candidate table of a round");
        // (4) batches: cut between two SAM records once there are batch_pairs candidates (or too many read bytes)
        const size_t n_round = cands.size();
        for (size_t c0 = 0; c0 < n_round;) {
            size_t c1 = std::min(n_round, c0 + batch_pairs);
            while (c1 < n_round && !cands[c1].first_of_record) ++c1;
            auto bytes_of = [&](size_t e) { return (e < n_round ? cands[e].roff : cands[n_round - 1].roff + cands[n_round - 1].rn) - cands[c0].roff; };
            if (c1 - c0 > R.cap_pairs) die("Error: one alignment has more candidates than a batch holds");
            if (bytes_of(c1) > limit_read_bytes) {
                size_t e = c1 - 1;                                       // the last cut between two records that keeps the bytes within the limit
                while (e > c0 && !(cands[e].first_of_record && bytes_of(e) <= limit_read_bytes)) --e;
                if (e == c0) {                                           // the first record by itself is over the limit: it goes alone, if it can
                    e = c0 + 1;
                    while (e < n_round && !cands[e].first_of_record) ++e;
                    if (bytes_of(e) > (uint64_t)R.cap_reads) die("Error: the reads of one alignment's candidates exceed the batch buffer (" + std::to_string(R.cap_reads) + " bytes)");
                }
                c1 = e;
            }
            submit_range(c0, c1);
            c0 = c1;
        }
        team.run([&](unsigned t) { sam.drop_pages(lo + (hi - lo) * t / nPieces, lo + (hi - lo) * (t + 1) / nPieces); });      // this round's text is not read again
        stage("  batches of a round");
        for (unsigned t = 0; t < last_piece; ++t) {
            const SamPiece& pc = pieces[t];
            if (pc.errorLine) {                                         // the records before the bad line were taken up, as a serial reader does
                finish_writer();
                out.close_file();
                if (pc.errorKind == 6) die("Error: bad integer '" + pc.errorText + "' as fragment name");
                DieSamLine(pc.errorKind, lineBase + pc.errorLine);
            }
            lineBase += pc.lines;
        }
        lo = hi;
    }
    report_reads();                          // a run without a single overlap still reports what the FASTQ files had to say
    finish_writer();
    if (collect) {
        // `sort -n -k 1` of the pipeline (scripts/defuse_run.pl:528) in the C locale: by fusion id, lines of one fusion in byte
        // order (sort's last-resort comparison).  Lines are indexed, grouped by id with a stable sort, and the groups — which
        // are independent — are ordered, written and (fused evaluation) evaluated by contiguous shares of the groups.
        struct Line { int id; uint32_t len; const char* p; };
        std::vector<Line> lines;
        {
            // the texts are indexed by contiguous shares of them, one per thread, and brought into id order by a counting sort
            // over the range of the ids (per-thread histograms, a thread's lines behind those of the threads before it: stable)
            const unsigned nt = collected.size() < 4 ? 1u : nThreads;
            std::vector<std::vector<Line>> mine(nt);
            std::vector<int> id_lo(nt, INT_MAX), id_hi(nt, INT_MIN);
            size_t total_bytes = 0;
            for (const Text& tx : collected) total_bytes += tx.size();
            std::vector<size_t> first_text(nt + 1, collected.size());
            {
                size_t acc = 0;
                unsigned t = 0;
                first_text[0] = 0;
                for (size_t k = 0; k < collected.size(); ++k) {
                    while (t + 1 < nt && acc >= total_bytes / nt * (t + 1)) first_text[++t] = k;
                    acc += collected[k].size();
                }
                while (t + 1 < nt) first_text[++t] = collected.size();
            }
            run_threads(nt, [&](unsigned t) {
                std::vector<Line> v;                         // (local: neighbouring vectors' headers share cache lines)
                int lo_seen = INT_MAX, hi_seen = INT_MIN;
                size_t bytes = 0;
                for (size_t k = first_text[t]; k < first_text[t + 1]; ++k) bytes += collected[k].size();
                v.reserve(bytes / 48 + 16);                  // alignment lines are longer than this
                for (size_t k = first_text[t]; k < first_text[t + 1]; ++k) {
                    const Text& tx = collected[k];
                    for (size_t pos = 0; pos < tx.size();) {
                        const char* nl = (const char*)memchr(tx.data() + pos, '\n', tx.size() - pos);
                        const size_t e = nl ? (size_t)(nl - tx.data()) + 1 : tx.size();
                        int id = 0;
                        const char* tab = (const char*)memchr(tx.data() + pos, '\t', e - pos);
                        field_int(tx.data() + pos, tab ? (size_t)(tab - (tx.data() + pos)) : 0, id);
                        v.push_back(Line{id, (uint32_t)(e - pos), tx.data() + pos});
                        lo_seen = std::min(lo_seen, id);
                        hi_seen = std::max(hi_seen, id);
                        pos = e;
                    }
                }
                id_lo[t] = lo_seen;
                id_hi[t] = hi_seen;
                mine[t].swap(v);
            });
            size_t n_lines = 0;
            int lo_id = INT_MAX, hi_id = INT_MIN;
            for (unsigned t = 0; t < nt; ++t) { n_lines += mine[t].size(); lo_id = std::min(lo_id, id_lo[t]); hi_id = std::max(hi_id, id_hi[t]); }
            lines.resize(n_lines);
            const uint64_t range = n_lines ? (uint64_t)((int64_t)hi_id - (int64_t)lo_id) + 1 : 0;
            if (n_lines && range <= std::max<uint64_t>(4 * (uint64_t)n_lines, (uint64_t)1 << 20) && range * nt <= ((uint64_t)1 << 28)) {
                std::vector<std::vector<size_t>> at(nt, std::vector<size_t>((size_t)range, 0));
                run_threads(nt, [&](unsigned t) {
                    size_t* mine_at = at[t].data();
                    for (const Line& l : mine[t]) ++mine_at[(size_t)(l.id - lo_id)];
                });
                size_t run = 0;
                for (size_t v = 0; v < (size_t)range; ++v)
                    for (unsigned t = 0; t < nt; ++t) { const size_t c = at[t][v]; at[t][v] = run; run += c; }
                run_threads(nt, [&](unsigned t) {
                    size_t* mine_at = at[t].data();
                    Line* to = lines.data();
                    for (const Line& l : mine[t]) to[mine_at[(size_t)(l.id - lo_id)]++] = l;
                });
            } else {
                size_t k = 0;
                for (unsigned t = 0; t < nt; ++t)
                    for (const Line& l : mine[t]) lines[k++] = l;
                std::stable_sort(lines.begin(), lines.end(), [](const Line& a, const Line& b) { return a.id < b.id; });
            }
        }
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
            std::string my_text;                             // (local, handed over at the end: the shared vectors' elements are neighbours in memory)
            EvalTexts my_ev;
            {
                size_t bytes = 0;
                for (size_t k = group[ng * t / nt]; k < group[ng * (t + 1) / nt]; ++k) bytes += lines[k].len;
                my_text.reserve(bytes);
            }
            for (size_t g = ng * t / nt; g < ng * (t + 1) / nt; ++g) {
                std::sort(lines.begin() + (std::ptrdiff_t)group[g], lines.begin() + (std::ptrdiff_t)group[g + 1], [](const Line& a, const Line& b) {
                    const int c = memcmp(a.p, b.p, std::min(a.len, b.len));
                    return c != 0 ? c < 0 : a.len < b.len;
                });
                alignments.clear();
                for (size_t k = group[g]; k < group[g + 1]; ++k) {
                    my_text.append(lines[k].p, lines[k].len);
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
                    EvaluateGroup(ti == tasks.end() ? emptyTask : ti->second, alignments, my_ev, kept, splitScore);
                }
            }
            sorted_text[t].swap(my_text);
            ev[t].seq.swap(my_ev.seq); ev[t].brk.swap(my_ev.brk); ev[t].pred.swap(my_ev.pred);
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
    stage("alignment + output behind the last round");
    if (timing && have_worker) {
        const int st = ch->status.load();
        std::cerr << "[dosplitalign] GPU worker (" << (inprocess ? "thread" : "process") << "): library loaded "
                  << (ch->t_loaded - ch->t_start) << " s, stream " << (ch->t_stream - ch->t_loaded) << " s, first batch + buffers "
                  << (ch->t_ready - ch->t_stream) << " s; " << (st == 1 ? "ready " : st < 0 ? "failed " : "not ready ")
                  << (st == 1 ? ch->t_ready - t_main : 0.0) << " s after the start of main()" << std::endl;
        std::cerr << "[dosplitalign] " << n_batches << " batches: building " << t_build << " s (waiting for a free slot " << t_slot_wait
                  << " s), of which GPU calls " << t_service << " s (submit to records in place, worker's clock), the writer waited " << t_wait_gpu
                  << " s for records (" << t_first_wait << " s of it for the first batch), formatting and writing " << t_format << " s" << std::endl;
        std::cerr << "[dosplitalign] first batches, ms in dsa_stream_submit / until the records were in place:";
        for (int k = 0; k < n_svc; ++k) std::cerr << " " << svc_ms[k][0] << "/" << svc_ms[k][1];
        std::cerr << std::endl;
    }
    if (!out.close_file()) die("Error: failed writing " + cmd.str("align"));
    team.run([&](unsigned t) {               // the FASTQ text goes the same way: the process's end has that much less to unmap by itself
        for (const ReadTable& f : reads.file) f.drop_pages(t, nPieces);
    });
    if (have_worker) {
        ch->quit.store(1);
        sem_post(&ch->submit_sem);
        if (inprocess && worker_thread.joinable()) worker_thread.join();
    }
    if (timing) std::cerr << "[dosplitalign] main() " << (now() - t_main) << " s" << std::endl;
    // The output is complete and closed.  The process ends here without unwinding its teams, tables and mapped inputs one by
    // one; the worker process ends with it (PR_SET_PDEATHSIG) and nobody waits for the GPU runtime's teardown.
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    if (inprocess && std::getenv("DEFUSE_FULL_EXIT")) return 0;            // under a profiler that writes its files at exit (rocprofv3)
    _exit(0);
}
