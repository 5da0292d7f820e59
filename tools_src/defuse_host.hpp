// defuse_host.hpp — host-side logic of the drop-in `dosplitalign` / `evalsplitalign` binaries.
//
// Written from scratch (C++17, no Boost, no samtools) to behave like the reference's C++98 tools
// for the split-alignment path; every block cites the reference code it mirrors (paths relative
// to the reference tree).  The DP itself is NOT here: candidates are batched and handed to the HIP
// library through include/defuse_dsa.h.
#pragma once
#include <fcntl.h>
#include <climits>
#include <malloc.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <cerrno>
#include <cctype>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace defuse {

enum Strand { PlusStrand = 0, MinusStrand = 1 };   // tools/Common.h:20-24

struct Region { int start = 0, end = 0; };
struct Location { std::string refName; int strand = 0, start = 0, end = 0; };

// A tool with helper threads that may sit inside a library (dlopen, a runtime coming up) installs a hook that ends the
// process in an orderly way — std::exit from an arbitrary thread would run exit handlers and library finalisers beside them.
inline std::function<void()>& die_hook()
{
    static std::function<void()> hook;
    return hook;
}
[[noreturn]] inline void die(const std::string& msg)
{
    std::cerr << msg << std::endl;
    if (die_hook()) die_hook()();
    std::exit(1);
}

// C++ `int / int` truncates toward zero — the bin arithmetic of the reference relies on it.
inline std::vector<std::string> split_tabs(const std::string& line, char sep = '\t')
{
    std::vector<std::string> out;   // boost::split(is_any_of("\t")): empty fields are kept
    size_t b = 0;
    for (;;) {
        size_t e = line.find(sep, b);
        if (e == std::string::npos) { out.emplace_back(line.substr(b)); break; }
        out.emplace_back(line.substr(b, e - b));
        b = e + 1;
    }
    return out;
}

// boost::lexical_cast<int>: optional sign, digits only, no whitespace, no trailing junk, range checked.
inline bool lexical_int(const std::string& s, int& out)
{
    if (s.empty()) return false;
    size_t k = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (k == s.size()) return false;
    long long v = 0;
    for (size_t i = k; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (s[i] - '0');
        if (v > 2147483648LL) return false;
    }
    if (s[0] == '-') v = -v;
    if (v > 2147483647LL || v < -2147483648LL) return false;
    out = (int)v;
    return true;
}
inline int lexical_int_or_die(const std::string& s, const std::string& context)
{
    int v;
    if (!lexical_int(s, v)) die("Error: bad integer '" + s + "' " + context);   // reference: uncaught bad_lexical_cast
    return v;
}

// lines of a text file without a std::string per line: blocks of 4 MiB, a line is [ptr, ptr + len)
class LineReader {
public:
    explicit LineReader(FILE* f) : f_(f), buf_(1 << 22) {}
    bool next(const char*& ptr, size_t& len)
    {
        for (;;) {
            const char* nl = (const char*)memchr(buf_.data() + pos_, '\n', end_ - pos_);
            if (nl) {
                ptr = buf_.data() + pos_;
                len = (size_t)(nl - ptr);
                pos_ = (size_t)(nl - buf_.data()) + 1;
                return true;
            }
            if (eof_) {
                if (pos_ == end_) return false;
                ptr = buf_.data() + pos_;          // last line without a newline, as std::getline returns it
                len = end_ - pos_;
                pos_ = end_;
                return true;
            }
            if (pos_ > 0) {
                memmove(buf_.data(), buf_.data() + pos_, end_ - pos_);
                end_ -= pos_;
                pos_ = 0;
            }
            if (end_ == buf_.size()) buf_.resize(buf_.size() * 2);
            const size_t got = fread(buf_.data() + end_, 1, buf_.size() - end_, f_);
            end_ += got;
            if (got == 0) eof_ = true;
        }
    }
private:
    FILE* f_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
};

// boost::lexical_cast<int> on a field (tools/AlignmentStream.cpp:170-186): optional sign, digits only, int range
inline bool field_int(const char* p, size_t n, int& out)
{
    if (n == 0) return false;
    size_t k = (p[0] == '+' || p[0] == '-') ? 1 : 0;
    if (k == n) return false;
    long long v = 0;
    for (; k < n; ++k) {
        if (p[k] < '0' || p[k] > '9') return false;
        v = v * 10 + (p[k] - '0');
        if (v > 2147483648LL) return false;
    }
    if (p[0] == '-') v = -v;
    if (v > 2147483647LL || v < -2147483648LL) return false;
    out = (int)v;
    return true;
}

// A tool that builds and drops gigabytes of small lists on many threads: keep what is freed (no trimming, no mmap per large
// block) — handing pages back to the system and faulting them in again is where such a run spends its system time.
// DEFUSE_MALLOC_DEFAULT=1 leaves the allocator alone.
inline void keep_freed_memory()
{
    if (std::getenv("DEFUSE_MALLOC_DEFAULT")) return;
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, INT_MAX);
    mallopt(M_TOP_PAD, 256 << 20);
}

// For a tool that defines DEFUSE_HUGE_NEW before this header (clustermatepairs): large blocks (tables, parsed records and output
// texts: gigabytes at full size) come from mappings of their own that ask for transparent huge pages.  Where the system grants them on request only (/sys/kernel/mm/transparent_hugepage/enabled =
// madvise, as on the MI355X boxes) a table's first touch is then one fault per 2 MiB instead of one per 4 KiB: clustermatepairs at
// 50 M fragments 5.7 -> 5.0-5.2 s (profiles/r04/tools/cmp50_huge_pages.txt; setcover, whose tables are indexed by fragment id and
// touched sparsely, loses 0.6 s with it and does not use it).  Everything below the
// threshold is malloc's.  DEFUSE_NO_HUGE_BLOCKS=1 (or DEFUSE_MALLOC_DEFAULT=1) switches it off; sanitizer builds keep their own
// operator new.
#if defined(DEFUSE_HUGE_NEW) && !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__)
namespace huge_blocks {
constexpr size_t ALIGN = (size_t)2 << 20;
inline size_t threshold()          // DEFUSE_HUGE_BLOCK_MIN (bytes): measurements
{
    static const size_t v = [] { const char* e = std::getenv("DEFUSE_HUGE_BLOCK_MIN"); return e ? (size_t)std::max(4096LL, std::atoll(e)) : (size_t)8 << 20; }();
    return v;
}
struct Block { void* base; size_t len; };
struct Registry {
    std::mutex m;
    std::unordered_map<void*, Block> blocks;        // by the pointer handed out (2 MiB aligned)
};
inline Registry& registry() { static Registry* r = new Registry; return *r; }     // never destroyed: deletes may come late
inline bool enabled() { static const bool on = !std::getenv("DEFUSE_MALLOC_DEFAULT") && !std::getenv("DEFUSE_NO_HUGE_BLOCKS"); return on; }
inline void* take(size_t n)
{
    const size_t len = ((n + 4095) & ~(size_t)4095) + ALIGN;
    void* base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (base == MAP_FAILED) return nullptr;
    void* user = (void*)(((uintptr_t)base + ALIGN - 1) & ~(uintptr_t)(ALIGN - 1));
    (void)madvise(user, len - ((uintptr_t)user - (uintptr_t)base), MADV_HUGEPAGE);
    Registry& r = registry();
    std::lock_guard<std::mutex> lk(r.m);
    r.blocks[user] = Block{base, len};
    return user;
}
inline bool give_back(void* p)
{
    if ((uintptr_t)p & (ALIGN - 1)) return false;       // (malloc's blocks are never 2 MiB aligned in practice; the table decides)
    Registry& r = registry();
    Block b;
    {
        std::lock_guard<std::mutex> lk(r.m);
        auto it = r.blocks.find(p);
        if (it == r.blocks.end()) return false;
        b = it->second;
        r.blocks.erase(it);
    }
    munmap(b.base, b.len);
    return true;
}
}  // namespace huge_blocks
}  // namespace defuse
void* operator new(size_t n)
{
    if (n >= defuse::huge_blocks::threshold() && defuse::huge_blocks::enabled())
        if (void* p = defuse::huge_blocks::take(n)) return p;
    void* p = std::malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept
{
    if (p && !defuse::huge_blocks::give_back(p)) std::free(p);
}
void operator delete[](void* p) noexcept { operator delete(p); }
void operator delete(void* p, size_t) noexcept { operator delete(p); }
void operator delete[](void* p, size_t) noexcept { operator delete(p); }
namespace defuse {
#endif

// host threads of a tool: DEFUSE_THREADS, else 8 (profiles/microbench/cmp_threads.sh: beyond that the joins cost more than
// the pieces save), never more than the machine has
inline unsigned host_threads()
{
    if (const char* e = std::getenv("DEFUSE_THREADS")) return (unsigned)std::max(1, std::atoi(e));
    return std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
}
inline void run_threads(unsigned n, const std::function<void(unsigned)>& fn)
{
    if (n <= 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n; ++t) th.emplace_back(fn, t);
    for (std::thread& x : th) x.join();
}

// A team of n threads that lives as long as the object: run(fn) executes fn(t) for t = 0..n-1 (t = 0 on the caller) and
// returns when all are done; inside fn, barrier() makes all n wait for each other.  The passes of a tool over one batch of
// input (count, prefix, place, ...) run as ONE run() with barriers between them instead of a thread start per pass.
class Team {
public:
    explicit Team(unsigned n) : n_(std::max(1u, n))
    {
        for (unsigned t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
    }
    ~Team()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (std::thread& x : th_) x.join();
    }
    Team(const Team&) = delete;
    Team& operator=(const Team&) = delete;
    unsigned size() const { return n_; }
    void run(const std::function<void(unsigned)>& fn)
    {
        if (n_ == 1) { fn(0); return; }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            pending_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
    void barrier()
    {
        if (n_ == 1) return;
        std::unique_lock<std::mutex> lk(bm_);
        const unsigned g = bgen_;
        if (++bcount_ == n_) {
            bcount_ = 0;
            ++bgen_;
            bcv_.notify_all();
        } else {
            bcv_.wait(lk, [&] { return bgen_ != g; });
        }
    }
    // [0, n) cut into the team's shares: the share of thread t
    static size_t lo(size_t n, unsigned t, unsigned of) { return n * t / of; }
private:
    void loop(unsigned t)
    {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(unsigned)>* fn;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            (*fn)(t);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    const unsigned n_;
    std::vector<std::thread> th_;
    std::mutex m_, bm_;
    std::condition_variable cv_, done_cv_, bcv_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned gen_ = 0, pending_ = 0, bgen_ = 0, bcount_ = 0;
    bool stop_ = false;
};

// 64-bit key -> 32-bit value, open addressing with linear probing (the bin pair tables of clustermatepairs: millions of
// keys, looked up once per alignment); key ~0 is reserved.  find_or_add returns the slot's value reference and whether
// the key was new.
class FlatMap64 {
public:
    explicit FlatMap64(size_t cap_pow2 = 1024) : keys_(cap_pow2, EMPTY), vals_(cap_pow2, 0), mask_(cap_pow2 - 1) {}
    uint32_t& find_or_add(uint64_t key, bool& added)
    {
        if ((n_ + 1) * 2 > keys_.size()) grow();
        size_t i = hash(key) & mask_;
        for (;; i = (i + 1) & mask_) {
            if (keys_[i] == key) { added = false; return vals_[i]; }
            if (keys_[i] == EMPTY) { keys_[i] = key; ++n_; added = true; return vals_[i]; }
        }
    }
    size_t size() const { return n_; }
    void release() { std::vector<uint64_t>().swap(keys_); std::vector<uint32_t>().swap(vals_); n_ = 0; mask_ = 0; }
private:
    static constexpr uint64_t EMPTY = ~(uint64_t)0;
    static uint64_t hash(uint64_t x)
    {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    }
    void grow()
    {
        std::vector<uint64_t> ok(keys_.size() * 2, EMPTY);
        std::vector<uint32_t> ov(keys_.size() * 2, 0);
        ok.swap(keys_); ov.swap(vals_);
        mask_ = keys_.size() - 1;
        for (size_t k = 0; k < ok.size(); ++k) {
            if (ok[k] == EMPTY) continue;
            size_t i = hash(ok[k]) & mask_;
            while (keys_[i] != EMPTY) i = (i + 1) & mask_;
            keys_[i] = ok[k]; vals_[i] = ov[k];
        }
    }
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> vals_;
    size_t mask_, n_ = 0;
};

// A whole text input in memory: a regular file is mapped (threads that parse pieces of it fault its pages in side by side),
// anything else ("-" = stdin, pipes) is collected in a plain buffer.
struct MappedText {
    char* p = nullptr;
    size_t n = 0, cap = 0;
    bool mapped = false;
    MappedText() = default;
    MappedText(const MappedText&) = delete;
    MappedText& operator=(const MappedText&) = delete;
    ~MappedText() { release(); }
    const char* data() const { return p; }
    size_t size() const { return n; }
    char operator[](size_t k) const { return p[k]; }
    void release()
    {
        if (p) { if (mapped) munmap(p, n); else free(p); }
        p = nullptr; n = cap = 0; mapped = false;
    }
    void load(const std::string& name, const std::string& open_error, bool populate = true)
    {
        if (!try_load(name, true, populate)) die(open_error + name);
    }
    // false: the file cannot be opened (or mapped); true otherwise, an empty file included.  "-" is stdin only when allowed.
    // populate = false leaves the page tables to the threads that read the text (a team that parses pieces side by side
    // faults a gigabyte in faster than one thread's MAP_POPULATE maps it: 0.19 s for 2.4 GB, profiles/r04/tools/).
    bool try_load(const std::string& name, bool dash_is_stdin = true, bool populate = true)
    {
        FILE* in = stdin;
        if (name == "-" && dash_is_stdin) {
            // a file redirected into stdin (`tool < file`, as the pipeline calls its text steps) is mapped like a named one
            struct stat st;
            if (fstat(STDIN_FILENO, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && lseek(STDIN_FILENO, 0, SEEK_CUR) == 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), STDIN_FILENO, 0);
                if (m != MAP_FAILED) { p = (char*)m; n = (size_t)st.st_size; mapped = true; return true; }
            }
        }
        if (name != "-" || !dash_is_stdin) {
            const int fd = open(name.c_str(), O_RDONLY);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) != 0 || S_ISDIR(st.st_mode)) { close(fd); return false; }
            if (S_ISREG(st.st_mode) && st.st_size > 0) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);
                close(fd);
                if (m == MAP_FAILED) return false;
                p = (char*)m; n = (size_t)st.st_size; mapped = true;
                return true;
            }
            in = fdopen(fd, "rb");
            if (!in) { close(fd); return false; }
        }
        for (;;) {
            if (cap - n < ((size_t)1 << 24)) {
                cap = std::max<size_t>(cap * 2, (size_t)1 << 26);
                p = (char*)realloc(p, cap);
                if (!p) die("Error: out of memory reading " + name);
            }
            const size_t got = fread(p + n, 1, cap - n, in);
            if (got == 0) break;
            n += got;
        }
        if (in != stdin) fclose(in);
        return true;
    }
    // The pages of [lo, hi) are not needed again: their page-table entries go now (whole pages inside the range only), on
    // the calling thread — several threads may do this for different ranges side by side, whereas the unmapping at the end of a
    // process is one thread's work (0.25 s for the 6 GB of text a ten-million-candidate dosplitalign run has mapped).
    void drop_pages(size_t lo, size_t hi) const
    {
        if (!mapped || hi <= lo) return;
        const uintptr_t a = ((uintptr_t)p + lo + 4095) & ~(uintptr_t)4095, b = ((uintptr_t)p + hi) & ~(uintptr_t)4095;
        if (b > a) (void)madvise((void*)a, b - a, MADV_DONTNEED);
    }
    // one past the newline of the line that contains pos (or the end of the text)
    size_t line_end(size_t pos) const
    {
        const char* nl = (const char*)memchr(p + pos, '\n', n - pos);
        return nl ? (size_t)(nl - p) + 1 : n;
    }
    // [lo, hi) cut into `pieces` ranges that begin at line starts (lo must be one)
    std::vector<size_t> cut_lines(size_t lo, size_t hi, unsigned pieces) const
    {
        std::vector<size_t> cut(pieces + 1, hi);
        cut[0] = lo;
        for (unsigned t = 1; t < pieces; ++t) {
            size_t pos = std::max(cut[t - 1], lo + (hi - lo) / pieces * t);
            if (pos > lo && pos < hi && p[pos - 1] != '\n') pos = std::min(hi, line_end(pos));
            cut[t] = pos;
        }
        return cut;
    }
};

// ---- cluster files (the text between clustermatepairs, setcover and dosplitalign) on a team of threads -----------------------
inline void append_int(std::string& buf, long long v);
struct Fields {
    const char* p[16];
    int n = 0;
    const char* end = nullptr;
    size_t len(int k) const { return (size_t)((k + 1 < n ? p[k + 1] - 1 : end) - p[k]); }
    std::string str(int k) const { return std::string(p[k], len(k)); }
};

// up to `want` leading tab-separated fields of a line (more are left inside the last one's tail)
inline void split_fields(const char* line, size_t len, int want, Fields& f)
{
    f.n = 0;
    f.end = line + len;
    f.p[f.n++] = line;
    for (const char* q = line; f.n < want;) {
        const char* tab = (const char*)memchr(q, '\t', (size_t)(f.end - q));
        if (!tab) break;
        f.p[f.n++] = q = tab + 1;
    }
}

// what a step would have died of: thrown, so that a team of threads working on pieces of the input can report the FIRST one in
// file order, as the one-pass scripts do; main() turns it into die()
struct GlueError { std::string msg; };
[[noreturn]] inline void fail(const std::string& msg) { throw GlueError{msg}; }

inline long long num(const Fields& f, int k, const char* what)
{
    int v;
    if (k >= f.n || !field_int(f.p[k], f.len(k), v)) fail(std::string("Error: bad ") + what + " '" + (k < f.n ? f.str(k) : std::string()) + "'");
    return v;
}

// The cluster files these steps read are hundreds of megabytes of text in runs of lines with one cluster id.  The input
// (stdin: mapped if it is a file) is cut into pieces that begin where the first column changes, every thread of a team works
// through its piece as the one-pass script would, and the pieces' outputs and errors are taken in file order.
struct ClusterPieces {
    MappedText text;
    std::vector<size_t> cut;
    unsigned pieces = 1;
    void load(const std::string& name)                                    // "-" = stdin
    {
        if (!text.try_load(name, true, false)) fail("Error: Unable to open clusters file " + name);
        size_t min_bytes = (size_t)1 << 22;                                 // below this one thread is as fast (DEFUSE_GLUE_MIN_BYTES: tests)
        if (const char* e = std::getenv("DEFUSE_GLUE_MIN_BYTES")) min_bytes = (size_t)std::atoll(e);
        const unsigned want = text.size() < std::max<size_t>(min_bytes, 1) ? 1u : host_threads();
        cut = text.cut_lines(0, text.size(), want);
        pieces = want;
        auto first_id = [&](size_t pos, bool& ok) -> long long {          // numeric value of the first column of the line at pos
            const size_t e = text.line_end(pos);
            size_t n = e - pos;
            if (n && text[e - 1] == '\n') --n;
            Fields f;
            split_fields(text.data() + pos, n, 2, f);
            int v = 0;
            ok = field_int(f.p[0], f.len(0), v);
            return v;
        };
        for (unsigned t = 1; t < pieces; ++t) {                             // forward to the next change of the cluster id
            size_t pos = std::max(cut[t], cut[t - 1]);
            if (pos > 0 && pos < text.size()) {
                size_t prev = pos - 1;                                      // start of the line before pos
                while (prev > 0 && text[prev - 1] != '\n') --prev;
                bool ok_a, ok_b;
                long long a = first_id(prev, ok_a);
                while (pos < text.size()) {
                    const long long b = first_id(pos, ok_b);
                    if (!ok_a || !ok_b || a != b) break;                   // (a bad line ends the search: its piece reports it)
                    pos = text.line_end(pos);
                }
            }
            cut[t] = pos;
        }
        for (unsigned t = 1; t <= pieces; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    }
    // fn(piece, line, len) for every line of every piece, pieces side by side; the first error in file order is thrown
    template <class Fn, class Done>
    void run(Fn&& fn, Done&& done)
    {
        std::vector<std::string> err(pieces);
        std::vector<char> failed(pieces, 0);
        run_threads(pieces, [&](unsigned t) {
            try {
                size_t pos = cut[t];
                const size_t hi = cut[t + 1];
                while (pos < hi) {
                    const size_t e = text.line_end(pos);
                    size_t n = e - pos;
                    if (n && text[e - 1] == '\n') --n;
                    fn(t, text.data() + pos, n);
                    pos = e;
                }
                done(t);
            } catch (const GlueError& g) {
                err[t] = g.msg;
                failed[t] = 1;
            }
        });
        for (unsigned t = 0; t < pieces; ++t)
            if (failed[t]) fail(err[t]);
    }
};


// scripts/get_align_regions.pl:14-53: per (cluster, end) the reference name and strand of its last line and the extent of all
// its alignments; clusters ascending, end 0 then 1; a cluster without exactly two ends is an error.  (bin/defuse_glue
// get_align_regions and the fused mode of dosplitalign.)
inline std::string align_regions_text(ClusterPieces& in)
{
    struct EndInfo { std::string ref, strand; long long start = 0, end = 0; bool have = false; };
    typedef std::map<long long, std::map<long long, EndInfo>> Clusters;
    std::vector<Clusters> part(in.pieces);
    in.run([&](unsigned t, const char* line, size_t len) {
        Fields f;
        split_fields(line, len, 9, f);
        if (f.n < 8) fail("Error: cluster line with fewer than 8 fields");
        const long long id = num(f, 0, "cluster id"), ce = num(f, 1, "cluster end"), start = num(f, 6, "start"), end = num(f, 7, "end");
        EndInfo& e = part[t][id][ce];
        e.ref.assign(f.p[4], f.len(4));
        e.strand.assign(f.p[5], f.len(5));
        if (!e.have) { e.start = start; e.end = end; e.have = true; }
        e.start = std::min(e.start, start);
        e.end = std::max(e.end, end);
    }, [](unsigned) {});
    // a cluster id that comes back in a later piece: name and strand of the LAST line, the extent of all of them
    Clusters& clusters = part[0];
    for (unsigned t = 1; t < in.pieces; ++t)
        for (auto& c : part[t]) {
            auto at = clusters.find(c.first);
            if (at == clusters.end()) { clusters.emplace_hint(clusters.end(), c.first, std::move(c.second)); continue; }
            for (auto& e : c.second) {
                EndInfo& into = at->second[e.first];
                const EndInfo& from = e.second;
                if (!into.have) { into = from; continue; }
                into.ref = from.ref;
                into.strand = from.strand;
                into.start = std::min(into.start, from.start);
                into.end = std::max(into.end, from.end);
            }
        }
    std::string out;
    for (const auto& c : clusters) {
        if (c.second.size() != 2) fail("Error: Did not find 2 ends for cluster " + std::to_string(c.first));
        for (const auto& e : c.second) {
            append_int(out, c.first); out += '\t';
            append_int(out, e.first); out += '\t';
            out += e.second.ref; out += '\t';
            out += e.second.strand; out += '\t';
            append_int(out, e.second.start); out += '\t';
            append_int(out, e.second.end); out += '\n';
        }
    }
    return out;
}

// An output file written in order by several threads: the texts of one round get consecutive ranges of the file (a prefix
// sum of their sizes) and every thread copies its own with pwrite, so the copy into the page cache runs side by side.
class OrderedFileWriter {
public:
    bool open_file(const std::string& name)
    {
        fd_ = open(name.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd_ < 0) return false;
        struct stat st;
        seekable_ = fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);      // /dev/stdout, FIFOs, process substitution: plain ordered writes
        return true;
    }
    void write_round(const std::vector<std::string>& texts, unsigned threads)
    {
        if (!seekable_) {
            for (const std::string& t : texts) ok_ = ok_ && write_all(t.data(), t.size(), -1);
            return;
        }
        std::vector<off_t> at(texts.size() + 1, pos_);
        for (size_t k = 0; k < texts.size(); ++k) at[k + 1] = at[k] + (off_t)texts[k].size();
        std::vector<int> bad(texts.size(), 0);
        const unsigned n = std::min<unsigned>(std::max(1u, threads), (unsigned)std::max<size_t>(1, texts.size()));
        run_threads(n, [&](unsigned t) {
            for (size_t k = t; k < texts.size(); k += n)
                if (!write_all(texts[k].data(), texts[k].size(), at[k])) bad[k] = 1;
        });
        for (int b : bad) ok_ = ok_ && !b;
        pos_ = at[texts.size()];
    }
    // The same round written behind the caller's back: the texts are taken over, a writer thread copies them into the file
    // while the caller formats the next round (one round in flight; the next call, and close_file, wait for it).  The copy
    // into the page cache is one core's work whatever the number of writers (the file's lock), so a tool whose output is
    // gigabytes overlaps it with its formatting instead of alternating the two.
    void write_round_async(std::vector<std::string>&& texts, unsigned threads)
    {
        std::unique_lock<std::mutex> lk(am_);
        if (!athread_.joinable()) athread_ = std::thread([this] { async_loop(); });
        acv_.wait(lk, [&] { return !ahave_; });
        atexts_ = std::move(texts);
        athreads_ = threads;
        ahave_ = true;
        acv_.notify_all();
    }
    void wait_async()
    {
        std::unique_lock<std::mutex> lk(am_);
        acv_.wait(lk, [&] { return !ahave_ && !abusy_; });
    }
    // For a caller whose own threads write: reserves consecutive ranges for parts of the given sizes (in order) and returns
    // where each begins; every thread then calls write_part with its own.  Not seekable: begin() returns false and the caller
    // writes the parts in order with append().
    bool seekable() const { return seekable_; }
    std::vector<off_t> reserve_parts(const std::vector<size_t>& sizes)
    {
        std::vector<off_t> at(sizes.size());
        for (size_t k = 0; k < sizes.size(); ++k) { at[k] = pos_; pos_ += (off_t)sizes[k]; }
        return at;
    }
    void write_part(const char* p, size_t n, off_t at)
    {
        if (!write_all(p, n, at)) failed_.store(true, std::memory_order_relaxed);
    }
    void append(const char* p, size_t n) { ok_ = ok_ && write_all(p, n, -1); }
    bool close_file()
    {
        if (athread_.joinable()) {
            wait_async();
            {
                std::lock_guard<std::mutex> lk(am_);
                astop_ = true;
            }
            acv_.notify_all();
            athread_.join();
        }
        if (failed_.load()) ok_ = false;
        if (fd_ >= 0 && close(fd_) != 0) ok_ = false;
        fd_ = -1;
        return ok_;
    }
private:
    // at >= 0: pwrite at that offset; at < 0: write at the file position.  Short writes continue, EINTR retries.
    bool write_all(const char* p, size_t left, off_t at)
    {
        while (left) {
            const ssize_t w = at >= 0 ? pwrite(fd_, p, left, at) : write(fd_, p, left);
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) return false;
            p += w; left -= (size_t)w;
            if (at >= 0) at += w;
        }
        return true;
    }
    void async_loop()
    {
        for (;;) {
            std::vector<std::string> texts;
            unsigned threads;
            {
                std::unique_lock<std::mutex> lk(am_);
                acv_.wait(lk, [&] { return ahave_ || astop_; });
                if (!ahave_) return;
                texts.swap(atexts_);
                threads = athreads_;
                ahave_ = false;
                abusy_ = true;
            }
            acv_.notify_all();
            write_round(texts, threads);
            texts.clear();
            {
                std::lock_guard<std::mutex> lk(am_);
                abusy_ = false;
            }
            acv_.notify_all();
        }
    }
    int fd_ = -1;
    off_t pos_ = 0;
    bool ok_ = true, seekable_ = true;
    std::atomic<bool> failed_{false};
    std::thread athread_;
    std::mutex am_;
    std::condition_variable acv_;
    std::vector<std::string> atexts_;
    unsigned athreads_ = 1;
    bool ahave_ = false, abusy_ = false, astop_ = false;
};

// A set of 64-bit keys for the hot de-duplication loops: open addressing, linear probing, grows at half full.
// insert() returns true when the key was new.  ~0 is reserved as the empty marker.
class FlatSet64 {
public:
    explicit FlatSet64(size_t cap_pow2 = 1024) : slots_(cap_pow2, EMPTY), mask_(cap_pow2 - 1) {}
    bool insert(uint64_t key)
    {
        if ((n_ + 1) * 2 > slots_.size()) grow();
        size_t i = hash(key) & mask_;
        for (;; i = (i + 1) & mask_) {
            if (slots_[i] == key) return false;
            if (slots_[i] == EMPTY) { slots_[i] = key; ++n_; return true; }
        }
    }
    static uint64_t hash(uint64_t x)
    {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        return x;
    }
    // the cache line insert(key) will look at first (a caller with many keys at hand asks for a few ahead of inserting them;
    // only useful when no insert in between can grow the table: reserve() first)
    void prefetch(uint64_t key) const { __builtin_prefetch(&slots_[hash(key) & mask_], 1, 1); }
    size_t size() const { return n_; }
    // room for n keys without another rehash on the way there
    void reserve(size_t n)
    {
        size_t want = slots_.size();
        while (want < 2 * n + 2) want *= 2;
        if (want > slots_.size()) grow(want);
    }
private:
    static constexpr uint64_t EMPTY = ~(uint64_t)0;
    void grow(size_t to = 0)
    {
        std::vector<uint64_t> old(to ? to : slots_.size() * 2, EMPTY);
        old.swap(slots_);
        mask_ = slots_.size() - 1;
        n_ = 0;
        for (uint64_t k : old)
            if (k != EMPTY) insert(k);
    }
    std::vector<uint64_t> slots_;
    size_t mask_, n_ = 0;
};

// decimal text of an integer appended to a buffer (what operator<< prints for an int)
inline void append_int(std::string& buf, long long v)
{
    char tmp[24];
    char* p = tmp + sizeof tmp;
    unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do { *--p = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    buf.append(p, (size_t)(tmp + sizeof tmp - p));
}

// the same text written at p (room for 11 bytes); returns one past the last byte
inline char* put_int(char* p, int v)
{
    static const char D2[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
                             "707172737475767778798081828384858687888990919293949596979899";
    uint32_t u = (uint32_t)v;
    if (v < 0) { *p++ = '-'; u = 0u - u; }
    const unsigned nd = u < 10 ? 1 : u < 100 ? 2 : u < 1000 ? 3 : u < 10000 ? 4 : u < 100000 ? 5 : u < 1000000 ? 6 : u < 10000000 ? 7 : u < 100000000 ? 8 : u < 1000000000 ? 9 : 10;
    char* const e = p + nd;
    char* q = e;
    while (u >= 100) {
        const unsigned r = u % 100;
        u /= 100;
        q -= 2;
        std::memcpy(q, D2 + 2 * r, 2);
    }
    if (u >= 10) std::memcpy(q - 2, D2 + 2 * u, 2);
    else q[-1] = (char)('0' + u);
    return e;
}

// tools/Common.cpp:32-54
inline void ReverseComplement(std::string& seq)
{
    std::reverse(seq.begin(), seq.end());
    for (char& c : seq) {
        switch (c) {
            case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'T': c = 'A'; break; case 'G': c = 'C'; break;
            case 'a': c = 't'; break; case 'c': c = 'g'; break; case 't': c = 'a'; break; case 'g': c = 'c'; break;
        }
    }
}

inline int InterpretStrand(const std::string& s)   // tools/Common.cpp:91-106
{
    if (s == "+") return PlusStrand;
    if (s == "-") return MinusStrand;
    die("Error: Unable to intepret strand " + s);
}

// ReadID / ClusterID bit-field unions (tools/Common.h:192-218): low 31 bits index, top bit end.
inline int pack_id(int index, int end) { return (int)(((uint32_t)index & 0x7FFFFFFFu) | ((uint32_t)end << 31)); }

// ---------------------------------------------------------------------------------------------
// Command line in the style of the reference's TCLAP usage (all arguments required, short or long
// name, "--" ends parsing, -h/--help, --version; parse error -> "PARSE ERROR:" on stderr, exit 1;
// include/tclap/CmdLine.h:306-326,370-416, StdOutput.h:130-152).
// ---------------------------------------------------------------------------------------------
class CmdLine {
public:
    struct Arg { std::string flag, name, desc, type; std::string value; bool set = false; bool required = true; std::string kind; };
    CmdLine(std::string message, std::string version = "none") : message_(std::move(message)), version_(std::move(version)) {}
    // type: what the usage text shows ("string", "integer", "float", ...: TCLAP's type description, free text in the
    // reference); kind: how the value is checked — "int", "float" or "string" (default: by the type text)
    void add(const std::string& flag, const std::string& name, const std::string& desc, const std::string& type, const std::string& kind = "")
    {
        args_.push_back({flag, name, desc, type, "", false, true, kind.empty() ? kind_of(type) : kind});
    }
    // an argument that may be left out (tclap ValueArg with req = false): keeps its default then
    void add_optional(const std::string& flag, const std::string& name, const std::string& desc, const std::string& type,
                      const std::string& default_value)
    {
        args_.push_back({flag, name, desc, type, default_value, false, false, kind_of(type)});
    }
    // a switch without a value (tclap SwitchArg, long form only when flag is empty): is_set(name) tells whether it was given
    void add_switch(const std::string& flag, const std::string& name, const std::string& desc)
    {
        args_.push_back({flag, name, desc, "switch", "", false, false, "switch"});
    }
    bool is_set(const std::string& name) const { return get(name).set; }
    void parse(int argc, char** argv)
    {
        prog_ = argc > 0 ? argv[0] : "prog";
        size_t slash = prog_.find_last_of('/');
        if (slash != std::string::npos) prog_ = prog_.substr(slash + 1);
        for (int i = 1; i < argc; ++i) {
            std::string tok = argv[i];
            if (tok == "--") break;
            if (tok == "-h" || tok == "--help") { usage(std::cout); std::exit(0); }
            if (tok == "--version") { std::cout << std::endl << prog_ << "  version: " << version_ << std::endl << std::endl; std::exit(0); }
            Arg* a = find(tok);
            if (!a) fail("Argument: " + tok, "Couldn't find match for argument");
            if (a->set) fail("Argument: " + id(*a), "Argument already set!");
            if (a->type == "switch") { a->set = true; continue; }
            if (i + 1 >= argc) fail("Argument: " + id(*a), "Missing a value for this argument!");
            a->value = argv[++i];
            a->set = true;
            if (a->kind == "int") { int v; if (!strict_int(a->value, v)) fail("Argument: " + id(*a), "Couldn't read argument value from string '" + a->value + "'"); }
            if (a->kind == "float") { double v; if (!strict_double(a->value, v)) fail("Argument: " + id(*a), "Couldn't read argument value from string '" + a->value + "'"); }
        }
        for (const Arg& a : args_)
            if (!a.set && a.required) fail(" ", "One or more required arguments missing!");      // the vendored TCLAP's text and its one-space argument id
    }
    std::string str(const std::string& name) const { return get(name).value; }
    int integer(const std::string& name) const { int v = 0; strict_int(get(name).value, v); return v; }
    double real(const std::string& name) const { double v = 0; strict_double(get(name).value, v); return v; }

private:
    static std::string kind_of(const std::string& type) { return (type == "integer" || type == "int") ? "int" : (type == "float" || type == "double") ? "float" : "string"; }
    static bool strict_int(const std::string& s, int& v)
    {
        std::istringstream is(s);
        is >> v;
        return !is.fail() && is.peek() == EOF;
    }
    static bool strict_double(const std::string& s, double& v)
    {
        std::istringstream is(s);
        is >> v;
        return !is.fail() && is.peek() == EOF;
    }
    static std::string id(const Arg& a) { return a.flag.empty() ? "(--" + a.name + ")" : "-" + a.flag + " (--" + a.name + ")"; }
    Arg* find(const std::string& tok)
    {
        for (Arg& a : args_)
            if ((!a.flag.empty() && tok == "-" + a.flag) || tok == "--" + a.name) return &a;
        return nullptr;
    }
    const Arg& get(const std::string& name) const
    {
        for (const Arg& a : args_)
            if (a.name == name) return a;
        die("internal error: unknown argument " + name);
    }
    // The vendored TCLAP (include/tclap of the reference) keeps its arguments in a list it pushes to the FRONT of, so the
    // usage texts name them in reverse order of definition, the built-in --, --version, -h (defined first) last; the program
    // name is followed by two spaces (oracle/tclap_ref.cpp runs that library; tests/test_cli_ref.py compares).
    void short_usage(std::ostream& os) const
    {
        std::string line = prog_ + " ";
        for (auto it = args_.rbegin(); it != args_.rend(); ++it) {
            const Arg& a = *it;
            if (a.type == "switch") { line += " [" + (a.flag.empty() ? "--" + a.name : "-" + a.flag) + "]"; continue; }
            line += (a.required ? " -" : " [-") + a.flag + " <" + a.type + ">" + (a.required ? "" : "]");
        }
        line += " [--] [--version] [-h]";
        space_print(os, line, 3, std::min<int>((int)prog_.size() + 2, 75 / 2));
    }
    // TCLAP's line folding (StdOutput::spacePrint with its width of 75): a text that does not fit is cut into lines of at most
    // 75 - indent characters, each cut moved back to the nearest position in front of a blank, comma or bar (a word longer than
    // a line is cut where the line ends; a newline in the text ends a line early); the first line is indented by `indent`, the
    // following ones by `indent + second` and that much shorter; blanks at the start of a continuation line are dropped.
    static void space_print(std::ostream& os, const std::string& s, int indent, int second)
    {
        const int len = (int)s.size(), width = 75;
        if (len + indent <= width) { os << std::string((size_t)indent, ' ') << s << std::endl; return; }
        int allowed = width - indent, start = 0;
        auto at = [&](int k) { return k >= 0 && k < len ? s[(size_t)k] : '\0'; };
        while (start < len) {
            int n = std::min(len - start, allowed);
            if (n == allowed)
                while (n >= 0 && at(start + n) != ' ' && at(start + n) != ',' && at(start + n) != '|') --n;
            if (n <= 0) n = allowed;
            for (int i = 0; i < n; ++i)
                if (at(start + i) == '\n') n = i + 1;
            os << std::string((size_t)indent, ' ');
            if (start == 0) { indent += second; allowed -= second; }
            os << s.substr((size_t)start, (size_t)n) << std::endl;
            while (start < len && at(start + n) == ' ') ++start;
            start += n;
        }
    }
    void usage(std::ostream& os) const
    {
        os << std::endl << "USAGE: " << std::endl << std::endl;
        short_usage(os);
        os << std::endl << std::endl << "Where: " << std::endl << std::endl;
        for (auto it = args_.rbegin(); it != args_.rend(); ++it) {
            const Arg& a = *it;
            if (a.type == "switch") space_print(os, (a.flag.empty() ? std::string() : "-" + a.flag + ",  ") + "--" + a.name, 3, 3);
            else space_print(os, "-" + a.flag + " <" + a.type + ">,  --" + a.name + " <" + a.type + ">", 3, 3);
            space_print(os, (a.required ? "(required)  " : "") + a.desc, 5, 0);
            os << std::endl;
        }
        space_print(os, "--,  --ignore_rest", 3, 3);
        space_print(os, "Ignores the rest of the labeled arguments following this flag.", 5, 0);
        os << std::endl;
        space_print(os, "--version", 3, 3);
        space_print(os, "Displays version information and exits.", 5, 0);
        os << std::endl;
        space_print(os, "-h,  --help", 3, 3);
        space_print(os, "Displays usage information and exits.", 5, 0);
        os << std::endl << std::endl;
        space_print(os, message_, 3, 0);
        os << std::endl;
    }
    // StdOutput::failure of the vendored TCLAP: the message on stderr — with the brief usage line itself on STDOUT, as that
    // version prints it — then exit(1)
    [[noreturn]] void fail(const std::string& arg_id, const std::string& error) const
    {
        std::cerr << "PARSE ERROR: " << arg_id << std::endl << "             " << error << std::endl << std::endl;
        std::cerr << "Brief USAGE: " << std::endl;
        std::cerr.flush();
        short_usage(std::cout);
        std::cout.flush();
        std::cerr << std::endl << "For complete USAGE and HELP type: " << std::endl << "   " << prog_ << " --help" << std::endl << std::endl;
        std::exit(1);
    }
    std::string message_, version_, prog_;
    std::vector<Arg> args_;
};

// ---------------------------------------------------------------------------------------------
// FASTA random access: tools/FastaIndex.cpp:18-61 on top of a .fai index with the semantics of
// external/samtools-0.1.8/faidx.c (fai_load builds "<fasta>.fai" when missing, :260-300; fai_fetch
// :305-357: 1-based inclusive region clipped to the sequence, non-graph characters skipped).
// ---------------------------------------------------------------------------------------------
class FastaIndex {
public:
    void Open(const std::string& fasta)
    {
        // samtools 0.1.8's faidx also reads RAZF-compressed FASTA (<fasta>.rz with its own index, faidx.c:260-303); deFuse's
        // pipeline never produces one (scripts/defuse_create_ref.pl writes plain FASTA) and this reader does not support it
        if (fasta.size() > 3 && fasta.compare(fasta.size() - 3, 3, ".rz") == 0)
            die("Error: RAZF-compressed FASTA (" + fasta + ") is not supported, give the plain FASTA file");
        const std::string fai = fasta + ".fai";
        std::ifstream in(fai.c_str(), std::ios::binary);
        if (!in.good()) {
            std::cerr << "[fai_load] build FASTA index." << std::endl;
            build(fasta, fai);
            in.open(fai.c_str(), std::ios::binary);
            if (!in.good()) die("[fai_load] fail to open FASTA index.");
        }
        std::string line;
        while (std::getline(in, line)) {
            size_t e = 0;
            while (e < line.size() && std::isgraph((unsigned char)line[e])) ++e;
            Entry en;
            std::istringstream rest(e < line.size() ? line.substr(e + 1) : "");
            rest >> en.len >> en.offset >> en.line_blen >> en.line_len;
            index_[line.substr(0, e)] = en;
        }
        fd_ = open(fasta.c_str(), O_RDONLY);
        if (fd_ < 0) die("[fai_load] fail to open FASTA file.");
        // a regular file is also mapped: a window is then a few hundred bytes copied from the page cache instead of a system
        // call and a buffer of its own (a hundred thousand fusions are four hundred thousand windows per dosplitalign process)
        struct stat st;
        if (fstat(fd_, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (m != MAP_FAILED) { map_ = (const char*)m; map_n_ = (size_t)st.st_size; }
        }
    }
    ~FastaIndex()
    {
        if (map_) munmap((void*)map_, map_n_);
        if (fd_ >= 0) close(fd_);
    }
    FastaIndex() = default;
    FastaIndex(const FastaIndex&) = delete;
    FastaIndex& operator=(const FastaIndex&) = delete;

    // FastaIndex::Get: start/length are in-out (tools/FastaIndex.h:24): clipped values flow back.
    void Get(const std::string& reference, int strand, int& start, int& length, std::string& sequence) const
    {
        if (length < 0) { sequence.clear(); return; }
        if (start < 1) { length -= 1 - start; start = 1; }
        const int end = start + length - 1;
        auto it = index_.find(reference);
        if (it == index_.end()) die("Error: Unable to find sequence for " + reference);
        const Entry& en = it->second;
        long long beg = start, e = end;       // the reference formats "name:start-end" and parses it back with atoi
        if (beg > 0) --beg;
        if (beg >= en.len) beg = en.len;
        // faidx.c:337 compares the int `end` with the unsigned field val.len: a negative end (a window wholly in front of the
        // sequence, "name:1--12") turns into a huge unsigned number and is clipped to the sequence END (tests/test_faidx_ref.py)
        if (e < 0 || e >= en.len) e = en.len;
        if (beg > e) beg = e;
        sequence.clear();
        if (e > beg && en.line_blen > 0) {
            // one pread of the span that holds the wanted bases (thread-safe: tasks are set up side by side), then the
            // non-graph bytes (line ends) are dropped as fai_fetch does
            const long long first = en.offset + beg / en.line_blen * en.line_len + beg % en.line_blen;
            const long long extra = std::max(1, en.line_len - en.line_blen);
            long long span = (e - beg) + ((e - beg) / en.line_blen + 2) * extra;
            sequence.reserve((size_t)(e - beg));
            std::vector<char> buf;
            long long at = first;
            if (map_) {
                // the same bytes, straight from the mapping: every byte that is not a graph character is dropped, as above
                const size_t want = (size_t)(e - beg);
                sequence.resize(want);
                char* out = &sequence[0];
                size_t have = 0;
                for (size_t k = (size_t)std::min<long long>(first, (long long)map_n_); k < map_n_ && have < want; ++k) {
                    const unsigned char c = (unsigned char)map_[k];
                    out[have] = (char)c;
                    have += (c > 32 && c < 127) ? 1 : 0;                  // isgraph in the C locale
                }
                sequence.resize(have);
                at = -1;
            }
            while (at >= 0 && (long long)sequence.size() < e - beg) {
                buf.resize((size_t)span);
                const ssize_t got = pread(fd_, buf.data(), buf.size(), (off_t)at);
                if (got <= 0) break;                                  // end of file: a truncated FASTA gives a short sequence, as there
                for (ssize_t k = 0; k < got && (long long)sequence.size() < e - beg; ++k)
                    if (std::isgraph((unsigned char)buf[(size_t)k])) sequence.push_back(buf[(size_t)k]);
                at += got;
                span = std::max<long long>(4096, e - beg - (long long)sequence.size() + 64);
            }
        }
        length = (int)sequence.size();
        if (strand == MinusStrand) ReverseComplement(sequence);
    }

private:
    const char* map_ = nullptr;
    size_t map_n_ = 0;
    struct Entry { long long len = 0, offset = 0; int line_blen = 0, line_len = 0; };
    static void build(const std::string& fasta, const std::string& fai)
    {
        // fai_build_core (faidx.c:62-139), character by character over the mapped file, with its state machine (1 = just after
        // a header, 0 = inside a sequence, 2 = a shorter line was seen, 3 = a line after that) and its refusals
        MappedText in;
        {
            const int fd = open(fasta.c_str(), O_RDONLY);
            if (fd < 0) die("[fai_build] fail to open the FASTA file " + fasta);
            close(fd);
        }
        in.load(fasta, "[fai_build] fail to open the FASTA file ");
        std::string text, name;
        long long len = -1, offset = 0;
        int line_len = -1, line_blen = -1, state = 0;
        auto insert = [&]() { text += name + "\t" + std::to_string(len) + "\t" + std::to_string(offset) + "\t" + std::to_string(line_blen) + "\t" + std::to_string(line_len) + "\n"; };
        const size_t n = in.size();
        size_t pos = 0;
        while (pos < n) {
            char c = in[pos++];
            if (c == '\n') {                                   // an empty line
                if (state == 1) { offset = (long long)pos; continue; }
                if ((state == 0 && len < 0) || state == 2) continue;
            }
            if (c == '>') {                                    // header
                if (len >= 0) insert();
                name.clear();
                bool more = false;
                while (pos < n) {
                    c = in[pos++];
                    if (std::isspace((unsigned char)c)) { more = true; break; }
                    name += c;
                }
                if (!more) die("[fai_build_core] the last entry has no sequence");
                if (c != '\n') while (pos < n && in[pos++] != '\n') {}
                state = 1;
                len = 0;
                offset = (long long)pos;
            } else {
                if (state == 3) die("[fai_build_core] inlined empty line is not allowed in sequence '" + name + "'.");
                if (state == 2) state = 3;
                int l1 = 0, l2 = 0;
                for (;;) {
                    ++l1;
                    if (std::isgraph((unsigned char)c)) ++l2;
                    if (pos >= n) break;
                    c = in[pos++];
                    if (c == '\n') break;
                }
                if (state == 3 && l2) die("[fai_build_core] different line length in sequence '" + name + "'.");
                ++l1;
                len += l2;
                if (l2 >= 0x10000) die("[fai_build_core] line length exceeds 65535 in sequence '" + name + "'.");
                if (state == 1) { line_len = l1; line_blen = l2; state = 0; }
                else if (state == 0 && (l1 != line_len || l2 != line_blen)) state = 2;
            }
        }
        insert();                                              // (as the reference: also for a file without any header)
        std::ofstream out(fai.c_str(), std::ios::binary);
        if (!out.good()) die("[fai_build] fail to write FASTA index " + fai);
        out << text;
    }
    std::unordered_map<std::string, Entry> index_;
    int fd_ = -1;
};

// ---------------------------------------------------------------------------------------------
// tools/ExonRegions.cpp (the parts the split-alignment path reaches).
// ---------------------------------------------------------------------------------------------
class ExonRegions {
public:
    bool Read(std::istream& in)   // :21-112
    {
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            std::vector<std::string> f = split_tabs(line);
            if (f.size() < 6) continue;
            const std::string &gene = f[0], &transcript = f[1], &chromosome = f[2], &strand = f[3];
            std::vector<Region> exons;
            for (size_t k = 5; k < f.size(); k += 2) {
                Region ex;
                if (!lexical_int(f[k - 1], ex.start) || !lexical_int(f[k], ex.end)) {
                    std::cout << "Failed to interpret exon:" << std::endl << line << std::endl;
                    std::exit(1);
                }
                exons.push_back(ex);
            }
            const int s = InterpretStrand(strand);
            int total = 0;
            for (const Region& r : exons) total += r.end - r.start + 1;
            chromosome_[transcript] = chromosome;
            strand_[transcript] = s;
            exons_[transcript] = exons;
            length_[transcript] = total;
            gene_[transcript] = gene;
            gene_transcripts_[gene].push_back(transcript);
            exons_str_[PlusStrand][transcript] = exons;
            std::vector<Region> minus;   // TransformExons :114-124
            for (auto it = exons.rbegin(); it != exons.rend(); ++it) minus.push_back(Region{-it->end, -it->start});
            exons_str_[MinusStrand][transcript] = minus;
            region_[transcript] = Region{exons.front().start, exons.back().end};
            for (int b = region_[transcript].start / kBin; b <= region_[transcript].end / kBin; ++b)
                lookup_[chromosome][b].push_back(transcript);
        }
        return true;
    }
    bool IsTranscript(const std::string& t) const { return gene_.count(t) != 0; }
    // GetGenes (:126-129): the reference lists its unordered_set of gene names; canonical order = ascending
    std::vector<std::string> GetGenes() const
    {
        std::vector<std::string> g;
        for (const auto& kv : gene_transcripts_) g.push_back(kv.first);
        return g;
    }
    const std::vector<std::string>& GetGeneTranscripts(const std::string& gene) const { return gene_transcripts_.at(gene); }   // file order
    int GetTranscriptLength(const std::string& t) const
    {
        auto it = length_.find(t);
        if (it == length_.end()) die("Error: Data mismatch, unable to find length for transcript " + t);
        return it->second;
    }
    const std::string& GetTranscriptGene(const std::string& t) const
    {
        auto it = gene_.find(t);
        if (it == gene_.end()) die("Error: Data mismatch, unable to find gene for transcript " + t);
        return it->second;
    }
    void GetRegionTranscripts(const std::string& chromosome, const Region& region, std::vector<std::string>& out) const   // :131-161
    {
        auto ci = lookup_.find(chromosome);
        if (ci == lookup_.end()) die("Error: Data mismatch, invalid chromosome " + chromosome);
        std::set<std::string> uniq;   // canonical order: ascending
        for (int b = region.start / kBin; b <= region.end / kBin; ++b) {
            auto bi = ci->second.find(b);
            if (bi == ci->second.end()) continue;
            for (const std::string& t : bi->second) {
                const Region& r = region_.at(t);
                if (!(r.end < region.start || r.start > region.end)) uniq.insert(t);
            }
        }
        out.insert(out.end(), uniq.begin(), uniq.end());
    }
    bool RemapTranscriptToGenome(const std::string& t, int strand, int position, std::string& chrom, int& rstrand, int& rpos) const   // :258-302
    {
        auto ei = exons_.find(t);
        if (ei == exons_.end() || ei->second.empty()) die("Error: Data mismatch, unable to find transcript " + t);
        const std::vector<Region>& ex = ei->second;
        const int tlen = length_.at(t), tstrand = strand_.at(t);
        chrom = chromosome_.at(t);
        rstrand = (tstrand == strand) ? PlusStrand : MinusStrand;
        if (tstrand == MinusStrand) position = tlen - position + 1;
        int off = 0;
        for (const Region& r : ex) {
            const int n = r.end - r.start + 1;
            if (position <= off + n) { rpos = position - (off + 1) + r.start; return true; }
            off += n;
        }
        rpos = position - tlen + ex.back().end;
        return true;
    }
    bool RemapThroughTranscript(const std::string& t, int position, int strand, int ext_min, int ext_max, int& rstrand, int& start, int& end) const   // :421-482
    {
        auto ei = exons_.find(t);
        if (ei == exons_.end() || ei->second.empty()) die("Error: Data mismatch, unable to find transcript " + t);
        const std::vector<Region>& ex = exons_str_[strand].at(t);
        const int tlen = length_.at(t), tstrand = strand_.at(t);
        rstrand = (strand == tstrand) ? PlusStrand : MinusStrand;
        const int sp = (strand == PlusStrand) ? position : -position;
        if (sp > ex.back().end) return false;
        int off = 0;
        for (const Region& r : ex) {
            const int n = r.end - r.start + 1;
            if (sp <= r.end) {
                const int rs = sp - r.start + ext_min + 1, re = sp - r.start + ext_max + 1;
                if (re < 1) return false;
                start = std::max(1, rs) + off;
                end = std::max(1, re) + off;
                break;
            }
            off += n;
        }
        if (end < 1 || start > tlen) return false;
        if (strand != tstrand) {
            start = tlen - start + 1;
            end = tlen - end + 1;
            std::swap(start, end);
        }
        return true;
    }

private:
    static constexpr int kBin = 100000;   // :19
    std::unordered_map<std::string, std::string> chromosome_, gene_;
    std::unordered_map<std::string, int> strand_, length_;
    std::map<std::string, std::vector<std::string>> gene_transcripts_;
    std::unordered_map<std::string, std::vector<Region>> exons_;
    mutable std::unordered_map<std::string, std::vector<Region>> exons_str_[2];
    std::unordered_map<std::string, Region> region_;
    std::unordered_map<std::string, std::unordered_map<int, std::vector<std::string>>> lookup_;
};

// tools/Common.cpp:117-131
inline bool ParseTranscriptID(const std::string& id, std::string& gene, std::string& transcript)
{
    std::vector<std::string> f = split_tabs(id, '|');
    if (f.size() < 2) return false;
    gene = f[0];
    transcript = f[1];
    return true;
}

// tools/Parsers.cpp:211-264
inline std::map<int, std::vector<Location>> ReadAlignRegionPairs(const std::string& filename)
{
    std::ifstream in(filename.c_str());
    if (!in.good()) die("Error: Unable to open align region pairs file " + filename);
    std::map<int, std::vector<Location>> pairs;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f = split_tabs(line);
        if (f.size() < 5) continue;
        int id, end;
        Location loc;
        bool ok = lexical_int(f[0], id) && lexical_int(f[1], end);
        if (ok && !(end == 0 || end == 1)) die("Error: DebugCheck pairEnd == 0 || pairEnd == 1 failed");
        if (ok) {
            loc.refName = f[2];
            loc.strand = InterpretStrand(f[3]);
            ok = f.size() > 5 && lexical_int(f[4], loc.start) && lexical_int(f[5], loc.end);
        }
        if (!ok) {
            std::cout << "Failed to interpret region:" << std::endl << line << std::endl;
            std::exit(1);
        }
        pairs[id].resize(2);
        pairs[id][end] = loc;
    }
    return pairs;
}

// ---------------------------------------------------------------------------------------------
// SplitAlignmentTask geometry: tools/SplitAlignment.cpp:31-175, :637-655, :657-686.
// ---------------------------------------------------------------------------------------------
struct SplitAlignmentTask {
    int mFusionID = 0;
    std::string mAlignRefName[2];
    int mAlignStrand[2] = {0, 0};
    int mSplitAlignSeqStart[2] = {0, 0}, mSplitAlignSeqLength[2] = {0, 0}, mSplitSeqStrand[2] = {0, 0};
    std::string mSplitAlignSeq[2], mSplitRemainderSeq[2];
    std::vector<Location> mMateRegions[2];

    static void CalculateBreakRegion(int minRead, int maxRead, int maxFrag, int alignStart, int alignEnd, int strand, int& breakStart, int& breakLength)
    {
        const int regionLength = alignEnd - alignStart + 1;
        const int push = std::min(maxRead, (int)(0.5 * regionLength));
        breakLength = maxFrag - regionLength - minRead + 2 * push;
        breakStart = (strand == PlusStrand) ? alignEnd - push + 1 : alignStart + push - 1;
    }

    bool Initialize(int id, const std::vector<Location>& alignPair, const FastaIndex& reference, const ExonRegions& exons,
                    double fragMean, double fragStdDev, int minReadLength, int maxReadLength)
    {
        mFusionID = id;
        const int minFrag = (int)(fragMean - 3 * fragStdDev), maxFrag = (int)(fragMean + 3 * fragStdDev);
        if (alignPair.size() != 2) {
            std::cerr << "Error: Incorrect input for SplitAlignment::Calculate()" << std::endl;
            return false;
        }
        for (int ce = 0; ce <= 1; ++ce) {
            const std::string& refName = alignPair[ce].refName;
            const int strand = alignPair[ce].strand, alignStart = alignPair[ce].start, alignEnd = alignPair[ce].end;
            mAlignRefName[ce] = refName;
            mAlignStrand[ce] = strand;
            const int refSeqStrand = (ce == 0) ? strand : 1 - strand;
            int breakStart, breakLength;
            CalculateBreakRegion(minReadLength, maxReadLength, maxFrag, alignStart, alignEnd, strand, breakStart, breakLength);
            mSplitSeqStrand[ce] = refSeqStrand;
            if (strand == PlusStrand) {
                mSplitAlignSeqStart[ce] = breakStart - maxReadLength;
                mSplitAlignSeqLength[ce] = breakLength + maxReadLength;
            } else {
                mSplitAlignSeqStart[ce] = breakStart - breakLength + 1;
                mSplitAlignSeqLength[ce] = breakLength + maxReadLength;
            }
            reference.Get(refName, refSeqStrand, mSplitAlignSeqStart[ce], mSplitAlignSeqLength[ce], mSplitAlignSeq[ce]);
            mSplitRemainderSeq[ce].clear();
            if (strand == PlusStrand) {
                if (alignStart < mSplitAlignSeqStart[ce]) {
                    int rs = alignStart, rl = mSplitAlignSeqStart[ce] - 1 - alignStart + 1;
                    reference.Get(refName, refSeqStrand, rs, rl, mSplitRemainderSeq[ce]);
                }
            } else if (alignEnd > mSplitAlignSeqStart[ce] + mSplitAlignSeqLength[ce] - 1) {
                int rs = mSplitAlignSeqStart[ce] + mSplitAlignSeqLength[ce], rl = alignEnd - rs + 1;
                reference.Get(refName, refSeqStrand, rs, rl, mSplitRemainderSeq[ce]);
            }
            std::string chromosome, gene, transcript;
            int genomeStrand, genomeBreakStart;
            if (ParseTranscriptID(refName, gene, transcript) && exons.IsTranscript(transcript)) {
                exons.RemapTranscriptToGenome(transcript, strand, breakStart, chromosome, genomeStrand, genomeBreakStart);
            } else {
                chromosome = refName;
                genomeStrand = strand;
                genomeBreakStart = breakStart;
            }
            const int mateMin = minFrag - breakLength - maxReadLength + 1, mateMax = maxFrag - minReadLength;
            Region mate;
            if (genomeStrand == PlusStrand) { mate.start = genomeBreakStart - mateMax; mate.end = genomeBreakStart - mateMin; }
            else { mate.start = genomeBreakStart + mateMin; mate.end = genomeBreakStart + mateMax; }
            mMateRegions[ce].push_back(Location{chromosome, genomeStrand, mate.start, mate.end});
            std::vector<std::string> transcripts;
            exons.GetRegionTranscripts(chromosome, mate, transcripts);
            for (const std::string& t : transcripts) {
                int rstrand, ms = 0, me = 0;
                if (exons.RemapThroughTranscript(t, genomeBreakStart, 1 - genomeStrand, mateMin, mateMax, rstrand, ms, me))
                    mMateRegions[ce].push_back(Location{exons.GetTranscriptGene(t) + "|" + t, 1 - rstrand, ms, me});
            }
        }
        return true;
    }
};

inline std::map<int, SplitAlignmentTask> CreateTasks(const std::string& fasta, const std::string& exonsFile, double fragMean,
                                                     double fragStdDev, int minRead, int maxRead,
                                                     const std::map<int, std::vector<Location>>& regions, unsigned threads = 0)
{
    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    auto clock_now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = clock_now();
    auto lap = [&](const char* what) {
        const double t = clock_now();
        if (timing) std::cerr << "[tasks]     " << what << " " << (t - t0) << " s" << std::endl;
        t0 = t;
    };
    FastaIndex reference;
    ExonRegions exons;
    reference.Open(fasta);
    lap("fasta index");
    std::ifstream ef(exonsFile.c_str());
    if (!ef.good() || !exons.Read(ef)) die("Error: Unable to read exon regions file " + exonsFile);
    lap("exon regions");
    std::map<int, SplitAlignmentTask> tasks;   // canonical iteration order: ascending fusion id
    std::vector<std::pair<SplitAlignmentTask*, const std::vector<Location>*>> work;
    std::vector<int> ids;
    for (const auto& kv : regions) {
        work.emplace_back(&tasks.emplace_hint(tasks.end(), kv.first, SplitAlignmentTask())->second, &kv.second);
        ids.push_back(kv.first);
    }
    lap("task table");
    // the fusions are independent and everything they read (index, exon tables, the FASTA through pread) is read-only
    const unsigned n = work.size() < 64 ? 1u : (threads ? threads : host_threads());
    run_threads(n, [&](unsigned t) {
        for (size_t i = t; i < work.size(); i += n)
            work[i].first->Initialize(ids[i], *work[i].second, reference, exons, fragMean, fragStdDev, minRead, maxRead);
    });
    lap("windows and mate regions");
    return tasks;
}

// tools/SplitAlignment.cpp:177-229.  The reference keeps a hash map of bins per strand and reference name; here the (strand,
// reference, bin, region) entries are collected flat, sorted once (Finish) and looked up through a table of bin starts per
// (strand, reference): a hundred thousand fusions' regions are binned in milliseconds, and a lookup is one hash of the name
// and an index.  Within a bin the regions keep the order in which they were added.
class BinnedLocations {
public:
    explicit BinnedLocations(int spacing) : spacing_(spacing) {}
    void Add(int id, const Location& loc)
    {
        const int idx = (int)ids_.size();
        ids_.push_back(id);
        regions_.push_back(Region{loc.start, loc.end});
        auto it = ref_index_[loc.strand].find(loc.refName);
        int ref;
        if (it == ref_index_[loc.strand].end()) {
            ref = (int)refs_.size();
            ref_index_[loc.strand].emplace(loc.refName, ref);
            refs_.emplace_back();
        } else {
            ref = it->second;
        }
        for (int b = loc.start / spacing_; b <= loc.end / spacing_; ++b) entries_.push_back(Entry{ref, b, idx});
        finished_ = false;
    }
    // after the last Add, before the first Overlapping
    void Finish()
    {
        std::sort(entries_.begin(), entries_.end(), [](const Entry& a, const Entry& b) {
            if (a.ref != b.ref) return a.ref < b.ref;
            if (a.bin != b.bin) return a.bin < b.bin;
            return a.idx < b.idx;
        });
        flat_.resize(entries_.size());
        for (size_t k = 0; k < entries_.size(); ++k) flat_[k] = entries_[k].idx;
        for (size_t k = 0; k < entries_.size();) {                      // per reference: its bins' starts in flat_, from its first bin on
            const int ref = entries_[k].ref;
            size_t e = k;
            while (e < entries_.size() && entries_[e].ref == ref) ++e;
            RefBins& rb = refs_[(size_t)ref];
            rb.first_bin = entries_[k].bin;
            const int n_bins = entries_[e - 1].bin - rb.first_bin + 1;
            rb.start.assign((size_t)n_bins + 1, 0);
            for (size_t x = k; x < e; ++x) ++rb.start[(size_t)(entries_[x].bin - rb.first_bin) + 1];
            rb.start[0] = (uint32_t)k;
            for (int v = 0; v < n_bins; ++v) rb.start[(size_t)v + 1] += rb.start[(size_t)v];
            k = e;
        }
        std::vector<Entry>().swap(entries_);
        finished_ = true;
    }
    // ids of the regions that overlap, each once, ascending as signed ints (the canonical visiting order)
    void Overlapping(const std::string& ref, int strand, const Region& region, std::vector<int>& ids) const
    {
        ids.clear();
        auto ri = ref_index_[strand].find(ref);
        if (ri == ref_index_[strand].end() || !finished_) return;
        const RefBins& rb = refs_[(size_t)ri->second];
        const int n_bins = (int)rb.start.size() - 1;
        for (int b = std::max(region.start / spacing_, rb.first_bin); b <= region.end / spacing_ && b - rb.first_bin < n_bins; ++b)
            for (uint32_t x = rb.start[(size_t)(b - rb.first_bin)]; x < rb.start[(size_t)(b - rb.first_bin) + 1]; ++x) {
                const int idx = flat_[x];
                if (regions_[idx].start <= region.end && regions_[idx].end >= region.start) ids.push_back(ids_[idx]);
            }
        if (ids.size() > 1) {
            std::sort(ids.begin(), ids.end());
            ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
        }
    }

private:
    struct Entry { int ref, bin, idx; };
    struct RefBins { int first_bin = 0; std::vector<uint32_t> start; };
    int spacing_;
    bool finished_ = true;
    std::unordered_map<std::string, int> ref_index_[2];       // (strand, name) -> refs_
    std::vector<RefBins> refs_;
    std::vector<Entry> entries_;
    std::vector<int> flat_;
    std::vector<int> ids_;
    std::vector<Region> regions_;
};

// The reads of both FASTQ files by ReadID (fragment index, read end): sequences in one byte pool, found through a table
// indexed by 2 * (fragment - first fragment seen) + end.  Fragment indices are the running numbers
// scripts/index_paired_fastq.pl gives the reads of the WHOLE run, and a tool process sees one chunk of them: the table is
// relative to the chunk's first index and only grows while it stays dense (at most eight slots per read stored, so its size
// follows the number of reads, not the largest id); anything else goes to a hash map.  A later read of the same id replaces
// the earlier one, as `reads[id] = sequence` does in the reference (tools/SplitAlignment.cpp:253-264).
class ReadStore {
public:
    void put(int frag, int end, const char* s, size_t n)
    {
        const uint64_t v = ((uint64_t)pool_.size() << 24) | (uint64_t)n;
        pool_.insert(pool_.end(), s, s + n);
        if (count_++ == 0) base_ = (int64_t)frag - ((int64_t)frag & 1023);        // some room below the first id
        const int64_t rel = (int64_t)frag - base_;
        if (rel >= 0 && rel < ((int64_t)1 << 28)) {
            const size_t k = (size_t)rel * 2 + (size_t)end;
            if (k < dense_.size()) { dense_[k] = v; return; }
            if (k + 1 <= 8 * count_ + 4096) {
                dense_.resize(std::min<size_t>(std::max(k + 1, dense_.size() * 2), 8 * count_ + 4096), NONE);
                dense_[k] = v;
                return;
            }
        }
        sparse_[pack_id(frag, end)] = v;
    }
    // false: no such read (it then aligns as the empty string, tools/SplitAlignment.cpp:286)
    bool get(int frag, int end, const char*& s, size_t& n) const
    {
        uint64_t v = NONE;
        const int64_t rel = (int64_t)frag - base_;
        if (rel >= 0 && (size_t)rel * 2 + (size_t)end < dense_.size()) v = dense_[(size_t)rel * 2 + (size_t)end];
        if (v == NONE && !sparse_.empty()) {      // (an id stored before the table reached it; the table has the later one if both exist)
            auto it = sparse_.find(pack_id(frag, end));
            if (it != sparse_.end()) v = it->second;
        }
        if (v == NONE) return false;
        s = pool_.data() + (v >> 24);
        n = (size_t)(v & 0xFFFFFF);
        return true;
    }
    size_t table_slots() const { return dense_.size(); }
private:
    static constexpr uint64_t NONE = ~(uint64_t)0;
    std::vector<char> pool_;
    std::vector<uint64_t> dense_;
    std::unordered_map<int, uint64_t> sparse_;
    int64_t base_ = 0;
    size_t count_ = 0;
};

struct ReadStorePair {
    ReadStore file[2];
    bool get(int frag, int end, const char*& s, size_t& n) const { return file[1].get(frag, end, s, n) || file[0].get(frag, end, s, n); }
};

// FASTQ: tools/ReadStream.cpp:18-32 (extension check), :57-104 (record parsing), AddReads SplitAlignment.cpp:253-264
// err receives what the reference prints on stderr; a condition on which the reference's process ends (an uncaught
// bad_lexical_cast, ...) is returned in *fatal instead of ending the process here, so that a caller that reads the FASTQ files
// beside other work can report things in the reference's order.
inline bool AddReads(const std::string& filename, ReadStore& reads, std::ostream& err, std::string* fatal)
{
    const size_t dot = filename.find_last_of('.');
    const std::string ext = filename.substr(dot + 1);
    if (ext != "fastq" && ext != "fq") {
        err << "Error: unrecognized extension " << ext << std::endl;
        return false;
    }
    FILE* in = fopen(filename.c_str(), "rb");
    if (!in) {
        err << "Error: unable to open file " << filename << std::endl;
        return false;
    }
    LineReader reader(in);
    std::string name, sequence;
    for (;;) {
        const char* l;
        size_t n;
        if (!reader.next(l, n)) break;                     // records of four lines; an incomplete one ends the file
        name.assign(l, n);
        if (!reader.next(l, n)) break;
        sequence.assign(l, n);
        if (!reader.next(l, n) || !reader.next(l, n)) break;
        if (name.empty() || name[0] != '@') { err << "Error: Unable to interpret read name " << name << std::endl; break; }
        const size_t slash = name.find_first_of('/');
        const char endc = (slash != std::string::npos && slash + 1 < name.size()) ? name[slash + 1] : '\0';
        if (endc != '1' && endc != '2') { err << "Error: Unable to interpret read end " << name << std::endl; break; }
        int frag;
        if (!field_int(name.data() + 1, slash - 1, frag)) { *fatal = "Error: bad integer '" + name.substr(1, slash - 1) + "' in read name " + name; break; }
        if (sequence.size() >= ((size_t)1 << 24)) { *fatal = "Error: read longer than 16 M bases: " + name; break; }
        reads.put(frag, endc == '1' ? 0 : 1, sequence.data(), sequence.size());
    }
    fclose(in);
    return fatal->empty();
}

// The reads of one FASTQ file by ReadID, built by a team of threads from the mapped file: the same records, messages and
// last-one-wins rule as AddReads / ReadStore above (tools/ReadStream.cpp:57-104, tools/SplitAlignment.cpp:253-264), but the
// sequences stay where they are in the mapped text and the table is filled side by side.  The file is cut into one piece per
// thread at line starts; a count of the newlines in front of every piece tells where its first whole record (four lines)
// begins, so every record is parsed exactly once, by the piece it starts in.  A record that ends the reading (bad name, bad
// end, ...) ends it for every later record too: the first such record in file order decides, as in a serial reader.
class ReadTable {
public:
    bool load(const std::string& filename, Team& team, std::ostream& err, std::string* fatal)
    {
        const size_t dot = filename.find_last_of('.');
        const std::string ext = filename.substr(dot + 1);
        if (ext != "fastq" && ext != "fq") {
            err << "Error: unrecognized extension " << ext << std::endl;
            return false;
        }
        if (!text_.try_load(filename, false, false)) {
            err << "Error: unable to open file " << filename << std::endl;
            return false;
        }
        const unsigned np = (text_.size() < ((size_t)1 << 16)) ? 1u : team.size();
        struct Entry { uint64_t off; uint32_t len; int frag; int end; };
        struct Piece {
            std::vector<Entry> entries;
            size_t newlines = 0;
            int stop = 0;                    // 0 none, 1 bad name, 2 bad end, 3 bad integer (fatal), 4 read too long (fatal)
            std::string name;                // of the record that stopped the reading
            int min_frag = INT_MAX, max_frag = INT_MIN;
        };
        std::vector<Piece> pieces(np);
        const std::vector<size_t> cut = text_.cut_lines(0, text_.size(), np);
        std::vector<size_t> lines_before(np + 1, 0);
        const char* const txt = text_.data();
        const size_t N = text_.size();
        // a line exists at pos iff pos < N; it ends at the next newline or at N
        auto line = [&](size_t pos, size_t& b, size_t& e, size_t& next) {
            if (pos >= N) return false;
            const char* nl = (const char*)memchr(txt + pos, '\n', N - pos);
            b = pos;
            e = nl ? (size_t)(nl - txt) : N;
            next = nl ? e + 1 : N;
            return true;
        };
        team.run([&](unsigned t) {
            if (t >= np) { team.barrier(); return; }
            Piece& pc = pieces[t];
            pc.newlines = (size_t)std::count(txt + cut[t], txt + cut[t + 1], '\n');
            team.barrier();
            size_t before = 0;
            for (unsigned u = 0; u < t; ++u) before += pieces[u].newlines;
            size_t pos = cut[t], b, e, next;
            for (size_t skip = (4 - before % 4) % 4; skip > 0; --skip) {
                if (!line(pos, b, e, next)) return;
                pos = next;
            }
            pc.entries.reserve((cut[t + 1] - cut[t]) / 128 + 16);
            while (pos < cut[t + 1]) {
                size_t nb, ne, sb, se, xb, xe;
                if (!line(pos, nb, ne, next)) break;                       // records of four lines; an incomplete one ends the file
                if (!line(next, sb, se, next)) break;
                if (!line(next, xb, xe, next) || !line(next, xb, xe, next)) break;
                pos = next;
                const char* name = txt + nb;
                const size_t nlen = ne - nb;
                if (nlen == 0 || name[0] != '@') { pc.stop = 1; pc.name.assign(name, nlen); break; }
                const char* slash = (const char*)memchr(name, '/', nlen);
                const char endc = (slash && slash + 1 < name + nlen) ? slash[1] : '\0';
                if (endc != '1' && endc != '2') { pc.stop = 2; pc.name.assign(name, nlen); break; }
                int frag;
                if (!field_int(name + 1, (size_t)(slash - name) - 1, frag)) { pc.stop = 3; pc.name.assign(name, nlen); break; }
                if (se - sb >= ((size_t)1 << 24)) { pc.stop = 4; pc.name.assign(name, nlen); break; }
                pc.entries.push_back(Entry{(uint64_t)sb, (uint32_t)(se - sb), frag, endc == '1' ? 0 : 1});
                pc.min_frag = std::min(pc.min_frag, frag);
                pc.max_frag = std::max(pc.max_frag, frag);
            }
        });
        // the first stop in file order ends the file
        unsigned used = np;
        for (unsigned t = 0; t < np; ++t)
            if (pieces[t].stop) { used = t + 1; break; }
        size_t count = 0;
        int min_frag = INT_MAX, max_frag = INT_MIN;
        for (unsigned t = 0; t < used; ++t) {
            count += pieces[t].entries.size();
            min_frag = std::min(min_frag, pieces[t].min_frag);
            max_frag = std::max(max_frag, pieces[t].max_frag);
        }
        if (used > 0 && pieces[used - 1].stop) {
            const Piece& pc = pieces[used - 1];
            if (pc.stop == 1) err << "Error: Unable to interpret read name " << pc.name << std::endl;
            else if (pc.stop == 2) err << "Error: Unable to interpret read end " << pc.name << std::endl;
            else if (pc.stop == 3) {
                const size_t slash = pc.name.find_first_of('/');
                *fatal = "Error: bad integer '" + pc.name.substr(1, slash - 1) + "' in read name " + pc.name;
            } else *fatal = "Error: read longer than 16 M bases: " + pc.name;
        }
        if (count) {
            // the table is relative to the smallest id of the file and as long as the ids are dense (at most eight slots per read
            // stored, so its size follows the number of reads, not the largest id); ids beyond it go to a hash map
            base_ = (int64_t)min_frag - ((int64_t)min_frag & 1023);
            const uint64_t span = (uint64_t)((int64_t)max_frag - base_) * 2 + 2;
            dense_n_ = (size_t)std::min<uint64_t>(span, 8 * (uint64_t)count + 4096);
            dense_.reset(new std::atomic<uint64_t>[dense_n_]);
            std::mutex sparse_mutex;
            team.run([&](unsigned t) {
                const unsigned nt = team.size();
                for (size_t k = dense_n_ * t / nt; k < dense_n_ * (t + 1) / nt; ++k) dense_[k].store(0, std::memory_order_relaxed);
                team.barrier();
                for (unsigned u = t; u < used; u += nt)
                    for (const Entry& en : pieces[u].entries) {
                        const uint64_t v = (en.off << 24) | (uint64_t)en.len;          // a later record has the larger offset: max = last wins
                        const uint64_t k = (uint64_t)((int64_t)en.frag - base_) * 2 + (uint64_t)en.end;
                        if (k < dense_n_) {
                            uint64_t cur = dense_[k].load(std::memory_order_relaxed);
                            while (cur < v && !dense_[k].compare_exchange_weak(cur, v, std::memory_order_relaxed)) {}
                        } else {
                            std::lock_guard<std::mutex> lk(sparse_mutex);
                            uint64_t& slot = sparse_[pack_id(en.frag, en.end)];
                            slot = std::max(slot, v);
                        }
                    }
            });
        }
        return fatal->empty();
    }
    // false: no such read (it then aligns as the empty string, tools/SplitAlignment.cpp:286)
    bool get(int frag, int end, const char*& s, size_t& n) const
    {
        uint64_t v = 0;
        const int64_t rel = (int64_t)frag - base_;
        if (rel >= 0 && (uint64_t)rel * 2 + (uint64_t)end < dense_n_) v = dense_[(size_t)rel * 2 + (size_t)end].load(std::memory_order_relaxed);
        else if (!sparse_.empty()) {
            auto it = sparse_.find(pack_id(frag, end));
            if (it != sparse_.end()) v = it->second;
        }
        if (v == 0) return false;             // (a sequence never starts at offset 0 of its file: its name line comes first)
        s = text_.data() + (v >> 24);
        n = (size_t)(v & 0xFFFFFF);
        return true;
    }
    size_t table_slots() const { return dense_n_; }
    // the end of the run: the mapped text goes, share by share
    void drop_pages(unsigned t, unsigned of) const { text_.drop_pages(text_.size() * t / of, text_.size() * (t + 1) / of); }
private:
    MappedText text_;
    std::unique_ptr<std::atomic<uint64_t>[]> dense_;
    size_t dense_n_ = 0;
    std::unordered_map<int, uint64_t> sparse_;
    int64_t base_ = 0;
};

struct ReadTablePair {
    ReadTable file[2];
    bool get(int frag, int end, const char*& s, size_t& n) const { return file[1].get(frag, end, s, n) || file[0].get(frag, end, s, n); }
};

// One SAM line (tools/AlignmentStream.cpp:39-130).  Returns 0 = a record, 1 = nothing to return (header line, rname "*"),
// 2.. = the reference dies: 2 empty line, 3 fewer than ten fields, 4 flag or position not an integer, 5 qname "x/y" with y
// not 1 or 2.  read_end is left alone when neither the name nor the flag bits tell it (the reference's object keeps the
// previous record's value); fragment points into the line.
struct SamFields { const char* fragment; size_t fragment_len; const char* reference; size_t reference_len; int strand; Region region; };
inline int ParseSamLine(const char* line, size_t len, SamFields& a, int& read_end)
{
    if (len == 0) return 2;
    if (line[0] == '@') return 1;
    const char* fs[11];
    int nf = 0;
    fs[nf++] = line;
    const char* end = line + len;
    for (const char* p = line; nf < 11;) {
        const char* tab = (const char*)memchr(p, '\t', (size_t)(end - p));
        if (!tab) break;
        fs[nf++] = p = tab + 1;
    }
    if (nf < 10) return 3;
    if (nf < 11) fs[10] = end + 1;                                   // field k is [fs[k], fs[k+1] - 1)
    auto flen = [&](int k) { return (size_t)(fs[k + 1] - 1 - fs[k]); };
    int flag, pos;
    if (!field_int(fs[1], flen(1), flag) || !field_int(fs[3], flen(3), pos)) return 4;
    if (flen(2) == 1 && fs[2][0] == '*') return 1;
    a.strand = (flag & 0x10) ? MinusStrand : PlusStrand;
    // qname split at '/': exactly two parts name the read end, anything else leaves it to the flag bits
    const char* slash = (const char*)memchr(fs[0], '/', flen(0));
    const bool two_parts = slash && !memchr(slash + 1, '/', (size_t)(fs[0] + flen(0) - slash - 1));
    if (two_parts) {
        const size_t tail = (size_t)(fs[0] + flen(0) - slash - 1);
        if (tail != 1 || (slash[1] != '1' && slash[1] != '2')) return 5;
        a.fragment = fs[0];
        a.fragment_len = (size_t)(slash - fs[0]);
        read_end = (slash[1] == '1') ? 0 : 1;
    } else {
        a.fragment = fs[0];
        a.fragment_len = flen(0);
        if (flag & 0x40) read_end = 0;
        else if (flag & 0x80) read_end = 1;
    }
    a.reference = fs[2];
    a.reference_len = flen(2);
    a.region.start = pos;
    a.region.end = pos + (int)flen(9) - 1;
    return 0;
}
[[noreturn]] inline void DieSamLine(int kind, size_t line_no)
{
    if (kind == 2) die("Error: Empty alignment line " + std::to_string(line_no));
    if (kind == 3) die("Error: Format error for alignment line " + std::to_string(line_no));
    if (kind == 4) die("Error: bad integer in sam line " + std::to_string(line_no));        // reference: uncaught bad_lexical_cast
    die("Error: Unable to interpret qname for alignment line " + std::to_string(line_no));
}

}  // namespace defuse
