// defuse_glue — the text glue between the path's tools, one binary, one subcommand per
// reference script, same arguments, same input and output formats:
//
//   defuse_glue merge_clusters f1 f2 ...            scripts/merge_clusters.pl:9-33        (stdout)
//   defuse_glue get_align_regions                   scripts/get_align_regions.pl:14-53    (stdin -> stdout)
//   defuse_glue remove_duplicates <min_size>        scripts/remove_duplicates.pl:11-104   (stdin -> stdout)
//   defuse_glue filter_unmatched                    scripts/filter_unmatched.pl:16-50     (stdin -> stdout)
//   defuse_glue divide_sam_chr_pairs -t <trans> -p <prefix>   scripts/divide_sam_chr_pairs.pl:9-177 (stdin -> files + list on stdout)
//
// A symlink named after a script (merge_clusters.pl -> defuse_glue) selects the subcommand by its own name, so the
// pipeline's `$scripts_directory/<script>.pl` can point at it unchanged.  Host text work only: nothing here has a
// device part (the files are a few hundred MB and every step is one pass), it is here because these steps sit between
// clustermatepairs, setcover and dosplitalign and were the slowest links once those ran on the GPU.
// Where a script's output order follows Perl's hash order (randomised per process), the canonical order of
// SURVEY 8(c) is used: ascending numeric ids, chromosome names in string order.
#include "defuse_host.hpp"

using namespace defuse;

namespace {

struct Fields {
    const char* p[16];
    int n = 0;
    const char* end = nullptr;
    size_t len(int k) const { return (size_t)((k + 1 < n ? p[k + 1] - 1 : end) - p[k]); }
    std::string str(int k) const { return std::string(p[k], len(k)); }
};

// up to `want` leading tab-separated fields of a line (more are left inside the last one's tail)
void split_fields(const char* line, size_t len, int want, Fields& f)
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

long long num(const Fields& f, int k, const char* what)
{
    int v;
    if (k >= f.n || !field_int(f.p[k], f.len(k), v)) die(std::string("Error: bad ") + what + " '" + (k < f.n ? f.str(k) : std::string()) + "'");
    return v;
}

class Out {
public:
    explicit Out(FILE* f) : f_(f) { buf_.reserve((1u << 22) + 4096); }
    ~Out() { flush(); }
    std::string& buf() { return buf_; }
    void maybe_flush() { if (buf_.size() > (1u << 22)) flush(); }
    void flush() { if (!buf_.empty()) { fwrite(buf_.data(), 1, buf_.size(), f_); buf_.clear(); } }
private:
    FILE* f_;
    std::string buf_;
};

// scripts/merge_clusters.pl: cluster ids renumbered consecutively over all files, a new id at every change of the
// first column and at every file boundary; the rest of each line is copied
int merge_clusters(int argc, char** argv)
{
    if (argc == 0) { std::cerr << "Usage merge_clusters clusters1 clusters2 ...\n"; return 1; }
    Out out(stdout);
    long long cluster_id = 0;
    for (int a = 0; a < argc; ++a) {
        FILE* in = fopen(argv[a], "rb");
        if (!in) die(std::string("Error: Unable to open ") + argv[a]);
        LineReader reader(in);
        const char* line;
        size_t len;
        bool have = false;
        long long prev = 0;
        Fields f;
        while (reader.next(line, len)) {
            split_fields(line, len, 2, f);
            const long long id = num(f, 0, "cluster id");
            if (have && prev != id) ++cluster_id;
            prev = id;
            have = true;
            append_int(out.buf(), cluster_id);
            out.buf().append(line + f.len(0), len - f.len(0));
            out.buf() += '\n';
            out.maybe_flush();
        }
        fclose(in);
        if (have) ++cluster_id;
    }
    return 0;
}

// scripts/get_align_regions.pl: per (cluster, end) the reference name and strand of its last line and the extent of all
// its alignments; clusters ascending, end 0 then 1; a cluster without exactly two ends is an error
int get_align_regions()
{
    struct EndInfo { std::string ref, strand; long long start = 0, end = 0; bool have = false; };
    std::map<long long, std::map<long long, EndInfo>> clusters;
    LineReader reader(stdin);
    const char* line;
    size_t len;
    Fields f;
    while (reader.next(line, len)) {
        split_fields(line, len, 9, f);
        if (f.n < 8) die("Error: cluster line with fewer than 8 fields");
        const long long id = num(f, 0, "cluster id"), ce = num(f, 1, "cluster end"), start = num(f, 6, "start"), end = num(f, 7, "end");
        EndInfo& e = clusters[id][ce];
        e.ref = f.str(4);
        e.strand = f.str(5);
        if (!e.have) { e.start = start; e.end = end; e.have = true; }
        e.start = std::min(e.start, start);
        e.end = std::max(e.end, end);
    }
    Out out(stdout);
    for (const auto& c : clusters) {
        if (c.second.size() != 2) die("Error: Did not find 2 ends for cluster " + std::to_string(c.first));
        for (const auto& e : c.second) {
            append_int(out.buf(), c.first); out.buf() += '\t';
            append_int(out.buf(), e.first); out.buf() += '\t';
            out.buf() += e.second.ref; out.buf() += '\t';
            out.buf() += e.second.strand; out.buf() += '\t';
            append_int(out.buf(), e.second.start); out.buf() += '\t';
            append_int(out.buf(), e.second.end); out.buf() += '\n';
        }
        out.maybe_flush();
    }
    return 0;
}

// scripts/remove_duplicates.pl: inside a cluster, fragments whose pair of positions (start on '+', end on '-', one per
// cluster end) was seen already are dropped; a cluster is kept if at least min_cluster_size fragments remain.
// Fragments are visited in ascending order (the script: Perl hash order), so of a set of duplicates the smallest
// fragment index stays.
int remove_duplicates(int argc, char** argv)
{
    int min_size;
    if (argc < 1 || !field_int(argv[0], strlen(argv[0]), min_size)) { std::cerr << "Usage: remove_duplicates min_cluster_size < in_clusters > out_clusters\n"; return 1; }
    struct Frag { long long pos[2] = {0, 0}; bool have[2] = {false, false}; std::string line[2]; };
    std::map<long long, Frag> frags;
    Out out(stdout);
    auto emit = [&]() {
        std::set<std::pair<long long, long long>> seen;
        std::vector<const Frag*> kept;
        for (const auto& kv : frags) {
            const Frag& fr = kv.second;
            if (!fr.have[0] || !fr.have[1]) die("Error: fragment " + std::to_string(kv.first) + " lacks a cluster end");   // the script dies on the undefined value
            if (!seen.insert(std::make_pair(fr.pos[0], fr.pos[1])).second) continue;
            kept.push_back(&fr);
        }
        if ((long long)kept.size() * 2 >= 2LL * min_size)
            for (const Frag* fr : kept)
                for (int e = 0; e < 2; ++e) { out.buf() += fr->line[e]; out.buf() += '\n'; }
        out.maybe_flush();
        frags.clear();
    };
    LineReader reader(stdin);
    const char* line;
    size_t len;
    Fields f;
    bool have = false;
    long long current = 0;
    while (reader.next(line, len)) {
        split_fields(line, len, 9, f);
        if (f.n < 8) die("Error: cluster line with fewer than 8 fields");
        const long long id = num(f, 0, "cluster id"), ce = num(f, 1, "cluster end"), frag = num(f, 2, "fragment id");
        if (have && current != id) emit();
        current = id;
        have = true;
        if (ce != 0 && ce != 1) continue;                       // the script stores it and never looks at it again
        Frag& fr = frags[frag];
        fr.pos[ce] = (f.len(5) == 1 && f.p[5][0] == '+') ? num(f, 6, "start") : num(f, 7, "end");
        fr.have[ce] = true;
        fr.line[ce].assign(line, len);
    }
    if (have) emit();
    return 0;
}

// qname =~ /(.*)\/([12])/ : the last "/1" or "/2" of the name
bool split_qname(const char* q, size_t n, size_t& frag_len, int& read_end)
{
    for (size_t k = n; k-- > 1;)
        if ((q[k] == '1' || q[k] == '2') && q[k - 1] == '/') { frag_len = k - 1; read_end = q[k] - '0'; return true; }
    return false;
}

// scripts/filter_unmatched.pl: the lines of a fragment (a run of equal numeric fragment ids) are kept if both read ends occur
int filter_unmatched()
{
    LineReader reader(stdin);
    Out out(stdout);
    const char* line;
    size_t len;
    Fields f;
    bool have = false, ends[3] = {false, false, false};
    long long current = 0;
    std::string held;
    auto emit = [&]() {
        if (ends[1] && ends[2]) { out.buf() += held; out.maybe_flush(); }
        held.clear();
        ends[1] = ends[2] = false;
    };
    while (reader.next(line, len)) {
        split_fields(line, len, 2, f);
        size_t fl;
        int re;
        if (!split_qname(f.p[0], f.len(0), fl, re)) die("Error: read name without /1 or /2: " + f.str(0));
        int frag;
        if (!field_int(f.p[0], fl, frag)) die("Error: fragment id is not a number: " + f.str(0));
        if (have && frag != current) emit();
        current = frag;
        have = true;
        ends[re] = true;
        held.append(line, len);
        held += '\n';
    }
    if (have) emit();
    return 0;
}

// scripts/divide_sam_chr_pairs.pl: compact spanning alignments (fragment, read end - 1, rname, strand, pos, pos + len(SEQ) - 1)
// of every fragment with both ends aligned, written to one file per sorted chromosome pair <prefix><chr1>-<chr2>; the list of
// files goes to stdout.  A transcript's chromosome comes from the -t table (gene, transcript, chromosome).
int divide_sam_chr_pairs(int argc, char** argv)
{
    std::string trans, prefix;
    for (int a = 0; a < argc; ++a) {
        const std::string t = argv[a];
        if ((t == "-t" || t == "--trans") && a + 1 < argc) trans = argv[++a];
        else if ((t == "-p" || t == "--prefix") && a + 1 < argc) prefix = argv[++a];
        else { std::cerr << "Usage: divide_sam_chr_pairs -t trans_chr_map -p prefix < sam\n"; return 1; }
    }
    if (trans.empty() || prefix.empty()) { std::cerr << "Usage: divide_sam_chr_pairs -t trans_chr_map -p prefix < sam\n"; return 1; }
    std::unordered_map<std::string, std::string> trans_chr;
    {
        FILE* in = fopen(trans.c_str(), "rb");
        if (!in) die("Error: Unable to open " + trans);
        LineReader reader(in);
        const char* line;
        size_t len;
        Fields f;
        while (reader.next(line, len)) {
            split_fields(line, len, 4, f);
            if (f.n < 3) continue;
            trans_chr[f.str(0) + "|" + f.str(1)] = f.str(2);
        }
        fclose(in);
    }
    // per output file: buffered text, appended to the file when it grows (the script: every 10000 alignments)
    struct Sink { std::string name, buf; bool created = false; };
    std::map<std::pair<std::string, std::string>, Sink> sinks;
    auto flush = [&](Sink& s) {
        FILE* o = fopen(s.name.c_str(), s.created ? "ab" : "wb");      // the first write replaces an old file (unlink in the script)
        if (!o) die("Error: Unable to write to " + s.name);
        fwrite(s.buf.data(), 1, s.buf.size(), o);
        fclose(o);
        s.created = true;
        s.buf.clear();
    };
    // the current fragment: per read end (1, 2) and chromosome the compact lines
    std::map<std::string, std::string> cur[3];
    auto process = [&]() {
        if (!cur[1].empty() && !cur[2].empty())
            for (const auto& c1 : cur[1])
                for (const auto& c2 : cur[2]) {
                    const bool fwd = c1.first <= c2.first;
                    Sink& s = sinks[fwd ? std::make_pair(c1.first, c2.first) : std::make_pair(c2.first, c1.first)];
                    if (s.name.empty()) s.name = prefix + (fwd ? c1.first : c2.first) + "-" + (fwd ? c2.first : c1.first);
                    s.buf += c1.second;
                    s.buf += c2.second;
                    if (s.buf.size() > (1u << 20)) flush(s);
                }
        cur[1].clear();
        cur[2].clear();
    };
    LineReader reader(stdin);
    const char* line;
    size_t len;
    Fields f;
    std::string current, key, rec;
    bool have = false;
    while (reader.next(line, len)) {
        if (len > 0 && line[0] == '@') continue;
        split_fields(line, len, 11, f);
        if (f.n < 10) die("Error: sam line with fewer than 10 fields");
        size_t fl;
        int re;
        if (!split_qname(f.p[0], f.len(0), fl, re)) die("Error: read name without /1 or /2: " + f.str(0));
        const long long flag = num(f, 1, "flag"), pos = num(f, 3, "position");
        if (have && (fl != current.size() || memcmp(current.data(), f.p[0], fl) != 0)) process();
        current.assign(f.p[0], fl);
        have = true;
        key.assign(f.p[2], f.len(2));
        auto tc = trans_chr.find(key);
        const std::string& chr = tc == trans_chr.end() ? key : tc->second;
        rec.assign(f.p[0], fl); rec += '\t';
        append_int(rec, re - 1); rec += '\t';
        rec += key; rec += '\t';
        rec += (flag & 0x10) ? '-' : '+'; rec += '\t';
        append_int(rec, pos); rec += '\t';
        append_int(rec, pos + (long long)f.len(9) - 1); rec += '\n';
        cur[re][chr] += rec;
    }
    if (have) process();
    Out out(stdout);
    for (auto& kv : sinks) {
        flush(kv.second);
        out.buf() += kv.first.first; out.buf() += '\t'; out.buf() += kv.first.second; out.buf() += '\t'; out.buf() += kv.second.name; out.buf() += '\n';
    }
    return 0;
}

}  // namespace

int main(int argc, char* argv[])
{
    std::string prog = argc > 0 ? argv[0] : "";
    const size_t slash = prog.find_last_of('/');
    if (slash != std::string::npos) prog = prog.substr(slash + 1);
    if (prog.size() > 3 && prog.compare(prog.size() - 3, 3, ".pl") == 0) prog.resize(prog.size() - 3);
    int first = 1;
    if (prog == "defuse_glue") {
        if (argc < 2) {
            std::cerr << "Usage: defuse_glue merge_clusters|get_align_regions|remove_duplicates|filter_unmatched|divide_sam_chr_pairs [args]\n";
            return 1;
        }
        prog = argv[1];
        first = 2;
    }
    if (prog == "merge_clusters") return merge_clusters(argc - first, argv + first);
    if (prog == "get_align_regions") return get_align_regions();
    if (prog == "remove_duplicates") return remove_duplicates(argc - first, argv + first);
    if (prog == "filter_unmatched") return filter_unmatched();
    if (prog == "divide_sam_chr_pairs") return divide_sam_chr_pairs(argc - first, argv + first);
    std::cerr << "defuse_glue: unknown subcommand " << prog << "\n";
    return 1;
}
