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
#include <chrono>

#include "defuse_host.hpp"

using namespace defuse;

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
    ClusterPieces in;
    in.load("-");
    const std::string text = align_regions_text(in);
    if (!text.empty() && fwrite(text.data(), 1, text.size(), stdout) != text.size()) fail("Error: failed writing the regions");
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
    struct Line { long long frag; int ce; long long pos; const char* text; size_t len; };
    struct Frag { long long frag, pos[2]; const char* text[2]; size_t len[2]; };
    struct Piece {
        std::vector<Line> lines;                 // of the current cluster, in file order
        std::vector<Frag> frags;
        std::vector<size_t> by_pos;
        std::vector<char> keep;
        std::string out;
        bool have = false;
        long long current = 0;
    };
    const double t_start = now();
    const bool timing = std::getenv("DEFUSE_TIMING") != nullptr;
    ClusterPieces in;
    in.load("-");
    std::vector<Piece> part(in.pieces);
    // a cluster's fragments ascending (the script: Perl hash order), of a fragment's lines per end the last one; of the fragments
    // with one pair of positions the smallest stays
    auto emit = [&](Piece& pc) {
        std::stable_sort(pc.lines.begin(), pc.lines.end(), [](const Line& a, const Line& b) { return a.frag < b.frag; });
        pc.frags.clear();
        for (size_t i = 0; i < pc.lines.size();) {
            Frag fr{pc.lines[i].frag, {0, 0}, {nullptr, nullptr}, {0, 0}};
            size_t j = i;
            for (; j < pc.lines.size() && pc.lines[j].frag == fr.frag; ++j) {
                const Line& l = pc.lines[j];
                fr.pos[l.ce] = l.pos; fr.text[l.ce] = l.text; fr.len[l.ce] = l.len;
            }
            if (!fr.text[0] || !fr.text[1]) fail("Error: fragment " + std::to_string(fr.frag) + " lacks a cluster end");   // the script dies on the undefined value
            pc.frags.push_back(fr);
            i = j;
        }
        pc.by_pos.resize(pc.frags.size());
        for (size_t k = 0; k < pc.by_pos.size(); ++k) pc.by_pos[k] = k;
        std::sort(pc.by_pos.begin(), pc.by_pos.end(), [&](size_t a, size_t b) {
            const Frag &x = pc.frags[a], &y = pc.frags[b];
            if (x.pos[0] != y.pos[0]) return x.pos[0] < y.pos[0];
            if (x.pos[1] != y.pos[1]) return x.pos[1] < y.pos[1];
            return a < b;                                    // ascending fragment id
        });
        pc.keep.assign(pc.frags.size(), 0);
        size_t kept = 0;
        for (size_t k = 0; k < pc.by_pos.size(); ++k) {
            const Frag& x = pc.frags[pc.by_pos[k]];
            if (k == 0 || x.pos[0] != pc.frags[pc.by_pos[k - 1]].pos[0] || x.pos[1] != pc.frags[pc.by_pos[k - 1]].pos[1]) { pc.keep[pc.by_pos[k]] = 1; ++kept; }
        }
        if ((long long)kept * 2 >= 2LL * min_size)
            for (size_t k = 0; k < pc.frags.size(); ++k)
                if (pc.keep[k])
                    for (int e = 0; e < 2; ++e) { pc.out.append(pc.frags[k].text[e], pc.frags[k].len[e]); pc.out += '\n'; }
        pc.lines.clear();
    };
    for (unsigned t = 0; t < in.pieces; ++t) part[t].out.reserve(in.cut[t + 1] - in.cut[t] + 1);      // never more than came in
    in.run([&](unsigned t, const char* line, size_t len) {
        Piece& pc = part[t];
        Fields f;
        split_fields(line, len, 9, f);
        if (f.n < 8) fail("Error: cluster line with fewer than 8 fields");
        const long long id = num(f, 0, "cluster id"), ce = num(f, 1, "cluster end"), frag = num(f, 2, "fragment id");
        if (pc.have && pc.current != id) emit(pc);
        pc.current = id;
        pc.have = true;
        if (ce != 0 && ce != 1) return;                         // the script stores it and never looks at it again
        const long long pos = (f.len(5) == 1 && f.p[5][0] == '+') ? num(f, 6, "start") : num(f, 7, "end");
        pc.lines.push_back(Line{frag, (int)ce, pos, line, len});    // the text stays mapped
    }, [&](unsigned t) { if (part[t].have) emit(part[t]); });
    if (timing) std::cerr << "[remove_duplicates] " << in.pieces << " piece(s) done " << now() - t_start << " s" << std::endl;
    // a file behind stdout: every thread copies its piece's output to its place; anything else: one after the other
    struct stat st;
    const off_t base = lseek(STDOUT_FILENO, 0, SEEK_CUR);
    if (in.pieces > 1 && fstat(STDOUT_FILENO, &st) == 0 && S_ISREG(st.st_mode) && base >= 0 && !(fcntl(STDOUT_FILENO, F_GETFL) & O_APPEND)) {
        std::vector<off_t> at(in.pieces + 1, base);
        for (unsigned t = 0; t < in.pieces; ++t) at[t + 1] = at[t] + (off_t)part[t].out.size();
        std::vector<char> bad(in.pieces, 0);
        run_threads(in.pieces, [&](unsigned t) {
            const std::string& o = part[t].out;
            for (size_t done = 0; done < o.size();) {
                const ssize_t w = pwrite(STDOUT_FILENO, o.data() + done, o.size() - done, at[t] + (off_t)done);
                if (w <= 0) { bad[t] = 1; return; }
                done += (size_t)w;
            }
        });
        for (char b : bad)
            if (b) fail("Error: failed writing the clusters");
        (void)lseek(STDOUT_FILENO, at[in.pieces], SEEK_SET);
    } else {
        for (const Piece& pc : part)
            if (!pc.out.empty() && fwrite(pc.out.data(), 1, pc.out.size(), stdout) != pc.out.size()) fail("Error: failed writing the clusters");
    }
    if (timing) std::cerr << "[remove_duplicates] written " << now() - t_start << " s" << std::endl;
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
    try {
        if (prog == "merge_clusters") return merge_clusters(argc - first, argv + first);
        if (prog == "get_align_regions") return get_align_regions();
        if (prog == "remove_duplicates") return remove_duplicates(argc - first, argv + first);
        if (prog == "filter_unmatched") return filter_unmatched();
        if (prog == "divide_sam_chr_pairs") return divide_sam_chr_pairs(argc - first, argv + first);
    } catch (const GlueError& g) {
        die(g.msg);
    }
    std::cerr << "defuse_glue: unknown subcommand " << prog << "\n";
    return 1;
}
