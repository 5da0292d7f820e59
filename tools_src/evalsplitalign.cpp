// evalsplitalign — drop-in replacement of the reference tool (tools/evalsplitalign.cpp:25-115):
// groups the fusion-id-sorted split alignments, picks the best supported breakpoint per fusion
// (SplitAlignmentTask::Evaluate, tools/SplitAlignment.cpp:484-594) and writes the .seq / .break /
// .predalign files (BreakPrediction::Write*, :596-624).  Host-only, like the reference: this stage has
// no DP and is I/O bound.  Ties between equally supported breakpoints go to the
// lexicographically smallest refSplit (canonical order of SURVEY.md 8(c)).
#include "evaluate.hpp"

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
    cmd.add("a", "align", "Split Alignments Filename", "string");
    cmd.add("q", "seq", "Sequences Filename", "string");
    cmd.add("b", "break", "Break Positions Filename", "string");
    cmd.add("p", "predalign", "Prediction Split Alignments Filename", "string");
    cmd.parse(argc, argv);

    const std::map<int, std::vector<Location>> regions = ReadAlignRegionPairs(cmd.str("regions"));
    std::map<int, SplitAlignmentTask> tasks = CreateTasks(cmd.str("fasta"), cmd.str("exons"), cmd.real("ufrag"), cmd.real("sfrag"),
                                                         cmd.integer("minread"), cmd.integer("maxread"), regions);

    // The alignment file is mapped and cut into one piece per host thread at group boundaries (a group = a run of lines
    // with one fusion id, as ReadSortedAlignments forms them); the pieces are evaluated side by side and their three texts
    // written in order.  A malformed line ends the run as in the reference: everything before it is written, then the
    // message.
    MappedText text;
    text.load(cmd.str("align"), "Error: Unable to open ");
    OrderedFileWriter seqFile, breakFile, predFile;
    if (!seqFile.open_file(cmd.str("seq"))) die("Error: Unable to open " + cmd.str("seq"));
    if (!breakFile.open_file(cmd.str("break"))) die("Error: Unable to open " + cmd.str("break"));
    if (!predFile.open_file(cmd.str("predalign"))) die("Error: Unable to open " + cmd.str("predalign"));
    const SplitAlignmentTask emptyTask;                  // operator[] of the reference: an unknown id evaluates against an empty task

    unsigned nPieces = host_threads();
    if (text.size() < ((size_t)1 << 20) && !std::getenv("DEFUSE_THREADS")) nPieces = 1;
    auto first_field = [&](size_t pos, size_t& n) {       // the fusion id column of the line at pos, as text
        const size_t e = text.line_end(pos);
        const char* tab = (const char*)memchr(text.data() + pos, '\t', e - pos);
        n = tab ? (size_t)(tab - (text.data() + pos)) : e - pos - ((e > pos && text[e - 1] == '\n') ? 1 : 0);
        return text.data() + pos;
    };
    std::vector<size_t> cut = text.cut_lines(0, text.size(), nPieces);
    for (unsigned t = 1; t < nPieces; ++t) {              // move every cut forward to the next change of the fusion id column
        size_t pos = std::max(cut[t], cut[t - 1]);
        while (pos > 0 && pos < text.size()) {
            size_t prev = pos - 1;                        // start of the previous line
            while (prev > 0 && text[prev - 1] != '\n') --prev;
            size_t na, nb;
            const char* a = first_field(prev, na);
            const char* b = first_field(pos, nb);
            int ia, ib;                                       // compared as the integers the reader casts them to ("013" continues group 13)
            if (!field_int(a, na, ia) || !field_int(b, nb, ib) || ia != ib) break;
            pos = text.line_end(pos);
        }
        cut[t] = pos;
    }
    struct Piece {
        std::string seq, brk, pred, error;
        size_t last_seq = 0, last_brk = 0, last_pred = 0;   // where the texts of the piece's last group begin
        bool cancels_previous = false;                      // its FIRST line is malformed in a way that ends the run inside the previous group's look-ahead
        bool empty = true;
    };
    std::vector<Piece> pieces(nPieces);
    run_threads(nPieces, [&](unsigned t) {
        Piece& out = pieces[t];
        std::vector<SplitAlignment> alignments;
        std::vector<const SplitAlignment*> kept;
        std::map<std::pair<int, int>, int> splitScore;
        SplitAlignment pending;
        EvalTexts texts;
        auto evaluate = [&]() {
            out.last_seq = out.seq.size(); out.last_brk = out.brk.size(); out.last_pred = out.pred.size();
            auto ti = tasks.find(alignments.front().fusionID);
            const SplitAlignmentTask& task = ti == tasks.end() ? emptyTask : ti->second;
            texts.seq.clear(); texts.brk.clear(); texts.pred.clear();
            EvaluateGroup(task, alignments, texts, kept, splitScore);
            out.seq += texts.seq; out.brk += texts.brk; out.pred += texts.pred;
            alignments.clear();
        };
        for (size_t pos = cut[t]; pos < cut[t + 1];) {
            const size_t e = text.line_end(pos);
            const size_t len = (e > pos && text[e - 1] == '\n') ? e - 1 - pos : e - pos;
            bool id_read = false;
            out.error = parse_line(text.data() + pos, len, pending, id_read);
            pos = e;
            const bool first_line = out.empty;
            out.empty = false;
            if (!out.error.empty()) {
                if (first_line && !id_read) out.cancels_previous = true;
                // the reference reads one line ahead: a malformed line that at least opens a new group (seven fields, a
                // readable id different from the running group's) lets the running group through first; any other malformed
                // line ends the run inside the reader, before the running group is evaluated
                if (!(id_read && !alignments.empty() && pending.fusionID != alignments.front().fusionID)) alignments.clear();
                break;
            }
            if (!alignments.empty() && pending.fusionID != alignments.front().fusionID) evaluate();
            alignments.push_back(pending);
        }
        if (!alignments.empty()) evaluate();
    });
    for (unsigned t = 0; t < nPieces; ++t) {
        for (unsigned u = t + 1; u < nPieces; ++u) {          // the next piece that holds lines
            if (pieces[u].empty) continue;
            if (pieces[u].cancels_previous) {                 // a sequential reader meets that line while it still collects this piece's last group
                pieces[t].seq.resize(pieces[t].last_seq); pieces[t].brk.resize(pieces[t].last_brk); pieces[t].pred.resize(pieces[t].last_pred);
            }
            break;
        }
        seqFile.write_round({pieces[t].seq}, 1);
        breakFile.write_round({pieces[t].brk}, 1);
        predFile.write_round({pieces[t].pred}, 1);
        if (!pieces[t].error.empty()) {
            seqFile.close_file(); breakFile.close_file(); predFile.close_file();
            die(pieces[t].error);
        }
    }
    if (!seqFile.close_file() || !breakFile.close_file() || !predFile.close_file()) die("Error: failed writing the predictions");
    return 0;
}
