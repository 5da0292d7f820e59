// evalsplitalign — drop-in replacement of the reference tool (tools/evalsplitalign.cpp:25-115):
// groups the fusion-id-sorted split alignments, picks the best supported breakpoint per fusion
// (SplitAlignmentTask::Evaluate, tools/SplitAlignment.cpp:484-594) and writes the .seq / .break /
// .predalign files (BreakPrediction::Write*, :596-624).  Host-only, like the reference: this stage has
// no DP and is I/O bound.  Ties between equally supported breakpoints go to the
// lexicographically smallest refSplit (canonical order of SURVEY.md 8(c)).
#include "defuse_host.hpp"

using namespace defuse;

namespace {

const int minAnchor = 4;   // tools/SplitAlignment.cpp:29

struct SplitAlignment {
    int fusionID = 0, fragmentIndex = 0, readEnd = 0, revComp = 0;
    std::pair<int, int> refSplit, readSplit;
    int score = 0;
    void Write(std::ostream& out) const   // tools/SplitAlignment.cpp:305-317
    {
        out << fusionID << "\t" << fragmentIndex << "\t" << readEnd << "\t" << revComp << "\t" << refSplit.first << "\t"
            << refSplit.second << "\t" << readSplit.first << "\t" << readSplit.second << "\t" << score << "\t" << std::endl;
    }
};

[[noreturn]] void debug_check_failed(const char* expr)
{
    std::cerr << "Error: " << expr << " failed in SplitAlignmentTask::Evaluate" << std::endl;
    std::exit(1);
}

// SplitAlignment::ReadSortedAlignments (tools/SplitAlignment.cpp:319-369) with one line of look-ahead
// instead of tellg/seekg.
class SortedAlignmentReader {
public:
    explicit SortedAlignmentReader(std::istream& in) : in_(in) {}
    bool NextGroup(std::vector<SplitAlignment>& group)
    {
        group.clear();
        SplitAlignment a;
        while (have_ || ReadOne()) {
            have_ = true;
            if (!group.empty() && pending_.fusionID != group.front().fusionID) return true;
            group.push_back(pending_);
            have_ = false;
        }
        return !group.empty();
    }

private:
    bool ReadOne()
    {
        std::string line;
        if (!std::getline(in_, line)) return false;
        std::vector<std::string> f = split_tabs(line);
        if (f.size() < 7) {
            std::cerr << "Error: Format error for candidate reads line:" << std::endl << line << std::endl;
            std::exit(1);
        }
        if (f.size() < 9) die("Error: Format error for candidate reads line:\n" + line);   // the reference indexes [8]
        auto num = [&](int k) { return lexical_int_or_die(f[k], "in candidate reads line: " + line); };
        pending_.fusionID = num(0);
        pending_.fragmentIndex = num(1);
        pending_.readEnd = num(2);
        if (f[3] != "0" && f[3] != "1") die("Error: bad boolean '" + f[3] + "' in candidate reads line: " + line);   // lexical_cast<bool>
        pending_.revComp = f[3] == "1";
        pending_.refSplit = std::make_pair(num(4), num(5));
        pending_.readSplit = std::make_pair(num(6), num(7));
        pending_.score = num(8);
        return true;
    }
    std::istream& in_;
    SplitAlignment pending_;
    bool have_ = false;
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
    cmd.add("a", "align", "Split Alignments Filename", "string");
    cmd.add("q", "seq", "Sequence Predictions Filename", "string");
    cmd.add("b", "break", "Breakpoint Predictions Filename", "string");
    cmd.add("p", "predalign", "Predicted Alignments Filename", "string");
    cmd.parse(argc, argv);

    const std::map<int, std::vector<Location>> regions = ReadAlignRegionPairs(cmd.str("regions"));
    std::map<int, SplitAlignmentTask> tasks = CreateTasks(cmd.str("fasta"), cmd.str("exons"), cmd.real("ufrag"), cmd.real("sfrag"),
                                                         cmd.integer("minread"), cmd.integer("maxread"), regions);

    std::ifstream alignFile(cmd.str("align").c_str());
    std::ofstream seqFile(cmd.str("seq").c_str()), breakFile(cmd.str("break").c_str()), predFile(cmd.str("predalign").c_str());
    if (!alignFile.good()) die("Error: Unable to open " + cmd.str("align"));
    if (!seqFile.good()) die("Error: Unable to open " + cmd.str("seq"));
    if (!breakFile.good()) die("Error: Unable to open " + cmd.str("break"));
    if (!predFile.good()) die("Error: Unable to open " + cmd.str("predalign"));

    SortedAlignmentReader reader(alignFile);
    std::vector<SplitAlignment> alignments;
    while (reader.NextGroup(alignments)) {
        const int fusionID = alignments.front().fusionID;
        const SplitAlignmentTask& task = tasks[fusionID];   // operator[]: an unknown id evaluates against an empty task

        // Evaluate (tools/SplitAlignment.cpp:484-594)
        std::map<std::pair<int, int>, int> splitScore;
        for (const SplitAlignment& a : alignments) splitScore[a.refSplit] += a.score;
        int maxScore = -1;
        std::pair<int, int> best;
        for (const auto& kv : splitScore)
            if (kv.second > maxScore) { best = kv.first; maxScore = kv.second; }
        std::string sequence = "N";
        int breakPos[2] = {0, 0}, count = 0;
        double posAvg = -1.0, minAvg = -1.0;
        std::vector<const SplitAlignment*> kept;
        if (maxScore == -1) {
            std::cerr << "Error: Unable to find max score split" << std::endl;
        } else {
            for (const SplitAlignment& a : alignments)
                if (a.refSplit == best) kept.push_back(&a);
            if (!(best.first <= (int)task.mSplitAlignSeq[0].length())) debug_check_failed("bestSplit.first <= mSplitAlignSeq[0].length()");
            if (!(best.second + 1 < (int)task.mSplitAlignSeq[1].length())) debug_check_failed("bestSplit.second + 1 < mSplitAlignSeq[1].length()");
            sequence = task.mSplitRemainderSeq[0] + task.mSplitAlignSeq[0].substr(0, best.first) + "|" +
                       task.mSplitAlignSeq[1].substr(best.second + 1) + task.mSplitRemainderSeq[1];
            breakPos[0] = task.mSplitSeqStrand[0] == PlusStrand ? task.mSplitAlignSeqStart[0] + best.first - 1
                                                                 : task.mSplitAlignSeqStart[0] + task.mSplitAlignSeqLength[0] - best.first;
            breakPos[1] = task.mSplitSeqStrand[1] == PlusStrand ? task.mSplitAlignSeqStart[1] + best.second + 1
                                                                 : task.mSplitAlignSeqStart[1] + task.mSplitAlignSeqLength[1] - best.second - 2;
            double posSum = 0.0, minSum = 0.0;
            for (const SplitAlignment* a : kept) {
                const int left = a->readSplit.first, right = a->readSplit.second;
                const double posRange = (double)(left + right - 2 * minAnchor);
                const double posValue = std::max(0, left - minAnchor);
                const double minRange = std::floor(0.5 * (double)(left + right - 2 * minAnchor));
                const double minValue = std::max(0, std::min(left - minAnchor, right - minAnchor));
                posSum += posValue / posRange;
                minSum += minValue / minRange;
            }
            count = (int)kept.size();
            posAvg = posSum / (double)kept.size();
            minAvg = minSum / kept.size();
        }
        // WriteSequence / WriteBreak / WriteAlignments (:596-624)
        seqFile << fusionID << "\t" << sequence << "\t" << "0" << "\t" << count << "\t" << posAvg << "\t" << minAvg << std::endl;
        for (int ce = 0; ce <= 1; ++ce)
            breakFile << fusionID << "\t" << ce << "\t" << task.mAlignRefName[ce] << "\t"
                      << (task.mAlignStrand[ce] == PlusStrand ? "+" : "-") << "\t" << breakPos[ce] << std::endl;
        for (const SplitAlignment* a : kept) a->Write(predFile);
    }
    return 0;
}
