// evaluate.hpp — what evalsplitalign and the fused mode of dosplitalign share: one line of the split alignment file and
// SplitAlignmentTask::Evaluate with the three writers (tools/SplitAlignment.cpp:305-369, :484-624).
#pragma once
#include "defuse_host.hpp"

namespace defuse {


constexpr int minAnchor = 4;   // tools/SplitAlignment.cpp:29

struct SplitAlignment {
    int fusionID = 0, fragmentIndex = 0, readEnd = 0, revComp = 0;
    std::pair<int, int> refSplit, readSplit;
    int score = 0;
    void Write(std::string& out) const    // tools/SplitAlignment.cpp:305-317
    {
        for (int v : {fusionID, fragmentIndex, readEnd, revComp, refSplit.first, refSplit.second, readSplit.first, readSplit.second, score}) {
            append_int(out, v);
            out += '\t';
        }
        out += '\n';
    }
};

[[noreturn]] inline void debug_check_failed(const char* expr)
{
    std::cerr << "Error: " << expr << " failed in SplitAlignmentTask::Evaluate" << std::endl;
    std::exit(1);
}

// One line of the sorted alignment file (SplitAlignment::ReadSortedAlignments, tools/SplitAlignment.cpp:319-369), parsed in
// place.  Returns an error text (empty = fine); the messages are the reference's.
inline std::string parse_line(const char* line, size_t len, SplitAlignment& a, bool& id_read)
{
    id_read = false;
    const char* fs[16];
    int nf = 0;
    fs[nf++] = line;
    const char* end = line + len;
    for (const char* p = line; nf < 15;) {
        const char* tab = (const char*)memchr(p, '\t', (size_t)(end - p));
        if (!tab) break;
        fs[nf++] = p = tab + 1;
    }
    fs[nf] = end + 1;                                   // field k is [fs[k], fs[k+1] - 1)
    const std::string text(line, len);
    // what the reference's look-ahead checks before it decides that a line opens the next group: seven fields and the id
    if (nf >= 7 && field_int(fs[0], (size_t)(fs[1] - 1 - fs[0]), a.fusionID)) id_read = true;
    if (nf < 9) return "Error: Format error for candidate reads line:\n" + text;      // < 7 in the reference, which then indexes [8]
    int v[9];
    for (int k = 0; k < 9; ++k) {
        const size_t n = (size_t)(fs[k + 1] - 1 - fs[k]);
        if (k == 3) {
            if (n != 1 || (fs[k][0] != '0' && fs[k][0] != '1'))
                return "Error: bad boolean '" + std::string(fs[k], n) + "' in candidate reads line: " + text;   // lexical_cast<bool>
            v[k] = fs[k][0] == '1';
        } else if (!field_int(fs[k], n, v[k])) {
            return "Error: bad integer '" + std::string(fs[k], n) + "' in candidate reads line: " + text;
        }
    }
    a.fusionID = v[0]; a.fragmentIndex = v[1]; a.readEnd = v[2]; a.revComp = v[3];
    a.refSplit = std::make_pair(v[4], v[5]);
    a.readSplit = std::make_pair(v[6], v[7]);
    a.score = v[8];
    return std::string();
}

inline void append_double(std::string& buf, double x)          // operator<<(double) at the stream's default precision: %g
{
    char tmp[40];
    buf.append(tmp, (size_t)snprintf(tmp, sizeof tmp, "%g", x));
}


struct EvalTexts { std::string seq, brk, pred; };

// Evaluate (tools/SplitAlignment.cpp:484-594) of one group of alignments (one fusion id) and WriteSequence / WriteBreak /
// WriteAlignments (:596-624) appended to out.  kept and splitScore are scratch.
inline void EvaluateGroup(const SplitAlignmentTask& task, const std::vector<SplitAlignment>& alignments, EvalTexts& out,
                          std::vector<const SplitAlignment*>& kept, std::map<std::pair<int, int>, int>& splitScore)
{
    const int fusionID = alignments.front().fusionID;
    splitScore.clear();
    for (const SplitAlignment& a : alignments) splitScore[a.refSplit] += a.score;
    int maxScore = -1;
    std::pair<int, int> best;
    for (const auto& kv : splitScore)
        if (kv.second > maxScore) { best = kv.first; maxScore = kv.second; }
    std::string sequence = "N";
    int breakPos[2] = {0, 0}, count = 0;
    double posAvg = -1.0, minAvg = -1.0;
    kept.clear();
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
    append_int(out.seq, fusionID);
    out.seq += '\t'; out.seq += sequence; out.seq += "\t0\t";
    append_int(out.seq, count);
    out.seq += '\t'; append_double(out.seq, posAvg);
    out.seq += '\t'; append_double(out.seq, minAvg);
    out.seq += '\n';
    for (int ce = 0; ce <= 1; ++ce) {
        append_int(out.brk, fusionID);
        out.brk += '\t'; append_int(out.brk, ce);
        out.brk += '\t'; out.brk += task.mAlignRefName[ce];
        out.brk += (task.mAlignStrand[ce] == PlusStrand ? "\t+\t" : "\t-\t");
        append_int(out.brk, breakPos[ce]);
        out.brk += '\n';
    }
    for (const SplitAlignment* a : kept) a->Write(out.pred);
}

}  // namespace defuse
