// localalign — drop-in replacement of the reference tool (tools/localalign.cpp:31-92): same command
// line (-m match, -x mismatch, -g gap, optional -t threshold), stdin lines `id \t reference \t sequence`,
// stdout lines `id \t score \t percent` for the pairs whose percent reaches the threshold.  The scores
// (SimpleAligner::Align, tools/SimpleAligner.cpp:24-64) come from the GPU through include/defuse_la.h;
// there is no CPU fallback: without a HIP device the tool exits 1.
#include "../include/defuse_dsa.h"
#include "../include/defuse_la.h"
#include "defuse_host.hpp"

#include <limits>

using namespace defuse;

namespace {

struct Batch {
    std::vector<std::string> ids;
    std::vector<la_item> items;
    std::vector<uint8_t> pool;
};

// scores the batch and writes its lines; the reference prints each line as soon as it has read it, so a
// later input error still leaves the earlier lines on stdout
void flush(Batch& b, int matchScore, int misMatchScore, int gapScore, double threshold)
{
    if (b.items.empty()) return;
    std::vector<int32_t> scores(b.items.size());
    // the smallest score a line needs to be printed at all: below it the device may stop early
    std::vector<int32_t> need(b.items.size());
    for (size_t k = 0; k < b.items.size(); ++k) {
        const int maxScore = (size_t)b.items[k].seq_len * matchScore;
        int32_t s = std::numeric_limits<int32_t>::min();
        if (maxScore > 0 && threshold > 0.0) {
            s = (int32_t)std::min<double>(std::ceil(threshold * (double)maxScore), 2147483000.0);
            while (s > 0 && !((double)(s - 1) / (double)maxScore < threshold)) --s;     // exactly the test of :89
            while ((double)s / (double)maxScore < threshold) ++s;
        }
        need[k] = s;
    }
    const int device = dsa_pick_device();                  // as the other tools: DEFUSE_GPU, else pid mod device count
    if (la_align_batch_min(device, matchScore, misMatchScore, gapScore, b.pool.data(), (int64_t)b.pool.size(), b.items.data(),
                           (int64_t)b.items.size(), need.data(), scores.data(), nullptr) != 0)
        die(std::string("Error: GPU alignment failed: ") + la_last_error());
    std::string out;
    char num[40];
    for (size_t k = 0; k < b.items.size(); ++k) {
        const int score = scores[k];
        const int maxScore = (size_t)b.items[k].seq_len * matchScore;      // tools/localalign.cpp:86
        const double percent = (double)score / (double)maxScore;
        if (percent < threshold) continue;
        out += b.ids[k];
        out += '\t';
        append_int(out, score);
        out += '\t';
        out.append(num, (size_t)snprintf(num, sizeof num, "%g", percent));   // operator<<(double): six significant digits
        out += '\n';
        if (out.size() > (1u << 22)) { fwrite(out.data(), 1, out.size(), stdout); out.clear(); }
    }
    fwrite(out.data(), 1, out.size(), stdout);
    fflush(stdout);
    b = Batch();
}

}  // namespace

int main(int argc, char* argv[])
{
    CmdLine cmd("Local realignment tool");
    cmd.add("m", "match", "Match Score", "int");
    cmd.add("x", "mismatch", "Mismatch Score", "int");
    cmd.add("g", "gap", "Gap Score", "int");
    cmd.add_optional("t", "threshold", "Percent Perfect Threshold", "float", "0");
    cmd.parse(argc, argv);
    const int matchScore = cmd.integer("match"), misMatchScore = cmd.integer("mismatch"), gapScore = cmd.integer("gap");
    const double threshold = cmd.real("threshold");

    Batch batch;
    const size_t flush_bytes = (size_t)1 << 30;           // bounds host memory on very large inputs
    LineReader reader(stdin);
    const char* line;
    size_t len;
    int lineNumber = 0;
    while (reader.next(line, len)) {
        lineNumber++;
        if (len == 0) {
            flush(batch, matchScore, misMatchScore, gapScore, threshold);
            std::cerr << "Error: Empty line " << lineNumber << std::endl;
            return 1;
        }
        const char* end = line + len;
        const char* t1 = (const char*)memchr(line, '\t', len);
        const char* t2 = t1 ? (const char*)memchr(t1 + 1, '\t', (size_t)(end - t1 - 1)) : nullptr;
        if (!t2) {
            flush(batch, matchScore, misMatchScore, gapScore, threshold);
            std::cerr << "Error: Format error for line " << lineNumber << std::endl;
            return 1;
        }
        const char* t3 = (const char*)memchr(t2 + 1, '\t', (size_t)(end - t2 - 1));     // further fields are ignored
        if (!t3) t3 = end;
        la_item it;
        it.ref_off = (int64_t)batch.pool.size();
        it.ref_len = (int32_t)(t2 - t1 - 1);
        batch.pool.insert(batch.pool.end(), t1 + 1, t2);
        it.seq_off = (int64_t)batch.pool.size();
        it.seq_len = (int32_t)(t3 - t2 - 1);
        batch.pool.insert(batch.pool.end(), t2 + 1, t3);
        batch.items.push_back(it);
        batch.ids.emplace_back(line, (size_t)(t1 - line));
        if (batch.pool.size() >= flush_bytes) flush(batch, matchScore, misMatchScore, gapScore, threshold);
    }
    flush(batch, matchScore, misMatchScore, gapScore, threshold);
    return 0;
}
