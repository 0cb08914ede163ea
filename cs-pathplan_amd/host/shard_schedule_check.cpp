// shard_schedule_check.cpp -- CPU check of the chunk / peer schedule csp_minsnap_solve_batch_sharded runs over RCCL
// (cs-pathplan_amd/csrc/minsnap_shard_schedule.h): the same template, driven over an in-memory RECORDING transport.
// "Devices" are host buffers, a send/recv pair is a memcpy that happens when its group ends, the "solve" is a known
// function of the inputs.  Checked: every trajectory's result reaches the root exactly once and is right; a piece is
// solved only after its inputs arrived and sent back only after it was solved; all scatters are queued before the first
// solve (what lets the gather of chunk i overlap the solve of chunk i + 1 on the real streams); groups are balanced.
// usage: shard_schedule_check B ndev root nchunks   -> exit 0 / 1, one line of statistics on stdout
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../csrc/minsnap_shard_schedule.h"

using csp::shard::Piece;

struct Recording {
    int64_t B;
    int ndev, root, nchunks;
    std::vector<double> in_root, out_root;                 // the caller's buffers on the root
    std::vector<std::vector<double>> in_dev, out_dev;      // per-device images of ITS shard (peers)
    std::vector<std::vector<char>> have_in, solved, sent;  // [dev][chunk]
    std::vector<int> written;                              // per trajectory: how often its result reached the root
    struct Copy { double *dst; const double *src; int64_t n; };
    std::vector<Copy> pending;
    bool in_group = false;
    int solves = 0, scatters_after_first_solve = 0, groups = 0;
    bool fail(const char *why) { std::fprintf(stderr, "schedule check: %s\n", why); return false; }
    bool ok = true;

    static double f(double x) { return 3.0 * x + 1.0; }
    void shard(int g, int64_t &lo, int64_t &hi) const { csp::shard::shard_range(B, ndev, g, lo, hi); }

    int scatter_begin(int) { if (in_group) ok = fail("nested group"); in_group = true; ++groups; return 0; }
    int scatter_piece(const Piece &p) {
        if (!in_group || p.dev == root) ok = fail("scatter outside a group / to the root");
        if (solves) ++scatters_after_first_solve;
        int64_t lo, hi;
        shard(p.dev, lo, hi);
        pending.push_back({in_dev[p.dev].data() + (p.lo - lo), in_root.data() + p.lo, p.hi - p.lo});
        have_in[p.dev][p.chunk] = 2;   // posted; arrives at group end
        return 0;
    }
    int scatter_end(int) {
        for (const Copy &c : pending) std::memcpy(c.dst, c.src, (size_t)c.n * sizeof(double));
        pending.clear();
        for (auto &v : have_in) for (char &x : v) if (x == 2) x = 1;
        in_group = false;
        return 0;
    }
    int solve(const Piece &p) {
        ++solves;
        if (in_group) ok = fail("solve inside a group");
        int64_t lo, hi;
        shard(p.dev, lo, hi);
        if (p.dev == root) {
            for (int64_t i = p.lo; i < p.hi; ++i) { out_root[i] = f(in_root[i]); ++written[i]; }
        } else {
            if (have_in[p.dev][p.chunk] != 1) ok = fail("piece solved before its inputs arrived");
            for (int64_t i = p.lo; i < p.hi; ++i) out_dev[p.dev][i - lo] = f(in_dev[p.dev][i - lo]);
        }
        solved[p.dev][p.chunk] = 1;
        return 0;
    }
    int gather_begin(int) { if (in_group) ok = fail("nested group"); in_group = true; ++groups; return 0; }
    int gather_piece(const Piece &p) {
        if (!in_group || p.dev == root) ok = fail("gather outside a group / from the root");
        if (!solved[p.dev][p.chunk]) ok = fail("piece sent back before it was solved");
        if (sent[p.dev][p.chunk]) ok = fail("piece sent twice");
        sent[p.dev][p.chunk] = 1;
        int64_t lo, hi;
        shard(p.dev, lo, hi);
        pending.push_back({out_root.data() + p.lo, out_dev[p.dev].data() + (p.lo - lo), p.hi - p.lo});
        for (int64_t i = p.lo; i < p.hi; ++i) ++written[i];
        return 0;
    }
    int gather_end(int) { return scatter_end(0); }
    int finish() { if (in_group) ok = fail("unfinished group"); return 0; }
};

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    Recording r;
    r.B = std::atoll(argv[1]); r.ndev = std::atoi(argv[2]); r.root = std::atoi(argv[3]); r.nchunks = std::atoi(argv[4]);
    r.in_root.resize((size_t)r.B); r.out_root.assign((size_t)r.B, -1.0); r.written.assign((size_t)r.B, 0);
    for (int64_t i = 0; i < r.B; ++i) r.in_root[(size_t)i] = 0.25 * (double)i - 7.0;
    r.in_dev.resize((size_t)r.ndev); r.out_dev.resize((size_t)r.ndev);
    r.have_in.assign((size_t)r.ndev, std::vector<char>((size_t)r.nchunks, 0));
    r.solved = r.have_in; r.sent = r.have_in;
    int64_t covered = 0, prev_hi = 0;
    for (int g = 0; g < r.ndev; ++g) {
        int64_t lo, hi;
        r.shard(g, lo, hi);
        if (lo != prev_hi) { std::fprintf(stderr, "shards not contiguous\n"); return 1; }
        prev_hi = hi;
        covered += hi - lo;
        r.in_dev[(size_t)g].assign((size_t)(hi - lo), -99.0);
        r.out_dev[(size_t)g].assign((size_t)(hi - lo), -99.0);
        int64_t plo = lo;
        for (int c = 0; c < r.nchunks; ++c) {   // a shard's pieces tile it
            const Piece p = csp::shard::piece_of(r.B, r.ndev, r.nchunks, g, c);
            if (p.lo != plo || p.hi < p.lo) { std::fprintf(stderr, "pieces do not tile the shard\n"); return 1; }
            plo = p.hi;
        }
        if (plo != hi) { std::fprintf(stderr, "pieces do not cover the shard\n"); return 1; }
    }
    if (covered != r.B || prev_hi != r.B) { std::fprintf(stderr, "shards do not cover the batch\n"); return 1; }
    const int rc = csp::shard::run(r, r.B, r.ndev, r.root, r.nchunks);
    if (rc != 0) { std::fprintf(stderr, "run() returned %d\n", rc); return 1; }
    for (int64_t i = 0; i < r.B; ++i) {
        if (r.written[(size_t)i] != 1) { std::fprintf(stderr, "trajectory %lld reached the root %d times\n", (long long)i, r.written[(size_t)i]); return 1; }
        if (r.out_root[(size_t)i] != Recording::f(r.in_root[(size_t)i])) { std::fprintf(stderr, "trajectory %lld wrong\n", (long long)i); return 1; }
    }
    if (r.scatters_after_first_solve) { std::fprintf(stderr, "a scatter was queued after the first solve\n"); return 1; }
    std::printf("ok B=%lld ndev=%d root=%d nchunks=%d groups=%d solves=%d\n", (long long)r.B, r.ndev, r.root, r.nchunks, r.groups, r.solves);
    return r.ok ? 0 : 1;
}
