// minsnap_shard_schedule.h -- the chunk / peer schedule of csp_minsnap_solve_batch_sharded for a batch that is RESIDENT ON ONE
// (root) DEVICE (north_star: "shards across the 8 GPUs of one node with only an RCCL scatter/gather").
// Plain C++ on purpose (no HIP, no RCCL): the schedule is written against a Transport concept so that the product runs it over
// RCCL (minsnap_sharded.hip) and tests/test_shard_schedule.py runs the very same code over a recording in-memory transport
// on the CPU.
//
// Partition: trajectories are independent (minimum_snap.cpp has no cross-trajectory term), so device g of n owns the
// contiguous range [B g / n, B (g+1) / n) and cuts it into `nchunks` contiguous pieces.  Schedule:
//
//     for c in chunks:  scatter(c)      one GROUPED point-to-point exchange: the root sends piece (g, c)'s inputs to every
//                                       peer g (RCCL has no scatter: ncclGroupStart, ncclSend x peers / ncclRecv, ncclGroupEnd);
//                                       536 B per solve at S = 16 -- all of it is queued before the first solve
//     for c in chunks:  solve(g, c) for every device g (the root's own pieces are solved in place, no copy), then
//                       gather(c)       grouped again: every peer sends piece (g, c)'s 3072 B per solve back, ordered after
//                                       ITS solve only -- on a communication stream, so that it overlaps the solve of c + 1
//     finish()                          wait for everything
#pragma once
#include <cstdint>
#include <vector>

namespace csp {
namespace shard {

struct Piece {
    int dev;          // position in the device list (0 .. ndev-1); `root` is one of them
    int chunk;
    int64_t lo, hi;   // trajectories [lo, hi) of the batch
};

inline void shard_range(int64_t B, int ndev, int g, int64_t &lo, int64_t &hi) {
    lo = B * g / ndev;
    hi = B * (g + 1) / ndev;
}

inline Piece piece_of(int64_t B, int ndev, int nchunks, int g, int c) {
    int64_t lo, hi;
    shard_range(B, ndev, g, lo, hi);
    const int64_t n = hi - lo;
    return Piece{g, c, lo + n * c / nchunks, lo + n * (c + 1) / nchunks};
}

// Transport concept:
//   int scatter_begin(int chunk);  int scatter_piece(const Piece &p);  int scatter_end(int chunk);   (peers' pieces only)
//   int solve(const Piece &p);                                                                       (every device)
//   int gather_begin(int chunk);   int gather_piece(const Piece &p);   int gather_end(int chunk);    (peers' pieces only)
//   int finish();
// every call returns 0 or an error code, which ends the schedule.
template <class Transport>
int run(Transport &t, int64_t B, int ndev, int root, int nchunks) {
    if (B <= 0 || ndev <= 0 || root < 0 || root >= ndev || nchunks <= 0) return -1;
    int rc;
    for (int c = 0; c < nchunks; ++c) {
        if ((rc = t.scatter_begin(c)) != 0) return rc;
        for (int g = 0; g < ndev; ++g) {
            if (g == root) continue;
            const Piece p = piece_of(B, ndev, nchunks, g, c);
            if (p.hi > p.lo && (rc = t.scatter_piece(p)) != 0) return rc;
        }
        if ((rc = t.scatter_end(c)) != 0) return rc;
    }
    for (int c = 0; c < nchunks; ++c) {
        for (int g = 0; g < ndev; ++g) {
            const Piece p = piece_of(B, ndev, nchunks, g, c);
            if (p.hi > p.lo && (rc = t.solve(p)) != 0) return rc;
        }
        if ((rc = t.gather_begin(c)) != 0) return rc;
        for (int g = 0; g < ndev; ++g) {
            if (g == root) continue;
            const Piece p = piece_of(B, ndev, nchunks, g, c);
            if (p.hi > p.lo && (rc = t.gather_piece(p)) != 0) return rc;
        }
        if ((rc = t.gather_end(c)) != 0) return rc;
    }
    return t.finish();
}

}  // namespace shard
}  // namespace csp
