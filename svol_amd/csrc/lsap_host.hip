// Host-side rectangular linear sum assignment: svol_lsap_solve (include/svol_hip.h).
//
// SURVEY.md §8(b) lists a HOST entry beside the device-batched one: it is what a reference-side caller that keeps its cost matrix
// on the CPU (matcher.py:86 `C.cpu()`, then scipy.optimize.linear_sum_assignment at matcher.py:93,158) binds instead of scipy.  The
// training / evaluation path of this build never calls it (svol_lsap_batched solves every layer's problems on the device without a
// host round trip); it exists for that caller and as a second, structurally independent implementation the device kernel is tested
// against (tests/test_abi.py::test_host_lsap_known_answers).
//
// Semantics = scipy's (third-party, 1.15.3 here; algorithm: Crouse 2016, shortest augmenting paths with dual variables in fp64):
//   * more rows than columns: solved on the transpose;
//   * NaN or -inf entries are invalid (scipy raises ValueError), +inf is an ordinary "forbidden" entry;
//   * among the not-yet-scanned columns the one with the lowest path cost wins; a column REPLACES the running best when it is
//     strictly lower, or equal and still unassigned (so of several equal unassigned columns the last one scanned wins, of several
//     equal assigned ones the first) — the scan order starts as nc-1, nc-2, ..., 0 and removing a column moves the order's last
//     entry into its slot, exactly as scipy's loop does;
//   * output pairs sorted by row.
#include <cmath>
#include <cstdint>
#include <limits>
#include <numeric>
#include <algorithm>
#include <vector>

#include "../../include/svol_hip.h"

namespace {

struct Sap {
    int64_t nr, nc;                 // nr <= nc
    std::vector<double> c;          // [nr, nc]
    std::vector<double> u, v, dist; // duals, shortest path cost to each column
    std::vector<int64_t> pred, col_of_row, row_of_col, order;
    std::vector<uint8_t> row_seen, col_seen;

    Sap(int64_t r, int64_t k) : nr(r), nc(k), c((size_t)(r * k)), u((size_t)r, 0.0), v((size_t)k, 0.0), dist((size_t)k),
                                pred((size_t)k, -1), col_of_row((size_t)r, -1), row_of_col((size_t)k, -1), order((size_t)k),
                                row_seen((size_t)r), col_seen((size_t)k) {}

    // grows one shortest augmenting path from `start`; returns its sink column or -1 (infeasible), the path's cost in `reach`
    int64_t grow(int64_t start, double& reach) {
        const double inf = std::numeric_limits<double>::infinity();
        std::fill(dist.begin(), dist.end(), inf);
        std::fill(row_seen.begin(), row_seen.end(), 0);
        std::fill(col_seen.begin(), col_seen.end(), 0);
        for (int64_t t = 0; t < nc; ++t) order[(size_t)t] = nc - 1 - t;
        int64_t live = nc, row = start;
        double base = 0.0;
        for (;;) {
            row_seen[(size_t)row] = 1;
            const double* crow = &c[(size_t)(row * nc)];
            const double ur = u[(size_t)row];
            int64_t best_slot = -1;
            double best = inf;
            for (int64_t t = 0; t < live; ++t) {
                const int64_t j = order[(size_t)t];
                const double through = base + crow[j] - ur - v[(size_t)j];   // ((base + c) - u) - v: scipy's association, bit for bit
                if (through < dist[(size_t)j]) { dist[(size_t)j] = through; pred[(size_t)j] = row; }
                const double dj = dist[(size_t)j];
                if (dj < best || (dj == best && row_of_col[(size_t)j] < 0)) { best = dj; best_slot = t; }
            }
            base = best;
            if (best == inf) return -1;
            const int64_t j = order[(size_t)best_slot];
            col_seen[(size_t)j] = 1;
            order[(size_t)best_slot] = order[(size_t)(--live)];
            if (row_of_col[(size_t)j] < 0) { reach = base; return j; }
            row = row_of_col[(size_t)j];
        }
    }

    bool solve() {
        for (int64_t r = 0; r < nr; ++r) {
            double reach = 0.0;
            const int64_t sink = grow(r, reach);
            if (sink < 0) return false;
            u[(size_t)r] += reach;
            for (int64_t i = 0; i < nr; ++i)
                if (row_seen[(size_t)i] && i != r) u[(size_t)i] += reach - dist[(size_t)col_of_row[(size_t)i]];
            for (int64_t j = 0; j < nc; ++j)
                if (col_seen[(size_t)j]) v[(size_t)j] -= reach - dist[(size_t)j];
            for (int64_t j = sink;;) {   // flip the path
                const int64_t i = pred[(size_t)j];
                row_of_col[(size_t)j] = i;
                std::swap(col_of_row[(size_t)i], j);
                if (i == r) break;
            }
        }
        return true;
    }
};

}  // namespace

extern "C" int svol_lsap_solve(const double* cost, int64_t nr, int64_t nc, int64_t* rows, int64_t* cols) {
    if (nr < 0 || nc < 0 || std::min(nr, nc) > 0x7fffffff) return SVOL_E_INVALID;
    if (nr == 0 || nc == 0) return 0;
    if (!cost || !rows || !cols) return SVOL_E_INVALID;
    const bool tall = nr > nc;
    Sap s(tall ? nc : nr, tall ? nr : nc);
    for (int64_t i = 0; i < nr; ++i)
        for (int64_t j = 0; j < nc; ++j) {
            const double x = cost[i * nc + j];
            if (x != x || x == -std::numeric_limits<double>::infinity()) return -1;   // scipy: "matrix contains invalid numeric entries"
            s.c[(size_t)(tall ? j * nr + i : i * nc + j)] = x;
        }
    if (!s.solve()) return -2;   // scipy: "cost matrix is infeasible"
    const int64_t n = s.nr;
    if (!tall) {
        for (int64_t i = 0; i < n; ++i) { rows[i] = i; cols[i] = s.col_of_row[(size_t)i]; }
    } else {   // rows of the transposed problem are the caller's columns: report pairs by ascending caller row
        std::vector<int64_t> by((size_t)n);
        std::iota(by.begin(), by.end(), (int64_t)0);
        std::stable_sort(by.begin(), by.end(), [&](int64_t a, int64_t b) { return s.col_of_row[(size_t)a] < s.col_of_row[(size_t)b]; });
        for (int64_t k = 0; k < n; ++k) { rows[k] = s.col_of_row[(size_t)by[(size_t)k]]; cols[k] = by[(size_t)k]; }
    }
    return (int)n;
}
