/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * CPU restatement of the rectangular linear-sum-assignment solver the
 * reference calls at lib/modeling/matcher.py:93 and :158
 * (scipy.optimize.linear_sum_assignment; third-party, `scipy` unpinned in the
 * reference's requirements.txt:3, 1.15.3 in the build image; its C++ source is
 * not on disk).  Algorithm: D. F. Crouse, "On implementing 2D rectangular
 * assignment algorithms", IEEE T-AES 52(4), 2016 — shortest augmenting path
 * with dual variables in fp64, tall matrices solved transposed, candidate
 * columns scanned in REVERSE initial order, tie rule "strictly lower, or equal
 * and the column is still unassigned".  Pinned by
 * tests/golden/lsap_known_answers.npz (answers produced by scipy 1.15.3) and
 * by live comparison with scipy in tests/test_lsap_oracle.py.
 *
 * Return: number of assigned pairs (min(nr,nc)) >= 0, -1 invalid entries
 * (NaN / -inf; scipy raises ValueError), -2 infeasible.
 * rows[] ascending; (rows[k], cols[k]) is the k-th pair.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t augment(int64_t nc, const double *cost, const double *u, const double *v,
                       int64_t *path, const int64_t *row4col, double *spc, int64_t i,
                       char *SR, char *SC, int64_t *remaining, int64_t nr, double *p_min)
{
    double min_val = 0.0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    memset(SR, 0, (size_t)nr);
    memset(SC, 0, (size_t)nc);
    for (int64_t j = 0; j < nc; ++j) spc[j] = INFINITY;

    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; ++it) {
            int64_t j = remaining[it];
            double r = min_val + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) {
                path[j] = i;
                spc[j] = r;
            }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        min_val = lowest;
        if (min_val == INFINITY) return -1;
        int64_t j = remaining[index];
        if (row4col[j] == -1) sink = j;
        else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = min_val;
    return sink;
}

typedef struct { int64_t key, idx; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    int64_t x = ((const kv_t *)a)->key, y = ((const kv_t *)b)->key;
    return (x > y) - (x < y);
}

int64_t svol_oracle_lsap_f64(const double *cost_in, int64_t nr, int64_t nc, int64_t *rows, int64_t *cols)
{
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    double *cost = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
    if (transpose) {
        for (int64_t i = 0; i < nr; ++i)
            for (int64_t j = 0; j < nc; ++j) cost[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
    } else {
        memcpy(cost, cost_in, sizeof(double) * (size_t)(nr * nc));
    }
    for (int64_t k = 0; k < nr * nc; ++k)
        if (cost[k] != cost[k] || cost[k] == -INFINITY) { free(cost); return -1; }

    double *u = (double *)calloc((size_t)nr, sizeof(double));
    double *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    char *SR = (char *)malloc((size_t)nr), *SC = (char *)malloc((size_t)nc);
    for (int64_t j = 0; j < nc; ++j) { path[j] = -1; row4col[j] = -1; }
    for (int64_t i = 0; i < nr; ++i) col4row[i] = -1;

    int64_t ret = nr;
    for (int64_t cur = 0; cur < nr; ++cur) {
        double min_val;
        int64_t sink = augment(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, nr, &min_val);
        if (sink < 0) { ret = -2; break; }
        u[cur] += min_val;
        for (int64_t i = 0; i < nr; ++i)
            if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
        for (int64_t j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= min_val - spc[j];
        int64_t j = sink;
        for (;;) {
            int64_t i = path[j];
            row4col[j] = i;
            int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    if (ret >= 0) {
        if (transpose) {
            kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)nr);
            for (int64_t i = 0; i < nr; ++i) { kv[i].key = col4row[i]; kv[i].idx = i; }
            qsort(kv, (size_t)nr, sizeof(kv_t), kv_cmp);
            for (int64_t i = 0; i < nr; ++i) { rows[i] = kv[i].key; cols[i] = kv[i].idx; }
            free(kv);
        } else {
            for (int64_t i = 0; i < nr; ++i) { rows[i] = i; cols[i] = col4row[i]; }
        }
    }
    free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining);
    free(SR); free(SC);
    return ret;
}

/* fp32 cost entry: scipy converts to float64 first (np.asarray(..., dtype=float64)). */
int64_t svol_oracle_lsap_f32(const float *cost_in, int64_t nr, int64_t nc, int64_t *rows, int64_t *cols)
{
    if (nr == 0 || nc == 0) return 0;
    double *c = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
    for (int64_t k = 0; k < nr * nc; ++k) c[k] = (double)cost_in[k];
    int64_t r = svol_oracle_lsap_f64(c, nr, nc, rows, cols);
    free(c);
    return r;
}
