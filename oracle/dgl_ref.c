/*
 * Plain-C CPU restatement of the integer/index and distance arithmetic of the
 * DGL decode step.  TEST INFRASTRUCTURE ONLY: linked/loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
 * library.
 *
 * What it restates (paths relative to the reference checkout):
 *   - the pairwise distance the reference gets from torch.cdist (TSP/TSPEnv.py:125,228;
 *     CVRP/CVRPEnv.py:262,475).  For more than 25 points ATen evaluates it as a
 *     4-term matrix product [-2x_i, -2y_i, |p_i|^2, 1] . [x_j, y_j, 1, |p_j|^2],
 *     which in IEEE-754 single precision is the chain
 *         n_i  = fl(fl(x_i*x_i) + fl(y_i*y_i))
 *         acc  = fl((-2 x_i) * x_j)
 *         acc  = fma(-2 y_i, y_j, acc)
 *         acc  = fl(acc + n_i)
 *         acc  = fl(acc + n_j)
 *         d_ij = sqrt(max(acc, 0))
 *     (asymmetric, non-zero diagonal).  For at most 25 points ATen evaluates every pair directly:
 *         d_ij = sqrt(fma(dy, dy, fl(dx * dx))),  dx = x_i - x_j, dy = y_i - y_j
 *     (symmetric, zero diagonal).  Both checked bit-for-bit against torch.cdist in
 *     tests/test_oracle_c.py.
 *   - row mean / population std of that matrix (TSP/TSPEnv.py:234-235), accumulated in
 *     double and rounded once (torch sums the mean in float with a cascade order that
 *     depends on the CPU's SIMD width: equal to within a few ulp, not bit for bit).
 *   - masked k-nearest selection (TSP/TSPModel.py:79-91, CVRP/CVRPModel.py:88-125):
 *     key = d(last, j) + 100 for visited j, k smallest keys, lowest index wins a tie,
 *     result ascending by node index.
 *   - the running-feature recurrences of env.step (TSP/TSPEnv.py:386,408-409;
 *     CVRP/CVRPEnv.py:711,733-738) and the capacity / feasibility update
 *     (CVRP/CVRPEnv.py:612-637).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf must be a real fused
 * multiply-add, which -mfma or glibc's software fmaf both guarantee).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DGLREF_API __attribute__((visibility("default")))

static inline float sqnorm(float x, float y) { return x * x + y * y; }

/* Which square root torch.sqrt / torch.cdist's final root stand for.  NULL (the default): the correctly rounded root.
 * Otherwise the root a CPU build of torch linked against MKL returns on an AVX-512 machine - the setting the golden
 * vectors were captured under: MKL refines the instruction VRSQRT14's estimate y once, s = x y, result =
 * fma(fma(-s, s, x), y / 2, s).  `table` is the recording of that instruction over [1, 4) reduced to 2^16 entries
 * (index = exponent parity << 15 | mantissa >> 8, value = (estimate's bits - 0x3F000000) >> 7; the estimate depends on no
 * lower mantissa bit, is exactly 1 for the exact powers of 4 and scales exactly with powers of 4), as
 * dgl_amd/sqrt_model.py::rsqrt14_table expands it from dgl_amd/data/rsqrt14_avx512.npz; the caller keeps it alive.
 * This restatement looks the estimate up in the recording; the device library evaluates it from constants. */
static const uint16_t *g_rsqrt14 = NULL;
DGLREF_API void dglref_set_rsqrt14(const uint16_t *table) { g_rsqrt14 = table; }

static float model_sqrtf(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    if (g_rsqrt14 == NULL || b == 0u || b >= 0x7F800000u || b < 0x00800000u) return sqrtf(x);   /* 0, negative, inf, NaN, denormal */
    int e = (int)(b >> 23) - 127, p = e & 1, q = (e - p) / 2;
    uint32_t mant = b & 0x7FFFFFu;
    uint32_t yb = (p == 0 && mant == 0u) ? 0x3F800000u : 0x3F000000u + ((uint32_t)g_rsqrt14[((uint32_t)p << 15) | (mant >> 8)] << 7);
    yb -= (uint32_t)q << 23;
    float y;
    memcpy(&y, &yb, 4);
    float s = x * y, h = 0.5f * y;
    return fmaf(fmaf(-s, s, x), h, s);
}
/* out[i] = the selected model's root of x[i] */
DGLREF_API void dglref_model_sqrt(const float *x, int n, float *out) {
    for (int i = 0; i < n; ++i) out[i] = model_sqrtf(x[i]);
}

/* d(i, j) of the matrix-product form of cdist: i is the ROW (query), j the column. */
static inline float chain_dist(float xi, float yi, float ni, float xj, float yj, float nj) {
    float acc = (-2.0f * xi) * xj;
    acc = fmaf(-2.0f * yi, yj, acc);
    acc = acc + ni;
    acc = acc + nj;
    return model_sqrtf(acc > 0.0f ? acc : 0.0f);
}

/* the per-pair form cdist takes for at most DIRECT_MAX_POINTS points (the scalar sqrt: correctly rounded under either model) */
#define DIRECT_MAX_POINTS 25
static inline float direct_dist(float xi, float yi, float xj, float yj) {
    float dx = xi - xj, dy = yi - yj;
    return sqrtf(fmaf(dy, dy, dx * dx));
}

/* out[j] = d(i, j), j = 0..n-1 (row i of the matrix) */
DGLREF_API void dglref_dist_row(const float *xy, int n, int i, float *out) {
    float xi = xy[2 * i], yi = xy[2 * i + 1], ni = sqnorm(xi, yi);
    for (int j = 0; j < n; ++j) {
        float xj = xy[2 * j], yj = xy[2 * j + 1];
        out[j] = n <= DIRECT_MAX_POINTS ? direct_dist(xi, yi, xj, yj) : chain_dist(xi, yi, ni, xj, yj, sqnorm(xj, yj));
    }
}

/* out[j] = d(j, c), j = 0..n-1 (column c of the matrix) */
DGLREF_API void dglref_dist_col(const float *xy, int n, int c, float *out) {
    float xc = xy[2 * c], yc = xy[2 * c + 1], nc = sqnorm(xc, yc);
    for (int j = 0; j < n; ++j) {
        float xj = xy[2 * j], yj = xy[2 * j + 1];
        out[j] = n <= DIRECT_MAX_POINTS ? direct_dist(xj, yj, xc, yc) : chain_dist(xj, yj, sqnorm(xj, yj), xc, yc, nc);
    }
}

/* avg[i] = mean_j d(i,j), sd[i] = population std_j d(i,j) for one instance */
DGLREF_API void dglref_row_mean_std(const float *xy, int n, float *avg, float *sd) {
    float *row = (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        dglref_dist_row(xy, n, i, row);
        double s = 0.0, m;
        for (int j = 0; j < n; ++j) s += row[j];
        m = s / n;
        avg[i] = (float)m;
        double q = 0.0;
        for (int j = 0; j < n; ++j) { double e = row[j] - m; q += e * e; }
        sd[i] = (float)sqrt(q / n);
    }
    free(row);
}

/* k smallest keys of key[0..n), ties -> lowest index; result ascending by index.
 * Selection by repeated minimum: O(n k), fine for a checker. */
static void select_k(const float *key, int n, int k, int32_t *idx_out) {
    uint8_t *taken = (uint8_t *)calloc((size_t)n, 1);
    for (int s = 0; s < k; ++s) {
        int best = -1;
        for (int j = 0; j < n; ++j)
            if (!taken[j] && (best < 0 || key[j] < key[best])) best = j;
        taken[best] = 1;
    }
    int c = 0;
    for (int j = 0; j < n && c < k; ++j)
        if (taken[j]) idx_out[c++] = j;
    free(taken);
}

/* TSP neighbourhood: k = min(knn, #unvisited) nearest unvisited nodes of `last`.
 * visited: n bytes (0/1).  Returns k. */
DGLREF_API int dglref_tsp_knn(const float *xy, int n, int last, const uint8_t *visited, int knn,
                              int32_t *idx_out) {
    int unvisited = 0;
    for (int j = 0; j < n; ++j) unvisited += !visited[j];
    int k = knn < unvisited ? knn : unvisited;
    float *key = (float *)malloc(sizeof(float) * (size_t)n);
    dglref_dist_row(xy, n, last, key);
    for (int j = 0; j < n; ++j)
        if (visited[j]) key[j] = key[j] + 100.0f;
    select_k(key, n, k, idx_out);
    free(key);
    return k;
}

/* CVRP neighbourhoods over nodes 0..n (n customers, node 0 = depot; xy has n+1 points).
 * kd' = min(depot_knn, #unvisited customers); last set = min(knn, kd') nearest of `last`,
 * depot set = kd' nearest of the depot; the depot itself is never a candidate.
 * Returns kd'; *k_last_out receives the size of the last set. */
DGLREF_API int dglref_cvrp_knn(const float *xy, int n, int last, const uint8_t *visited, int knn,
                               int depot_knn, int32_t *idx_last, int32_t *idx_depot, int *k_last_out) {
    int m = n + 1, unvisited = 0;
    for (int j = 1; j < m; ++j) unvisited += !visited[j];
    int kd = depot_knn < unvisited ? depot_knn : unvisited;
    int kl = knn < kd ? knn : kd;
    float *key = (float *)malloc(sizeof(float) * (size_t)m);
    dglref_dist_row(xy, m, last, key);
    for (int j = 0; j < m; ++j)
        if (visited[j] || j == 0) key[j] = key[j] + 100.0f;
    select_k(key, m, kl, idx_last);
    dglref_dist_row(xy, m, 0, key);
    for (int j = 0; j < m; ++j)
        if (visited[j] || j == 0) key[j] = key[j] + 100.0f;
    select_k(key, m, kd, idx_depot);
    free(key);
    *k_last_out = kl;
    return kd;
}

/* One env.step of the running features for one row: m nodes, `cur` just selected,
 * u = number of nodes still unvisited AFTER this step.  clamp != 0 is the CVRP variant
 * (negative radicands set to 0).  d_out[j] = d(j, cur). */
DGLREF_API void dglref_step_features(const float *xy, int m, int cur, int u, int clamp, float *avg,
                                     float *sd, float *d_out) {
    dglref_dist_col(xy, m, cur, d_out);
    float up1 = (float)(u + 1), uf = (float)u;
    for (int j = 0; j < m; ++j) {
        float d = d_out[j], a0 = avg[j];
        avg[j] = (a0 * up1 - d) / uf;
        float e = d - a0;
        float item = (sd[j] * sd[j] * up1 - e * e) / uf;
        if (clamp && item < 0.0f) item = 0.0f;
        sd[j] = model_sqrtf(item);
    }
}

/* sin/cos of the direction from node j to `dest` as the reference defines them
 * (TSP/TSPEnv.py:448-471): (y_dest - y_j) / (d(j,dest) + 1e-7), (x_dest - x_j) / (...). */
DGLREF_API void dglref_direction(const float *xy, int m, int dest, float *sin_out, float *cos_out) {
    float *d = (float *)malloc(sizeof(float) * (size_t)m);
    dglref_dist_col(xy, m, dest, d);
    for (int j = 0; j < m; ++j) {
        float den = d[j] + 1e-7f;
        sin_out[j] = (xy[2 * dest + 1] - xy[2 * j + 1]) / den;
        cos_out[j] = (xy[2 * dest] - xy[2 * j]) / den;
    }
    free(d);
}

/* CVRP action decoding + capacity + feasibility (CVRP/CVRPEnv.py:612-637) for one row.
 * demand: n+1 floats (depot 0).  infeasible_out[j] = 1 where capacity + 1e-6 < demand[j].
 * Returns the node; *flag_out = 1 when the move goes via the depot. */
DGLREF_API int dglref_cvrp_apply(int n, int action, const float *demand, float *capacity,
                                 uint8_t *infeasible_out, int *flag_out) {
    int via = action > n;
    int node = via ? action - (n + 1) : action;
    *capacity = via ? 1.0f - demand[node] : *capacity - demand[node];
    float c = *capacity + 0.000001f;
    for (int j = 0; j <= n; ++j) infeasible_out[j] = c < demand[j];
    *flag_out = via;
    return node;
}

/* Closed TSP tour length, direct formula sqrt(fl(dx^2)+fl(dy^2)) per segment, summed in
 * double (the reference sums in float; compare with a tolerance). */
DGLREF_API double dglref_tsp_length(const float *xy, const int32_t *tour, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int a = tour[i], b = tour[(i + 1) % n];
        float dx = xy[2 * a] - xy[2 * b], dy = xy[2 * a + 1] - xy[2 * b + 1];
        s += sqrtf(dx * dx + dy * dy);
    }
    return s;
}

/* CVRP route length (CVRP/CVRPEnv.py:951-993): step i goes prev->cur directly when
 * flag[i]==0, else prev->depot->cur; step 0 always starts at the depot; plus the
 * closing leg last->depot.  Summed in double. */
DGLREF_API double dglref_cvrp_length(const float *xy, const int32_t *nodes, const int8_t *flags, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int cur = nodes[i], prev = nodes[(i + n - 1) % n];
        int mid = flags[i] ? 0 : cur;
        int mid2 = (i == 0) ? 0 : mid;
        float ax = xy[2 * cur] - xy[2 * mid], ay = xy[2 * cur + 1] - xy[2 * mid + 1];
        float bx = xy[2 * prev] - xy[2 * mid2], by = xy[2 * prev + 1] - xy[2 * mid2 + 1];
        s += sqrtf(ax * ax + ay * ay) + sqrtf(bx * bx + by * by);
    }
    return s;
}
