/*
 * dgl_amd.h - C ABI of the MI355X (gfx950) decode-step library for DGL
 * (Dynamic Global-Local aggregation TSP/CVRP solver).
 *
 * The reference has no FFI of its own: its hot path is a Python object protocol
 * (model.forward / env.step / Beamsearch.advance).  Each entry point below replaces
 * the body of one or more of those methods; the file:line of the reference code it
 * stands in for is given with each declaration (paths relative to the reference
 * checkout).  The Python classes in dgl_amd/ that keep the reference's signatures are
 * thin callers of these functions.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless stated otherwise;
 *  - nothing is allocated or freed inside any entry point;
 *  - every call is asynchronous on the given HIP stream (a hipStream_t passed as
 *    void*; NULL = the default stream) and re-entrant per stream;
 *  - return value 0 = success, negative = error (see dgl_last_error_string());
 *    no C++ exception crosses the boundary;
 *  - rows: R = B * rows_per_instance; row r belongs to instance r / rows_per_instance
 *    (rows_per_instance = pomo * beam, TSP/TSPEnv.py:230-232);
 *  - node ids and positions are int32 on the device (the reference uses int64 on the
 *    host side; the Python layer converts at the boundary);
 *  - all floating point state is IEEE binary32; distances use the arithmetic of
 *    torch.cdist's matrix-product path (see dgl_tsp_reset).
 */
#ifndef DGL_AMD_H
#define DGL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGL_AMD_VERSION 100 /* major*10000 + minor*100 + patch */

#define DGL_OK 0
#define DGL_ERR_INVALID_ARGUMENT (-1)
#define DGL_ERR_UNSUPPORTED (-2)
#define DGL_ERR_LAUNCH (-3)

#define DGL_TILE_STRIDE 12 /* floats per token in a feature tile (7 used for TSP, 9 for CVRP) */
#define DGL_MAX_TOKENS 32  /* tokens per row of the 32-token kernels of dgl_policy_forward (and its default row width) */
#define DGL_MAX_WIDE_TOKENS 128 /* tokens per row dgl_policy_forward takes under DGL_ARITH_SPLIT_F16 (one workgroup per row) */

#if defined(__GNUC__)
#define DGL_API __attribute__((visibility("default")))
#else
#define DGL_API
#endif

typedef void *dgl_stream_t;

DGL_API int dgl_version(void);
/* Thread-local, valid until the next failing call on the same thread. */
DGL_API const char *dgl_last_error_string(void);

/* ------------------------------------------------------- square-root model
 * The reference's distances and running standard deviations come out of torch.cdist / torch.sqrt (TSP/TSPEnv.py:125,228,
 * 408-409; CVRP/CVRPEnv.py:262,475,733-738), and on CPU torch hands contiguous float tensors to the math library it
 * was built against.  With MKL on an AVX-512 machine - where this build's golden vectors were captured - that root is the
 * instruction VRSQRT14's estimate refined once, which is one ulp below the correctly rounded root on 0.6 % of all
 * inputs; the running standard deviations amplify such last-bit differences by N / #unvisited and the min-max
 * normalisation by another hundredfold, so the last decisions of a tour can depend on them.  No counterpart in the
 * reference's own interface: it selects which of the two the state kernels (reset, step, knn, state_data) reproduce.
 *   DGL_SQRT_MKL_AVX512  MKL's AVX-512 float sqrt bit for bit, and the reset's means in torch.mean's own summation
 *                        order (8-lane cascade sum): the device then carries the reference's state to the last bit.
 *                        The default.  VRSQRT14 is evaluated in closed form - over [1, 4) the instruction is piecewise
 *                        linear in the mantissa, 64 segments with one pair of integers each, compiled into the kernels
 *                        (csrc/common.h; solved from a recording of the instruction over all 2^24 inputs of [1, 4),
 *                        dgl_amd/data/rsqrt14_avx512.npz) - so the model needs nothing from the caller.
 *   DGL_SQRT_IEEE        correctly rounded roots (what torch without MKL, or on another CPU path, computes up to its
 *                        own library's rounding); the reset's means are binary64 sums rounded once.  Step kernels up
 *                        to 4 % faster at saturating batches, rollouts about 1 %.
 * The model also decides how tour lengths are rounded (dgl_tsp_tour_length, dgl_cvrp_route_length).
 * The setting is per device (the current one) and process-wide; set it before the reset of the states it should apply to.
 * The Python host (dgl_amd/sqrt_model.py) sets it on every device it creates an engine on (environment: DGL_SQRT_MODEL=ieee
 * or mkl_avx512).  dgl_tsp_greedy_decode implements DGL_SQRT_IEEE only and refuses to run under the other model. */
#define DGL_SQRT_IEEE 0
#define DGL_SQRT_MKL_AVX512 1
DGL_API int dgl_set_sqrt_model(int32_t model);
/* out[i] = the model's square root of x[i]: what torch.sqrt returns for x[i] under the model (the root the reference
 * takes at TSP/TSPEnv.py:409 and CVRP/CVRPEnv.py:738), for checking a model against recorded torch.sqrt values. */
DGL_API int dgl_sqrt_apply(const float *x, float *out, int64_t n, dgl_stream_t stream);

/* ------------------------------------------------------------------ TSP state
 * Compact per-row state that replaces the reference's Step_State
 * (TSP/TSPEnv.py:39-43: data [R,N,7], avg/std [R,N]) and its dense dis_matrix
 * [B,N,N] (TSP/TSPEnv.py:125,228).  Distances are recomputed from coordinates. */
typedef struct {
    int32_t B;                 /* instances */
    int32_t N;                 /* nodes per instance */
    int32_t rows_per_instance; /* pomo * beam */
    int32_t R;                 /* B * rows_per_instance */
    const float *xy;           /* [B][N][2] */
    float *avg;                /* [R][N] running mean distance to the unvisited nodes */
    float *sd;                 /* [R][N] running std of the same */
    uint64_t *visited;         /* [R][ceil(N/64)] bit j of word j/64 set = node visited */
    int32_t *tour;             /* [R][N] position -> node (the reference's selected_node_list) */
} dgl_tsp_state_t;

/* TSPEnv.reset('test') (TSP/TSPEnv.py:216-235) + load_problems' cdist (:125):
 * avg[r][i] = mean_j d(i,j), sd[r][i] = population std_j d(i,j) for every row of the
 * instance, visited cleared.  d(i,j) is the binary32 chain
 *   n_i = fl(fl(x_i^2)+fl(y_i^2)); a = fl((-2x_i) x_j); a = fma(-2y_i, y_j, a);
 *   a = fl(a + n_i); a = fl(a + n_j); d = sqrt(max(a,0))
 * i.e. what torch.cdist computes for more than 25 points.  No [N,N] matrix is stored. */
DGL_API int dgl_tsp_reset(const dgl_tsp_state_t *st, dgl_stream_t stream);

/* TSPEnv.step (TSP/TSPEnv.py:318-516) fused with TSPModel.get_knn_state,_get_node,
 * _norm_node (TSP/TSPModel.py:50-100,141-172).
 *  selected[r]: node chosen for position t-1 (t = number of visited nodes AFTER this call).
 *  Updates tour, visited, avg, sd (u = N - t unvisited).  If t < N and idx_out != NULL
 *  also builds the next decision's neighbourhood: k = min(knn, N - t) nearest unvisited
 *  nodes of `selected`, key = d(selected, j) (+100 for visited), lowest index wins ties,
 *  idx_out[r][0..k) ascending (rest -1; row stride knn), and the min-max normalised
 *  token tile tile_out[r][k+1][DGL_TILE_STRIDE]: tokens 0..k-1 = neighbours, token k =
 *  the current node, columns x, y, d(j,cur), avg, sd, sin_to_start, cos_to_start. */
DGL_API int dgl_tsp_step(const dgl_tsp_state_t *st, const int32_t *selected, int32_t t, int32_t knn,
                 int32_t *idx_out, float *tile_out, dgl_stream_t stream);

/* Beam-search step without a separate state gather (TSPEnv.perm_attr, TSP/TSPEnv.py:278-308, followed by
 * TSPEnv.step): row r of `dst` continues row (r / beam) * beam + prev_k[r] of `src` (two copies of the
 * state over the same instances); avg / sd / visited are read from that row, updated for `selected[r]` and
 * written to row r of `dst`, the neighbourhood outputs are those of dgl_tsp_step.  Tours are not copied:
 * the chosen node and the parent row of step t go to hist_node / hist_parent [N][R] (entry (t-1)*R + r) and
 * dgl_tsp_beam_backtrack rebuilds dst->tour for steps t_first..t_last (2 <= t_first) once at the end;
 * entry 0 (the start node, shared by a beam group) is kept by every linked step. */
DGL_API int dgl_tsp_step_beam(const dgl_tsp_state_t *src, const dgl_tsp_state_t *dst, const int32_t *prev_k,
                              int32_t beam, const int32_t *selected, int32_t t, int32_t knn, int32_t *idx_out,
                              float *tile_out, int32_t *hist_node, int32_t *hist_parent, dgl_stream_t stream);
DGL_API int dgl_tsp_beam_backtrack(const dgl_tsp_state_t *st, const int32_t *hist_node, const int32_t *hist_parent,
                                   int32_t t_first, int32_t t_last, dgl_stream_t stream);

/* TSPModel.get_knn_state/_get_node/_norm_node alone (TSP/TSPModel.py:50-100,141-172): the
 * neighbourhood and token tile of the current node tour[r][t-1] from the state as it stands
 * after t steps, without touching the state.  Same outputs as dgl_tsp_step. */
DGL_API int dgl_tsp_knn(const dgl_tsp_state_t *st, int32_t t, int32_t knn, int32_t *idx_out, float *tile_out,
                dgl_stream_t stream);

/* The reference's step_state.data [R][N][7] (TSP/TSPEnv.py:371,393,415,456,475),
 * materialised from the compact state after t >= 1 steps; for API parity and tests. */
DGL_API int dgl_tsp_state_data(const dgl_tsp_state_t *st, int32_t t, float *data_out, dgl_stream_t stream);

/* TSPEnv._get_travel_distance (TSP/TSPEnv.py:554-564): closed tour length per row, segments sqrt(fl(dx^2)+fl(dy^2)).
 * Under DGL_SQRT_MKL_AVX512 the float the reference's torch build returns: its roots, summed in torch.sum's order
 * (binary32 cascade sum); under DGL_SQRT_IEEE correctly rounded roots accumulated in binary64 and rounded once. */
DGL_API int dgl_tsp_tour_length(const dgl_tsp_state_t *st, float *length_out, dgl_stream_t stream);

/* ----------------------------------------------------------------- CVRP state
 * Replaces CVRP/CVRPEnv.py:170-180 (Step_State) and the dense dis_matrix.
 * Nodes 0..N, node 0 = depot; M = N + 1 points per instance. */
typedef struct {
    int32_t B;
    int32_t N;                 /* customers; M = N + 1 nodes */
    int32_t rows_per_instance;
    int32_t R;
    const float *xy;           /* [B][M][2] depot first */
    const float *demand;       /* [B][M] demand / vehicle capacity, depot 0 */
    float *avg;                /* [R][M] */
    float *sd;                 /* [R][M] */
    float *capacity;           /* [R] remaining capacity (starts at 1) */
    uint64_t *visited;         /* [R][ceil(M/64)] */
    int32_t *nodes;            /* [R][N] position -> customer */
    int8_t *flags;             /* [R][N] 1 = reached via the depot */
    int32_t demand_in_distance; /* 0: distances over (x, y) (test/valid reset, CVRP/CVRPEnv.py:475);
                                   1: over (x, y, demand) as reset('train') builds them (:519) */
} dgl_cvrp_state_t;

/* CVRPEnv.reset('test') (CVRP/CVRPEnv.py:456-482). */
DGL_API int dgl_cvrp_reset(const dgl_cvrp_state_t *st, dgl_stream_t stream);

/* CVRPEnv.step (CVRP/CVRPEnv.py:591-873) fused with CVRPModel.get_knn_state/_norm_node and
 * the tile assembly of CVRPModel.forward (CVRP/CVRPModel.py:51-187,258-309).
 *  action[r] in [0, 2(N+1)): node a, or node a-(N+1) via the depot.
 *  kd' = min(depot_knn, N - t), kl = min(knn, kd').  idx_last_out[r][0..kl),
 *  idx_depot_out[r][0..kd') ascending (row strides knn / depot_knn, rest -1);
 *  infeasible_out[r][j] (uint8, row stride knn) = 1 where neighbour j of the last set has
 *  demand > capacity + 1e-6; tile_out[r][kl+1+kd'+1][DGL_TILE_STRIDE] with columns
 *  x, y, demand(/capacity for the last set), d, avg, sd, sin, cos, group flag. */
DGL_API int dgl_cvrp_step(const dgl_cvrp_state_t *st, const int32_t *action, int32_t t, int32_t knn,
                  int32_t depot_knn, int32_t *idx_last_out, int32_t *idx_depot_out,
                  uint8_t *infeasible_out, float *tile_out, dgl_stream_t stream);

/* Beam-search step without a separate state gather (CVRPEnv.perm_attr, CVRP/CVRPEnv.py:536-581, followed by
 * CVRPEnv.step): as dgl_tsp_step_beam; capacity is read from the parent row as well.  hist_action holds the
 * action ids (node, or node + N + 1 for a move via the depot), dgl_cvrp_beam_backtrack rebuilds dst->nodes and
 * dst->flags for steps t_first..t_last (2 <= t_first); entry 0 (the first move, shared by a beam group) is
 * kept by every linked step. */
DGL_API int dgl_cvrp_step_beam(const dgl_cvrp_state_t *src, const dgl_cvrp_state_t *dst, const int32_t *prev_k,
                               int32_t beam, const int32_t *action, int32_t t, int32_t knn, int32_t depot_knn,
                               int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out, float *tile_out,
                               int32_t *hist_action, int32_t *hist_parent, dgl_stream_t stream);
DGL_API int dgl_cvrp_beam_backtrack(const dgl_cvrp_state_t *st, const int32_t *hist_action, const int32_t *hist_parent,
                                    int32_t t_first, int32_t t_last, dgl_stream_t stream);

/* CVRPModel.get_knn_state/_norm_node + tile assembly alone (CVRP/CVRPModel.py:51-187,258-309) for
 * the state after t steps.  Like the reference's get_knn_state it nudges capacities below 1e-5 by
 * +1e-7 in place (CVRP/CVRPModel.py:164-166), so call it at most once per decode step. */
DGL_API int dgl_cvrp_knn(const dgl_cvrp_state_t *st, int32_t t, int32_t knn, int32_t depot_knn,
                 int32_t *idx_last_out, int32_t *idx_depot_out, uint8_t *infeasible_out, float *tile_out,
                 dgl_stream_t stream);

/* step_state.data [R][M][8] (CVRP/CVRPEnv.py:696,718,746,782,801). */
DGL_API int dgl_cvrp_state_data(const dgl_cvrp_state_t *st, int32_t t, float *data_out, dgl_stream_t stream);

/* CVRPEnv.cal_length (CVRP/CVRPEnv.py:951-993); roots and summation per square-root model as dgl_tsp_tour_length. */
DGL_API int dgl_cvrp_route_length(const dgl_cvrp_state_t *st, float *length_out, dgl_stream_t stream);

/* --------------------------------------------------------------------- policy
 * Encoder (1 block) + decoder (3 blocks) of TSPModel / CVRPModel
 * (TSP/TSPModel.py:238-457, CVRP/CVRPModel.py:378-642), fp32 throughout. */
#define DGL_PROBLEM_TSP 0
#define DGL_PROBLEM_CVRP 1

/* Size in floats of the packed weight buffer for dgl_policy_pack. */
DGL_API int64_t dgl_policy_packed_floats(int32_t problem);

/* Raw weights in the reference's state_dict layout (nn.Linear: weight [out][in]). */
typedef struct {
    const float *wq, *wk, *wv;  /* [128][128] */
    const float *wc, *bc;       /* multi_head_combine [128][128], [128] */
    const float *w1, *b1;       /* feedForward.W1 [512][128], [512] */
    const float *w2, *b2;       /* feedForward.W2 [128][512], [128] */
} dgl_block_weights_t;

typedef struct {
    int32_t problem;            /* DGL_PROBLEM_TSP (7 features) or _CVRP (9 features) */
    const float *emb_w, *emb_b; /* encoder.embedding [128][F], [128] */
    dgl_block_weights_t block[4]; /* encoder.layers.0, decoder.layers.0..2 */
    const float *anchor_w[2], *anchor_b[2]; /* TSP: [0] = decoder.embedding_last_node;
                                   CVRP: [0] = ..._not_via_depot, [1] = ..._via_depot */
    const float *final_w, *final_b; /* decoder.Linear_final [1][128], [1] */
} dgl_policy_weights_t;

/* One-time re-layout of the weights into MFMA operand order (device -> device). */
DGL_API int dgl_policy_pack(const dgl_policy_weights_t *w, float *packed, dgl_stream_t stream);

/* TSPModel.forward / CVRPModel.forward after the tile is built (TSP/TSPModel.py:257-339,
 * CVRP/CVRPModel.py:397-523): logits, masked softmax, small-probability floor, greedy pick.
 *  tile: row r starts at tile + r*tile_row_stride, n tokens of DGL_TILE_STRIDE floats each;
 *  n <= DGL_MAX_TOKENS tokens per row; under DGL_ARITH_SPLIT_F16 also DGL_MAX_TOKENS < n <= DGL_MAX_WIDE_TOKENS
 *  (neighbourhoods of 50 + 50 nodes, CVRP/test_cvrplib.py:326-327): then out_stride >= n is required and the
 *  outputs have out_stride floats per row.  More tokens, or such rows under DGL_ARITH_F32: dgl_policy_general.
 *  TSP : tokens 0..n-2 candidates, token n-1 = current node (anchor 0).
 *  CVRP: tokens 0..kl-1 direct candidates, kl = anchor 0 (current node),
 *        kl+1..n-2 via-depot candidates, n-1 = anchor 1 (depot).
 *  infeasible (CVRP, may be NULL): [R][knn_stride] 1 = direct move masked out.
 *  Outputs (each may be NULL): logits_out [R][DGL_MAX_TOKENS] (token order as above),
 *  prob_out [R][DGL_MAX_TOKENS] = floored softmax probability per token (anchors 0),
 *  selected_out [R] = greedy action: TSP node id; CVRP action id (node, or node+N+1 via
 *  depot); ties -> lowest action id. */
/* Arithmetic of the weight x activation contractions (everything else is binary32 either way):
 *  DGL_ARITH_SPLIT_F16: every operand is split into hi = binary16(x), lo = binary16(x - hi) and the
 *    product is evaluated as hi*hi + hi*lo + lo*hi on the binary16 matrix cores with binary32
 *    accumulation (22 significant bits per operand; measured error no larger than a binary32 FMA
 *    chain of the same length).  Operands beyond +-65504 saturate: io->overflow_out reports it.
 *  DGL_ARITH_F32: binary32-input matrix instructions, bitwise equal to FMA chains. */
#define DGL_ARITH_SPLIT_F16 0
#define DGL_ARITH_F32 1

typedef struct {
    int32_t problem;
    int32_t R;
    int32_t n;            /* tokens per row at this step */
    int32_t k_last;       /* CVRP: size of the direct set; TSP: n-1 */
    int32_t N;            /* customers (CVRP) / nodes (TSP): via-depot offset is N+1 */
    int32_t idx_stride;   /* row stride of idx_last / infeasible */
    int32_t idx_dep_stride;
    int32_t tile_row_stride; /* floats between consecutive rows of `tile` (>= n*DGL_TILE_STRIDE) */
    int32_t arithmetic;   /* DGL_ARITH_* : how the dense contractions are evaluated */
    int32_t out_stride;   /* floats per row of logits_out / prob_out; 0 = DGL_MAX_TOKENS */
    const float *tile;
    const int32_t *idx_last;    /* [R][idx_stride] */
    const int32_t *idx_depot;   /* [R][idx_dep_stride] (CVRP) */
    const uint8_t *infeasible;  /* [R][idx_stride] (CVRP) */
    float *logits_out;
    float *prob_out;
    int32_t *selected_out;
    int32_t *overflow_out;      /* may be NULL.  DGL_ARITH_SPLIT_F16 stores 1 here when some operand of a dense contraction
                                   reached the binary16 range (|v| >= 65504: the result is then unreliable and the caller
                                   repeats the call with DGL_ARITH_F32); never cleared by the library */
} dgl_policy_io_t;

/* Runs the network described above for every row (TSP/TSPModel.py:174-231,
 * CVRP/CVRPModel.py:278-371, test/valid branch). */
DGL_API int dgl_policy_forward(const float *packed_weights, const dgl_policy_io_t *io, dgl_stream_t stream);

/* Dense [R][A] probabilities as the reference returns them (1e-20 where the action is not
 * a candidate; TSP/TSPModel.py:328-337, CVRP/CVRPModel.py:495-516).  A = N (TSP) or
 * 2(N+1) (CVRP). */
DGL_API int dgl_policy_scatter_probs(const dgl_policy_io_t *io, const float *prob, float *dense_out,
                             int32_t A, dgl_stream_t stream);

/* ---------------------------------------------------------------- fused greedy decode
 * The decode loop of TSPTester.beamsearch_tour_nodes_shortest for beam 1 (TSP/TSPTester.py:163-180: model(...),
 * arg-max, env.step(...), repeated) in one launch: for t = t_done .. t_end-1 the policy decision on the neighbourhood
 * the previous step built (as dgl_policy_forward, DGL_ARITH_SPLIT_F16, greedy pick to io->selected_out) and then
 * dgl_tsp_step(st, io->selected_out, t + 1, knn, io->idx_last, io->tile).  Bit-identical to that sequence of calls.
 *  st  : state after t_done >= 1 steps; io->idx_last (row stride knn) and io->tile (row stride (knn+1)*DGL_TILE_STRIDE)
 *        hold the neighbourhood of that state (dgl_tsp_step / dgl_tsp_knn wrote them) and are overwritten;
 *  io  : problem TSP, R, N as in st; n / k_last are set per step by the kernel; logits_out / prob_out are ignored;
 *        overflow_out as for dgl_policy_forward.  knn + 1 <= DGL_MAX_TOKENS.
 * Returns DGL_ERR_UNSUPPORTED when two rows' step arrays do not fit a workgroup's LDS (N above about 4000): use the
 * step-by-step calls then. */
DGL_API int dgl_tsp_greedy_decode(const dgl_tsp_state_t *st, const float *packed_weights, const dgl_policy_io_t *io,
                                  int32_t t_done, int32_t t_end, int32_t knn, dgl_stream_t stream);

/* ------------------------------------------------------- general forward / training step
 * The same network for ANY number of tokens per row (2 <= n <= 256; e.g. knn = depot_knn = 100,
 * CVRP/train.py:130-131), evaluated as a sequence of matrix and attention kernels over the flattened
 * token axis (binary32 tensors; products on the f16 matrix cores with operands split into two
 * binary16 halves, 22 significant bits, binary32 accumulation), and its backward pass: the model
 * side of CVRPTrainer._train_one_batch
 * (CVRP/CVRPTrainer.py:268-271: model(...) in mode 'train' -> probability of the teacher's
 * action, CVRP/CVRPModel.py:311-345 / TSP/TSPModel.py:206-217; loss = -log(prob).mean();
 * loss.backward()).
 *  w     : raw weights (device pointers, reference state_dict layout), as for dgl_policy_pack.
 *  grad  : NULL = inference only.  Otherwise the same struct over gradient buffers (one per
 *          tensor, same shapes, pointers cast to const float*): gradients are ADDED to them,
 *          so zero them first (model.zero_grad()).  Sums over tokens are ordered sums of partial
 *          sums: the same call on the same data gives the same bits.
 *  teacher [R] (may be NULL when grad is NULL): the teacher's action per row (TSP node id; CVRP
 *          node id, + N+1 when the move goes via the depot).  teacher_prob_out [R] (may be NULL)
 *          receives the floored probability of that action, 1e-20 when it is not a candidate
 *          token (the dense tensor the reference indexes holds 1e-20 there).
 *  grad_scale: d(loss)/d(-log p_r), i.e. 1 / (rows of the global batch) for the reference's mean.
 *  ws    : dgl_policy_general_workspace_floats(R, n) floats of scratch, 16-byte aligned
 *          (activations are kept there between the forward and the backward half of the call,
 *          with the weights repacked for the matrix cores and the partial sums of the gradients).
 *  io    : as for dgl_policy_forward (logits_out / prob_out / selected_out are written when
 *          given, with io->out_stride floats per row); io->arithmetic is ignored. */
DGL_API int64_t dgl_policy_general_workspace_floats(int32_t R, int32_t n);
DGL_API int dgl_policy_general(const dgl_policy_weights_t *w, const dgl_policy_weights_t *grad,
                               const dgl_policy_io_t *io, const int32_t *teacher, float grad_scale,
                               float *ws, float *teacher_prob_out, dgl_stream_t stream);

/* dgl_policy_general without a workspace argument, for callers that cannot allocate device memory: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the
 * reference.  The workspace is a device array inside the library of DGL_STATIC_WORKSPACE_FLOATS floats (64 MiB,
 * held on every device the library runs on until the process ends), so
 * dgl_policy_general_workspace_floats(io->R, io->n) must not exceed that (about 2000 tokens: 10 rows of 202
 * tokens, 65 rows of 32); larger calls return DGL_ERR_INVALID_ARGUMENT and belong to dgl_policy_general.
 * No hipMalloc / hipFree or other allocating call is made.  The first call on a device looks the array's
 * address up (make it outside graph capture).  All calls on one device share the array: calls that overlap in
 * time on one device race.  The workspace used is the current device's: `stream` must belong to the current
 * device (anything else returns DGL_ERR_INVALID_ARGUMENT).  Arguments and results otherwise exactly as
 * dgl_policy_general. */
#define DGL_STATIC_WORKSPACE_FLOATS (16 * 1024 * 1024)
DGL_API int dgl_policy_general_static(const dgl_policy_weights_t *w, const dgl_policy_weights_t *grad,
                                      const dgl_policy_io_t *io, const int32_t *teacher, float grad_scale,
                                      float *teacher_prob_out, dgl_stream_t stream);

/* optimizer.step() of torch.optim.Adam(lr, betas, eps, weight_decay) as the reference configures
 * it (CVRP/train.py:49-54, CVRP/CVRPTrainer.py:272) over a flat parameter vector of n floats;
 * step = 1, 2, ... ; grads are multiplied by grad_scale first (1/world_size after a sum
 * all-reduce). */
DGL_API int dgl_adam_step(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                          int32_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                          float grad_scale, dgl_stream_t stream);

/* ----------------------------------------------------------------- augmentation
 * mix_aug of the reference (utils/utils.py:790-878, used by reset('test') in aug_test mode,
 * TSP/TSPEnv.py:222-224, CVRP/CVRPEnv.py:462-467, and by reset('train')): instance b is copied
 * (kind 0), rotated (kind 1, cos_sin[b] = cos t, sin t) or reflected (kind 2, cos_sin[b] = cos 2t,
 * sin 2t), shifted by +10 and rescaled to the unit square by its larger extent.  xy, xy_out
 * [B][M][2]; the host draws the angles (the reference's torch.rand stream). */
DGL_API int dgl_augment(const float *xy, int32_t B, int32_t M, const float *cos_sin, const int32_t *kind,
                        float *xy_out, dgl_stream_t stream);

/* ---------------------------------------------------------- self-improvement labels
 * Nearest-neighbour construction the trainers start from.  TSP (TSPEnv.greedy_search,
 * TSP/TSPEnv.py:140-169): from node 0 always to the nearest unvisited node by the entries of the
 * reference's cdist matrix (row = current node; equal distances go to the lower index);
 * solution [B][N] int32, length [B] = sum of those entries around the closed tour.  The entries' roots follow the
 * square-root model (dgl_set_sqrt_model); under DGL_SQRT_MKL_AVX512 the sum is torch.sum's float (TSP/TSPEnv.py:167),
 * under DGL_SQRT_IEEE a binary64 sum rounded once.  At most 9600 nodes per instance (the workgroup's LDS).
 * CVRP (CVRPEnv.greedy_search, CVRP/CVRPEnv.py:340-409): from the depot to the nearest unvisited
 * customer; a move goes via the depot (flag 1, capacity 1 - demand) when capacity + 1e-6 < demand,
 * the first move always does.  xy [B][N+1][2], demand [B][N+1]; nodes [B][N] int32, flags [B][N]
 * int8 (the length follows from dgl_cvrp_route_length on a state whose nodes/flags point here). */
DGL_API int dgl_tsp_nearest_neighbour(const float *xy, int32_t B, int32_t N, int32_t *solution, float *length,
                                      dgl_stream_t stream);
DGL_API int dgl_cvrp_nearest_neighbour(const float *xy, const float *demand, int32_t B, int32_t N, int32_t *nodes,
                                       int8_t *flags, dgl_stream_t stream);
/* CVRPEnv.data_augmentation (CVRP/CVRPEnv.py:325-338) with the random draws as inputs: sub-route s
 * (1-based, from the s-th set flag to the next) is reversed when flip[b][s] != 0 for s = 1..S-1
 * (the reference draws torch.rand(1) > 0.5 per sub-route; the last one is never reversed), then
 * label and flags are rotated to start at sub-route (shift[b] mod S) (0-based; the reference draws
 * randint(0, S)).  flip [B][N] uint8, shift [B] int32 >= 0; not in place. */
DGL_API int dgl_cvrp_solution_augment(const int32_t *nodes, const int8_t *flags, int32_t B, int32_t N,
                                      const uint8_t *flip, const int32_t *shift, int32_t *nodes_out,
                                      int8_t *flags_out, dgl_stream_t stream);

/* ----------------------------------------------------------------------- beam
 * Beamsearch.advance (utils/beamsearch.py:52-131) on dense log-probabilities:
 * log_prob [G][beam][A]; scores [G][beam] in/out; step = current_step (>= 1; at step 1
 * only beam 0 counts).  Outputs prev_k [G][beam], action [G][beam] (int32), best beam
 * scores written back to `scores`.  Ties -> lowest flat index. */
DGL_API int dgl_beam_advance(const float *log_prob, float *scores, int32_t G, int32_t beam, int32_t A,
                     int32_t step, int32_t *prev_k_out, int32_t *action_out, dgl_stream_t stream);

/* The same decision (utils/beamsearch.py:65-99) from the compact per-token probabilities
 * (prob [R][DGL_MAX_TOKENS] as written
 * by dgl_policy_forward, R = G*beam): candidates are the tokens' actions plus the implicit
 * 1e-20 entries of every other action, exactly as the dense form.  beam <= 256; any number of
 * tokens per row (beam 100 over rows of 202 tokens is the reference's training setting,
 * CVRP/train.py:70,130-131). */
DGL_API int dgl_beam_advance_compact(const dgl_policy_io_t *io, const float *prob, float *scores, int32_t G,
                             int32_t beam, int32_t A, int32_t step, int32_t *prev_k_out,
                             int32_t *action_out, dgl_stream_t stream);

/* TSPEnv.perm_attr / CVRPEnv.perm_attr (TSP/TSPEnv.py:278-308, CVRP/CVRPEnv.py:536-581):
 * re-index the per-row state by back-pointer, src -> dst (distinct buffers). */
DGL_API int dgl_tsp_permute(const dgl_tsp_state_t *src, const dgl_tsp_state_t *dst, const int32_t *prev_k,
                    int32_t beam, int32_t t, dgl_stream_t stream);
DGL_API int dgl_cvrp_permute(const dgl_cvrp_state_t *src, const dgl_cvrp_state_t *dst, const int32_t *prev_k,
                     int32_t beam, int32_t t, dgl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DGL_AMD_H */
