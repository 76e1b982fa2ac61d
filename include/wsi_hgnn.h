/*
 * wsi_hgnn.h — C-ABI of libwsi_hgnn.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * message-passing hot path of HKU-MedAI/WSI-HGNN.
 *
 * The reference has NO native layer: its hot path is Python calling DGL's message-passing API and
 * torch.nn.Linear.  Each entry point below therefore cites the reference *call site*
 * (file:line under /root/reference) whose DGL / BLAS kernels it replaces; INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors in the shipped host code);
 *     the library never allocates, frees or retains user-visible memory; scratch is passed in;
 *   - row-major contiguous fp32 data, int32 indices; `ld*` are row strides in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it, no internal sync;
 *   - returns 0 on success, a negative errno-style code otherwise (WSI_E*); wsi_last_error() gives
 *     a thread-local message.  No C++ exception crosses the boundary;
 *   - re-entrant, NO global mutable state: the caller selects the device (hipSetDevice / torch.cuda.device) before the
 *     call; the only library-held datum is the thread-local error string.  What used to be process-wide is now passed per
 *     call: the GEMM arithmetic mode is an argument of wsi_gemm_grouped, and the side stream the attention entry points
 *     use for hub nodes lives in a caller-owned wsi_context_t (one per device, or per thread; NULL = no side stream).
 */
#ifndef WSI_HGNN_H
#define WSI_HGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSI_OK        0
#define WSI_EINVAL  (-22)   /* bad argument (shape / alignment / null pointer) */
#define WSI_ENOSYS  (-38)   /* shape not supported by the compiled kernel set   */
#define WSI_EFAULT  (-14)   /* HIP runtime reported a launch error              */
#define WSI_ENOMEM  (-12)   /* caller-provided workspace too small              */

#define WSI_ABI_VERSION 23

int         wsi_abi_version(void);
const char* wsi_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Caller-owned execution context ("one context per device for multi-GPU processes").
 * Holds one non-blocking side stream + a fork/join event pair on the device that is current at creation.  The attention
 * entry points launch the few long-running hub-node workgroups on it, forked from and joined to the caller's stream with
 * events (every effect stays ordered on `stream` from the caller's point of view), so they run UNDER the main launch.
 * A context may be shared by threads (fork..join is serialised by a mutex inside it); pass NULL to run everything in
 * order on `stream`.  Destroy only after the work that used it has completed.
 */
typedef struct wsi_context wsi_context_t;
int  wsi_context_create(wsi_context_t** out);
void wsi_context_destroy(wsi_context_t* ctx);

/* ------------------------------------------------------------------------------------------------
 * HEAT relation attention (per-relation edge softmax + weighted neighbour sum + cross-relation mean)
 *
 * Replaces, for ALL relations of a layer in one launch:
 *   models/HEATNet4.py:103      ea = e_linear(sim)                               (a4)
 *   models/HEATNet4.py:109,111  apply_edges(fn.v_dot_u('q','k','t')) * ea / sqrt_dk   (DGL SDDMM, a5)
 *   models/HEATNet4.py:113      edge_softmax(sub_graph, score)                   (5 DGL kernels, a6)
 *   models/HEATNet4.py:118-119  multi_update_all(u_mul_e -> sum, cross_reducer='mean')  (DGL SpMM, a7)
 * (identically models/HEATNet2.py:78-94).
 *
 * Graph layout (two-level CSR by destination, see wsi-hgnn_amd/graph.py):
 *   node_seg[N+1] : relation-slot segments of dst node w are [node_seg[w], node_seg[w+1])
 *   rowptr[S+1]   : in-edges of segment s are [rowptr[s], rowptr[s+1])  (edge ids in "CSR order")
 *   src[E]        : global source node id of each edge, CSR order;  sim[E]: edge scalar, CSR order
 *   order[N]      : optional (may be NULL) processing order of dst nodes
 *   num_heavy     : 0, or the number of leading entries of `order` that hold the highest in-degree nodes (kNN hubs).
 *                   Those of them with more than 32 in-edges are processed by a cooperative instantiation of the kernel
 *                   (one workgroup per node: 8 waves x 4 gathered rows in flight instead of 1 x 2, partial softmax
 *                   states merged through LDS in a fixed order), launched ahead of the main one: a launch ends when its
 *                   longest serial gather chain ends, and a hub with hundreds of in-edges IS that chain.
 *                   Deterministic; differs from the single-wave order only in fp32 rounding.  Ignored by the generic kernels.
 * Tables: q/k/v rows of node i start at q + i*ldq (etc.); D = H*d_k floats per row.
 *   t[w, :]   = (1/#segments(w)) * sum_s sum_{e in s} softmax_s(score)[e,h] * v[src[e], h, :]
 *   score[e,h]= (q[w,h,:] . k[src[e],h,:]) * (e_weight*sim[e] + e_bias) / sqrt(d_k)
 * Saved for backward: score[E,H] (raw logits) and lse[S,H] (log-sum-exp per segment and head).
 * Specialised (coalesced 16-byte lane loads, DPP head reductions): D in {128,256,512} x H in {1,2,4,8,16} with
 * 16-byte aligned tables; any other D <= 1024, H <= 16, D % H == 0 runs a generic (slower) kernel; else WSI_ENOSYS.
 */
#define WSI_ATTN_XCD_CONTIGUOUS 1   /* flags: every XCD walks one contiguous eighth of `order` (workgroup b runs on XCD b % 8): for
                                      orders with locality (wsi-hgnn_amd/graph.py::apply_locality_order, kNN graphs) the rows gathered
                                      by the waves in flight on an XCD then share its 4 MiB L2; pointless (slightly negative) for
                                      random graphs, whose default order is heaviest-first */
#define WSI_ATTN_HUB_DEGREE(d) (((d) & 0xffff) << 8)   /* flags bits 8..23: in-degree above which a leading entry of `order` goes to the
                                      cooperative hub kernel (0 = 32); must be the threshold the caller used to choose num_heavy */
int wsi_heat_attn_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                      int32_t num_nodes, int32_t D, int32_t H,
                      const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                      const int32_t* order, int32_t num_heavy, int32_t flags,
                      const float* e_weight, const float* e_bias,
                      float* t, int64_t ldt, float* score, float* lse,
                      uint32_t* t_absmax,   /* optional [N] (one part per row): receives the absmax bits (see wsi_gemm_group_t.a_absmax)
                                               of every row of t - the WSI_GEMM_FP16X3 scale of the projection that consumes t, for
                                               free while the row is in registers; NULL = not wanted */
                      wsi_context_t* ctx, void* stream);

/* Forward of a layer whose aggregate t is only read through per-segment sums (wsi_attn_pool_t): scores and softmax statistics only -
 * `score` [E, H] and `lse` [num_segs, H] exactly as wsi_heat_attn_fwd writes them; no v, no t.  Fast kernels only (D in {128, 256, 512}). */
int wsi_heat_attn_scores_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, int32_t num_nodes, int32_t D, int32_t H,
                             const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                             const int32_t* order, int32_t num_heavy, int32_t flags, const float* e_weight, const float* e_bias,
                             float* score, float* lse, wsi_context_t* ctx, void* stream);

/* ---- The same attention, blocked for the 4 MiB L2 of an XCD (csrc/heat_attn_tiled.hip; replaces the same reference lines,
 * models/HEATNet4.py:103-119).  The caller cuts the processing order into SPANS (contiguous pieces of `order` that lie inside ONE
 * graph of the batch) and deals them to 8 PARTS (part p runs on XCD p: workgroup b lands on XCD b % 8); a part walks its spans in
 * turn and, inside a span, one HEAD of every node before the next head, so that the d_k-column table slice its gathers touch
 * (rows of the graph x d_k floats) stays L2-resident.  Per-(edge, head) arrays of these entry points are HEAD-MAJOR:
 * score[H][E], lse[H][S], and the backward's a / ga / gsc / gea likewise.  d_k = D / H in {32, 64, 128}, 16-byte aligned rows;
 * anything else WSI_ENOSYS (the caller then uses wsi_heat_attn_fwd / _bwd).  The table is a host structure passed by value to the
 * kernels (no device copy; capturable). */
#define WSI_ATTN_MAX_SPANS 200
typedef struct wsi_attn_tiles {
    int32_t part_ptr[9];                 /* spans of part p: [part_ptr[p], part_ptr[p+1]); part_ptr[0] = 0, part_ptr[8] <= WSI_ATTN_MAX_SPANS */
    int32_t begin[WSI_ATTN_MAX_SPANS];   /* positions in the processing order: [begin, end) */
    int32_t end[WSI_ATTN_MAX_SPANS];
} wsi_attn_tiles_t;

/* v == NULL: scores and lse only (the forward of a layer under a sum / mean readout, as wsi_heat_attn_scores_fwd).
 * t_absmax: optional [N][H] (H parts per row). flags bits 4..7: rows in flight per lane group (measurement; 0 = default). */
int wsi_heat_attn_tiled_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            int32_t num_nodes, int32_t num_edges, int32_t num_segs, int32_t D, int32_t H,
                            const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                            const int32_t* order, const wsi_attn_tiles_t* tiles, int32_t flags,
                            const float* e_weight, const float* e_bias,
                            float* t, int64_t ldt, float* score, float* lse, uint32_t* t_absmax, void* stream);

/* Backward of wsi_heat_attn_tiled_fwd: p1 (gathers v: a, ga), p2 (gathers k: g_q, gsc, gea), p3 (CSC: g_k from q, then g_v from g_t)
 * + the float64 two-stage e_linear reduction.  All per-(edge, head) arrays head-major [H][E]; `tiles` must describe `order_dst` AND
 * `order_src` (both graph-major with the same graph boundaries; wsi-hgnn_amd/graph.py::attn_tiles).  g_t_row as in wsi_heat_attn_bwd.
 * v == NULL (a layer under a sum / mean readout): the caller has filled a and ga (wsi_heat_attn_bwd's pooled pass 1 in head-major form)
 * and takes g_v through its S x H factors: passes 2 and 3 (g_k) only; gv / g_t may be NULL.
 * g_absmax: optional [rows][3H] parts (g_q: slots [0,H), g_k: [H,2H), g_v: [2H,3H)); [rows][2H] when v == NULL. */
int wsi_heat_attn_tiled_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            int32_t num_nodes, int32_t num_edges, int32_t num_segs, int32_t D, int32_t H,
                            const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                            const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, const float* inv_rd,
                            const int32_t* order_dst, const int32_t* order_src, const wsi_attn_tiles_t* tiles, int32_t flags,
                            const float* e_weight, const float* e_bias,
                            const float* g_t, int64_t ldgt, const int32_t* g_t_row,
                            const float* score, const float* lse, float* a, float* ga, float* gsc, float* gea, float* red_ws,
                            float* gq, int64_t ldgq, float* gk, int64_t ldgk, float* gv, int64_t ldgv,
                            float* g_e, uint32_t* g_absmax, void* stream);

/* ctab[u, b, h] = sum over the out-edges e of source node u into destination node type b of exp(score[e,h] - lse[edge_seg[e],h]) / R_dst:
 * the coefficient with which v[u]_h enters the SUM of t over the (type b, graph of u) segment.  edge_seg[E]: softmax segment (row of lse) of
 * every CSR edge; CSC arrays and inv_rd as in wsi_heat_attn_bwd; row_seg / segs_per_type / n_types as in wsi_attn_pool_t. */
int wsi_heat_pool_coeff(const float* score, const float* lse, const int32_t* edge_seg,
                        const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, const float* inv_rd,
                        const int32_t* row_seg, int32_t segs_per_type, int32_t n_types, int32_t H, int32_t num_src,
                        float* ctab, void* stream);

/* gtab[u, b, h] = h[u, :] . y[type(u), b * segs_per_type + graph(u), h, :] + beta[type(u), (same segment), h]  for every node row u: what an edge
 * from u into a destination of type b adds to pass 1's ga, per head (wsi_attn_pool_t.gtab).  chunk_row / chunk_seg: the readout plan's chunk tables
 * (a chunk = at most 128 rows of ONE (type, graph) segment, wsi_segment_reduce_bwd's arguments).  D in {128, 256, 512}; n_types * H * (D + 4) * 4 <= 64 KB. */
int wsi_heat_pool_gtab(const float* h, int64_t ldh, int32_t D, int32_t H, const float* y, const float* beta,
                       const int32_t* chunk_row, const int32_t* chunk_seg, int32_t num_chunks,
                       int32_t segs_per_type, int32_t n_types, float* gtab, void* stream);

/*
 * Backward of the above (the autograd of DGL's SDDMM/SpMM/edge_softmax that loss.backward() reaches
 * from trainer/train_gnn.py:70).  Three deterministic, atomic-free passes:
 *   pass 1 (dst-major, gathers v): a[e,h] = exp(score - lse)  (written over `score`),
 *                                  ga[e,h] = (g_t[w]/R_w)[h,:] . v[src,h,:]
 *   pass 2 (dst-major, gathers k): delta, g_s = a*(ga-delta); g_q[w]; gsc[e,h] = g_s*ea/sqrt_dk;
 *                                  gea[e,h] = g_s * (q.k)/sqrt_dk
 *   pass 3 (src-major over CSC)  : g_k[u] = sum gsc*q[w],  g_v[u] = sum a*g_t[w]/R_w
 * and a fixed-shape two-stage reduction  g_e_weight = sum gea*sim,  g_e_bias = sum gea.
 * Every row of gq/gk/gv is written (zeros for nodes without edges): no memset needed.
 *   colptr[num_src+1], csc_eid[E] (CSR edge id), csc_dst[E] (global dst): CSC by source ROW of the k/v tables
 *   (num_src = N for HEAT, where src[] holds global node ids; for HGT the k/v tables hold one row per
 *   (relation, source node) — models/HGT.py:92-97 — and src[] / the CSC index those stacked rows).
 *   inv_rd[N]: 1/#segments of each node.  order_dst/order_src: optional processing orders; num_heavy as in the forward
 *   (applies to passes 1 and 2, which walk order_dst).
 *   ga, gsc, gea: caller scratch, E*H floats each.  red_ws: >= 1024 floats, 8-byte aligned (the e_linear gradients are summed in float64: 512 partial sums).  g_e[2] = {g_weight, g_bias}.
 */
/* Optional descriptor for the backward of a layer whose output is read ONLY through a sum / mean readout over S = n_types * segs_per_type
 * segments of node rows (segment = node type * segs_per_type + graph; models/HEATNet4.py:219 after :128-135).  The gradient of t then
 * has S distinct rows (g_t / g_t_row), and g_v - a [N, D] matrix of rank <= S*H - is never formed: pass 3 writes, per source node u,
 *   ctab[u, b, h] = sum over u's out-edges e into destination type b of a[e, h] / R_dst      (the coefficients of g_v[u]_h = sum_b ctab * g_t[seg]_h)
 *   r_out[u, :]   = omg[type(u)] * g_row[seg(u), :] + sum_{b, h} ctab[u, b, h] * y[type(u), b * segs_per_type + graph(u), h, :]
 * i.e. the residual term of the K|Q|V dX epilogue with g_v W_v already added (y[tau, s, h, :] = (W_v^tau rows of head h)^T g_t[s]_h, a
 * [D] vector per source type, segment and head, prepared by the caller with one small grouped GEMM).  The caller gets dW_v from
 * wsi_segment_weighted_sums(h, ctab) and runs the dX / dW projections on the K and Q chunks only.  gv may be NULL.  Fast kernels only. */
typedef struct wsi_attn_pool {
    const int32_t* row_seg;      /* [N] */
    int32_t segs_per_type, n_types;       /* n_types <= 8 */
    const float* y;              /* [n_types][S][H][D] */
    const float* g_row;          /* [S][D]: gradient of every output row of the segment */
    const float* omg;            /* [n_types]: 1 - sigmoid(skip) of the type (1 where the layer passes h through) */
    float* r_out; int64_t ldr;   /* [N][D] */
    float* ctab;                 /* [N][n_types][H] */
    int32_t ctab_ready;          /* 1: ctab was filled by the forward (wsi_heat_pool_coeff); pass 3 reads it instead of binning again */
    const float* h; int64_t ldh; /* optional: the layer input [N][D].  With it the layer never computed V at all (forward: wsi_heat_attn_scores_fwd +
                                    wsi_heat_pool_coeff + weighted sums): pass 1 gathers h[src] and takes
                                    ga[e,h] = (h[src] . y[type(src), seg(dst), h, :] + beta[type(src), seg(dst), h]) / R_dst;  v may then be NULL */
    const float* beta;           /* [n_types][S][H]: g_t[seg]_h . b_v^tau (rows of head h); required with h */
    const float* gtab;           /* optional [N][n_types][H] from wsi_heat_pool_gtab: those dot products taken once per SOURCE node; pass 1 is then one
                                    flat per-(edge, head) lookup, ga[e,h] = gtab[src, type(dst), h] / R_dst, and gathers no row at all (h / beta unused) */
    const int32_t* edge_seg;     /* with gtab: [E] softmax segment (row of lse) of every CSR edge */
    const int32_t* seg_dst;      /* with gtab: [num softmax segments] destination node of the segment */
} wsi_attn_pool_t;

int wsi_heat_attn_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                      int32_t num_nodes, int32_t num_src, int32_t num_edges, int32_t D, int32_t H,
                      const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                      const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
                      const float* inv_rd, const int32_t* order_dst, int32_t num_heavy, const int32_t* order_src, int32_t flags,
                      const float* e_weight, const float* e_bias,
                      const float* g_t, int64_t ldgt,
                      const int32_t* g_t_row, /* optional [N]: node w's gradient row is g_t[g_t_row[w]] - for a gradient with few DISTINCT
                                                 rows (the layer under a sum / mean readout gets one per (graph, node type)): the caller
                                                 passes that small table instead of N broadcast rows and pass 3's per-edge gathers of it
                                                 stay in the L2; NULL = row w of an [N, D] g_t */
                      const float* score,   /* the logits [E, H] the forward wrote (read only); NULL = they are in score_a (in place) */
                      float* score_a,       /* receives the attention probabilities exp(score - lse): E*H floats, may be the score buffer itself */
                      const float* lse,
                      float* ga, float* gsc, float* gea, float* red_ws,
                      float* gq, int64_t ldgq, float* gk, int64_t ldgk, float* gv, int64_t ldgv,
                      float* g_e,
                      uint32_t* g_absmax,   /* optional [max(N, num_src)][2] (two parts per row; zero it first): slot 0 of row r receives
                                               the absmax bits of gq[r], slot 1 those of gk[r] and gv[r] together - the row scale of a
                                               [N, 3D] g_k|g_q|g_v table that feeds one WSI_GEMM_FP16X3 dX projection; NULL = not wanted */
                      const wsi_attn_pool_t* pool,   /* optional, see above; NULL = the plain backward */
                      wsi_context_t* ctx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Grouped fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
 *
 * Replaces the torch.nn.Linear calls of the path, grouped over node types in ONE launch:
 *   models/HEATNet4.py:202      adapt_ws[t](feat)                      (a2)
 *   models/HEATNet4.py:100-102  k/v/q_linears[t](h)  (deduplicated per node type)   (a3)
 *   models/HEATNet4.py:134-135  a_linears[t](t) + sigmoid-gated skip   (a8, fused epilogue)
 *   models/HEATNet4.py:219,243-245  linears_prediction / head_*        (a10)
 * and their autograd (dX = dY W, dW = dY^T X).
 *
 * op:  WSI_GEMM_NT  C[M,N] = A[M,K] * B[N,K]^T          (forward:  Y = X W^T)
 *      WSI_GEMM_NN  C[M,N] = A[M,K] * B[K,N]            (dX = dY W, W stored [out,in] = [K,N]); the
 *                   reduction may be split over up to 3 B matrices of `b_chunk` rows each (B, B1, B2):
 *                   dX = [dK|dQ|dV] * [W_k; W_q; W_v] in one launch without concatenating the weights
 *      WSI_GEMM_TN  C[M,N] = A[K,M]^T * B[K,N]          (dW = dY^T X; reduction over rows, split-K
 *                                                        through `workspace`, deterministic)
 * epilogue flags, applied in this order to x = sum_k a*b (s = sigmoid(*gate)):
 *      WSI_EPI_BIAS        x += bias[n]                                   (NT/NN)
 *      WSI_EPI_GELU        x  = gelu(x)   (exact erf form; models/HGT.py:180)   (NT/NN)
 *      WSI_EPI_MUL_M       x *= Mm[m,n]   (the nn.Dropout of models/HEATNet4.py:135 `self.drop(self.a_linears[..](t))`:
 *                                          Mm = keep mask / (1-p), drawn by the caller)           (NT/NN)
 *      WSI_EPI_SCALE_GATE  x *= s                                         (all ops)
 *      WSI_EPI_ADD_R       x += R[m,n] * (WSI_EPI_R_1MG ? (1-s) : 1)      (NT/NN)
 *      WSI_EPI_ACCUMULATE  x += C_old[m,n]                                (all ops)
 *   WSI_EPI_GATED_SKIP = BIAS|SCALE_GATE|ADD_R|R_1MG :  C = s*(x+bias) + (1-s)*R   (HEATNet4.py:128,135)
 *   A NULL `gate` in a group means s = 1 for that group.
 * Alignment: fastest path needs lda/ldb multiples of 4 elements and 16-byte aligned A/B; anything else
 * (odd strides, K tails, edge tiles) takes a guarded scalar-load path inside the same kernel.
 */
typedef struct wsi_gemm_group {
    const float* A;
    const float* B;
    float*       C;
    const float* bias;   /* [N] or NULL */
    const float* R;      /* residual [M,N] for ADD_R or NULL */
    const float* gate;   /* device scalar for SCALE_GATE / R_1MG, or NULL (s = 1) */
    const float* B1;     /* NN only: 2nd / 3rd chunk of the reduction dimension, or NULL */
    const float* B2;
    int64_t lda, ldb, ldc, ldr;
    int32_t M, N, K;
    int32_t b_chunk;     /* NN with B1/B2: rows of the reduction per B matrix (multiple of 32); else 0 */
    const float* Mm;     /* MUL_M: [M,N] multiplier with leading dimension ldm, else NULL */
    int64_t  ldm;
    float*   colsum_out; /* TN only, may be NULL: receives sum_k A[k][m] for m in [0,M) (x sigmoid(*gate) under SCALE_GATE,
                            added to its previous contents under ACCUMULATE, exactly like C): the bias gradient
                            colsum(dY) computed from the tiles the dW GEMM stages anyway */
    /* WSI_GEMM_FP16X3 scale exchange between producers and consumers on the path (pointers may be NULL; ignored by the
       other precisions).  "absmax bits" = the IEEE bit pattern of a max of |x| (unsigned compare = magnitude compare).  A
       row's scale is given as `parts` PARTIAL maxima, row-major [rows][parts]; the consumer takes their maximum - so every
       producer writes its own slot with a plain store (device-scope atomics cost more than the pass they would save). */
    const uint32_t* a_absmax; /* NT / NN: partial absmax bits of the M rows of A over its K columns, already known to the
                                 caller (written by the kernels that produced A): the call skips its own pass over A */
    uint32_t*       c_absmax; /* NT / NN: receives partial absmax bits of the rows of C this group writes (final values,
                                 after the epilogue): slot c_absmax_first + j of row m (at c_absmax[m * c_absmax_parts + ..])
                                 for j < wsi_gemm_absmax_parts(N); groups writing other column blocks of the same rows use
                                 other slots; slots nobody writes must hold 0 */
    int32_t a_absmax_parts;   /* slots per row of a_absmax (>= 1 when a_absmax is given) */
    int32_t c_absmax_parts;   /* slots per row of c_absmax (its row pitch) */
    int32_t c_absmax_first;   /* first slot this group writes */
    int32_t reserved;
    /* WSI_EPI_DROPOUT (NT / NN): the keep mask of nn.Dropout(p) as a FUNCTION of (seed, element) instead of a tensor - see wsi_dropout_keep below */
    uint32_t drop_seed;       /* the draw: one value per (layer, forward call) */
    uint32_t drop_threshold;  /* element kept iff its 16 hash bits >= drop_threshold = round(p * 65536); 0 keeps everything */
    float    drop_scale;      /* 1 / (1 - p) */
    int32_t  drop_row0;       /* row of the masked tensor that row 0 of this group's C is (the mask belongs to the tensor, not to the grouping) */
    int32_t  drop_cols;       /* columns of the masked tensor (its row pitch in the index space of the hash) */
    int32_t  drop_col0;       /* column of the masked tensor that column 0 of this group's C is */
    const uint32_t* drop_seed_base; /* optional DEVICE word added to drop_seed (mod 2^32) when the kernel runs: a step replayed as one hipGraph
                                       advances that word between replays and so draws new masks with the launch arguments frozen */
    /* COLUMN statistics exchange for the scaled-fp16 weight gradients (WSI_GEMM_TN under FP16X3 / AUTO).  Such a launch scales every COLUMN of
       A and of B by a power of two taken from the column's absmax over the group's K rows, and - with colsum_out - sums the columns of A.  Without
       the fields below it makes its own pass over both operands; a producer on the path that writes the tensor anyway can leave the statistics
       instead.  Layout: PARTIAL tables, row-major [parts][ld]: row p holds the statistic over SOME of the rows (a producer: one part per 128-row
       tile of its group, in row order); the consumer combines the parts in order (maxima: any order; sums: part 0 first - deterministic). */
    uint32_t*       c_colmax;    /* NT / NN: receives partial absmax bits of the COLUMNS of C (final values): part m / 128, columns 0 .. N of the
                                    group at c_colmax[part * c_col_ld + n]; NULL = not wanted.  Honoured only when wsi_gemm_writes_colstats says so */
    float*          c_colsum;    /* ... and the partial column sums, same layout, or NULL */
    int64_t         c_col_ld;
    const uint32_t* a_colmax;    /* TN: partial absmax bits of A's M columns over the K rows, [a_col_parts][a_col_ld], or NULL (own pass) */
    const float*    a_colsum;    /* TN with colsum_out: partial sums of A's columns, same layout, or NULL (own pass, over A again) */
    const uint32_t* b_colmax;    /* TN: the same for B's N columns, [b_col_parts][b_col_ld] */
    int64_t         a_col_ld, b_col_ld;
    int32_t         a_col_parts, b_col_parts;
    /* WSI_GEMM_FP16X3, NT / NN: B (the weights) already in the kernel's packed form - wsi_gemm_packed_b_bytes(N, K) bytes, 16-byte aligned, filled by
       wsi_gemm_pack_b from the SAME B / B1 / B2 / ldb / N / K / b_chunk - or NULL: the call packs B itself (one more launch per call).  Weights
       change only in the optimizer step: a trainer packs them all once behind it.  Ignored by the other precisions and by TN. */
    void*           b_packed;
} wsi_gemm_group_t;

#define WSI_GEMM_NT 0
#define WSI_GEMM_NN 1
#define WSI_GEMM_TN 2

#define WSI_EPI_BIAS        1
#define WSI_EPI_ACCUMULATE  2
#define WSI_EPI_SCALE_GATE  4
#define WSI_EPI_GELU        8
#define WSI_EPI_ADD_R       16
#define WSI_EPI_R_1MG       32
#define WSI_EPI_MUL_M       64
#define WSI_EPI_DROPOUT     256  /* x *= keep(seed, row, col) ? drop_scale : 0 at the position of MUL_M (NT/NN; not together with MUL_M): the nn.Dropout of
                                    models/HEATNet4.py:135 drawn INSIDE the epilogue from a counter-based generator - no mask tensor is written, read or kept
                                    for the backward, which regenerates it (wsi_dropout_apply) */
#define WSI_EPI_BACKGROUND  128  /* TN only; a launch hint, not arithmetic: at most ONE workgroup of this launch per CU, so that a memory-bound kernel
                                    the caller runs on another stream at the same time (the attention backward) keeps half of every CU's
                                    registers and LDS.  Results are bit-identical with or without it. */
#define WSI_EPI_GATED_SKIP  (WSI_EPI_BIAS | WSI_EPI_SCALE_GATE | WSI_EPI_ADD_R | WSI_EPI_R_1MG)

#define WSI_GEMM_MAX_GROUPS 24
#define WSI_GEMM_ABSMAX_PARTS(N) (2 * (((N) + 127) / 128))   /* slots a group of N output columns writes per row of c_absmax */

/* Arithmetic of one wsi_gemm_grouped call (`precision`; the reference has one knob of this kind too: torch's
 * `torch.backends.cuda.matmul.allow_tf32`, which trainer/train_gnn.py leaves at its fp32 default):
 *   WSI_GEMM_FP32   v_mfma_f32_32x32x2_f32: products and sums in IEEE fp32.
 *   WSI_GEMM_BF16X6 every fp32 operand is split exactly into 3 bf16 terms and x*y is summed in fp32 from the 6 largest
 *                   cross products on the bf16 matrix cores; per-product relative error <= ~2^-22, i.e. results agree
 *                   with the fp32 path to fp32 rounding noise (NOT a reduced-precision mode), at up to 16/6 the rate.
 *   WSI_GEMM_FP16X3 the same idea with half the matrix work.  NT / NN: every row of A and every output column of B - TN (weight gradients,
 *                   C = A^T B: the contraction runs over the rows): every COLUMN of A and of B, the absmax taken over the group's K rows
 *                   (a_colmax / b_colmax, or a pass of the call's own) - is
 *                   scaled by a power of two so that its largest element lies in [2^14, 2^15), split into 2 fp16 terms with
 *                   round-to-nearest at both levels (|x - x0 - x1| <= 2^-23 |x|: one bit short of fp32), the second one stored
 *                   times 2^11 so that both are normal fp16 numbers for every element within 2^-28 of its row's largest (smaller
 *                   ones are flushed: absolute error <= 2^-28 of the row's largest; an element 2^-d below the row's largest keeps
 *                   22 bits for d <= 17, 39 - d beyond); x*y is summed in fp32 from 3 products on the fp16 matrix cores (the
 *                   dropped x1*y1 is <= 2^-22 |x y|), the two cross products in an accumulator of their own that is folded in
 *                   with weight 2^-11; the scales are undone exactly (v_ldexp_f32) before the epilogue.  Per product the error is
 *                   <= 2^-21 |x y| (exact fp32: none, its error is 2^-24 per accumulation step): for K >~ 100 a dot product is as
 *                   accurate as fp32's (accumulation dominates: ::test_gemm_emulated_error_vs_fp32_mfma), for short ones the bound
 *                   allows 4x fp32's element-wise error (measured at K = 16 / 64 / 128: 0.97x / 0.37x / 0.52x of fp32's maximum, mean
 *                   1.3x at K = 16: ::test_gemm_fp16x3_short_dot_products); a large element that meets an exact zero exposes the
 *                   narrower window of the small ones (bound 2^-19 per product at d = 20; measured 2^-23.1 of the remaining sum at
 *                   K = 256, fp32 2^-21.6: ::test_gemm_fp16x3_outlier_times_zero).  NT / NN need workspace for the
 *                   absmax pre-pass and the packed planes of B (wsi_gemm_workspace_bytes says how much; 16-byte aligned).
 * A per-call argument, not library state: two callers in one process may use different modes concurrently. */
#define WSI_GEMM_FP32   0
#define WSI_GEMM_BF16X6 1
#define WSI_GEMM_FP16X3 2
#define WSI_GEMM_AUTO   3   /* the faster of the two fp32-class emulations for the launch's shape: FP16X3 when the launch is large
                               enough to amortise its pre-pass - NT / NN: >= 12 GFLOP in total and every K >= 384; TN: >= 30 GFLOP, every group >= 2048 rows;
                               on a large batch (a group of >= 24576 rows) NT / NN from 5 GFLOP and K >= 256, TN from 4 GFLOP with >= 8192 rows
                               per group; TN outputs at least 192 x 192 always -,
                               else BF16X6; c_absmax is honoured either way, so scales keep flowing between
                               mixed launches */

/* The counter-based dropout mask (WSI_EPI_DROPOUT).  Element (row, col) of a [rows, cols] tensor is KEPT iff
 *     bits16(fmix32((row * ceil(cols / 2) + col / 2) * 0x9E3779B1 + seed), col & 1) >= threshold        (fmix32 = MurmurHash3's 32-bit finaliser;
 * bits16(h, 0) = h & 0xffff, bits16(h, 1) = h >> 16; all arithmetic modulo 2^32): a pure function of (seed, row, col), so the forward's epilogue and the
 * backward regenerate the same mask, and a test can replay it on the host (wsi_hgnn_amd.ops.dropout_keep_mask).  Keep probability 1 - threshold / 65536.
 * seed = the call's seed argument + *seed_base (a device word, optional; see wsi_gemm_group_t.drop_seed_base).
 * wsi_dropout_apply: out[r, c] = keep ? x[r, c] * scale : 0 for the rows [row0, row0 + rows) of the masked tensor - the backward of the dropout
 * (g_y = g_out * mask) without a stored mask; in place when out == x. */
int wsi_dropout_apply(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t rows, int32_t cols, int32_t row0, int32_t tensor_cols, int32_t col0,
                      uint32_t seed, const uint32_t* seed_base, uint32_t threshold, float scale, void* stream);

/* bytes of workspace wsi_gemm_grouped needs for this call (0 for NT/NN unless the launch runs scaled-fp16); same `precision` as the
 * call (the split-K plan depends on it). */
int64_t wsi_gemm_workspace_bytes(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups);

/* absmax bits (one part per row, see wsi_gemm_group_t.a_absmax) of the rows of X[rows, cols]: for operands that do not change from
 * step to step (the input features of a resident graph, models/HEATNet4.py:202): taken once, handed to every projection as a_absmax */
int wsi_row_absmax(const float* x, int64_t ld, int32_t rows, int32_t cols, uint32_t* out, void* stream);

/* the arithmetic a call with these arguments runs in (resolves WSI_GEMM_AUTO): for
 * callers that account matrix-core work; < 0 on a bad precision */
int32_t wsi_gemm_kernel_precision(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups);

int wsi_gemm_grouped(int32_t op, int32_t epilogue, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* The packed form of B for WSI_GEMM_FP16X3 NT / NN launches (wsi_gemm_group_t.b_packed): per output column the scale word (absmax over the
 * reduction), then the two scaled fp16 planes in MFMA fragment order.  wsi_gemm_pack_b fills b_packed of every group (fields read: B, B1, B2, ldb,
 * N, K, b_chunk) in ONE launch. */
int64_t wsi_gemm_packed_b_bytes(int32_t N, int32_t K);
int wsi_gemm_pack_b(int32_t op, const wsi_gemm_group_t* groups, int32_t ngroups, void* stream);

/* 1 when a wsi_gemm_grouped call with these arguments fills c_colmax / c_colsum of its groups (the LDS-DMA scaled-fp16 kernel of NT / NN launches
 * does; the other kernels leave the tables untouched), else 0: a caller asks before it hands the tables to a weight-gradient launch */
int32_t wsi_gemm_writes_colstats(int32_t op, int32_t precision, const wsi_gemm_group_t* groups, int32_t ngroups);

/* PARTIAL column statistics of X[rows, cols] in the layout of wsi_gemm_group_t.a_colmax / a_colsum: part p = rows [256 p, 256 p + 256), wsi_col_stats_parts(rows)
 * parts; part_max[p * part_ld + c] = bits(max |X[r, c]|), part_sum[p * part_ld + c] = sum X[r, c] (part_sum may be NULL); part_ld >= cols rounded up to 4,
 * tables 16-byte aligned.  For a weight-gradient operand whose producer leaves no statistics (the attention gradients): the caller runs it on a stream
 * of its own beside the matrix-bound projection that reads the same tensor, instead of letting the weight gradient make the pass in line. */
int32_t wsi_col_stats_parts(int32_t rows);
int wsi_col_stats(const float* x, int64_t ld, int32_t rows, int32_t cols, uint32_t* part_max, float* part_sum, int64_t part_ld, void* stream);

/* absmax bits of every COLUMN of X[rows, cols] over its rows (out[cols]; wsi_gemm_group_t.b_colmax with one part): for a weight-gradient operand
 * that does not change from step to step (the input features of a resident graph).  workspace: wsi_col_absmax_workspace_bytes(rows, cols). */
int64_t wsi_col_absmax_workspace_bytes(int32_t rows, int32_t cols);
int wsi_col_absmax(const float* x, int64_t ld, int32_t rows, int32_t cols, uint32_t* out, void* workspace, int64_t workspace_bytes, void* stream);

/* Both gradients of small Linear layers in ONE launch: dx[i]: dX = dY W (the WSI_GEMM_NN form, M <= 32 rows), dw[i]: dW = dY^T X with the bias
 * gradient in colsum_out (the WSI_GEMM_TN form, K <= 32 rows) - the backward of the classifier head behind the readout (head_2 / head_1 / head,
 * linears_prediction: models/HEATNet4.py:219,243-245; HEATNet2.py:183-190), whose levels are one row per graph.  Same group fields, same
 * fp32 fma chains in a fixed order as wsi_gemm_grouped takes for such shapes; n_dx + n_dw <= 16; epilogues: dx BIAS / ACCUMULATE / SCALE_GATE /
 * ADD_R / R_1MG, dw ACCUMULATE / SCALE_GATE.  WSI_ENOSYS when a group is not small (the caller then makes two wsi_gemm_grouped calls). */
int wsi_gemm_small_pair(const wsi_gemm_group_t* dx, int32_t n_dx, int32_t dx_epilogue,
                        const wsi_gemm_group_t* dw, int32_t n_dw, int32_t dw_epilogue, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Segmented row reduction: per-(graph, node type) readout and per-type bias gradients.
 *
 * Replaces dgl.readout.{mean,sum,max}_nodes behind
 *   pooling/avg_pooling.py:15-17, pooling/sum_pooling.py:14-16, pooling/max_pooling.py:15-17   (a9)
 * (called from models/HEATNet4.py:219).  Rows of a segment are contiguous.  Chunk tables (built once
 * per graph batch, see wsi-hgnn_amd/ops.py::ReducePlan) make the reduction two-stage and deterministic:
 *   chunk_row[C+1]  : chunk c covers rows [chunk_row[c], chunk_row[c+1]); chunks never straddle segments
 *   seg_chunk[S+1]  : chunks of segment s are [seg_chunk[s], seg_chunk[s+1])
 * op: 0 = sum, 1 = mean (empty segment -> 0), 2 = max (empty segment -> 0; argmax[S,D] receives the row).
 * partial: caller scratch, C*D floats (+ C*D int32 when op == max, placed after the floats).
 */
#define WSI_RED_SUM  0
#define WSI_RED_MEAN 1
#define WSI_RED_MAX  2

int wsi_segment_reduce_fwd(const float* x, int64_t ldx, int32_t D, int32_t op,
                           const int32_t* chunk_row, int32_t num_chunks,
                           const int32_t* seg_chunk, int32_t num_segs,
                           float* partial, float* out, int64_t ldo, int32_t* argmax, void* stream);

/* gx[r,:] = gout[seg(r),:] * (op==mean ? 1/count : 1)   (sum/mean);  max: scatter through argmax.
 * chunk_seg[C]: segment of each chunk.  For max, gx must be zero-filled by the caller. */
int wsi_segment_reduce_bwd(const float* gout, int64_t ldgo, int32_t D, int32_t op,
                           const int32_t* chunk_row, const int32_t* chunk_seg, int32_t num_chunks,
                           const int32_t* seg_chunk, int32_t num_segs,
                           const int32_t* argmax, float* gx, int64_t ldgx, void* stream);

/* out[s, j, :] = sum_{r in segment s} w[r, j] * x[r, :]   (J weighted sums of the segment's rows; x [rows, D], w [rows, J] with row
 * stride ldw, out [num_segs, J, D] contiguous).  The weight-gradient side of wsi_attn_pool_t: with x = the layer input h, w = ctab and the
 * (source type, graph) segments this gives sum_u c[u, b, h] * h[u, :], from which dW_v is an [S]-deep product.  Same chunk tables as
 * wsi_segment_reduce_fwd; two-stage, deterministic.  partial: caller scratch, num_chunks * J * D floats. */
int wsi_segment_weighted_sums(const float* x, int64_t ldx, int32_t D, const float* w, int64_t ldw, int32_t J,
                              const int32_t* chunk_row, int32_t num_chunks, const int32_t* seg_chunk, int32_t num_segs,
                              float* partial, float* out, void* stream);

/* One Adam step over `count` parameter tensors in ONE launch: the optimizer step of the reference's trainer (torch.optim.Adam(lr, weight_decay),
 * parser.py:33-38; trainer/train_gnn.py:72) with torch's arithmetic, amsgrad = False, maximize = False:
 *   g += weight_decay * p;  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` = the 1-based step count AFTER this step (torch increments before use).  p, m, v are updated in place; contiguous fp32 tensors. */
typedef struct wsi_adam_tensor { float* p; const float* g; float* m; float* v; int64_t n; } wsi_adam_tensor_t;
int wsi_adam_step(const wsi_adam_tensor_t* tensors, int32_t count, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int64_t step, void* stream);      /* (hyper-parameters in double, as torch holds them: 1 - beta2 is taken in double) */

/* Mean cross entropy of logits [B, C] against int64 labels and its gradient factor in ONE launch (torch.nn.CrossEntropyLoss() with its
 * defaults, parser.py:182-183, applied at trainer/train_gnn.py:67):  *loss = mean over the VALID rows b of (logsumexp(logits[b]) - logits[b, y_b]);
 * dlogits[b, c] = (softmax(logits[b])[c] - [c == y_b]) / #valid  (the caller multiplies it by the incoming gradient of the loss).
 * A label equal to -100 (torch's default ignore_index) is ignored as torch ignores it: zero gradient row, not counted (no valid row: loss = NaN).
 * Any other label outside [0, C) (torch: a device assert) zeroes its gradient row, turns *loss into NaN and sets *bad_label (optional) to 1 -
 * every element of dlogits is written in every case.  B * C <= 65536. */
int wsi_cross_entropy(const float* logits, const int64_t* labels, int32_t B, int32_t C, float* loss, float* dlogits, int32_t* bad_label, void* stream);

/* out[s] = sum_{r in segment s} sum_c g[r,c] * (a[r,c] - b[r,c])   — the reduction behind d(loss)/d(skip) of
 * the sigmoid-gated residual `alpha*y + (1-alpha)*h` (models/HEATNet4.py:128,135; autograd of torch.sigmoid /
 * broadcasting mul in the reference).  Same chunk tables as wsi_segment_reduce_fwd; two-stage, deterministic.
 * partial: caller scratch, num_chunks * ceil(D/256) floats. */
int wsi_segment_dot_diff(const float* g, int64_t ldg, const float* a, int64_t lda, const float* b, int64_t ldb,
                         int32_t D, const int32_t* chunk_row, int32_t num_chunks,
                         const int32_t* seg_chunk, int32_t num_segs,
                         float* partial, float* out, void* stream);

/* d(loss)/d(skip) of a HEAT layer (models/HEATNet4.py:128,135: `alpha = sigmoid(skip[n_id])`, `alpha * y + (1 - alpha) * h`) in two launches:
 *   g_skip[gate] = (1 - sigmoid(skip[gate])) * sum over the segments s with seg_gate[s] == gate of  sum_{r in s} g[r,:] . (a[r,:] - b[r,:])
 * = wsi_segment_dot_diff followed by the gate map and the sigmoid factor (the alpha factor of d sigmoid is already in g . (a - b) = g . alpha (y - h)).
 * seg_gate[s] < 0: the segment's node type has no gated output (passed through).  Every entry of g_skip [n_gates] is written.  num_segs <= 8192.
 * partial: caller scratch, num_chunks * ceil(D/256) floats. */
int wsi_gate_grad(const float* g, int64_t ldg, const float* a, int64_t lda, const float* b, int64_t ldb,
                  int32_t D, const int32_t* chunk_row, int32_t num_chunks, const int32_t* seg_chunk, int32_t num_segs,
                  const int32_t* seg_gate, const float* skip, int32_t n_gates, float* partial, float* g_skip, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The S-row algebra of a HEAT layer under a sum / mean readout (see wsi_attn_pool_t; no single reference call site: the LAST HEATLayer,
 * models/HEATNet4.py:213-214, together with pools[0] at :219).  T node types, H heads, dk = D / H, Bg graphs, S = T * Bg segments numbered
 * type * Bg + graph.  Plain fp32 sums in a fixed order.
 *
 * wsi_pool_factors: from the per-source coefficients w = ctab [rows, T * H] (column b * H + hh: destination type b, head hh) and x = the layer
 *   input h, over the SOURCE segments (tau, g) of the chunk tables (numbered tau * Bg + g):
 *     hp[b * Bg + g][hh][tau][:] = sum_u w[u, b * H + hh] * x[u, :]          ([S][H][T][D]: the T source types of a destination segment side by side)
 *     csum[tau][b * Bg + g][hh]  = sum_u w[u, b * H + hh]                    ([T][S][H])
 *     x_mean[tau * Bg + g][:]    = the mean of the segment's rows of x   (optional, [S][D]: the same pass with one more weight column of ones - the
 *                                  layer needs mean_seg(h) as well and would otherwise read h once more)
 *   partial: caller scratch, num_chunks * (T * H + 1) * (D + 1) floats.
 * wsi_pool_tmean: t_mean[s, c] = scale[s] * ( sum_tau tpart[tau, s, c] + sum_tau csum[tau, s, c / dk] * bv[tau][c] ) - the segment means of the
 *   aggregate t from the per-source-type partial products tpart [T][S][D] (hp through W_v) and the value biases; bv: HOST array of T device
 *   pointers ([D] each, NULL = no bias); scale [S] or NULL.
 * wsi_pool_bwd_prep: the head of the layer's backward from the gradient of its pooled output g_pool [S, D] (op = WSI_RED_SUM / WSI_RED_MEAN,
 *   counts [S] = rows per segment as floats, z_mean / h_mean [S, D] = segment means of the never-formed output and of the input):
 *     g_row = gradient of every output row of the segment, g_sum = ... summed over the segment's rows,
 *     g_skip[gate] = (1 - sigmoid(skip[gate])) * sum_{s: seg_gate[s] == gate} g_sum[s,:] . (z_mean[s,:] - h_mean[s,:]),
 *     omg[i] = 1 - sigmoid(skip[type_gate[i]]) (1 where type_gate[i] < 0: the layer passes that type through).   S <= 8192.
 * wsi_pool_bwd_bias: beta[tau, s, hh] = gt_seg[s, head hh] . bv[tau][head hh]  and  gbv[tau, c] = sum_s csum[tau, s, c / dk] * gt_seg[s, c]
 *   (wsi_attn_pool_t.beta and the gradient of the value biases); either output may be NULL. */
int wsi_pool_factors(const float* x, int64_t ldx, int32_t D, const float* w, int64_t ldw, int32_t T, int32_t H, int32_t Bg,
                     const int32_t* chunk_row, int32_t num_chunks, const int32_t* seg_chunk,
                     float* partial, float* hp, float* csum, float* x_mean, void* stream);
int wsi_pool_tmean(const float* tpart, int32_t T, int32_t S, int32_t D, int32_t H, const float* csum, const float* const* bv,
                   const float* scale, float* t_mean, void* stream);
int wsi_pool_bwd_prep(const float* g_pool, int32_t S, int32_t D, int32_t op, const float* counts, const float* z_mean, const float* h_mean,
                      const int32_t* seg_gate, const float* skip, int32_t n_gates, const int32_t* type_gate, int32_t T,
                      float* g_row, float* g_sum, float* g_skip, float* omg, void* stream);
int wsi_pool_bwd_bias(const float* gt_seg, int32_t T, int32_t S, int32_t D, int32_t H, const float* const* bv, const float* csum,
                      float* beta, float* gbv, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Kernel plan of a block-diagonal batch (`dgl.batch` + `g.to(device)` in front of every step of a loader-fed run: trainer/train_gnn.py:48-65):
 * every table of the batch's plan (rowptr, colptr, src, csc_eid, csc_dst, sim, the processing orders, node_seg, inv_rd) is a concatenation of
 * per-graph pieces with per-piece offsets - ONE launch over a table of segment descriptors.  desc: DEVICE array of int64, nsegs rows of 10 words
 *   [out, in1, in2, tab_off, key, add, stride, n, mode, block_start]
 * followed by the lookup tables the tab_off's index (in words from desc).  Segment s writes n elements:
 *   mode 0 (int32 out):  out[i] = (in1 ? in1[i] : 0) + add + i * stride + (tab_off >= 0 ? desc[tab_off + key + (in2 ? in2[i] : 0)] : 0)     (in1, in2: int64)
 *   mode 1 (float out):  out[i] = ((const float*)in1)[i]
 *   mode 2 (float out):  out[i] = the float whose bits are the low 32 bits of add
 * block_start = number of 1024-element blocks of the segments before it (segments with n = 0 are not listed); total_blocks = their sum. */
int wsi_plan_assemble(const int64_t* desc, int32_t nsegs, int32_t total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise kernels of the HGT / GCN siblings of the path.
 *
 * wsi_layernorm_*: torch.nn.LayerNorm(out_dim) per node type, models/HGT.py:57 (creation), :124 (use).
 *   row_param[r] (may be NULL = 0) selects the gamma/beta row (node type -> norms[n_id]); gamma/beta are [P, D].
 *   stats[n,2] receives (mean, rstd) for backward.  bwd also writes xhat_gy = gy * xhat [n,D] whose per-type
 *   column sums are d(gamma) (d(beta) = column sums of gy) — reduce them with wsi_segment_reduce_fwd.
 * wsi_gelu_*: F.gelu after the input projection, models/HGT.py:180, models/HetRGCN.py:98 (exact erf form).
 * wsi_spmm_sum: the copy_u -> sum message passing + degree norms + bias + ReLU of dgl.nn.pytorch.GraphConv
 *   (norm='both'), models/GCN.py:30-33 / models/GCN_NTPool.py:34-37:
 *     out[w] = act(oscale[w] * sum_{e in [ptr[w], ptr[w+1])} edge_w[e] * iscale[idx[e]] * x[idx[e]] + bias)
 *   (edge_w: optional per-edge weight in the order of idx - the explicit edge_weight of PyG's GCNConv / LEConv in pooling/ASAP.py:45-61,157; NULL = 1)
 *   forward: (ptr, idx) = CSR by destination; backward: CSC by source with the scales swapped.  relu_ref
 *   (may be NULL): rows of x are masked by relu_ref[idx] > 0 before being summed (ReLU backward fused in).
 *   Any D <= 1024. */
int wsi_layernorm_fwd(const float* x, int64_t ldx, int32_t n, int32_t D, float eps,
                      const float* gamma, const float* beta, const int32_t* row_param,
                      float* y, int64_t ldy, float* stats, void* stream);
int wsi_layernorm_bwd(const float* gy, int64_t ldgy, const float* x, int64_t ldx, int32_t n, int32_t D,
                      const float* gamma, const int32_t* row_param, const float* stats,
                      float* gx, int64_t ldgx, float* xhat_gy, int64_t ldp, void* stream);
int wsi_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int wsi_gelu_bwd(const float* x, const float* gy, float* gx, int64_t n, void* stream);
int wsi_spmm_sum(const float* x, int64_t ldx, int32_t n_out, int32_t D,
                 const int32_t* ptr, const int32_t* idx, const float* edge_w, const float* iscale, const float* oscale,
                 const float* bias, int32_t relu, const float* relu_ref, int64_t ldref,
                 float* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Graph-construction edge step: exact L2 kNN + per-edge Pearson correlation.
 *
 * Replaces, behind wsi-hgnn_amd/construct.py::construct_graph,
 *   construct_graph/graph_constructor.py:265-273   Hnsw(space='l2').fit(features); query(features[v], topn=radius)[1:]
 *   construct_graph/graph_constructor.py:276-282   scipy.stats.pearsonr(features[a], features[b])[0] per edge, type = corr > 0
 *
 * wsi_row_sqnorm : out[i] = sum_c x[i,c]^2.
 * wsi_knn_select : `dots` = rows [row0, row0+rows) of X X^T (ld = ldd, N columns; from wsi_gemm_grouped NT).  For every
 *                  row writes the kc (<= 32) columns j != row0+i with the smallest sqnorm[j] - 2 dots[i,j] (i.e. the
 *                  smallest |x_i - x_j|^2 up to the GEMM's rounding), ascending, ties -> smaller j; -1 pads when N-1 < kc.
 * wsi_pair_stats : for every row i and each of its kc (<= 64) candidates j: exact d2 = sum (x_i-x_j)^2 and Pearson r
 *                  from centred sums (NaN if either vector is constant, as scipy); keeps the `keep` candidates with the
 *                  smallest (d2, j), ascending, into nbr / dist2 / corr [n, keep] (caller pre-fills nbr with -1).
 */
int wsi_row_sqnorm(const float* x, int64_t ldx, int32_t n, int32_t F, float* out, void* stream);
int wsi_knn_select(const float* dots, int64_t ldd, const float* sqnorm, int32_t row0, int32_t rows, int32_t N,
                   int32_t kc, int32_t* cand, void* stream);
int wsi_pair_stats(const float* x, int64_t ldx, int32_t n, int32_t F, const int32_t* cand, int32_t kc, int32_t keep,
                   int32_t* nbr, float* dist2, float* corr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge kernels of ASAPPooling (pooling/ASAP.py:142-199).
 *
 * Edges are grouped by the node i that aggregates them: ptr[n+1], idx[E] = gathered node j of each edge in that
 * ("CSR") order; the backward passes that reduce onto j use the CSC of the same edge numbering
 * (colptr[n+1], csc_eid[E] = CSR position, csc_dst[E] = aggregating node).  One wave per node, any D <= 1024,
 * no atomics (replaces torch_scatter's atomic scatter_max / scatter_add and PyG's softmax).
 *
 * wsi_csr_gather_max_fwd : pooling/ASAP.py:158,163  X_q = scatter_max(x_pool[j], i):  out[i,c] = max_e x[idx[e],c],
 *                          arg[i,c] (int32 [n,D]) = CSR position of the first maximum; empty group -> 0 / -1.
 * wsi_csr_gather_max_bwd : gx[j,c] = sum of g_out[i,c] over the edges whose arg[i,c] is that edge.
 * wsi_asap_attend_fwd    : pooling/ASAP.py:167-179 with gat_att(cat(M_q[i], x_pool[j])) pre-split into per-node scalars
 *                          a[i] (includes the bias) and b[j]:  s_e = leaky_relu(a[i] + b[j], negative_slope);
 *                          score[e] = exp(s_e - max_i) / (sum_i exp + 1e-16)   (torch_geometric.utils.softmax);
 *                          out[i,:] = sum_e score[e] * x[idx[e],:].   score is in CSR order.
 * wsi_asap_attend_bwd    : from g_out: g_a[n], g_b[n], gx[n,D]; gpre[E] is caller scratch (receives d loss / d pre-activation).
 */
int wsi_csr_gather_max_fwd(const float* x, int64_t ldx, int32_t n, int32_t D, const int32_t* ptr, const int32_t* idx,
                           float* out, int64_t ldo, int32_t* arg, void* stream);
int wsi_csr_gather_max_bwd(const float* g_out, int64_t ldg, const int32_t* arg, int32_t n_src, int32_t D,
                           const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
                           float* gx, int64_t ldgx, void* stream);
int wsi_asap_attend_fwd(const float* a, const float* b, const float* x, int64_t ldx, int32_t n, int32_t D,
                        const int32_t* ptr, const int32_t* idx, float negative_slope,
                        float* score, float* out, int64_t ldo, void* stream);
int wsi_asap_attend_bwd(const float* a, const float* b, const float* x, int64_t ldx, int32_t n, int32_t D,
                        const int32_t* ptr, const int32_t* idx,
                        const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, float negative_slope,
                        const float* score, const float* g_out, int64_t ldg,
                        float* gpre, float* g_a, float* g_b, float* gx, int64_t ldgx, void* stream);

/* wsi_graph_topk : pooling/ASAP.py:184  perm = topk(fitness, ratio, batch)  (torch_geometric.nn.pool.topk_pool.topk).
 *                  score[n] fp32, batch[n] int64 graph id of every node (any arrangement; ids in [0, num_graphs)),
 *                  out_start[num_graphs+1] int64 = exclusive prefix of the per-graph counts k_b = ceil(ratio * n_b) (caller),
 *                  perm[out_start[num_graphs]] int64 out: graph by graph, the k_b highest-scoring nodes of graph b in descending
 *                  score order, equal scores in node order.  Rank-by-counting through LDS tiles: exact, deterministic, no sort. */
int wsi_graph_topk(const float* score, const int64_t* batch, int32_t n, int32_t num_graphs,
                   const int64_t* out_start, int32_t* rank_ws /* caller scratch, n int32 */, int64_t* perm, void* stream);

/* wsi_stas : pooling/ASAP.py:68-117  E = S^T A S of graph_connectivity (A = the edge weights, 1 when edge_weight=None - the only way the class
 *            is called), replacing torch_sparse.spspmm x2 + coalesce x4.  Same edge layout as the attention kernels: CSR by centre
 *            (rowptr/idx, score[E] in that order = the attention scores of wsi_asap_attend_fwd) and CSC by neighbour (colptr /
 *            csc_eid / csc_dst).  perm[kN] = selected centres (wsi_graph_topk), n_idx[n] = pooled index of a node or -1.
 *            fill = 0: row_count[kN] <- number of distinct columns c2 != c1 of every row (0 and *overflow = 1 for a row with more
 *                      than 1536 of them: the caller falls back to its sparse-matrix path);
 *            fill = 1: row_start[kN+1] (exclusive prefix of row_count) in, out_col/out_val[row_start[kN]] <- columns ascending and
 *                      values of every row (rows ascending = coalesced COO order); unit self loops are NOT included (:113-115).
 *            Values are sums of score*score products accumulated in 2^-40 fixed point with integer atomics: order-independent,
 *            hence bit-reproducible, and within 2^-41 per term of the exact sum. */
int wsi_stas(int32_t fill, int32_t kN, const int64_t* perm, const int32_t* n_idx,
             const int32_t* rowptr, const int32_t* idx, const float* score, const float* edge_w /* optional [E], CSR order: the values of A (NULL = 1) */,
             const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
             int32_t* row_count, const int64_t* row_start, int64_t* out_col, float* out_val, int32_t* overflow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WSI_HGNN_H */
