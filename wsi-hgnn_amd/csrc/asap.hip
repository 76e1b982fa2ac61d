// Edge kernels of ASAPPooling (pooling/ASAP.py:142-199) for gfx950.  Contracts: include/wsi_hgnn.h.
// Same shape as rowwise.hip: one 64-lane wave per graph node, lane l owns feature columns l, l+64, ... (256-byte
// coalesced wave accesses for any width <= 1024); edges arrive grouped by the node that aggregates them (CSR) and, for the
// backward passes that reduce onto the gathered node, by that node (CSC) — no atomics, fixed summation order.
//   pooling/ASAP.py:158,163  x_pool_j = x_pool[j]; X_q = scatter_max(x_pool_j, i)           -> wsi_csr_gather_max_fwd/bwd
//   pooling/ASAP.py:167-179  score = softmax_i(leaky_relu(gat_att([M_q[i], x_pool[j]])));    -> wsi_asap_attend_fwd/bwd
//                            out = scatter_add(x[j] * score, i)
#include "common.h"
#include <math.h>

namespace wsi {

constexpr int AS_BLOCK = 256;
constexpr int AS_WAVES = AS_BLOCK / 64;

__device__ __forceinline__ int as_row(int n) {
    int w = (int)blockIdx.x * AS_WAVES + (int)(threadIdx.x >> 6);
    w = __builtin_amdgcn_readfirstlane(w);
    return w < n ? w : -1;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x = fmaxf(x, __shfl_xor(x, m));
    return x;
}

// out[w,c] = max over the segment of x[idx[e],c]; arg[w,c] = CSR position e of the FIRST maximum; empty segment: 0 / -1
// (torch_scatter's scatter(..., reduce='max') leaves 0 where nothing was scattered).
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void gather_max_fwd_kernel(const float* __restrict__ x, int64_t ldx, int n, int D,
                                                                  const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                                  float* __restrict__ out, int64_t ldo, int32_t* __restrict__ arg) {
    const int w = as_row(n);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    float best[NV];
    int where[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { best[i] = -INFINITY; where[i] = -1; }
    const int e0 = ptr[w], e1 = ptr[w + 1];
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) {
                const float v = x[(int64_t)u * ldx + c];
                if (v > best[i] || where[i] < 0) { best[i] = v; where[i] = e; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) {
            out[(int64_t)w * ldo + c] = where[i] < 0 ? 0.f : best[i];
            arg[(int64_t)w * D + c] = where[i];
        }
    }
}

// gx[u,c] = sum over the out-edges (u -> w, CSR position eid) of g[w,c] where arg[w,c] == eid      (CSC by gathered node u)
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void gather_max_bwd_kernel(const float* __restrict__ g, int64_t ldg, const int32_t* __restrict__ arg,
                                                                  int n_src, int D, const int32_t* __restrict__ colptr,
                                                                  const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
                                                                  float* __restrict__ gx, int64_t ldgx) {
    const int u = as_row(n_src);
    if (u < 0) return;
    const int lane = threadIdx.x & 63;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; ++j) {
        const int eid = csc_eid[j], w = csc_dst[j];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D && arg[(int64_t)w * D + c] == eid) acc[i] += g[(int64_t)w * ldg + c];
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) gx[(int64_t)u * ldgx + c] = acc[i];
    }
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// score[e] = exp(s_e - max) / (sum exp + 1e-16), s_e = leaky_relu(a[w] + b[idx[e]])  (torch_geometric.utils.softmax);
// out[w,:] = sum_e score[e] * x[idx[e],:]
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void asap_attend_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                   const float* __restrict__ x, int64_t ldx, int n, int D,
                                                                   const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, float slope,
                                                                   float* __restrict__ score, float* __restrict__ out, int64_t ldo) {
    const int w = as_row(n);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    const int e0 = ptr[w], e1 = ptr[w + 1];
    const float aw = a[w];
    float m = -INFINITY;
    for (int e = e0 + lane; e < e1; e += 64) m = fmaxf(m, leaky(aw + b[idx[e]], slope));
    m = wave_max(m);
    float l = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) l += expf(leaky(aw + b[idx[e]], slope) - m);
    l = wave_sum(l);
    const float inv = 1.f / (l + 1e-16f);
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
        const float p = expf(leaky(aw + b[u], slope) - m) * inv;
        if (lane == 0) score[e] = p;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) acc[i] = fmaf(p, x[(int64_t)u * ldx + c], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) out[(int64_t)w * ldo + c] = acc[i];
    }
}

// dst-major backward: gp_e = g_out[w,:] . x[idx[e],:];  g_s_e = p_e (gp_e - sum_k p_k gp_k);
//                     gpre[e] = g_s_e * (pre_e > 0 ? 1 : slope);  g_a[w] = sum_e gpre[e]
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void asap_attend_bwd_dst_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                       const float* __restrict__ x, int64_t ldx, int n, int D,
                                                                       const int32_t* __restrict__ ptr, const int32_t* __restrict__ idx, float slope,
                                                                       const float* __restrict__ score, const float* __restrict__ g_out, int64_t ldg,
                                                                       float* __restrict__ gpre, float* __restrict__ g_a) {
    const int w = as_row(n);
    if (w < 0) return;
    const int lane = threadIdx.x & 63;
    const int e0 = ptr[w], e1 = ptr[w + 1];
    float go[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; go[i] = c < D ? g_out[(int64_t)w * ldg + c] : 0.f; }
    float delta = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int u = idx[e];
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int c = lane + 64 * i; if (c < D) d = fmaf(go[i], x[(int64_t)u * ldx + c], d); }
        d = wave_sum(d);
        if (lane == 0) gpre[e] = d;                        // gp_e, finished below
        delta = fmaf(score[e], d, delta);
    }
    __threadfence_block();                                 // lane 0's gp_e stores must be visible to the other lanes of this wave
    const float aw = a[w];
    float ga = 0.f;
    for (int e = e0 + lane; e < e1; e += 64) {
        const float pre = aw + b[idx[e]];
        const float gs = score[e] * (gpre[e] - delta) * (pre > 0.f ? 1.f : slope);
        gpre[e] = gs;
        ga += gs;
    }
    ga = wave_sum(ga);
    if (lane == 0) g_a[w] = ga;
}

// src-major backward (CSC by gathered node u):  g_x[u,:] = sum score[eid] * g_out[dst,:];  g_b[u] = sum gpre[eid]
template <int NV>
__global__ __launch_bounds__(AS_BLOCK) void asap_attend_bwd_src_kernel(const float* __restrict__ g_out, int64_t ldg, int n_src, int D,
                                                                       const int32_t* __restrict__ colptr, const int32_t* __restrict__ csc_eid,
                                                                       const int32_t* __restrict__ csc_dst, const float* __restrict__ score,
                                                                       const float* __restrict__ gpre, float* __restrict__ gx, int64_t ldgx,
                                                                       float* __restrict__ g_b) {
    const int u = as_row(n_src);
    if (u < 0) return;
    const int lane = threadIdx.x & 63;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    float gb = 0.f;
    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; ++j) {
        const int eid = csc_eid[j], w = csc_dst[j];
        const float p = score[eid];
        gb += gpre[eid];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = lane + 64 * i;
            if (c < D) acc[i] = fmaf(p, g_out[(int64_t)w * ldg + c], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < D) gx[(int64_t)u * ldgx + c] = acc[i];
    }
    if (lane == 0) g_b[u] = gb;
}

#define AS_NV_DISPATCH(D, CALL)                  \
    {                                            \
        const int nv_ = ((D) + 63) / 64;         \
        if (nv_ <= 1) { CALL(1); }               \
        else if (nv_ <= 2) { CALL(2); }          \
        else if (nv_ <= 4) { CALL(4); }          \
        else if (nv_ <= 8) { CALL(8); }          \
        else if (nv_ <= 16) { CALL(16); }        \
        else { set_error("feature width %d > 1024 unsupported", (D)); return WSI_ENOSYS; } \
    }

}  // namespace wsi

using namespace wsi;

// ------------------------------------------------------------------------------------------------ per-graph top-k
// pooling/ASAP.py:184  perm = topk(fitness, ratio, batch)   (torch_geometric.nn.pool.topk_pool.topk): per graph the k_b highest
// scores, graphs in order, descending score inside a graph, equal scores in node order.  Instead of a device-wide sort +
// per-graph ranking, every node counts the nodes of its own graph that precede it in that order — rank(i) = #{j : batch j ==
// batch i, s_j > s_i or (s_j == s_i and j < i)} — and, if rank < k_b, writes itself to perm[start_b + rank].  All nodes are
// staged through LDS in tiles (broadcast reads); a tile whose graph-id range cannot intersect the block's is skipped, so for
// the usual layouts (nodes grouped by graph, or by node type and then graph) the work is ~n^2 / #graphs compare-adds:
// 80k nodes in 4 graphs = 1.6e9 pairs.  The j range is cut into chunks over grid.y (a 313-workgroup grid would leave the chip
// at one wave per SIMD) and the partial ranks are summed with integer atomics: exact, deterministic, no sort.
constexpr int TK_TILE = 1024;
constexpr int TK_CHUNK = 8 * TK_TILE;      // nodes j one workgroup compares its 256 nodes i against (grid.y = ceil(n / TK_CHUNK))

// Scores are compared as integer keys that carry torch.sort's TOTAL order (descending, stable): every NaN is the largest value
// (it sorts first, NaNs among themselves in node order), -0 == +0; for everything else the usual monotone map of the IEEE bits.
// (A float comparison is false both ways on NaN: all NaN nodes would get rank 0, collide on one slot of perm and leave others
// unwritten.)
__device__ __forceinline__ int topk_key(float s) {
    if (s != s) return 0x7fffffff;
    if (s == 0.f) return 0;
    const int b = __float_as_int(s);
    return b ^ ((b >> 31) & 0x7fffffff);
}

// pass 1: rank[i] += #{j in this workgroup's chunk that precede i}; integer atomics: the sum is order-independent.
// "j precedes i" (s_j > s_i, or equal and j < i) is ONE unsigned 64-bit comparison of composite keys (order-preserving score key : ~index), and a
// tile whose live entries all belong to one graph (the usual case: nodes come grouped by graph inside a node type) needs no per-entry graph test -
// two vector instructions per pair instead of eight (round 5: 0.49 -> 0.2 ms on 80 k nodes in 4 graphs).
__device__ __forceinline__ unsigned long long topk_key64(int key, int index) {
    return ((unsigned long long)((unsigned)key ^ 0x80000000u) << 32) | (unsigned)~index;        // > 0 for every live node
}
__global__ __launch_bounds__(256) void graph_topk_rank_kernel(const float* __restrict__ score, const int64_t* __restrict__ batch, int32_t n,
                                                              int32_t* __restrict__ rank_out) {
    __shared__ __attribute__((aligned(16))) unsigned long long lk[TK_TILE];
    __shared__ __attribute__((aligned(16))) int lb[TK_TILE];
    __shared__ int red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = (int)blockIdx.x * 256 + tid;
    const bool live = i < n;
    const unsigned long long ki = live ? topk_key64(topk_key(score[i]), i) : ~0ull;          // (a dead lane counts nothing)
    const int bi = live ? (int)batch[i] : -1;
    int mn = live ? bi : 0x7fffffff, mx = live ? bi : -1;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { mn = min(mn, __shfl_xor(mn, m)); mx = max(mx, __shfl_xor(mx, m)); }
    if (lane == 0) { red[0][wv] = mn; red[1][wv] = mx; }
    __syncthreads();
    const int bmin = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
    const int bmax = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
    __syncthreads();
    int rank = 0;
    const int c0 = (int)blockIdx.y * TK_CHUNK, c1 = min(n, c0 + TK_CHUNK);
    for (int t0 = c0; t0 < c1; t0 += TK_TILE) {
        int tmn = 0x7fffffff, tmx = -1;
#pragma unroll
        for (int q = 0; q < TK_TILE / 256; ++q) {
            const int j = t0 + q * 256 + tid;
            lk[q * 256 + tid] = j < n ? topk_key64(topk_key(score[j]), j) : 0ull;     // key 0 precedes nobody
            const int b = j < n ? (int)batch[j] : -2;          // -2 never equals a graph id
            lb[q * 256 + tid] = b;
            if (j < n) { tmn = min(tmn, b); tmx = max(tmx, b); }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { tmn = min(tmn, __shfl_xor(tmn, m)); tmx = max(tmx, __shfl_xor(tmx, m)); }
        if (lane == 0) { red[0][wv] = tmn; red[1][wv] = tmx; }
        __syncthreads();
        const int lo = min(min(red[0][0], red[0][1]), min(red[0][2], red[0][3]));
        const int hi = max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3]));
        if (lo <= bmax && hi >= bmin) {                        // block-uniform: the tile may hold nodes of our graphs
            const int lim = min(TK_TILE, n - t0);
            if (lo == hi) {                                    // one graph in the tile: count without looking at graph ids
                int cnt = 0;
#pragma unroll 4
                for (int jj = 0; jj < lim; jj += 2) {          // (entries beyond n carry key 0)
                    const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(lk + jj);
                    cnt += (k2.x > ki);
                    cnt += (k2.y > ki);
                }
                rank += (lo == bi) ? cnt : 0;
            } else {
#pragma unroll 2
                for (int jj = 0; jj < lim; jj += 2) {
                    const ulonglong2 k2 = *reinterpret_cast<const ulonglong2*>(lk + jj);
                    const int2 b2 = *reinterpret_cast<const int2*>(lb + jj);
                    rank += (b2.x == bi) & (k2.x > ki);
                    rank += (b2.y == bi) & (k2.y > ki);
                }
            }
        }
        __syncthreads();
    }
    if (live && rank) atomicAdd(rank_out + i, rank);
}

// pass 2: the k_b best of every graph, in rank order
__global__ __launch_bounds__(256) void graph_topk_write_kernel(const int64_t* __restrict__ batch, int32_t n, const int32_t* __restrict__ rank,
                                                               const int64_t* __restrict__ out_start, int64_t* __restrict__ perm) {
    const int i = (int)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = (int)batch[i];
    const int64_t st = out_start[b], k = out_start[b + 1] - st;
    const int r = rank[i];
    if (r < k) perm[st + r] = i;
}

// ------------------------------------------------------------------------------------------------ E = S^T A S
// pooling/ASAP.py:68-117 (StAS + graph_connectivity; torch_sparse.spspmm x2 + coalesce x4 in the reference).
// S[j, c] = score of edge (centre perm[c] -> neighbour j), A[j1, j2] = weight of edge (j1 -> j2) (1 when edge_weight is None), so
//   E[c1, c2] = sum over paths  perm[c1] -e1-> j1 -e2-> j2 <-e3- perm[c2]  of  score[e1] * A[e2] * score[e3].
// One workgroup per pooled node c1 walks those paths straight off the CSR (by centre) / CSC (by neighbour) of the SAME edge
// list the attention kernels use — no coalesce, no intermediate A*S — and accumulates into an LDS hash table keyed by c2.
// Determinism without a sort of the candidates: every term is converted to 2^-40 fixed point and added with INTEGER atomics
// (associative: the sum does not depend on the order the lanes arrive in); which slot a key lands in does depend on that
// order, so the fill pass ranks the row's keys by counting and writes them in ascending c2 order — rows ascending, columns
// ascending inside a row, i.e. exactly the coalesced order torch_sparse returns.  Self loops (c2 == c1) are dropped here
// (ASAP.py:113); the caller appends the unit loops of :114-115.  Two passes: COUNT (unique c2 per row) and FILL.
// Two table sizes: most rows have a few dozen to a few hundred distinct columns, so each pass first runs with ST_SMALL slots (4x less
// LDS to clear / compact, more workgroups per CU) and flags the rows that do not fit (> 3/4 full: a function of the row's
// key SET, not of the insertion order); a second launch with ST_CAP slots handles only those.
constexpr int ST_CAP = 2048;               // hash slots per row in the large pass; rows with more than ST_CAP*3/4 distinct columns overflow
constexpr int ST_SMALL = 512;
constexpr int ST_EMPTY = -1;
constexpr int ST_RETRY = -1;               // row_count marker between the two count launches
constexpr double ST_SCALE = 1099511627776.0;   // 2^40

template <bool FILL, int CAP>
__global__ __launch_bounds__(256) void stas_kernel(int32_t kN, const int64_t* __restrict__ perm, const int32_t* __restrict__ n_idx,
                                                   const int32_t* __restrict__ rowptr, const int32_t* __restrict__ idx,
                                                   const float* __restrict__ score, const float* __restrict__ edge_w, const int32_t* __restrict__ colptr,
                                                   const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
                                                   int32_t* __restrict__ row_count, const int64_t* __restrict__ row_start,
                                                   int64_t* __restrict__ out_col, float* __restrict__ out_val, int32_t* __restrict__ overflow) {
    constexpr bool SMALL = CAP < ST_CAP;
    constexpr int HSHIFT = 32 - (CAP == 2048 ? 11 : 9);
    static_assert(CAP == 2048 || CAP == 512, "hash shift is written for these two sizes");
    __shared__ int keys[CAP];
    __shared__ unsigned long long vals[CAP];
    __shared__ int ck[FILL ? CAP : 1];
    __shared__ unsigned long long cv[FILL ? CAP : 1];
    __shared__ int cnt, ovf, npacked;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c1 = blockIdx.x;
    if (FILL) {                                                         // block-uniform: which launch owns this row
        const int64_t u = row_start[c1 + 1] - row_start[c1];
        if (u == 0 || SMALL != (u <= ST_SMALL * 3 / 4)) return;
    } else if (!SMALL && row_count[c1] != ST_RETRY) return;
    for (int s = tid; s < CAP; s += 256) { keys[s] = ST_EMPTY; vals[s] = 0ull; }
    if (tid == 0) { cnt = 0; ovf = 0; npacked = 0; }
    __syncthreads();
    const int i1 = (int)perm[c1];
    const int a0 = rowptr[i1], a1 = rowptr[i1 + 1];
    for (int e1 = a0 + wv; e1 < a1; e1 += 4) {                          // the 4 waves share the centre's edges
        const int j1 = idx[e1];
        const float s1 = score[e1];
        const int b0 = rowptr[j1], b1 = rowptr[j1 + 1];
        // 16 second hops x 4 of the centres that reach each of them per wave step (WSI graphs have a handful of edges per
        // node: a lane-per-third-hop loop alone left ~4 of 64 lanes busy)
        for (int eb = b0; eb < b1; eb += 16) {
            const int e2 = eb + (lane >> 2);
            int d0 = 0, d1 = 0;
            float w2 = 1.f;                                              // A[j1][j2]: the weight of the middle hop (1 when the graph has none)
            if (e2 < b1) {
                const int j2 = idx[e2];
                if (edge_w) w2 = edge_w[e2];
                d0 = colptr[j2];
                d1 = colptr[j2 + 1];
            }
            for (int e3 = d0 + (lane & 3); __any(e3 < d1); e3 += 4) {
                // a row that has outgrown this table is decided (it goes to the large launch / is reported): stop inserting -
                // probing a full table costs CAP atomics per path
                if (e3 >= d1 || *reinterpret_cast<volatile int*>(&cnt) > CAP * 3 / 4) continue;
                const int c2 = n_idx[csc_dst[e3]];
                if (c2 < 0 || c2 == c1) continue;
                const long long q = __double2ll_rn((double)s1 * (double)w2 * (double)score[csc_eid[e3]] * ST_SCALE);
                unsigned h = ((unsigned)c2 * 2654435761u) >> HSHIFT;
                int probes = 0;
                for (;;) {
                    const int k = atomicCAS(&keys[h], ST_EMPTY, c2);
                    if (k == ST_EMPTY || k == c2) {
                        atomicAdd(&vals[h], (unsigned long long)q);
                        if (k == ST_EMPTY) atomicAdd(&cnt, 1);
                        break;
                    }
                    h = (h + 1) & (CAP - 1);
                    if (++probes >= CAP) { ovf = 1; break; }
                }
            }
        }
    }
    __syncthreads();
    const int U = cnt;
    const bool over = ovf || U > CAP * 3 / 4;
    if (!FILL) {
        if (tid == 0) {
            if (!over) row_count[c1] = U;
            else if (SMALL) row_count[c1] = ST_RETRY;
            else { row_count[c1] = 0; atomicExch(overflow, 1); }
        }
        return;
    }
    if (over) return;                                                   // (cannot happen: the count pass sized the row)
    // compact the occupied slots (any order), then rank the keys by counting and write in ascending column order.  (A counter of
    // its own: re-using `cnt` here let thread 0 zero it while slower waves had not read U yet - they then wrote nothing.)
    for (int s = tid; s < CAP; s += 256) {
        if (keys[s] != ST_EMPTY) {
            const int p = atomicAdd(&npacked, 1);
            ck[p] = keys[s];
            cv[p] = vals[s];
        }
    }
    __syncthreads();
    const int64_t base = row_start[c1];
    for (int a = tid; a < U; a += 256) {
        const int key = ck[a];
        int rank = 0;
        for (int t = 0; t < U; ++t) rank += ck[t] < key;
        out_col[base + rank] = key;
        out_val[base + rank] = (float)((double)(long long)cv[a] * (1.0 / ST_SCALE));
    }
}

extern "C" int wsi_csr_gather_max_fwd(const float* x, int64_t ldx, int32_t n, int32_t D, const int32_t* ptr, const int32_t* idx,
                                      float* out, int64_t ldo, int32_t* arg, void* stream) {
    if (n < 0 || D <= 0) { set_error("csr_gather_max_fwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!x || !ptr || !out || !arg) { set_error("csr_gather_max_fwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n + AS_WAVES - 1) / AS_WAVES), b(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) hipLaunchKernelGGL((gather_max_fwd_kernel<NV>), g, b, 0, st, x, ldx, n, D, ptr, idx, out, ldo, arg)
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("csr_gather_max_fwd");
}

extern "C" int wsi_csr_gather_max_bwd(const float* g_out, int64_t ldg, const int32_t* arg, int32_t n_src, int32_t D,
                                      const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
                                      float* gx, int64_t ldgx, void* stream) {
    if (n_src < 0 || D <= 0) { set_error("csr_gather_max_bwd: bad shape"); return WSI_EINVAL; }
    if (n_src == 0) return WSI_OK;
    if (!g_out || !arg || !colptr || !gx) { set_error("csr_gather_max_bwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n_src + AS_WAVES - 1) / AS_WAVES), b(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) hipLaunchKernelGGL((gather_max_bwd_kernel<NV>), g, b, 0, st, g_out, ldg, arg, n_src, D, colptr, csc_eid, csc_dst, gx, ldgx)
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("csr_gather_max_bwd");
}

extern "C" int wsi_asap_attend_fwd(const float* a, const float* b, const float* x, int64_t ldx, int32_t n, int32_t D,
                                   const int32_t* ptr, const int32_t* idx, float negative_slope,
                                   float* score, float* out, int64_t ldo, void* stream) {
    if (n < 0 || D <= 0) { set_error("asap_attend_fwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!a || !b || !x || !ptr || !score || !out) { set_error("asap_attend_fwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n + AS_WAVES - 1) / AS_WAVES), bl(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) hipLaunchKernelGGL((asap_attend_fwd_kernel<NV>), g, bl, 0, st, a, b, x, ldx, n, D, ptr, idx, negative_slope, score, out, ldo)
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("asap_attend_fwd");
}

extern "C" int wsi_asap_attend_bwd(const float* a, const float* b, const float* x, int64_t ldx, int32_t n, int32_t D,
                                   const int32_t* ptr, const int32_t* idx,
                                   const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, float negative_slope,
                                   const float* score, const float* g_out, int64_t ldg,
                                   float* gpre, float* g_a, float* g_b, float* gx, int64_t ldgx, void* stream) {
    if (n < 0 || D <= 0) { set_error("asap_attend_bwd: bad shape"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!a || !b || !x || !ptr || !colptr || !score || !g_out || !gpre || !g_a || !g_b || !gx) { set_error("asap_attend_bwd: null pointer"); return WSI_EINVAL; }
    const dim3 g((n + AS_WAVES - 1) / AS_WAVES), bl(AS_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define CALL(NV) { hipLaunchKernelGGL((asap_attend_bwd_dst_kernel<NV>), g, bl, 0, st, a, b, x, ldx, n, D, ptr, idx, negative_slope, score, g_out, ldg, gpre, g_a); \
                   hipLaunchKernelGGL((asap_attend_bwd_src_kernel<NV>), g, bl, 0, st, g_out, ldg, n, D, colptr, csc_eid, csc_dst, score, (const float*)gpre, gx, ldgx, g_b); }
    AS_NV_DISPATCH(D, CALL)
#undef CALL
    return check_launch("asap_attend_bwd");
}

extern "C" int wsi_graph_topk(const float* score, const int64_t* batch, int32_t n, int32_t num_graphs,
                              const int64_t* out_start, int32_t* rank_ws, int64_t* perm, void* stream) {
    if (n < 0 || num_graphs < 0) { set_error("graph_topk: bad shape n=%d graphs=%d", n, num_graphs); return WSI_EINVAL; }
    if (n == 0 || num_graphs == 0) return WSI_OK;
    if (!score || !batch || !out_start || !perm || !rank_ws) { set_error("graph_topk: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(rank_ws, 0, (size_t)n * sizeof(int32_t), st) != hipSuccess) return check_launch("graph_topk(memset)");
    hipLaunchKernelGGL(graph_topk_rank_kernel, dim3((n + 255) / 256, (n + TK_CHUNK - 1) / TK_CHUNK), dim3(256), 0, st, score, batch, n, rank_ws);
    hipLaunchKernelGGL(graph_topk_write_kernel, dim3((n + 255) / 256), dim3(256), 0, st, batch, n, (const int32_t*)rank_ws, out_start, perm);
    return check_launch("graph_topk");
}

extern "C" int wsi_stas(int32_t fill, int32_t kN, const int64_t* perm, const int32_t* n_idx,
                        const int32_t* rowptr, const int32_t* idx, const float* score, const float* edge_w,
                        const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
                        int32_t* row_count, const int64_t* row_start, int64_t* out_col, float* out_val, int32_t* overflow, void* stream) {
    if (kN < 0) { set_error("stas: bad kN=%d", kN); return WSI_EINVAL; }
    if (kN == 0) return WSI_OK;
    if (!perm || !n_idx || !rowptr || !idx || !score || !colptr || !csc_eid || !csc_dst || !overflow) { set_error("stas: null pointer"); return WSI_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (!fill) {
        if (!row_count) { set_error("stas(count): null row_count"); return WSI_EINVAL; }
        hipLaunchKernelGGL((stas_kernel<false, ST_SMALL>), dim3(kN), dim3(256), 0, st, kN, perm, n_idx, rowptr, idx, score, edge_w, colptr, csc_eid, csc_dst,
                           row_count, row_start, out_col, out_val, overflow);
        hipLaunchKernelGGL((stas_kernel<false, ST_CAP>), dim3(kN), dim3(256), 0, st, kN, perm, n_idx, rowptr, idx, score, edge_w, colptr, csc_eid, csc_dst,
                           row_count, row_start, out_col, out_val, overflow);
    } else {
        if (!row_start || !out_col || !out_val) { set_error("stas(fill): null output"); return WSI_EINVAL; }
        hipLaunchKernelGGL((stas_kernel<true, ST_SMALL>), dim3(kN), dim3(256), 0, st, kN, perm, n_idx, rowptr, idx, score, edge_w, colptr, csc_eid, csc_dst,
                           row_count, row_start, out_col, out_val, overflow);
        hipLaunchKernelGGL((stas_kernel<true, ST_CAP>), dim3(kN), dim3(256), 0, st, kN, perm, n_idx, rowptr, idx, score, edge_w, colptr, csc_eid, csc_dst,
                           row_count, row_start, out_col, out_val, overflow);
    }
    return check_launch("stas");
}
