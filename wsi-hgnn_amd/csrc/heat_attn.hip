// HEAT relation attention for gfx950: fused per-relation edge softmax + neighbour aggregation (forward)
// and its three-pass atomic-free backward.  See include/wsi_hgnn.h for the contract and the reference
// call sites (models/HEATNet4.py:103-119) these kernels replace.
//
// Mapping (MI355X-first, not a translation of DGL's SDDMM/SpMM pair):
//   * one 64-lane wavefront owns one destination node (forward, backward passes 1-2) or one source
//     node (pass 3); lane l holds V = D/64 contiguous floats of every 2 KB feature row, so a row
//     gather is one fully coalesced wave-wide access and head h lives in LPH = 64/H adjacent lanes;
//   * per-head dot products reduce over those LPH lanes with DPP (quad_perm / row_mirror), no LDS;
//   * all indices (segment pointers, edge ids, src ids, sim) are wave-uniform -> scalar (SMEM) loads,
//     leaving the vector-memory pipe to the row gathers; U edges are kept in flight per wave;
//   * rows are consumed exactly once per wave, straight from the vector loads into registers: an LDS
//     stage would be a pure round trip here (no cross-wave reuse), so none is used;
//   * the softmax is online (running max / sum) and the cross-relation mean is folded into the same
//     wave by walking all relation slots of the node, so `m` (per-relation messages) and the
//     per-edge attention tensor DGL materialises never exist; raw logits + one LSE per segment are
//     what backward needs.
//   * nodes are visited heaviest-first (`order`) so kNN hub nodes do not form the tail of the launch.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <new>

namespace wsi {

struct AttnGraph {
    const int32_t* node_seg;
    const int32_t* rowptr;
    const int32_t* src;
    const float* sim;
    const int32_t* order;
    int32_t num_nodes;
    // Hub split (fast kernels): the first `heavy_n` entries of `order` are the highest in-degree nodes.  pass 0 = one
    // launch handles everything; pass 1 = "light" launch, skips entries < heavy_n whose in-degree exceeds kHeavyDegree;
    // pass 2 = "heavy" launch over those entries only: one WORKGROUP per node, its 8 waves each gathering 4 rows per round
    // (32 edges in flight per node instead of 2) and merging their partial softmax states through LDS: a launch ends when
    // its longest serial gather chain ends, and a kNN hub with hundreds of in-edges at 2 rows per round IS that chain.
    int32_t heavy_n;
    int32_t pass;
    int32_t heavy_deg; // in-degree above which a leading entry of `order` goes to the cooperative (hub) kernel
    int32_t xcd;       // 1: walk `order` XCD-contiguously (workgroup b runs on XCD b % 8; remapped so that each XCD takes one
                       // contiguous eighth of the order: with a locality order, the rows in flight on an XCD share its L2)
    uint32_t* absmax;  // optional: receives the absmax bits of the row each node's wave(s) write (t forward, g_q backward; see the C API)
    const float* score_in; // backward: the logits the forward saved; pass 1 reads them and writes the probabilities to its score_a argument (the same buffer: in place)
    const int32_t* gt_row; // backward, optional: node w reads row gt_row[w] of g_t (a gradient with few distinct rows, e.g. under a mean
                           // readout: the gathers of pass 3 then hit a table that stays in the L2); null = row w
};

// same bijective remap as the GEMMs' tile order (gemm_common.h)
__device__ __forceinline__ int attn_xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int kHeavyDegree = 32;
constexpr int kHeavyUnroll = 4;

struct AttnTables {
    const float* q; int64_t ldq;
    const float* k; int64_t ldk;
    const float* v; int64_t ldv;
};

// Backward of a layer whose output is only read through a sum / mean readout over S = (node type, graph) segments (see the C API,
// wsi_attn_pool_t): pass 3 then never forms g_v.  Segment numbering: seg = type * segs_per_type + graph.
struct AttnPool {
    const int32_t* row_seg;     // [N] segment of every node row
    int32_t segs_per_type;      // graphs in the batch
    int32_t n_types;            // <= kPoolTypes
    int32_t num_segs;           // n_types * segs_per_type
    const float* y;             // [n_types (source type)][num_segs][H][D]: (Wv_h)^T g_t[seg]_h
    const float* g_row;         // [num_segs][D]: gradient of every output row of the segment
    const float* omg;           // [n_types]: 1 - sigmoid(skip) of the node type (1 for a passthrough type)
    float* r_out; int64_t ldr;  // [N][D]: omg * g_row[seg(u)] + sum_{bin,h} c[u,bin,h] * y[type(u), seg(bin), h, :]
    float* ctab;                // [N][n_types][H]: c[u, bin, h] = sum over u's out-edges into dst type `bin` of a[e,h] / R_dst
    int32_t ctab_ready;         // 1: ctab was filled by the forward (wsi_heat_pool_coeff): pass 3 reads it instead of binning again
    const float* h; int64_t ldh;   // optional: the layer input; with it pass 1 gathers h[src] and never touches v:
    const float* beta;             //   ga[e,h] = (h[src] . y[type(src), seg(dst), h, :] + beta[type(src), seg(dst), h]) / R_dst,  beta = g_t[seg]_h . b_v_h
    const float* gtab;             // optional [N][n_types][H]: those dot products taken once per SOURCE node (wsi_heat_pool_gtab): pass 1 is then a
    const int32_t* edge_seg;       //   flat per-(edge, head) lookup - edge_seg[E]: softmax segment of every CSR edge, seg_dst[num softmax segments]:
    const int32_t* seg_dst;        //   destination node of every softmax segment
};
constexpr int kPoolTypes = 8;

constexpr int kBlock = 512;          // 8 waves per workgroup (A/B on one MI355X: 128 / 256 / 512 / 1024 threads -> hub batch 3.17 / 2.66 / 2.48 / 3.41 ms, uniform batch unchanged)
constexpr int kWavesPerBlock = kBlock / 64;

// COOP: the whole workgroup (kWavesPerBlock waves) works on ONE node, wave `part` taking every kWavesPerBlock-th group of U
// edges of each segment.
template <bool COOP = false>
__device__ __forceinline__ int wave_uniform_node(const AttnGraph& g, int& lane) {
    lane = threadIdx.x & 63;
    const int blk = (g.xcd && !COOP) ? attn_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int wave = COOP ? blk : blk * kWavesPerBlock + (int)(threadIdx.x >> 6);
    wave = __builtin_amdgcn_readfirstlane(wave);
    if (wave >= g.num_nodes) return -1;
    int w = g.order ? g.order[wave] : wave;
    w = __builtin_amdgcn_readfirstlane(w);
    if (g.pass != 0) {
        bool heavy = false;
        if (wave < g.heavy_n) heavy = (g.rowptr[g.node_seg[w + 1]] - g.rowptr[g.node_seg[w]]) > g.heavy_deg;
        if (heavy != (g.pass == 2)) return -1;
    }
    return w;
}

// bits of max |r[i]| over the wave (all lanes get it)
template <int NV>
__device__ __forceinline__ uint32_t wave_absmax_bits(const float (&r)[NV]) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) m = fmaxf(m, fabsf(r[i]));
    uint32_t b = __float_as_uint(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, o, 64));
    return b;
}

// ------------------------------------------------------------------------------------------ forward
// NOV: scores, online softmax statistics and nothing else (no v gather, no t) - the forward of a layer whose aggregate is only read
// through per-segment sums (wsi_heat_attn_scores_fwd)
template <int V, int LPH, int U, bool COOP = false, bool NOV = false>
__global__ __launch_bounds__(kBlock) void heat_attn_fwd_kernel(
    AttnTables tb, AttnGraph g, const float* __restrict__ e_weight, const float* __restrict__ e_bias,
    float inv_sqrt_dk, float* __restrict__ t, int64_t ldt, float* __restrict__ score, float* __restrict__ lse) {
    constexpr int H = 64 / LPH;
    __shared__ float sm_ml[COOP ? 2 * kWavesPerBlock * 64 : 1];
    __shared__ float sm_acc[COOP ? kWavesPerBlock * 64 * V : 1];
    int lane;
    const int w = wave_uniform_node<COOP>(g, lane);
    if (w < 0) return;
    const int part = COOP ? (int)(threadIdx.x >> 6) : 0;
    constexpr int ESTEP = COOP ? kWavesPerBlock * U : U;
    const int head = lane / LPH;
    const bool leader = (lane % LPH) == 0;
    const int col = lane * V;

    float q[V];
    load_vec<V>(q, tb.q + (int64_t)w * tb.ldq + col);
    const float we = *e_weight, be = *e_bias;

    float tacc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) tacc[i] = 0.f;

    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    for (int s = s0; s < s1; ++s) {
        const int e0 = g.rowptr[s], e1 = g.rowptr[s + 1];
        if (e0 == e1) continue;
        float m = -INFINITY, l = 0.f;
        float acc[V];
#pragma unroll
        for (int i = 0; i < V; ++i) acc[i] = 0.f;

        for (int e = e0 + part * U; e < e1; e += ESTEP) {
            float kk[U][V], vv[U][V];
            float c[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (e + j < e1) {
                    const int u = g.src[e + j];
                    c[j] = (we * g.sim[e + j] + be) * inv_sqrt_dk;
                    load_vec<V>(kk[j], tb.k + (int64_t)u * tb.ldk + col);
                    if constexpr (!NOV) load_vec<V>(vv[j], tb.v + (int64_t)u * tb.ldv + col);
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (e + j < e1) {
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < V; ++i) d = fmaf(q[i], kk[j][i], d);
                    d = group_sum<LPH>(d);
                    const float sc = d * c[j];
                    if (leader) score[(int64_t)(e + j) * H + head] = sc;
                    const float mn = fmaxf(m, sc);
                    const float scale = expf(m - mn);
                    const float pr = expf(sc - mn);
                    l = l * scale + pr;
                    if constexpr (!NOV) {
#pragma unroll
                        for (int i = 0; i < V; ++i) acc[i] = fmaf(pr, vv[j][i], acc[i] * scale);
                    }
                    m = mn;
                }
            }
        }
        if constexpr (COOP) {       // merge the waves' online-softmax states (fixed order: deterministic)
            sm_ml[part * 64 + lane] = m;
            sm_ml[(kWavesPerBlock + part) * 64 + lane] = l;
            if constexpr (!NOV) {
#pragma unroll
                for (int i = 0; i < V; ++i) sm_acc[(part * V + i) * 64 + lane] = acc[i];
            }
            __syncthreads();
            float M = -INFINITY;
#pragma unroll
            for (int pp = 0; pp < kWavesPerBlock; ++pp) M = fmaxf(M, sm_ml[pp * 64 + lane]);
            l = 0.f;
#pragma unroll
            for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
            for (int pp = 0; pp < kWavesPerBlock; ++pp) {
                const float f = expf(sm_ml[pp * 64 + lane] - M);          // exp(-inf) = 0 for a wave that saw no edge
                l = fmaf(sm_ml[(kWavesPerBlock + pp) * 64 + lane], f, l);
                if constexpr (!NOV) {
#pragma unroll
                    for (int i = 0; i < V; ++i) acc[i] = fmaf(sm_acc[(pp * V + i) * 64 + lane], f, acc[i]);
                }
            }
            m = M;
            __syncthreads();
        }
        const float inv_l = 1.f / l;
#pragma unroll
        for (int i = 0; i < V; ++i) tacc[i] = fmaf(acc[i], inv_l, tacc[i]);
        if (leader && part == 0) lse[(int64_t)s * H + head] = m + logf(l);
    }
    if constexpr (!NOV) {
        const float inv_r = (s1 > s0) ? 1.f / (float)(s1 - s0) : 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) tacc[i] *= inv_r;
        if (part == 0) store_vec<V>(t + (int64_t)w * ldt + col, tacc);
        if (g.absmax) {                                  // one writer per node: a plain store
            const uint32_t b = wave_absmax_bits<V>(tacc);
            if (part == 0 && lane == 0) g.absmax[w] = b;
        }
    }
}

// every lane holds one partial per head (p[h]: its V columns' share of a D-long dot product taken once per head); returns, in the lanes of
// head h, the wave-wide sum of p[h]: log2(H) halving exchanges between the head groups, then the sum inside the group - H - 1 + log2(LPH)
// shuffles instead of H full wave reductions
template <int LPH>
__device__ __forceinline__ float heads_reduce(float (&p)[64 / LPH], int lane) {
    constexpr int H = 64 / LPH;
    if constexpr (H >= 2) {
#pragma unroll
        for (int w = 32, n = H; w >= LPH; w >>= 1, n >>= 1) {       // n: heads this lane still carries
            const bool upper = (lane & w) != 0;
#pragma unroll
            for (int k = 0; k < H / 2; ++k) {
                if (k < n / 2) {
                    const float send = upper ? p[k] : p[k + n / 2];
                    const float recv = __shfl_xor(send, w, 64);
                    p[k] = (upper ? p[k + n / 2] : p[k]) + recv;
                }
            }
        }
    }
    return group_sum<LPH>(p[0]);
}

// ------------------------------------------------------------------------------------------ backward pass 1
// dst-major, gathers v:  a = exp(score - lse) (in place),  ga[e,h] = (g_t[w]/R_w)[h,:] . v[src,h,:]
template <int V, int LPH, int U, bool COOP = false, bool POOL = false>
__global__ __launch_bounds__(kBlock) void heat_attn_bwd_p1_kernel(
    AttnTables tb, AttnGraph g, const float* __restrict__ g_t, int64_t ldgt,
    float* score_a, const float* __restrict__ lse, float* __restrict__ ga, AttnPool pool) {
    constexpr int H = 64 / LPH;
    int lane;
    const int w = wave_uniform_node<COOP>(g, lane);
    if (w < 0) return;
    const int part = COOP ? (int)(threadIdx.x >> 6) : 0;
    constexpr int ESTEP = COOP ? kWavesPerBlock * U : U;
    const int head = lane / LPH;
    const bool leader = (lane % LPH) == 0;
    const int col = lane * V;
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    if (s1 == s0) return;
    const float inv_r = 1.f / (float)(s1 - s0);
    float gm[V];
    if constexpr (!POOL) {
        load_vec<V>(gm, g_t + (int64_t)(g.gt_row ? g.gt_row[w] : w) * ldgt + col);
#pragma unroll
        for (int i = 0; i < V; ++i) gm[i] *= inv_r;
    }
    const int sw = POOL ? pool.row_seg[w] : 0;          // POOL: the readout segment of the destination

    for (int s = s0; s < s1; ++s) {
        const int e0 = g.rowptr[s], e1 = g.rowptr[s + 1];
        if (e0 == e1) continue;
        const float ls = lse[(int64_t)s * H + head];
        // POOL: v = h W_v^T + b_v is not stored; the sources of one relation share a node type tau, so
        //   ga[e,h] = (g_t[seg]_h . v[src]_h) / R = (h[src] . y[tau, seg, h, :] + beta[tau, seg, h]) / R
        float yv[POOL ? H : 1][V];
        float beta = 0.f;
        if constexpr (POOL) {
            const int tau = pool.row_seg[g.src[e0]] / pool.segs_per_type;
            const int64_t yo = ((int64_t)tau * pool.num_segs + sw) * H;
#pragma unroll
            for (int h = 0; h < H; ++h) load_vec<V>(yv[h], pool.y + (yo + h) * (V * 64) + col);
            beta = pool.beta[yo + head];
        }
        for (int e = e0 + part * U; e < e1; e += ESTEP) {
            float vv[U][V];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (e + j < e1) {
                    const int u = g.src[e + j];
                    if constexpr (POOL) load_vec<V>(vv[j], pool.h + (int64_t)u * pool.ldh + col);
                    else load_vec<V>(vv[j], tb.v + (int64_t)u * tb.ldv + col);
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (e + j < e1) {
                    float d;
                    if constexpr (POOL) {
                        float p[H];
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            p[h] = 0.f;
#pragma unroll
                            for (int i = 0; i < V; ++i) p[h] = fmaf(yv[h][i], vv[j][i], p[h]);
                        }
                        d = (heads_reduce<LPH>(p, lane) + beta) * inv_r;
                    } else {
                        d = 0.f;
#pragma unroll
                        for (int i = 0; i < V; ++i) d = fmaf(gm[i], vv[j][i], d);
                        d = group_sum<LPH>(d);
                    }
                    if (leader) {
                        const int64_t o = (int64_t)(e + j) * H + head;
                        score_a[o] = expf(g.score_in[o] - ls);
                        ga[o] = d;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ pooled pass 1 without row gathers
// gtab[u, b, h] = h[u] . y[type(u), (b, graph(u)), h, :] + beta[type(u), (b, graph(u)), h]: what an edge from source u into a destination of type b
// contributes to ga, per head - T*H dot products per SOURCE node instead of H per EDGE.  Block = one chunk (<= 128 rows of one (type, graph) segment):
// the segment's J = T*H vectors go to the LDS once, every wave takes rows round-robin with its lanes along the columns.
// A [rows, D] x [D, J] product per (type, graph) segment, J = T*H <= 32.  Block = one chunk (<= 128 rows of one segment): the segment's J vectors
// go to the LDS once (rows padded by 4 floats: the four j-groups of a lane quad hit different banks); thread = (pair of rows, group of JP
// j's): 2 x JP dot products in registers, each fetched h value feeding JP of them and each y value two.  A wave covers 32 rows per pass
// (the first version gave every thread one row and one j: the 16 lanes of a row all fetched the same 16 bytes, and the address unit spends
// its cycles per LANE - 82 us for a pass HBM serves in 35).
constexpr int kGtabPad = 4;
template <int D, int JP>
__global__ __launch_bounds__(256) void heat_pool_gtab_kernel(
    const float* __restrict__ h, int64_t ldh, const float* __restrict__ y, const float* __restrict__ beta,
    const int32_t* __restrict__ chunk_row, const int32_t* __restrict__ chunk_seg, int32_t segs_per_type, int32_t n_types, int32_t H,
    float* __restrict__ gtab) {
    extern __shared__ float ylds[];                       // [J][D + kGtabPad]
    const int c = blockIdx.x;
    const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
    const int sig = chunk_seg[c];
    const int tau = sig / segs_per_type, gb = sig - tau * segs_per_type;
    const int S = n_types * segs_per_type, J = n_types * H;
    for (int idx = threadIdx.x; idx < J * (D / 4); idx += 256) {
        const int j = idx / (D / 4), q4 = idx - j * (D / 4);
        const int b = j / H, hh = j - b * H;
        const float4 v = *reinterpret_cast<const float4*>(y + (((int64_t)tau * S + (b * segs_per_type + gb)) * H + hh) * D + q4 * 4);
        *reinterpret_cast<float4*>(ylds + j * (D + kGtabPad) + q4 * 4) = v;
    }
    __syncthreads();
    const int rg = threadIdx.x >> 2, jg = threadIdx.x & 3;
    const int j0 = jg * JP;
    for (int rb = r0; rb < r1; rb += 128) {
        const int ra = rb + 2 * rg, rbb = ra + 1;
        if (ra >= r1) break;
        const bool two = rbb < r1;
        const float* ha = h + (int64_t)ra * ldh;
        const float* hb = h + (int64_t)(two ? rbb : ra) * ldh;
        float acc_a[JP], acc_b[JP];
#pragma unroll
        for (int q = 0; q < JP; ++q) { acc_a[q] = 0.f; acc_b[q] = 0.f; }
#pragma unroll 4
        for (int k = 0; k < D; k += 4) {
            const float4 va = *reinterpret_cast<const float4*>(ha + k);
            const float4 vb = *reinterpret_cast<const float4*>(hb + k);
#pragma unroll
            for (int q = 0; q < JP; ++q) {
                const int j = min(j0 + q, J - 1);
                const float4 yv = *reinterpret_cast<const float4*>(ylds + j * (D + kGtabPad) + k);
                acc_a[q] = fmaf(va.x, yv.x, fmaf(va.y, yv.y, fmaf(va.z, yv.z, fmaf(va.w, yv.w, acc_a[q]))));
                acc_b[q] = fmaf(vb.x, yv.x, fmaf(vb.y, yv.y, fmaf(vb.z, yv.z, fmaf(vb.w, yv.w, acc_b[q]))));
            }
        }
#pragma unroll
        for (int q = 0; q < JP; ++q) {
            const int j = j0 + q;
            if (j < J) {
                const int b = j / H, hh = j - b * H;
                const float bj = beta[((int64_t)tau * S + (b * segs_per_type + gb)) * H + hh];
                gtab[(int64_t)ra * J + j] = acc_a[q] + bj;
                if (two) gtab[(int64_t)rbb * J + j] = acc_b[q] + bj;
            }
        }
    }
}

// thread = (CSR edge e, head h):  a = exp(score - lse) in place,  ga[e,h] = gtab[src, type(dst), h] / R_dst
__global__ __launch_bounds__(256) void heat_attn_bwd_p1_flat_kernel(
    const int32_t* __restrict__ src, const int32_t* __restrict__ edge_seg, const int32_t* __restrict__ seg_dst,
    const int32_t* __restrict__ row_seg, int32_t segs_per_type, int32_t n_types, const float* __restrict__ inv_rd,
    const float* __restrict__ gtab, const float* __restrict__ lse, int64_t EH, int32_t H, const float* score_in, float* score_a, float* __restrict__ ga) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= EH) return;
    const int64_t e = o / H;
    const int hh = (int)(o - e * H);
    const int s = edge_seg[e];
    const int w = seg_dst[s];
    const int b = row_seg[w] / segs_per_type;
    score_a[o] = expf(score_in[o] - lse[(int64_t)s * H + hh]);
    ga[o] = gtab[((int64_t)src[e] * n_types + b) * H + hh] * inv_rd[w];
}

// ------------------------------------------------------------------------------------------ pooled coefficients (forward of a readout-fused layer)
// thread = (source node u, head h):  ctab[u, b, h] = sum over u's out-edges e into destination type b of exp(score[e,h] - lse[seg(e),h]) / R_dst
__global__ __launch_bounds__(256) void heat_pool_coeff_kernel(
    const float* __restrict__ score, const float* __restrict__ lse, const int32_t* __restrict__ edge_seg,
    const int32_t* __restrict__ colptr, const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
    const float* __restrict__ inv_rd, const int32_t* __restrict__ row_seg, int32_t segs_per_type, int32_t n_types, int32_t H,
    int32_t num_src, float* __restrict__ ctab) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)num_src * H) return;
    const int u = (int)(idx / H), h = (int)(idx - (int64_t)u * H);
    float cb[kPoolTypes];
#pragma unroll
    for (int b = 0; b < kPoolTypes; ++b) cb[b] = 0.f;
    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; ++j) {
        const int e = csc_eid[j], w = csc_dst[j];
        const float a = expf(score[(int64_t)e * H + h] - lse[(int64_t)edge_seg[e] * H + h]) * inv_rd[w];
        const int bin = row_seg[w] / segs_per_type;
#pragma unroll
        for (int b = 0; b < kPoolTypes; ++b) cb[b] += (bin == b) ? a : 0.f;
    }
#pragma unroll
    for (int b = 0; b < kPoolTypes; ++b)
        if (b < n_types) ctab[((int64_t)u * n_types + b) * H + h] = cb[b];
}

// ------------------------------------------------------------------------------------------ backward pass 2
// dst-major, gathers k:  delta_h = sum_e a*ga;  g_s = a*(ga - delta);  g_q[w] += g_s*c*k[src];
//                        gsc[e,h] = g_s*c;  gea[e,h] = g_s*(q.k)/sqrt_dk      (c = ea/sqrt_dk)
template <int V, int LPH, int U, bool COOP = false>
__global__ __launch_bounds__(kBlock) void heat_attn_bwd_p2_kernel(
    AttnTables tb, AttnGraph g, const float* __restrict__ e_weight, const float* __restrict__ e_bias,
    float inv_sqrt_dk, const float* __restrict__ a, const float* __restrict__ ga,
    float* __restrict__ gsc, float* __restrict__ gea, float* __restrict__ gq, int64_t ldgq) {
    constexpr int H = 64 / LPH;
    __shared__ float sm_acc[COOP ? kWavesPerBlock * 64 * V : 1];
    int lane;
    const int w = wave_uniform_node<COOP>(g, lane);
    if (w < 0) return;
    const int part = COOP ? (int)(threadIdx.x >> 6) : 0;
    constexpr int ESTEP = COOP ? kWavesPerBlock * U : U;
    const int head = lane / LPH;
    const bool leader = (lane % LPH) == 0;
    const int col = lane * V;

    float gqa[V];
#pragma unroll
    for (int i = 0; i < V; ++i) gqa[i] = 0.f;

    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    if (s1 > s0 && g.rowptr[s0] != g.rowptr[s1]) {
        float q[V];
        load_vec<V>(q, tb.q + (int64_t)w * tb.ldq + col);
        const float we = *e_weight, be = *e_bias;
        for (int s = s0; s < s1; ++s) {
            const int e0 = g.rowptr[s], e1 = g.rowptr[s + 1];
            if (e0 == e1) continue;
            float delta = 0.f;
            for (int e = e0; e < e1; ++e) {
                const int64_t o = (int64_t)e * H + head;
                delta = fmaf(a[o], ga[o], delta);
            }
            for (int e = e0 + part * U; e < e1; e += ESTEP) {
                float kk[U][V];
                float c[U], aa[U], gg[U];
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    if (e + j < e1) {
                        const int u = g.src[e + j];
                        c[j] = (we * g.sim[e + j] + be) * inv_sqrt_dk;
                        const int64_t o = (int64_t)(e + j) * H + head;
                        aa[j] = a[o];
                        gg[j] = ga[o];
                        load_vec<V>(kk[j], tb.k + (int64_t)u * tb.ldk + col);
                    }
                }
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    if (e + j < e1) {
                        float d = 0.f;
#pragma unroll
                        for (int i = 0; i < V; ++i) d = fmaf(q[i], kk[j][i], d);
                        d = group_sum<LPH>(d);
                        const float gs = aa[j] * (gg[j] - delta);
                        const float gc = gs * c[j];
#pragma unroll
                        for (int i = 0; i < V; ++i) gqa[i] = fmaf(gc, kk[j][i], gqa[i]);
                        if (leader) {
                            const int64_t o = (int64_t)(e + j) * H + head;
                            gsc[o] = gc;
                            gea[o] = gs * d * inv_sqrt_dk;
                        }
                    }
                }
            }
        }
    }
    if constexpr (COOP) {           // sum the waves' partial g_q in a fixed order
#pragma unroll
        for (int i = 0; i < V; ++i) sm_acc[(part * V + i) * 64 + lane] = gqa[i];
        __syncthreads();
        if (part != 0) return;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float x = 0.f;
#pragma unroll
            for (int pp = 0; pp < kWavesPerBlock; ++pp) x += sm_acc[(pp * V + i) * 64 + lane];
            gqa[i] = x;
        }
    }
    store_vec<V>(gq + (int64_t)w * ldgq + col, gqa);
    if (g.absmax) {                                  // slot 0 of the row (pass 3 writes slot 1: g_k, g_v): plain stores
        const uint32_t b = wave_absmax_bits<V>(gqa);
        if (lane == 0) g.absmax[2 * (int64_t)w] = b;
    }
}

// ------------------------------------------------------------------------------------------ backward pass 3
// src-major over the CSC:  g_k[u] = sum_j gsc[eid_j]*q[w_j];   g_v[u] = sum_j a[eid_j]*g_t[w_j]/R_{w_j}
template <int V, int LPH, int U, bool POOL = false>
__global__ __launch_bounds__(kBlock) void heat_attn_bwd_p3_kernel(
    const float* __restrict__ qtab, int64_t ldq, const float* __restrict__ g_t, int64_t ldgt,
    const int32_t* __restrict__ colptr, const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
    const float* __restrict__ inv_rd, const int32_t* __restrict__ order, int32_t num_nodes, int32_t xcd,
    const float* __restrict__ a, const float* __restrict__ gsc,
    float* __restrict__ gk, int64_t ldgk, float* __restrict__ gv, int64_t ldgv, uint32_t* __restrict__ absmax,
    const int32_t* __restrict__ gt_row, AttnPool pool) {
    constexpr int H = 64 / LPH;
    const int lane = threadIdx.x & 63;
    const int blk = xcd ? attn_xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    int wave = blk * kWavesPerBlock + (int)(threadIdx.x >> 6);
    wave = __builtin_amdgcn_readfirstlane(wave);
    if (wave >= num_nodes) return;
    int u = order ? order[wave] : wave;
    u = __builtin_amdgcn_readfirstlane(u);
    const int head = lane / LPH;
    const int col = lane * V;

    float gka[V], gva[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { gka[i] = 0.f; gva[i] = 0.f; }
    float cb[POOL ? kPoolTypes : 1];            // POOL: this lane's head's coefficient per destination type
#pragma unroll
    for (int b = 0; b < (POOL ? kPoolTypes : 1); ++b) cb[b] = 0.f;

    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; j += U) {
        float qq[U][V], gt[U][V];
        float aa[U], gg[U];
        int bin[U];
#pragma unroll
        for (int x = 0; x < U; ++x) {
            if (j + x < j1) {
                const int eid = csc_eid[j + x];
                const int w = csc_dst[j + x];
                const int64_t o = (int64_t)eid * H + head;
                aa[x] = a[o] * inv_rd[w];
                gg[x] = gsc[o];
                load_vec<V>(qq[x], qtab + (int64_t)w * ldq + col);
                if constexpr (POOL) bin[x] = pool.ctab_ready ? -1 : pool.row_seg[w] / pool.segs_per_type;
                else load_vec<V>(gt[x], g_t + (int64_t)(gt_row ? gt_row[w] : w) * ldgt + col);
            }
        }
#pragma unroll
        for (int x = 0; x < U; ++x) {
            if (j + x < j1) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    gka[i] = fmaf(gg[x], qq[x][i], gka[i]);
                    if constexpr (!POOL) gva[i] = fmaf(aa[x], gt[x][i], gva[i]);
                }
                if constexpr (POOL) {
#pragma unroll
                    for (int b = 0; b < kPoolTypes; ++b) cb[b] += (bin[x] == b) ? aa[x] : 0.f;      // (static register indices only)
                }
            }
        }
    }
    store_vec<V>(gk + (int64_t)u * ldgk + col, gka);
    if constexpr (POOL) {
        // g_v is never formed: its two consumers take it through the S x H factors instead (dW_v from weighted sums of h, wsi_segment_weighted_sums;
        // the g_h term right here: sum_{bin,h} c[u,bin,h] * y[type(u), seg(bin), h, :], added to the residual term the dX epilogue reads)
        const int su = pool.row_seg[u];
        const int tu = su / pool.segs_per_type, gu = su - tu * pool.segs_per_type;
        float racc[V];
        load_vec<V>(racc, pool.g_row + (int64_t)su * (V * 64) + col);
        const float om = pool.omg[tu];
#pragma unroll
        for (int i = 0; i < V; ++i) racc[i] *= om;
        const float* ybase = pool.y + (int64_t)tu * pool.num_segs * H * (V * 64);
#pragma unroll
        for (int b = 0; b < kPoolTypes; ++b) {
            if (b < pool.n_types) {
                if (pool.ctab_ready) cb[b] = pool.ctab[((int64_t)u * pool.n_types + b) * H + head];
                const float* yrow = ybase + (int64_t)(b * pool.segs_per_type + gu) * H * (V * 64) + col;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float c = __shfl(cb[b], h * LPH, 64);
                    float yv[V];
                    load_vec<V>(yv, yrow + h * (V * 64));
#pragma unroll
                    for (int i = 0; i < V; ++i) racc[i] = fmaf(c, yv[i], racc[i]);
                }
                if (!pool.ctab_ready && (lane % LPH) == 0) pool.ctab[((int64_t)u * pool.n_types + b) * H + head] = cb[b];
            }
        }
        store_vec<V>(pool.r_out + (int64_t)u * pool.ldr + col, racc);
        if (absmax) {
            const uint32_t bmax = wave_absmax_bits<V>(gka);
            if (lane == 0) absmax[2 * (int64_t)u + 1] = bmax;
        }
    } else {
        store_vec<V>(gv + (int64_t)u * ldgv + col, gva);
        if (absmax) {
            const uint32_t b = max(wave_absmax_bits<V>(gka), wave_absmax_bits<V>(gva));
            if (lane == 0) absmax[2 * (int64_t)u + 1] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------ e_linear grads
// Fixed-shape two-stage reduction (deterministic):  g_w = sum_e sim[e]*sum_h gea[e,h],  g_b = sum gea.
// Both scalars add E x H signed per-(edge, head) terms (2.5 M on the bench batch) that largely cancel: summed in fp32 the result carries ~1e-4
// of relative error from the ACCUMULATION alone (the reference's own fp32 arithmetic has the same problem).  The terms are fp32, the sums run in
// float64 - two scalars, 2.5 M double additions per layer: no measurable time - so the accumulation adds nothing to the terms' own rounding.
constexpr int kRedBlocks = 256;

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ __launch_bounds__(256) void heat_egrad_stage1(const float* __restrict__ gea, const float* __restrict__ sim,
                                                          int32_t E, int32_t H, double* __restrict__ part) {
    double sw = 0.0, sb = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)kRedBlocks * 256) {
        double r = 0.0;
        for (int h = 0; h < H; ++h) r += (double)gea[e * H + h];
        sw += r * (double)sim[e];
        sb += r;
    }
    sw = wave_sum_d(sw);
    sb = wave_sum_d(sb);
    __shared__ double sh[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = sw; sh[1][wv] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[kRedBlocks + blockIdx.x] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

__global__ __launch_bounds__(256) void heat_egrad_stage2(const double* __restrict__ part, float* __restrict__ g_e) {
    double sw = part[threadIdx.x], sb = part[kRedBlocks + threadIdx.x];
    sw = wave_sum_d(sw);
    sb = wave_sum_d(sb);
    __shared__ double sh[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = sw; sh[1][wv] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        g_e[0] = (float)((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]));
        g_e[1] = (float)((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
    }
}


// ------------------------------------------------------------------------------------------ generic path
// Any D <= 1024 and any H <= 16 with D % H == 0 (e.g. hidden 200 / 4 heads of the HGT configs, or the tiny
// widths of the golden fixtures).  Lane l holds elements l, l+64, ... (NV per lane, coalesced 256-byte
// wave accesses); the head of element e is e / d_k, so a head's dot product is a masked wave-wide sum.
// Same math, same saved tensors and the same three-pass backward as the specialised kernels; slower
// (H wave reductions per edge), used only when the (D,H) pair has no specialised instantiation.
constexpr int kHMax = 16;

template <int NV>
struct GenLane {
    int hd[NV];      // head of each owned element (or -1 beyond D)
    int col[NV];
    __device__ __forceinline__ void init(int lane, int D, int dk) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            col[i] = lane + 64 * i;
            hd[i] = (col[i] < D) ? col[i] / dk : -1;
        }
    }
    __device__ __forceinline__ void load(float (&r)[NV], const float* __restrict__ p) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = (hd[i] >= 0) ? p[col[i]] : 0.f;
    }
    __device__ __forceinline__ void store(float* __restrict__ p, const float (&r)[NV]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) if (hd[i] >= 0) p[col[i]] = r[i];
    }
    __device__ __forceinline__ float head_dot(const float (&a)[NV], const float (&b)[NV], int h) const {
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) d += (hd[i] == h) ? a[i] * b[i] : 0.f;
        return wave_sum(d);
    }
};

template <int NV>
__global__ __launch_bounds__(kBlock) void heat_attn_fwd_generic(
    AttnTables tb, AttnGraph g, const float* __restrict__ e_weight, const float* __restrict__ e_bias,
    float inv_sqrt_dk, int D, int H, float* __restrict__ t, int64_t ldt, float* __restrict__ score, float* __restrict__ lse) {
    int lane;
    const int w = wave_uniform_node(g, lane);
    if (w < 0) return;
    GenLane<NV> L;
    L.init(lane, D, D / H);
    float q[NV], tacc[NV];
    L.load(q, tb.q + (int64_t)w * tb.ldq);
#pragma unroll
    for (int i = 0; i < NV; ++i) tacc[i] = 0.f;
    const float we = *e_weight, be = *e_bias;
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    for (int s = s0; s < s1; ++s) {
        const int e0 = g.rowptr[s], e1 = g.rowptr[s + 1];
        if (e0 == e1) continue;
        float m[kHMax], l[kHMax], acc[NV];
#pragma unroll
        for (int h = 0; h < kHMax; ++h) { m[h] = -INFINITY; l[h] = 0.f; }
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int u = g.src[e];
            const float c = (we * g.sim[e] + be) * inv_sqrt_dk;
            float kk[NV], vv[NV];
            L.load(kk, tb.k + (int64_t)u * tb.ldk);
            L.load(vv, tb.v + (int64_t)u * tb.ldv);
#pragma unroll
            for (int h = 0; h < kHMax; ++h) {
                if (h < H) {
                    const float sc = L.head_dot(q, kk, h) * c;
                    if (lane == 0) score[(int64_t)e * H + h] = sc;
                    const float mn = fmaxf(m[h], sc);
                    const float scale = expf(m[h] - mn), pr = expf(sc - mn);
                    l[h] = l[h] * scale + pr;
                    m[h] = mn;
#pragma unroll
                    for (int i = 0; i < NV; ++i) if (L.hd[i] == h) acc[i] = fmaf(pr, vv[i], acc[i] * scale);
                }
            }
        }
#pragma unroll
        for (int h = 0; h < kHMax; ++h) {
            if (h < H) {
                const float inv_l = 1.f / l[h];
                if (lane == 0) lse[(int64_t)s * H + h] = m[h] + logf(l[h]);
#pragma unroll
                for (int i = 0; i < NV; ++i) if (L.hd[i] == h) tacc[i] = fmaf(acc[i], inv_l, tacc[i]);
            }
        }
    }
    const float inv_r = (s1 > s0) ? 1.f / (float)(s1 - s0) : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) tacc[i] *= inv_r;
    L.store(t + (int64_t)w * ldt, tacc);
    if (g.absmax) {
        const uint32_t b = wave_absmax_bits<NV>(tacc);
        if (lane == 0) g.absmax[w] = b;
    }
}

template <int NV>
__global__ __launch_bounds__(kBlock) void heat_attn_bwd_p1_generic(
    AttnTables tb, AttnGraph g, const float* __restrict__ g_t, int64_t ldgt, int D, int H,
    float* score_a, const float* __restrict__ lse, float* __restrict__ ga) {
    int lane;
    const int w = wave_uniform_node(g, lane);
    if (w < 0) return;
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    if (s1 == s0) return;
    GenLane<NV> L;
    L.init(lane, D, D / H);
    float gm[NV];
    L.load(gm, g_t + (int64_t)(g.gt_row ? g.gt_row[w] : w) * ldgt);
    const float inv_r = 1.f / (float)(s1 - s0);
#pragma unroll
    for (int i = 0; i < NV; ++i) gm[i] *= inv_r;
    for (int s = s0; s < s1; ++s) {
        const int e0 = g.rowptr[s], e1 = g.rowptr[s + 1];
        for (int e = e0; e < e1; ++e) {
            const int u = g.src[e];
            float vv[NV];
            L.load(vv, tb.v + (int64_t)u * tb.ldv);
            for (int h = 0; h < H; ++h) {
                const float d = L.head_dot(gm, vv, h);
                if (lane == 0) {
                    const int64_t o = (int64_t)e * H + h;
                    score_a[o] = expf(g.score_in[o] - lse[(int64_t)s * H + h]);
                    ga[o] = d;
                }
            }
        }
    }
}

template <int NV>
__global__ __launch_bounds__(kBlock) void heat_attn_bwd_p2_generic(
    AttnTables tb, AttnGraph g, const float* __restrict__ e_weight, const float* __restrict__ e_bias,
    float inv_sqrt_dk, int D, int H, const float* __restrict__ a, const float* __restrict__ ga,
    float* __restrict__ gsc, float* __restrict__ gea, float* __restrict__ gq, int64_t ldgq) {
    int lane;
    const int w = wave_uniform_node(g, lane);
    if (w < 0) return;
    GenLane<NV> L;
    L.init(lane, D, D / H);
    float gqa[NV], q[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) gqa[i] = 0.f;
    L.load(q, tb.q + (int64_t)w * tb.ldq);
    const float we = *e_weight, be = *e_bias;
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    for (int s = s0; s < s1; ++s) {
        const int e0 = g.rowptr[s], e1 = g.rowptr[s + 1];
        if (e0 == e1) continue;
        float delta[kHMax];
#pragma unroll
        for (int h = 0; h < kHMax; ++h) {
            delta[h] = 0.f;
            if (h < H) for (int e = e0; e < e1; ++e) delta[h] = fmaf(a[(int64_t)e * H + h], ga[(int64_t)e * H + h], delta[h]);
        }
        for (int e = e0; e < e1; ++e) {
            const int u = g.src[e];
            const float c = (we * g.sim[e] + be) * inv_sqrt_dk;
            float kk[NV];
            L.load(kk, tb.k + (int64_t)u * tb.ldk);
#pragma unroll
            for (int h = 0; h < kHMax; ++h) {
                if (h < H) {
                    const int64_t o = (int64_t)e * H + h;
                    const float d = L.head_dot(q, kk, h);
                    const float gs = a[o] * (ga[o] - delta[h]);
                    const float gc = gs * c;
#pragma unroll
                    for (int i = 0; i < NV; ++i) if (L.hd[i] == h) gqa[i] = fmaf(gc, kk[i], gqa[i]);
                    if (lane == 0) { gsc[o] = gc; gea[o] = gs * d * inv_sqrt_dk; }
                }
            }
        }
    }
    L.store(gq + (int64_t)w * ldgq, gqa);
    if (g.absmax) {
        const uint32_t b = wave_absmax_bits<NV>(gqa);
        if (lane == 0) g.absmax[2 * (int64_t)w] = b;
    }
}

template <int NV>
__global__ __launch_bounds__(kBlock) void heat_attn_bwd_p3_generic(
    const float* __restrict__ qtab, int64_t ldq, const float* __restrict__ g_t, int64_t ldgt, int D, int H,
    const int32_t* __restrict__ colptr, const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
    const float* __restrict__ inv_rd, const int32_t* __restrict__ order, int32_t num_nodes,
    const float* __restrict__ a, const float* __restrict__ gsc,
    float* __restrict__ gk, int64_t ldgk, float* __restrict__ gv, int64_t ldgv, uint32_t* __restrict__ absmax,
    const int32_t* __restrict__ gt_row) {
    const int lane = threadIdx.x & 63;
    int wave = (int)blockIdx.x * kWavesPerBlock + (int)(threadIdx.x >> 6);
    wave = __builtin_amdgcn_readfirstlane(wave);
    if (wave >= num_nodes) return;
    int u = order ? order[wave] : wave;
    u = __builtin_amdgcn_readfirstlane(u);
    GenLane<NV> L;
    L.init(lane, D, D / H);
    float gka[NV], gva[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { gka[i] = 0.f; gva[i] = 0.f; }
    const int j0 = colptr[u], j1 = colptr[u + 1];
    for (int j = j0; j < j1; ++j) {
        const int eid = csc_eid[j], w = csc_dst[j];
        const float ir = inv_rd[w];
        float qq[NV], gt[NV];
        L.load(qq, qtab + (int64_t)w * ldq);
        L.load(gt, g_t + (int64_t)(gt_row ? gt_row[w] : w) * ldgt);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (L.hd[i] >= 0) {
                const int64_t o = (int64_t)eid * H + L.hd[i];
                gka[i] = fmaf(gsc[o], qq[i], gka[i]);
                gva[i] = fmaf(a[o] * ir, gt[i], gva[i]);
            }
        }
    }
    L.store(gk + (int64_t)u * ldgk, gka);
    L.store(gv + (int64_t)u * ldgv, gva);
    if (absmax) {
        const uint32_t b = max(wave_absmax_bits<NV>(gka), wave_absmax_bits<NV>(gva));
        if (lane == 0) absmax[2 * (int64_t)u + 1] = b;
    }
}

template <int NV>
int launch_fwd_generic(const AttnTables& tb, const AttnGraph& g, const float* ew, const float* eb, float isd, int D, int H,
                       float* t, int64_t ldt, float* score, float* lse, hipStream_t st) {
    const int blocks = (g.num_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks == 0) return WSI_OK;
    hipLaunchKernelGGL((heat_attn_fwd_generic<NV>), dim3(blocks), dim3(kBlock), 0, st, tb, g, ew, eb, isd, D, H, t, ldt, score, lse);
    return check_launch("heat_attn_fwd(generic)");
}

template <int NV>
int launch_bwd_generic(const AttnTables& tb, const AttnGraph& gd, int32_t num_src, int32_t E, int D, int H,
                       const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, const float* inv_rd,
                       const int32_t* order_src, const float* ew, const float* eb, float isd,
                       const float* g_t, int64_t ldgt, float* score_a, const float* lse, float* ga, float* gsc, float* gea,
                       float* red_ws, float* gq, int64_t ldgq, float* gk, int64_t ldgk, float* gv, int64_t ldgv,
                       float* g_e, hipStream_t st) {
    const int blocks = (gd.num_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    const int sblocks = (num_src + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 0) {
        hipLaunchKernelGGL((heat_attn_bwd_p1_generic<NV>), dim3(blocks), dim3(kBlock), 0, st, tb, gd, g_t, ldgt, D, H, score_a, lse, ga);
        hipLaunchKernelGGL((heat_attn_bwd_p2_generic<NV>), dim3(blocks), dim3(kBlock), 0, st, tb, gd, ew, eb, isd, D, H,
                           (const float*)score_a, (const float*)ga, gsc, gea, gq, ldgq);
    }
    if (sblocks > 0)
        hipLaunchKernelGGL((heat_attn_bwd_p3_generic<NV>), dim3(sblocks), dim3(kBlock), 0, st, tb.q, tb.ldq, g_t, ldgt, D, H,
                           colptr, csc_eid, csc_dst, inv_rd, order_src, num_src, (const float*)score_a, (const float*)gsc,
                           gk, ldgk, gv, ldgv, gd.absmax, gd.gt_row);
    hipLaunchKernelGGL(heat_egrad_stage1, dim3(kRedBlocks), dim3(256), 0, st, (const float*)gea, gd.sim, E, H, reinterpret_cast<double*>(red_ws));
    hipLaunchKernelGGL(heat_egrad_stage2, dim3(1), dim3(256), 0, st, reinterpret_cast<const double*>(red_ws), g_e);
    return check_launch("heat_attn_bwd(generic)");
}

#define WSI_ATTN_GENERIC(CALL)                      \
    {                                               \
        const int nv = (D + 63) / 64;               \
        if (nv <= 1) CALL(1);                       \
        else if (nv <= 2) CALL(2);                  \
        else if (nv <= 4) CALL(4);                  \
        else if (nv <= 8) CALL(8);                  \
        else if (nv <= 16) CALL(16);                \
    }

// ------------------------------------------------------------------------------------------ side stream of the hub kernels
// The few long-running hub workgroups run CONCURRENTLY with the main launch on the non-blocking stream of the CALLER-OWNED
// context (wsi_context_create), forked from / joined to the caller's stream with events (so from the caller's point of view
// everything is still ordered on `stream`).  Without a context the hub kernels simply run in-order on the caller's stream.
// The library itself keeps no stream, event or other mutable state.
}  // namespace wsi

struct wsi_context {
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    std::mutex use;          // held from fork to join: the event pair is shared by every thread using this context
};

namespace wsi {
using SideStream = wsi_context;

// returns the stream the hub kernels go to (the side stream after a fork, else `st`)
static hipStream_t hub_fork(hipStream_t st, SideStream* ctx, SideStream*& side) {
    static const bool off = [] { const char* e = knob("WSI_HUB_SIDE_STREAM"); return e && e[0] == '0'; }();   // A/B knob, read once
    side = (ctx && ctx->s && !off) ? ctx : nullptr;
    if (side) {
        side->use.lock();
        if (hipEventRecord(side->fork, st) == hipSuccess && hipStreamWaitEvent(side->s, side->fork, 0) == hipSuccess) return side->s;
        side->use.unlock();
    }
    side = nullptr;
    return st;
}
static void hub_join(hipStream_t st, SideStream* side) {
    if (!side) return;
    (void)hipEventRecord(side->join, side->s);
    (void)hipStreamWaitEvent(st, side->join, 0);
    side->use.unlock();
}

// ------------------------------------------------------------------------------------------ dispatch
template <int V, int LPH>
struct Unroll { static constexpr int value = (V >= 8) ? 2 : 4; };

template <int V, int LPH>
int launch_fwd(const AttnTables& tb, const AttnGraph& g, const float* ew, const float* eb, float isd,
               float* t, int64_t ldt, float* score, float* lse, SideStream* ctx, hipStream_t st) {
    constexpr int U = Unroll<V, LPH>::value;
    const int blocks = (g.num_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks == 0) return WSI_OK;
    if (g.heavy_n > 0) {
        AttnGraph gh = g, gl = g;
        gh.pass = 2; gh.num_nodes = g.heavy_n;
        gl.pass = 1;
        SideStream* side;
        hipStream_t hs = hub_fork(st, ctx, side);
        hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, kHeavyUnroll, true>), dim3(g.heavy_n), dim3(kBlock), 0, hs,
                           tb, gh, ew, eb, isd, t, ldt, score, lse);
        hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, U>), dim3(blocks), dim3(kBlock), 0, st,
                           tb, gl, ew, eb, isd, t, ldt, score, lse);
        hub_join(st, side);
        return check_launch("heat_attn_fwd");
    }
#ifdef WSI_ABLATE
    if (const char* v = knob("WSI_ATTN_U")) {            // measurement build: rows in flight per wave (the shipped choice is Unroll<V, LPH>)
        if (v[0] == '4') { hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, 4>), dim3(blocks), dim3(kBlock), 0, st, tb, g, ew, eb, isd, t, ldt, score, lse); return check_launch("heat_attn_fwd"); }
        if (v[0] == '1') { hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, 1>), dim3(blocks), dim3(kBlock), 0, st, tb, g, ew, eb, isd, t, ldt, score, lse); return check_launch("heat_attn_fwd"); }
        if (v[0] == '3') { hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, 3>), dim3(blocks), dim3(kBlock), 0, st, tb, g, ew, eb, isd, t, ldt, score, lse); return check_launch("heat_attn_fwd"); }
    }
#endif
    hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, U>), dim3(blocks), dim3(kBlock), 0, st,
                       tb, g, ew, eb, isd, t, ldt, score, lse);
    return check_launch("heat_attn_fwd");
}

template <int V, int LPH>
int launch_scores_fwd(const AttnTables& tb, const AttnGraph& g, const float* ew, const float* eb, float isd,
                      float* score, float* lse, SideStream* ctx, hipStream_t st) {
    constexpr int U = Unroll<V, LPH>::value;
    const int blocks = (g.num_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks == 0) return WSI_OK;
    if (g.heavy_n > 0) {
        AttnGraph gh = g, gl = g;
        gh.pass = 2; gh.num_nodes = g.heavy_n;
        gl.pass = 1;
        SideStream* side;
        hipStream_t hs = hub_fork(st, ctx, side);
        hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, kHeavyUnroll, true, true>), dim3(g.heavy_n), dim3(kBlock), 0, hs,
                           tb, gh, ew, eb, isd, (float*)nullptr, (int64_t)0, score, lse);
        hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, U, false, true>), dim3(blocks), dim3(kBlock), 0, st,
                           tb, gl, ew, eb, isd, (float*)nullptr, (int64_t)0, score, lse);
        hub_join(st, side);
        return check_launch("heat_attn_scores_fwd");
    }
    hipLaunchKernelGGL((heat_attn_fwd_kernel<V, LPH, U, false, true>), dim3(blocks), dim3(kBlock), 0, st,
                       tb, g, ew, eb, isd, (float*)nullptr, (int64_t)0, score, lse);
    return check_launch("heat_attn_scores_fwd");
}

template <int V, int LPH>
int launch_bwd(const AttnTables& tb, const AttnGraph& gd, int32_t num_src, int32_t E,
               const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, const float* inv_rd,
               const int32_t* order_src, const float* ew, const float* eb, float isd,
               const float* g_t, int64_t ldgt, float* score_a, const float* lse, float* ga, float* gsc, float* gea,
               float* red_ws, float* gq, int64_t ldgq, float* gk, int64_t ldgk, float* gv, int64_t ldgv,
               float* g_e, const AttnPool* pool, SideStream* ctx, hipStream_t st) {
    constexpr int U = Unroll<V, LPH>::value;
    constexpr int H = 64 / LPH;
    const int blocks = (gd.num_nodes + kWavesPerBlock - 1) / kWavesPerBlock;
    const int sblocks = (num_src + kWavesPerBlock - 1) / kWavesPerBlock;
    const bool ph = pool && pool->h;            // pass 1 gathers the layer input instead of v
    const bool pflat = pool && pool->gtab;      // pass 1 is a per-edge lookup: one flat launch ahead of everything else
    if (pflat && E > 0) {
        const int64_t EH = (int64_t)E * H;
        hipLaunchKernelGGL(heat_attn_bwd_p1_flat_kernel, dim3((unsigned)((EH + 255) / 256)), dim3(256), 0, st, gd.src, pool->edge_seg, pool->seg_dst,
                           pool->row_seg, pool->segs_per_type, pool->n_types, inv_rd, pool->gtab, lse, EH, (int32_t)H, gd.score_in, score_a, ga);
    }
    if (blocks > 0 && gd.heavy_n > 0) {
        AttnGraph gh = gd, gl = gd;
        gh.pass = 2; gh.num_nodes = gd.heavy_n;
        gl.pass = 1;
        const dim3 hb(gd.heavy_n);
        // two independent chains (pass 2 of a node only needs pass 1 of the same node): hubs on the side stream
        SideStream* side;
        hipStream_t hs = hub_fork(st, ctx, side);
        if (pflat) {}
        else if (ph) hipLaunchKernelGGL((heat_attn_bwd_p1_kernel<V, LPH, kHeavyUnroll, true, true>), hb, dim3(kBlock), 0, hs, tb, gh, g_t, ldgt, score_a, lse, ga, *pool);
        else hipLaunchKernelGGL((heat_attn_bwd_p1_kernel<V, LPH, kHeavyUnroll, true>), hb, dim3(kBlock), 0, hs, tb, gh, g_t, ldgt, score_a, lse, ga, AttnPool{});
        hipLaunchKernelGGL((heat_attn_bwd_p2_kernel<V, LPH, kHeavyUnroll, true>), hb, dim3(kBlock), 0, hs,
                           tb, gh, ew, eb, isd, (const float*)score_a, (const float*)ga, gsc, gea, gq, ldgq);
        if (pflat) {}
        else if (ph) hipLaunchKernelGGL((heat_attn_bwd_p1_kernel<V, LPH, U, false, true>), dim3(blocks), dim3(kBlock), 0, st, tb, gl, g_t, ldgt, score_a, lse, ga, *pool);
        else hipLaunchKernelGGL((heat_attn_bwd_p1_kernel<V, LPH, U>), dim3(blocks), dim3(kBlock), 0, st, tb, gl, g_t, ldgt, score_a, lse, ga, AttnPool{});
        hipLaunchKernelGGL((heat_attn_bwd_p2_kernel<V, LPH, U>), dim3(blocks), dim3(kBlock), 0, st,
                           tb, gl, ew, eb, isd, (const float*)score_a, (const float*)ga, gsc, gea, gq, ldgq);
        hub_join(st, side);
    } else if (blocks > 0) {
        if (pflat) {}
        else if (ph) hipLaunchKernelGGL((heat_attn_bwd_p1_kernel<V, LPH, U, false, true>), dim3(blocks), dim3(kBlock), 0, st,
                                        tb, gd, g_t, ldgt, score_a, lse, ga, *pool);
        else hipLaunchKernelGGL((heat_attn_bwd_p1_kernel<V, LPH, U>), dim3(blocks), dim3(kBlock), 0, st,
                                tb, gd, g_t, ldgt, score_a, lse, ga, AttnPool{});
        hipLaunchKernelGGL((heat_attn_bwd_p2_kernel<V, LPH, U>), dim3(blocks), dim3(kBlock), 0, st,
                           tb, gd, ew, eb, isd, (const float*)score_a, (const float*)ga, gsc, gea, gq, ldgq);
    }
    if (sblocks > 0 && pool)
        hipLaunchKernelGGL((heat_attn_bwd_p3_kernel<V, LPH, U, true>), dim3(sblocks), dim3(kBlock), 0, st,
                           tb.q, tb.ldq, g_t, ldgt, colptr, csc_eid, csc_dst, inv_rd, order_src, num_src, gd.xcd,
                           (const float*)score_a, (const float*)gsc, gk, ldgk, gv, ldgv, gd.absmax, gd.gt_row, *pool);
    else if (sblocks > 0)
        hipLaunchKernelGGL((heat_attn_bwd_p3_kernel<V, LPH, U>), dim3(sblocks), dim3(kBlock), 0, st,
                           tb.q, tb.ldq, g_t, ldgt, colptr, csc_eid, csc_dst, inv_rd, order_src, num_src, gd.xcd,
                           (const float*)score_a, (const float*)gsc, gk, ldgk, gv, ldgv, gd.absmax, gd.gt_row, AttnPool{});
    hipLaunchKernelGGL(heat_egrad_stage1, dim3(kRedBlocks), dim3(256), 0, st, (const float*)gea, gd.sim, E, H, reinterpret_cast<double*>(red_ws));
    hipLaunchKernelGGL(heat_egrad_stage2, dim3(1), dim3(256), 0, st, reinterpret_cast<const double*>(red_ws), g_e);
    return check_launch("heat_attn_bwd");
}

#define WSI_ATTN_DISPATCH(CALL)                                                        \
    switch (D) {                                                                       \
        case 512: switch (H) { case 1: CALL(8, 64); case 2: CALL(8, 32); case 4: CALL(8, 16);   \
                               case 8: CALL(8, 8); case 16: CALL(8, 4); default: break; } break; \
        case 256: switch (H) { case 1: CALL(4, 64); case 2: CALL(4, 32); case 4: CALL(4, 16);   \
                               case 8: CALL(4, 8); case 16: CALL(4, 4); default: break; } break; \
        case 128: switch (H) { case 1: CALL(2, 64); case 2: CALL(2, 32); case 4: CALL(2, 16);   \
                               case 8: CALL(2, 8); case 16: CALL(2, 4); default: break; } break; \
        default: break;                                                                \
    }

// flags bits 8..23: hub threshold (0 = the default kHeavyDegree)
static int32_t heavy_degree(int32_t flags) { const int32_t t = (flags >> 8) & 0xffff; return t ? t : kHeavyDegree; }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_heat_attn_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                 int32_t num_nodes, int32_t D, int32_t H,
                                 const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                                 const int32_t* order, int32_t num_heavy, int32_t flags, const float* e_weight, const float* e_bias,
                                 float* t, int64_t ldt, float* score, float* lse, uint32_t* t_absmax, wsi_context_t* ctx, void* stream) {
    if (num_nodes < 0 || D <= 0 || H <= 0 || D % H != 0) { set_error("heat_attn_fwd: bad shape N=%d D=%d H=%d", num_nodes, D, H); return WSI_EINVAL; }
    if (num_nodes == 0) return WSI_OK;
    if (!q || !k || !v || !node_seg || !rowptr || !e_weight || !e_bias || !t || !score || !lse) { set_error("heat_attn_fwd: null pointer"); return WSI_EINVAL; }
    const bool al = (ldq | ldk | ldv | ldt) % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) && aligned16(t);
    AttnTables tb{q, ldq, k, ldk, v, ldv};
    if (num_heavy < 0 || num_heavy > num_nodes || (num_heavy > 0 && !order)) { set_error("heat_attn_fwd: num_heavy=%d needs an order of num_nodes entries", num_heavy); return WSI_EINVAL; }
    AttnGraph g{node_seg, rowptr, src, sim, order, num_nodes, num_heavy, 0, heavy_degree(flags), (flags & WSI_ATTN_XCD_CONTIGUOUS) ? 1 : 0, t_absmax, nullptr};
    const float isd = 1.0f / sqrtf((float)(D / H));
    hipStream_t st = (hipStream_t)stream;
    if (al) {
#define CALL(V, LPH) return launch_fwd<V, LPH>(tb, g, e_weight, e_bias, isd, t, ldt, score, lse, ctx, st)
        WSI_ATTN_DISPATCH(CALL)
#undef CALL
    }
    if (D <= 1024 && H <= kHMax) {
#define CALL(NV) return launch_fwd_generic<NV>(tb, g, e_weight, e_bias, isd, D, H, t, ldt, score, lse, st)
        WSI_ATTN_GENERIC(CALL)
#undef CALL
    }
    set_error("heat_attn_fwd: unsupported (D=%d, H=%d): D <= 1024, H <= 16, D %% H == 0", D, H);
    return WSI_ENOSYS;
}

extern "C" int wsi_heat_attn_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                 int32_t num_nodes, int32_t num_src, int32_t num_edges, int32_t D, int32_t H,
                                 const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                                 const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst,
                                 const float* inv_rd, const int32_t* order_dst, int32_t num_heavy, const int32_t* order_src, int32_t flags,
                                 const float* e_weight, const float* e_bias,
                                 const float* g_t, int64_t ldgt, const int32_t* g_t_row, const float* score, float* score_a, const float* lse,
                                 float* ga, float* gsc, float* gea, float* red_ws,
                                 float* gq, int64_t ldgq, float* gk, int64_t ldgk, float* gv, int64_t ldgv,
                                 float* g_e, uint32_t* g_absmax, const wsi_attn_pool_t* pool, wsi_context_t* ctx, void* stream) {
    if (num_nodes < 0 || num_src < 0 || num_edges < 0 || D <= 0 || H <= 0 || D % H != 0) { set_error("heat_attn_bwd: bad shape"); return WSI_EINVAL; }
    if (!q || !k || (!v && !(pool && (pool->h || pool->gtab))) || !node_seg || !rowptr || !colptr || !inv_rd || !e_weight || !e_bias || !g_t || !score_a ||
        !lse || !ga || !gsc || !gea || !red_ws || !gq || !gk || (!gv && !pool) || !g_e) { set_error("heat_attn_bwd: null pointer"); return WSI_EINVAL; }
    if (reinterpret_cast<uintptr_t>(red_ws) & 7) { set_error("heat_attn_bwd: red_ws must be 8-byte aligned (it holds 512 doubles)"); return WSI_EINVAL; }
    AttnPool ap{};
    if (pool) {
        if (!pool->row_seg || !pool->y || !pool->g_row || !pool->omg || !pool->r_out || !pool->ctab || pool->segs_per_type <= 0 ||
            pool->n_types <= 0 || pool->n_types > kPoolTypes || pool->ldr % 4 != 0 || !aligned16(pool->y) || !aligned16(pool->g_row) ||
            !aligned16(pool->r_out) || num_src != num_nodes) {
            set_error("heat_attn_bwd: bad pool descriptor (1..%d node types, 16-byte aligned tables, num_src == num_nodes)", kPoolTypes);
            return WSI_EINVAL;
        }
        if (pool->h && (!pool->beta || pool->ldh % 4 != 0 || !aligned16(pool->h))) { set_error("heat_attn_bwd: pool.h needs pool.beta and 16-byte aligned rows"); return WSI_EINVAL; }
        ap = AttnPool{pool->row_seg, pool->segs_per_type, pool->n_types, pool->n_types * pool->segs_per_type, pool->y, pool->g_row, pool->omg,
                      pool->r_out, pool->ldr, pool->ctab, pool->ctab_ready, pool->h, pool->ldh, pool->beta, pool->gtab, pool->edge_seg, pool->seg_dst};
        if (pool->gtab && (!pool->edge_seg || !pool->seg_dst)) { set_error("heat_attn_bwd: pool.gtab needs pool.edge_seg and pool.seg_dst"); return WSI_EINVAL; }
    }
    const bool al = (ldq | ldk | (v ? ldv : 0) | ldgt | ldgq | ldgk | (gv ? ldgv : 0)) % 4 == 0 && aligned16(q) && aligned16(k) && aligned16(v) &&
                    aligned16(g_t) && aligned16(gq) && aligned16(gk) && aligned16(gv);
    AttnTables tb{q, ldq, k, ldk, v, ldv};
    if (num_heavy < 0 || num_heavy > num_nodes || (num_heavy > 0 && !order_dst)) { set_error("heat_attn_bwd: bad num_heavy=%d", num_heavy); return WSI_EINVAL; }
    AttnGraph gd{node_seg, rowptr, src, sim, order_dst, num_nodes, num_heavy, 0, heavy_degree(flags), (flags & WSI_ATTN_XCD_CONTIGUOUS) ? 1 : 0, g_absmax, score ? score : score_a, g_t_row};
    const float isd = 1.0f / sqrtf((float)(D / H));
    hipStream_t st = (hipStream_t)stream;
#define CALL(V, LPH) return launch_bwd<V, LPH>(tb, gd, num_src, num_edges, colptr, csc_eid, csc_dst, inv_rd, order_src, e_weight, \
                                               e_bias, isd, g_t, ldgt, score_a, lse, ga, gsc, gea, red_ws, gq, ldgq, gk,  \
                                               ldgk, gv, ldgv, g_e, pool ? &ap : nullptr, ctx, st)
    if (al) { WSI_ATTN_DISPATCH(CALL) }
#undef CALL
    if (pool) { set_error("heat_attn_bwd: the pooled backward needs D in {128, 256, 512}, H | 64 and 16-byte aligned rows (D=%d, H=%d)", D, H); return WSI_ENOSYS; }
    if (D <= 1024 && H <= kHMax) {
#define CALL(NV) return launch_bwd_generic<NV>(tb, gd, num_src, num_edges, D, H, colptr, csc_eid, csc_dst, inv_rd, order_src, e_weight, e_bias, \
                                               isd, g_t, ldgt, score_a, lse, ga, gsc, gea, red_ws, gq, ldgq, gk, ldgk, gv, ldgv, g_e, st)
        WSI_ATTN_GENERIC(CALL)
#undef CALL
    }
    set_error("heat_attn_bwd: unsupported (D=%d, H=%d)", D, H);
    return WSI_ENOSYS;
}

extern "C" int wsi_heat_attn_scores_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, int32_t num_nodes, int32_t D, int32_t H,
                                        const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                                        const int32_t* order, int32_t num_heavy, int32_t flags, const float* e_weight, const float* e_bias,
                                        float* score, float* lse, wsi_context_t* ctx, void* stream) {
    if (num_nodes < 0 || D <= 0 || H <= 0 || D % H != 0) { set_error("heat_attn_scores_fwd: bad shape N=%d D=%d H=%d", num_nodes, D, H); return WSI_EINVAL; }
    if (num_nodes == 0) return WSI_OK;
    if (!q || !k || !node_seg || !rowptr || !e_weight || !e_bias || !score || !lse) { set_error("heat_attn_scores_fwd: null pointer"); return WSI_EINVAL; }
    if (num_heavy < 0 || num_heavy > num_nodes || (num_heavy > 0 && !order)) { set_error("heat_attn_scores_fwd: bad num_heavy=%d", num_heavy); return WSI_EINVAL; }
    const bool al = (ldq | ldk) % 4 == 0 && aligned16(q) && aligned16(k);
    AttnTables tb{q, ldq, k, ldk, nullptr, 0};
    AttnGraph g{node_seg, rowptr, src, sim, order, num_nodes, num_heavy, 0, heavy_degree(flags), (flags & WSI_ATTN_XCD_CONTIGUOUS) ? 1 : 0, nullptr, nullptr};
    const float isd = 1.0f / sqrtf((float)(D / H));
    hipStream_t st = (hipStream_t)stream;
    if (al) {
#define CALL(V, LPH) return launch_scores_fwd<V, LPH>(tb, g, e_weight, e_bias, isd, score, lse, ctx, st)
        WSI_ATTN_DISPATCH(CALL)
#undef CALL
    }
    set_error("heat_attn_scores_fwd: needs D in {128, 256, 512}, H | 64 and 16-byte aligned rows (D=%d, H=%d)", D, H);
    return WSI_ENOSYS;
}

extern "C" int wsi_heat_pool_coeff(const float* score, const float* lse, const int32_t* edge_seg,
                                   const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, const float* inv_rd,
                                   const int32_t* row_seg, int32_t segs_per_type, int32_t n_types, int32_t H, int32_t num_src,
                                   float* ctab, void* stream) {
    if (num_src < 0 || H <= 0 || segs_per_type <= 0 || n_types <= 0 || n_types > kPoolTypes) { set_error("heat_pool_coeff: bad argument (1..%d node types)", kPoolTypes); return WSI_EINVAL; }
    if (num_src == 0) return WSI_OK;
    if (!score || !lse || !edge_seg || !colptr || !csc_eid || !csc_dst || !inv_rd || !row_seg || !ctab) { set_error("heat_pool_coeff: null pointer"); return WSI_EINVAL; }
    const int64_t threads = (int64_t)num_src * H;
    hipLaunchKernelGGL(heat_pool_coeff_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       score, lse, edge_seg, colptr, csc_eid, csc_dst, inv_rd, row_seg, segs_per_type, n_types, H, num_src, ctab);
    return check_launch("heat_pool_coeff");
}

extern "C" int wsi_heat_pool_gtab(const float* h, int64_t ldh, int32_t D, int32_t H, const float* y, const float* beta,
                                  const int32_t* chunk_row, const int32_t* chunk_seg, int32_t num_chunks,
                                  int32_t segs_per_type, int32_t n_types, float* gtab, void* stream) {
    if (D <= 0 || H <= 0 || num_chunks < 0 || segs_per_type <= 0 || n_types <= 0 || n_types > kPoolTypes) { set_error("heat_pool_gtab: bad argument"); return WSI_EINVAL; }
    if (num_chunks == 0) return WSI_OK;
    if (!h || !y || !beta || !chunk_row || !chunk_seg || !gtab) { set_error("heat_pool_gtab: null pointer"); return WSI_EINVAL; }
    if (ldh % 4 != 0 || !aligned16(h) || !aligned16(y)) { set_error("heat_pool_gtab: 16-byte aligned rows needed"); return WSI_EINVAL; }
    const size_t lds = (size_t)n_types * H * (D + kGtabPad) * sizeof(float);
    if (lds > 64 * 1024) { set_error("heat_pool_gtab: n_types * H * (D + 4) = %d floats exceed 64 KB of LDS", n_types * H * (D + kGtabPad)); return WSI_ENOSYS; }
    hipStream_t st = (hipStream_t)stream;
    const int J = n_types * H;
    if (J > 32) { set_error("heat_pool_gtab: n_types * H = %d > 32", J); return WSI_ENOSYS; }
#define WSI_GTAB(DD)                                                                                                                                   \
    {                                                                                                                                                  \
        if (J <= 12) hipLaunchKernelGGL((heat_pool_gtab_kernel<DD, 3>), dim3(num_chunks), dim3(256), lds, st, h, ldh, y, beta, chunk_row, chunk_seg, segs_per_type, n_types, H, gtab); \
        else if (J <= 24) hipLaunchKernelGGL((heat_pool_gtab_kernel<DD, 6>), dim3(num_chunks), dim3(256), lds, st, h, ldh, y, beta, chunk_row, chunk_seg, segs_per_type, n_types, H, gtab); \
        else hipLaunchKernelGGL((heat_pool_gtab_kernel<DD, 8>), dim3(num_chunks), dim3(256), lds, st, h, ldh, y, beta, chunk_row, chunk_seg, segs_per_type, n_types, H, gtab); \
    }
    switch (D) {
        case 512: WSI_GTAB(512) break;
        case 256: WSI_GTAB(256) break;
        case 128: WSI_GTAB(128) break;
        default: set_error("heat_pool_gtab: D must be 128, 256 or 512 (D=%d)", D); return WSI_ENOSYS;
    }
#undef WSI_GTAB
    return check_launch("heat_pool_gtab");
}

extern "C" int wsi_context_create(wsi_context_t** out) {
    if (!out) { set_error("context_create: null out pointer"); return WSI_EINVAL; }
    wsi_context* c = new (std::nothrow) wsi_context;
    if (!c) { set_error("context_create: out of host memory"); return WSI_ENOMEM; }
    // highest priority: the hub workgroups are the longest serial chains of the phase, so they should win every free CU slot over
    // the main launch's short ones (heaviest-first, applied to dispatch) unless WSI_HUB_PRIORITY=0
    int lo = 0, hi = 0;
    static const bool prio = [] { const char* e = knob("WSI_HUB_PRIORITY"); return !(e && e[0] == '0'); }();
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void)hipGetLastError(); }
    if (hipStreamCreateWithPriority(&c->s, hipStreamNonBlocking, prio ? hi : lo) != hipSuccess ||
        hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join, hipEventDisableTiming) != hipSuccess) {
        set_error("context_create: %s", hipGetErrorString(hipGetLastError()));
        wsi_context_destroy(c);
        return WSI_EFAULT;
    }
    *out = c;
    return WSI_OK;
}

extern "C" void wsi_context_destroy(wsi_context_t* c) {
    if (!c) return;
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->s) (void)hipStreamDestroy(c->s);
    delete c;
}
