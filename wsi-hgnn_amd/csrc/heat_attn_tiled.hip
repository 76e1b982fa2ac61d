// HEAT relation attention, blocked for the 4 MiB L2 of an XCD (gfx950).  Same arithmetic and the same atomic-free backward as
// heat_attn.hip (models/HEATNet4.py:103-119); what changes is WHICH bytes a gather touches while they are cache-resident.
//
// heat_attn.hip gives a destination node to a wave and gathers full 2 KB rows: on a random graph the K|V working set of one
// slide (41 MB at 10k nodes, D = 512) streams from the Infinity Cache for every edge (L2 hit 0.21-0.40, 4x the compulsory
// fabric traffic).  Here a pass gathers ONE HEAD's d_k-float slice of ONE table:
//   * the order of destination (or source) nodes is cut into SPANS - contiguous pieces of the processing order that lie inside
//     one graph - and every span belongs to one of 8 PARTS = XCDs (workgroup b runs on XCD b % 8); a part walks its spans one
//     after the other and, inside a span, head 0 of every node chunk, then head 1, ...  While an XCD works on (span, head) the
//     table slice its gathers touch is rows_of_the_graph x d_k floats (2.56 MB for 10k nodes, d_k = 64): L2-resident after the
//     first touch of every row;
//   * a group of LPG = d_k / 4 adjacent lanes owns one (node, head): 16 bytes per lane, dot products reduce over the group with
//     DPP; a 256-thread workgroup covers 256 / LPG nodes of one head.  Edge indices and per-edge scalars are loaded ONCE per
//     group (lane j of the group takes edge j of the chunk) and handed round with a DPP row broadcast; rows past the end of a
//     relation slot are clamped to its last edge (an L1 hit) and weighted zero, so the U row gathers of a chunk are issued
//     back to back with no branch between them;
//   * K and V slices of one head do not fit together, so the forward is two launches (scores + log-sum-exp; aggregation) and the
//     backward four (p1: V, p2: K, p3k: Q, p3v: g_t); the per-(edge, head) exchange arrays are HEAD-MAJOR ([H][E], [H][S]) so
//     that a pass reads and writes contiguous runs.
// Everything is atomic-free and has a fixed summation order: bit-reproducible.
#include "common.h"
#include <math.h>
#include <utility>

namespace wsi {

struct TileMap {                    // by-value kernel argument (the host table of wsi_attn_tiles_t)
    int32_t part_ptr[9];            // spans of part p: [part_ptr[p], part_ptr[p+1])
    int32_t begin[WSI_ATTN_MAX_SPANS];
    int32_t end[WSI_ATTN_MAX_SPANS];
};

struct TiledGraph {
    const int32_t* node_seg;
    const int32_t* rowptr;
    const int32_t* src;
    const float* sim;
    const int32_t* order;
    int32_t num_edges;              // E: stride of the head-major per-edge arrays
    int32_t num_segs;               // S: stride of the head-major lse
};

constexpr int kTBlock = 256;

// workgroup -> (head, first position in the order, end of the span); false: nothing to do
template <int LPG>
__device__ __forceinline__ bool tile_decode(const TileMap& tm, int H, int& head, int& pos, int& pos_end) {
    constexpr int NPB = kTBlock / LPG;          // nodes per workgroup
    const int part = (int)blockIdx.x & 7;
    int i = (int)blockIdx.x >> 3;
    const int k1 = tm.part_ptr[part + 1];
    for (int k = tm.part_ptr[part]; k < k1; ++k) {
        const int b = tm.begin[k], e = tm.end[k];
        const int nch = (e - b + NPB - 1) / NPB;
        const int nb = nch * H;
        if (i < nb) {
            head = i / nch;
            pos = b + (i - head * nch) * NPB;
            pos_end = e;
            return true;
        }
        i -= nb;
    }
    return false;
}

template <int... Js, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Js...>, F&& f) { (f(std::integral_constant<int, Js>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

// value of lane J of the caller's LPG-lane group
template <int LPG, int J>
__device__ __forceinline__ int group_bcast(int x) {
    if constexpr (LPG == 16) return __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xf, 0xf, true);     // row_newbcast:J
    else return __shfl(x, J, LPG);
}
template <int LPG, int J>
__device__ __forceinline__ float group_bcastf(float x) { return __int_as_float(group_bcast<LPG, J>(__float_as_int(x))); }

template <int LPG>
__device__ __forceinline__ uint32_t group_max_bits(uint32_t b) {
    b = max(b, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xf, 0xf, true));
    b = max(b, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xf, 0xf, true));
    if (LPG >= 8) b = max(b, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x141, 0xf, 0xf, true));
    if (LPG >= 16) b = max(b, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x140, 0xf, 0xf, true));
    if (LPG >= 32) b = max(b, (uint32_t)__shfl_xor((int)b, 16, 64));
    return b;
}
__device__ __forceinline__ uint32_t absmax4_bits(const float4& a) {
    return __float_as_uint(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v) {
    acc.x = fmaf(s, v.x, acc.x); acc.y = fmaf(s, v.y, acc.y); acc.z = fmaf(s, v.z, acc.z); acc.w = fmaf(s, v.w, acc.w);
}
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------------------------------ forward, pass A: scores + lse
// score[h][e] = (q[w,h,:] . k[src[e],h,:]) * (e_weight*sim[e] + e_bias) / sqrt(d_k);  lse[h][s] = log sum_e exp(score) per relation slot
template <int LPG, int U>
__global__ __launch_bounds__(kTBlock) void heat_tiled_scores_kernel(
    TileMap tm, TiledGraph g, const float* __restrict__ qtab, int64_t ldq, const float* __restrict__ ktab, int64_t ldk, int H,
    const float* __restrict__ e_weight, const float* __restrict__ e_bias, float inv_sqrt_dk,
    float* __restrict__ score, float* __restrict__ lse) {
    static_assert(U <= LPG, "a chunk's indices are held one per lane of the group");
    int head, pos, pos_end;
    if (!tile_decode<LPG>(tm, H, head, pos, pos_end)) return;
    const int gl = threadIdx.x & (LPG - 1);
    pos += (int)threadIdx.x / LPG;
    if (pos >= pos_end) return;
    const int w = g.order ? g.order[pos] : pos;
    const int col = head * (LPG * 4) + gl * 4;
    const float4 q = ldg4(qtab + (int64_t)w * ldq + col);
    const float we = *e_weight, be = *e_bias;
    float* __restrict__ sc_h = score + (int64_t)head * g.num_edges;
    float* __restrict__ lse_h = lse + (int64_t)head * g.num_segs;
    const float* __restrict__ kcol = ktab + col;

    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    int e_lo = (s1 > s0) ? g.rowptr[s0] : 0;
    for (int s = s0; s < s1; ++s) {
        const int e0 = e_lo, e1 = g.rowptr[s + 1];
        e_lo = e1;
        if (e0 == e1) continue;
        float m = -INFINITY, l = 0.f;
        for (int e = e0; e < e1; e += U) {
            const int ie = min(e + gl, e1 - 1);
            const int my_u = g.src[ie];
            const float my_c = (we * g.sim[ie] + be) * inv_sqrt_dk;
            float4 kk[U];
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                kk[j] = ldg4(kcol + (int64_t)group_bcast<LPG, j>(my_u) * ldk);
            });
            float mine = 0.f;
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                float sc = group_sum<LPG>(dot4(q, kk[j])) * group_bcastf<LPG, j>(my_c);
                sc = (e + j < e1) ? sc : -INFINITY;          // (j = 0 is always inside the slot: m is finite from the first step on)
                mine = (gl == j) ? sc : mine;
                const float mn = fmaxf(m, sc);
                l = l * __expf(m - mn) + __expf(sc - mn);
                m = mn;
            });
            if (gl < U && e + gl < e1) sc_h[e + gl] = mine;
        }
        if (gl == 0) lse_h[s] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------ forward, pass B: aggregation
// t[w, h, :] = (1/#slots(w)) * sum_s sum_{e in s} exp(score[h][e] - lse[h][s]) * v[src[e], h, :]
template <int LPG, int U>
__global__ __launch_bounds__(kTBlock) void heat_tiled_aggregate_kernel(
    TileMap tm, TiledGraph g, const float* __restrict__ vtab, int64_t ldv, int H,
    const float* __restrict__ score, const float* __restrict__ lse, float* __restrict__ t, int64_t ldt, uint32_t* __restrict__ absmax) {
    int head, pos, pos_end;
    if (!tile_decode<LPG>(tm, H, head, pos, pos_end)) return;
    const int gl = threadIdx.x & (LPG - 1);
    pos += (int)threadIdx.x / LPG;
    if (pos >= pos_end) return;
    const int w = g.order ? g.order[pos] : pos;
    const int col = head * (LPG * 4) + gl * 4;
    const float* __restrict__ sc_h = score + (int64_t)head * g.num_edges;
    const float* __restrict__ lse_h = lse + (int64_t)head * g.num_segs;
    const float* __restrict__ vcol = vtab + col;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    int e_lo = (s1 > s0) ? g.rowptr[s0] : 0;
    for (int s = s0; s < s1; ++s) {
        const int e0 = e_lo, e1 = g.rowptr[s + 1];
        e_lo = e1;
        if (e0 == e1) continue;
        const float ls = lse_h[s];
        for (int e = e0; e < e1; e += U) {
            const int ie = min(e + gl, e1 - 1);
            const int my_u = g.src[ie];
            const float my_p = (e + gl < e1) ? __expf(sc_h[ie] - ls) : 0.f;        // once per edge, not once per lane
            float4 vv[U];
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                vv[j] = ldg4(vcol + (int64_t)group_bcast<LPG, j>(my_u) * ldv);
            });
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                fma4(acc, group_bcastf<LPG, j>(my_p), vv[j]);
            });
        }
    }
    const float inv_r = (s1 > s0) ? 1.f / (float)(s1 - s0) : 0.f;
    acc.x *= inv_r; acc.y *= inv_r; acc.z *= inv_r; acc.w *= inv_r;
    *reinterpret_cast<float4*>(t + (int64_t)w * ldt + col) = acc;
    if (absmax) {                                   // one part per (row, head): plain stores, the consumer takes the maximum
        const uint32_t b = group_max_bits<LPG>(absmax4_bits(acc));
        if (gl == 0) absmax[(int64_t)w * H + head] = b;
    }
}

// ------------------------------------------------------------------------------------------ backward pass 1 (gathers v)
// a[h][e] = exp(score[h][e] - lse[h][s]);  ga[h][e] = (g_t[w]/R_w)[h,:] . v[src[e],h,:]
template <int LPG, int U>
__global__ __launch_bounds__(kTBlock) void heat_tiled_bwd_p1_kernel(
    TileMap tm, TiledGraph g, const float* __restrict__ vtab, int64_t ldv, int H,
    const float* __restrict__ g_t, int64_t ldgt, const int32_t* __restrict__ gt_row,
    const float* __restrict__ score, const float* __restrict__ lse, float* __restrict__ a, float* __restrict__ ga) {
    int head, pos, pos_end;
    if (!tile_decode<LPG>(tm, H, head, pos, pos_end)) return;
    const int gl = threadIdx.x & (LPG - 1);
    pos += (int)threadIdx.x / LPG;
    if (pos >= pos_end) return;
    const int w = g.order ? g.order[pos] : pos;
    const int col = head * (LPG * 4) + gl * 4;
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    if (s1 == s0) return;
    const float inv_r = 1.f / (float)(s1 - s0);
    float4 gm = ldg4(g_t + (int64_t)(gt_row ? gt_row[w] : w) * ldgt + col);
    gm.x *= inv_r; gm.y *= inv_r; gm.z *= inv_r; gm.w *= inv_r;
    const int64_t ho = (int64_t)head * g.num_edges;
    const float* __restrict__ sc_h = score + ho;
    const float* __restrict__ lse_h = lse + (int64_t)head * g.num_segs;
    float* __restrict__ a_h = a + ho;
    float* __restrict__ ga_h = ga + ho;
    const float* __restrict__ vcol = vtab + col;

    int e_lo = g.rowptr[s0];
    for (int s = s0; s < s1; ++s) {
        const int e0 = e_lo, e1 = g.rowptr[s + 1];
        e_lo = e1;
        if (e0 == e1) continue;
        const float ls = lse_h[s];
        for (int e = e0; e < e1; e += U) {
            const int ie = min(e + gl, e1 - 1);
            const int my_u = g.src[ie];
            const float my_sc = sc_h[ie];
            float4 vv[U];
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                vv[j] = ldg4(vcol + (int64_t)group_bcast<LPG, j>(my_u) * ldv);
            });
            float mine = 0.f;
            static_for<U>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const float d = group_sum<LPG>(dot4(gm, vv[j]));
                mine = (gl == j) ? d : mine;
            });
            if (gl < U && e + gl < e1) {
                a_h[ie] = __expf(my_sc - ls);
                ga_h[ie] = mine;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward pass 2 (gathers k)
// delta = sum_{e in slot} a*ga;  g_s = a*(ga - delta);  g_q[w,h,:] += g_s*c*k[src,h,:];  gsc = g_s*c;  gea = g_s*(q.k)/sqrt_dk   (c = ea/sqrt_dk)
template <int LPG, int U>
__global__ __launch_bounds__(kTBlock) void heat_tiled_bwd_p2_kernel(
    TileMap tm, TiledGraph g, const float* __restrict__ qtab, int64_t ldq, const float* __restrict__ ktab, int64_t ldk, int H,
    const float* __restrict__ e_weight, const float* __restrict__ e_bias, float inv_sqrt_dk,
    const float* __restrict__ a, const float* __restrict__ ga, float* __restrict__ gsc, float* __restrict__ gea,
    float* __restrict__ gq, int64_t ldgq, uint32_t* __restrict__ absmax, int absmax_parts, int absmax_first) {
    int head, pos, pos_end;
    if (!tile_decode<LPG>(tm, H, head, pos, pos_end)) return;
    const int gl = threadIdx.x & (LPG - 1);
    pos += (int)threadIdx.x / LPG;
    if (pos >= pos_end) return;
    const int w = g.order ? g.order[pos] : pos;
    const int col = head * (LPG * 4) + gl * 4;
    const int64_t ho = (int64_t)head * g.num_edges;
    const float* __restrict__ a_h = a + ho;
    const float* __restrict__ ga_h = ga + ho;
    float* __restrict__ gsc_h = gsc + ho;
    float* __restrict__ gea_h = gea + ho;
    const float* __restrict__ kcol = ktab + col;
    const float we = *e_weight, be = *e_bias;

    float4 gqa = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s0 = g.node_seg[w], s1 = g.node_seg[w + 1];
    if (s1 > s0) {
        const float4 q = ldg4(qtab + (int64_t)w * ldq + col);
        int e_lo = g.rowptr[s0];
        for (int s = s0; s < s1; ++s) {
            const int e0 = e_lo, e1 = g.rowptr[s + 1];
            e_lo = e1;
            if (e0 == e1) continue;
            float delta = 0.f;
            for (int e = e0; e < e1; e += LPG) {
                const int ie = e + gl;
                delta += (ie < e1) ? a_h[ie] * ga_h[ie] : 0.f;
            }
            delta = group_sum<LPG>(delta);
            for (int e = e0; e < e1; e += U) {
                const int ie = min(e + gl, e1 - 1);
                const bool ok = e + gl < e1;
                const int my_u = g.src[ie];
                const float my_c = (we * g.sim[ie] + be) * inv_sqrt_dk;
                const float my_gs = ok ? a_h[ie] * (ga_h[ie] - delta) : 0.f;
                const float my_gc = my_gs * my_c;
                float4 kk[U];
                static_for<U>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    kk[j] = ldg4(kcol + (int64_t)group_bcast<LPG, j>(my_u) * ldk);
                });
                float mine = 0.f;
                static_for<U>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    const float d = group_sum<LPG>(dot4(q, kk[j]));
                    mine = (gl == j) ? d : mine;
                    fma4(gqa, group_bcastf<LPG, j>(my_gc), kk[j]);
                });
                if (gl < U && ok) {
                    gsc_h[ie] = my_gc;
                    gea_h[ie] = my_gs * mine * inv_sqrt_dk;
                }
            }
        }
    }
    *reinterpret_cast<float4*>(gq + (int64_t)w * ldgq + col) = gqa;
    if (absmax) {
        const uint32_t b = group_max_bits<LPG>(absmax4_bits(gqa));
        if (gl == 0) absmax[(int64_t)w * absmax_parts + absmax_first + head] = b;
    }
}

// ------------------------------------------------------------------------------------------ backward pass 3 (src-major over the CSC)
// VPASS = false:  g_k[u,h,:] = sum_j gsc[h][eid_j] * q[w_j,h,:]
// VPASS = true:   g_v[u,h,:] = sum_j a[h][eid_j] / R_{w_j} * g_t[row(w_j),h,:]
template <int LPG, int U, bool VPASS>
__global__ __launch_bounds__(kTBlock) void heat_tiled_bwd_p3_kernel(
    TileMap tm, const int32_t* __restrict__ order, int H, int32_t num_edges,
    const int32_t* __restrict__ colptr, const int32_t* __restrict__ csc_eid, const int32_t* __restrict__ csc_dst,
    const float* __restrict__ inv_rd, const int32_t* __restrict__ gt_row,
    const float* __restrict__ tab, int64_t ldtab, const float* __restrict__ coef,
    float* __restrict__ out, int64_t ldo, uint32_t* __restrict__ absmax, int absmax_parts, int absmax_first) {
    int head, pos, pos_end;
    if (!tile_decode<LPG>(tm, H, head, pos, pos_end)) return;
    const int gl = threadIdx.x & (LPG - 1);
    pos += (int)threadIdx.x / LPG;
    if (pos >= pos_end) return;
    const int u = order ? order[pos] : pos;
    const int col = head * (LPG * 4) + gl * 4;
    const float* __restrict__ coef_h = coef + (int64_t)head * num_edges;
    const float* __restrict__ tcol = tab + col;

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c0 = colptr[u], c1 = colptr[u + 1];
    for (int j0 = c0; j0 < c1; j0 += U) {
        const int ij = min(j0 + gl, c1 - 1);
        const bool ok = j0 + gl < c1;
        const int my_eid = csc_eid[ij];
        int my_w = csc_dst[ij];
        float my_c = ok ? coef_h[my_eid] : 0.f;
        if constexpr (VPASS) {
            my_c *= inv_rd[my_w];
            if (gt_row) my_w = gt_row[my_w];
        }
        float4 rr[U];
        static_for<U>([&](auto J) {
            constexpr int j = decltype(J)::value;
            rr[j] = ldg4(tcol + (int64_t)group_bcast<LPG, j>(my_w) * ldtab);
        });
        static_for<U>([&](auto J) {
            constexpr int j = decltype(J)::value;
            fma4(acc, group_bcastf<LPG, j>(my_c), rr[j]);
        });
    }
    *reinterpret_cast<float4*>(out + (int64_t)u * ldo + col) = acc;
    if (absmax) {
        const uint32_t b = group_max_bits<LPG>(absmax4_bits(acc));
        if (gl == 0) absmax[(int64_t)u * absmax_parts + absmax_first + head] = b;
    }
}

// ------------------------------------------------------------------------------------------ e_linear grads (head-major gea)
// g_w = sum_e sim[e] * sum_h gea[h][e],  g_b = sum gea: fixed-shape two-stage reduction in float64 (see heat_attn.hip)
constexpr int kTRedBlocks = 256;

__device__ __forceinline__ double wave_sum_dd(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

__global__ __launch_bounds__(256) void heat_tiled_egrad_stage1(const float* __restrict__ gea, const float* __restrict__ sim,
                                                                int32_t E, int32_t H, double* __restrict__ part) {
    double sw = 0.0, sb = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)kTRedBlocks * 256) {
        double r = 0.0;
        for (int h = 0; h < H; ++h) r += (double)gea[(int64_t)h * E + e];
        sw += r * (double)sim[e];
        sb += r;
    }
    sw = wave_sum_dd(sw);
    sb = wave_sum_dd(sb);
    __shared__ double sh[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = sw; sh[1][wv] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[kTRedBlocks + blockIdx.x] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

__global__ __launch_bounds__(256) void heat_tiled_egrad_stage2(const double* __restrict__ part, float* __restrict__ g_e) {
    double sw = part[threadIdx.x], sb = part[kTRedBlocks + threadIdx.x];
    sw = wave_sum_dd(sw);
    sb = wave_sum_dd(sb);
    __shared__ double sh[2][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[0][wv] = sw; sh[1][wv] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        g_e[0] = (float)((sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]));
        g_e[1] = (float)((sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
    }
}

static int tile_grid(const wsi_attn_tiles_t* tiles, int H, int lpg, TileMap& tm) {
    const int npb = kTBlock / lpg;
    int worst = 0;
    for (int p = 0; p < 8; ++p) {
        int nb = 0;
        for (int k = tiles->part_ptr[p]; k < tiles->part_ptr[p + 1]; ++k)
            nb += ((tiles->end[k] - tiles->begin[k] + npb - 1) / npb) * H;
        worst = nb > worst ? nb : worst;
    }
    for (int p = 0; p < 9; ++p) tm.part_ptr[p] = tiles->part_ptr[p];
    for (int k = 0; k < tiles->part_ptr[8]; ++k) { tm.begin[k] = tiles->begin[k]; tm.end[k] = tiles->end[k]; }
    return worst * 8;
}

static bool tiles_ok(const wsi_attn_tiles_t* tiles, int32_t num_nodes) {
    if (!tiles || tiles->part_ptr[0] != 0) return false;
    for (int p = 0; p < 8; ++p)
        if (tiles->part_ptr[p + 1] < tiles->part_ptr[p]) return false;
    if (tiles->part_ptr[8] > WSI_ATTN_MAX_SPANS) return false;
    for (int k = 0; k < tiles->part_ptr[8]; ++k)
        if (tiles->begin[k] < 0 || tiles->end[k] < tiles->begin[k] || tiles->end[k] > num_nodes) return false;
    return true;
}

static bool aligned16t(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace wsi

using namespace wsi;

#define WSI_TILED_DISPATCH(CALL)                 \
    switch (dk) {                                \
        case 32: CALL(8); break;                 \
        case 64: CALL(16); break;                \
        case 128: CALL(32); break;               \
        default: break;                          \
    }

extern "C" int wsi_heat_attn_tiled_fwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       int32_t num_nodes, int32_t num_edges, int32_t num_segs, int32_t D, int32_t H,
                                       const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                                       const int32_t* order, const wsi_attn_tiles_t* tiles, int32_t flags,
                                       const float* e_weight, const float* e_bias,
                                       float* t, int64_t ldt, float* score, float* lse, uint32_t* t_absmax, void* stream) {
    if (num_nodes < 0 || num_edges < 0 || num_segs < 0 || D <= 0 || H <= 0 || D % H != 0) { set_error("heat_attn_tiled_fwd: bad shape N=%d D=%d H=%d", num_nodes, D, H); return WSI_EINVAL; }
    if (num_nodes == 0) return WSI_OK;
    if (!q || !k || !node_seg || !rowptr || !src || !sim || !e_weight || !e_bias || !score || !lse || (v && !t)) { set_error("heat_attn_tiled_fwd: null pointer"); return WSI_EINVAL; }
    if (!tiles_ok(tiles, num_nodes)) { set_error("heat_attn_tiled_fwd: bad tile table (<= %d spans inside [0, num_nodes), 8 parts)", WSI_ATTN_MAX_SPANS); return WSI_EINVAL; }
    const int dk = D / H;
    if ((ldq | ldk | (v ? (ldv | ldt) : 0)) % 4 != 0 || !aligned16t(q) || !aligned16t(k) || !aligned16t(v) || !aligned16t(t)) {
        set_error("heat_attn_tiled_fwd: 16-byte aligned rows needed"); return WSI_ENOSYS;
    }
    TiledGraph g{node_seg, rowptr, src, sim, order, num_edges, num_segs};
    const float isd = 1.0f / sqrtf((float)dk);
    hipStream_t st = (hipStream_t)stream;
    const int u = (flags >> 4) & 0xf;          // measurement: rows in flight per group (0 = default)
#define LAUNCH(LPG, UU)                                                                                                                    \
    {                                                                                                                                      \
        TileMap tm;                                                                                                                        \
        const int grid = tile_grid(tiles, H, LPG, tm);                                                                                     \
        if (grid == 0) return WSI_OK;                                                                                                      \
        hipLaunchKernelGGL((heat_tiled_scores_kernel<LPG, UU>), dim3(grid), dim3(kTBlock), 0, st, tm, g, q, ldq, k, ldk, (int)H,          \
                           e_weight, e_bias, isd, score, lse);                                                                             \
        if (v) hipLaunchKernelGGL((heat_tiled_aggregate_kernel<LPG, UU>), dim3(grid), dim3(kTBlock), 0, st, tm, g, v, ldv, (int)H,        \
                                  (const float*)score, (const float*)lse, t, ldt, t_absmax);                                               \
        return check_launch("heat_attn_tiled_fwd");                                                                                        \
    }
#define CALL(LPG)                                  \
    {                                              \
        if (u == 2) LAUNCH(LPG, 2)                 \
        else if (u == 8) LAUNCH(LPG, 8)            \
        else LAUNCH(LPG, 4)                        \
    }
    WSI_TILED_DISPATCH(CALL)
#undef CALL
#undef LAUNCH
    set_error("heat_attn_tiled_fwd: needs d_k = D / H in {32, 64, 128} (D=%d, H=%d)", D, H);
    return WSI_ENOSYS;
}

extern "C" int wsi_heat_attn_tiled_bwd(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       int32_t num_nodes, int32_t num_edges, int32_t num_segs, int32_t D, int32_t H,
                                       const int32_t* node_seg, const int32_t* rowptr, const int32_t* src, const float* sim,
                                       const int32_t* colptr, const int32_t* csc_eid, const int32_t* csc_dst, const float* inv_rd,
                                       const int32_t* order_dst, const int32_t* order_src, const wsi_attn_tiles_t* tiles, int32_t flags,
                                       const float* e_weight, const float* e_bias,
                                       const float* g_t, int64_t ldgt, const int32_t* g_t_row,
                                       const float* score, const float* lse, float* a, float* ga, float* gsc, float* gea, float* red_ws,
                                       float* gq, int64_t ldgq, float* gk, int64_t ldgk, float* gv, int64_t ldgv,
                                       float* g_e, uint32_t* g_absmax, void* stream) {
    if (num_nodes < 0 || num_edges < 0 || num_segs < 0 || D <= 0 || H <= 0 || D % H != 0) { set_error("heat_attn_tiled_bwd: bad shape"); return WSI_EINVAL; }
    if (!q || !k || !node_seg || !rowptr || !src || !sim || !colptr || !csc_eid || !csc_dst || !inv_rd || !e_weight || !e_bias || !score || !lse ||
        !a || !ga || !gsc || !gea || !red_ws || !gq || !gk || !g_e || (v && (!g_t || !gv))) { set_error("heat_attn_tiled_bwd: null pointer"); return WSI_EINVAL; }
    if (reinterpret_cast<uintptr_t>(red_ws) & 7) { set_error("heat_attn_tiled_bwd: red_ws must be 8-byte aligned (it holds 512 doubles)"); return WSI_EINVAL; }
    if (!tiles_ok(tiles, num_nodes)) { set_error("heat_attn_tiled_bwd: bad tile table"); return WSI_EINVAL; }
    const int dk = D / H;
    if ((ldq | ldk | ldgq | ldgk | (v ? (ldv | ldgt | ldgv) : 0)) % 4 != 0 || !aligned16t(q) || !aligned16t(k) || !aligned16t(v) || !aligned16t(g_t) ||
        !aligned16t(gq) || !aligned16t(gk) || !aligned16t(gv)) { set_error("heat_attn_tiled_bwd: 16-byte aligned rows needed"); return WSI_ENOSYS; }
    TiledGraph g{node_seg, rowptr, src, sim, order_dst, num_edges, num_segs};
    const float isd = 1.0f / sqrtf((float)dk);
    hipStream_t st = (hipStream_t)stream;
    const int u = (flags >> 4) & 0xf;
    // v == NULL: the caller has filled a and ga itself (a layer under a sum / mean readout: wsi_heat_attn_bwd's pooled pass 1) and takes
    // g_v through its S x H factors; passes 2 and 3k only.  g_absmax: [rows][3H] parts (g_q: [0,H), g_k: [H,2H), g_v: [2H,3H)), or [rows][2H] without v.
    const int parts = v ? 3 * H : 2 * H;
#define LAUNCH(LPG, UU)                                                                                                                    \
    {                                                                                                                                      \
        TileMap tm;                                                                                                                        \
        const int grid = tile_grid(tiles, H, LPG, tm);                                                                                     \
        if (grid > 0) {                                                                                                                    \
            if (v) hipLaunchKernelGGL((heat_tiled_bwd_p1_kernel<LPG, UU>), dim3(grid), dim3(kTBlock), 0, st, tm, g, v, ldv, (int)H,       \
                                      g_t, ldgt, g_t_row, score, lse, a, ga);                                                              \
            hipLaunchKernelGGL((heat_tiled_bwd_p2_kernel<LPG, UU>), dim3(grid), dim3(kTBlock), 0, st, tm, g, q, ldq, k, ldk, (int)H,      \
                               e_weight, e_bias, isd, (const float*)a, (const float*)ga, gsc, gea, gq, ldgq, g_absmax, parts, 0);          \
            hipLaunchKernelGGL((heat_tiled_bwd_p3_kernel<LPG, UU, false>), dim3(grid), dim3(kTBlock), 0, st, tm, order_src, (int)H,       \
                               num_edges, colptr, csc_eid, csc_dst, inv_rd, (const int32_t*)nullptr, q, ldq, (const float*)gsc,            \
                               gk, ldgk, g_absmax, parts, (int)H);                                                                         \
            if (v) hipLaunchKernelGGL((heat_tiled_bwd_p3_kernel<LPG, UU, true>), dim3(grid), dim3(kTBlock), 0, st, tm, order_src, (int)H, \
                                      num_edges, colptr, csc_eid, csc_dst, inv_rd, g_t_row, g_t, ldgt, (const float*)a,                    \
                                      gv, ldgv, g_absmax, parts, 2 * (int)H);                                                              \
        }                                                                                                                                  \
        hipLaunchKernelGGL(heat_tiled_egrad_stage1, dim3(kTRedBlocks), dim3(256), 0, st, (const float*)gea, sim, num_edges, H,            \
                           reinterpret_cast<double*>(red_ws));                                                                             \
        hipLaunchKernelGGL(heat_tiled_egrad_stage2, dim3(1), dim3(256), 0, st, reinterpret_cast<const double*>(red_ws), g_e);             \
        return check_launch("heat_attn_tiled_bwd");                                                                                        \
    }
#define CALL(LPG)                                  \
    {                                              \
        if (u == 2) LAUNCH(LPG, 2)                 \
        else if (u == 8) LAUNCH(LPG, 8)            \
        else LAUNCH(LPG, 4)                        \
    }
    WSI_TILED_DISPATCH(CALL)
#undef CALL
#undef LAUNCH
    set_error("heat_attn_tiled_bwd: needs d_k = D / H in {32, 64, 128} (D=%d, H=%d)", D, H);
    return WSI_ENOSYS;
}

// (the stream-form experiment of profiles/r06_l2_blocking.md - heat_stream_aggregate_kernel, wsi_heat_attn_stream_aggregate - is compiled into the
// measurement library only)
#ifdef WSI_ABLATE
#include "heat_attn_stream_ablate.inc"
#endif
