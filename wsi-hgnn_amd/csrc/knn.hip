// Graph-construction edge step on the GPU: exact L2 k-nearest-neighbour selection and the per-edge
// Pearson correlation that types the edges.  Replaces, behind wsi-hgnn_amd/construct.py,
//   construct_graph/graph_constructor.py:265-273  nmslib HNSW(space='l2') fit + one knnQuery per patch (approximate)
//   construct_graph/graph_constructor.py:276-282  scipy.stats.pearsonr in a Python loop over all E pairs of 1024-d vectors
// Pipeline (host side in construct.py):  X X^T row blocks on the matrix cores (wsi_gemm_grouped NT)  ->
// wsi_knn_select: per row the kc = k+pad columns with the smallest |x_j|^2 - 2 x_i.x_j  (one streaming pass, HBM-bound) ->
// wsi_pair_stats: for those candidates the EXACT sum (x_i-x_j)^2 and Pearson r from centred sums (gathers of 4 KB rows),
// final top-k by the exact distance.  The GEMM form of the distance cancels catastrophically for near neighbours; it is
// only used to shortlist, never to rank the output.
#include "common.h"
#include <limits.h>
#include <math.h>

namespace wsi {

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int64_t ldx, int n, int F, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const float* p = x + (int64_t)row * ldx;
    float s = 0.f;
    for (int c = lane; c < F; c += 64) s = fmaf(p[c], p[c], s);
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}

// lexicographic (key, index) minimum over the wave; every lane gets the result
__device__ __forceinline__ void wave_argmin(float& k, int& i) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const float k2 = __shfl_xor(k, m);
        const int i2 = __shfl_xor(i, m);
        if (k2 < k || (k2 == k && i2 < i)) { k = k2; i = i2; }
    }
}

// One wave per query row.  Each lane keeps the KC best (smallest key, then smallest column) of the columns it streams,
// sorted, in registers; the wave then pops the global minimum kc_out times.
template <int KC>
__global__ __launch_bounds__(256) void knn_select_kernel(const float* __restrict__ dots, int64_t ldd, const float* __restrict__ sqn,
                                                         int row0, int rows, int N, int kc_out, int32_t* __restrict__ cand) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= rows) return;
    const float* d = dots + (int64_t)w * ldd;
    const int self = row0 + w;
    float key[KC];
    int idx[KC];
#pragma unroll
    for (int p = 0; p < KC; ++p) { key[p] = INFINITY; idx[p] = INT_MAX; }
    auto offer = [&](float v, int j) {
        if (j == self || !(v < key[KC - 1])) return;      // strict <: of equal keys the earlier (smaller) column stays
#pragma unroll
        for (int p = KC - 1; p >= 1; --p) {
            const bool up = v < key[p - 1];                // element p-1 moves up one slot
            const bool here = !up && v < key[p];
            const float nk = up ? key[p - 1] : (here ? v : key[p]);
            const int ni = up ? idx[p - 1] : (here ? j : idx[p]);
            key[p] = nk; idx[p] = ni;
        }
        if (v < key[0]) { key[0] = v; idx[0] = j; }
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(sqn)) & 15) == 0;
    int j0 = 0;
    if (vec) {
        const int n4 = N & ~3;
        for (int j = 4 * lane; j < n4; j += 256) {
            const float4 dv = *reinterpret_cast<const float4*>(d + j);
            const float4 sv = *reinterpret_cast<const float4*>(sqn + j);
            offer(fmaf(-2.f, dv.x, sv.x), j);
            offer(fmaf(-2.f, dv.y, sv.y), j + 1);
            offer(fmaf(-2.f, dv.z, sv.z), j + 2);
            offer(fmaf(-2.f, dv.w, sv.w), j + 3);
        }
        j0 = n4;
    }
    for (int j = j0 + lane; j < N; j += 64) offer(fmaf(-2.f, d[j], sqn[j]), j);

    for (int o = 0; o < kc_out; ++o) {
        float bk = key[0];
        int bi = idx[0];
        wave_argmin(bk, bi);
        if (idx[0] == bi && bi != INT_MAX) {               // the owning lane pops its head
#pragma unroll
            for (int p = 0; p < KC - 1; ++p) { key[p] = key[p + 1]; idx[p] = idx[p + 1]; }
            key[KC - 1] = INFINITY; idx[KC - 1] = INT_MAX;
        }
        if (lane == 0) cand[(int64_t)w * kc_out + o] = (bi == INT_MAX) ? -1 : bi;
    }
}

// One wave per row i: exact squared distance and Pearson r to each of its kc candidates, then the `keep` nearest.
__global__ __launch_bounds__(256) void pair_stats_kernel(const float* __restrict__ x, int64_t ldx, int n, int F,
                                                         const int32_t* __restrict__ cand, int kc, int keep,
                                                         int32_t* __restrict__ nbr, float* __restrict__ dist2, float* __restrict__ corr) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float* xi = x + (int64_t)i * ldx;
    float si = 0.f;
    for (int c = lane; c < F; c += 64) si += xi[c];
    const float mi = wave_sum(si) / (float)F;
    float sxx = 0.f;
    for (int c = lane; c < F; c += 64) { const float a = xi[c] - mi; sxx = fmaf(a, a, sxx); }
    sxx = wave_sum(sxx);

    float my_d2 = INFINITY, my_r = 0.f;
    int my_j = INT_MAX;
    for (int cidx = 0; cidx < kc; ++cidx) {
        const int j = cand[(int64_t)i * kc + cidx];
        if (j < 0) continue;                                   // wave-uniform
        const float* xj = x + (int64_t)j * ldx;
        float sj = 0.f;
        for (int c = lane; c < F; c += 64) sj += xj[c];
        const float mj = wave_sum(sj) / (float)F;
        float sxy = 0.f, syy = 0.f, d2 = 0.f;
        for (int c = lane; c < F; c += 64) {
            const float a = xi[c], b = xj[c];
            const float ac = a - mi, bc = b - mj, df = a - b;
            sxy = fmaf(ac, bc, sxy);
            syy = fmaf(bc, bc, syy);
            d2 = fmaf(df, df, d2);
        }
        sxy = wave_sum(sxy); syy = wave_sum(syy); d2 = wave_sum(d2);
        if (lane == cidx) {
            my_d2 = d2; my_j = j;
            // scipy.stats.pearsonr: r = <xm/|xm|, ym/|ym|> clipped to [-1, 1]; a constant vector gives nan (kept: the
            // reference then types the edge 'neg' because `nan > 0` is False)
            my_r = fminf(1.f, fmaxf(-1.f, sxy / (sqrtf(sxx) * sqrtf(syy))));
            if (sxx == 0.f || syy == 0.f) my_r = NAN;
        }
    }
    // rank of this lane's candidate among all candidates by (exact d2, column)
    int rank = 0;
    for (int c = 0; c < kc; ++c) {
        const float od = __shfl(my_d2, c);
        const int oj = __shfl(my_j, c);
        rank += (od < my_d2 || (od == my_d2 && oj < my_j)) ? 1 : 0;
    }
    if (lane < kc && my_j != INT_MAX && rank < keep) {
        nbr[(int64_t)i * keep + rank] = my_j;
        dist2[(int64_t)i * keep + rank] = my_d2;
        corr[(int64_t)i * keep + rank] = my_r;
    }
}

}  // namespace wsi

using namespace wsi;

extern "C" int wsi_row_sqnorm(const float* x, int64_t ldx, int32_t n, int32_t F, float* out, void* stream) {
    if (n < 0 || F < 0 || (n > 0 && (!x || !out))) { set_error("row_sqnorm: bad arguments"); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, n, F, out);
    return check_launch("row_sqnorm");
}

extern "C" int wsi_knn_select(const float* dots, int64_t ldd, const float* sqnorm, int32_t row0, int32_t rows, int32_t N,
                              int32_t kc, int32_t* cand, void* stream) {
    if (rows < 0 || N < 0 || kc < 1 || kc > 32) { set_error("knn_select: kc=%d outside 1..32 (or negative sizes)", kc); return WSI_EINVAL; }
    if (rows == 0) return WSI_OK;
    if (!dots || !sqnorm || !cand) { set_error("knn_select: null pointer"); return WSI_EINVAL; }
    const dim3 g((rows + 3) / 4), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (kc <= 16) hipLaunchKernelGGL(knn_select_kernel<16>, g, b, 0, st, dots, ldd, sqnorm, row0, rows, N, kc, cand);
    else hipLaunchKernelGGL(knn_select_kernel<32>, g, b, 0, st, dots, ldd, sqnorm, row0, rows, N, kc, cand);
    return check_launch("knn_select");
}

extern "C" int wsi_pair_stats(const float* x, int64_t ldx, int32_t n, int32_t F, const int32_t* cand, int32_t kc, int32_t keep,
                              int32_t* nbr, float* dist2, float* corr, void* stream) {
    if (n < 0 || F < 1 || kc < 1 || kc > 64 || keep < 1 || keep > kc) {
        set_error("pair_stats: need 1 <= keep <= kc <= 64 and F >= 1 (kc=%d keep=%d F=%d)", kc, keep, F); return WSI_EINVAL; }
    if (n == 0) return WSI_OK;
    if (!x || !cand || !nbr || !dist2 || !corr) { set_error("pair_stats: null pointer"); return WSI_EINVAL; }
    hipLaunchKernelGGL(pair_stats_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, n, F, cand, kc, keep, nbr, dist2, corr);
    return check_launch("pair_stats");
}
