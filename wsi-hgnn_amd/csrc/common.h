// Shared device/host helpers for libwsi_hgnn.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/wsi_hgnn.h"

namespace wsi {

void set_error(const char* fmt, ...);

// Measurement switches (kernel variants, ablations, A/B knobs of tools/) exist ONLY in a library built with -DWSI_ABLATE
// (wsi_hgnn_amd.build.build_native(ablate=True) -> libwsi_hgnn_ablate.so, loaded by tools/ through _native.set_library_path).
// The product library reads no environment variable: knob() is a constant there, the variants are not instantiated, and
// tests/test_boundary.py checks that the shared object does not even import getenv.
#ifdef WSI_ABLATE
inline const char* knob(const char* name) { return getenv(name); }
#else
constexpr const char* knob(const char*) { return nullptr; }
#endif

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return WSI_EFAULT;
    }
    return WSI_OK;
}

// ---------------------------------------------------------------- cross-lane (DPP within a 16-lane row)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// Sum over aligned groups of LPH consecutive lanes; every lane of the group gets the total.
// Steps 1,2 = quad_perm, 4 = row_half_mirror, 8 = row_mirror (DPP, no LDS traffic); 16/32 cross
// the 16-lane DPP row and go through ds_bpermute (__shfl_xor).
template <int LPH>
__device__ __forceinline__ float group_sum(float x) {
    if (LPH >= 2) x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    if (LPH >= 4) x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    if (LPH >= 8) x += dpp_mov<0x141>(x);  // row_half_mirror
    if (LPH >= 16) x += dpp_mov<0x140>(x); // row_mirror
    if (LPH >= 32) x += __shfl_xor(x, 16);
    if (LPH >= 64) x += __shfl_xor(x, 32);
    return x;
}

__device__ __forceinline__ float wave_sum(float x) { return group_sum<64>(x); }

// ---------------------------------------------------------------- V contiguous floats per lane
template <int V>
__device__ __forceinline__ void load_vec(float (&r)[V], const float* __restrict__ p) {
    if constexpr (V % 4 == 0) {
#pragma unroll
        for (int i = 0; i < V / 4; ++i) {
            const float4 t = reinterpret_cast<const float4*>(p)[i];
            r[4 * i + 0] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
        }
    } else if constexpr (V == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r[0] = t.x; r[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = p[i];
    }
}

template <int V>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&r)[V]) {
    if constexpr (V % 4 == 0) {
#pragma unroll
        for (int i = 0; i < V / 4; ++i)
            reinterpret_cast<float4*>(p)[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    } else if constexpr (V == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) p[i] = r[i];
    }
}

}  // namespace wsi
