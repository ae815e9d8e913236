// Shared host/device helpers for libdss_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/dss_hip.h"

#define DSS_WAVE 64
#define DSS_TILE 8           // screen tile side in pixels (one 256-thread workgroup per tile)
#define DSS_TILE_PIX (DSS_TILE * DSS_TILE)

namespace dss {

void set_error(const char *fmt, ...);

// Pixel index -> NDC centre.  Same expression, same fp32 rounding as PixToNdc
// (reference rasterization_utils.cuh:8-11): -1 + (2*i + 1.0f) / S.
__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1 + (2 * i + 1.0f) / S; }

// Same value as pix_to_ndc for every S: when S is a power of two, multiplying by the exactly
// representable 1/S rounds identically to the division (one v_mul instead of a ~10-instruction
// IEEE divide in inner loops); otherwise fall back to the division.  `pow2` is wave-uniform.
struct NdcMap {
    int S;
    float invS;
    bool pow2;
    __device__ __forceinline__ explicit NdcMap(int S_) : S(S_), invS(1.0f / (float)S_), pow2((S_ & (S_ - 1)) == 0) {}
    __device__ __forceinline__ float operator()(int i) const
    {
        const float t = 2 * i + 1.0f;
        return pow2 ? -1 + t * invS : -1 + t / S;
    }
};

// Cloud that owns packed point p (N is small; clouds are disjoint index ranges).
__device__ __forceinline__ int find_cloud(int64_t p, const int64_t *__restrict__ first_idx,
                                          const int64_t *__restrict__ num_pts, int N)
{
    for (int n = 0; n < N; ++n) {
        const int64_t f = first_idx[n];
        if (p >= f && p < f + num_pts[n]) return n;
    }
    return -1;
}

// Range [lo, hi] of NDC pixel indices i in [0,S) whose centre may satisfy |ndc(i) - x| <= r.
// Conservative (one pixel of slack each side); the exact fp32 test runs later per pixel.
// Non-finite inputs select the whole axis.  Returns false if the range is empty.
__device__ __forceinline__ bool ndc_index_range(float x, float r, int S, int &lo, int &hi)
{
    const float flo = ((x - r + 1.0f) * S - 1.0f) * 0.5f;
    const float fhi = ((x + r + 1.0f) * S - 1.0f) * 0.5f;
    lo = 0;
    hi = S - 1;
    if (flo == flo && fhi == fhi) {  // not NaN
        if (fhi < -2.0f || flo > (float)S + 1.0f) return false;
        const float a = fmaxf(flo, -2.0f), b = fminf(fhi, (float)S + 1.0f);
        lo = max(0, (int)floorf(a) - 1);
        hi = min(S - 1, (int)ceilf(b) + 1);
    }
    return lo <= hi;
}

// Wave-wide sum without LDS traffic: `__shfl_xor` lowers to ds_bpermute_b32 (an LDS round trip per
// stage, six dependent stages); DPP row operations are plain VALU.  Four DPP stages leave the sum of
// each 16-lane row in every lane of the row, then four v_readlane + scalar-operand adds combine the
// rows.  Result is wave-uniform; the summation order is fixed (deterministic).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Same range with only 0.01 px of slack (the fp32 evaluation error of the bounds and of the pixel centres
// is ~1e-4 px at S = 4096).  Used by the backward gathers: a window that creeps from 30 to 33 columns
// because of whole-pixel slack falls off the 32-lane tiling and doubles the work.
__device__ __forceinline__ bool ndc_index_range_tight(float x, float r, int S, int &lo, int &hi)
{
    const float flo = ((x - r + 1.0f) * S - 1.0f) * 0.5f;
    const float fhi = ((x + r + 1.0f) * S - 1.0f) * 0.5f;
    lo = 0;
    hi = S - 1;
    if (flo == flo && fhi == fhi) {  // not NaN
        if (fhi < -1.0f || flo > (float)S) return false;
        lo = max(0, (int)ceilf(fmaxf(flo, -1.0f) - 0.01f));
        hi = min(S - 1, (int)floorf(fminf(fhi, (float)S) + 0.01f));
    }
    return lo <= hi;
}

__device__ __forceinline__ float wave_sum(float v)
{
    v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141>(v);  // row_half_mirror
    v += dpp_f32<0x140>(v);  // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}

// Order-preserving float <-> int map (atomicMin/Max on floats of either sign).
__device__ __forceinline__ int f2ord(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__device__ __forceinline__ float eps_denom_py(float d)  // DSS/utils/mathHelper.py:10-14
{
    const float s = (float)((d > 0) - (d < 0)) + (d == 0.0f ? 1.0f : 0.0f);
    return s * fmaxf(fabsf(d), 1e-17f);
}

// EWA fragment weight exp(-Q/2) * scaler (renderer.py:53) on the transcendental unit: exp(-q/2) = 2^(-q/2 * log2 e),
// one v_exp_f32 (~1 ulp) instead of the ~15-instruction expf; 1/x as one v_rcp_f32 (1 ulp) instead of the IEEE divide
// sequence.  The image is checked at 1e-4 against the oracle (observed ~3e-7); every kernel that forms weights uses
// these two helpers, so the fused and the stand-alone blend agree bit for bit.
__device__ __forceinline__ float ewa_weight(float q, float scaler) { return __builtin_amdgcn_exp2f(-0.72134752f * q) * scaler; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int check_launch(const char *what);
int option(int which);                         // api.hip: value set by dss_set_option (0 = default)
#define DSS_MAX_DEVICES 64
#define DSS_DEV_CACHE_SLOTS 24
std::atomic<int> *device_cache(int dev);       // api.hip: DSS_DEV_CACHE_SLOTS zero-initialised slots per device ordinal (nullptr beyond 64)

// knn.hip: bbox (N,6) ordered ints = min xyz, max xyz of every cloud (NaN coordinates skipped)
int launch_cloud_bbox(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int *bbox,
                      hipStream_t st);

}  // namespace dss
