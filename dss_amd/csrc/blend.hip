// Blend (weights + normalised weighted sum + RGBA assembly) and its backward, gfx950.
//
// Replaces three elementwise torch passes + two permute copies + pytorch3d's compositor kernels
// (DSS/core/renderer.py:53-78, rasterizer.py:631-633) with one pass over the fragments each way.
//   w_k   = exp(-0.5*Q_k) * scaler[idx_k]
//   img   = sum_k feat[idx_k] * w_k / max(sum_k w_k, 1e-4)      (norm_weighted_sum, kEpsilon = 1e-4)
//   out   = (img, occ)
#include "common.h"

namespace dss {

#define BLEND_MAX_C 8

template <int C>
__global__ __launch_bounds__(256) void blend_forward_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ qv, const float *__restrict__ occ,
    const float *__restrict__ scaler, const float *__restrict__ feat, size_t npix, int K, int Crt,
    float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int Cn = (C > 0) ? C : Crt;
    float cum = 0.0f;
    for (int k = 0; k < K; ++k) {
        const int32_t p = idx[i * K + k];
        if (p < 0) continue;
        cum += expf(-0.5f * qv[i * K + k]) * scaler[p];
    }
    if (cum < 1e-4f) cum = 1e-4f;
    float acc[(C > 0) ? C : BLEND_MAX_C];
#pragma unroll
    for (int ch = 0; ch < ((C > 0) ? C : BLEND_MAX_C); ++ch) acc[ch] = 0.0f;
    for (int k = 0; k < K; ++k) {
        const int32_t p = idx[i * K + k];
        if (p < 0) continue;
        const float w = expf(-0.5f * qv[i * K + k]) * scaler[p];
#pragma unroll
        for (int ch = 0; ch < ((C > 0) ? C : BLEND_MAX_C); ++ch)
            if (ch < Cn) acc[ch] += feat[(size_t)p * Cn + ch] * w / cum;
    }
    float *o = out + i * (Cn + 1);
    if (C == 3) {
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], occ[i]);
    } else {
#pragma unroll
        for (int ch = 0; ch < ((C > 0) ? C : BLEND_MAX_C); ++ch)
            if (ch < Cn) o[ch] = acc[ch];
        o[Cn] = occ[i];
    }
}

template <int C>
__global__ __launch_bounds__(256) void blend_backward_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx, const float *__restrict__ qv,
    const float *__restrict__ scaler, size_t npix, int K, int Crt, float *__restrict__ grad_feat,
    float *__restrict__ grad_occ)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int Cn = (C > 0) ? C : Crt;
    float g[(C > 0) ? C : BLEND_MAX_C];
    const float *go = grad_out + i * (Cn + 1);
    if (C == 3) {
        const float4 v = *reinterpret_cast<const float4 *>(go);
        g[0] = v.x; g[1] = v.y; g[2] = v.z;
        grad_occ[i] = v.w;
    } else {
#pragma unroll
        for (int ch = 0; ch < ((C > 0) ? C : BLEND_MAX_C); ++ch) g[ch] = (ch < Cn) ? go[ch] : 0.0f;
        grad_occ[i] = go[Cn];
    }
    if (idx[i * K] < 0) {
        // fragments are packed front to back: an empty first slot means an empty pixel
        bool any = false;
        for (int k = 1; k < K; ++k) any = any || (idx[i * K + k] >= 0);
        if (!any) return;
    }
    float cum = 0.0f;
    for (int k = 0; k < K; ++k) {
        const int32_t p = idx[i * K + k];
        if (p < 0) continue;
        cum += expf(-0.5f * qv[i * K + k]) * scaler[p];
    }
    if (cum < 1e-4f) cum = 1e-4f;
    for (int k = 0; k < K; ++k) {
        const int32_t p = idx[i * K + k];
        if (p < 0) continue;
        const float w = expf(-0.5f * qv[i * K + k]) * scaler[p];
#pragma unroll
        for (int ch = 0; ch < ((C > 0) ? C : BLEND_MAX_C); ++ch)
            if (ch < Cn) atomicAdd(&grad_feat[(size_t)p * Cn + ch], g[ch] * w / cum);
    }
}

}  // namespace dss

using namespace dss;

extern "C" int dss_blend_forward(const int32_t *idx, const float *qvalue, const float *occ, const float *scaler,
                                 const float *feat, int N, int rows, int S, int K, int C, float *out, void *stream)
{
    if (N <= 0 || rows <= 0 || S <= 0 || K <= 0 || C < 1 || C > BLEND_MAX_C) {
        set_error("dss_blend_forward: bad sizes N=%d rows=%d S=%d K=%d C=%d", N, rows, S, K, C);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!idx || !qvalue || !occ || !scaler || !feat || !out) {
        set_error("dss_blend_forward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t npix = (size_t)N * rows * S;
    const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    hipStream_t st = as_stream(stream);
    if (C == 3)
        hipLaunchKernelGGL(blend_forward_kernel<3>, grid, block, 0, st, idx, qvalue, occ, scaler, feat, npix, K, C, out);
    else
        hipLaunchKernelGGL(blend_forward_kernel<0>, grid, block, 0, st, idx, qvalue, occ, scaler, feat, npix, K, C, out);
    return check_launch("dss_blend_forward");
}

extern "C" int dss_blend_backward(const float *grad_out, const int32_t *idx, const float *qvalue,
                                  const float *scaler, int N, int rows, int S, int K, int C, int64_t P,
                                  float *grad_feat, float *grad_occ, void *stream)
{
    if (N <= 0 || rows <= 0 || S <= 0 || K <= 0 || C < 1 || C > BLEND_MAX_C || P < 0) {
        set_error("dss_blend_backward: bad sizes N=%d rows=%d S=%d K=%d C=%d", N, rows, S, K, C);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!grad_out || !idx || !qvalue || !scaler || !grad_occ || (P > 0 && !grad_feat)) {
        set_error("dss_blend_backward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = as_stream(stream);
    if (P > 0 && hipMemsetAsync(grad_feat, 0, (size_t)P * C * sizeof(float), st) != hipSuccess)
        return check_launch("memset grad_feat");
    const size_t npix = (size_t)N * rows * S;
    const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    if (C == 3)
        hipLaunchKernelGGL(blend_backward_kernel<3>, grid, block, 0, st, grad_out, idx, qvalue, scaler, npix, K, C,
                           grad_feat, grad_occ);
    else
        hipLaunchKernelGGL(blend_backward_kernel<0>, grid, block, 0, st, grad_out, idx, qvalue, scaler, npix, K, C,
                           grad_feat, grad_occ);
    return check_launch("dss_blend_backward");
}
