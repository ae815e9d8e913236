// Blend (weights + normalised weighted sum + RGBA assembly) and its backward, gfx950.
//
// Replaces three elementwise torch passes + two permute copies + pytorch3d's compositor kernels
// (DSS/core/renderer.py:53-78, rasterizer.py:631-633) with one pass over the fragments each way.
//   w_k   = exp(-0.5*Q_k) * scaler[idx_k]
//   img   = sum_k feat[idx_k] * w_k / max(sum_k w_k, 1e-4)      (norm_weighted_sum, kEpsilon = 1e-4)
//   out   = (img, occ)
#include "point_bodies.h"

namespace dss {

template <int C>
__global__ __launch_bounds__(256) void blend_forward_kernel(
    const int32_t *__restrict__ idx, const float *__restrict__ qv, const float *__restrict__ occ,
    const float *__restrict__ scaler, const float *__restrict__ feat, size_t npix, int K, int Crt,
    float *__restrict__ out, float *__restrict__ wsum)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int Cn = (C > 0) ? C : Crt;
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    // Summation order = the fused epilogue's (fine_tile, raster_forward.hip): lane j of a pixel's quad sums the fragments
    // k = j, j + 4, ... in ascending k, the four partial sums are combined as (p0 + p1) + (p2 + p3) -- bit-identical results.
    float pc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k0 = 0; k0 < K; k0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            if (k < K) {
                const int32_t p = idx[i * K + k];
                if (p >= 0) pc[j] += ewa_weight(qv[i * K + k], scaler[p]);
            }
        }
    }
    float cum = (pc[0] + pc[1]) + (pc[2] + pc[3]);
    if (cum < 1e-4f) cum = 1e-4f;
    if (wsum) wsum[i] = cum;
    const float inv_cum = fast_rcp(cum);
    float pa[4][CM];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ch = 0; ch < CM; ++ch) pa[j][ch] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j;
            if (k < K) {
                const int32_t p = idx[i * K + k];
                if (p >= 0) {
                    // normalised weight once per fragment (same arithmetic as the fused epilogue of the fine pass)
                    const float w = ewa_weight(qv[i * K + k], scaler[p]) * inv_cum;
#pragma unroll
                    for (int ch = 0; ch < CM; ++ch)
                        if (ch < Cn) pa[j][ch] += feat[(size_t)p * Cn + ch] * w;
                }
            }
        }
    }
    float acc[CM];
#pragma unroll
    for (int ch = 0; ch < CM; ++ch) acc[ch] = (pa[0][ch] + pa[1][ch]) + (pa[2][ch] + pa[3][ch]);
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

// Backward to the per-point features, point-centric: one wavefront per visible point gathers over
// the pixels of the point's own bounding box (a fragment (pixel,k) with idx == p can only exist where
// the hit test passed, i.e. inside |dx|<=rx, |dy|<=ry), finds its slot in the pixel's K-list and
// accumulates grad_out * w / cum.  No atomics (pytorch3d's norm_weighted_sum backward scatters with
// atomicAdd per fragment and channel), deterministic, and invisible points cost one byte load.
template <int C>
__global__ __launch_bounds__(256) void blend_backward_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx, const float *__restrict__ qv,
    const float *__restrict__ wsum, const float *__restrict__ scaler, const float *__restrict__ points,
    const float *__restrict__ radii, const uint8_t *__restrict__ visible,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t P, int S, int K,
    int Crt, int row0, int rows, float *__restrict__ grad_feat)
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    const int Cn = (C > 0) ? C : Crt;
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    float acc[CM];
#pragma unroll
    for (int ch = 0; ch < CM; ++ch) acc[ch] = 0.0f;
    if (visible[p] != 0) {
        const int n = find_cloud(p, first_idx, num_pts, N);
        if (n >= 0)
            blend_point_gather<C>(lane, p, n, grad_out, idx, qv, wsum, scaler, points, radii, S, K, Cn, row0, rows, acc);
    }
#pragma unroll
    for (int ch = 0; ch < CM; ++ch) {
        if (ch < Cn) {
            const float v = wave_sum(acc[ch]);
            if (lane == 0) grad_feat[(size_t)p * Cn + ch] = v;
        }
    }
}

// Pixel-centric scatter variant (the shape of pytorch3d's norm_weighted_sum backward): used only when
// the caller has fragments but not the splat geometry.  One atomicAdd per fragment and channel.
template <int C>
__global__ __launch_bounds__(256) void blend_backward_scatter_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx, const float *__restrict__ qv,
    const float *__restrict__ scaler, size_t npix, int K, int Crt, float *__restrict__ grad_feat)
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int Cn = (C > 0) ? C : Crt;
    float cum = 0.0f;
    bool any = false;
    for (int k = 0; k < K; ++k) {
        const int32_t p = idx[i * K + k];
        if (p < 0) continue;
        any = true;
        cum += ewa_weight(qv[i * K + k], scaler[p]);
    }
    if (!any) return;
    if (cum < 1e-4f) cum = 1e-4f;
    float g[CM];
#pragma unroll
    for (int ch = 0; ch < CM; ++ch) g[ch] = (ch < Cn) ? grad_out[i * (Cn + 1) + ch] : 0.0f;
    for (int k = 0; k < K; ++k) {
        const int32_t p = idx[i * K + k];
        if (p < 0) continue;
        const float w = ewa_weight(qv[i * K + k], scaler[p]);
#pragma unroll
        for (int ch = 0; ch < CM; ++ch)
            if (ch < Cn) atomicAdd(&grad_feat[(size_t)p * Cn + ch], g[ch] * w / cum);
    }
}

}  // namespace dss

using namespace dss;

extern "C" int dss_blend_forward(const int32_t *idx, const float *qvalue, const float *occ, const float *scaler,
                                 const float *feat, int N, int rows, int S, int K, int C, float *out, float *wsum,
                                 void *stream)
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
        hipLaunchKernelGGL(blend_forward_kernel<3>, grid, block, 0, st, idx, qvalue, occ, scaler, feat, npix, K, C, out, wsum);
    else
        hipLaunchKernelGGL(blend_forward_kernel<0>, grid, block, 0, st, idx, qvalue, occ, scaler, feat, npix, K, C, out, wsum);
    return check_launch("dss_blend_forward");
}

extern "C" int dss_blend_backward(const float *grad_out, const int32_t *idx, const float *qvalue, const float *wsum,
                                  const float *scaler, const float *points, const float *radii,
                                  const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                  int64_t P, int S, int K, int C, int row0, int row1, float *grad_feat, void *stream)
{
    if (N <= 0 || S <= 0 || K <= 0 || C < 1 || C > BLEND_MAX_C || P < 0 || row0 < 0 || row1 > S || row0 >= row1) {
        set_error("dss_blend_backward: bad sizes N=%d S=%d K=%d C=%d rows=[%d,%d)", N, S, K, C, row0, row1);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!grad_out || !idx || !qvalue || !scaler || !points || !radii || !visible || !first_idx || !num_pts || !grad_feat) {
        set_error("dss_blend_backward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const long long blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffll) { set_error("dss_blend_backward: P too large"); return DSS_ERR_UNSUPPORTED; }
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)blocks), block(256);
    if (C == 3)
        hipLaunchKernelGGL(blend_backward_kernel<3>, grid, block, 0, st, grad_out, idx, qvalue, wsum, scaler, points,
                           radii, visible, first_idx, num_pts, N, P, S, K, C, row0, row1 - row0, grad_feat);
    else
        hipLaunchKernelGGL(blend_backward_kernel<0>, grid, block, 0, st, grad_out, idx, qvalue, wsum, scaler, points,
                           radii, visible, first_idx, num_pts, N, P, S, K, C, row0, row1 - row0, grad_feat);
    return check_launch("dss_blend_backward");
}

extern "C" int dss_blend_backward_scatter(const float *grad_out, const int32_t *idx, const float *qvalue,
                                          const float *scaler, int N, int rows, int S, int K, int C, int64_t P,
                                          float *grad_feat, void *stream)
{
    if (N <= 0 || rows <= 0 || S <= 0 || K <= 0 || C < 1 || C > BLEND_MAX_C || P < 0) {
        set_error("dss_blend_backward_scatter: bad sizes N=%d rows=%d S=%d K=%d C=%d", N, rows, S, K, C);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!grad_out || !idx || !qvalue || !scaler || (P > 0 && !grad_feat)) {
        set_error("dss_blend_backward_scatter: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipStream_t st = as_stream(stream);
    if (P > 0 && hipMemsetAsync(grad_feat, 0, (size_t)P * C * sizeof(float), st) != hipSuccess)
        return check_launch("memset grad_feat");
    const size_t npix = (size_t)N * rows * S;
    const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
    if (C == 3)
        hipLaunchKernelGGL(blend_backward_scatter_kernel<3>, grid, block, 0, st, grad_out, idx, qvalue, scaler, npix, K,
                           C, grad_feat);
    else
        hipLaunchKernelGGL(blend_backward_scatter_kernel<0>, grid, block, 0, st, grad_out, idx, qvalue, scaler, npix, K,
                           C, grad_feat);
    return check_launch("dss_blend_backward_scatter");
}
