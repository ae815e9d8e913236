// Backward of the EWA splat rasterizer for gfx950 (MI355X).
//
// The reference (DSS/core/rasterizer.py:853-977) builds a 2-D FRNN grid over the visible points
// in Python (three third-party CUDA launches per cloud plus host syncs) and then runs a
// pixel-centric kernel that scatters with global atomics (rasterize_points_backward.cu:30-212).
// Here the same sum is evaluated point-centric: one wavefront per visible point gathers over the
// pixel window of radius rs around the point, reduces in registers / across the wave, and writes
// its gradient once.  No grid, no atomics, no host sync, bit-reproducible.
//
//   median_radius_kernel   rs[n] = lower median of the visible radii * radii_s (radix select,
//                          wave-aggregated LDS histograms)               rasterizer.py:885-888
//   occ_backward_kernel    occupancy surrogate gradient                   rasterize_points_backward.cu:141-178
//   zbuf_backward_kernel   z_grad scatter                                 rasterize_points.cu:823-846
//   clip_grad_kernel       per-point norm clip hook                       rasterizer.py:667-673
#include "common.h"

namespace dss {

// order-preserving float -> uint key
__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(b);
}

// LDS histogram increment with wave-level aggregation of equal bins (radii cluster in a few
// exponent bins, which would serialise plain LDS atomics 64-way).
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t bin, bool active)
{
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const unsigned long long act = __ballot(active);
        if (act == 0ull) return;
        const int leader = __builtin_ctzll(act);
        const uint32_t lbin = __shfl(bin, leader, 64);
        const bool same = active && (bin == lbin);
        const unsigned long long m = __ballot(same);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lbin], (uint32_t)__popcll(m));
        active = active && !same;
    }
    if (active) atomicAdd(&hist[bin], 1u);
}

#define MED_THREADS 1024
#define MED_BINS 2048

// Find the bin that holds rank k in hist[0..MED_BINS); returns bin and the rank inside it.
__device__ void select_bin(uint32_t *hist, uint32_t k, uint32_t *s_scan, uint32_t *s_res)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    uint32_t x = h0 + h1;
    const uint32_t v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) s_scan[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wid; ++w) woff += s_scan[w];
    const uint32_t excl = woff + x - v;
    if (k >= excl && k < excl + v) {
        if (k < excl + h0) { s_res[0] = 2 * tid; s_res[1] = k - excl; }
        else { s_res[0] = 2 * tid + 1; s_res[1] = k - excl - h0; }
    }
    __syncthreads();
}

__global__ __launch_bounds__(MED_THREADS) void median_radius_kernel(
    const float *__restrict__ radii, const uint8_t *__restrict__ visible,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, float radii_s,
    float *__restrict__ rs)
{
    __shared__ uint32_t hist[MED_BINS];
    __shared__ uint32_t s_scan[MED_THREADS / 64];
    __shared__ uint32_t s_res[2];
    __shared__ uint32_t s_total;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = first_idx[n], cnt_pts = num_pts[n];

    uint32_t prefix = 0, prefix_mask = 0, k = 0;
    // digits: bits [31:21], [20:10], [9:0]
    const int shifts[3] = {21, 10, 0};
    const uint32_t widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = tid; i < MED_BINS; i += MED_THREADS) hist[i] = 0;
        if (tid == 0) s_total = 0;
        __syncthreads();
        const int sh = shifts[pass];
        const uint32_t dmask = (1u << widths[pass]) - 1u;
        for (int64_t base = 0; base < cnt_pts; base += MED_THREADS) {
            const int64_t i = base + tid;
            bool act = (i < cnt_pts) && (visible[p0 + i] != 0);
            uint32_t kx = 0, ky = 0;
            if (act) {
                const float2 r = reinterpret_cast<const float2 *>(radii)[p0 + i];
                kx = float_key(r.x);
                ky = float_key(r.y);
            }
            hist_add(hist, (kx >> sh) & dmask, act && ((kx & prefix_mask) == prefix));
            hist_add(hist, (ky >> sh) & dmask, act && ((ky & prefix_mask) == prefix));
        }
        __syncthreads();
        if (pass == 0) {
            // total number of values = sum of the histogram
            uint32_t s = hist[2 * tid] + hist[2 * tid + 1];
            uint32_t x = s;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if ((tid & 63) == 0) atomicAdd(&s_total, x);
            __syncthreads();
            const uint32_t total = s_total;
            if (total == 0) {
                if (tid == 0) rs[n] = 0.0f;
                return;
            }
            k = (total - 1) / 2;  // lower median (torch.median)
        }
        select_bin(hist, k, s_scan, s_res);
        const uint32_t bin = s_res[0];
        k = s_res[1];
        prefix |= bin << sh;
        prefix_mask |= dmask << sh;
        __syncthreads();
    }
    if (tid == 0) rs[n] = key_float(prefix) * radii_s;
}

// ---------------------------------------------------------------------------------------------
// Occupancy surrogate gradient: one wavefront per point.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void occ_backward_kernel(
    const float *__restrict__ points, const float *__restrict__ radii,
    const uint8_t *__restrict__ visible, const float *__restrict__ rs,
    const float *__restrict__ grad_occ, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, int N, int64_t P, int S, int row0, int rows,
    float *__restrict__ grad_pts)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    float gx = 0.0f, gy = 0.0f;
    bool act = visible[p] != 0;
    int n = -1;
    if (act) {
        n = find_cloud(p, first_idx, num_pts, N);
        act = n >= 0;
    }
    if (act) {
        const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
        const float rx = radii[2 * p], ry = radii[2 * p + 1];
        const float cur_r = rs[n];
        const float cur_r2 = cur_r * cur_r;
        // rasterize_points_backward.cu:141-143
        act = !(pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f);
        int xlo, xhi, ylo, yhi;
        act = act && ndc_index_range(px, cur_r, S, xlo, xhi) && ndc_index_range(py, cur_r, S, ylo, yhi);
        if (act) {
            // band rows: image row = S-1-yi in [row0, row0+rows)
            ylo = max(ylo, S - row0 - rows);
            yhi = min(yhi, S - 1 - row0);
            const int w = xhi - xlo + 1;
            // lane tiling of the window: LW columns x (64/LW) rows per sweep
            const int lw_log = (w <= 8) ? 3 : (w <= 16) ? 4 : (w <= 32) ? 5 : 6;
            const int LW = 1 << lw_log, LH = 64 >> lw_log;
            const int lxx = lane & (LW - 1), lyy = lane >> lw_log;
            for (int yi = ylo + lyy; yi <= yhi; yi += LH) {
                const float yf = pix_to_ndc(yi, S);
                const float dy = yf - py;
                const float *grow = grad_occ + ((size_t)n * rows + (S - 1 - yi - row0)) * S;
                for (int xi = xlo + lxx; xi <= xhi; xi += LW) {
                    const float g = grow[S - 1 - xi];
                    if (g == 0.0f) continue;
                    const float xf = pix_to_ndc(xi, S);
                    const float dx = xf - px;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 > cur_r2) continue;
                    const bool outside = (fabsf(dx) > rx) || (fabsf(dy) > ry);
                    if (g > 0.0f && outside) continue;
                    if (d2 == 0.0f) continue;  // reference yields 0/0 here; see include/dss_hip.h
                    const float den = fmaxf(d2, 1e-10f);
                    gx += dx / den * g;
                    gy += dy / den * g;
                }
            }
        }
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    if (lane == 0) {
        grad_pts[3 * p] = gx;
        grad_pts[3 * p + 1] = gy;
        grad_pts[3 * p + 2] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void zbuf_backward_kernel(const int32_t *__restrict__ idx,
                                                            const float *__restrict__ grad_zbuf,
                                                            size_t npix, int K, float *__restrict__ grad_pts)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    for (int k = 0; k < K; ++k) {
        const float g = grad_zbuf[i * K + k];
        if (g == 0.0f) continue;
        const int32_t p = idx[i * K + k];
        if (p < 0) break;
        atomicAdd(&grad_pts[3 * (size_t)p + 2], g);
    }
}

__global__ __launch_bounds__(256) void clip_grad_kernel(float *__restrict__ grad, int64_t P, float clip)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float gx = grad[3 * p], gy = grad[3 * p + 1], gz = grad[3 * p + 2];
    const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
    const float scaler = fminf(nrm, clip);
    const float den = fmaxf(nrm, 1e-12f);
    grad[3 * p] = gx / den * scaler;
    grad[3 * p + 1] = gy / den * scaler;
    grad[3 * p + 2] = gz / den * scaler;
}

}  // namespace dss

using namespace dss;

extern "C" size_t dss_backward_radius_workspace(int N, int64_t P)
{
    (void)N; (void)P;
    return 256;
}

extern "C" int dss_backward_radius(const float *radii, const uint8_t *visible, const int64_t *first_idx,
                                   const int64_t *num_pts, int N, int64_t P, float radii_s, float *rs,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    (void)workspace; (void)workspace_bytes;
    if (N <= 0 || P < 0 || !rs || !first_idx || !num_pts || (P > 0 && (!radii || !visible))) {
        set_error("dss_backward_radius: bad arguments (N=%d P=%lld)", N, (long long)P);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(median_radius_kernel, dim3(N), dim3(MED_THREADS), 0, as_stream(stream), radii, visible,
                       first_idx, num_pts, radii_s, rs);
    return check_launch("dss_backward_radius");
}

extern "C" int dss_occ_backward(const float *points, const float *radii, const uint8_t *visible,
                                const float *rs, const float *grad_occ, const int64_t *first_idx,
                                const int64_t *num_pts, int N, int64_t P, int S, int row0, int row1,
                                float *grad_pts, void *stream)
{
    if (N <= 0 || P < 0 || S <= 0 || row0 < 0 || row1 > S || row0 >= row1) {
        set_error("dss_occ_backward: bad sizes N=%d P=%lld S=%d rows=[%d,%d)", N, (long long)P, S, row0, row1);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!points || !radii || !visible || !rs || !grad_occ || !first_idx || !num_pts || !grad_pts) {
        set_error("dss_occ_backward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const long long blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffll) { set_error("dss_occ_backward: P too large"); return DSS_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(occ_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), points, radii,
                       visible, rs, grad_occ, first_idx, num_pts, N, P, S, row0, row1 - row0, grad_pts);
    return check_launch("dss_occ_backward");
}

extern "C" int dss_zbuf_backward(const int32_t *idx, const float *grad_zbuf, int N, int rows, int S, int K,
                                 float *grad_pts, void *stream)
{
    if (N <= 0 || rows <= 0 || S <= 0 || K <= 0 || !idx || !grad_zbuf || !grad_pts) {
        set_error("dss_zbuf_backward: bad arguments");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t npix = (size_t)N * rows * S;
    hipLaunchKernelGGL(zbuf_backward_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream),
                       idx, grad_zbuf, npix, K, grad_pts);
    return check_launch("dss_zbuf_backward");
}

extern "C" int dss_clip_grad(float *grad_pts, int64_t P, float clip, void *stream)
{
    if (P < 0 || (P > 0 && !grad_pts)) { set_error("dss_clip_grad: bad arguments"); return DSS_ERR_INVALID_ARGUMENT; }
    if (!(clip > 0.0f) || P == 0) return DSS_OK;
    hipLaunchKernelGGL(clip_grad_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream),
                       grad_pts, P, clip);
    return check_launch("dss_clip_grad");
}

extern "C" size_t dss_splat_backward_workspace(int N, int64_t P)
{
    (void)P;
    return align_up((size_t)(N > 0 ? N : 1) * 4, 256) + dss_backward_radius_workspace(N, P);
}

extern "C" int dss_splat_backward(const float *points, const float *radii, const uint8_t *visible,
                                  const int32_t *idx, const float *grad_occ, const float *grad_zbuf,
                                  const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S, int K,
                                  float radii_s, float clip, float *grad_pts, float *rs_out, void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    if (N <= 0) { set_error("dss_splat_backward: N=%d", N); return DSS_ERR_INVALID_ARGUMENT; }
    float *rs = rs_out;
    if (!rs) {
        if (!workspace || workspace_bytes < dss_splat_backward_workspace(N, P)) {
            set_error("dss_splat_backward: workspace too small");
            return DSS_ERR_WORKSPACE;
        }
        rs = reinterpret_cast<float *>(workspace);
    }
    int rc = dss_backward_radius(radii, visible, first_idx, num_pts, N, P, radii_s, rs, nullptr, 0, stream);
    if (rc) return rc;
    rc = dss_occ_backward(points, radii, visible, rs, grad_occ, first_idx, num_pts, N, P, S, 0, S, grad_pts, stream);
    if (rc) return rc;
    if (grad_zbuf) {
        if (!idx) { set_error("dss_splat_backward: grad_zbuf given without idx"); return DSS_ERR_INVALID_ARGUMENT; }
        rc = dss_zbuf_backward(idx, grad_zbuf, N, S, S, K, grad_pts, stream);
        if (rc) return rc;
    }
    return dss_clip_grad(grad_pts, P, clip, stream);
}
