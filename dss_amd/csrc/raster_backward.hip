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
#include "point_bodies.h"

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

#define MED_THREADS 256
#define MED_BINS 2048
#define MED_PTS_PER_WG 2048

// Rank-k selection in a 2048-bin histogram (global or LDS) by one 256-thread workgroup:
// returns the bin that holds rank k and the rank inside that bin (uniform across the workgroup).
// If total_out != nullptr the sum of all bins is stored there.
__device__ __forceinline__ void select_bin(const uint32_t *hist, uint32_t k, bool k_is_lower_median,
                                           uint32_t *s_scan /*[4+3]*/, uint32_t &bin_out, uint32_t &k_out,
                                           uint32_t &total_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t h[8];
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = hist[8 * tid + i];
        v += h[i];
    }
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();  // s_scan reuse
    if (lane == 63) s_scan[wid] = x;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < MED_THREADS / 64; ++w) {
        if (w < wid) woff += s_scan[w];
        total += s_scan[w];
    }
    if (k_is_lower_median) k = (total > 0) ? (total - 1) / 2 : 0;  // torch.median = lower median
    uint32_t excl = woff + x - v;
    if (total > 0 && k >= excl && k < excl + v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (k >= excl && k < excl + h[i]) {
                s_scan[4] = 8 * tid + i;
                s_scan[5] = k - excl;
            }
            excl += h[i];
        }
    }
    __syncthreads();
    bin_out = s_scan[4];
    k_out = s_scan[5];
    total_out = total;
}

// digits of the order-preserving key: bits [31:21], [20:10], [9:0]
__device__ __forceinline__ int med_shift(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__device__ __forceinline__ uint32_t med_mask(int pass) { return pass == 2 ? 0x3ffu : 0x7ffu; }

// One radix-select pass over all packed points, many workgroups.  hist: (3, N, MED_BINS) zeroed by
// the host wrapper.  Pass p first re-derives the prefix chosen by passes < p from their finished
// histograms (cheap: 2048 bins per pass), then histograms digit p of the matching visible radii.
template <int PASS>
__global__ __launch_bounds__(MED_THREADS) void median_hist_kernel(
    const float *__restrict__ radii, const uint8_t *__restrict__ visible,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t P,
    uint32_t *__restrict__ hist)
{
    __shared__ uint32_t lh[MED_BINS];
    __shared__ uint32_t s_scan[8];
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * MED_PTS_PER_WG;
    const int64_t c1 = min(c0 + MED_PTS_PER_WG, P);
    for (int n = 0; n < N; ++n) {
        const int64_t lo = max(c0, first_idx[n]), hi = min(c1, first_idx[n] + num_pts[n]);
        if (lo >= hi) continue;  // uniform
        uint32_t prefix = 0, pmask = 0;
        if (PASS > 0) {
            uint32_t k = 0, bin, tot;
#pragma unroll
            for (int q = 0; q < PASS; ++q) {
                select_bin(hist + ((size_t)q * N + n) * MED_BINS, k, q == 0, s_scan, bin, k, tot);
                prefix |= bin << med_shift(q);
                pmask |= med_mask(q) << med_shift(q);
            }
        }
        __syncthreads();
        for (int i = tid; i < MED_BINS; i += MED_THREADS) lh[i] = 0;
        __syncthreads();
        const int sh = med_shift(PASS);
        const uint32_t dm = med_mask(PASS);
        // 8 points per thread; all loads are issued before the first use (latency overlapped).
        // Plain LDS atomics: a 64-way same-bin conflict costs ~64 LDS cycles, far less than the loads.
        constexpr int PER = MED_PTS_PER_WG / MED_THREADS;
        uint8_t vis[PER];
        float2 rr[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t i = lo + tid + (int64_t)u * MED_THREADS;
            const bool in = i < hi;
            vis[u] = in ? visible[i] : (uint8_t)0;
            rr[u] = in ? reinterpret_cast<const float2 *>(radii)[i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (vis[u]) {
                const uint32_t kx = float_key(rr[u].x), ky = float_key(rr[u].y);
                if ((kx & pmask) == prefix) atomicAdd(&lh[(kx >> sh) & dm], 1u);
                if ((ky & pmask) == prefix) atomicAdd(&lh[(ky >> sh) & dm], 1u);
            }
        }
        __syncthreads();
        uint32_t *gh = hist + ((size_t)PASS * N + n) * MED_BINS;
        for (int i = tid; i < MED_BINS; i += MED_THREADS) {
            const uint32_t c = lh[i];
            if (c) atomicAdd(&gh[i], c);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(MED_THREADS) void median_final_kernel(const uint32_t *__restrict__ hist, int N,
                                                                   float radii_s, float *__restrict__ rs)
{
    __shared__ uint32_t s_scan[8];
    const int n = blockIdx.x;
    uint32_t key = 0, k = 0, bin, tot, total0 = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        select_bin(hist + ((size_t)q * N + n) * MED_BINS, k, q == 0, s_scan, bin, k, tot);
        if (q == 0) total0 = tot;
        key |= bin << med_shift(q);
    }
    if (threadIdx.x == 0) rs[n] = total0 ? key_float(key) * radii_s : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Occupancy surrogate gradient: one wavefront per point.
// ---------------------------------------------------------------------------------------------
#ifdef DSS_FINE_TIMING
__device__ long long *g_occ_timing = nullptr;  // (P, 6) int64, developer tool only (tools/occ_timing.py)
#define OT_MARK(slot)                                                                              \
    do {                                                                                           \
        if (g_occ_timing && (threadIdx.x & 63) == 0) g_occ_timing[p * 6 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define OT_VAL(slot, v)                                                                            \
    do {                                                                                           \
        if (g_occ_timing && (threadIdx.x & 63) == 0) g_occ_timing[p * 6 + (slot)] = (long long)(v); \
    } while (0)
#else
#define OT_MARK(slot)
#define OT_VAL(slot, v)
#endif
__global__ __launch_bounds__(256) void occ_backward_kernel(
    const float *__restrict__ points, const float *__restrict__ radii,
    const uint8_t *__restrict__ visible, const float *__restrict__ rs,
    const float *__restrict__ grad_occ, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, int N, int64_t P, int S, int row0, int rows, int gstride, float clip,
    float *__restrict__ grad_pts)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    OT_MARK(0);
    OT_VAL(4, __builtin_amdgcn_s_memrealtime());
    float gx = 0.0f, gy = 0.0f;
    if (visible[p] != 0) {
        const int n = find_cloud(p, first_idx, num_pts, N);
        OT_MARK(1);
        if (n >= 0) occ_point_gather(lane, p, n, points, radii, rs, grad_occ, S, row0, rows, gstride, gx, gy);
    }
    OT_MARK(2);
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    OT_MARK(3);
    OT_VAL(5, __builtin_amdgcn_s_memrealtime());
    if (lane == 0) {
        if (clip > 0.0f) {
            // fused per-point clip hook (rasterizer.py:667-673) when no zbuf gradient follows (z grad = 0)
            const float nrm = sqrtf(gx * gx + gy * gy);
            gx = gx / fmaxf(nrm, 1e-12f) * fminf(nrm, clip);
            gy = gy / fmaxf(nrm, 1e-12f) * fminf(nrm, clip);
        }
        grad_pts[3 * p] = gx;
        grad_pts[3 * p + 1] = gy;
        grad_pts[3 * p + 2] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused single-GPU backward (dss_render_backward): the stand-alone kernels launch one wavefront per
// packed point, and tools/occ_timing.py shows that at DSS sizes they are bound by the workgroup
// DISPATCH rate (8171 workgroups take ~26 us to start, 60 % of them only to find their point
// invisible), not by the gather itself.  Here
//   visible_scan_kernel     = radix-select pass 0 + compaction of the visible point ids + zero fill of
//                             the gradients of invisible points (one pass over the flags)
//   render_backward_kernel  = persistent wavefronts (one full-occupancy grid) that walk the compacted
//                             list and evaluate BOTH gathers (blend backward + occupancy backward + clip)
//                             per visible point.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MED_THREADS) void visible_scan_kernel(
    const float *__restrict__ radii, const uint8_t *__restrict__ visible,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t P,
    uint32_t *__restrict__ hist, uint32_t *__restrict__ vis_count, int32_t *__restrict__ vis_list,
    float *__restrict__ grad_pts, float *__restrict__ grad_feat, int C)
{
    __shared__ uint32_t lh[MED_BINS];
    __shared__ uint32_t s_wave[MED_THREADS / 64];
    __shared__ uint32_t s_base;
    constexpr int PER = MED_PTS_PER_WG / MED_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * MED_PTS_PER_WG;
    const int64_t c1 = min(c0 + MED_PTS_PER_WG, P);
    // all loads first (latency overlapped)
    uint8_t vis[PER];
    float2 rr[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = c0 + tid + (int64_t)u * MED_THREADS;
        const bool in = i < c1;
        vis[u] = in ? visible[i] : (uint8_t)0;
        rr[u] = in ? reinterpret_cast<const float2 *>(radii)[i] : make_float2(0.f, 0.f);
    }
    // ---- compaction: wave ballots -> ranks; one global atomic per workgroup ----
    uint32_t rank[PER];
    uint32_t wave_cnt = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const unsigned long long m = __ballot(vis[u] != 0);
        rank[u] = wave_cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        wave_cnt += (uint32_t)__popcll(m);
    }
    if (lane == 0) s_wave[wid] = wave_cnt;
    for (int i = tid; i < MED_BINS; i += MED_THREADS) lh[i] = 0;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < MED_THREADS / 64; ++w) tot += s_wave[w];
        s_base = tot ? atomicAdd(vis_count, tot) : 0u;
    }
    __syncthreads();
    uint32_t base = s_base;
    for (int w = 0; w < wid; ++w) base += s_wave[w];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = c0 + tid + (int64_t)u * MED_THREADS;
        if (vis[u]) {
            vis_list[base + rank[u]] = (int32_t)i;
        } else if (i < c1) {
            grad_pts[3 * i] = 0.0f; grad_pts[3 * i + 1] = 0.0f; grad_pts[3 * i + 2] = 0.0f;
            if (grad_feat)
                for (int ch = 0; ch < C; ++ch) grad_feat[(size_t)i * C + ch] = 0.0f;
        }
    }
    // ---- radix-select pass 0 (digit [31:21]) per cloud overlapping this chunk ----
    for (int n = 0; n < N; ++n) {
        const int64_t lo = max(c0, first_idx[n]), hi = min(c1, first_idx[n] + num_pts[n]);
        if (lo >= hi) continue;  // uniform
        bool any = false;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t i = c0 + tid + (int64_t)u * MED_THREADS;
            if (vis[u] && i >= lo && i < hi) {
                atomicAdd(&lh[(float_key(rr[u].x) >> 21) & 0x7ffu], 1u);
                atomicAdd(&lh[(float_key(rr[u].y) >> 21) & 0x7ffu], 1u);
                any = true;
            }
        }
        (void)any;
        __syncthreads();
        uint32_t *gh = hist + (size_t)n * MED_BINS;  // pass 0 block of the (3,N,BINS) array
        for (int i = tid; i < MED_BINS; i += MED_THREADS) {
            const uint32_t c = lh[i];
            if (c) {
                atomicAdd(&gh[i], c);
                lh[i] = 0;
            }
        }
        __syncthreads();
    }
}

template <int C>
__global__ __launch_bounds__(256) void render_backward_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx, const float *__restrict__ qv,
    const float *__restrict__ wsum, const float *__restrict__ scaler, const float *__restrict__ points,
    const float *__restrict__ radii, const float *__restrict__ rs, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, const uint32_t *__restrict__ vis_count,
    const int32_t *__restrict__ vis_list, int N, int S, int K, int Crt, float clip, int row0, int rows,
    float *__restrict__ grad_feat, float *__restrict__ grad_pts)
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    const int Cn = (C > 0) ? C : Crt;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * 4;
    const uint32_t count = *vis_count;
#ifdef DSS_FINE_TIMING
    long long tm_rt0 = __builtin_amdgcn_s_memrealtime(), tm_occ = 0, tm_blend = 0, tm_pro = 0, tm_tasks = 0;
#endif
    for (uint32_t t = wave; t < count; t += n_waves) {
#ifdef DSS_FINE_TIMING
        const long long tm0 = __builtin_amdgcn_s_memtime();
#endif
        const int64_t p = vis_list[t];
        const int n = find_cloud(p, first_idx, num_pts, N);
        if (n < 0) continue;
        float gx = 0.0f, gy = 0.0f;
#ifdef DSS_FINE_TIMING
        const long long tm1 = __builtin_amdgcn_s_memtime();
#endif
        // occupancy gradient = alpha channel of the image gradient, read in place
        occ_point_gather(lane, p, n, points, radii, rs, grad_out + Cn, S, row0, rows, Cn + 1, gx, gy);
#ifdef DSS_FINE_TIMING
        gx = wave_sum(gx) * (1.0f / 64.0f) * 64.0f / 64.0f;  // force completion of the gather before the stamp
        const long long tm2 = __builtin_amdgcn_s_memtime();
        tm_pro += tm1 - tm0; tm_occ += tm2 - tm1; tm_tasks += 1;
#endif
        float acc[CM];
#pragma unroll
        for (int ch = 0; ch < CM; ++ch) acc[ch] = 0.0f;
        if (grad_feat)
            blend_point_gather<C>(lane, p, n, grad_out, idx, qv, wsum, scaler, points, radii, S, K, Cn, row0, rows, acc);
        gx = wave_sum(gx);
        gy = wave_sum(gy);
#pragma unroll
        for (int ch = 0; ch < CM; ++ch)
            if (ch < Cn) acc[ch] = wave_sum(acc[ch]);
#ifdef DSS_FINE_TIMING
        tm_blend += (long long)__builtin_amdgcn_s_memtime() - tm2;
#endif
        if (lane == 0) {
            if (clip > 0.0f) {  // rasterizer.py:667-673 (z gradient is 0 on this path)
                const float nrm = sqrtf(gx * gx + gy * gy);
                gx = gx / fmaxf(nrm, 1e-12f) * fminf(nrm, clip);
                gy = gy / fmaxf(nrm, 1e-12f) * fminf(nrm, clip);
            }
            grad_pts[3 * p] = gx;
            grad_pts[3 * p + 1] = gy;
            grad_pts[3 * p + 2] = 0.0f;
            if (grad_feat) {
#pragma unroll
                for (int ch = 0; ch < CM; ++ch)
                    if (ch < Cn) grad_feat[(size_t)p * Cn + ch] = acc[ch];
            }
        }
    }
#ifdef DSS_FINE_TIMING
    if (g_occ_timing && lane == 0) {
        long long *o = g_occ_timing + (size_t)wave * 6;
        o[0] = tm_rt0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tm_tasks; o[3] = tm_occ; o[4] = tm_blend; o[5] = tm_pro;
    }
#endif
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
    (void)P;
    return align_up((size_t)3 * (N > 0 ? N : 1) * MED_BINS * sizeof(uint32_t), 256);
}

extern "C" int dss_backward_radius(const float *radii, const uint8_t *visible, const int64_t *first_idx,
                                   const int64_t *num_pts, int N, int64_t P, float radii_s, float *rs,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    if (N <= 0 || P < 0 || !rs || !first_idx || !num_pts || (P > 0 && (!radii || !visible))) {
        set_error("dss_backward_radius: bad arguments (N=%d P=%lld)", N, (long long)P);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t need = dss_backward_radius_workspace(N, P);
    if (!workspace || workspace_bytes < need) {
        set_error("dss_backward_radius: workspace %zu bytes < required %zu", workspace_bytes, need);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    uint32_t *hist = reinterpret_cast<uint32_t *>(workspace);
    if (hipMemsetAsync(hist, 0, need, st) != hipSuccess) return check_launch("memset median hist");
    const unsigned blocks = (unsigned)((P + MED_PTS_PER_WG - 1) / MED_PTS_PER_WG);
    if (blocks > 0) {
        hipLaunchKernelGGL(median_hist_kernel<0>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
        hipLaunchKernelGGL(median_hist_kernel<1>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
        hipLaunchKernelGGL(median_hist_kernel<2>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
    }
    hipLaunchKernelGGL(median_final_kernel, dim3(N), dim3(MED_THREADS), 0, st, hist, N, radii_s, rs);
    return check_launch("dss_backward_radius");
}

static int occ_backward_impl(const float *points, const float *radii, const uint8_t *visible,
                             const float *rs, const float *grad_occ, const int64_t *first_idx,
                             const int64_t *num_pts, int N, int64_t P, int S, int row0, int row1,
                             int grad_pixel_stride, float fused_clip, float *grad_pts, void *stream)
{
    if (N <= 0 || P < 0 || S <= 0 || row0 < 0 || row1 > S || row0 >= row1 || grad_pixel_stride < 1) {
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
                       visible, rs, grad_occ, first_idx, num_pts, N, P, S, row0, row1 - row0, grad_pixel_stride, fused_clip,
                       grad_pts);
    return check_launch("dss_occ_backward");
}

extern "C" int dss_occ_backward(const float *points, const float *radii, const uint8_t *visible,
                                const float *rs, const float *grad_occ, const int64_t *first_idx,
                                const int64_t *num_pts, int N, int64_t P, int S, int row0, int row1,
                                int grad_pixel_stride, float clip, float *grad_pts, void *stream)
{
    return occ_backward_impl(points, radii, visible, rs, grad_occ, first_idx, num_pts, N, P, S, row0, row1,
                             grad_pixel_stride, clip, grad_pts, stream);
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
    return align_up((size_t)(N > 0 ? N : 1) * 4, 256) + dss_backward_radius_workspace(N, P);
}

extern "C" int dss_splat_backward(const float *points, const float *radii, const uint8_t *visible,
                                  const int32_t *idx, const float *grad_occ, const float *grad_zbuf,
                                  const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S, int K,
                                  int grad_pixel_stride, float radii_s, float clip, float *grad_pts, float *rs_out,
                                  void *workspace,
                                  size_t workspace_bytes, void *stream)
{
    if (N <= 0) { set_error("dss_splat_backward: N=%d", N); return DSS_ERR_INVALID_ARGUMENT; }
    if (!workspace || workspace_bytes < dss_splat_backward_workspace(N, P)) {
        set_error("dss_splat_backward: workspace too small");
        return DSS_ERR_WORKSPACE;
    }
    const size_t rs_bytes = align_up((size_t)N * 4, 256);
    float *rs = rs_out ? rs_out : reinterpret_cast<float *>(workspace);
    int rc = dss_backward_radius(radii, visible, first_idx, num_pts, N, P, radii_s, rs,
                                 reinterpret_cast<char *>(workspace) + rs_bytes, workspace_bytes - rs_bytes, stream);
    if (rc) return rc;
    // without a zbuf gradient the z column is 0 and the clip hook is fused into the gather kernel
    rc = occ_backward_impl(points, radii, visible, rs, grad_occ, first_idx, num_pts, N, P, S, 0, S, grad_pixel_stride,
                           grad_zbuf ? -1.0f : clip, grad_pts, stream);
    if (rc) return rc;
    if (!grad_zbuf) return DSS_OK;
    {
        if (!idx) { set_error("dss_splat_backward: grad_zbuf given without idx"); return DSS_ERR_INVALID_ARGUMENT; }
        rc = dss_zbuf_backward(idx, grad_zbuf, N, S, S, K, grad_pts, stream);
        if (rc) return rc;
    }
    return dss_clip_grad(grad_pts, P, clip, stream);
}

extern "C" size_t dss_render_backward_workspace(int N, int64_t P)
{
    const int n = N > 0 ? N : 1;
    return align_up((size_t)3 * n * MED_BINS * 4 + 256, 256)  // histograms + visible counter
           + align_up((size_t)(P > 0 ? P : 1) * 4, 256)       // compacted visible list
           + align_up((size_t)n * 4, 256);                    // rs
}

extern "C" int dss_render_backward(const float *grad_out, const int32_t *idx, const float *qvalue, const float *wsum,
                                   const float *scaler, const float *points, const float *radii,
                                   const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                   int64_t P, int S, int K, int C, int row0, int row1, float radii_s, float clip,
                                   float *grad_feat, float *grad_pts, float *rs_out, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    if (N <= 0 || P < 0 || S <= 0 || K <= 0 || C < 1 || C > BLEND_MAX_C || row0 < 0 || row1 > S || row0 >= row1) {
        set_error("dss_render_backward: bad sizes N=%d P=%lld S=%d K=%d C=%d", N, (long long)P, S, K, C);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (P > 0x7ffffff0ll) { set_error("dss_render_backward: P too large"); return DSS_ERR_UNSUPPORTED; }
    if (!grad_out || !points || !radii || !visible || !first_idx || !num_pts || !grad_pts ||
        (grad_feat && (!idx || !qvalue || !scaler))) {
        set_error("dss_render_backward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t need = dss_render_backward_workspace(N, P);
    if (!workspace || workspace_bytes < need) {
        set_error("dss_render_backward: workspace %zu bytes < required %zu", workspace_bytes, need);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    char *w = reinterpret_cast<char *>(workspace);
    const size_t hist_bytes = (size_t)3 * N * MED_BINS * 4;
    uint32_t *hist = reinterpret_cast<uint32_t *>(w);
    uint32_t *vis_count = reinterpret_cast<uint32_t *>(w + hist_bytes);
    size_t off = align_up(hist_bytes + 256, 256);
    int32_t *vis_list = reinterpret_cast<int32_t *>(w + off);
    off += align_up((size_t)P * 4, 256);
    float *rs = rs_out ? rs_out : reinterpret_cast<float *>(w + off);
    if (hipMemsetAsync(hist, 0, hist_bytes + 256, st) != hipSuccess) return check_launch("memset render_backward");
    const unsigned blocks = (unsigned)((P + MED_PTS_PER_WG - 1) / MED_PTS_PER_WG);
    hipLaunchKernelGGL(visible_scan_kernel, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx, num_pts,
                       N, P, hist, vis_count, vis_list, grad_pts, grad_feat, C);
    hipLaunchKernelGGL(median_hist_kernel<1>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                       num_pts, N, P, hist);
    hipLaunchKernelGGL(median_hist_kernel<2>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                       num_pts, N, P, hist);
    hipLaunchKernelGGL(median_final_kernel, dim3(N), dim3(MED_THREADS), 0, st, hist, N, radii_s, rs);
    // persistent grid = exactly the resident capacity of the chip for this kernel (a larger grid would
    // leave late workgroups waiting for slots while their share of the list sits idle)
    static int cap3 = 0, cap0 = 0;  // benign race: every thread computes the same value
    int &cap = (C == 3) ? cap3 : cap0;
    if (cap == 0) {
        int dev = 0, cus = 256, per_cu = 4;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (C == 3)
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<3>, 256, 0);
        else
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<0>, 256, 0);
        if (per_cu < 1) per_cu = 1;
        cap = cus * per_cu;
        (void)hipGetLastError();
    }
    const unsigned pgrid = (unsigned)((P + 3) / 4 < cap ? (P + 3) / 4 : cap);
    if (C == 3)
        hipLaunchKernelGGL(render_backward_kernel<3>, dim3(pgrid), dim3(256), 0, st, grad_out, idx, qvalue, wsum, scaler,
                           points, radii, rs, first_idx, num_pts, vis_count, vis_list, N, S, K, C, clip, row0, row1 - row0,
                           grad_feat, grad_pts);
    else
        hipLaunchKernelGGL(render_backward_kernel<0>, dim3(pgrid), dim3(256), 0, st, grad_out, idx, qvalue, wsum, scaler,
                           points, radii, rs, first_idx, num_pts, vis_count, vis_list, N, S, K, C, clip, row0, row1 - row0,
                           grad_feat, grad_pts);
    return check_launch("dss_render_backward");
}

#ifdef DSS_FINE_TIMING
extern "C" __attribute__((visibility("default"))) int dss_debug_set_occ_timing(long long *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(dss::g_occ_timing), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
