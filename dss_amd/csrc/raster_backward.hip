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
#include <stdlib.h>
#include <type_traits>
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
// `DSS._C._splat_points_occ_backward` on CUDA tensors (ext.cpp:10,16; RasterizePointsOccBackwardCudaKernel,
// rasterize_points.cu:672-757): the reference's older, box-supported occupancy surrogate.  NOT on the training
// path (`backward_occ_fast = True`, rasterizer.py:816) -- kept as a same-name mirror, so a plain gather:
// one wavefront per point over the pixel window |dx| <= rx*s, |dy| <= ry*s.
//   skip if pz<0 or |px|>1 or |py|>1;  R = radii*radii_s;  skip if |dx|>Rx or |dy|>Ry;
//   skip if g>0 and (|dx| > Rx/radii_s or |dy| > Ry/radii_s);  grad += (dx,dy)/max(d2,1e-10)*g   (d2 == 0: 0)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void occ_box_backward_kernel(
    const float *__restrict__ points, const float *__restrict__ radii, const float *__restrict__ grad_occ,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t P, int S, float radii_s,
    float *__restrict__ grad_xy)
{
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    float gx = 0.0f, gy = 0.0f;
    const int n = find_cloud(p, first_idx, num_pts, N);
    const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
    const float Rx = radii[2 * p] * radii_s, Ry = radii[2 * p + 1] * radii_s;
    int xlo, xhi, ylo, yhi;
    if (n >= 0 && !(pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f) && ndc_index_range(px, Rx, S, xlo, xhi) &&
        ndc_index_range(py, Ry, S, ylo, yhi)) {
        const int w = xhi - xlo + 1;
        const int total = w * (yhi - ylo + 1);
        const float rx1 = Rx / radii_s, ry1 = Ry / radii_s;   // (rasterize_points.cu:741: scaled back, as the reference does)
        for (int t = lane; t < total; t += 64) {
            const int yi = ylo + t / w, xi = xlo + t % w;
            const float g = grad_occ[((size_t)n * S + (S - 1 - yi)) * S + (S - 1 - xi)];
            if (g == 0.0f) continue;
            const float dx = pix_to_ndc(xi, S) - px, dy = pix_to_ndc(yi, S) - py;
            if (fabsf(dx) > Rx || fabsf(dy) > Ry) continue;
            if (g > 0.0f && (fabsf(dx) > rx1 || fabsf(dy) > ry1)) continue;
            const float d2 = dx * dx + dy * dy;
            if (d2 == 0.0f) continue;
            const float den = fmaxf(d2, 1e-10f);
            gx += dx / den * g;
            gy += dy / den * g;
        }
    }
    gx = wave_sum(gx);
    gy = wave_sum(gy);
    if (lane == 0) {
        grad_xy[2 * p] = gx;
        grad_xy[2 * p + 1] = gy;
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
    float *__restrict__ grad_pts, float *__restrict__ grad_feat, int C,
    int zero_all = 0 /* row band: also the visible points start from zero (contiguous stores here; the band filter of
                        cell_hist_kernel then drops a point without the scattered zero stores it issued before -- 60 us of its
                        92 us on a rank of configs[4]) */)
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
        if (vis[u]) vis_list[base + rank[u]] = (int32_t)i;
        if ((!vis[u] || zero_all) && i < c1) {
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

// ---------------------------------------------------------------------------------------------
// Small-cloud preparation (P <= PREP_MAX_POINTS): TWO launches, no memset, no global atomics, replace
// memset + visible_scan + median_hist<1> + median_hist<2> + median_final (five launches of pure latency,
// ~26 us for ~300 KB of traffic at DSS sizes).
//   backward_compact_kernel  one 1024-thread workgroup per 2048-point chunk of the packed array: compacts the
//                            chunk's visible ids AND radius keys into the chunk's own segment
//                            (vis_list / vis_keys [chunk * 2048 + rank], seg_count[chunk]), zero-fills the
//                            gradients of invisible points and writes the chunk's pass-0 histogram (8-bit
//                            digit [31:24], per overlapping cloud) with plain stores.
//   median_visible_kernel    one 1024-thread workgroup per cloud: sums the chunk histograms -> digit 0, then
//                            selects digits [23:12] and [11:0] over the COMPACTED keys held in registers
//                            (up to 32 visible points per thread; larger clouds are re-streamed per pass).
// One CU scanning every packed radius three times (a single-launch variant, measured) took 23-29 us: the
// wave-instruction issue rate of ONE CU, not memory, is the limit, hence the split: all-points work on many
// CUs, single-CU work only on the visible ~40 %.
// Pass-0 histograms are replicated 32x by lane (address = bin * 32 + lane % 32): screen-space radii share one
// or two exponent bytes and same-address LDS atomics serialise.
// ---------------------------------------------------------------------------------------------
#define PREP_THREADS 1024
#define PREP_MAX_SEG 64          // segments a wavefront can scan in registers
#define PREP_MAX_PER 4           // points per thread of the compaction kernel: 2 or 4 -> segments of 2048 / 4096 points
                                 // (8 per thread needs > 128 VGPRs at 1024 threads: spills)
#define PREP_MAX_POINTS (PREP_MAX_PER * PREP_THREADS * PREP_MAX_SEG)   // 262,144
// segment size for P points: the smallest of 2048 / 4096 that needs at most PREP_MAX_SEG segments
static inline int prep_points_per_thread(int64_t P)
{
    int per = 2;
    while (per < PREP_MAX_PER && (int64_t)per * PREP_THREADS * PREP_MAX_SEG < P) per *= 2;
    return per;
}
#define PREP_RES 32              // visible points per thread resident in registers in the median workgroup

#ifdef DSS_FINE_TIMING
#define PREP_MARK(slot)                                                                                       \
    do {                                                                                                      \
        if (g_occ_timing && threadIdx.x == 0)                                                                 \
            g_occ_timing[(size_t)blockIdx.x * 12 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime();     \
    } while (0)
#else
#define PREP_MARK(slot)
#endif

// Wave-wide inclusive scan on the VALU (DPP row shifts + row broadcasts; `__shfl_up` would be six dependent
// ds_bpermute round trips).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += dpp_u32_zero<0x111, 0xf>(v);  // row_shr:1
    v += dpp_u32_zero<0x112, 0xf>(v);  // row_shr:2
    v += dpp_u32_zero<0x114, 0xf>(v);  // row_shr:4
    v += dpp_u32_zero<0x118, 0xf>(v);  // row_shr:8  -> inclusive scan inside each row of 16
    v += dpp_u32_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_u32_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// Sum of the 32 lane-replicas of bin (tid >> 2), valid in the lanes with (tid & 3) == 0 (others return 0).
__device__ __forceinline__ uint32_t replica_sum_256(const uint32_t *lh)
{
    int b = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    asm volatile("" : "+v"(b));  // keep the eight LDS addresses out of the long-lived register set
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v += lh[b * 32 + ((q * 8 + i + b) & 31)];  // rotated by the bin: <= 2-way conflicts
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    return q == 0 ? v : 0u;
}

// Rank-k selection by a 1024-thread workgroup: thread owns bins first_bin .. first_bin + 3 with counts h[].
__device__ __forceinline__ void block_select(const uint32_t (&h)[4], uint32_t first_bin, uint32_t k,
                                             bool k_is_lower_median, uint32_t *s_w /*[16]*/, uint32_t *s_sel /*[2]*/,
                                             uint32_t &bin_out, uint32_t &k_out, uint32_t &total_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t v = (h[0] + h[1]) + (h[2] + h[3]);
    const uint32_t x = wave_incl_scan(v);
    __syncthreads();  // s_w / s_sel reuse
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < PREP_THREADS / 64; ++w) {
        const uint32_t c = s_w[w];
        if (w < wid) woff += c;
        total += c;
    }
    if (k_is_lower_median) k = (total > 0) ? (total - 1) / 2 : 0;  // torch.median = lower median
    uint32_t excl = woff + x - v;
    if (total > 0 && k >= excl && k < excl + v) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (k >= excl && k < excl + h[i]) {
                s_sel[0] = first_bin + i;
                s_sel[1] = k - excl;
            }
            excl += h[i];
        }
    }
    __syncthreads();
    bin_out = s_sel[0];
    k_out = s_sel[1];
    total_out = total;
}

// Dense copy of the alpha channel of the image gradient: plane[i] = grad_out[i * (C + 1) + C].
#define ALPHA_PIX_PER_WG 8192
__device__ __forceinline__ void alpha_plane_body(unsigned block, unsigned threads, const float *__restrict__ grad_out,
                                                 float *__restrict__ plane, size_t npix, int C)
{
    const size_t i0 = (size_t)block * ALPHA_PIX_PER_WG;
    const size_t i1 = min(i0 + ALPHA_PIX_PER_WG, npix);
    if (C == 3) {
        const float4 *g4 = reinterpret_cast<const float4 *>(grad_out);  // (N,rows,S,4): 16-byte aligned pixels
        for (size_t i = i0 + threadIdx.x; i < i1; i += threads) plane[i] = g4[i].w;
    } else {
        for (size_t i = i0 + threadIdx.x; i < i1; i += threads) plane[i] = grad_out[i * (C + 1) + C];
    }
}

__global__ __launch_bounds__(1024) void alpha_plane_kernel(const float *__restrict__ grad_out, float *__restrict__ plane,
                                                           size_t npix, int C)
{
    alpha_plane_body(blockIdx.x, blockDim.x, grad_out, plane, npix, C);
}

template <int PER>
__global__ __launch_bounds__(PREP_THREADS) void backward_compact_kernel(
    const float *__restrict__ radii, const uint8_t *__restrict__ visible, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, int N, int64_t P, int chunks, uint32_t *__restrict__ seg_count,
    int32_t *__restrict__ vis_list, uint2 *__restrict__ vis_keys, uint32_t *__restrict__ chunk_hist /*(N,chunks,256)*/,
    uint2 *__restrict__ seg_range /*(N,chunks)*/, float *__restrict__ grad_pts, float *__restrict__ grad_feat, int C,
    const float *__restrict__ grad_out, float *__restrict__ alpha_plane, size_t npix)
{
    if ((int)blockIdx.x >= chunks) {  // extra workgroups of the launch: dense alpha plane for the gather kernel
        alpha_plane_body(blockIdx.x - chunks, PREP_THREADS, grad_out, alpha_plane, npix, C);
        return;
    }
    constexpr int PREP_CHUNK = PER * PREP_THREADS;
    __shared__ uint32_t lh[256 * 32];
    __shared__ uint32_t s_w[PER * PREP_THREADS / 64];
    __shared__ uint32_t s_rng[2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned chunk = blockIdx.x;
    const int64_t c0 = (int64_t)chunk * PREP_CHUNK;
    const int64_t c1 = min(c0 + PREP_CHUNK, P);
    PREP_MARK(0);
    uint32_t vmask = 0;  // bit u: point c0 + tid + u * 1024 is visible
    uint2 kk[PER];
    {
        // unconditional loads from clamped addresses (a bounds branch around each load made the compiler wait
        // for every load in turn and spill), all in flight together
        uint8_t vb[PER];
        float2 rr[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t ic = min(c0 + tid + (int64_t)u * PREP_THREADS, P - 1);
            vb[u] = visible[ic];
            rr[u] = reinterpret_cast<const float2 *>(radii)[ic];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const bool in = c0 + tid + (int64_t)u * PREP_THREADS < c1;
            vmask |= ((in && vb[u] != 0) ? 1u : 0u) << u;
            kk[u] = make_uint2(float_key(rr[u].x), float_key(rr[u].y));
        }
    }
    for (int i = tid; i < 256 * 32; i += PREP_THREADS) lh[i] = 0;
    // order-preserving ranks (entries of a segment are sorted by point id, so the entries of one cloud are a
    // contiguous range of it): rank = visible points before (u, wave, lane) in that order
    uint32_t rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const unsigned long long m = __ballot((vmask >> u) & 1u);
        rank[u] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (lane == 0) s_w[u * (PREP_THREADS / 64) + wid] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        uint32_t before = tot;
#pragma unroll
        for (int w = 0; w < PREP_THREADS / 64; ++w) {
            const uint32_t c = s_w[u * (PREP_THREADS / 64) + w];
            if (w < wid) before += c;
            tot += c;
        }
        rank[u] += before;
    }
    if (tid == 0) seg_count[chunk] = tot;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t i = c0 + tid + (int64_t)u * PREP_THREADS;
        if ((vmask >> u) & 1u) {
            vis_list[c0 + rank[u]] = (int32_t)i;
            vis_keys[c0 + rank[u]] = kk[u];
        } else if (i < c1 && grad_pts) {
            grad_pts[3 * i] = 0.0f; grad_pts[3 * i + 1] = 0.0f; grad_pts[3 * i + 2] = 0.0f;
            if (grad_feat)
                for (int ch = 0; ch < C; ++ch) grad_feat[(size_t)i * C + ch] = 0.0f;
        }
    }
    // pass-0 histogram of this chunk, one per cloud that overlaps it (normally one)
    for (int n = 0; n < N; ++n) {
        const int64_t lo = max(c0, first_idx[n]), hi = min(c1, first_idx[n] + num_pts[n]);
        if (lo >= hi) continue;  // uniform
        if (tid < 2) s_rng[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int64_t i = c0 + tid + (int64_t)u * PREP_THREADS;
            const bool vis_u = (vmask >> u) & 1u;
            const bool mine = vis_u && i >= lo && i < hi;
            const unsigned long long mb = __ballot(vis_u && i < lo), mi = __ballot(mine);
            if (lane == 0) {
                if (mb) atomicAdd(&s_rng[0], (uint32_t)__popcll(mb));
                if (mi) atomicAdd(&s_rng[1], (uint32_t)__popcll(mi));
            }
            if (mine) {
                atomicAdd(&lh[(kk[u].x >> 24) * 32 + (lane & 31)], 1u);
                atomicAdd(&lh[(kk[u].y >> 24) * 32 + (lane & 31)], 1u);
            }
        }
        __syncthreads();
        // where cloud n's entries sit inside this chunk's segment: (first entry, count)
        if (tid == 0) seg_range[(size_t)n * chunks + chunk] = make_uint2(s_rng[0], s_rng[1]);
        const uint32_t v = replica_sum_256(lh);
        if ((tid & 3) == 0) chunk_hist[((size_t)n * chunks + chunk) * 256 + (tid >> 2)] = v;
        __syncthreads();
        for (int i = tid; i < 256 * 32; i += PREP_THREADS) lh[i] = 0;
        __syncthreads();
    }
    PREP_MARK(1);
}

__global__ __launch_bounds__(PREP_THREADS) void median_visible_kernel(
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int64_t P, int chunks, int seg_pts,
    const uint2 *__restrict__ seg_range, const uint2 *__restrict__ vis_keys, const uint32_t *__restrict__ chunk_hist,
    float radii_s, float *__restrict__ rs)
{
    __shared__ uint32_t lh[4096];
    __shared__ uint32_t s_w[PREP_THREADS / 64];
    __shared__ uint32_t s_sel[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = blockIdx.x;
    const int64_t f = first_idx[n];
    const int64_t cnt = max((int64_t)0, min(num_pts[n], P - f));
    PREP_MARK(6);
    if (cnt <= 0) {
        if (tid == 0) rs[n] = 0.0f;
        return;
    }
    const int c_lo = (int)(f / seg_pts), c_hi = (int)((f + cnt - 1) / seg_pts);
    const int n_c = c_hi - c_lo + 1;  // <= PREP_MAX_SEG
    // ---- all global loads are issued up front: chunk histograms, segment ranges, then the keys ----
    const int hb = tid >> 2, hq = tid & 3;
    uint32_t hv = 0;
    for (int c = c_lo + hq; c <= c_hi; c += 4) hv += chunk_hist[((size_t)n * chunks + c) * 256 + hb];
    // entries of cloud n, flattened in units of WAVE SLOTS (64 consecutive entries of one segment): slot g ->
    // segment with one ballot over the per-lane slot prefix (every wave keeps the table in its lanes)
    const uint2 rg = lane < n_c ? seg_range[(size_t)n * chunks + c_lo + lane] : make_uint2(0u, 0u);
    const uint32_t seg_start = rg.x, seg_cnt = rg.y;
    const uint32_t seg_slots = (seg_cnt + 63u) >> 6;
    const uint32_t slot_incl = wave_incl_scan(seg_slots);
    const uint32_t slot_excl = slot_incl - seg_slots;
    const uint32_t n_slots = (uint32_t)__builtin_amdgcn_readlane((int)slot_incl, 63);
    constexpr uint32_t WAVES = PREP_THREADS / 64;
    constexpr uint32_t RES_SLOTS = (uint32_t)PREP_RES * WAVES;  // wave slots resident in registers per round
    const uint32_t wid = (uint32_t)tid >> 6;
    uint2 kk[PREP_RES];
    uint32_t vm = 0;
    uint32_t resident = 0;
    auto load_round = [&](uint32_t base) {  // up to 512 wave slots (~32768 visible points)
        const uint32_t rem = min(n_slots - base, RES_SLOTS);
        vm = 0;
#pragma unroll
        for (int u = 0; u < PREP_RES; ++u) {
            kk[u] = make_uint2(0u, 0u);
            if ((uint32_t)u * WAVES < rem) {  // uniform
                const uint32_t g = base + (uint32_t)u * WAVES + wid;  // wave-uniform
                if (g < n_slots) {
                    const int seg = (int)__popcll(__ballot(slot_incl <= g));  // first segment with incl > g
                    const uint32_t off = (g - (uint32_t)__builtin_amdgcn_readlane((int)slot_excl, seg)) * 64u + lane;
                    if (off < (uint32_t)__builtin_amdgcn_readlane((int)seg_cnt, seg)) {
                        kk[u] = vis_keys[(size_t)(c_lo + seg) * seg_pts +
                                         (uint32_t)__builtin_amdgcn_readlane((int)seg_start, seg) + off];
                        vm |= 1u << u;
                    }
                }
            }
        }
    };
    if (n_slots > 0) load_round(0);
    // ---- digit 0: sum of the chunk histograms (4 partial sums per bin, quad reduce) ----
    uint32_t prefix, pmask, k, total0;
    {
        hv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv, 0xB1, 0xf, 0xf, true);
        hv += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv, 0x4E, 0xf, 0xf, true);
        const uint32_t h[4] = {hq == 0 ? hv : 0u, 0u, 0u, 0u};
        uint32_t bin;
        block_select(h, (uint32_t)hb, 0u, true, s_w, s_sel, bin, k, total0);
        prefix = bin << 24;
        pmask = 0xff000000u;
    }
    PREP_MARK(7);
    if (total0 == 0) {  // nothing visible in this cloud
        if (tid == 0) rs[n] = 0.0f;
        return;
    }
#pragma unroll
    for (int pass = 1; pass < 3; ++pass) {
        for (int i = tid; i < 4096; i += PREP_THREADS) lh[i] = 0;
        __syncthreads();
        const int sh = pass == 1 ? 12 : 0;
        for (uint32_t base = 0; base < n_slots; base += RES_SLOTS) {
            const uint32_t rem = min(n_slots - base, RES_SLOTS);
            if (resident != base) {  // uniform; only clouds with more than ~32768 visible points re-stream
                load_round(base);
                resident = base;
            }
#pragma unroll
            for (int u = 0; u < PREP_RES; ++u) {
                if ((uint32_t)u * WAVES < rem) {  // uniform
                    const uint32_t kx = kk[u].x, ky = kk[u].y;
                    const bool v = (vm >> u) & 1u;
                    const bool mx = v && (kx & pmask) == prefix, my = v && (ky & pmask) == prefix;
                    if (mx || my) {
                        if (mx) atomicAdd(&lh[(kx >> sh) & 0xfffu], 1u);
                        if (my) atomicAdd(&lh[(ky >> sh) & 0xfffu], 1u);
                    }
                }
            }
        }
        __syncthreads();
        PREP_MARK(6 + 2 * pass);
        int t4 = tid;
        asm volatile("" : "+v"(t4));
        const uint4 q = reinterpret_cast<const uint4 *>(lh)[t4];
        const uint32_t h[4] = {q.x, q.y, q.z, q.w};
        uint32_t bin, tot;
        block_select(h, 4u * tid, k, false, s_w, s_sel, bin, k, tot);
        prefix |= bin << sh;
        pmask |= 0xfffu << sh;
        __syncthreads();  // lh is re-zeroed by the next pass
        PREP_MARK(7 + 2 * pass);
    }
    if (tid == 0) rs[n] = key_float(prefix) * radii_s;
}

// Band-local row <-> image row of a (possibly tile-row-cyclic) band, see TileGrid::tshift in raster_forward.hip:
// image row of band row l = row0 + ((l >> 3) << tshift) + (l & 7); contiguous band: tshift = 3, i.e. row0 + l.
__device__ __forceinline__ int band_image_row(int l, int row0, int tshift) { return row0 + ((l >> 3) << tshift) + (l & 7); }
// smallest band row whose image row is >= r (may equal `rows`: none)
__device__ __forceinline__ int band_row_ceil(int r, int row0, int tshift)
{
    const int x = r - row0;
    if (x <= 0) return 0;
    const int q = x >> tshift, m = x & ((1 << tshift) - 1);
    return m < 8 ? 8 * q + m : 8 * (q + 1);
}
// largest band row whose image row is <= r (-1: none)
__device__ __forceinline__ int band_row_floor(int r, int row0, int tshift)
{
    const int x = r - row0;
    if (x < 0) return -1;
    const int q = x >> tshift, m = x & ((1 << tshift) - 1);
    return 8 * q + min(m, 7);
}

// is image row r one of the band's rows?
__device__ __forceinline__ bool band_owns_row(int r, int row0, int rows, int tshift)
{
    const int x = r - row0;
    if (x < 0) return false;
    const int m = x & ((1 << tshift) - 1);
    return m < 8 && 8 * (x >> tshift) + m < rows;
}
// OWNER mode of a row band (dss_render_backward_owned): the occupancy surrogate of a point -- its whole window, over all image
// rows, read from the FULL image gradient every rank holds -- is computed by the one rank whose band contains the image row
// of the point's centre; the blend half stays with the rows that hold the fragments.  The position gradient of a (camera,
// point) pair is then complete on its owner (the other ranks store zeros): the non-linear per-point clip can run before the
// reduction over the ranks, which shrinks from a dense (N P, 6) bucket to the world-space (P_cloud, 6) sums.
struct OwnArgs {
    const float *alpha;   // alpha channel of the full image gradient, (N, S, S) pixels `astride` floats apart; NULL: off
    int astride;
};
__device__ __forceinline__ int centre_image_row(float py, int S)
{
    const float fy = (1.0f - py) * 0.5f * (float)S;   // pixel row of the centre (any fixed partition of the axis does)
    return min(max((int)fy, 0), S - 1);
}

// Owner mode, long lists: the dense alpha plane in FULL-image layout (N,S,S), filled only where an owned window can reach
// -- the rows within the largest search radius (read from rs on the device: the medians run in front of this launch) of one
// of the band's rows.  (Reading the alpha channel in place, 16 bytes apart, costs the gather four times the cache lines.)
__global__ __launch_bounds__(256) void alpha_rows_kernel(const float *__restrict__ grad_full, float *__restrict__ plane,
                                                         const float *__restrict__ rs, int N, int S, int C, int row0, int rows,
                                                         int tshift)
{
    const int r = blockIdx.x, n = blockIdx.y;
    float rmax = 0.0f;
    for (int c = 0; c < N; ++c) rmax = fmaxf(rmax, rs[c]);
    const int H = (int)ceilf(fminf(rmax, 4.0f) * 0.5f * (float)S) + 2;   // pixels (NDC spans 2 over S pixels)
    const int lf = band_row_floor(r, row0, tshift), lc = band_row_ceil(r, row0, tshift);
    bool need = false;
    if (lf >= 0 && r - band_image_row(min(lf, rows - 1), row0, tshift) <= H) need = true;
    if (lc < rows && band_image_row(lc, row0, tshift) - r <= H) need = true;
    if (!need) return;
    const size_t base = ((size_t)n * S + r) * S;
    for (int c = threadIdx.x; c < S; c += 256) plane[base + c] = grad_full[(base + c) * (size_t)(C + 1) + C];
}

// Row band (multi-GPU), after the median: drop the visible points that cannot reach this rank's rows from the
// segment lists, in place (one workgroup per segment: all entries are read into registers before any is
// written), and zero their gradient rows (this band's partial sum for them is zero).  Same conservative test
// as the gather kernel.  Without it every rank walks the whole visible list: 35 us of rejected tasks per step
// at 8 ranks even for a rank whose band is empty.
template <int PER>
__global__ __launch_bounds__(PREP_THREADS) void band_filter_kernel(
    const float *__restrict__ points, const float *__restrict__ radii, const float *__restrict__ rs,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int S, int row0, int rows,
    uint32_t *__restrict__ seg_count, int32_t *__restrict__ vis_list, float *__restrict__ grad_pts,
    float *__restrict__ grad_feat, int C, int tshift, int own = 0 /* owner mode: only the splat's own box must meet the band */)
{
    __shared__ uint32_t s_w[PER * PREP_THREADS / 64];
    constexpr int SEG_PTS = PER * PREP_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned seg = blockIdx.x;
    const uint32_t count = seg_count[seg];
    // (tile-row-cyclic band: the last band row's image row bounds the band from below; the owned-row test follows)
    const int last_row = row0 + (((rows - 1) >> 3) << tshift) + ((rows - 1) & 7);
    const float band_lo = -1 + (2 * (S - 1 - last_row)) / (float)S;     // lower edge of the lowest pixel row
    const float band_hi = -1 + (2 * (S - 1 - row0) + 2.0f) / (float)S;  // upper edge of the highest one
    int32_t id[PER];
    uint32_t keep = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const uint32_t e = (uint32_t)tid + (uint32_t)u * PREP_THREADS;
        id[u] = e < count ? vis_list[(size_t)seg * SEG_PTS + e] : -1;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (id[u] >= 0) {
            const int n = find_cloud(id[u], first_idx, num_pts, N);
            const float py = points[3 * (size_t)id[u] + 1], ry = radii[2 * (size_t)id[u] + 1];
            const float reach = own ? ry : fmaxf(n >= 0 ? rs[n] : 0.0f, ry);
            bool in_band = n >= 0 && !(py + reach < band_lo || py - reach > band_hi);
            if (in_band && tshift > 3) {
                // cyclic band: some OWNED row must lie within reach (conservative pixel range, one pixel of slack each side)
                int ylo, yhi;
                in_band = ndc_index_range(py, reach, S, ylo, yhi) &&
                          band_row_ceil(S - 1 - yhi, row0, tshift) <= min(band_row_floor(S - 1 - ylo, row0, tshift), rows - 1);
            }
            keep |= (in_band ? 1u : 0u) << u;
        }
    }
    uint32_t rank[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const unsigned long long m = __ballot((keep >> u) & 1u);
        rank[u] = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (lane == 0) s_w[u * (PREP_THREADS / 64) + wid] = (uint32_t)__popcll(m);
    }
    __syncthreads();  // every entry of the segment has been read
    uint32_t tot = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        uint32_t before = tot;
#pragma unroll
        for (int w = 0; w < PREP_THREADS / 64; ++w) {
            const uint32_t c = s_w[u * (PREP_THREADS / 64) + w];
            if (w < wid) before += c;
            tot += c;
        }
        rank[u] += before;
    }
    if (tid == 0) seg_count[seg] = tot;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (id[u] < 0) continue;
        if ((keep >> u) & 1u) {
            vis_list[(size_t)seg * SEG_PTS + rank[u]] = id[u];
        } else {
            const size_t i = (size_t)id[u];
            grad_pts[3 * i] = 0.0f; grad_pts[3 * i + 1] = 0.0f; grad_pts[3 * i + 2] = 0.0f;
            if (grad_feat)
                for (int ch = 0; ch < C; ++ch) grad_feat[i * C + ch] = 0.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Screen-cell order of the visible list (long lists only).  The gather of a visible point reads the image-gradient
// window around it (~20 x 20 pixels) and the fragments of its own box; in point-id order (random on screen) every task
// pulls its rows from HBM again: 14 GB of fetches per launch at 8 x 1M points @1024^2 for 0.7 GB of distinct data --
// the kernel was HBM-bound on re-reads.  A counting sort by (camera, 32 x 32-pixel cell) puts tasks that share rows next
// to each other, and the gather deals whole runs of the sorted list to the same XCD (one L2).  The order of the tasks
// does not change any result (every task writes only its own point).
// ---------------------------------------------------------------------------------------------
struct CellGrid {
    int shift;      // cell side = 1 << shift pixels
    int cx, cy;     // cells per image row / column
    int total;      // N * cx * cy  (<= CELL_MAX: one LDS histogram)
};
#define CELL_MAX 16384          // 64 KB of LDS counters
#define CELL_THREADS 1024
#define CELL_PER_THREAD 16      // list entries per thread: a workgroup sorts 16384 entries
#define CELL_CHUNK (CELL_THREADS * CELL_PER_THREAD)
static CellGrid make_cells(int N, int S)
{
    CellGrid c;
    c.shift = 5;   // 32 x 32-pixel cells (16 measures the same, 64 is 4 % slower)
    for (;;) {
        c.cx = ((S - 1) >> c.shift) + 1;
        c.cy = c.cx;
        c.total = N * c.cx * c.cy;
        if (c.total <= CELL_MAX || c.shift >= 14) break;
        ++c.shift;
    }
    return c;
}
static inline size_t cell_blocks(int64_t P) { return (size_t)((P + CELL_CHUNK - 1) / CELL_CHUNK); }

// A counting sort without global atomics (they run at ~16 G/s on this part: 2 x 2.4M of them cost 0.45 ms, more than the
// sort saves): pass 1, per workgroup a histogram of its 16384 list entries in LDS -> block_hist[block][cell]; pass 2, a
// thread per cell walks the active blocks (running sum = each block's first position within the cell) and one workgroup
// scans the cell totals; pass 3, every workgroup ranks its entries again in LDS and writes them to
// cell_start[cell] + block_base[block][cell] + rank.
// Row band (multi-GPU, round 5): rs is known by now (median_final_kernel precedes this launch), so the entries that cannot
// reach the rank's rows -- neither with their own box nor with the search window; the conservative test of
// band_filter_kernel -- get no cell: they drop out of the sorted list the gather walks (at 8 ranks seven eighths of the
// ~3.2M tasks of 8 x 1M points were rejected one by one inside the gather: 0.6 ms per rank and step for 0.16 ms of work) and
// their zero partial sums are stored here.
#define CELL_NONE 0xffffffffu
__global__ __launch_bounds__(CELL_THREADS) void cell_hist_kernel(
    const float *__restrict__ points, const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int S,
    CellGrid cg, const uint32_t *__restrict__ vis_count, const int32_t *__restrict__ vis_list,
    uint32_t *__restrict__ cell_of, uint32_t *__restrict__ block_hist,
    const float *__restrict__ radii /* NULL: whole image, no filter */, const float *__restrict__ rs, int row0, int rows,
    int tshift, int own = 0)
{
    extern __shared__ uint32_t s_hist[];
    const uint32_t count = *vis_count;
    const uint32_t b0 = blockIdx.x * (uint32_t)CELL_CHUNK;
    if (b0 >= count) return;   // (inactive blocks: pass 2 never reads their rows)
    for (int c = threadIdx.x; c < cg.total; c += CELL_THREADS) s_hist[c] = 0u;
    __syncthreads();
#pragma unroll 4
    for (int u = 0; u < CELL_PER_THREAD; ++u) {
        const uint32_t i = b0 + (uint32_t)u * CELL_THREADS + threadIdx.x;
        if (i < count) {
            const int32_t p = vis_list[i];
            const int n = max(find_cloud(p, first_idx, num_pts, N), 0);
            const float px = points[3 * (size_t)p], py = points[3 * (size_t)p + 1];
            // pixel column / row of the point (any monotone map of NDC does: only neighbourhood matters)
            const float fx = (1.0f - px) * 0.5f * (float)S, fy = (1.0f - py) * 0.5f * (float)S;
            const int ix = min(max((int)fx, 0), S - 1) >> cg.shift, iy = min(max((int)fy, 0), S - 1) >> cg.shift;
            uint32_t key = (uint32_t)((n * cg.cy + iy) * cg.cx + ix);
            if (radii) {
                const int last_row = row0 + (((rows - 1) >> 3) << tshift) + ((rows - 1) & 7);
                const float band_lo = -1 + (2 * (S - 1 - last_row)) / (float)S;     // lower edge of the lowest pixel row
                const float band_hi = -1 + (2 * (S - 1 - row0) + 2.0f) / (float)S;  // upper edge of the highest one
                const float reach = own ? radii[2 * (size_t)p + 1] : fmaxf(rs[n], radii[2 * (size_t)p + 1]);
                bool in_band = !(py + reach < band_lo || py - reach > band_hi);
                if (in_band && tshift > 3) {
                    int ylo, yhi;
                    in_band = ndc_index_range(py, reach, S, ylo, yhi) &&
                              band_row_ceil(S - 1 - yhi, row0, tshift) <= min(band_row_floor(S - 1 - ylo, row0, tshift), rows - 1);
                }
                if (!in_band) key = CELL_NONE;   // (its partial sums over this band are zero: visible_scan_kernel stored them)
            }
            cell_of[i] = key;
            if (key != CELL_NONE) atomicAdd(&s_hist[key], 1u);
        }
    }
    __syncthreads();
    uint32_t *row = block_hist + (size_t)blockIdx.x * cg.total;
    for (int c = threadIdx.x; c < cg.total; c += CELL_THREADS) row[c] = s_hist[c];
}
// thread per cell: block_hist[b][c] <- number of the cell's entries in blocks < b; cell_total[c]
__global__ __launch_bounds__(256) void cell_block_scan_kernel(const uint32_t *__restrict__ vis_count, int total,
                                                              uint32_t *__restrict__ block_hist,
                                                              uint32_t *__restrict__ cell_total)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    const uint32_t nb = (*vis_count + (uint32_t)CELL_CHUNK - 1u) / (uint32_t)CELL_CHUNK;
    uint32_t run = 0;
    uint32_t b = 0;
    // eight independent loads in flight per trip (one at a time the walk over ~150 blocks was 40 us of load latencies)
    for (; b + 8 <= nb; b += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = block_hist[(size_t)(b + u) * total + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            block_hist[(size_t)(b + u) * total + c] = run;
            run += v[u];
        }
    }
    for (; b < nb; ++b) {
        const uint32_t v = block_hist[(size_t)b * total + c];
        block_hist[(size_t)b * total + c] = run;
        run += v;
    }
    cell_total[c] = run;
}
// one workgroup: exclusive scan of the cell totals -> cell_start
__global__ __launch_bounds__(1024) void cell_scan_kernel(const uint32_t *__restrict__ cell_total, uint32_t *__restrict__ cell_start,
                                                         int total, uint32_t *__restrict__ sorted_count /* entries of the sorted list */)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (total + 1023) / 1024;
    const int b = tid * per, e = min(b + per, total);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += cell_total[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (int i = b; i < e; ++i) {
        cell_start[i] = run;
        run += cell_total[i];
    }
    if (tid == 1023) *sorted_count = s_part[1023];
}
__global__ __launch_bounds__(CELL_THREADS) void cell_scatter_kernel(
    CellGrid cg, const uint32_t *__restrict__ vis_count, const int32_t *__restrict__ vis_list,
    const uint32_t *__restrict__ cell_of, const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ block_base,
    int32_t *__restrict__ sorted)
{
    extern __shared__ uint32_t s_hist[];
    const uint32_t count = *vis_count;
    const uint32_t b0 = blockIdx.x * (uint32_t)CELL_CHUNK;
    if (b0 >= count) return;
    const uint32_t *base = block_base + (size_t)blockIdx.x * cg.total;
    for (int c = threadIdx.x; c < cg.total; c += CELL_THREADS) s_hist[c] = cell_start[c] + base[c];
    __syncthreads();
#pragma unroll 4
    for (int u = 0; u < CELL_PER_THREAD; ++u) {
        const uint32_t i = b0 + (uint32_t)u * CELL_THREADS + threadIdx.x;
        if (i < count) {
            const uint32_t key = cell_of[i];
            if (key != CELL_NONE) sorted[atomicAdd(&s_hist[key], 1u)] = vis_list[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused backward gather, TPW visible points per wavefront (TPW = 4, 2 or 1; 64 / TPW lanes each).
//
// Round 1 gave every visible point a whole wavefront with wave-uniform bookkeeping: ~840 VALU + ~570 SALU
// instructions per point, of which the rule itself (rasterize_points_backward.cu:141-178: ~12 operations for each of
// the ~350 pixels within rs) needs a sixth; the rest was per-task overhead executed redundantly by 64 lanes -- window
// ranges, lane tilings, record unpacking, five wave reductions, the clip -- plus a third of the lanes idle in 21-wide
// windows on 32-lane tilings.  Here the overhead is ordinary SIMD work of TPW independent tasks: every value that used
// to be wave-uniform lives in a VGPR that is uniform within the task's lanes.  A task's lanes are 16 columns x RP row
// phases (RP = 4 / TPW): the window is swept with two column slots per lane (packed fp32), rows rp, rp + RP, ... per
// lane row, eight rows of loads in flight per trip; the blend box is walked in 4 x 4 pixel patches, one per lane row.
// The five sums of a task are reduced with four DPP steps inside each 16-lane row plus RP - 1 scalar adds.
// Throughput-bound lists (large clouds) run four tasks per wavefront -- 1.9x faster than round 1 at 8 x 1M points --,
// short lists (a few tasks per wavefront, latency-bound) fewer tasks with more lanes each.  Tasks stay statically
// dealt (group q = wave, wave + n_waves, ...), ids one group ahead.
// ---------------------------------------------------------------------------------------------
// load from a uniform (SGPR) base + a 32-bit unsigned BYTE offset: one global_load with saddr + voffset, no 64-bit VALU
// address arithmetic (a 64-bit index costs v_ashr + v_lshl_add_u64 per load, a 64-bit product much more)
struct __attribute__((packed, aligned(4))) Frag4 { int32_t a, b, c, d; };   // four consecutive dwords, 4-byte aligned
template <typename T>
__device__ __forceinline__ T ld_off(const T *__restrict__ base, uint32_t byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <int CTRL>
__device__ __forceinline__ float row_add(float v) { return v + dpp_f32<CTRL>(v); }
// sum over the 16 lanes of a DPP row, left in every lane of the row (fixed order: deterministic)
__device__ __forceinline__ float row_sum16(float v)
{
    v = row_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = row_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = row_add<0x141>(v);  // row_half_mirror
    v = row_add<0x140>(v);  // row_mirror
    return v;
}
// sum over the lanes of a task (RP rows of 16), left in every lane of the task
template <int RP>
__device__ __forceinline__ float task_sum(float v, int grp)
{
    v = row_sum16(v);
    if (RP == 1) return v;
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    if (RP == 4) return (a + b) + (c + d);
    return grp == 0 ? a + b : c + d;
}
// max over the tasks of a wavefront of an int that is uniform within each task -> wave-uniform (SGPR)
template <int TPW>
__device__ __forceinline__ int tasks_max(int v)
{
    int m = __builtin_amdgcn_readlane(v, 0);
    if (TPW >= 2) m = max(m, __builtin_amdgcn_readlane(v, 32));
    if (TPW >= 4) m = max(m, max(__builtin_amdgcn_readlane(v, 16), __builtin_amdgcn_readlane(v, 48)));
    return m;
}


// ---------------------------------------------------------------------------------------------
// Backward preparation of short lists, round 4: bucket-sorted radius keys, a median that only touches one bucket, and
// the median inside the gather launch.
//
// Round 3: backward_compact_kernel (5.1 us) -> median_visible_kernel (10.6 us: ONE workgroup radix-selects over all
// ~52k radius keys, bound by the instruction issue of one CU) -> gather (20.2 us).  Now:
//   fb_prep_kernel    one 256-thread workgroup per segment of P/64 points (64 workgroups instead of 16): visible ids in id
//                     order, zero rows for invisible points, and per cloud the segment's radius keys counting-sorted into
//                     256 monotone buckets of 1/16 octave + the exclusive prefix of the bucket counts; extra workgroups
//                     copy the alpha channel of the image gradient into its dense plane
//   fb_median         sums the 64 prefix rows -> bucket of the median -> reads ONLY that bucket's keys (a contiguous run of
//                     every segment, ~1/24 of all keys) -> exact selection by 1024-bin refinement passes in LDS
// and the medians run in the first N workgroups of the GATHER launch (DSS_OPT_BACKWARD_FUSED 4; 5 = a launch of their own):
// the gather's persistent workgroups are resident from the start, do the half of every task that does not need the search
// radius (the blend backward over the splat's own box) while workgroup n selects the median of cloud n, pick rs up
// through memory and continue with the occupancy sweep.
// Synchronisation is by TAG words: the median workgroup stores the launch's tag behind rs; the others poll until the
// word equals the tag.  The tag is the AQL dispatch id of the launch mixed with its queue address: unique per launch --
// also per replay of a captured graph, where the kernel arguments are frozen -- so no word is ever reset, no memset
// precedes the launch and a workspace may hold anything (a stale word equals the new tag only if this very dispatch wrote
// it).  The median workgroups are the first of the grid and wait for nobody: no deadlock even if the grid exceeded the
// resident capacity.  The wait is bounded (a lost producer gives wrong numbers, not a hung GPU).
// The L2s of the eight XCDs are not coherent, and agent-scope fences (L2 write-back / invalidate) cost tens of microseconds
// here (measured: a launch that also held the compaction and synchronised it with fences took 56 us against 36 us for the
// three round-3 launches; with the alpha plane behind an acquire in every workgroup 103 us): rs and its tag are WRITTEN
// with agent-scope (write-through) stores and READ with agent-scope loads, nothing else crosses workgroups in a launch.
// ---------------------------------------------------------------------------------------------
extern "C" __device__ unsigned long long dss_dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");

#define DSS_BACKWARD_FUSED_DEFAULT 4   // automatic choice of DSS_OPT_BACKWARD_FUSED (see render_backward_impl)
#define FB_THREADS 256
#define FB_BUCKETS 256          // level-0 buckets of the radius keys
#define FB_CAND_MAX 4096        // candidate keys of the median bucket held in LDS (more: streamed from memory per pass)
#define FB_HIST 1024            // bins of a refinement pass
#define FB_TAG_RS 0             // 8 copies (one per XCD of the polling workgroup, each on its own cache lines) x 64 clouds
#define FB_TAGS (8 * 64)
#define FB_SPIN_LIMIT (1 << 20)
#define FB_ALPHA_PER_WG 2048    // pixels of the alpha plane per (256-thread) workgroup of fb_prep_kernel
#define FB_MAX_SEG 256          // segments of the list (= FB_THREADS: fb_median scans them one per thread)
#define FB_MEDIAN_LDS (FB_CAND_MAX + FB_HIST + 3 * FB_MAX_SEG + 32)   // words of LDS of fb_median
// level-0 bucket of an order-preserving radius key: 16 octaves [2^-16, 1) of NDC radius in 256 buckets (1/16 octave each;
// everything below / above lands in the end buckets: any monotone map keeps the selection exact)
#define FB_KEY_LO 0xB7800000u   // float_key(2^-16)
#define FB_KEY_HI 0xBF800000u   // float_key(1.0)
__device__ __forceinline__ uint32_t fb_bucket(uint32_t key)
{
    const uint32_t c = min(max(key, FB_KEY_LO), FB_KEY_HI - 1u);
    return (c - FB_KEY_LO) >> 19;
}

struct FusedPrep {
    const uint8_t *visible;       // (P)
    uint32_t *seg_count;          // (chunks) visible points per segment
    int32_t *vis_list;            // (P) per segment: visible ids, ascending
    uint32_t *keys;               // (2 P) per (cloud, segment): radius keys counting-sorted by level-0 bucket
    uint32_t *chunk_hist;         // (N, chunks, FB_BUCKETS): EXCLUSIVE prefix of the bucket counts of (cloud, segment)
    uint2 *seg_range;             // (N, chunks): first entry / entries of cloud n inside segment c
    unsigned long long *tags;     // FB_TAGS words
    float *rs;                    // (N) out
    int64_t P;
    int chunks, per;              // segments (<= 64); points per thread of a segment (segment = per * 256 points)
    float radii_s;
};

__device__ __forceinline__ unsigned long long fb_launch_tag()
{
    const unsigned long long q = (unsigned long long)__builtin_amdgcn_queue_ptr();
    return dss_dispatch_id() * 0x9E3779B97F4A7C15ull ^ (q << 17) ^ (q >> 7) ^ 0x5851F42D4C957F2Dull;
}

// exclusive scan over the FB_THREADS threads of a workgroup (4 wavefronts); total left in every thread
__device__ __forceinline__ uint32_t fb_block_excl_scan(uint32_t v, uint32_t *s_w /*[4]*/, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t x = wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_w[wid] = x;
    __syncthreads();
    uint32_t woff = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < FB_THREADS / 64; ++w) {
        const uint32_t c = s_w[w];
        if (w < wid) woff += c;
        total += c;
    }
    return woff + x - v;
}

// Segment `c`: compaction + per-cloud bucket-sorted keys / bucket prefix.  s_mem: >= 2 * FB_BUCKETS + 160 words.
// PER (= F.per) is a template parameter so that the loops over a thread's points unroll: all of their loads in flight
// (as a runtime loop every load waited for the previous one: 5.5 us per segment instead of 3.5).
template <int PER>
__device__ __forceinline__ void fb_prep_segment(const FusedPrep &F, int c, const float *__restrict__ radii,
                                                const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts,
                                                int N, float *__restrict__ grad_pts, float *__restrict__ grad_feat, int C,
                                                uint32_t *s_mem)
{
    uint32_t *s_hist = s_mem, *s_cur = s_mem + FB_BUCKETS, *s_w = s_mem + 2 * FB_BUCKETS /*[64]*/,
             *s_pre = s_mem + 2 * FB_BUCKETS + 64 /*[64]*/, *s_tot = s_mem + 2 * FB_BUCKETS + 128;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int per = PER;
    const int64_t seg = (int64_t)per * FB_THREADS;
    const int64_t c0 = (int64_t)c * seg, c1 = min(c0 + seg, F.P);
    uint32_t vmask = 0;   // bit u: point c0 + u * 256 + tid is visible
    // the radius keys of the thread's points: requested together with the flags (one memory round trip less) and kept in
    // registers when there are few of them; larger segments re-read them where they are used
    constexpr bool KEEP = PER <= 4;
    const float2 *r2 = reinterpret_cast<const float2 *>(radii);
    float2 rr[KEEP ? PER : 1];
#pragma unroll
    for (int u = 0; u < per; ++u) {
        const int64_t i = c0 + (int64_t)u * FB_THREADS + tid;
        const uint8_t v = F.visible[min(i, F.P - 1)];
        if (KEEP) rr[KEEP ? u : 0] = r2[min(i, F.P - 1)];
        vmask |= ((i < c1 && v != 0) ? 1u : 0u) << u;
    }
    // visible points per (u, wavefront), in list order
#pragma unroll
    for (int u = 0; u < per; ++u) {
        const unsigned long long m = __ballot((vmask >> u) & 1u);
        if (lane == 0) s_w[u * 4 + wid] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (wid == 0) {
        const uint32_t v = lane < per * 4 ? s_w[lane] : 0u;
        const uint32_t x = wave_incl_scan(v);
        s_pre[lane] = x - v;
        if (lane == 63) s_tot[0] = x;
    }
    __syncthreads();
    const uint32_t tot = s_tot[0];
    if (tid == 0) F.seg_count[c] = tot;
#pragma unroll
    for (int u = 0; u < per; ++u) {
        const int64_t i = c0 + (int64_t)u * FB_THREADS + tid;
        const bool vis = (vmask >> u) & 1u;
        const unsigned long long m = __ballot(vis);
        if (vis) {
            const uint32_t pos = s_pre[u * 4 + wid] +
                                 __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            F.vis_list[c0 + pos] = (int32_t)i;
        } else if (i < c1 && grad_pts) {
            grad_pts[3 * i] = 0.0f; grad_pts[3 * i + 1] = 0.0f; grad_pts[3 * i + 2] = 0.0f;
            if (grad_feat)
                for (int ch = 0; ch < C; ++ch) grad_feat[(size_t)i * C + ch] = 0.0f;
        }
    }
    // per cloud that overlaps the segment (normally one): bucket counts, their prefix, then the keys in bucket order
    for (int n = 0; n < N; ++n) {
        const int64_t lo = max(c0, first_idx[n]), hi = min(c1, first_idx[n] + num_pts[n]);
        if (lo >= hi) continue;  // uniform
        __syncthreads();         // previous cloud's cursors consumed
        s_hist[tid] = 0;
        if (tid < 2) s_w[tid] = 0;
        __syncthreads();
    #pragma unroll
    for (int u = 0; u < per; ++u) {
            const int64_t i = c0 + (int64_t)u * FB_THREADS + tid;
            const bool vis = (vmask >> u) & 1u;
            const bool mine = vis && i >= lo && i < hi;
            const unsigned long long mb = __ballot(vis && i < lo), mi = __ballot(mine);
            if (lane == 0) {
                if (mb) atomicAdd(&s_w[0], (uint32_t)__popcll(mb));
                if (mi) atomicAdd(&s_w[1], (uint32_t)__popcll(mi));
            }
            if (mine) {
                const float2 r = KEEP ? rr[KEEP ? u : 0] : r2[i];
                atomicAdd(&s_hist[fb_bucket(float_key(r.x))], 1u);
                atomicAdd(&s_hist[fb_bucket(float_key(r.y))], 1u);
            }
        }
        __syncthreads();
        const uint32_t h = s_hist[tid];
        const uint32_t first_entry = s_w[0], entries = s_w[1];
        uint32_t total;
        const uint32_t excl = fb_block_excl_scan(h, s_w + 4, total);
        F.chunk_hist[((size_t)n * F.chunks + c) * FB_BUCKETS + tid] = excl;
        s_cur[tid] = excl;
        if (tid == 0) F.seg_range[(size_t)n * F.chunks + c] = make_uint2(first_entry, entries);
        __syncthreads();
        uint32_t *kb = F.keys + 2 * ((size_t)c0 + first_entry);
    #pragma unroll
    for (int u = 0; u < per; ++u) {
            const int64_t i = c0 + (int64_t)u * FB_THREADS + tid;
            if (((vmask >> u) & 1u) && i >= lo && i < hi) {
                const float2 r = KEEP ? rr[KEEP ? u : 0] : r2[i];
                const uint32_t kx = float_key(r.x), ky = float_key(r.y);
                kb[atomicAdd(&s_cur[fb_bucket(kx)], 1u)] = kx;
                kb[atomicAdd(&s_cur[fb_bucket(ky)], 1u)] = ky;
            }
        }
    }
}

// rank-k bin of a histogram held as h[NB] consecutive bins per thread (first bin of the thread: NB * tid)
template <int NB>
__device__ __forceinline__ void fb_select(const uint32_t (&h)[NB], uint32_t k, uint32_t *s_w /*[8]*/, uint32_t &bin_out,
                                          uint32_t &k_out, uint32_t &cnt_out)
{
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) v += h[i];
    uint32_t total;
    uint32_t excl = fb_block_excl_scan(v, s_w, total);
    __syncthreads();
    if (total > 0 && k >= excl && k < excl + v) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (k >= excl && k < excl + h[i]) {
                s_w[4] = (uint32_t)(NB * threadIdx.x + i);
                s_w[5] = k - excl;
                s_w[7] = h[i];
            }
            excl += h[i];
        }
    }
    __syncthreads();
    bin_out = s_w[4];
    k_out = s_w[5];
    cnt_out = s_w[7];
}

// Median of cloud n (lower median of the radius keys of its visible points, rasterizer.py:885-888) from the segments'
// bucket prefixes and bucket-sorted keys (written by an EARLIER launch: plain loads).  s_mem: FB_CAND_MAX + FB_HIST + 4 * 64 +
// 16 words.  Every global load of a step is issued before the first is used (a loop of dependent single loads took 25 us).
__device__ __forceinline__ float fb_median(const FusedPrep &F, int n, const int64_t *__restrict__ first_idx,
                                           const int64_t *__restrict__ num_pts, uint32_t *s_mem)
{
    uint32_t *s_cand = s_mem /* also: 4 x 256 partial sums */, *s_hist = s_mem + FB_CAND_MAX /* also: 257 prefix sums */,
             *s_cnt = s_hist + FB_HIST, *s_dst = s_cnt + FB_MAX_SEG, *s_base = s_dst + FB_MAX_SEG, *s_w = s_base + FB_MAX_SEG;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t f = first_idx[n];
    const int64_t cnt = max((int64_t)0, min(num_pts[n], F.P - f));
    if (cnt <= 0) return 0.0f;   // uniform
    const int64_t seg = (int64_t)F.per * FB_THREADS;
    const int c_lo = (int)(f / seg), c_hi = (int)((f + cnt - 1) / seg);
    const int n_c = c_hi - c_lo + 1;   // <= FB_MAX_SEG (= FB_THREADS: one thread per segment in the scans below)
    const uint32_t *H = F.chunk_hist + ((size_t)n * F.chunks + c_lo) * FB_BUCKETS;
    const uint2 *R = F.seg_range + (size_t)n * F.chunks + c_lo;
    PREP_MARK(4);
    // ---- level 0: S[b] = keys of the cloud in buckets < b, summed over the segments' prefix rows.  Thread = (group of 16
    // segments, four consecutive buckets): sixteen 16-byte loads in flight ----
    {
        const int g = wid, qd = lane;
        const uint2 rg = tid < n_c ? R[tid] : make_uint2(0u, 0u);   // (first entry, entries) of segment tid
        uint4 acc = make_uint4(0u, 0u, 0u, 0u);
        for (int blk = 0; blk < n_c; blk += 64) {   // (one trip for up to 64 segments)
            uint4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = blk + g * 16 + u;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (c < n_c) v[u] = reinterpret_cast<const uint4 *>(H + (size_t)c * FB_BUCKETS)[qd];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        reinterpret_cast<uint4 *>(s_cand + g * FB_BUCKETS)[qd] = acc;
        uint32_t all_keys;
        (void)fb_block_excl_scan(2u * rg.y, s_w + 16, all_keys);   // two keys per visible point
        if (tid == 0) s_hist[FB_BUCKETS] = all_keys;               // all keys of the cloud
        s_cnt[tid] = 2u * rg.y;                                    // (keys of the segment, for the last bucket's end)
        s_base[tid] = 2u * (uint32_t)((int64_t)(c_lo + tid) * seg) + 2u * rg.x;   // first key of (cloud, segment) in F.keys
    }
    __syncthreads();
    s_hist[tid] = (s_cand[tid] + s_cand[FB_BUCKETS + tid]) + (s_cand[2 * FB_BUCKETS + tid] + s_cand[3 * FB_BUCKETS + tid]);
    __syncthreads();
    const uint32_t total0 = s_hist[FB_BUCKETS];
    if (total0 == 0) return 0.0f;   // uniform: nothing visible in this cloud
    uint32_t k = (total0 - 1) / 2;  // torch.median = lower median
    {
        const uint32_t lo_b = s_hist[tid], hi_b = s_hist[tid + 1];
        if (lo_b <= k && k < hi_b) { s_w[4] = (uint32_t)tid; s_w[5] = k - lo_b; s_w[6] = hi_b - lo_b; }
    }
    __syncthreads();
    const uint32_t b1 = s_w[4], m = s_w[6];   // bucket of the median, its keys
    k = s_w[5];
    PREP_MARK(5);
    // ---- the bucket's keys: a contiguous run of every segment's sorted keys ----
    {
        uint32_t e0 = 0, e1 = 0;
        if (tid < n_c) {
            const uint32_t *hc = H + (size_t)tid * FB_BUCKETS;
            e0 = hc[b1];
            e1 = b1 + 1 < FB_BUCKETS ? hc[b1 + 1] : s_cnt[tid];
        }
        const uint32_t v = e1 - e0;
        uint32_t tot_unused;
        const uint32_t ex = fb_block_excl_scan(v, s_w + 16, tot_unused);
        s_dst[tid] = ex;
        s_cnt[tid] = v;
        s_base[tid] += e0;
    }
    __syncthreads();
    const bool in_lds = m <= FB_CAND_MAX;
    // 4 threads per segment when the cloud has up to 64 of them; fewer, larger segments get more threads each (the run of a
    // segment inside the bucket is then long: 8 segments of 4096 points = ~270 keys per run, six dependent trips of 4 x 12)
    const int tsh = n_c <= 8 ? 5 : (n_c <= 16 ? 4 : (n_c <= 32 ? 3 : 2));
    const int tps = 1 << tsh;   // threads per segment (uniform)
    const int q = tid & (tps - 1);
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    constexpr int KB = 12;   // loads in flight per thread: a segment's ~40 keys of the bucket in ONE round trip
    for (int cc = tid >> tsh; cc < n_c; cc += FB_THREADS >> tsh) {
        const uint32_t my_cnt = s_cnt[cc];
        const uint32_t *my_keys = F.keys + (size_t)s_base[cc];
        for (uint32_t j0 = q; j0 < my_cnt; j0 += (uint32_t)tps * KB) {
            uint32_t key[KB];
#pragma unroll
            for (int u = 0; u < KB; ++u) key[u] = my_keys[min(j0 + (uint32_t)tps * u, my_cnt - 1u)];
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                if (j0 + (uint32_t)tps * u < my_cnt) {
                    if (in_lds) s_cand[s_dst[cc] + j0 + (uint32_t)tps * u] = key[u];
                    kmin = min(kmin, key[u]);
                    kmax = max(kmax, key[u]);
                }
            }
        }
    }
    // min / max over the workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
    }
    __syncthreads();
    if (lane == 0) { s_w[8 + wid] = kmin; s_w[12 + wid] = kmax; }
    __syncthreads();
    uint32_t lo = min(min(s_w[8], s_w[9]), min(s_w[10], s_w[11]));
    uint32_t hi = max(max(s_w[12], s_w[13]), max(s_w[14], s_w[15]));
    PREP_MARK(6);
    // ---- refinement: 1024-bin passes over the candidates inside [lo, hi] until one key value is left ----
    while (lo != hi) {   // uniform
        const uint32_t range = hi - lo;
        const int bits = 32 - __builtin_clz(range);
        const int shift = bits > 10 ? bits - 10 : 0;
        for (int i = tid; i < FB_HIST; i += FB_THREADS) s_hist[i] = 0;
        __syncthreads();
        if (in_lds) {
            for (uint32_t j = tid; j < m; j += FB_THREADS) {
                const uint32_t key = s_cand[j];
                if (key >= lo && key <= hi) atomicAdd(&s_hist[(key - lo) >> shift], 1u);
            }
        } else {
            for (int cc = tid >> tsh; cc < n_c; cc += FB_THREADS >> tsh) {
                const uint32_t my_cnt = s_cnt[cc];
                const uint32_t *my_keys = F.keys + (size_t)s_base[cc];
                for (uint32_t j = q; j < my_cnt; j += (uint32_t)tps) {
                    const uint32_t key = my_keys[j];
                    if (key >= lo && key <= hi) atomicAdd(&s_hist[(key - lo) >> shift], 1u);
                }
            }
        }
        __syncthreads();
        const uint4 hq = reinterpret_cast<const uint4 *>(s_hist)[tid];
        const uint32_t h[4] = {hq.x, hq.y, hq.z, hq.w};
        uint32_t bin, left;
        fb_select<4>(h, k, s_w, bin, k, left);
        const uint32_t nlo = lo + (bin << shift);
        const uint32_t span = (1u << shift) - 1u;
        hi = min(hi, nlo + span < nlo ? 0xffffffffu : nlo + span);
        lo = nlo;
        __syncthreads();
        if (lo != hi && left <= 64u && in_lds) {
            // few keys left (the usual second step: ~3 of ~2500): collect them and rank them in one wavefront instead of
            // another histogram pass
            if (tid == 0) s_w[6] = 0;
            __syncthreads();
            for (uint32_t j = tid; j < m; j += FB_THREADS) {
                const uint32_t key = s_cand[j];
                if (key >= lo && key <= hi) s_hist[atomicAdd(&s_w[6], 1u)] = key;
            }
            __syncthreads();
            if (wid == 0) {
                const uint32_t mine = (uint32_t)lane < left ? s_hist[lane] : 0xffffffffu;
                uint32_t rank = 0;
                for (uint32_t j = 0; j < left; ++j) {
                    const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)j);
                    rank += (kj < mine || (kj == mine && j < (uint32_t)lane)) ? 1u : 0u;
                }
                if ((uint32_t)lane < left && rank == k) s_w[6] = mine;
            }
            __syncthreads();
            lo = hi = s_w[6];
        }
    }
    PREP_MARK(7);
    return key_float(lo) * F.radii_s;
}

// rs[n] and, behind it, the eight copies of its tag (median workgroup of a fused launch)
__device__ __forceinline__ void fb_publish_rs(const FusedPrep &F, int n, float value, unsigned long long tag)
{
    if (threadIdx.x == 0) {
        __hip_atomic_store(&F.rs[n], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // s_waitcnt vmcnt(0): the write-through store has completed
    }
    __syncthreads();
    if (threadIdx.x < 8)
        __hip_atomic_store(&F.tags[FB_TAG_RS + (size_t)threadIdx.x * 64 + n], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every other workgroup of a fused launch: wait until the search radius of every cloud is published (N <= 64).  Workgroup b
// runs on XCD b mod 8 and polls that XCD's copy of the tags (eight cache lines instead of one under ~1500 pollers).
__device__ __forceinline__ void fb_wait_rs(const FusedPrep &F, int N, unsigned long long tag)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned long long *t = F.tags + FB_TAG_RS + (size_t)(blockIdx.x & 7u) * 64;
        for (int spin = 0; spin < FB_SPIN_LIMIT; ++spin) {
            const bool ok = lane >= N || __hip_atomic_load(&t[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag;
            if (__ballot(!ok) == 0ull) break;
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
}

// the two preparation stages as launches of their own
template <int PER>
__global__ __launch_bounds__(FB_THREADS) void fb_prep_kernel(const FusedPrep F, const float *__restrict__ radii,
                                                             const int64_t *__restrict__ first_idx,
                                                             const int64_t *__restrict__ num_pts, int N,
                                                             float *__restrict__ grad_pts, float *__restrict__ grad_feat, int C,
                                                             const float *__restrict__ grad_out, float *__restrict__ alpha_plane,
                                                             size_t npix)
{
    if ((int)blockIdx.x >= F.chunks) {  // extra workgroups of the launch: dense alpha plane for the gather kernel
        const size_t i0 = (size_t)(blockIdx.x - F.chunks) * FB_ALPHA_PER_WG + threadIdx.x;
        if (C == 3) {   // (N,rows,S,4): 16-byte aligned pixels; all eight loads of the thread in flight
            const float4 *g4 = reinterpret_cast<const float4 *>(grad_out);
            float a[FB_ALPHA_PER_WG / FB_THREADS];
#pragma unroll
            for (int u = 0; u < FB_ALPHA_PER_WG / FB_THREADS; ++u) a[u] = g4[min(i0 + (size_t)u * FB_THREADS, npix - 1)].w;
#pragma unroll
            for (int u = 0; u < FB_ALPHA_PER_WG / FB_THREADS; ++u)
                if (i0 + (size_t)u * FB_THREADS < npix) alpha_plane[i0 + (size_t)u * FB_THREADS] = a[u];
        } else {
            for (int u = 0; u < FB_ALPHA_PER_WG / FB_THREADS; ++u) {
                const size_t i = i0 + (size_t)u * FB_THREADS;
                if (i < npix) alpha_plane[i] = grad_out[i * (C + 1) + C];
            }
        }
        return;
    }
    __shared__ uint32_t s_fb[2 * FB_BUCKETS + 160];
    fb_prep_segment<PER>(F, (int)blockIdx.x, radii, first_idx, num_pts, N, grad_pts, grad_feat, C, s_fb);
}
__global__ __launch_bounds__(FB_THREADS) void fb_median_kernel(const FusedPrep F, const int64_t *__restrict__ first_idx,
                                                               const int64_t *__restrict__ num_pts)
{
    __shared__ uint32_t s_fb[FB_MEDIAN_LDS];
    const float v = fb_median(F, (int)blockIdx.x, first_idx, num_pts, s_fb);
    if (threadIdx.x == 0) F.rs[blockIdx.x] = v;
}

// CYC: tile-row-cyclic band (tshift > 3): the windows are walked in BAND rows (the owned rows of a window are a
// contiguous range of band rows), the NDC y of each comes from band_image_row.  CYC = false is the contiguous band.
// BAND (round 5; multi-GPU row bands on the two-phase launch, PREP && SEG): the visible list holds the points visible on ANY
// rank, of which a rank's rows are reached by a fraction (own box: ~1/G; search window: 1/G + 2 rs / S for contiguous bands,
// about half for 8-row cyclic units at 8 ranks).  Round 3 filtered the list in a launch of its own behind the median
// (band_filter_kernel, 9.6 us + the 11.8 us median launch + a 9.7 us compaction: 7 launches per step and rank); here every
// worker workgroup filters ITS share of the list (entry e belongs to workgroup e mod n_wg: a strided deal, so that a cloud
// whose ids are spatially ordered still spreads over all workgroups) inside the launch -- once by the splat's own box before
// the blend half, once by the search radius behind the wait for the medians -- into a list in LDS, which its four wavefronts
// then share through an LDS counter; rejected points get their zero partial sums from the filtering thread.  No extra
// launch, nothing crosses workgroups but rs.
#ifndef DSS_BAND_RB
#define DSS_BAND_RB 8    // rows of window loads in flight per trip of a BAND task (14 = a 12.8-pixel window in one trip for two
                         // row phases: measured no faster -- 103 VGPRs, four workgroups per CU instead of five)
#endif
#define BAND_SHARE 512   // list entries per worker workgroup (two per thread); the host keeps P <= BAND_SHARE * workers
template <int C, bool SEG, int TPW, bool A32, bool CYC = false, bool PREP = false, bool BAND = false>
__global__ __launch_bounds__(256) void render_backward_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ grad_alpha /* dense (N,rows,S); PREP: or grad_out + C, see astride */,
    const int32_t *__restrict__ idx, const float *__restrict__ qv,
    const float *__restrict__ wsum, const float *__restrict__ scaler, const float *__restrict__ points,
    const float *__restrict__ radii, const float *__restrict__ rs, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, const uint32_t *__restrict__ vis_count,
    const int32_t *__restrict__ vis_list, int n_seg, int seg_pts, int N, int S, int K, int Crt, float clip, int row0,
    int rows, uint32_t large_waves, float *__restrict__ grad_feat, float *__restrict__ grad_pts, int tshift = 3,
    const float *__restrict__ world = nullptr /* (P,3): fused projection backward, see the epilogue */,
    const float *__restrict__ Mproj = nullptr /* (N,4,4) */,
    const FusedPrep F = FusedPrep() /* PREP: preparation stages inside this launch, see fb_prep_segment */,
    int astride = 1 /* elements between two pixels of grad_alpha: 1 = dense plane, C + 1 = the alpha channel of grad_out in place */,
    const OwnArgs OW = OwnArgs{nullptr, 1} /* owner mode of a row band, see OwnArgs */)
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    constexpr int KF = 8;            // fragment slots held in registers; deeper lists take the loop
    constexpr int GS = 64 / TPW;     // lanes per task
    constexpr int RP = GS / 16;      // row phases per task
    const int Cn = (C > 0) ? C : Crt;
    const int lane = threadIdx.x & 63, grp = lane / GS, rp = (lane % GS) >> 4, l = lane & 15;
    // PREP: the first N workgroups select the medians (fb_median) and take no gather tasks
    const uint32_t first_block = PREP ? (uint32_t)N : 0u;
    // BAND: one pool of LDS -- the median workgroups use it as fb_median's scratch, the workers for their share's records
    // and task lists (both at once would cost 42 KB per workgroup and the fifth resident workgroup of a CU)
    constexpr int BAND_WORDS = 9 * BAND_SHARE;   // 7 record arrays + 2 lists
    __shared__ uint32_t s_pool[BAND ? (FB_MEDIAN_LDS > BAND_WORDS ? FB_MEDIAN_LDS : BAND_WORDS) : 1];
    if (PREP && blockIdx.x < first_block) {
        __shared__ uint32_t s_fb_own[(PREP && !BAND) ? FB_MEDIAN_LDS : 1];
        uint32_t *s_fb = BAND ? s_pool : s_fb_own;
        PREP_MARK(0);
        __builtin_amdgcn_s_setprio(3);   // everybody else's second half waits for these few wavefronts
        const float v = fb_median(F, (int)blockIdx.x, first_idx, num_pts, s_fb);
        fb_publish_rs(F, (int)blockIdx.x, v, fb_launch_tag());
        PREP_MARK(1);
        return;
    }
    const uint32_t wave = (blockIdx.x - first_block) * 4 + (threadIdx.x >> 6);
    uint32_t n_waves = (gridDim.x - first_block) * 4;
    // SEG: vis_count[0..n_seg) are per-segment counts written by backward_compact_kernel (n_seg <= 64): every
    // wavefront scans them in registers once; a task index t maps to (segment, offset) with one ballot.
    uint32_t count, seg_excl = 0;  // first task index of segment `lane`
    __shared__ uint32_t s_bseg[BAND ? FB_MAX_SEG + 1 : 1];   // BAND: first list entry of every segment (+ the total)
    __shared__ uint32_t s_bscan[4];
    if (BAND) {
        // up to FB_MAX_SEG segments (one per thread): exclusive scan of their counts over the workgroup
        const uint32_t c = (int)threadIdx.x < n_seg ? vis_count[threadIdx.x] : 0u;
        const uint32_t ex = fb_block_excl_scan(c, s_bscan, count);
        s_bseg[threadIdx.x] = ex;
        if (threadIdx.x == 0) s_bseg[FB_MAX_SEG] = count;
    } else if (SEG) {
        const uint32_t seg_cnt = lane < n_seg ? vis_count[lane] : 0u;
        uint32_t seg_incl = seg_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(seg_incl, o, 64);
            if (lane >= o) seg_incl += y;
        }
        count = (uint32_t)__builtin_amdgcn_readlane((int)seg_incl, 63);
        seg_excl = seg_incl - seg_cnt;
    } else {
        count = *vis_count;
    }
    const uint32_t n_groups = (count + TPW - 1u) / TPW;
    // long lists use 6 workgroups per CU: more resident gathers only evict each other's image rows from L2
    const bool long_list = n_groups > 8u * n_waves;
    if (long_list) n_waves = min(n_waves, large_waves);
    if (!PREP && !BAND && wave >= n_waves) return;
    // Dealing of the groups.  Short lists: group t -> wave t mod n_waves.  Long lists are in screen-cell order (see
    // cell_count_kernel): runs of CH consecutive groups -- about one cell -- go to ONE XCD (blocks are dispatched to the
    // XCDs round robin), so that the rows a cell's tasks share are fetched into one L2 instead of eight.
    constexpr uint32_t CH = 64u;
    const bool chunked = !SEG && long_list && CH > 0u && (n_waves >> 2) >= 8u;
    const uint32_t xcd = blockIdx.x & 7u;
    uint32_t t_stride = n_waves, t0 = wave;
    if (chunked) {
        const uint32_t blocks_eff = n_waves >> 2;
        t_stride = ((blocks_eff - xcd + 7u) >> 3) * 4u;     // waves of this XCD
        t0 = (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6);
    }
    auto qof = [&](uint32_t t) -> uint32_t {
        return chunked ? ((t / max(CH, 1u)) * 8u + xcd) * CH + (t % max(CH, 1u)) : t;
    };
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)qof(t0));
    if (!PREP && !BAND && wave_u >= n_groups) return;
    const bool has_tasks = wave < n_waves && wave_u < n_groups;   // (PREP: wavefronts without tasks still take part in the wait)
    const uint32_t t_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)t0);

    // point id of this lane's task in group q (-1 beyond the list); uniform within the task's lanes
    auto task_ids = [&](uint32_t q) -> int {
        int off[TPW];
#pragma unroll
        for (int g = 0; g < TPW; ++g) {
            const uint32_t t = (uint32_t)TPW * q + (uint32_t)g;   // uniform
            off[g] = -1;
            if (t < count) {
                if (SEG) {
                    // last segment that starts at or before t (empty segments share their successor's start)
                    const int seg = (int)__popcll(__ballot(seg_excl <= t)) - 1;
                    const uint32_t excl = (uint32_t)__builtin_amdgcn_readlane((int)seg_excl, seg);
                    off[g] = seg * seg_pts + (int)(t - excl);
                } else {
                    off[g] = (int)t;
                }
            }
        }
        int o = off[0];
#pragma unroll
        for (int g = 1; g < TPW; ++g) o = (grp == g) ? off[g] : o;
        return o >= 0 ? vis_list[o] : -1;
    };
    const NdcMap ndc(S);
    const size_t plane = (size_t)rows * S;
    // PH 0: whole tasks; 1: the blend half only (needs no search radius); 2: the occupancy half only
    // PREP: the groups of a WORKGROUP (those its four wavefronts would take statically: 4 b + w + k n_waves) are shared by
    // its wavefronts through an LDS counter -- with ~4.3 groups per wavefront at the DSS sizes a static deal leaves every
    // fourth wavefront one group (25 %) more than its neighbours and the launch ends with them.  i-th group of the
    // workgroup: 4 b + (i & 3) + (i >> 2) n_waves (increasing in i).
    __shared__ uint32_t s_deal[2];
    const uint32_t blk4 = (blockIdx.x - first_block) * 4u;
    auto dyn_group = [&](uint32_t i) -> uint32_t { return blk4 + (i & 3u) + (i >> 2) * n_waves; };
    // ---- BAND: this workgroup's share of the visible list, filtered into LDS ---------------------------------------------
    // share slot t (= thread + 256 j): record arrays [id | cloud << 24, px, py, pz, rx, ry, scaler][BAND_SHARE], loaded ONCE
    // by the filter pass (the tasks of both halves then start from LDS: one dependent memory round trip less per round);
    // s_band: [0, SHARE): slots of the blend half's tasks, [SHARE, 2 SHARE): occupancy half
    float *s_rec = reinterpret_cast<float *>(s_pool);
    int32_t *s_band = reinterpret_cast<int32_t *>(s_pool) + (BAND ? 7 * BAND_SHARE : 0);
    __shared__ uint32_t s_bcnt[2];
    __shared__ float s_brs[BAND ? 64 : 1];                  // rs of every cloud, read once behind the wait
    constexpr int BAND_NSH = 24;                            // list entry = cloud << 24 | point id (P <= 262,144, N <= 64)
    int32_t b_id[2] = {-1, -1};    // this thread's (at most two) list entries: point id, cloud, NDC y, own / search reach
    int b_n[2] = {-1, -1};
    float b_py[2] = {0.f, 0.f}, b_ry[2] = {0.f, 0.f};
    if (BAND) {
        if (threadIdx.x == 0) { s_bcnt[0] = 0; s_bcnt[1] = 0; }
        __syncthreads();
        const uint32_t n_wg = gridDim.x - first_block, wg = blockIdx.x - first_block;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t e = wg + ((uint32_t)threadIdx.x + 256u * (uint32_t)j) * n_wg;
            if (e < count) {
                // last segment that starts at or before e (empty segments share their successor's start)
                int sg = 0;
#pragma unroll
                for (int st = FB_MAX_SEG / 2; st >= 1; st >>= 1) sg += (sg + st < FB_MAX_SEG && s_bseg[sg + st] <= e) ? st : 0;
                b_id[j] = vis_list[(size_t)sg * seg_pts + (e - s_bseg[sg])];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (b_id[j] >= 0) {
                const size_t i = (size_t)b_id[j];
                const int slot = (int)threadIdx.x + 256 * j;
                const float px_ = points[3 * i], py_ = points[3 * i + 1], pz_ = points[3 * i + 2];
                const float2 rr_ = reinterpret_cast<const float2 *>(radii)[i];
                const float sc_ = scaler ? scaler[i] : 0.0f;
                b_py[j] = py_;
                b_ry[j] = rr_.y;
                b_n[j] = find_cloud(b_id[j], first_idx, num_pts, N);
                reinterpret_cast<int32_t *>(s_rec)[slot] = b_id[j] | (max(b_n[j], 0) << BAND_NSH);
                s_rec[BAND_SHARE + slot] = px_;
                s_rec[2 * BAND_SHARE + slot] = py_;
                s_rec[3 * BAND_SHARE + slot] = pz_;
                s_rec[4 * BAND_SHARE + slot] = rr_.x;
                s_rec[5 * BAND_SHARE + slot] = rr_.y;
                s_rec[6 * BAND_SHARE + slot] = sc_;
            }
        }
    }
    // which = 0: by the splat's own box (blend half), 1: by the search radius (occupancy half; rs published).  Conservative
    // (the task evaluates the exact rows); a rejected point's partial sum over this band is zero and is stored here.
    auto band_filter = [&](auto which_tag) {
        constexpr int WHICH = decltype(which_tag)::value;
        const int last_row = row0 + (((rows - 1) >> 3) << tshift) + ((rows - 1) & 7);
        const float band_lo = -1 + (2 * (S - 1 - last_row)) / (float)S;     // lower edge of the lowest pixel row
        const float band_hi = -1 + (2 * (S - 1 - row0) + 2.0f) / (float)S;  // upper edge of the highest one
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bool keep = false;
            if (b_id[j] >= 0 && b_n[j] >= 0) {
                const float reach = WHICH == 0 ? b_ry[j] : s_brs[b_n[j]];
                const float py = b_py[j];
                keep = !(py + reach < band_lo || py - reach > band_hi);
                if (WHICH == 1 && OW.alpha != nullptr) {
                    keep = band_owns_row(centre_image_row(py, S), row0, rows, tshift);   // owner mode: whole windows of the own centres
                } else
                if (keep && CYC) {
                    int ylo, yhi;
                    keep = ndc_index_range(py, reach, S, ylo, yhi) &&
                           band_row_ceil(S - 1 - yhi, row0, tshift) <= min(band_row_floor(S - 1 - ylo, row0, tshift), rows - 1);
                }
            }
            const unsigned long long m = __ballot(keep);
            uint32_t base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_bcnt[WHICH], (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (keep) {
                s_band[WHICH * BAND_SHARE + base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
                    (int)threadIdx.x + 256 * j;   // the share slot: its record is in s_rec
            } else if (b_id[j] >= 0) {
                const size_t i = (size_t)b_id[j];
                if (WHICH == 0) {
                    if (grad_feat)
                        for (int ch = 0; ch < Cn; ++ch) grad_feat[i * Cn + ch] = 0.0f;
                } else {
                    grad_pts[3 * i] = 0.0f; grad_pts[3 * i + 1] = 0.0f; grad_pts[3 * i + 2] = 0.0f;
                }
            }
        }
        __syncthreads();
    };
    const uint32_t wave_in_wg = threadIdx.x >> 6;
    auto band_ids = [&](uint32_t q, int which) -> int {
        const uint32_t t = (uint32_t)TPW * q + (uint32_t)grp;
        return t < s_bcnt[which] ? s_band[which * BAND_SHARE + t] : -1;
    };
    if (PREP || BAND) {
        if (threadIdx.x < 2) s_deal[threadIdx.x] = 4u;   // (indices 0..3 are the wavefronts' first groups)
        __syncthreads();
    }
    auto run_tasks = [&](auto phase_tag) {
    constexpr int PH = decltype(phase_tag)::value;
    uint32_t t_cur = t_first;
    const uint32_t b_groups = BAND ? (s_bcnt[PH == 2 ? 1 : 0] + TPW - 1u) / TPW : 0u;   // groups of this workgroup's own list
    int p_nx = BAND ? band_ids(wave_in_wg, PH == 2 ? 1 : 0) : task_ids(wave_u);
    for (;;) {
        // BAND: p_nx is a share slot; id and cloud come from its record
        const int b_slot = BAND ? max(p_nx, 0) : 0;
        const int b_word = (BAND && p_nx >= 0) ? reinterpret_cast<const int32_t *>(s_rec)[b_slot] : -1;
        const int p = BAND ? (b_word >= 0 ? (b_word & ((1 << BAND_NSH) - 1)) : -1) : p_nx;
        const int p_cloud = b_word >= 0 ? (b_word >> BAND_NSH) : -1;
        uint32_t t_next, q_next;
        if (BAND) {
            uint32_t i_next = 0;
            if (lane == 0) i_next = atomicAdd(&s_deal[PH == 2 ? 1 : 0], 1u);
            q_next = t_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)i_next);
        } else if (PREP) {
            uint32_t i_next = 0;
            if (lane == 0) i_next = atomicAdd(&s_deal[PH == 2 ? 1 : 0], 1u);
            q_next = t_next = dyn_group((uint32_t)__builtin_amdgcn_readfirstlane((int)i_next));
        } else {
            t_next = t_cur + t_stride;
            q_next = qof(t_next);
        }
        const bool more = q_next < (BAND ? b_groups : n_groups);
        if (more) p_nx = BAND ? band_ids(q_next, PH == 2 ? 1 : 0) : task_ids(q_next);  // in flight during this group
        // ---- record + cloud of the task's point ------------------------------------------------------------
        float px = 0.f, py = 0.f, pz = -1.f, rx = 0.f, ry = 0.f, sc = 0.f, cur_r = 0.f;
        float wx = 0.f, wy = 0.f, wz = 0.f;   // world position (fused projection backward only)
        int n = -1;
        if (BAND) {
            if (p >= 0) {
                px = s_rec[BAND_SHARE + b_slot]; py = s_rec[2 * BAND_SHARE + b_slot]; pz = s_rec[3 * BAND_SHARE + b_slot];
                rx = s_rec[4 * BAND_SHARE + b_slot]; ry = s_rec[5 * BAND_SHARE + b_slot]; sc = s_rec[6 * BAND_SHARE + b_slot];
                n = p_cloud;
                if (PH != 1) cur_r = s_brs[n];
            }
        } else if (p >= 0) {
            px = points[3 * (size_t)p]; py = points[3 * (size_t)p + 1]; pz = points[3 * (size_t)p + 2];
            const float2 rr = reinterpret_cast<const float2 *>(radii)[p];
            rx = rr.x; ry = rr.y;
            sc = scaler ? scaler[p] : 0.0f;
            if (world) { wx = world[3 * (size_t)p]; wy = world[3 * (size_t)p + 1]; wz = world[3 * (size_t)p + 2]; }
            for (int cld = 0; cld < N; ++cld) {  // scalar loads; N is small
                const int64_t f = first_idx[cld];
                const bool own = (int64_t)p >= f && (int64_t)p < f + num_pts[cld];
                n = own ? cld : n;
                if (PH != 1) cur_r = own ? (PREP ? __hip_atomic_load(&rs[cld], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rs[cld]) : cur_r;
            }
        }
        const int nn = max(n, 0);
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 gx2 = {0.0f, 0.0f}, gy2 = {0.0f, 0.0f};  // the lane's two column slots, summed at the end
        if constexpr (PH != 1) {
        // ---- occupancy window (rasterize_points_backward.cu:141-178) ---------------------------------------
        const float cur_r2 = cur_r * cur_r;
        int xlo = 0, xhi = -1, ylo = 0, yhi = -1;
        bool o_ok = n >= 0 && !(pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f) &&
                    ndc_index_range_tight(px, cur_r, S, xlo, xhi) && ndc_index_range_tight(py, cur_r, S, ylo, yhi);
        // owner mode: the whole window over the full image, for the points whose centre row is one of this band's rows
        const bool own_m = OW.alpha != nullptr;
        const bool o_cyc = CYC && !own_m;
        const int o_row0 = own_m ? 0 : row0, o_rows = own_m ? S : rows;
        const size_t o_plane = own_m ? (size_t)S * S : plane;
        const int o_astride = own_m ? OW.astride : astride;
        const float *__restrict__ o_alpha = own_m ? OW.alpha : grad_alpha;
        if (own_m) o_ok = o_ok && band_owns_row(centre_image_row(py, S), row0, rows, tshift);
        int l_hi = -1;   // CYC: band row of window row 0 (the window's rows are band rows l_hi, l_hi - 1, ..., l_hi - oh + 1)
        if (o_ok) {
            if (o_cyc) {
                // image rows S-1-yhi .. S-1-ylo -> the band rows among them
                const int l_lo = band_row_ceil(S - 1 - yhi, row0, tshift);
                l_hi = min(band_row_floor(S - 1 - ylo, row0, tshift), rows - 1);
                o_ok = l_lo <= l_hi;
                ylo = 0;
                yhi = l_hi - l_lo;   // (only yhi - ylo + 1 = the number of window rows is used below)
            } else {
                ylo = max(ylo, S - o_row0 - o_rows);
                yhi = min(yhi, S - 1 - o_row0);
                o_ok = ylo <= yhi;
            }
        }
        if (!o_ok) { xlo = 0; xhi = -1; ylo = 0; yhi = -1; l_hi = 0; }
        const int ow = xhi - xlo + 1, oh = yhi - ylo + 1;          // 0 for an empty window
        const int ncp = (tasks_max<TPW>(ow) + 31) >> 5;             // column-slot pairs: wave-uniform
        const int nrow = (tasks_max<TPW>(oh) + RP - 1) / RP;        // rows per lane row: wave-uniform
        const float *__restrict__ gimg = o_alpha + (size_t)nn * o_plane * o_astride;
        for (int cp = 0; cp < ncp; ++cp) {
            // this lane's two columns of the pass: x0 = xlo + 32 cp + l, x1 = x0 + 16 (clamped: masked, in bounds)
            const int x0 = xlo + 32 * cp + l, x1 = x0 + 16;
            const bool c0 = x0 <= xhi, c1 = x1 <= xhi;
            const int x0c = c0 ? x0 : max(xhi, 0), x1c = c1 ? x1 : max(xhi, 0);
            const f2 dx = {ndc(x0c) - px, ndc(x1c) - px};
            // a column slot beyond the window gets an infinite squared distance: the radius test then drops it (the sums see
            // the same +0 as from a masked gradient), and the loads of a trip need no per-slot mask
            const f2 dx2 = {c0 ? dx.x * dx.x : __builtin_inff(), c1 ? dx.y * dx.y : __builtin_inff()};
            // "g > 0 and outside the splat's box": with ry_eff = -1 for out-of-box columns the row test alone decides
            const f2 ry_eff = {(fabsf(dx.x) > rx) ? -1.0f : ry, (fabsf(dx.y) > rx) ? -1.0f : ry};
            // image (row, col) of NDC (y, x) is (S-1-y, S-1-x); band row = S-1-y-row0: one image row up per NDC row
            const int top = o_cyc ? max(l_hi, 0) : (S - 1 - o_row0 - max(ylo, 0));   // band row of window row 0
            const int i0 = top * S + (S - 1 - x0c);  // element offsets in the camera's plane
            const int i1 = top * S + (S - 1 - x1c);
            // RB rows per trip: all their loads are issued before the first is used (one memory round trip per trip; a
            // row-at-a-time loop spent ~1 us per ROW waiting)
            constexpr int RB = BAND ? DSS_BAND_RB : 8;
            // NDC y of the lane's rows: for S = 2^k every pixel centre and every centre-to-centre distance is an exact
            // fp32 multiple of 1/S, so the rows advance by exact additions (one v_add per row instead of convert +
            // multiply + add behind a branch); other sizes evaluate the reference expression per row
            const float y_step = (float)(2 * RP) * ndc.invS;          // distance of two consecutive rows of a lane row
            for (int ib = 0; ib < nrow; ib += RB) {
                float g0[RB], g1[RB];
                if (A32) {
                    // unconditional loads from clamped (always valid) addresses, masked afterwards: no exec-mask branch
                    // around every load, 32-bit byte offsets from the tensor base
                    // (without the dense plane: the alpha channel of grad_out in place, `astride` floats per pixel)
                    const uint32_t e4 = 4u * (uint32_t)o_astride;
                    const uint32_t S4 = (uint32_t)S * e4;
                    const uint32_t w0 = ((uint32_t)nn * (uint32_t)o_plane + (uint32_t)(o_ok ? i0 : 0)) * e4;
                    const uint32_t w1 = ((uint32_t)nn * (uint32_t)o_plane + (uint32_t)(o_ok ? i1 : 0)) * e4;
                    // a trip whose RB rows lie inside the window of every task of the wavefront (two of three trips at the
                    // bench sizes) walks two running offsets: no row test, no clamp, no mask (5 VALU per row less of 26);
                    // a task without a window stays on pixel 0 of its plane (step 0) and is dropped by its infinite dx2
                    const bool lane_full = !o_ok || rp + RP * (ib + RB - 1) < oh;
                    if (__ballot(!lane_full) == 0ull) {
                        const uint32_t step4 = o_ok ? (uint32_t)RP * S4 : 0u;
                        const uint32_t first = __umul24((uint32_t)(o_ok ? rp + RP * ib : 0), S4);
                        uint32_t q0 = w0 - first, q1 = w1 - first;
#pragma unroll
                        for (int u = 0; u < RB; ++u) {
                            g0[u] = ld_off(o_alpha, q0);
                            g1[u] = ld_off(o_alpha, q1);
                            q0 -= step4;
                            q1 -= step4;
                        }
                    } else {
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int i = rp + RP * (ib + u);   // window row of this lane row
                        const bool r_ok = i < oh;
                        const uint32_t back = __umul24((uint32_t)(r_ok ? i : 0), S4);
                        const float a0 = ld_off(o_alpha, w0 - back), a1 = ld_off(o_alpha, w1 - back);
                        g0[u] = r_ok ? a0 : 0.0f;
                        g1[u] = r_ok ? a1 : 0.0f;
                    }
                    }
                } else {
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int i = rp + RP * (ib + u);   // window row of this lane row
                    const bool r_ok = i < oh;
                    g0[u] = 0.0f;
                    g1[u] = 0.0f;
                    if (r_ok && c0) g0[u] = gimg[(size_t)(i0 - i * S) * o_astride];
                    if (r_ok && c1) g1[u] = gimg[(size_t)(i1 - i * S) * o_astride];
                }
                }
                const float y_ib = ndc(ylo + rp + RP * ib);
                auto consume = [&](auto pow2_tag) {
                constexpr bool POW2 = decltype(pow2_tag)::value;
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if (ib + u >= nrow) break;  // uniform
                    // CYC: window row i is band row l_hi - i; its image row comes from the band map
                    const float yv = o_cyc ? ndc(S - 1 - band_image_row(max(l_hi - (rp + RP * (ib + u)), 0), row0, tshift))
                                           : (POW2 ? y_ib + (float)u * y_step : ndc(ylo + rp + RP * (ib + u)));
                    const float dy = yv - py;
                    const float dy2 = dy * dy;
                    const f2 gg = {g0[u], g1[u]};
                    const f2 d2 = dx2 + dy2;
                    const bool s0 = (int)(d2.x > cur_r2) | ((int)(gg.x > 0.0f) & (int)(fabsf(dy) > ry_eff.x));
                    const bool s1 = (int)(d2.y > cur_r2) | ((int)(gg.y > 0.0f) & (int)(fabsf(dy) > ry_eff.y));
                    // (dx, dy) / max(d2, 1e-10) * g with a 1-ulp reciprocal and fused accumulation; out-of-window slots
                    // have g = 0, the d2 == 0 pair has dx = dy = 0: both contribute 0 without a test of their own
                    const f2 rc = {__builtin_amdgcn_rcpf(fmaxf(d2.x, 1e-10f)), __builtin_amdgcn_rcpf(fmaxf(d2.y, 1e-10f))};
                    f2 sgl = rc * gg;
                    sgl.x = s0 ? 0.0f : sgl.x;
                    sgl.y = s1 ? 0.0f : sgl.y;
                    const f2 dyy = {dy, dy};
                    gx2 = __builtin_elementwise_fma(dx, sgl, gx2);
                    gy2 = __builtin_elementwise_fma(dyy, sgl, gy2);
                }
                };
                if (ndc.pow2) consume(std::true_type{}); else consume(std::false_type{});
            }
        }
        }
        float gx = gx2.x + gx2.y, gy = gy2.x + gy2.y;
        // ---- blend backward over the splat's own bounding box (a fragment with idx == p can only exist there) ---
        float acc[CM];
#pragma unroll
        for (int ch = 0; ch < CM; ++ch) acc[ch] = 0.0f;
        if (PH != 2 && grad_feat != nullptr) {
            int bxlo = 0, bxhi = -1, bylo = 0, byhi = -1;
            bool b_ok = n >= 0 && ndc_index_range_tight(px, rx, S, bxlo, bxhi) && ndc_index_range_tight(py, ry, S, bylo, byhi);
            int bl_hi = 0;   // CYC: band row of box row bylo (box rows are band rows bl_hi, bl_hi - 1, ...)
            if (b_ok) {
                if (CYC) {
                    const int bl_lo = band_row_ceil(S - 1 - byhi, row0, tshift);
                    bl_hi = min(band_row_floor(S - 1 - bylo, row0, tshift), rows - 1);
                    b_ok = bl_lo <= bl_hi;
                    bylo = 0;
                    byhi = bl_hi - bl_lo;
                } else {
                    bylo = max(bylo, S - row0 - rows);
                    byhi = min(byhi, S - 1 - row0);
                    b_ok = bylo <= byhi;
                }
            }
            if (!b_ok) { bxlo = 0; bxhi = -1; bylo = 0; byhi = -1; bl_hi = 0; }
            // a lane row = one patch of 16 pixels; the task's RP lane rows take patches rp, rp + RP, ... of its box (row-major,
            // ptx patches per row), all tasks in a common loop over the largest box
            // patch shape: 4 x 4 pixels per lane row -- 8 x 2 when a task has ONE lane row (TPW = 4): a 5..7-pixel box then
            // takes 3 patches instead of 4 (the long lists are bound by instruction issue: every iteration counts)
            constexpr int PWS = (RP == 1) ? 3 : 2, PHS = 4 - PWS;   // log2 of the patch width / height
            constexpr int PW = 1 << PWS, PH = 1 << PHS;
            const int ptx = (bxhi - bxlo + PW) >> PWS, pty = (byhi - bylo + PH) >> PHS;   // 0 for an empty box
            const int npatch = (tasks_max<TPW>(ptx * pty) + RP - 1) / RP;
            const float inv_ptx = fast_rcp((float)max(ptx, 1));   // (pi + 0.5) / ptx is never within 0.5 / ptx of an integer
            for (int it = 0; it < npatch; ++it) {
                const int pi_ = rp + RP * it;
                const int pty_i = (int)(((float)pi_ + 0.5f) * inv_ptx);   // pi_ / ptx (exact: small integers)
                const int ptx_i = pi_ - pty_i * ptx;
                const int xi = bxlo + PW * ptx_i + (l & (PW - 1)), yi = bylo + PH * pty_i + (l >> PWS);
                if (A32 && K <= KF && wsum != nullptr) {
                    // 32-bit byte offsets from the tensor bases, unconditional loads from pixel 0 of the camera for the
                    // lanes outside the box (masked by `on` below)
                    const bool on = !(pi_ >= ptx * pty || xi > bxhi || yi > byhi);
                    const uint32_t pix32 = on ? ((uint32_t)nn * (uint32_t)rows + (uint32_t)(CYC ? bl_hi - yi : S - 1 - yi - row0)) * (uint32_t)S +
                                                    (uint32_t)(S - 1 - xi)
                                              : (uint32_t)nn * (uint32_t)plane;
                    const uint32_t oK = pix32 * (uint32_t)K * 4u, oC = pix32 * (uint32_t)(Cn + 1) * 4u;
                    if (TPW >= 2) {
                        // Throughput-bound lists: two dependent rounds with fewer loads -- the K fragment ids first, then
                        // ONLY the matching slot's Q value, the weight sum and the image gradient (10 loads per pixel slot
                        // instead of 14; the extra round trip hides behind the other resident wavefronts)
                        int hitk = -1;
                        if (K == 5) {
                            // the K = 5 ids of a pixel as one 16-byte + one 4-byte load (4-byte aligned: global memory takes
                            // unaligned vector accesses) instead of five dword loads
                            const Frag4 v4 = ld_off(reinterpret_cast<const Frag4 *>(idx), oK);
                            const int32_t v5 = ld_off(idx, oK + 16u);
                            hitk = v4.a == (int32_t)p ? 0 : hitk;
                            hitk = v4.b == (int32_t)p ? 1 : hitk;
                            hitk = v4.c == (int32_t)p ? 2 : hitk;
                            hitk = v4.d == (int32_t)p ? 3 : hitk;
                            hitk = v5 == (int32_t)p ? 4 : hitk;
                        } else
#pragma unroll
                        for (int k = 0; k < KF; ++k)
                            if (k < K) hitk = (ld_off(idx, oK + 4u * k) == (int32_t)p) ? k : hitk;
                        const bool f2 = on && hitk >= 0;
                        const float q2 = ld_off(qv, oK + 4u * (uint32_t)max(hitk, 0));
                        const float cum2 = ld_off(wsum, pix32 * 4u);
                        float g2[CM];
                        if (C == 3) {   // RGBA pixel of the image gradient: one aligned 16-byte load
                            const float4 gg4 = ld_off(reinterpret_cast<const float4 *>(grad_out), oC);
                            g2[0] = gg4.x; g2[CM > 1 ? 1 : 0] = gg4.y; g2[CM > 2 ? 2 : 0] = gg4.z;
                        } else {
#pragma unroll
                            for (int ch = 0; ch < CM; ++ch) g2[ch] = (ch < Cn) ? ld_off(grad_out, oC + 4u * ch) : 0.0f;
                        }
                        const float wn2 = f2 ? ewa_weight(q2, sc) * fast_rcp(cum2) : 0.0f;
#pragma unroll
                        for (int ch = 0; ch < CM; ++ch)
                            if (ch < Cn) acc[ch] = fmaf(g2[ch], wn2, acc[ch]);
                        continue;
                    }
                    int32_t vi[KF];
                    float qk[KF];
#pragma unroll
                    for (int k = 0; k < KF; ++k) { vi[k] = -1; qk[k] = 0.0f; }
                    if (K == 5) {   // 2 x (16 + 4)-byte loads instead of ten dword loads (4-byte aligned vector accesses)
                        const Frag4 v4 = ld_off(reinterpret_cast<const Frag4 *>(idx), oK);
                        const Frag4 q4 = ld_off(reinterpret_cast<const Frag4 *>(qv), oK);
                        vi[0] = v4.a; vi[1] = v4.b; vi[2] = v4.c; vi[3] = v4.d; vi[4] = ld_off(idx, oK + 16u);
                        qk[0] = __int_as_float(q4.a); qk[1] = __int_as_float(q4.b); qk[2] = __int_as_float(q4.c);
                        qk[3] = __int_as_float(q4.d); qk[4] = ld_off(qv, oK + 16u);
                    } else
#pragma unroll
                    for (int k = 0; k < KF; ++k)
                        if (k < K) { vi[k] = ld_off(idx, oK + 4u * k); qk[k] = ld_off(qv, oK + 4u * k); }
                    const float cum32 = ld_off(wsum, pix32 * 4u);
                    float g32[CM];
                    if (C == 3) {
                        const float4 gg4 = ld_off(reinterpret_cast<const float4 *>(grad_out), oC);
                        g32[0] = gg4.x; g32[CM > 1 ? 1 : 0] = gg4.y; g32[CM > 2 ? 2 : 0] = gg4.z;
                    } else
#pragma unroll
                    for (int ch = 0; ch < CM; ++ch) g32[ch] = (ch < Cn) ? ld_off(grad_out, oC + 4u * ch) : 0.0f;
                    float q32 = 0.0f;
                    bool f32_ = false;
#pragma unroll
                    for (int k = 0; k < KF; ++k) {
                        const bool hit = (k < K) && vi[k] == (int32_t)p;
                        q32 = hit ? qk[k] : q32;
                        f32_ = f32_ || hit;
                    }
                    const float wn32 = (f32_ && on) ? ewa_weight(q32, sc) * fast_rcp(cum32) : 0.0f;
#pragma unroll
                    for (int ch = 0; ch < CM; ++ch)
                        if (ch < Cn) acc[ch] = fmaf(g32[ch], wn32, acc[ch]);
                    continue;
                }
                if (pi_ >= ptx * pty || xi > bxhi || yi > byhi) continue;
                const size_t pix = ((size_t)nn * rows + (CYC ? bl_hi - yi : S - 1 - yi - row0)) * S + (S - 1 - xi);
                const int32_t *pi = idx + pix * K;
                const float *pq = qv + pix * K;
                const float *go = grad_out + pix * (Cn + 1);
                float q_sel = 0.0f;
                bool found = false;
                float cum;
                float gch[CM];
                if (K <= KF && wsum != nullptr) {
                    // everything the pixel needs in ONE round trip (the image gradient is read whether or not the point
                    // is among the pixel's fragments: it almost always is)
                    int32_t vi[KF];
                    float qk[KF];
#pragma unroll
                    for (int k = 0; k < KF; ++k) {
                        vi[k] = -1; qk[k] = 0.0f;
                        if (k < K) { vi[k] = pi[k]; qk[k] = pq[k]; }
                    }
                    cum = wsum[pix];
#pragma unroll
                    for (int ch = 0; ch < CM; ++ch) gch[ch] = (ch < Cn) ? go[ch] : 0.0f;
#pragma unroll
                    for (int k = 0; k < KF; ++k) {
                        const bool hit = (k < K) && vi[k] == (int32_t)p;
                        q_sel = hit ? qk[k] : q_sel;
                        found = found || hit;
                    }
                } else {
                    cum = 0.0f;
                    for (int k = 0; k < K; ++k) {
                        const int32_t v = pi[k];
                        if (v == (int32_t)p) { found = true; q_sel = pq[k]; }
                        if (!wsum && v >= 0) cum += ewa_weight(pq[k], scaler[v]);
                    }
                    if (wsum) cum = wsum[pix];
                    else if (cum < 1e-4f) cum = 1e-4f;
#pragma unroll
                    for (int ch = 0; ch < CM; ++ch) gch[ch] = (ch < Cn) ? go[ch] : 0.0f;
                }
                if (!found) continue;
                const float wn = ewa_weight(q_sel, sc) * fast_rcp(cum);
#pragma unroll
                for (int ch = 0; ch < CM; ++ch)
                    if (ch < Cn) acc[ch] = fmaf(gch[ch], wn, acc[ch]);
            }
        }
        // ---- per-task reductions, clip, stores -------------------------------------------------------------
        if (PH != 1) {
            gx = task_sum<RP>(gx, grp);
            gy = task_sum<RP>(gy, grp);
        }
        if (PH != 2) {
#pragma unroll
            for (int ch = 0; ch < CM; ++ch)
                if (ch < Cn) acc[ch] = task_sum<RP>(acc[ch], grp);
        }
        if (PH == 1) {
            if (l == 0 && rp == 0 && p >= 0 && n >= 0 && grad_feat) {
#pragma unroll
                for (int ch = 0; ch < CM; ++ch)
                    if (ch < Cn) grad_feat[(size_t)p * Cn + ch] = acc[ch];
            }
        } else if (l == 0 && rp == 0 && p >= 0 && n >= 0) {
            if (clip > 0.0f) {  // rasterizer.py:667-673 (z gradient is 0 on this path)
                const float nrm = sqrtf(gx * gx + gy * gy);
                const float k = fminf(nrm, clip) * fast_rcp(fmaxf(nrm, 1e-12f));   // 1-ulp reciprocal instead of two divides
                gx *= k;
                gy *= k;
            }
            float o0 = gx, o1 = gy, o2 = 0.0f;
            if (world) {
                // Fused backward of the projection (project_backward_kernel, setup.hip, operation for operation; the z
                // gradient is 0 on this path): the task already holds its point's clipped screen-space gradient, so the
                // world-space gradient leaves from here and the separate launch (4 us of the 80 us step at 32k points)
                // disappears.  Only for clouds that are not shared between cameras (packed index == world index): a
                // shared cloud sums over its cameras in a fixed order in the separate kernel.
                const float *m = Mproj + 16 * nn;
                const float cx = wx * m[0] + wy * m[4] + wz * m[8] + m[12];
                const float cy = wx * m[1] + wy * m[5] + wz * m[9] + m[13];
                const float w = wx * m[3] + wy * m[7] + wz * m[11] + m[15];
                const float iw = 1.0f / w;
                const float nx = cx * iw, ny = cy * iw;
                float t3[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const float jx = (m[i * 4 + 0] - nx * m[i * 4 + 3]) * iw;
                    const float jy = (m[i * 4 + 1] - ny * m[i * 4 + 3]) * iw;
                    t3[i] = jx * gx + jy * gy + 0.0f;
                }
                o0 = t3[0]; o1 = t3[1]; o2 = t3[2];
            }
            grad_pts[3 * (size_t)p] = o0;
            grad_pts[3 * (size_t)p + 1] = o1;
            grad_pts[3 * (size_t)p + 2] = o2;
            if (PH == 0 && grad_feat) {
#pragma unroll
                for (int ch = 0; ch < CM; ++ch)
                    if (ch < Cn) grad_feat[(size_t)p * Cn + ch] = acc[ch];
            }
        }
        if (!more) break;
        t_cur = t_next;
    }
    };
    if (BAND) {
        // row band: own-box filter -> blend half -> (medians) -> search-radius filter -> occupancy half
        PREP_MARK(0);
        if (grad_feat) {
            band_filter(std::integral_constant<int, 0>{});
            PREP_MARK(8);
            if (wave_in_wg * TPW < s_bcnt[0]) run_tasks(std::integral_constant<int, 1>{});
        }
        PREP_MARK(1);
        if (PREP) fb_wait_rs(F, N, fb_launch_tag());
        if ((int)threadIdx.x < N)
            s_brs[threadIdx.x] = PREP ? __hip_atomic_load(&rs[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rs[threadIdx.x];
        __syncthreads();
        PREP_MARK(2);
        band_filter(std::integral_constant<int, 1>{});
        PREP_MARK(9);
        if (wave_in_wg * TPW < s_bcnt[1]) run_tasks(std::integral_constant<int, 2>{});
        PREP_MARK(3);
    } else if (!PREP) {
        run_tasks(std::integral_constant<int, 0>{});
    } else {
        // the half of every task that needs no search radius, while the first workgroups select the medians
        PREP_MARK(0);
        if (has_tasks) run_tasks(std::integral_constant<int, 1>{});
        PREP_MARK(1);
        fb_wait_rs(F, N, fb_launch_tag());
        PREP_MARK(2);
        if (has_tasks) run_tasks(std::integral_constant<int, 2>{});
        PREP_MARK(3);
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

// Workspace of the two-launch preparation (P <= PREP_MAX_POINTS), relative to its own base.
struct PrepLayout {
    size_t seg_count, vis_list, vis_keys, chunk_hist, seg_range, rs, alpha, tags, bytes;
    int chunks, per;  // segments, points per thread of the compaction kernel (segment = per * 1024 points)
    int f_chunks, f_per;  // two-launch backward (fb_prep_segment): segments of f_per * 256 points, at most 64 (whole image:
                          // the gather maps a task to its segment with one ballot over 64 lanes) or FB_MAX_SEG (row band: the
                          // BAND variants search a table in LDS; at 8 x 32,684 points 64 segments of 4096 points kept a
                          // quarter of the CUs busy for 22 us, 256 of 1024 take a quarter of that)
};
static PrepLayout prep_layout(int N, int64_t P, int S, bool band = false)
{
    PrepLayout L;
    const size_t n = N > 0 ? N : 1, p = P > 0 ? (size_t)P : 1;
    L.per = prep_points_per_thread(P);
    const size_t seg = (size_t)L.per * PREP_THREADS;
    L.chunks = (int)((p + seg - 1) / seg);
    L.f_per = 1;
    const size_t f_max_seg = (band && p > (size_t)2 * FB_THREADS * PREP_MAX_SEG) ? FB_MAX_SEG : PREP_MAX_SEG;
    while (L.f_per < 16 && (size_t)L.f_per * FB_THREADS * f_max_seg < p) L.f_per *= 2;
    L.f_chunks = (int)((p + (size_t)L.f_per * FB_THREADS - 1) / ((size_t)L.f_per * FB_THREADS));
    size_t off = 0;
    L.seg_count = off;  off += (size_t)FB_MAX_SEG * 4;                           // segment counters (any segmentation)
    L.vis_list = off;   off += align_up(p * 4, 256);
    L.vis_keys = off;   off += align_up(p * 8, 256);
    L.chunk_hist = off; off += align_up(n * (size_t)FB_MAX_SEG * 256 * 4, 256);   // (any segmentation: <= FB_MAX_SEG segments)
    L.seg_range = off;  off += align_up(n * (size_t)FB_MAX_SEG * 8, 256);
    L.tags = off;       off += align_up((size_t)FB_TAGS * 8, 256);   // rs tags of the fused gather launch
    L.rs = off;         off += align_up(n * 4, 256);
    L.alpha = off;      off += align_up(n * (size_t)(S > 0 ? S : 0) * (size_t)(S > 0 ? S : 0) * 4, 256);  // S = 0: none
    L.bytes = off;
    return L;
}

static void launch_prep(const float *radii, const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts,
                        int N, int64_t P, float radii_s, float *rs, char *w, const PrepLayout &L, float *grad_pts,
                        float *grad_feat, int C, const float *grad_out, size_t npix, hipStream_t st)
{
    float *alpha = reinterpret_cast<float *>(w + L.alpha);
    const unsigned alpha_wgs = grad_out ? (unsigned)((npix + ALPHA_PIX_PER_WG - 1) / ALPHA_PIX_PER_WG) : 0u;
    uint32_t *seg_count = reinterpret_cast<uint32_t *>(w + L.seg_count);
    int32_t *vis_list = reinterpret_cast<int32_t *>(w + L.vis_list);
    uint2 *vis_keys = reinterpret_cast<uint2 *>(w + L.vis_keys);
    uint32_t *chunk_hist = reinterpret_cast<uint32_t *>(w + L.chunk_hist);
    uint2 *seg_range = reinterpret_cast<uint2 *>(w + L.seg_range);
#define DSS_LAUNCH_COMPACT(PER_)                                                                                       \
    hipLaunchKernelGGL(backward_compact_kernel<PER_>, dim3(L.chunks + alpha_wgs), dim3(PREP_THREADS), 0, st, radii,       \
                       visible, first_idx, num_pts, N, P, L.chunks, seg_count, vis_list, vis_keys, chunk_hist, seg_range, \
                       grad_pts, grad_feat, C, grad_out, alpha, npix)
    if (L.per == 2) DSS_LAUNCH_COMPACT(2);
    else DSS_LAUNCH_COMPACT(4);
#undef DSS_LAUNCH_COMPACT
    hipLaunchKernelGGL(median_visible_kernel, dim3(N), dim3(PREP_THREADS), 0, st, first_idx, num_pts, P, L.chunks,
                       L.per * PREP_THREADS, seg_range, vis_keys, chunk_hist, radii_s, rs);
}

extern "C" size_t dss_backward_radius_workspace(int N, int64_t P)
{
    if (P <= PREP_MAX_POINTS) return prep_layout(N, P, 0).bytes;
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
    if (P <= PREP_MAX_POINTS) {
        if (P > 0)
            launch_prep(radii, visible, first_idx, num_pts, N, P, radii_s, rs, reinterpret_cast<char *>(workspace),
                        prep_layout(N, P, 0), nullptr, nullptr, 0, nullptr, 0, st);
        else
            (void)hipMemsetAsync(rs, 0, (size_t)N * 4, st);
        return check_launch("dss_backward_radius");
    }
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

extern "C" int dss_occ_backward_box(const float *points, const float *radii, const float *grad_occ,
                                    const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S,
                                    float radii_s, float *grad_xy, void *stream)
{
    if (N <= 0 || P < 0 || S <= 0 || !(radii_s > 0.0f)) {
        set_error("dss_occ_backward_box: bad sizes N=%d P=%lld S=%d radii_s=%g", N, (long long)P, S, (double)radii_s);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!points || !radii || !grad_occ || !first_idx || !num_pts || !grad_xy) {
        set_error("dss_occ_backward_box: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const long long blocks = (P + 3) / 4;
    if (blocks > 0x7fffffffll) { set_error("dss_occ_backward_box: P too large"); return DSS_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(occ_box_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), points, radii,
                       grad_occ, first_idx, num_pts, N, P, S, radii_s, grad_xy);
    return check_launch("dss_occ_backward_box");
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

extern "C" size_t dss_render_backward_workspace(int N, int64_t P, int S)
{
    const size_t n = N > 0 ? N : 1, s = S > 0 ? S : 1;
    if (P <= PREP_MAX_POINTS) return prep_layout(N, P, S).bytes;
    const size_t cells = (size_t)make_cells(N > 0 ? N : 1, S > 0 ? S : 1).total;
    const size_t p1 = (size_t)(P > 0 ? P : 1);
    return align_up((size_t)3 * n * MED_BINS * 4 + 256, 256)  // histograms + visible counter
           + align_up(p1 * 4, 256)                            // compacted visible list
           + align_up(n * 4, 256)                             // rs
           + align_up(n * s * s * 4, 256)                     // dense alpha-gradient plane
           + 2 * align_up(p1 * 4, 256)                        // cell of every list entry, cell-ordered list
           + 2 * align_up(cells * 4, 256)                     // entries per cell, first list position of every cell
           + align_up(cell_blocks(P) * cells * 4, 256);       // per sort block: entries of every cell in earlier blocks
}

static int render_backward_impl(bool run_prep, const float *grad_out, const int32_t *idx, const float *qvalue, const float *wsum,
                                const float *scaler, const float *points, const float *radii,
                                const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                int64_t P, int S, int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                float *grad_feat, float *grad_pts, float *rs_out, const float *world, const float *Mproj,
                                void *workspace, size_t workspace_bytes, void *stream,
                                const float *grad_full = nullptr /* (N,S,S,C+1): owner mode of a row band, see OwnArgs */,
                                const float *occ_full = nullptr /* (N,S,S) dense: owner mode fed by the all-gathered alpha-gradient plane */)
{
    if (N <= 0 || P < 0 || S <= 0 || K <= 0 || C < 1 || C > BLEND_MAX_C || row0 < 0 || row1 > S || row0 >= row1 ||
        row_cycle < 1 || (row_cycle & (row_cycle - 1)) || row_cycle > 4096) {
        set_error("dss_render_backward: bad sizes N=%d P=%lld S=%d K=%d C=%d rows=[%d,%d) cycle %d", N, (long long)P, S, K, C,
                  row0, row1, row_cycle);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (world != nullptr) {
        // fused projection backward: grad_pts then receives WORLD-space gradients (see the kernel's epilogue)
        if (!Mproj || row0 != 0 || row1 != S || row_cycle > 1 || C != 3) {
            set_error("dss_render_backward: the fused projection needs M, the whole image (no row band) and C == 3");
            return DSS_ERR_INVALID_ARGUMENT;
        }
    }
    const int rows = dss_band_rows(row0, row1, row_cycle);   // band-local rows
    int tshift = 3;
    for (int c = row_cycle; c > 1; c >>= 1) ++tshift;
    const bool cyc = row_cycle > 1;
    if (P == 0) return DSS_OK;
    if (P > 0x7ffffff0ll) { set_error("dss_render_backward: P too large"); return DSS_ERR_UNSUPPORTED; }
    // owner mode only means something on a band (the whole image owns every centre: the plain form)
    OwnArgs OW = {(grad_full != nullptr && (rows < S || cyc)) ? grad_full + C : nullptr, C + 1};
    // (the dense plane of the occupancy gradient handed in: nothing to extract, the owned windows read it as it is)
    const bool own_given = occ_full != nullptr && (rows < S || cyc);
    if (own_given) { OW.alpha = occ_full; OW.astride = 1; }
    const int own = OW.alpha != nullptr ? 1 : 0;
    if (!grad_out || !points || !radii || !visible || !first_idx || !num_pts || !grad_pts ||
        (grad_feat && (!idx || !qvalue || !scaler))) {
        set_error("dss_render_backward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t need = dss_render_backward_workspace(N, P, S);
    if (!workspace || workspace_bytes < need) {
        set_error("dss_render_backward: workspace %zu bytes < required %zu", workspace_bytes, need);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    char *w = reinterpret_cast<char *>(workspace);
    const bool small = P <= PREP_MAX_POINTS;
    int n_seg = 0, seg_pts = 0;
    uint32_t *vis_count;  // small: PREP_MAX_SEG per-segment counters; otherwise one global counter
    int32_t *vis_list;
    float *rs;
    const size_t npix = (size_t)N * rows * S;
    const float *alpha;
    if (((uintptr_t)grad_out & 15u) && C == 3) { set_error("dss_render_backward: grad_out must be 16-byte aligned"); return DSS_ERR_INVALID_ARGUMENT; }
    // persistent grid = exactly the resident capacity of the chip for this kernel (a larger grid would
    // leave late workgroups waiting for slots while their share of the list sits idle)
    // Cached per DEVICE ordinal (api.hip: four atomic slots per device -- CU count, capacity for C == 3, capacity for the
    // generic-channel kernel): sized by the <C, true, 4, true> instantiation, the variant with the most registers, so the
    // grid never exceeds the resident capacity of whichever variant is launched below.  Racing first callers compute and
    // store the same values.
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int> *dc = device_cache(dev);
    int n_cus = dc ? dc[0].load(std::memory_order_relaxed) : 0;
    int cap = dc ? dc[C == 3 ? 1 : 2].load(std::memory_order_relaxed) : 0;
    if (cap == 0 || n_cus == 0) {
        int cus = 256, per_cu = 4;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (C == 3)
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<3, true, 4, true>, 256, 0);
        else
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<0, true, 4, true>, 256, 0);
        if (per_cu < 1) per_cu = 1;
        n_cus = cus;
        cap = cus * per_cu;
        (void)hipGetLastError();
        if (dc) {
            dc[0].store(n_cus, std::memory_order_relaxed);
            dc[C == 3 ? 1 : 2].store(cap, std::memory_order_relaxed);
        }
    }
    const unsigned pgrid = (unsigned)((P + 3) / 4 < cap ? (P + 3) / 4 : cap);
    const uint32_t large_waves = 6u * (uint32_t)n_cus * 4u;
    // tasks per wavefront: four when the list is long enough to keep every resident wavefront busy with whole groups
    // (throughput-bound), fewer -- more lanes per task, shorter dependent chains -- for short lists.  The visible count is
    // only known on the device; P bounds it and the visible fraction of a rendered cloud is 30-60 %.
    const int tpw_opt = option(DSS_OPT_BACKWARD_TPW);
    const long long est_tasks = (long long)P * 2 / 5;
    int tpw = est_tasks >= 16ll * cap * 4 ? 4 : (est_tasks >= 4ll * cap * 4 ? 2 : 1);
    if (tpw_opt == 1 || tpw_opt == 2 || tpw_opt == 4) tpw = tpw_opt;
    // 32-bit byte offsets from the tensor bases (one VALU per gather address instead of 64-bit index arithmetic) whenever
    // every gathered tensor is smaller than 4 GB; larger problems take the 64-bit addressing, four tasks per wavefront
    unsigned long long widest = (unsigned long long)N * (unsigned long long)rows * (unsigned long long)S *
                                (unsigned long long)(K > C + 1 ? K : C + 1) * 4ull;
    if (own) widest = std::max(widest, (unsigned long long)N * (unsigned long long)S * (unsigned long long)S * (unsigned long long)(own_given ? 1 : C + 1) * 4ull);
    // (DSS_OPT_BACKWARD_ADDR64 forces the 64-bit variant: it only exists for tensors nobody allocates in a test)
    const bool a32 = widest < (1ull << 32) && option(DSS_OPT_BACKWARD_ADDR64) != 1;
    if (!a32) tpw = 4;
    if (!a32 && world) { set_error("dss_render_backward: the fused projection needs gathered tensors below 4 GB"); return DSS_ERR_UNSUPPORTED; }
    // Round-4 preparation of short lists (fb_prep_kernel / fb_median): whole image, 32-bit offsets, at most 64 clouds.
    // DSS_OPT_BACKWARD_FUSED: 0 = automatic, 1 = the round-3 launch sequence, 4 = two launches (segments + alpha plane |
    // medians + gather), 5 = three (segments + alpha plane | medians | gather).
    int fused_opt = option(DSS_OPT_BACKWARD_FUSED);
    if (fused_opt == 0) fused_opt = DSS_BACKWARD_FUSED_DEFAULT;
    // grid of a BAND variant (below): the resident capacity of the instantiation, cached per device in slots 10.. of the cache
    auto band_grid = [&](int tpw_) -> unsigned {
        const int slot = 10 + (C == 3 ? (cyc ? 3 : 0) : 6) + (tpw_ == 4 ? 2 : (tpw_ == 2 ? 1 : 0));
        int fcap = (dc && slot < DSS_DEV_CACHE_SLOTS) ? dc[slot].load(std::memory_order_relaxed) : 0;
        if (fcap == 0) {
            int per_cu = 0;
#define DSS_OCC_B(CC, TT, YY) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<CC, true, TT, true, YY, true, true>, 256, 0)
            if (C == 3 && cyc) { if (tpw_ == 4) DSS_OCC_B(3, 4, true); else if (tpw_ == 2) DSS_OCC_B(3, 2, true); else DSS_OCC_B(3, 1, true); }
            else if (C == 3) { if (tpw_ == 4) DSS_OCC_B(3, 4, false); else if (tpw_ == 2) DSS_OCC_B(3, 2, false); else DSS_OCC_B(3, 1, false); }
            else { if (tpw_ == 4) DSS_OCC_B(0, 4, false); else if (tpw_ == 2) DSS_OCC_B(0, 2, false); else DSS_OCC_B(0, 1, false); }
#undef DSS_OCC_B
            (void)hipGetLastError();
            fcap = n_cus * (per_cu > 0 ? per_cu : 1);
            if (dc && slot < DSS_DEV_CACHE_SLOTS) dc[slot].store(fcap, std::memory_order_relaxed);
        }
        unsigned g = pgrid;
        if ((unsigned)fcap < g) g = (unsigned)fcap;
        if (g < (unsigned)N + 8u) g = (unsigned)N + 8u;
        return g;
    };
    FusedPrep FP = FusedPrep();
    const int astride = 1;
    bool fused = false, band = false;
    if (small) {
        band = rows < S || cyc;
        const PrepLayout L = prep_layout(N, P, S, band);
        // row band (multi-GPU): the two-launch form with the band filter inside the gather launch (BAND variants: fused_opt 4,
        // RGB or generic features on a contiguous band, RGB on a tile-row-cyclic one, every worker workgroup's share of the
        // list within BAND_SHARE entries); anything else takes the round-3 sequence
        fused = (fused_opt == 4 || fused_opt == 5) && a32 && N <= 64 && (unsigned)N + 8u <= pgrid &&
                (!band || (fused_opt == 4 && (!cyc || C == 3) && world == nullptr));
        if (fused && fused_opt == 4 && !(tpw_opt == 1 || tpw_opt == 2 || tpw_opt == 4)) {
            // two-phase gather with its groups dealt per workgroup: two tasks per wavefront win as soon as there is about a
            // task per resident wavefront (32,684 points @512^2: 58.7 us per step against 60.5 with one, 59.3 with four)
            // (a band keeps about (rows + two search radii) / S of the visible points: estimated as 2 rows / S)
            const long long est = band ? est_tasks * 2 * rows / S : est_tasks;
            tpw = est >= 16ll * cap * 4 ? 4 : (est >= 1ll * cap * 4 ? 2 : 1);
            // a tile-row-cyclic band owns at most eight consecutive rows of any window: many small tasks, one lane row each
            // (8 ranks x 32,684 points: 101-105 us per rank and step with four tasks per wavefront, 110-114 with two)
            if (cyc) tpw = 4;
        }
        // (every worker workgroup of a BAND launch holds its share of the list in LDS)
        if (fused && band && (long long)BAND_SHARE * ((long long)band_grid(tpw) - N) < (long long)P) fused = false;
        if (fused) {
            n_seg = L.f_chunks;
            seg_pts = L.f_per * FB_THREADS;
            vis_count = reinterpret_cast<uint32_t *>(w + L.seg_count);
            vis_list = reinterpret_cast<int32_t *>(w + L.vis_list);
            rs = rs_out ? rs_out : reinterpret_cast<float *>(w + L.rs);
            FP.visible = visible; FP.seg_count = vis_count; FP.vis_list = vis_list;
            FP.keys = reinterpret_cast<uint32_t *>(w + L.vis_keys);
            FP.chunk_hist = reinterpret_cast<uint32_t *>(w + L.chunk_hist);
            FP.seg_range = reinterpret_cast<uint2 *>(w + L.seg_range);
            FP.tags = reinterpret_cast<unsigned long long *>(w + L.tags);
            FP.rs = rs; FP.P = P;
            FP.chunks = L.f_chunks; FP.per = L.f_per; FP.radii_s = radii_s;
            float *plane = reinterpret_cast<float *>(w + L.alpha);
            alpha = plane;
            // owner mode: the dense plane holds the alpha channel of the FULL image gradient (the medians run inside the next
            // launch, so the rows an owned window can reach are not known yet; the blend half does not read the plane)
            const float *plane_src = own ? grad_full : grad_out;
            const size_t plane_px = own_given ? (size_t)0 : (own ? (size_t)N * S * S : npix);
            if (own && !own_given) { OW.alpha = plane; OW.astride = 1; }
            if (run_prep) {
                // stage 1: one workgroup per segment + the dense alpha plane in extra workgroups
                const unsigned alpha_wgs = (unsigned)((plane_px + FB_ALPHA_PER_WG - 1) / FB_ALPHA_PER_WG);
#define DSS_LAUNCH_FB_PREP(PER_)                                                                                          \
    hipLaunchKernelGGL(fb_prep_kernel<PER_>, dim3((unsigned)L.f_chunks + alpha_wgs), dim3(FB_THREADS), 0, st, FP, radii,   \
                       first_idx, num_pts, N, grad_pts, grad_feat, C, plane_src, plane, plane_px)
                switch (L.f_per) {
                    case 1: DSS_LAUNCH_FB_PREP(1); break;
                    case 2: DSS_LAUNCH_FB_PREP(2); break;
                    case 4: DSS_LAUNCH_FB_PREP(4); break;
                    case 8: DSS_LAUNCH_FB_PREP(8); break;
                    default: DSS_LAUNCH_FB_PREP(16); break;
                }
#undef DSS_LAUNCH_FB_PREP
                if (fused_opt == 5)
                    hipLaunchKernelGGL(fb_median_kernel, dim3((unsigned)N), dim3(FB_THREADS), 0, st, FP, first_idx, num_pts);
            }
        }
    }
    if (small && !fused) {
        const PrepLayout L = prep_layout(N, P, S);
        n_seg = L.chunks;
        seg_pts = L.per * PREP_THREADS;
        alpha = reinterpret_cast<const float *>(w + L.alpha);
        vis_count = reinterpret_cast<uint32_t *>(w + L.seg_count);
        vis_list = reinterpret_cast<int32_t *>(w + L.vis_list);
        rs = rs_out ? rs_out : reinterpret_cast<float *>(w + L.rs);
        if (run_prep)
            launch_prep(radii, visible, first_idx, num_pts, N, P, radii_s, rs, w, L, grad_pts, grad_feat, C, grad_out, npix, st);
        if (run_prep && rows < S) {  // row band: keep only the points that can reach it
            if (L.per == 2)
                hipLaunchKernelGGL(band_filter_kernel<2>, dim3(L.chunks), dim3(PREP_THREADS), 0, st, points, radii, rs,
                                   first_idx, num_pts, N, S, row0, rows, vis_count, vis_list, grad_pts, grad_feat, C, tshift, own);
            else
                hipLaunchKernelGGL(band_filter_kernel<4>, dim3(L.chunks), dim3(PREP_THREADS), 0, st, points, radii, rs,
                                   first_idx, num_pts, N, S, row0, rows, vis_count, vis_list, grad_pts, grad_feat, C, tshift, own);
        }
    } else if (!small) {
        const size_t hist_bytes = (size_t)3 * N * MED_BINS * 4;
        const CellGrid cg = make_cells(N, S);
        uint32_t *hist = reinterpret_cast<uint32_t *>(w);
        vis_count = reinterpret_cast<uint32_t *>(w + hist_bytes);
        size_t off = align_up(hist_bytes + 256, 256);
        vis_list = reinterpret_cast<int32_t *>(w + off);
        off += align_up((size_t)P * 4, 256);
        rs = rs_out ? rs_out : reinterpret_cast<float *>(w + off);
        off += align_up((size_t)N * 4, 256);
        float *plane = reinterpret_cast<float *>(w + off);
        alpha = plane;
        off += align_up((size_t)N * S * S * 4, 256);
        uint32_t *cell_of = reinterpret_cast<uint32_t *>(w + off);
        off += align_up((size_t)P * 4, 256);
        int32_t *sorted = reinterpret_cast<int32_t *>(w + off);
        off += align_up((size_t)P * 4, 256);
        uint32_t *cell_start = reinterpret_cast<uint32_t *>(w + off);
        off += align_up((size_t)cg.total * 4, 256);
        uint32_t *cell_total = reinterpret_cast<uint32_t *>(w + off);
        off += align_up((size_t)cg.total * 4, 256);
        uint32_t *block_hist = reinterpret_cast<uint32_t *>(w + off);
        int32_t *unsorted = vis_list;
        vis_list = sorted;

        if (run_prep) {
        if (!own)
            hipLaunchKernelGGL(alpha_plane_kernel, dim3((unsigned)((npix + ALPHA_PIX_PER_WG - 1) / ALPHA_PIX_PER_WG)), dim3(1024),
                               0, st, grad_out, plane, npix, C);
        if (hipMemsetAsync(hist, 0, hist_bytes + 256, st) != hipSuccess) return check_launch("memset render_backward");
        const unsigned blocks = (unsigned)((P + MED_PTS_PER_WG - 1) / MED_PTS_PER_WG);
        hipLaunchKernelGGL(visible_scan_kernel, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist, vis_count, unsorted, grad_pts, grad_feat, C, (rows < S || cyc) ? 1 : 0);
        hipLaunchKernelGGL(median_hist_kernel<1>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
        hipLaunchKernelGGL(median_hist_kernel<2>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
        hipLaunchKernelGGL(median_final_kernel, dim3(N), dim3(MED_THREADS), 0, st, hist, N, radii_s, rs);
        if (own && !own_given)   // (the band's own plane above is not read in owner mode: its region holds the full-layout one)
            hipLaunchKernelGGL(alpha_rows_kernel, dim3((unsigned)S, (unsigned)N), dim3(256), 0, st, grad_full, plane, rs, N, S, C,
                               row0, rows, tshift);
        const unsigned cb = (unsigned)cell_blocks(P);   // (the visible count is only known on the device: P bounds it)
        const size_t lds = (size_t)cg.total * 4;
        const bool band_l = rows < S || cyc;   // row band: the entries that cannot reach it drop out of the sorted list
        hipLaunchKernelGGL(cell_hist_kernel, dim3(cb), dim3(CELL_THREADS), lds, st, points, first_idx, num_pts, N, S, cg,
                           vis_count, unsorted, cell_of, block_hist, band_l ? radii : nullptr, rs, row0, rows, tshift, own);
        hipLaunchKernelGGL(cell_block_scan_kernel, dim3((unsigned)((cg.total + 255) / 256)), dim3(256), 0, st, vis_count,
                           cg.total, block_hist, cell_total);
        hipLaunchKernelGGL(cell_scan_kernel, dim3(1), dim3(1024), 0, st, cell_total, cell_start, cg.total, vis_count + 1);
        hipLaunchKernelGGL(cell_scatter_kernel, dim3(cb), dim3(CELL_THREADS), lds, st, cg, vis_count, unsorted, cell_of,
                           cell_start, block_hist, sorted);
        }
        if (own && !own_given) { OW.alpha = plane; OW.astride = 1; }
        vis_count += 1;   // the gather walks the SORTED list: its length (written by cell_scan_kernel) is the second word
    }
    if (fused) {
        // the gather launch, with the preparation stages the mode puts inside it (run_prep false: the gather stage alone,
        // on what a preceding full call left in the workspace)
        // Its workgroups WAIT for the medians, so the grid must not exceed what is resident at once for THIS instantiation
        // (a late workgroup would start its two halves when the others end theirs): capacity per variant, cached in
        // slots 4.. of the device cache.
        unsigned fgrid = pgrid;
        if (band) {
            // BAND variants: medians always inside the launch (run_prep false only skips the segment launch: the keys of the
            // preceding full call are still in the workspace), grid = resident capacity of the variant
            fgrid = band_grid(tpw);
#define DSS_LAUNCH_RB_B(CC, TT, YY)                                                                                         \
    hipLaunchKernelGGL((render_backward_kernel<CC, true, TT, true, YY, true, true>), dim3(fgrid), dim3(256), 0, st, grad_out, alpha, idx, \
                       qvalue, wsum, scaler, points, radii, rs, first_idx, num_pts, vis_count, vis_list, n_seg, seg_pts, N, S, K, C, \
                       clip, row0, rows, large_waves, grad_feat, grad_pts, tshift, nullptr, nullptr, FP, astride, OW)
            if (C == 3 && cyc) { if (tpw == 4) DSS_LAUNCH_RB_B(3, 4, true); else if (tpw == 2) DSS_LAUNCH_RB_B(3, 2, true); else DSS_LAUNCH_RB_B(3, 1, true); }
            else if (C == 3) { if (tpw == 4) DSS_LAUNCH_RB_B(3, 4, false); else if (tpw == 2) DSS_LAUNCH_RB_B(3, 2, false); else DSS_LAUNCH_RB_B(3, 1, false); }
            else { if (tpw == 4) DSS_LAUNCH_RB_B(0, 4, false); else if (tpw == 2) DSS_LAUNCH_RB_B(0, 2, false); else DSS_LAUNCH_RB_B(0, 1, false); }
#undef DSS_LAUNCH_RB_B
            return check_launch("dss_render_backward");
        }
        if (run_prep && fused_opt != 5) {
            const int slot = 4 + (C == 3 ? 0 : 3) + (tpw == 4 ? 2 : (tpw == 2 ? 1 : 0));
            int fcap = dc ? dc[slot].load(std::memory_order_relaxed) : 0;
            if (fcap == 0) {
                int per_cu = 0;
#define DSS_OCC_F(CC, TT) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<CC, true, TT, true, false, true>, 256, 0)
                if (C == 3) { if (tpw == 4) DSS_OCC_F(3, 4); else if (tpw == 2) DSS_OCC_F(3, 2); else DSS_OCC_F(3, 1); }
                else { if (tpw == 4) DSS_OCC_F(0, 4); else if (tpw == 2) DSS_OCC_F(0, 2); else DSS_OCC_F(0, 1); }
#undef DSS_OCC_F
                (void)hipGetLastError();
                fcap = n_cus * (per_cu > 0 ? per_cu : 1);
                if (dc) dc[slot].store(fcap, std::memory_order_relaxed);
            }
            if ((unsigned)fcap < fgrid) fgrid = (unsigned)fcap;
            if (fgrid < (unsigned)N + 8u) fgrid = (unsigned)N + 8u;
        }
#define DSS_LAUNCH_RB_F(CC, TT, PP)                                                                                         \
    hipLaunchKernelGGL((render_backward_kernel<CC, true, TT, true, false, PP>), dim3(fgrid), dim3(256), 0, st, grad_out, alpha, idx, \
                       qvalue, wsum, scaler, points, radii, rs, first_idx, num_pts, vis_count, vis_list, n_seg, seg_pts, N, S, K, C, \
                       clip, row0, rows, large_waves, grad_feat, grad_pts, 3, world, Mproj, FP, astride, OW)
#define DSS_LAUNCH_RB_FT(CC, PP)                                                                                            \
    do {                                                                                                                   \
        if (tpw == 4) DSS_LAUNCH_RB_F(CC, 4, PP); else if (tpw == 2) DSS_LAUNCH_RB_F(CC, 2, PP); else DSS_LAUNCH_RB_F(CC, 1, PP); \
    } while (0)
        if (run_prep && fused_opt != 5) { if (C == 3) DSS_LAUNCH_RB_FT(3, true); else DSS_LAUNCH_RB_FT(0, true); }
        else { if (C == 3) DSS_LAUNCH_RB_FT(3, false); else DSS_LAUNCH_RB_FT(0, false); }
#undef DSS_LAUNCH_RB_FT
#undef DSS_LAUNCH_RB_F
        return check_launch("dss_render_backward");
    }
    // The grid is persistent and was sized for the variant the device cache measured (<C, true, 4, true>: 79 VGPRs, six
    // workgroups per CU); some variants hold more registers (two tasks per wavefront without the fused preparation: 83, the
    // 64-bit addressing: 100) -- a workgroup that is not resident at launch would start its share when the others end theirs,
    // so every launch is clamped to what ITS instantiation keeps resident (occupancy queried once per instantiation).
    auto resident_grid = [&](const void *fn) -> unsigned {
        static std::atomic<const void *> keys[64];
        static std::atomic<int> vals[64];
        int per_cu = 0;
        for (int i = 0; i < 64; ++i) {
            const void *k = keys[i].load(std::memory_order_acquire);
            if (k == fn) { per_cu = vals[i].load(std::memory_order_relaxed); break; }
            if (k == nullptr) {
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0);
                (void)hipGetLastError();
                if (per_cu < 1) per_cu = 1;
                const void *expect = nullptr;
                if (keys[i].compare_exchange_strong(expect, fn, std::memory_order_acq_rel)) vals[i].store(per_cu, std::memory_order_relaxed);
                break;   // (lost the race for the slot: this call uses its own query, a later one finds the entry or the next slot)
            }
        }
        if (per_cu < 1) return pgrid;
        const unsigned g = (unsigned)n_cus * (unsigned)per_cu;
        return g < pgrid ? g : pgrid;
    };
#define DSS_LAUNCH_RB_A(CC, SS, TT, AA)                                                                                 \
    hipLaunchKernelGGL((render_backward_kernel<CC, SS, TT, AA>), dim3(resident_grid((const void *)render_backward_kernel<CC, SS, TT, AA>)), dim3(256), 0, st, grad_out, alpha, idx, qvalue, wsum, \
                       scaler, points, radii, rs, first_idx, num_pts, vis_count, vis_list, n_seg, seg_pts, N, S, K, C, clip, \
                       row0, rows, large_waves, grad_feat, grad_pts, 3, world, Mproj, FusedPrep(), 1, OW)
#define DSS_LAUNCH_RB(CC, SS, TT) DSS_LAUNCH_RB_A(CC, SS, TT, true)
#define DSS_LAUNCH_RB_T(CC, SS)                                                                                        \
    do {                                                                                                               \
        if (!a32) DSS_LAUNCH_RB_A(CC, SS, 4, false);                                                                   \
        else if (tpw == 4) DSS_LAUNCH_RB(CC, SS, 4); else if (tpw == 2) DSS_LAUNCH_RB(CC, SS, 2); else DSS_LAUNCH_RB(CC, SS, 1); \
    } while (0)
    if (cyc) {
        // tile-row-cyclic band (multi-GPU): built for the training configuration (RGB features, 32-bit offsets)
        if (C != 3 || !a32) {
            set_error("dss_render_backward: a tile-row-cyclic band needs C == 3 and gathered tensors below 4 GB");
            return DSS_ERR_UNSUPPORTED;
        }
#define DSS_LAUNCH_RB_C(SS, TT)                                                                                          \
    hipLaunchKernelGGL((render_backward_kernel<3, SS, TT, true, true>), dim3(resident_grid((const void *)render_backward_kernel<3, SS, TT, true, true>)), dim3(256), 0, st, grad_out, alpha, idx, qvalue, \
                       wsum, scaler, points, radii, rs, first_idx, num_pts, vis_count, vis_list, n_seg, seg_pts, N, S, K, C, clip, \
                       row0, rows, large_waves, grad_feat, grad_pts, tshift, nullptr, nullptr, FusedPrep(), 1, OW)
        if (small) { if (tpw == 4) DSS_LAUNCH_RB_C(true, 4); else if (tpw == 2) DSS_LAUNCH_RB_C(true, 2); else DSS_LAUNCH_RB_C(true, 1); }
        else { if (tpw == 4) DSS_LAUNCH_RB_C(false, 4); else if (tpw == 2) DSS_LAUNCH_RB_C(false, 2); else DSS_LAUNCH_RB_C(false, 1); }
#undef DSS_LAUNCH_RB_C
    } else if (C == 3) {
        if (small) DSS_LAUNCH_RB_T(3, true); else DSS_LAUNCH_RB_T(3, false);
    } else {
        if (small) DSS_LAUNCH_RB_T(0, true); else DSS_LAUNCH_RB_T(0, false);
    }
#undef DSS_LAUNCH_RB_T
#undef DSS_LAUNCH_RB
#undef DSS_LAUNCH_RB_A
    return check_launch("dss_render_backward");
}

extern "C" int dss_render_backward(const float *grad_out, const int32_t *idx, const float *qvalue, const float *wsum,
                                   const float *scaler, const float *points, const float *radii,
                                   const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                   int64_t P, int S, int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                   float *grad_feat, float *grad_pts, float *rs_out, const float *world, const float *M, void *workspace,
                                   size_t workspace_bytes, void *stream)
{
    return render_backward_impl(true, grad_out, idx, qvalue, wsum, scaler, points, radii, visible, first_idx, num_pts, N, P, S,
                                K, C, row0, row1, row_cycle, radii_s, clip, grad_feat, grad_pts, rs_out, world, M, workspace, workspace_bytes, stream);
}

// Owner mode of a row band (see OwnArgs; include/dss_hip.h): grad_out_full = the image gradient of ALL rows
extern "C" int dss_render_backward_owned(const float *grad_out, const float *grad_out_full, const int32_t *idx, const float *qvalue,
                                         const float *wsum, const float *scaler, const float *points, const float *radii,
                                         const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                         int64_t P, int S, int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                         float *grad_feat, float *grad_pts, float *rs_out, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    if (!grad_out_full) { set_error("dss_render_backward_owned: grad_out_full is NULL"); return DSS_ERR_INVALID_ARGUMENT; }
    return render_backward_impl(true, grad_out, idx, qvalue, wsum, scaler, points, radii, visible, first_idx, num_pts, N, P, S,
                                K, C, row0, row1, row_cycle, radii_s, clip, grad_feat, grad_pts, rs_out, nullptr, nullptr, workspace,
                                workspace_bytes, stream, grad_out_full);
}

// Owner mode fed by the dense plane of the occupancy gradient (N,S,S) -- what a rank holds after an all-gather of the alpha
// channel of the band-local loss gradients (dss_gather_rows puts the gathered rows in image order)
extern "C" int dss_render_backward_owned_plane(const float *grad_out, const float *grad_occ_full, const int32_t *idx, const float *qvalue,
                                               const float *wsum, const float *scaler, const float *points, const float *radii,
                                               const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                               int64_t P, int S, int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                               float *grad_feat, float *grad_pts, float *rs_out, void *workspace, size_t workspace_bytes,
                                               void *stream)
{
    if (!grad_occ_full) { set_error("dss_render_backward_owned_plane: grad_occ_full is NULL"); return DSS_ERR_INVALID_ARGUMENT; }
    return render_backward_impl(true, grad_out, idx, qvalue, wsum, scaler, points, radii, visible, first_idx, num_pts, N, P, S,
                                K, C, row0, row1, row_cycle, radii_s, clip, grad_feat, grad_pts, rs_out, nullptr, nullptr, workspace,
                                workspace_bytes, stream, nullptr, grad_occ_full);
}

// Second stage alone (the persistent gather kernel), on the workspace (visible lists, alpha plane, rs) and the zero-filled
// gradients a preceding dss_render_backward call with the SAME arguments left behind.  For per-kernel timing (bench.py).
extern "C" int dss_render_backward_gather(const float *grad_out, const int32_t *idx, const float *qvalue, const float *wsum,
                                          const float *scaler, const float *points, const float *radii,
                                          const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                          int64_t P, int S, int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                          float *grad_feat, float *grad_pts, float *rs_out, const float *world, const float *M, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    return render_backward_impl(false, grad_out, idx, qvalue, wsum, scaler, points, radii, visible, first_idx, num_pts, N, P, S,
                                K, C, row0, row1, row_cycle, radii_s, clip, grad_feat, grad_pts, rs_out, world, M, workspace, workspace_bytes, stream);
}

#ifdef DSS_FINE_TIMING
extern "C" __attribute__((visibility("default"))) int dss_debug_set_occ_timing(long long *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(dss::g_occ_timing), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
