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
    float *__restrict__ grad_feat, int C)
{
    __shared__ uint32_t s_w[PER * PREP_THREADS / 64];
    constexpr int SEG_PTS = PER * PREP_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const unsigned seg = blockIdx.x;
    const uint32_t count = seg_count[seg];
    const float band_lo = -1 + (2 * (S - row0 - rows)) / (float)S;      // lower edge of the lowest pixel row
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
            const float reach = fmaxf(n >= 0 ? rs[n] : 0.0f, ry);
            const bool in_band = n >= 0 && !(py + reach < band_lo || py - reach > band_hi);
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

template <int C, bool SEG>
__global__ __launch_bounds__(256) void render_backward_kernel(
    const float *__restrict__ grad_out, const float *__restrict__ grad_alpha /* dense (N,rows,S) */,
    const int32_t *__restrict__ idx, const float *__restrict__ qv,
    const float *__restrict__ wsum, const float *__restrict__ scaler, const float *__restrict__ points,
    const float *__restrict__ radii, const float *__restrict__ rs, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, const uint32_t *__restrict__ vis_count,
    const int32_t *__restrict__ vis_list, int n_seg, int seg_pts, int N, int S, int K, int Crt, float clip, int row0,
    int rows, uint32_t large_waves, float *__restrict__ grad_feat, float *__restrict__ grad_pts)
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    const int Cn = (C > 0) ? C : Crt;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t n_waves = gridDim.x * 4;
    // SEG: vis_count[0..n_seg) are per-segment counts written by backward_compact_kernel (n_seg <= 64): every
    // wavefront scans them in registers once; a task index t maps to (segment, offset) with one ballot.
    uint32_t count, seg_excl = 0;  // first task index of segment `lane`
    if (SEG) {
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
    // Long lists use 6 workgroups per CU: more resident gathers only evict each other's image rows from L2
    // (8 x 1M points at 1024^2: 1169 Msplats/s with 6 per CU, 1058 with 7).  Short lists (a few tasks per
    // wavefront) use every resident wavefront; equalising the task counts (ceil(count / rounds) wavefronts)
    // was measured and is slower (329 vs 348 Msplats/s on the 512^2 bunny).
    if (count > 8u * n_waves) n_waves = min(n_waves, large_waves);
    if (wave >= n_waves) return;
#ifdef DSS_FINE_TIMING
    long long tm_rt0 = __builtin_amdgcn_s_memrealtime(), tm_occ = 0, tm_blend = 0, tm_pro = 0, tm_tasks = 0;
#endif
    // One task = one visible point, handled by the whole wavefront.  Everything that identifies the task is
    // wave-uniform, so it is kept in SGPRs (readfirstlane) and fetched with scalar loads; the NEXT task's
    // point id and record are requested before the current task's gathers and arrive during them (the
    // prologue was ~10 % of a task: two dependent round trips before the first gather load could issue).
    // Cloud of a point without touching memory: every wavefront keeps (first, count, rs) of cloud `lane` in its
    // lanes; the owner is found with one ballot.  (The scalar-load loop of find_cloud cost ~2 us per TASK at
    // 8 clouds: two dependent loads per cloud.)  More than 64 clouds fall back to the loop.
    const int64_t cl_first = lane < N ? first_idx[lane] : (int64_t)0x7fffffffffffffffll;
    const int64_t cl_count = lane < N ? num_pts[lane] : 0;
    const float cl_rs = lane < N ? rs[lane] : 0.0f;
    auto cloud_of = [&](int64_t p, float &rs_n) -> int {
        int n;
        if (N <= 64) {
            const unsigned long long own = __ballot(p >= cl_first && p < cl_first + cl_count);
            n = own ? (int)__builtin_ctzll(own) : -1;
        } else {
            n = find_cloud(p, first_idx, num_pts, N);
        }
        rs_n = (n >= 0 && N <= 64) ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cl_rs), n < 0 ? 0 : n))
                                   : (n >= 0 ? rs[n] : 0.0f);
        return n;
    };
    auto task_point = [&](uint32_t t) -> int {
        if (SEG) {
            // last segment that starts at or before t (starts are non-decreasing over lanes; empty segments
            // share their successor's start and lose the tie)
            const int seg = (int)__popcll(__ballot(seg_excl <= t)) - 1;
            const uint32_t excl = (uint32_t)__builtin_amdgcn_readlane((int)seg_excl, seg);
            return vis_list[(size_t)seg * seg_pts + (t - excl)];
        }
        return vis_list[t];
    };
    // the six floats of a record come from three arrays: ONE vector load with a different address per lane
    // (six scalar loads per task throttle on the scalar cache once tasks are short, measured on 8 x 1M points)
    auto issue_rec = [&](int p) -> float {
        const float *a = lane < 3 ? points + 3 * (size_t)p + lane
                       : lane < 5 ? radii + 2 * (size_t)p + (lane - 3)
                                  : (scaler ? scaler + p : points + 3 * (size_t)p);
        return lane < 6 ? *a : 0.0f;
    };
    auto unpack_rec = [&](float v, SplatRec &R) {
        R.px = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
        R.py = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 1));
        R.pz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 2));
        R.rx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 3));
        R.ry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 4));
        R.sc = scaler ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 5)) : 0.0f;
    };
    // Static schedule: task t = wave, wave + n_waves, ...  (A dynamic one -- 64 atomic queue heads, one returning
    // atomic per task, next position prefetched -- was measured at HALF the speed on every configuration: a
    // returning global atomic is slower than a whole gather trip and sits in the same in-order return queue as
    // the gather loads.)  Software pipeline, three tasks deep: during task i the point id of task i+3 (scalar load)
    // and the records of tasks i+1 and i+2 are in flight.  A record is requested a full task before it is
    // needed: in a row band (multi-GPU) most tasks are rejected without any gather, and with a shallower
    // pipeline each of those waited one memory round trip for the next record.
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    uint32_t t = wave_u;
    if (t >= count) return;
    int p_cur = __builtin_amdgcn_readfirstlane(task_point(t));
    int p_nx = (t + n_waves < count) ? __builtin_amdgcn_readfirstlane(task_point(t + n_waves)) : 0;
    int p_nx2 = (t + 2 * n_waves < count) ? __builtin_amdgcn_readfirstlane(task_point(t + 2 * n_waves)) : 0;
    SplatRec cur;
    unpack_rec(issue_rec(p_cur), cur);
    float v_nx = (t + n_waves < count) ? issue_rec(p_nx) : 0.0f;
    for (;;) {
#ifdef DSS_FINE_TIMING
        const long long tm0 = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t t_next = t + n_waves;
        const bool have_next = t_next < count;
        const bool have_next2 = t_next + n_waves < count;
        int p_nx3 = 0;
        float v_nx2 = 0.0f;
        if (t_next + 2 * n_waves < count) p_nx3 = task_point(t_next + 2 * n_waves);
        const int64_t p = p_cur;
        float rs_n;
        const int n = cloud_of(p, rs_n);
        float gx = 0.0f, gy = 0.0f;
        float acc[CM];
#pragma unroll
        for (int ch = 0; ch < CM; ++ch) acc[ch] = 0.0f;
#ifdef DSS_FINE_TIMING
        const long long tm1 = __builtin_amdgcn_s_memtime();
#endif
        auto mid = [&]() {
            if (have_next2) v_nx2 = issue_rec(p_nx2);
        };
        // Row band (multi-GPU): most visible points cannot reach this rank's rows (7 of 8 at 8 ranks).  One
        // conservative test on the wave-uniform record skips both gathers for them (~300 of the ~400 fixed
        // instructions of a task); the reductions then sum zeros and the point's partial is written as zero.
        bool in_band = true;
        if (rows < S && n >= 0) {
            const float reach = fmaxf(rs_n, cur.ry);
            const float band_lo = -1 + (2 * (S - row0 - rows)) / (float)S;      // lower edge of the lowest pixel row
            const float band_hi = -1 + (2 * (S - 1 - row0) + 2.0f) / (float)S;  // upper edge of the highest one
            in_band = !(cur.py + reach < band_lo || cur.py - reach > band_hi);
        }
        if (n >= 0 && in_band) {
            // occupancy gradient = dense copy of the alpha channel of the image gradient; blend loads overlapped
            occ_blend_point_gather<C>(lane, p, n, cur, rs_n, grad_alpha, 1, grad_out, idx, qv, wsum, scaler, S, K, Cn,
                                      row0, rows, grad_feat != nullptr, gx, gy, acc, mid);
        } else {
            mid();
        }
#ifdef DSS_FINE_TIMING
        const long long tm2 = __builtin_amdgcn_s_memtime();
        tm_pro += tm1 - tm0; tm_occ += tm2 - tm1; tm_tasks += 1;
#endif
        gx = wave_sum(gx);
        gy = wave_sum(gy);
#pragma unroll
        for (int ch = 0; ch < CM; ++ch)
            if (ch < Cn) acc[ch] = wave_sum(acc[ch]);
#ifdef DSS_FINE_TIMING
        tm_blend += (long long)__builtin_amdgcn_s_memtime() - tm2;
#endif
        if (lane == 0 && n >= 0) {
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
        if (!have_next) break;
        t = t_next;
        p_cur = p_nx;
        p_nx = p_nx2;
        p_nx2 = __builtin_amdgcn_readfirstlane(p_nx3);
        unpack_rec(v_nx, cur);
        v_nx = v_nx2;
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

// Workspace of the two-launch preparation (P <= PREP_MAX_POINTS), relative to its own base.
struct PrepLayout {
    size_t seg_count, vis_list, vis_keys, chunk_hist, seg_range, rs, alpha, bytes;
    int chunks, per;  // segments, points per thread of the compaction kernel (segment = per * 1024 points)
};
static PrepLayout prep_layout(int N, int64_t P, int S)
{
    PrepLayout L;
    const size_t n = N > 0 ? N : 1, p = P > 0 ? (size_t)P : 1;
    L.per = prep_points_per_thread(P);
    const size_t seg = (size_t)L.per * PREP_THREADS;
    L.chunks = (int)((p + seg - 1) / seg);
    size_t off = 0;
    L.seg_count = off;  off += 256;                                              // PREP_MAX_SEG counters
    L.vis_list = off;   off += align_up(p * 4, 256);
    L.vis_keys = off;   off += align_up(p * 8, 256);
    L.chunk_hist = off; off += align_up(n * (size_t)L.chunks * 256 * 4, 256);
    L.seg_range = off;  off += align_up(n * (size_t)L.chunks * 8, 256);
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
    return align_up((size_t)3 * n * MED_BINS * 4 + 256, 256)  // histograms + visible counter
           + align_up((size_t)(P > 0 ? P : 1) * 4, 256)       // compacted visible list
           + align_up(n * 4, 256)                             // rs
           + align_up(n * s * s * 4, 256);                    // dense alpha-gradient plane
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
    const size_t npix = (size_t)N * (row1 - row0) * S;
    const float *alpha;
    if (((uintptr_t)grad_out & 15u) && C == 3) { set_error("dss_render_backward: grad_out must be 16-byte aligned"); return DSS_ERR_INVALID_ARGUMENT; }
    if (small) {
        const PrepLayout L = prep_layout(N, P, S);
        n_seg = L.chunks;
        seg_pts = L.per * PREP_THREADS;
        alpha = reinterpret_cast<const float *>(w + L.alpha);
        vis_count = reinterpret_cast<uint32_t *>(w + L.seg_count);
        vis_list = reinterpret_cast<int32_t *>(w + L.vis_list);
        rs = rs_out ? rs_out : reinterpret_cast<float *>(w + L.rs);
        launch_prep(radii, visible, first_idx, num_pts, N, P, radii_s, rs, w, L, grad_pts, grad_feat, C, grad_out, npix, st);
        if (row1 - row0 < S) {  // row band: keep only the points that can reach it
            if (L.per == 2)
                hipLaunchKernelGGL(band_filter_kernel<2>, dim3(L.chunks), dim3(PREP_THREADS), 0, st, points, radii, rs,
                                   first_idx, num_pts, N, S, row0, row1 - row0, vis_count, vis_list, grad_pts, grad_feat, C);
            else
                hipLaunchKernelGGL(band_filter_kernel<4>, dim3(L.chunks), dim3(PREP_THREADS), 0, st, points, radii, rs,
                                   first_idx, num_pts, N, S, row0, row1 - row0, vis_count, vis_list, grad_pts, grad_feat, C);
        }
    } else {
        const size_t hist_bytes = (size_t)3 * N * MED_BINS * 4;
        uint32_t *hist = reinterpret_cast<uint32_t *>(w);
        vis_count = reinterpret_cast<uint32_t *>(w + hist_bytes);
        size_t off = align_up(hist_bytes + 256, 256);
        vis_list = reinterpret_cast<int32_t *>(w + off);
        off += align_up((size_t)P * 4, 256);
        rs = rs_out ? rs_out : reinterpret_cast<float *>(w + off);
        off += align_up((size_t)N * 4, 256);
        float *plane = reinterpret_cast<float *>(w + off);
        alpha = plane;
        hipLaunchKernelGGL(alpha_plane_kernel, dim3((unsigned)((npix + ALPHA_PIX_PER_WG - 1) / ALPHA_PIX_PER_WG)), dim3(1024),
                           0, st, grad_out, plane, npix, C);
        if (hipMemsetAsync(hist, 0, hist_bytes + 256, st) != hipSuccess) return check_launch("memset render_backward");
        const unsigned blocks = (unsigned)((P + MED_PTS_PER_WG - 1) / MED_PTS_PER_WG);
        hipLaunchKernelGGL(visible_scan_kernel, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist, vis_count, vis_list, grad_pts, grad_feat, C);
        hipLaunchKernelGGL(median_hist_kernel<1>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
        hipLaunchKernelGGL(median_hist_kernel<2>, dim3(blocks), dim3(MED_THREADS), 0, st, radii, visible, first_idx,
                           num_pts, N, P, hist);
        hipLaunchKernelGGL(median_final_kernel, dim3(N), dim3(MED_THREADS), 0, st, hist, N, radii_s, rs);
    }
    // persistent grid = exactly the resident capacity of the chip for this kernel (a larger grid would
    // leave late workgroups waiting for slots while their share of the list sits idle)
    static int cap3 = 0, cap0 = 0, n_cus = 256;  // benign race: every thread computes the same value
    int &cap = (C == 3) ? cap3 : cap0;
    if (cap == 0) {
        int dev = 0, cus = 256, per_cu = 4;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (C == 3)
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<3, true>, 256, 0);
        else
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, render_backward_kernel<0, true>, 256, 0);
        if (per_cu < 1) per_cu = 1;
        n_cus = cus;
        cap = cus * per_cu;
        (void)hipGetLastError();
    }
    const unsigned pgrid = (unsigned)((P + 3) / 4 < cap ? (P + 3) / 4 : cap);
    const uint32_t large_waves = 6u * (uint32_t)n_cus * 4u;
#define DSS_LAUNCH_RB(CC, SS)                                                                                          \
    hipLaunchKernelGGL((render_backward_kernel<CC, SS>), dim3(pgrid), dim3(256), 0, st, grad_out, alpha, idx, qvalue, wsum, \
                       scaler, points, radii, rs, first_idx, num_pts, vis_count, vis_list, n_seg, seg_pts, N, S, K, C, clip, \
                       row0, row1 - row0, large_waves, grad_feat, grad_pts)
    if (C == 3) {
        if (small) DSS_LAUNCH_RB(3, true); else DSS_LAUNCH_RB(3, false);
    } else {
        if (small) DSS_LAUNCH_RB(0, true); else DSS_LAUNCH_RB(0, false);
    }
#undef DSS_LAUNCH_RB
    return check_launch("dss_render_backward");
}

#ifdef DSS_FINE_TIMING
extern "C" __attribute__((visibility("default"))) int dss_debug_set_occ_timing(long long *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(dss::g_occ_timing), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif
