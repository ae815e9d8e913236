// Per-point gather bodies shared by the stand-alone backward kernels (one wavefront per packed point)
// and by the fused persistent backward kernel (wavefronts loop over the compacted visible list).
// Both are executed by a full wavefront for ONE point; results are per-lane partial sums that the
// caller reduces with wave_sum().
#pragma once
#include <type_traits>
#include "common.h"

namespace dss {

#define BLEND_MAX_C 8

// lane tiling of a W-column pixel window: LW columns x (64/LW) rows per sweep
struct LaneTiling {
    int LW, LH, lh_log, lxx, lyy;
    __device__ __forceinline__ LaneTiling(int w, int lane)
    {
        const int lw_log = (w <= 8) ? 3 : (w <= 16) ? 4 : (w <= 32) ? 5 : 6;
        LW = 1 << lw_log;
        LH = 64 >> lw_log;
        lh_log = 6 - lw_log;  // divisions by LH are shifts (a runtime integer divide is ~25 instructions)
        lxx = lane & (LW - 1);
        lyy = lane >> lw_log;
    }
};

// Occupancy surrogate gradient of point p (cloud n) over the band rows [row0, row0+rows):
//   for every pixel with g = grad_occ != 0 and d2 = dx^2+dy^2 <= rs^2:
//       skip if g>0 and (|dx|>rx or |dy|>ry);  (gx,gy) += (dx,dy)/max(d2,1e-10)*g
// (rasterize_points_backward.cu:141-178; a pair with d2 == 0 contributes 0, see dss_hip.h).
// grad_occ is read with an element stride `gstride` per pixel (alpha channel of an image gradient).
// screen-space record of one splat as the gathers need it
struct SplatRec {
    float px, py, pz, rx, ry, sc;
};

__device__ __forceinline__ void occ_point_gather(int lane, int n, const SplatRec &R, float cur_r,
                                                 const float *__restrict__ grad_occ, int S, int row0, int rows,
                                                 int gstride, float &gx, float &gy);  // defined at the end of the file

__device__ __forceinline__ void occ_point_gather(int lane, int64_t p, int n, const float *__restrict__ points,
                                                 const float *__restrict__ radii, const float *__restrict__ rs,
                                                 const float *__restrict__ grad_occ, int S, int row0, int rows,
                                                 int gstride, float &gx, float &gy)
{
    SplatRec R;
    R.px = points[3 * p]; R.py = points[3 * p + 1]; R.pz = points[3 * p + 2];
    R.rx = radii[2 * p]; R.ry = radii[2 * p + 1]; R.sc = 0.0f;
    occ_point_gather(lane, n, R, rs[n], grad_occ, S, row0, rows, gstride, gx, gy);
}

// Blend backward of point p: sum over the pixels of the point's own bounding box (a fragment with
// idx == p can only exist where the hit test passed) of grad_out * w / wsum.
template <int C>
__device__ __forceinline__ void blend_point_gather(int lane, int64_t p, int n, const SplatRec &R,
                                                   const float *__restrict__ grad_out,
                                                   const int32_t *__restrict__ idx, const float *__restrict__ qv,
                                                   const float *__restrict__ wsum, const float *__restrict__ scaler,
                                                   int S, int K, int Cn, int row0, int rows,
                                                   float (&acc)[(C > 0) ? C : BLEND_MAX_C])
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    const float px = R.px, py = R.py, rx = R.rx, ry = R.ry, sc = R.sc;
    int xlo, xhi, ylo, yhi;
    if (!ndc_index_range_tight(px, rx, S, xlo, xhi) || !ndc_index_range_tight(py, ry, S, ylo, yhi)) return;
    ylo = max(ylo, S - row0 - rows);
    yhi = min(yhi, S - 1 - row0);
    const LaneTiling T(xhi - xlo + 1, lane);
    for (int yi = ylo + T.lyy; yi <= yhi; yi += T.LH) {
        const size_t rowbase = ((size_t)n * rows + (S - 1 - yi - row0)) * S;
        for (int xi = xlo + T.lxx; xi <= xhi; xi += T.LW) {
            const size_t pix = rowbase + (S - 1 - xi);
            const int32_t *pi = idx + pix * K;
            // loads that do not depend on the slot search are issued first (one round trip for all)
            const float *go = grad_out + pix * (Cn + 1);
            float gch[CM];
#pragma unroll
            for (int ch = 0; ch < CM; ++ch) gch[ch] = (ch < Cn) ? go[ch] : 0.0f;
            float cum = wsum ? wsum[pix] : 0.0f;
            int kk = -1;
            for (int k = 0; k < K; ++k) {
                const int32_t v = pi[k];
                if (v == (int32_t)p) kk = k;
            }
            if (kk < 0) continue;
            if (!wsum) {
                for (int k = 0; k < K; ++k) {
                    const int32_t v = pi[k];
                    if (v >= 0) cum += ewa_weight(qv[pix * K + k], scaler[v]);
                }
                if (cum < 1e-4f) cum = 1e-4f;
            }
            // exp(-q/2) = 2^(-q/2 * log2 e): v_exp_f32 (~1e-6 rel.) is ample for a gradient checked at 1e-3
            const float wgt = __builtin_amdgcn_exp2f(-0.72134752f * qv[pix * K + kk]) * sc;
            const float wn = wgt * __builtin_amdgcn_rcpf(cum);
#pragma unroll
            for (int ch = 0; ch < CM; ++ch)
                if (ch < Cn) acc[ch] = fmaf(gch[ch], wn, acc[ch]);
        }
    }
}

template <int C>
__device__ __forceinline__ void blend_point_gather(int lane, int64_t p, int n, const float *__restrict__ grad_out,
                                                   const int32_t *__restrict__ idx, const float *__restrict__ qv,
                                                   const float *__restrict__ wsum, const float *__restrict__ scaler,
                                                   const float *__restrict__ points, const float *__restrict__ radii,
                                                   int S, int K, int Cn, int row0, int rows,
                                                   float (&acc)[(C > 0) ? C : BLEND_MAX_C])
{
    SplatRec R;
    R.px = points[3 * p]; R.py = points[3 * p + 1]; R.pz = 0.0f;
    R.rx = radii[2 * p]; R.ry = radii[2 * p + 1]; R.sc = scaler[p];
    blend_point_gather<C>(lane, p, n, R, grad_out, idx, qv, wsum, scaler, S, K, Cn, row0, rows, acc);
}

// Both gathers of one visible point in the fused backward, arranged so that their memory round trips overlap.
// A splat's own bounding box is almost always a single sweep of the wavefront (<= 64 pixels in the lane tiling):
// in that case the blend loads (K fragment ids and Q values, the image gradient, the weight sum) are issued
// FIRST, then the first trip of the occupancy window; the fragment-slot search runs when the first trip has
// arrived (loads return in order), the occupancy arithmetic follows.  Sequentially the two gathers cost four to
// five dependent round trips per point; like this the blend gather adds none.  Same arithmetic, same
// accumulation order per lane as occ_point_gather + blend_point_gather (checked bit for bit against the
// unfused kernels in tests/test_gpu_raster.py).
// `mid` is called once, right after the first loads of the point have been issued (the persistent kernel uses
// it to request the next task's record, which then arrives during this task's arithmetic).
template <int C, typename Mid>
__device__ __forceinline__ void occ_blend_point_gather(int lane, int64_t p, int n, const SplatRec &R, float cur_r,
                                                       const float *__restrict__ grad_occ, int gstride,
                                                       const float *__restrict__ grad_out,
                                                       const int32_t *__restrict__ idx, const float *__restrict__ qv,
                                                       const float *__restrict__ wsum,
                                                       const float *__restrict__ scaler, int S, int K, int Cn, int row0,
                                                       int rows, bool want_blend, float &gx, float &gy,
                                                       float (&acc)[(C > 0) ? C : BLEND_MAX_C], Mid mid)
{
    constexpr int CM = (C > 0) ? C : BLEND_MAX_C;
    constexpr int KF = 8;  // fragment slots the single-sweep path keeps in registers
    const float px = R.px, py = R.py, pz = R.pz, rx = R.rx, ry = R.ry;
    // ---- blend window; single-sweep prefetch ----
    int bxlo = 0, bxhi = -1, bylo = 0, byhi = -1;
    bool b_any = want_blend && ndc_index_range_tight(px, rx, S, bxlo, bxhi) && ndc_index_range_tight(py, ry, S, bylo, byhi);
    if (b_any) {
        bylo = max(bylo, S - row0 - rows);
        byhi = min(byhi, S - 1 - row0);
        b_any = bylo <= byhi;
    }
    const LaneTiling TB(b_any ? bxhi - bxlo + 1 : 1, lane);
    const bool b_single = b_any && wsum != nullptr && K <= KF && (bxhi - bxlo + 1 <= TB.LW) && (byhi - bylo + 1 <= TB.LH);
    int32_t vi[KF];
    float qk[KF], gch[CM], cum = 1.0f;
    bool b_in = false;
#pragma unroll
    for (int k = 0; k < KF; ++k) { vi[k] = -1; qk[k] = 0.0f; }
#pragma unroll
    for (int ch = 0; ch < CM; ++ch) gch[ch] = 0.0f;
    if (b_single) {
        const int yi = bylo + TB.lyy, xi = bxlo + TB.lxx;
        b_in = yi <= byhi && xi <= bxhi;
        if (b_in) {
            const size_t pix = ((size_t)n * rows + (S - 1 - yi - row0)) * S + (S - 1 - xi);
#pragma unroll
            for (int k = 0; k < KF; ++k)
                if (k < K) {
                    vi[k] = idx[pix * K + k];
                    qk[k] = qv[pix * K + k];
                }
            const float *go = grad_out + pix * (Cn + 1);
#pragma unroll
            for (int ch = 0; ch < CM; ++ch) gch[ch] = (ch < Cn) ? go[ch] : 0.0f;
            cum = wsum[pix];
        }
    }
    // ---- occupancy window (rasterize_points_backward.cu:141-178), see occ_point_gather ----
    const float cur_r2 = cur_r * cur_r;
    int xlo = 0, xhi = -1, ylo = 0, yhi = -1;
    bool o_ok = !(pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f) && ndc_index_range_tight(px, cur_r, S, xlo, xhi) &&
                ndc_index_range_tight(py, cur_r, S, ylo, yhi);
    if (o_ok) {
        ylo = max(ylo, S - row0 - rows);
        yhi = min(yhi, S - 1 - row0);
        o_ok = ylo <= yhi;
    }
    const LaneTiling T(o_ok ? xhi - xlo + 1 : 1, lane);
    const int n_u = __builtin_amdgcn_readfirstlane(n);
    // grad_occ: occupancy gradient with an element stride `gstride` per pixel (1 = dense plane)
    const float *__restrict__ gimg = grad_occ + (size_t)n_u * rows * S * gstride;
    const int rowstride = S * gstride;
    const NdcMap ndc(S);
    constexpr int RPT = 8;
    float g[RPT];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 gx2 = {0.0f, 0.0f}, gy2 = {0.0f, 0.0f};  // even / odd row slots (packed fp32 pipes), summed at the end
    // one trip = up to RPT row slots of the lane tiling, columns xb + lxx, rows yb + lyy + u * LH.
    // The whole kernel is bound by VALU issue (~800 wave instructions per point, tools/render_bwd_timing.py),
    // so the per-slot arithmetic is pared down: out-of-window slots are zeroed once at the load (g = 0 makes
    // the contribution 0 without further tests, as does dx = dy = 0 for the d2 == 0 pair), pixel centres
    // advance by an exact increment when S is a power of two, and two row slots share one packed instruction.
    auto trip_load = [&](int xb, int yb) {
        const int xi = xb + T.lxx;
        const bool col_ok = xi <= xhi;
        // row slot u of this trip is image row S-1-(yb + lyy + u LH): a wave-uniform row pointer (scalar
        // arithmetic) plus ONE per-lane offset for all slots; lanes outside the window are masked, not clamped
        const int lane_off = (S - 1 - min(xi, xhi)) * gstride - T.lyy * rowstride;
        const int nrow = min(RPT, ((yhi - yb) >> T.lh_log) + 1);  // uniform, >= 1
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            g[u] = 0.0f;
            if (u < nrow) {
                const float *__restrict__ rowp = gimg + (S - 1 - row0 - yb - u * T.LH) * rowstride;  // uniform
                if (col_ok && yb + T.lyy + u * T.LH <= yhi) g[u] = rowp[lane_off];
            }
        }
    };
    auto trip_compute_t = [&](int xb, int yb, auto pow2_tag) {
        constexpr bool POW2 = decltype(pow2_tag)::value;
        const int xi = xb + T.lxx;
        const float dx = ndc(xi) - px;
        const float dx2 = dx * dx;
        // "g > 0 and outside the splat's box" (|dx| > rx or |dy| > ry): with ry_eff = -1 for out-of-box columns
        // the row test alone decides (|dy| > -1 is always true)
        const float ry_eff = (fabsf(dx) > rx) ? -1.0f : ry;
        const int nrow = min(RPT, ((yhi - yb) >> T.lh_log) + 1);
        const int yi0 = yb + T.lyy;
        const float nd0 = ndc(yi0);
        // pixel centres are integer multiples of 1/S: for S = 2^k, nd0 + u * (2 LH / S) is exact, i.e. ndc(yi0 + u LH)
        const float stepf = (float)(2 * T.LH) * ndc.invS;
#pragma unroll
        for (int u = 0; u < RPT; u += 2) {
            if (u < nrow) {
                f2 nd;
                if (POW2) {
                    nd.x = nd0 + (float)u * stepf;
                    nd.y = nd0 + (float)(u + 1) * stepf;
                } else {
                    nd.x = ndc(yi0 + u * T.LH);
                    nd.y = ndc(yi0 + (u + 1) * T.LH);
                }
                const f2 gg = {g[u], g[u + 1]};
                const f2 dy = nd - py;
                const f2 d2 = dx2 + dy * dy;
                // bitwise, not short-circuit: three compares and two mask operations per slot
                const bool skip0 = (d2.x > cur_r2) | ((gg.x > 0.0f) & (fabsf(dy.x) > ry_eff));
                const bool skip1 = (d2.y > cur_r2) | ((gg.y > 0.0f) & (fabsf(dy.y) > ry_eff));
                // (dx, dy) / max(d2, 1e-10) * g with a 1-ulp reciprocal and fused accumulation (tolerance-checked,
                // not bit-pinned: the reference accumulates with unordered fp32 atomics anyway)
                const f2 rc = {__builtin_amdgcn_rcpf(fmaxf(d2.x, 1e-10f)), __builtin_amdgcn_rcpf(fmaxf(d2.y, 1e-10f))};
                f2 sgl = rc * gg;
                sgl.x = skip0 ? 0.0f : sgl.x;
                sgl.y = skip1 ? 0.0f : sgl.y;
                const f2 dxx = {dx, dx};
                gx2 = __builtin_elementwise_fma(dxx, sgl, gx2);
                gy2 = __builtin_elementwise_fma(dy, sgl, gy2);
            }
        }
    };
    auto trip_compute = [&](int xb, int yb) {
        if (ndc.pow2) trip_compute_t(xb, yb, std::true_type{});
        else trip_compute_t(xb, yb, std::false_type{});
    };
    if (o_ok) trip_load(xlo, ylo);
    mid();
    // ---- fragment-slot search of the prefetched blend pixel ----
    float q_sel = 0.0f;
    bool found = false;
    if (b_single) {
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const bool hit = (k < K) && vi[k] == (int32_t)p;
            q_sel = hit ? qk[k] : q_sel;
            found = found || hit;
        }
    }
    if (o_ok) {
        trip_compute(xlo, ylo);
        for (int xb = xlo; xb <= xhi; xb += T.LW) {
            for (int yb = (xb == xlo) ? ylo + RPT * T.LH : ylo; yb <= yhi; yb += RPT * T.LH) {
                trip_load(xb, yb);
                trip_compute(xb, yb);
            }
        }
    }
    gx += gx2.x + gx2.y;
    gy += gy2.x + gy2.y;
    // ---- blend finish ----
    if (b_single) {
        if (b_in && found) {
            const float wgt = __builtin_amdgcn_exp2f(-0.72134752f * q_sel) * R.sc;
            const float wn = wgt * __builtin_amdgcn_rcpf(cum);
#pragma unroll
            for (int ch = 0; ch < CM; ++ch)
                if (ch < Cn) acc[ch] = fmaf(gch[ch], wn, acc[ch]);
        }
    } else if (b_any) {
        blend_point_gather<C>(lane, p, n, R, grad_out, idx, qv, wsum, scaler, S, K, Cn, row0, rows, acc);
    }
}

// Occupancy gather alone (stand-alone kernels): the same code path as the fused gather, hence the same bits.
__device__ __forceinline__ void occ_point_gather(int lane, int n, const SplatRec &R, float cur_r,
                                                 const float *__restrict__ grad_occ, int S, int row0, int rows,
                                                 int gstride, float &gx, float &gy)
{
    float acc[BLEND_MAX_C];
    occ_blend_point_gather<0>(lane, 0, n, R, cur_r, grad_occ, gstride, nullptr, nullptr, nullptr, nullptr, nullptr, S, 1,
                              0, row0, rows, false, gx, gy, acc, []() {});
}

}  // namespace dss
