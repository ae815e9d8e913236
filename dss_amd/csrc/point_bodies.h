// Per-point gather bodies shared by the stand-alone backward kernels (one wavefront per packed point)
// and by the fused persistent backward kernel (wavefronts loop over the compacted visible list).
// Both are executed by a full wavefront for ONE point; results are per-lane partial sums that the
// caller reduces with wave_sum().
#pragma once
#include "common.h"

namespace dss {

#define BLEND_MAX_C 8

// lane tiling of a W-column pixel window: LW columns x (64/LW) rows per sweep
struct LaneTiling {
    int LW, LH, lxx, lyy;
    __device__ __forceinline__ LaneTiling(int w, int lane)
    {
        const int lw_log = (w <= 8) ? 3 : (w <= 16) ? 4 : (w <= 32) ? 5 : 6;
        LW = 1 << lw_log;
        LH = 64 >> lw_log;
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
                                                 int gstride, float &gx, float &gy)
{
    const float px = R.px, py = R.py, pz = R.pz, rx = R.rx, ry = R.ry;
    const float cur_r2 = cur_r * cur_r;
    // rasterize_points_backward.cu:141-143
    if (pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f) return;
    int xlo, xhi, ylo, yhi;
    if (!ndc_index_range_tight(px, cur_r, S, xlo, xhi) || !ndc_index_range_tight(py, cur_r, S, ylo, yhi)) return;
    // band rows: image row = S-1-yi in [row0, row0+rows)
    ylo = max(ylo, S - row0 - rows);
    yhi = min(yhi, S - 1 - row0);
    if (ylo > yhi) return;
    const LaneTiling T(xhi - xlo + 1, lane);
    // wave-uniform image base (SGPR) + 32-bit element offsets (one band of one cloud is < 2^31 elements)
    const int n_u = __builtin_amdgcn_readfirstlane(n);
    const float *__restrict__ gimg = grad_occ + (size_t)n_u * rows * S * gstride;
    const int rowstride = S * gstride;
    const NdcMap ndc(S);
    for (int xi = xlo + T.lxx; xi <= xhi; xi += T.LW) {
        // column-invariant terms hoisted out of the row loop
        const float dx = ndc(xi) - px;
        const float dx2 = dx * dx;
        const bool out_x = fabsf(dx) > rx;
        const int coff = (S - 1 - xi) * gstride - row0 * rowstride;
        // RPT rows per trip: the loads are independent and issue back to back, so a 29-row window
        // (rs = 14 px) costs two memory round trips; offsets are unsigned 32-bit (saddr-form loads)
        constexpr int RPT = 8;
        for (int y0 = ylo + T.lyy; y0 <= yhi; y0 += RPT * T.LH) {
            // rows of this trip that exist for at least one lane (wave-uniform): small windows (rs of a few
            // pixels at high point density) must not pay for eight row slots
            const int yb = y0 - T.lyy;                       // uniform
            const int nrow = min(RPT, (yhi - yb) / T.LH + 1);  // uniform, >= 1
            float g[RPT];
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                g[u] = 0.0f;
                if (u < nrow) {
                    const int yc = min(y0 + u * T.LH, yhi);  // clamped: always a legal address
                    g[u] = gimg[(unsigned)((S - 1 - yc) * rowstride + coff)];
                }
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                if (u < nrow) {
                    const int yi = y0 + u * T.LH;
                    const float dy = ndc(yi) - py;
                    const float d2 = dx2 + dy * dy;
                    const bool outside = out_x || (fabsf(dy) > ry);
                    const bool use = (yi <= yhi) && (g[u] != 0.0f) && !(d2 > cur_r2) && !(g[u] > 0.0f && outside) &&
                                     (d2 != 0.0f);
                    // dx / max(d2,1e-10) * g with a 1-ulp reciprocal and fused accumulation (tolerance-checked,
                    // not bit-pinned: the reference accumulates with unordered fp32 atomics anyway)
                    const float sgl = use ? __builtin_amdgcn_rcpf(fmaxf(d2, 1e-10f)) * g[u] : 0.0f;
                    gx = fmaf(dx, sgl, gx);
                    gy = fmaf(dy, sgl, gy);
                }
            }
        }
    }
}

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
                    if (v >= 0) cum += expf(-0.5f * qv[pix * K + k]) * scaler[v];
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

}  // namespace dss
