/*
 * oracle/dss_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the DSS EWA surface-splatting hot path (OpenMP only on loops whose
 * iterations are independent by construction, so results do not depend on the thread count)
 * (SURVEY.md section 8a).  It is the checker for the HIP library in dss_amd/csrc; it is
 * never linked, imported or called by the product path (only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * Compile with -ffp-contract=off so fp32 expressions round exactly like the reference's CPU
 * build (x86-64 baseline: no FMA contraction).
 *
 * Parity pinning status (see oracle/README.md, tests/test_oracle_pinning.py):
 *   forward rasterizer      PINNED   bit-exact vs the compiled reference CPU naive path
 *                                    (oracle/_ref, DSS/csrc/rasterize_points_cpu.cpp:27-144)
 *   slow occupancy backward PINNED   bit-exact vs reference CPU (rasterize_points_cpu.cpp:380-477)
 *   zbuf backward           PINNED   bit-exact vs reference CPU (rasterize_points_cpu.cpp:479-513)
 *   fast occupancy backward PINNED   vs the reference CUDA kernel itself (rasterize_points_backward.cu:30-212), host-compiled
 *                                    unmodified by oracle/ref_cuda_host.cpp and EXECUTED inside the reference's own
 *                                    EllipticalRasterizer.backward (tests/golden/make_golden_fast_backward.py)
 *   blend fwd/bwd           PINNED   vs the reference's own renderer.py:36-82 + gather_with_neg_idx + its weighted-sum CUDA
 *                                    kernels (weighted_sum.cu:38-134) host-compiled and executed (make_golden_blend.py);
 *                                    the 1e-4-clamped normalisation of pytorch3d's NormWeightedCompositor is restated
 *   per-point EWA setup     PINNED   vs the reference's own Python (rasterizer.py:293-565) run with stubbed
 *                                    third-party imports (tests/golden/make_golden_setup.py); the projection
 *                                    itself (pytorch3d cameras) stays unpinned
 *   visible set / median rs PINNED   vs the Python half of EllipticalRasterizer.backward (rasterizer.py:853-913)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DSS_ORACLE_API __attribute__((visibility("default")))

/* rasterization_utils.cuh:8-11 / rasterize_points_cpu.cpp:10-14 */
static inline float pix_to_ndc(int i, int S) { return -1 + (2 * i + 1.0f) / S; }

/* rasterize_points_cpu.cpp:22-25, rasterize_points.cu:98 (same expression, same order) */
static inline float qvalue_of(float dx, float dy, float a, float b, float c)
{
    return a * dx * dx + b * dx * dy + c * dy * dy;
}

typedef struct { float z; int32_t idx; float q; } frag_t;

/* strict total order used for the K-nearest set: (z, idx) ascending.
 * rasterize_points_cpu.cpp:85 keeps the K smallest std::tuple<z, idx, q>; the CUDA kernels
 * (rasterize_points.cu:99-123) visit points in index order and only evict on strict z<max,
 * which yields the same set whenever z values are distinct. */
static inline int frag_less(const frag_t *a, const frag_t *b)
{
    if (a->z < b->z) return 1;
    if (a->z > b->z) return 0;
    return a->idx < b->idx;
}

/* insert into an ascending list of at most K entries */
static inline void frag_insert(frag_t *q, int *count, int K, frag_t e)
{
    int n = *count;
    if (n == K) {
        if (!frag_less(&e, &q[K - 1])) return;
        n = K - 1;
    }
    int k = n;
    while (k > 0 && frag_less(&e, &q[k - 1])) { q[k] = q[k - 1]; --k; }
    q[k] = e;
    *count = n + 1;
}

/* hit test of one (pixel, point) pair.
 * CUDA form rasterize_points.cu:79-96: skip if pz<0; skip if |dx|>rx OR |dy|>ry; skip if Q>cutoff.
 * CPU form rasterize_points_cpu.cpp:92-104 uses AND in the bbox pre-test (flag bit 0 selects it;
 * only used to pin this file against the compiled CPU reference on arbitrary radii). */
static inline int pair_hit(float xf, float yf, float px, float py, float pz,
                           float a, float b, float c, float rx, float ry, float cutoff,
                           int cpu_bbox_and, float *q_out)
{
    if (pz < 0) return 0;
    const float dx = xf - px;
    const float dy = yf - py;
    if (cpu_bbox_and) {
        if (fabsf(dx) > rx && fabsf(dy) > ry) return 0;
    } else {
        if (fabsf(dx) > rx || fabsf(dy) > ry) return 0;
    }
    const float q = qvalue_of(dx, dy, a, b, c);
    if (q > cutoff) return 0;
    *q_out = q;
    return 1;
}

static void write_pixel(const frag_t *q, int count, int K, float thr,
                        int32_t *idx, float *zbuf, float *qv, float *occ)
{
    /* outputs pre-filled with -1 / 0: rasterize_points.cu:254-257, 634-637 */
    for (int k = 0; k < K; ++k) { idx[k] = -1; zbuf[k] = -1.0f; qv[k] = -1.0f; }
    *occ = 0.0f;
    if (count == 0) return;
    *occ = 1.0f; /* rasterize_points.cu:196-200 / :581-585 (pz>=0 for every hit) */
    for (int k = 0; k < count; ++k) {
        /* depth merge, rasterize_points.cu:201-210 / :586-595 */
        if (q[k].z - q[0].z > thr) break;
        idx[k] = q[k].idx; zbuf[k] = q[k].z; qv[k] = q[k].q;
    }
}

/* ---------------------------------------------------------------------------------------
 * Forward rasterizer, literal per-pixel brute force.
 * Restates RasterizePointsNaiveCudaKernel (rasterize_points.cu:131-212) ==
 * RasterizePointsNaiveCpu (rasterize_points_cpu.cpp:27-144).
 * Image pixel (r, c) has NDC centre (pix_to_ndc(S-1-c), pix_to_ndc(S-1-r)).
 * flags bit0: CPU-variant bbox pre-test (see pair_hit).
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API int oracle_splat_forward_brute(
    const float *points, const float *ellipse, const float *cutoff, const float *radii,
    const int64_t *first_idx, const int64_t *num_pts, int N, int S, int K, float thr, int flags,
    int32_t *idx, float *zbuf, float *qv, float *occ)
{
    if (K <= 0 || S <= 0) return -1;
    frag_t *q = (frag_t *)malloc(sizeof(frag_t) * (size_t)K);
    if (!q) return -2;
    for (int n = 0; n < N; ++n) {
        const int64_t p0 = first_idx[n], p1 = p0 + num_pts[n];
        for (int r = 0; r < S; ++r) {
            const float yf = pix_to_ndc(S - 1 - r, S);
            for (int c = 0; c < S; ++c) {
                const float xf = pix_to_ndc(S - 1 - c, S);
                int count = 0;
                for (int64_t p = p0; p < p1; ++p) {
                    float qq;
                    if (!pair_hit(xf, yf, points[3 * p], points[3 * p + 1], points[3 * p + 2],
                                  ellipse[3 * p], ellipse[3 * p + 1], ellipse[3 * p + 2],
                                  radii[2 * p], radii[2 * p + 1], cutoff[p], flags & 1, &qq))
                        continue;
                    frag_t e = { points[3 * p + 2], (int32_t)p, qq };
                    frag_insert(q, &count, K, e);
                }
                const size_t pix = ((size_t)n * S + r) * S + c;
                write_pixel(q, count, K, thr, idx + pix * K, zbuf + pix * K, qv + pix * K, occ + pix);
            }
        }
    }
    free(q);
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * Forward rasterizer, same per-pair rule but driven per splat over a conservative pixel
 * window (bbox +-2 px) so that BASELINE-sized inputs finish in seconds.  Because the pair
 * test is evaluated with the identical expression and the K-set is defined by the total
 * order (z, idx), the result is identical to the brute-force loop (checked in tests).
 * CUDA-form bbox test only.
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API int oracle_splat_forward(
    const float *points, const float *ellipse, const float *cutoff, const float *radii,
    const int64_t *first_idx, const int64_t *num_pts, int N, int S, int K, float thr,
    int32_t *idx, float *zbuf, float *qv, float *occ)
{
    if (K <= 0 || S <= 0) return -1;
    const size_t npix = (size_t)S * S;
    frag_t *lists = (frag_t *)malloc(sizeof(frag_t) * npix * (size_t)K);
    int *counts = (int *)malloc(sizeof(int) * npix);
    if (!lists || !counts) { free(lists); free(counts); return -2; }
    for (int n = 0; n < N; ++n) {
        memset(counts, 0, sizeof(int) * npix);
        const int64_t p0 = first_idx[n], p1 = p0 + num_pts[n];
        /* row bands: every thread owns the pixels of its rows and visits the splats in index order, so each
         * pixel's K-list sees exactly the sequence of the serial loop */
        #pragma omp parallel
        {
        int band0 = 0, band1 = S;
#ifdef _OPENMP
        const int nt = omp_get_num_threads(), tid = omp_get_thread_num();
        band0 = (int)((int64_t)S * tid / nt); band1 = (int)((int64_t)S * (tid + 1) / nt);
#endif
        if (band0 < band1)
        for (int64_t p = p0; p < p1; ++p) {
            const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
            const float rx = radii[2 * p], ry = radii[2 * p + 1];
            if (pz < 0) continue;
            /* NaN / inf radii or positions: fall back to the whole image for this splat */
            int c_lo = 0, c_hi = S - 1, r_lo = 0, r_hi = S - 1;
            if (isfinite(px) && isfinite(rx)) {
                /* ndc index i = S-1-c ; centre = -1 + (2i+1)/S  ->  i = ((x+1)*S-1)/2 */
                double i_lo = (((double)px - (double)rx + 1.0) * S - 1.0) / 2.0;
                double i_hi = (((double)px + (double)rx + 1.0) * S - 1.0) / 2.0;
                if (i_hi < -3.0 || i_lo > S + 2.0) continue;
                int ilo = (int)floor(fmax(i_lo, -4.0)) - 2, ihi = (int)ceil(fmin(i_hi, S + 4.0)) + 2;
                if (ilo < 0) ilo = 0;
                if (ihi > S - 1) ihi = S - 1;
                c_lo = S - 1 - ihi; c_hi = S - 1 - ilo;
            }
            if (isfinite(py) && isfinite(ry)) {
                double i_lo = (((double)py - (double)ry + 1.0) * S - 1.0) / 2.0;
                double i_hi = (((double)py + (double)ry + 1.0) * S - 1.0) / 2.0;
                if (i_hi < -3.0 || i_lo > S + 2.0) continue;
                int ilo = (int)floor(fmax(i_lo, -4.0)) - 2, ihi = (int)ceil(fmin(i_hi, S + 4.0)) + 2;
                if (ilo < 0) ilo = 0;
                if (ihi > S - 1) ihi = S - 1;
                r_lo = S - 1 - ihi; r_hi = S - 1 - ilo;
            }
            if (r_lo < band0) r_lo = band0;
            if (r_hi > band1 - 1) r_hi = band1 - 1;
            for (int r = r_lo; r <= r_hi; ++r) {
                const float yf = pix_to_ndc(S - 1 - r, S);
                for (int c = c_lo; c <= c_hi; ++c) {
                    const float xf = pix_to_ndc(S - 1 - c, S);
                    float qq;
                    if (!pair_hit(xf, yf, px, py, pz, ellipse[3 * p], ellipse[3 * p + 1],
                                  ellipse[3 * p + 2], rx, ry, cutoff[p], 0, &qq))
                        continue;
                    const size_t pix = (size_t)r * S + c;
                    frag_t e = { pz, (int32_t)p, qq };
                    frag_insert(lists + pix * K, &counts[pix], K, e);
                }
            }
        }
        }  /* omp parallel */
        #pragma omp parallel for schedule(static)
        for (size_t pix = 0; pix < npix; ++pix) {
            const size_t o = (size_t)n * npix + pix;
            write_pixel(lists + pix * K, counts[pix], K, thr, idx + o * K, zbuf + o * K, qv + o * K, occ + o);
        }
    }
    free(lists); free(counts);
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * Per-point visibility: point p is visible iff it appears in the fragment list of any pixel
 * whose first slot is filled.  DSS/utils/__init__.py:320-340 (forward, rasterizer.py:639-641)
 * and rasterizer.py:854-860 (backward).
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_visibility(const int32_t *idx, int N, int S, int K, int64_t P, uint8_t *vis)
{
    memset(vis, 0, (size_t)P);
    const size_t npix = (size_t)N * S * S;
    for (size_t i = 0; i < npix; ++i) {
        if (idx[i * K] < 0) continue;
        for (int k = 0; k < K; ++k) {
            const int32_t p = idx[i * K + k];
            if (p >= 0 && p < P) vis[p] = 1;
        }
    }
}

static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* rs[n] = median(flattened (n_vis, 2) radii of the visible points of cloud n) * radii_s
 * rasterizer.py:885-888 (torch.median = lower median).  Clouds without visible points get 0. */
DSS_ORACLE_API void oracle_backward_radius(const float *radii, const uint8_t *vis,
                                           const int64_t *first_idx, const int64_t *num_pts, int N,
                                           float radii_s, float *rs)
{
    for (int n = 0; n < N; ++n) {
        const int64_t p0 = first_idx[n], p1 = p0 + num_pts[n];
        int64_t cnt = 0;
        for (int64_t p = p0; p < p1; ++p) cnt += vis[p] ? 2 : 0;
        rs[n] = 0.0f;
        if (cnt == 0) continue;
        float *tmp = (float *)malloc(sizeof(float) * (size_t)cnt);
        int64_t j = 0;
        for (int64_t p = p0; p < p1; ++p)
            if (vis[p]) { tmp[j++] = radii[2 * p]; tmp[j++] = radii[2 * p + 1]; }
        qsort(tmp, (size_t)cnt, sizeof(float), cmp_float);
        rs[n] = tmp[(cnt - 1) / 2] * radii_s;
        free(tmp);
    }
}

/* ---------------------------------------------------------------------------------------
 * FAST occupancy backward (the one train_mvr.py uses, rasterizer.py:816 backward_occ_fast=True).
 * Restates RasterizePointsBackwardCudaFastKernel, rasterize_points_backward.cu:85-96 (pixel ->
 * NDC), :141-178 (per-pair rule).  The FRNN grid (rasterizer.py:889-933) only enumerates the
 * visible points within rs of the pixel, so the result equals this brute force over the
 * visible points of cloud n.
 *   for pixel (r,c), g = grad_occ != 0, xf = ndc(S-1-c), yf = ndc(S-1-r):
 *     for p visible in cloud n with pz>=0, |px|<=1, |py|<=1:
 *        d2 = dx*dx+dy*dy ; skip if d2 > rs^2
 *        skip if g>0 and (|dx|>rx or |dy|>ry)
 *        den = eps_denom(d2, 1e-10)  (rasterization_utils.cuh:38-43)
 *        grad[p] += (dx,dy)/den*g
 * Documented divergence: eps_denom(0) = 0 gives 0/0 = NaN in the reference when a point sits
 * exactly on a pixel centre; here (and in the HIP path) such a pair contributes 0.
 * Accumulation is in double (the order-independent limit of the reference's fp32 atomics);
 * each per-pair term is rounded in fp32 exactly as the reference computes it.
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_occ_backward_fast(
    const float *points, const float *radii, const uint8_t *vis, const float *rs,
    const float *grad_occ, const int64_t *first_idx, const int64_t *num_pts,
    int N, int64_t P, int S, float *grad_xy /* (P,2) */)
{
    double *acc = (double *)calloc((size_t)P * 2, sizeof(double));
    for (int n = 0; n < N; ++n) {
        const int64_t p0 = first_idx[n], p1 = p0 + num_pts[n];
        const float cur_r = rs[n];
        const float cur_r2 = cur_r * cur_r;
        #pragma omp parallel for schedule(dynamic, 512)
        for (int64_t p = p0; p < p1; ++p) {
            if (!vis[p]) continue;
            const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
            if (pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f) continue;
            const float rx = radii[2 * p], ry = radii[2 * p + 1];
            /* conservative pixel window around the point (+-2 px slack); exact test inside */
            double ix_lo = (((double)px - (double)cur_r + 1.0) * S - 1.0) / 2.0;
            double ix_hi = (((double)px + (double)cur_r + 1.0) * S - 1.0) / 2.0;
            double iy_lo = (((double)py - (double)cur_r + 1.0) * S - 1.0) / 2.0;
            double iy_hi = (((double)py + (double)cur_r + 1.0) * S - 1.0) / 2.0;
            int xlo = 0, xhi = S - 1, ylo = 0, yhi = S - 1;
            if (isfinite(ix_lo) && isfinite(ix_hi)) {
                xlo = (int)floor(fmax(ix_lo, -4.0)) - 2; xhi = (int)ceil(fmin(ix_hi, S + 4.0)) + 2;
                if (xlo < 0) xlo = 0;
                if (xhi > S - 1) xhi = S - 1;
            }
            if (isfinite(iy_lo) && isfinite(iy_hi)) {
                ylo = (int)floor(fmax(iy_lo, -4.0)) - 2; yhi = (int)ceil(fmin(iy_hi, S + 4.0)) + 2;
                if (ylo < 0) ylo = 0;
                if (yhi > S - 1) yhi = S - 1;
            }
            double gx = 0.0, gy = 0.0;
            for (int yi = ylo; yi <= yhi; ++yi) {
                const float yf = pix_to_ndc(yi, S);
                const int r = S - 1 - yi;
                for (int xi = xlo; xi <= xhi; ++xi) {
                    const int c = S - 1 - xi;
                    const float g = grad_occ[((size_t)n * S + r) * S + c];
                    if (g == 0.0f) continue;
                    const float xf = pix_to_ndc(xi, S);
                    const float dx = xf - px, dy = yf - py;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 > cur_r2) continue;
                    const int outside = (fabsf(dx) > rx) || (fabsf(dy) > ry);
                    if (g > 0.0f && outside) continue;
                    if (d2 == 0.0f) continue; /* documented divergence (reference: NaN) */
                    const float den = fmaxf(d2, 1e-10f);
                    gx += (double)(dx / den * g);
                    gy += (double)(dy / den * g);
                }
            }
            acc[2 * p] = gx; acc[2 * p + 1] = gy;
        }
    }
    for (int64_t i = 0; i < 2 * P; ++i) grad_xy[i] = (float)acc[i];
    free(acc);
}

/* ---------------------------------------------------------------------------------------
 * SLOW occupancy backward, CPU semantics: RasterizePointsOccBackwardCpu,
 * rasterize_points_cpu.cpp:380-477 (box support radii*radii_s with AND, eps 1e-8, every point
 * of the cloud).  Same loop nest and fp32 accumulation order -> bit-exact vs oracle/_ref.
 * Not on the default training path (rasterizer.py:816); restated only to pin the shared
 * pixel->NDC / skip-rule arithmetic against runnable reference code.
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_occ_backward_slow_cpu(
    const float *points, const float *radii, const float *grad_occ,
    const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S, float radii_s,
    float *grad_xy /* (P,2), zeroed here */)
{
    memset(grad_xy, 0, sizeof(float) * (size_t)P * 2);
    for (int n = 0; n < N; ++n) {
        const int p0 = (int)first_idx[n], p1 = p0 + (int)num_pts[n];
        for (int yi = 0; yi < S; ++yi) {
            const float yf = pix_to_ndc(S - 1 - yi, S);
            for (int xi = 0; xi < S; ++xi) {
                const float xf = pix_to_ndc(S - 1 - xi, S);
                const float g = grad_occ[((size_t)n * S + yi) * S + xi];
                if (g == 0.0f) continue;
                for (int p = p0; p < p1; ++p) {
                    const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
                    if (pz < 0 || fabsf(py) > 1.0 || fabsf(px) > 1.0) continue;
                    const float dx = xf - px, dy = yf - py;
                    const float rxs = radii[2 * p] * radii_s, rys = radii[2 * p + 1] * radii_s;
                    const int outside = (fabsf(dx) > rxs / radii_s) || (fabsf(dy) > rys / radii_s);
                    if (g > 0.0f && outside) continue;
                    if (fabsf(dx) > rxs && fabsf(dy) > rys) continue;
                    const float d2 = dx * dx + dy * dy;
                    const float den = d2 > 1e-8f ? d2 : 1e-8f;
                    grad_xy[2 * p] += dx / den * g;
                    grad_xy[2 * p + 1] += dy / den * g;
                }
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * SLOW occupancy backward, CUDA semantics: RasterizePointsOccBackwardCudaKernel, rasterize_points.cu:672-757
 * (`DSS._C._splat_points_occ_backward` on CUDA tensors; logical OR in the support test, eps 1e-10, d2 == 0 -> device
 * eps_denom gives 0/0 = NaN in the reference, 0 here).  Pinned against the kernel itself, host-compiled and executed
 * (oracle/ref_cuda_host.cpp, tests/golden/make_golden_fast_backward.py).  Double accumulation like the fast form.
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_occ_backward_slow_cuda(
    const float *points, const float *radii, const float *grad_occ,
    const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S, float radii_s,
    float *grad_xy /* (P,2) */)
{
    memset(grad_xy, 0, sizeof(float) * (size_t)P * 2);
    for (int n = 0; n < N; ++n) {
        const int64_t p0 = first_idx[n], p1 = p0 + num_pts[n];
        #pragma omp parallel for schedule(dynamic, 64)
        for (int64_t p = p0; p < p1; ++p) {
            const float px = points[3 * p], py = points[3 * p + 1], pz = points[3 * p + 2];
            if (pz < 0 || fabsf(py) > 1.0f || fabsf(px) > 1.0f) continue;
            const float radiix = radii[2 * p] * radii_s, radiiy = radii[2 * p + 1] * radii_s;
            double gx = 0.0, gy = 0.0;
            for (int yi = 0; yi < S; ++yi) {
                const float yf = pix_to_ndc(yi, S);
                const float dy = yf - py;
                if (fabsf(dy) > radiiy) continue;
                for (int xi = 0; xi < S; ++xi) {
                    const float g = grad_occ[((size_t)n * S + (S - 1 - yi)) * S + (S - 1 - xi)];
                    if (g == 0.0f) continue;
                    const float dx = pix_to_ndc(xi, S) - px;
                    if (fabsf(dx) > radiix) continue;
                    const int outside = (fabsf(dx) > radiix / radii_s) || (fabsf(dy) > radiiy / radii_s);
                    if (g > 0.0f && outside) continue;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 == 0.0f) continue;
                    const float den = fmaxf(d2, 1e-10f);
                    gx += (double)(dx / den * g);
                    gy += (double)(dy / den * g);
                }
            }
            grad_xy[2 * p] = (float)gx; grad_xy[2 * p + 1] = (float)gy;
        }
    }
}

/* ---------------------------------------------------------------------------------------
 * zbuf backward: z_grad[idx[n,y,x,k]] += grad_zbuf[n,y,x,k]; zero grads skipped, stop at the
 * first idx<0.  rasterize_points.cu:823-846 == rasterize_points_cpu.cpp:479-513.
 * fp32 accumulation in raster order (bit-exact vs the CPU reference).  Accumulates IN PLACE.
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_zbuf_backward(const int32_t *idx, const float *grad_zbuf,
                                         int N, int S, int K, float *z_grad /* (P,) in/out */)
{
    const size_t npix = (size_t)N * S * S;
    for (size_t i = 0; i < npix; ++i)
        for (int k = 0; k < K; ++k) {
            const float g = grad_zbuf[i * K + k];
            if (g == 0.0f) continue;
            const int32_t p = idx[i * K + k];
            if (p < 0) break;
            z_grad[p] += g;
        }
}

/* ---------------------------------------------------------------------------------------
 * Per-point gradient clipping hook: grad <- normalize(grad) * min(||grad||, clip)
 * rasterizer.py:667-673 (F.normalize eps = 1e-12), installed at :735-737.
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_clip_grad(float *grad /* (P,3) */, int64_t P, float clip)
{
    for (int64_t p = 0; p < P; ++p) {
        float *g = grad + 3 * p;
        const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        const float scaler = nrm < clip ? nrm : clip;       /* clamp(0, value) */
        const float den = nrm > 1e-12f ? nrm : 1e-12f;       /* F.normalize */
        for (int j = 0; j < 3; ++j) g[j] = g[j] / den * scaler;
    }
}

/* ---------------------------------------------------------------------------------------
 * Blend.  weights: renderer.py:53 with the scaler gather of rasterizer.py:631-633
 * (utils/__init__.py:172-185):  w_k = exp(-0.5*Q_k) * scaler[idx_k]  (0 where idx_k < 0).
 * Compositor: pytorch3d NormWeightedCompositor / norm_weighted_sum at renderer.py:67-72
 * [third party, pytorch3d 0.2.5-0.4.0, not under /root/reference; published algorithm]:
 *     cum = sum_k w_k (idx_k>=0) ; cum = max(cum, 1e-4)
 *     img[ch] = sum_k f[idx_k][ch] * w_k / cum
 * RGBA assembly renderer.py:75-78: out[..., C] = occupancy.
 * features are (P, C) row-major (= Pointclouds.features_packed()).
 * ------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_blend_forward(
    const int32_t *idx, const float *qv, const float *occ, const float *scaler,
    const float *feat, int N, int S, int K, int C, float *out /* (N,S,S,C+1) */)
{
    const size_t npix = (size_t)N * S * S;
    #pragma omp parallel for schedule(static)
    for (size_t i = 0; i < npix; ++i) {
        float cum = 0.0f;
        for (int k = 0; k < K; ++k) {
            const int32_t p = idx[i * K + k];
            if (p < 0) continue;
            cum += expf(-0.5f * qv[i * K + k]) * scaler[p];
        }
        if (cum < 1e-4f) cum = 1e-4f;
        for (int ch = 0; ch < C; ++ch) {
            float acc = 0.0f;
            for (int k = 0; k < K; ++k) {
                const int32_t p = idx[i * K + k];
                if (p < 0) continue;
                const float w = expf(-0.5f * qv[i * K + k]) * scaler[p];
                acc += feat[(size_t)p * C + ch] * w / cum;
            }
            out[i * (C + 1) + ch] = acc;
        }
        out[i * (C + 1) + C] = occ[i];
    }
}

/* Blend backward to the per-point features (colours):
 *     grad_f[idx_k][ch] += grad_out[ch] * w_k / cum
 * (pytorch3d norm_weighted_sum backward; the gradient w.r.t. the weights is dead in DSS because
 * EllipticalRasterizer.backward ignores qvalue_grad, rasterizer.py:788-789, and the EWA terms
 * are detached, :562-565).  The alpha channel's gradient is returned as grad_occ (N,S,S).
 * Double accumulation (order-independent limit of the reference's atomics). */
DSS_ORACLE_API void oracle_blend_backward(
    const float *grad_out /* (N,S,S,C+1) */, const int32_t *idx, const float *qv,
    const float *scaler, int N, int S, int K, int C, int64_t P,
    float *grad_feat /* (P,C) */, float *grad_occ /* (N,S,S) */)
{
    double *acc = (double *)calloc((size_t)P * C, sizeof(double));
    const size_t npix = (size_t)N * S * S;
    for (size_t i = 0; i < npix; ++i) {
        grad_occ[i] = grad_out[i * (C + 1) + C];
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
            for (int ch = 0; ch < C; ++ch)
                acc[(size_t)p * C + ch] += (double)(grad_out[i * (C + 1) + ch] * w / cum);
        }
    }
    for (size_t i = 0; i < (size_t)P * C; ++i) grad_feat[i] = (float)acc[i];
    free(acc);
}

/* ---------------------------------------------------------------------------------------
 * Per-point EWA setup, fp32.
 *
 * Inputs per camera n: M (4x4, row-vector convention p_h @ M, = pytorch3d
 * cameras.get_full_projection_transform().get_matrix()[n]) and V (4x4 world->view, same
 * convention); per point: world position, unit normal, source-space variance scale h.
 *
 *   projection (pytorch3d PointsRasterizer.transform at rasterizer.py:614 [third party]):
 *       clip = p_h @ M ; ndc_xy = clip.xy / clip.w ; z = (p_h @ V).z   (view-space depth)
 *   Jacobian (rasterizer.py:443-496, _compute_WJk):
 *       w = p_h . M[:,3] ; xy = p_h @ M[:, :2]
 *       Jk[0][0] = Jk[1][1] = 1/eps_denom(w) ; Jk[3][j] = -xy[j]/eps_denom(w*w)
 *       WJk = M[:3,:] @ Jk                                   (3x2)
 *   source variance (rasterizer.py:293-342): Vrk = h * Sk^T Sk = h * (I - n^ n^^T), n^ = n/|n|
 *       (Sk is a random orthonormal tangent basis; the product is basis independent)
 *   Vk = WJk^T Vrk WJk ; GV = Vk + sigma*I*(2/S)^2            (rasterizer.py:404-441)
 *   |detMk| = |det(Sk @ WJk)| = |n^ . (WJk[:,0] x WJk[:,1])|   (Binet-Cauchy, basis independent)
 *   GVinv = inverse(GV); (a,b,c) = (GVinv00, GVinv01+GVinv10, GVinv11)   (rasterizer.py:541-550)
 *   radii (rasterizer.py:498-523): den = eps_denom(4ac-b^2); rx = sqrt(eps_sqrt(4*c*C/den)),
 *       ry = sqrt(eps_sqrt(4*a*C/den))
 *   scaler = |detMk| / eps_denom(sqrt(eps_sqrt(det(GV)*4*pi^2)))        (rasterizer.py:556-558)
 * eps_denom / eps_sqrt: DSS/utils/mathHelper.py:10-21 (eps 1e-17).
 * ------------------------------------------------------------------------------------- */
static inline float eps_denom_py(float d)
{
    const float s = (d > 0) - (d < 0) + (d == 0.0f ? 1.0f : 0.0f);
    const float a = fabsf(d);
    return s * (a > 1e-17f ? a : 1e-17f);
}
static inline float eps_sqrt_py(float d) { const float a = fabsf(d); return a > 1e-17f ? a : 1e-17f; }

/* ---------------------------------------------------------------------------------------------
 * Anisotropic source variance (rasterizer.py:256-291 + mathHelper.py:34-92): for every point, the K nearest
 * points of its cloud (the point itself included, pytorch3d knn_points) are centred on their mean; the
 * singular values of the (K,3) difference matrix give curvature = sigma^2 / K = eigenvalues of the neighbourhood
 * covariance C = (1/K) sum d d^T, ascending, with the right singular vectors as frame.  The frame's last two
 * columns F (tangent directions) and curvatures give Vrk = F diag(c1, c2) F^T = C - c0 e0 e0^T, and Sk = F^T.
 * Restated with a cyclic Jacobi eigen-solver in double precision (the reference uses an fp32 batched SVD).
 * knn_idx holds PACKED point ids.  Outputs: vr6 (xx,xy,xz,yy,yz,zz), frame_n = e0, curv (ascending).
 * ------------------------------------------------------------------------------------------- */
static void jacobi_eig3(double A[3][3], double V[3][3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
}

DSS_ORACLE_API void oracle_local_frames(const float *pts /* (P,3) */, const int64_t *knn_idx /* (P,K) packed ids */,
                                        int64_t P, int K, float *vr6 /* (P,6) */, float *frame_n /* (P,3) */,
                                        float *curv /* (P,3) */)
{
    for (int64_t p = 0; p < P; ++p) {
        double mean[3] = {0, 0, 0};
        for (int k = 0; k < K; ++k)
            for (int d = 0; d < 3; ++d) mean[d] += (double)pts[3 * knn_idx[p * K + k] + d];
        for (int d = 0; d < 3; ++d) mean[d] /= K;
        double C[3][3] = {{0}};
        for (int k = 0; k < K; ++k) {
            double df[3];
            for (int d = 0; d < 3; ++d) df[d] = (double)pts[3 * knn_idx[p * K + k] + d] - mean[d];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) C[i][j] += df[i] * df[j] / K;
        }
        double A[3][3], V[3][3];
        memcpy(A, C, sizeof(A));
        jacobi_eig3(A, V);
        int o[3] = {0, 1, 2};  /* ascending eigenvalues */
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j)
                if (A[o[j]][o[j]] < A[o[i]][o[i]]) { const int t = o[i]; o[i] = o[j]; o[j] = t; }
        const double l0 = A[o[0]][o[0]];
        const double e0[3] = {V[0][o[0]], V[1][o[0]], V[2][o[0]]};
        for (int d = 0; d < 3; ++d) {
            curv[3 * p + d] = (float)A[o[d]][o[d]];
            frame_n[3 * p + d] = (float)e0[d];
        }
        const int ii[6] = {0, 0, 0, 1, 1, 2}, jj[6] = {0, 1, 2, 1, 2, 2};
        for (int q = 0; q < 6; ++q) vr6[6 * p + q] = (float)(C[ii[q]][jj[q]] - l0 * e0[ii[q]] * e0[jj[q]]);
    }
}

DSS_ORACLE_API void oracle_point_setup(
    const float *pts_world /* (P,3) */, const float *normals /* (P,3) */, const float *h /* (P,) */,
    const int32_t *cloud_of /* (P,) */, const float *M /* (N,4,4) */, const float *V /* (N,4,4) */,
    int64_t P, int S, float cutoffC, float sigma,
    const float *vr6 /* (P,6) xx,xy,xz,yy,yz,zz or NULL */, const float *frame_n /* (P,3) or NULL */,
    float *pts_screen /* (P,3) */, float *ellipse /* (P,3) */, float *radii /* (P,2) */,
    float *scaler /* (P,) */, float *cutoff /* (P,) */)
{
    /* vr6 != NULL: anisotropic source variance (rasterizer.py:256-291): Vrk is given per point and the tangent
     * frame of det(Sk WJk) is the PCA frame, whose normal is frame_n (the cloud normals are not used). */
    const float pixel = 2.0f / (float)S;
    #pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < P; ++p) {
        const float *m = M + 16 * cloud_of[p];
        const float *v = V + 16 * cloud_of[p];
        const float ph[4] = { pts_world[3 * p], pts_world[3 * p + 1], pts_world[3 * p + 2], 1.0f };
        float clip[4];
        for (int j = 0; j < 4; ++j)
            clip[j] = ph[0] * m[0 * 4 + j] + ph[1] * m[1 * 4 + j] + ph[2] * m[2 * 4 + j] + ph[3] * m[3 * 4 + j];
        const float zview = ph[0] * v[0 * 4 + 2] + ph[1] * v[1 * 4 + 2] + ph[2] * v[2 * 4 + 2] + ph[3] * v[3 * 4 + 2];
        const float w = clip[3];
        pts_screen[3 * p] = clip[0] / w;
        pts_screen[3 * p + 1] = clip[1] / w;
        pts_screen[3 * p + 2] = zview;

        const float dw = eps_denom_py(w), dw2 = eps_denom_py(w * w);
        /* WJk[i][j] = M[i][j]/dw + M[i][3] * (-clip[j]/dw2) , i<3, j<2 */
        float WJ[3][2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j)
                WJ[i][j] = m[i * 4 + j] * (1.0f / dw) + m[i * 4 + 3] * (-1.0f / dw2 * clip[j]);
        /* Sk is built from NORMALISED cross products (rasterizer.py:337-341, F.normalize eps 1e-12), so
         * Sk^T Sk = I - n^ n^^T for the unit normal n^ whatever the length of the stored normal
         * (bunny-8000.ply stores |n| = 56.25); a zero normal gives Sk = 0. */
        const float *nraw = vr6 ? frame_n + 3 * p : normals + 3 * p;
        const float nlen = sqrtf(nraw[0] * nraw[0] + nraw[1] * nraw[1] + nraw[2] * nraw[2]);
        const float nden = nlen > 1e-12f ? nlen : 1e-12f;
        const float nn[3] = { nraw[0] / nden, nraw[1] / nden, nraw[2] / nden };
        const float hh = h[p];
        const float hv = nlen > 1e-12f ? hh : 0.0f;
        /* Vrk = h (I - n^ n^^T) */
        float Vr[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                Vr[i][j] = hv * ((i == j ? 1.0f : 0.0f) - nn[i] * nn[j]);
        if (vr6) {
            const float *q = vr6 + 6 * p;
            Vr[0][0] = q[0]; Vr[0][1] = Vr[1][0] = q[1]; Vr[0][2] = Vr[2][0] = q[2];
            Vr[1][1] = q[3]; Vr[1][2] = Vr[2][1] = q[4]; Vr[2][2] = q[5];
        }
        float T[3][2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 2; ++j)
                T[i][j] = Vr[i][0] * WJ[0][j] + Vr[i][1] * WJ[1][j] + Vr[i][2] * WJ[2][j];
        float Vk[2][2];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                Vk[i][j] = WJ[0][i] * T[0][j] + WJ[1][i] * T[1][j] + WJ[2][i] * T[2][j];
        const float detVk = Vk[0][0] * Vk[1][1] - Vk[0][1] * Vk[1][0];
        /* |det(Sk WJk)| (rasterizer.py:430-439 torch.det(Mk)) = |n^ . (w0 x w1)| with w_j = WJk[:, j]
         * (Binet-Cauchy: det([u0;u1][w0 w1]) = (u0 x u1).(w0 x w1), u0 x u1 = +-n^): basis independent and,
         * unlike sqrt(det Vk)/h, well conditioned for splats seen edge-on. */
        const float cx0 = WJ[1][0] * WJ[2][1] - WJ[2][0] * WJ[1][1];
        const float cx1 = WJ[2][0] * WJ[0][1] - WJ[0][0] * WJ[2][1];
        const float cx2 = WJ[0][0] * WJ[1][1] - WJ[1][0] * WJ[0][1];
        const float absdetMk = nlen > 1e-12f ? fabsf(nn[0] * cx0 + nn[1] * cx1 + nn[2] * cx2) : 0.0f;
        (void)detVk;
        const float G00 = Vk[0][0] + sigma * (pixel * pixel), G11 = Vk[1][1] + sigma * (pixel * pixel);
        const float G01 = Vk[0][1], G10 = Vk[1][0];
        const float detG = G00 * G11 - G01 * G10;
        const float a = G11 / detG, c = G00 / detG, b = (-G01 / detG) + (-G10 / detG);
        ellipse[3 * p] = a; ellipse[3 * p + 1] = b; ellipse[3 * p + 2] = c;
        const float den = eps_denom_py(4.0f * a * c - b * b);
        radii[2 * p] = sqrtf(eps_sqrt_py(4.0f * c * cutoffC / den));
        radii[2 * p + 1] = sqrtf(eps_sqrt_py(4.0f * a * cutoffC / den));
        const float sc = sqrtf(eps_sqrt_py(detG * 4.0f * (float)M_PI * (float)M_PI));
        scaler[p] = absdetMk / eps_denom_py(sc);
        cutoff[p] = cutoffC;
    }
}


/* ---------------------------------------------------------------------------------------------
 * Phong shading of the points: LightingTexture.forward (DSS/core/texture.py:65-125) = apply_lighting
 * (:26-63) with `diffuse` (DSS/core/lighting.py:10-77) and `specular` (:80-172), L lights per cloud, as
 * PointLights (direction = location - point, lighting.py:270-276) or DirectionalLights:
 *   n^ = n / max(|n|, 1e-6) (F.normalize), same for the light direction d^ and the view direction v^
 *   diffuse  = sum_l kd_l relu(n^.d^)
 *   specular = sum_l ks_l (relu(v^.(-d^ + 2 (n^.d^) n^)) [n^.d^ > 0])^shininess
 *   out = rgb (ambient + diffuse) + specular                                   (texture.py:118-122)
 * fp32 like the reference.  cloud_of (P,) gives the cloud of every packed point.
 * ------------------------------------------------------------------------------------------- */
static void normalize3(const float *v, float *o)
{
    float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (n < 1e-6f) n = 1e-6f;
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

DSS_ORACLE_API void oracle_phong_forward(const float *pts, const float *normals, const float *rgb, const int32_t *cloud_of,
                                         int64_t P, const float *ambient /* (N,3) */, const float *kd /* (N,L,3) */,
                                         const float *ks, const float *lvec, int L, int point_lights,
                                         const float *cam /* (N,3) */, float shininess, float *out /* (P,3) */,
                                         float *out_diffuse, float *out_specular /* (P,3) or NULL */)
{
    for (int64_t p = 0; p < P; ++p) {
        const int n = cloud_of[p];
        float nh[3], v[3], dif[3] = {0, 0, 0}, spec[3] = {0, 0, 0};
        normalize3(normals + 3 * p, nh);
        const float w[3] = {cam[3 * n] - pts[3 * p], cam[3 * n + 1] - pts[3 * p + 1], cam[3 * n + 2] - pts[3 * p + 2]};
        normalize3(w, v);
        for (int l = 0; l < L; ++l) {
            const float *lv = lvec + ((size_t)n * L + l) * 3;
            float u[3] = {lv[0], lv[1], lv[2]}, d[3];
            if (point_lights) { u[0] -= pts[3 * p]; u[1] -= pts[3 * p + 1]; u[2] -= pts[3 * p + 2]; }
            normalize3(u, d);
            const float ca = nh[0] * d[0] + nh[1] * d[1] + nh[2] * d[2];
            const float r[3] = {-d[0] + 2.0f * (ca * nh[0]), -d[1] + 2.0f * (ca * nh[1]), -d[2] + 2.0f * (ca * nh[2])};
            float a0 = v[0] * r[0] + v[1] * r[1] + v[2] * r[2];
            if (a0 < 0) a0 = 0;
            const float alpha = ca > 0 ? a0 : 0.0f;
            const float D = ca > 0 ? ca : 0.0f, S = powf(alpha, shininess);
            for (int ch = 0; ch < 3; ++ch) {
                dif[ch] += kd[((size_t)n * L + l) * 3 + ch] * D;
                spec[ch] += ks[((size_t)n * L + l) * 3 + ch] * S;
            }
        }
        for (int ch = 0; ch < 3; ++ch) {
            out[3 * p + ch] = rgb[3 * p + ch] * (ambient[3 * n + ch] + dif[ch]) + spec[ch];
            if (out_diffuse) out_diffuse[3 * p + ch] = dif[ch];
            if (out_specular) out_specular[3 * p + ch] = spec[ch];
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Point-cloud regularisers of the training iteration (DSS/training/losses.py:145-459; trainer.py:134-137 builds
 * both with knn_k = 12, dss.yml:30 weights the projection term with 0.01).  Restated in double precision on the
 * PACKED layout; the neighbour lists are the self query knn_points(p, p, K = knn_k) with the point itself as entry
 * 0, entries 1..K-1 are the neighbourhood (losses.py:177-179 drops column 0).  knn_idx are cloud-local ids.
 *
 *   phi_k      = max(0, 1 - d_k / (4 mean_k d_k))^4                                  (get_phi, :262-278)
 *   mollified  = sum_k phi_k n_j / eps_denom(sum_k phi_k); points with keep[p] (visibility & inmask) keep n_p
 *                (_denoise_normals, :181-222; normals are NOT re-normalised)
 *   normal_w_k = exp(-|nm_j/|nm_j| - nm_i/|nm_i||^2 / sigma^2)     F.normalize eps 1e-12 (get_normal_w, :224-246)
 *   projection (:296-392):  w_k = phi_k normal_w_k (visible[j] ? 1 : 0.1);  sdf_k = (x_j - p_i) . nm_j
 *                loss_i = sum_k w_k sdf_k^2 / eps_denom(sum_k w_k);   only p_i carries gradient
 *   repulsion  (:395-492):  s_k = exp(-|x_j - p_i|^2 inv_sigma[n])  (get_spatial_w :248-260; inv_sigma = num_points /
 *                bbox_diag^2 * filter_scale);  w_k = s_k normal_w_k;  density = 1 + sum_k s_k
 *                proj_k = (p_i - x_j) - ((p_i - x_j) . nm_j) nm_j
 *                r = sum_k proj_k w_k / eps_denom(sum_k w_k) * density;   loss_ic = exp(-|r_c|)  (3 per point)
 * Gradients are those autograd produces with every weight detached: d loss_i / d p_i only.
 * ------------------------------------------------------------------------------------------- */
static inline double eps_denom_d(double d)
{
    const double s = d > 0 ? 1.0 : (d < 0 ? -1.0 : 1.0);
    const double a = fabs(d);
    return s * (a > 1e-17 ? a : 1e-17);
}

DSS_ORACLE_API void oracle_mollify_normals(const float *normals, const float *knn_d2, const int64_t *knn_idx,
                                           const uint8_t *keep /* or NULL */, const int64_t *first_of /* (P,) packed
                                           id of the first point of p's cloud */, int64_t P, int K, float *out)
{
    for (int64_t p = 0; p < P; ++p) {
        double mean = 0;
        for (int k = 1; k < K; ++k) mean += (double)knn_d2[p * K + k];
        mean /= (K - 1);
        const double h = 4.0 * mean;
        double acc[3] = {0, 0, 0}, wsum = 0;
        for (int k = 1; k < K; ++k) {
            double w = 1.0 - (double)knn_d2[p * K + k] / h;
            if (w < 0) w = 0;
            w = w * w; w = w * w;
            const int64_t j = first_of[p] + knn_idx[p * K + k];
            for (int d = 0; d < 3; ++d) acc[d] += w * (double)normals[3 * j + d];
            wsum += w;
        }
        const int kept = keep && keep[p];
        for (int d = 0; d < 3; ++d) out[3 * p + d] = kept ? normals[3 * p + d] : (float)(acc[d] / eps_denom_d(wsum));
    }
}

static void unit3d(const float *v, double *o)
{
    double n = sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]);
    if (n < 1e-12) n = 1e-12;
    for (int d = 0; d < 3; ++d) o[d] = (double)v[d] / n;
}

DSS_ORACLE_API void oracle_projection_loss(const float *points, const float *mollified, const float *knn_d2,
                                           const int64_t *knn_idx, const uint8_t *visible /* or NULL */,
                                           const int64_t *first_of, int64_t P, int K, float sigma,
                                           const float *grad_loss /* (P,) or NULL */, float *loss /* (P,) */,
                                           float *grad_points /* (P,3) or NULL */)
{
    const double inv_s = 1.0 / ((double)sigma * (double)sigma);
    for (int64_t p = 0; p < P; ++p) {
        double mean = 0;
        for (int k = 1; k < K; ++k) mean += (double)knn_d2[p * K + k];
        const double h = 4.0 * mean / (K - 1);
        double ni[3];
        unit3d(mollified + 3 * p, ni);
        double num = 0, den = 0, g[3] = {0, 0, 0};
        for (int k = 1; k < K; ++k) {
            double phi = 1.0 - (double)knn_d2[p * K + k] / h;
            if (phi < 0) phi = 0;
            phi = phi * phi; phi = phi * phi;
            const int64_t j = first_of[p] + knn_idx[p * K + k];
            double nj[3], dn = 0, sdf = 0;
            unit3d(mollified + 3 * j, nj);
            for (int d = 0; d < 3; ++d) {
                dn += (nj[d] - ni[d]) * (nj[d] - ni[d]);
                sdf += ((double)points[3 * j + d] - (double)points[3 * p + d]) * (double)mollified[3 * j + d];
            }
            const double w = phi * exp(-dn * inv_s) * ((!visible || visible[j]) ? 1.0 : (double)0.1f);
            num += w * sdf * sdf;
            den += w;
            for (int d = 0; d < 3; ++d) g[d] += -2.0 * w * sdf * (double)mollified[3 * j + d];
        }
        den = eps_denom_d(den);
        loss[p] = (float)(num / den);
        if (grad_points)
            for (int d = 0; d < 3; ++d) grad_points[3 * p + d] = (float)((grad_loss ? grad_loss[p] : 1.0f) * g[d] / den);
    }
}

DSS_ORACLE_API void oracle_repulsion_loss(const float *points, const float *mollified, const int64_t *knn_idx,
                                          const int64_t *first_of, const float *inv_sigma_of /* (P,) per point's
                                          cloud */, int64_t P, int K, float sigma, const float *grad_loss /* (P,3) or
                                          NULL */, float *loss /* (P,3) */, float *grad_points /* (P,3) or NULL */)
{
    const double inv_s = 1.0 / ((double)sigma * (double)sigma);
    for (int64_t p = 0; p < P; ++p) {
        double ni[3];
        unit3d(mollified + 3 * p, ni);
        double acc[3] = {0, 0, 0}, wsum = 0, ssum = 0, A[3][3] = {{0}};
        for (int k = 1; k < K; ++k) {
            const int64_t j = first_of[p] + knn_idx[p * K + k];
            double nj[3], dn = 0, d2 = 0, df[3], dot = 0;
            unit3d(mollified + 3 * j, nj);
            for (int d = 0; d < 3; ++d) {
                dn += (nj[d] - ni[d]) * (nj[d] - ni[d]);
                df[d] = (double)points[3 * p + d] - (double)points[3 * j + d];
                d2 += df[d] * df[d];
                dot += df[d] * (double)mollified[3 * j + d];
            }
            const double s = exp(-d2 * (double)inv_sigma_of[p]);
            const double w = s * exp(-dn * inv_s);
            ssum += s;
            wsum += w;
            for (int d = 0; d < 3; ++d) {
                acc[d] += w * (df[d] - dot * (double)mollified[3 * j + d]);
                for (int e = 0; e < 3; ++e)
                    A[d][e] += w * ((d == e ? 1.0 : 0.0) - (double)mollified[3 * j + d] * (double)mollified[3 * j + e]);
            }
        }
        const double den = eps_denom_d(wsum), density = ssum + 1.0;
        double r[3], dl[3];
        for (int d = 0; d < 3; ++d) {
            r[d] = acc[d] / den * density;
            const double l = exp(-fabs(r[d]));
            loss[3 * p + d] = (float)l;
            dl[d] = (grad_loss ? (double)grad_loss[3 * p + d] : 1.0) * (r[d] > 0 ? -l : (r[d] < 0 ? l : 0.0));
        }
        if (grad_points)
            for (int e = 0; e < 3; ++e) {
                double g = 0;
                for (int d = 0; d < 3; ++d) g += dl[d] * A[d][e] / den * density;
                grad_points[3 * p + e] = (float)g;
            }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Image loss of the training iteration: Trainer.calc_dr_loss (DSS/training/trainer.py:332-372) with the loss
 * objects of Trainer.__init__ (:138-141), restated in double precision together with its gradient with respect
 * to the rendered RGBA image (what autograd hands to the renderer's backward):
 *   inside = mask != 0 and alpha != 0                          (mask_img.bool() & mask_img_pred.bool(), :351)
 *   rgb    = sum_inside sum_c |img_c - pred_c| / #inside       (L1Loss + BaseLoss: channel sum, mean; 0 if none)
 *   sil    = mean |mask - alpha| + 0.01 mean_n (1 - I_n / eps_denom(U_n)),  I = sum mask alpha,
 *            U = sum (alpha + mask - alpha mask)                (:361-367, IouLoss losses.py:498-513)
 *   total  = lambda_rgb rgb + lambda_sil sil
 * losses[4] = total, lambda_rgb rgb, lambda_sil sil, IoU term (unweighted mean of 1 - I/U).
 * ------------------------------------------------------------------------------------------- */
DSS_ORACLE_API void oracle_image_loss(const float *rgba /* (N,H,W,4) */, const float *img /* (N,H,W,3) */,
                                      const float *mask /* (N,H,W) */, int N, int H, int W, float lambda_rgb,
                                      float lambda_sil, float *losses /* (4) */, float *grad_rgba /* (N,H,W,4) or NULL */)
{
    const int64_t HW = (int64_t)H * W;
    double cnt = 0, srgb = 0, smask = 0, iou = 0;
    double *I = (double *)calloc((size_t)N, sizeof(double)), *U = (double *)calloc((size_t)N, sizeof(double));
    for (int n = 0; n < N; ++n)
        for (int64_t i = 0; i < HW; ++i) {
            const int64_t q = n * HW + i;
            const double a = rgba[4 * q + 3], t = mask[q];
            if (t != 0 && a != 0) {
                cnt += 1;
                for (int c = 0; c < 3; ++c) srgb += fabs((double)img[3 * q + c] - (double)rgba[4 * q + c]);
            }
            smask += fabs(t - a);
            I[n] += a * t;
            U[n] += a + t - a * t;
        }
    for (int n = 0; n < N; ++n) iou += 1.0 - I[n] / eps_denom_d(U[n]);
    iou /= N;
    const double rgb = cnt > 0 ? srgb / cnt : 0.0;
    const double sil = smask / ((double)N * HW) + 0.01 * iou;
    losses[0] = (float)(lambda_rgb * rgb + lambda_sil * sil);
    losses[1] = (float)(lambda_rgb * rgb);
    losses[2] = (float)(lambda_sil * sil);
    losses[3] = (float)iou;
    if (grad_rgba)
        for (int n = 0; n < N; ++n) {
            const double Un = eps_denom_d(U[n]);
            for (int64_t i = 0; i < HW; ++i) {
                const int64_t q = n * HW + i;
                const double a = rgba[4 * q + 3], t = mask[q];
                const int inside = t != 0 && a != 0;
                for (int c = 0; c < 3; ++c) {
                    const double d = (double)rgba[4 * q + c] - (double)img[3 * q + c];
                    grad_rgba[4 * q + c] = inside ? (float)(lambda_rgb * (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0)) / cnt) : 0.0f;
                }
                const double dm = a - t;
                /* d(1 - I/U)/da = -(t U - I (1 - t)) / U^2; with the denominator clamped (U == 0) only -t / eps is left */
                const double diou = fabs(U[n]) > 1e-17 ? -(t * Un - I[n] * (1.0 - t)) / (Un * Un) : -t / Un;
                grad_rgba[4 * q + 3] = (float)(lambda_sil * ((dm > 0 ? 1.0 : (dm < 0 ? -1.0 : 0.0)) / ((double)N * HW) +
                                                             0.01 * diou / N));
            }
        }
    free(I);
    free(U);
}

/* ---------------------------------------------------------------------------------------------
 * In-mask filter of the regularisers (DSS/models/point_modeling.py:183-208): for every view the target mask is
 * sampled at the projection of the point with get_tensor_values (DSS/utils/__init__.py:266-317) =
 * F.grid_sample(mask, p, mode='bilinear', padding_mode='reflection') [align_corners False], p = clamp(-ndc_xy, -1, 1);
 * inmask = any_n(value != 0) & visibility.  grid_sample is restated from its definition (ATen GridSampler: unnormalise
 * ((g + 1) size - 1) / 2, reflect about [-0.5, size - 0.5], clip to [0, size - 1], bilinear taps nw/ne/sw/se with
 * bounds checks); the projection is the row-vector product p_h @ M of pytorch3d's transform_points, in fp32 in the
 * order the HIP kernel uses (the reference's bmm may round differently: points that project within an ulp of a pixel
 * boundary can differ, see the pinning test).
 * ------------------------------------------------------------------------------------------- */
static float reflect_coord(float in, float twice_low, float twice_high)
{
    if (twice_low == twice_high) return 0.0f;
    const float mn = twice_low / 2, span = (twice_high - twice_low) / 2;
    in = fabsf(in - mn);
    const float extra = fmodf(in, span);
    const int flips = (int)floorf(in / span);
    return (flips % 2 == 0) ? extra + mn : span - extra + mn;
}

static float grid_sample_bilinear_reflect(const float *img, int H, int W, float gx, float gy)
{
    float ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    ix = reflect_coord(ix, -1.0f, 2.0f * W - 1.0f);
    iy = reflect_coord(iy, -1.0f, 2.0f * H - 1.0f);
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float x_nw = floorf(ix), y_nw = floorf(iy);
    const float x_se = x_nw + 1, y_se = y_nw + 1;
    const float nw = (x_se - ix) * (y_se - iy), ne = (ix - x_nw) * (y_se - iy);
    const float sw = (x_se - ix) * (iy - y_nw), se = (ix - x_nw) * (iy - y_nw);
    float out = 0.0f;
    const int x0 = (int)x_nw, y0 = (int)y_nw, x1 = x0 + 1, y1 = y0 + 1;
#define IN_IMG(yy, xx) ((yy) >= 0 && (yy) < H && (xx) >= 0 && (xx) < W)
    if (IN_IMG(y0, x0)) out += img[(size_t)y0 * W + x0] * nw;
    if (IN_IMG(y0, x1)) out += img[(size_t)y0 * W + x1] * ne;
    if (IN_IMG(y1, x0)) out += img[(size_t)y1 * W + x0] * sw;
    if (IN_IMG(y1, x1)) out += img[(size_t)y1 * W + x1] * se;
#undef IN_IMG
    return out;
}

DSS_ORACLE_API void oracle_grid_sample_points(const float *mask /* (N,H,W) */, const float *grid /* (N,P,2) */, int N,
                                              int64_t P, int H, int W, float *out /* (N,P) */)
{
    for (int n = 0; n < N; ++n)
        for (int64_t p = 0; p < P; ++p)
            out[n * P + p] = grid_sample_bilinear_reflect(mask + (size_t)n * H * W, H, W, grid[(n * P + p) * 2],
                                                          grid[(n * P + p) * 2 + 1]);
}

DSS_ORACLE_API void oracle_points_inmask(const float *points /* (P,3) */, const float *M /* (N,4,4) */,
                                         const float *mask /* (N,H,W) */, const uint8_t *visible /* (P,) or NULL */,
                                         int N, int64_t P, int H, int W, uint8_t *inmask)
{
    for (int64_t p = 0; p < P; ++p) {
        const float x = points[3 * p], y = points[3 * p + 1], z = points[3 * p + 2];
        int in = 0;
        for (int n = 0; n < N && !in; ++n) {
            const float *m = M + 16 * n;
            const float X = ((x * m[0] + y * m[4]) + z * m[8]) + m[12];
            const float Y = ((x * m[1] + y * m[5]) + z * m[9]) + m[13];
            const float Wc = ((x * m[3] + y * m[7]) + z * m[11]) + m[15];
            const float gx = fminf(fmaxf(-(X / Wc), -1.0f), 1.0f), gy = fminf(fmaxf(-(Y / Wc), -1.0f), 1.0f);
            if (!(gx == gx && gy == gy)) continue;
            in = grid_sample_bilinear_reflect(mask + (size_t)n * H * W, H, W, gx, gy) != 0.0f;
        }
        inmask[p] = (uint8_t)(in && (!visible || visible[p]));
    }
}
