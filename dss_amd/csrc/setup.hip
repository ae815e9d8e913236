// Fused per-point setup for gfx950: culling + projection + projection Jacobian + EWA variance +
// ellipse coefficients + axis-aligned radii + Gaussian normaliser, one thread per (camera, point).
//
// Replaces, in one pass over the points with no Python list rebuilds and no host syncs:
//   SurfaceSplatting.filter_renderable            DSS/core/rasterizer.py:219-254 (+ :148-217)
//   PointsRasterizer.transform (pytorch3d)        called at rasterizer.py:614
//   _compute_WJk                                  rasterizer.py:443-496
//   _compute_global_Vrk / _compute_isotropic_Vrk  rasterizer.py:293-402   (Vrk = h (I - n n^T))
//   _compute_variance_and_detMk                   rasterizer.py:404-441
//   _get_per_point_info + _get_ellipse_axis_aligned_radius   rasterizer.py:525-565, 498-523
// The arithmetic follows oracle/dss_oracle.c:oracle_point_setup operation by operation (fp32,
// no FMA contraction) so the two agree bit for bit.
//
// Culled points are not compacted away (that would need a host round trip for the new sizes):
// they stay in the packed arrays with valid=0 and z=-1, which every downstream kernel ignores
// (forward: pz<0, rasterize_points.cu:79-80; backward: never visible).
#include "setup_body.h"

namespace dss {

__global__ __launch_bounds__(256) void point_setup_kernel(const SetupArgs A)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    float px, py, pz, rx, ry;
    setup_point(A, p, find_cloud(p, A.first_idx, A.num_pts, A.N), px, py, pz, rx, ry);
}

// Backward of the projection (the autograd of pytorch3d's transform, rasterizer.py:614):
//   d ndc_x / d world = (M[:3,0] - ndc_x M[:3,3]) / w   (same for y),   d z / d world = V[:3,2]
// One thread per WORLD point; for a shared cloud the N cameras are summed in a fixed order
// (deterministic, no atomics).
// One thread per WORLD point; a cloud shared by the N cameras sums their contributions in camera order.  The per-camera
// inputs (valid flag, screen gradient, optionally the feature gradient) of up to eight cameras are requested together before
// any of them is used: as a loop of dependent loads with `continue` branches the eight-camera step of the multi-GPU bench
// spent 12.7 us here (one memory round trip per camera).  `grad_feat` != NULL also reduces the per-camera feature gradients
// (N Pw, C) of a shared cloud to the cloud's (Pw, C) -- the `grad.view(N, Pw, C).sum(0)` every caller of a shared cloud
// otherwise runs as a launch of its own.
__global__ __launch_bounds__(256) void project_backward_kernel(
    const float *__restrict__ world, const float *__restrict__ M, const float *__restrict__ V,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t Pw, int shared,
    const float *__restrict__ grad_screen, const uint8_t *__restrict__ valid, float clip,
    float *__restrict__ grad_world, const float *__restrict__ grad_feat, int C, float *__restrict__ grad_feat_world)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= Pw) return;
    const float x = world[3 * wi], y = world[3 * wi + 1], z = world[3 * wi + 2];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    float fs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // feature-gradient sums (C <= 8)
    const int n_lo = shared ? 0 : find_cloud(wi, first_idx, num_pts, N);
    const int n_hi = shared ? N : n_lo + 1;
    constexpr int NB = 8;
    for (int nb = max(n_lo, 0); nb < n_hi && n_lo >= 0; nb += NB) {
        int64_t pp[NB];
        bool on[NB];
        uint8_t vl[NB];
        float gs[NB][3];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int n = nb + u;
            on[u] = n < n_hi;
            pp[u] = wi;
            if (on[u] && shared) {
                on[u] = wi < num_pts[n];
                pp[u] = first_idx[n] + wi;
            }
            if (!on[u]) pp[u] = wi;   // (a valid address: masked below)
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            vl[u] = valid[pp[u]];
            gs[u][0] = grad_screen[3 * pp[u]]; gs[u][1] = grad_screen[3 * pp[u] + 1]; gs[u][2] = grad_screen[3 * pp[u] + 2];
        }
        if (grad_feat) {
#pragma unroll
            for (int u = 0; u < NB; ++u)
                if (on[u])
                    for (int ch = 0; ch < C; ++ch) fs[ch] += grad_feat[(size_t)pp[u] * C + ch];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (!on[u] || !vl[u]) continue;
            const int n = nb + u;
            const float *m = M + 16 * n;
            const float *v = V + 16 * n;
            const float cx = x * m[0] + y * m[4] + z * m[8] + m[12];
            const float cy = x * m[1] + y * m[5] + z * m[9] + m[13];
            const float w = x * m[3] + y * m[7] + z * m[11] + m[15];
            const float iw = 1.0f / w;
            const float nx = cx * iw, ny = cy * iw;
            float gx = gs[u][0], gy = gs[u][1], gz = gs[u][2];
            if (clip > 0.0f) {  // the per-point norm clip hook (rasterizer.py:667-673), same arithmetic as clip_grad_kernel
                const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
                const float sc = fminf(nrm, clip), den = fmaxf(nrm, 1e-12f);
                gx = gx / den * sc;
                gy = gy / den * sc;
                gz = gz / den * sc;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float jx = (m[i * 4 + 0] - nx * m[i * 4 + 3]) * iw;
                const float jy = (m[i * 4 + 1] - ny * m[i * 4 + 3]) * iw;
                const float t = jx * gx + jy * gy + v[i * 4 + 2] * gz;
                if (i == 0) g0 += t;
                else if (i == 1) g1 += t;
                else g2 += t;
            }
        }
    }
    grad_world[3 * wi] = g0; grad_world[3 * wi + 1] = g1; grad_world[3 * wi + 2] = g2;
    if (grad_feat)
        for (int ch = 0; ch < C; ++ch) grad_feat_world[(size_t)wi * C + ch] = fs[ch];
}

// PCA frames of the K-neighbourhoods -> anisotropic source variance (rasterizer.py:256-291, mathHelper.py:34-92):
// covariance of the K nearest points about their mean, cyclic Jacobi in fp32 on the trace-normalised matrix,
// Vrk = C - c0 e0 e0^T (= the two largest principal components), frame normal e0, curvatures ascending.
__global__ __launch_bounds__(256) void local_frames_kernel(const float *__restrict__ pts, const int64_t *__restrict__ knn_idx,
                                                           const int64_t *__restrict__ first_idx,
                                                           const int64_t *__restrict__ num_pts, int N, int64_t P, int K,
                                                           float *__restrict__ vr6, float *__restrict__ frame_n,
                                                           float *__restrict__ curv)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    float C[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    float e0[3] = {0.f, 0.f, 1.f}, lam[3] = {0.f, 0.f, 0.f};
    if (n >= 0) {
        const int64_t f0 = first_idx[n];
        const int kk = (int)min((int64_t)K, num_pts[n]);
        // differences to the query point first: the neighbourhood is tiny compared with the coordinates
        const float qx = pts[3 * p], qy = pts[3 * p + 1], qz = pts[3 * p + 2];
        float mx = 0.f, my = 0.f, mz = 0.f;
        for (int k = 0; k < kk; ++k) {
            const int64_t j = f0 + knn_idx[p * K + k];
            mx += pts[3 * j] - qx; my += pts[3 * j + 1] - qy; mz += pts[3 * j + 2] - qz;
        }
        const float ik = 1.0f / (float)(kk > 0 ? kk : 1);
        mx *= ik; my *= ik; mz *= ik;
        for (int k = 0; k < kk; ++k) {
            const int64_t j = f0 + knn_idx[p * K + k];
            const float d[3] = {pts[3 * j] - qx - mx, pts[3 * j + 1] - qy - my, pts[3 * j + 2] - qz - mz};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) C[a][b] += d[a] * d[b] * ik;
        }
        const float tr = C[0][0] + C[1][1] + C[2][2];
        if (tr > 0.0f) {
            const float it = 1.0f / tr;
            float A[3][3], V[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) { A[a][b] = C[a][b] * it; V[a][b] = (a == b) ? 1.0f : 0.0f; }
            for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int qi = pi + 1; qi < 3; ++qi) {
                        const float apq = A[pi][qi];
                        if (fabsf(apq) > 1e-20f) {
                            const float theta = (A[qi][qi] - A[pi][pi]) / (2.0f * apq);
                            const float t = (theta >= 0.0f ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
                            const float c = 1.0f / sqrtf(t * t + 1.0f), sn = t * c;
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const float akp = A[k][pi], akq = A[k][qi];
                                A[k][pi] = c * akp - sn * akq;
                                A[k][qi] = sn * akp + c * akq;
                            }
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const float apk = A[pi][k], aqk = A[qi][k];
                                A[pi][k] = c * apk - sn * aqk;
                                A[qi][k] = sn * apk + c * aqk;
                            }
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const float vkp = V[k][pi], vkq = V[k][qi];
                                V[k][pi] = c * vkp - sn * vkq;
                                V[k][qi] = sn * vkp + c * vkq;
                            }
                        }
                    }
            }
            // ascending eigenvalues without dynamic indexing
            const float l0 = A[0][0], l1 = A[1][1], l2 = A[2][2];
            const bool m0 = l0 <= l1 && l0 <= l2, m1 = !m0 && l1 <= l2;
            const float lmin = m0 ? l0 : (m1 ? l1 : l2);
            const float lmax = fmaxf(l0, fmaxf(l1, l2));
            const float lmid = (l0 + l1 + l2) - lmin - lmax;
#pragma unroll
            for (int k = 0; k < 3; ++k) e0[k] = m0 ? V[k][0] : (m1 ? V[k][1] : V[k][2]);
            lam[0] = lmin * tr; lam[1] = lmid * tr; lam[2] = lmax * tr;
        }
    }
    vr6[6 * p + 0] = C[0][0] - lam[0] * e0[0] * e0[0];
    vr6[6 * p + 1] = C[0][1] - lam[0] * e0[0] * e0[1];
    vr6[6 * p + 2] = C[0][2] - lam[0] * e0[0] * e0[2];
    vr6[6 * p + 3] = C[1][1] - lam[0] * e0[1] * e0[1];
    vr6[6 * p + 4] = C[1][2] - lam[0] * e0[1] * e0[2];
    vr6[6 * p + 5] = C[2][2] - lam[0] * e0[2] * e0[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        frame_n[3 * p + k] = e0[k];
        if (curv) curv[3 * p + k] = lam[k];
    }
}

}  // namespace dss

using namespace dss;

extern "C" int dss_local_frames(const float *points, const int64_t *knn_idx, const int64_t *first_idx,
                                const int64_t *num_pts, int N, int64_t P, int K, float *vr6, float *frame_n,
                                float *curvature, void *stream)
{
    if (N <= 0 || P < 0 || K < 1) {
        set_error("dss_local_frames: bad sizes N=%d P=%lld K=%d", N, (long long)P, K);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!points || !knn_idx || !first_idx || !num_pts || !vr6 || !frame_n) {
        set_error("dss_local_frames: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(local_frames_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), points,
                       knn_idx, first_idx, num_pts, N, P, K, vr6, frame_n, curvature);
    return check_launch("dss_local_frames");
}
extern "C" int dss_point_setup(const float *world, const float *normals, const float *h_point, const float *h_cloud,
                               const float *vr6, const float *frame_normals, const float *M, const float *V, const float *znear, const float *zfar,
                               const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int shared_cloud,
                               int backface_culling, int S, float cutoff_threshold, float antialiasing_sigma,
                               float *pts_screen, float *ellipse, float *radii, float *scaler, float *cutoff,
                               uint8_t *valid, void *stream)
{
    if (N <= 0 || P < 0 || S <= 0) {
        set_error("dss_point_setup: bad sizes N=%d P=%lld S=%d", N, (long long)P, S);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!world || !normals || (!h_point && !h_cloud && !vr6) || (vr6 && !frame_normals) || !M || !V || !znear || !zfar ||
        !first_idx || !num_pts || !pts_screen || !ellipse || !radii || !scaler || !cutoff || !valid) {
        set_error("dss_point_setup: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    SetupArgs A;
    A.world = world; A.normals = normals; A.h_point = h_point; A.h_cloud = h_cloud; A.M = M; A.V = V;
    A.vr6 = vr6; A.frame_n = frame_normals;
    A.znear = znear; A.zfar = zfar; A.first_idx = first_idx; A.num_pts = num_pts; A.N = N; A.P = P;
    A.shared = shared_cloud; A.backface = backface_culling; A.S = S; A.cutoffC = cutoff_threshold;
    A.sigma = antialiasing_sigma; A.screen = pts_screen; A.ellipse = ellipse; A.radii = radii; A.scaler = scaler;
    A.cutoff = cutoff; A.valid = valid; A.rec = nullptr; A.feat = nullptr;
    hipLaunchKernelGGL(point_setup_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), A);
    return check_launch("dss_point_setup");
}

// A cloud shared by N >= 2 cameras: EIGHT lanes per world point, lane c of a point's eight handles cameras c, c + 8, ...; the
// eight partial sums are added in a fixed tree (lane pairs 1, 2, 4 apart: deterministic), so a point's camera contributions
// are read and evaluated in parallel instead of one after the other by one thread (32,684 points x 8 cameras: 128 workgroups
// of threads that each walked eight cameras -- 10.7 us of the multi-GPU step -- become 1,022 workgroups).  The order of the
// additions differs from the one-thread kernel's (camera order) by float rounding only.
__global__ __launch_bounds__(256) void project_backward_shared_kernel(
    const float *__restrict__ world, const float *__restrict__ M, const float *__restrict__ V,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t Pw,
    const float *__restrict__ grad_screen, const uint8_t *__restrict__ valid, float clip,
    float *__restrict__ grad_world, const float *__restrict__ grad_feat, int C, float *__restrict__ grad_feat_world)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t wi = t >> 3;
    const int c0 = (int)(t & 7);
    const bool live = wi < Pw;
    const int64_t wc = live ? wi : Pw - 1;
    const float x = world[3 * wc], y = world[3 * wc + 1], z = world[3 * wc + 2];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    float fs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int n = c0; n < N; n += 8) {
        if (!live || wi >= num_pts[n]) continue;
        const int64_t p = first_idx[n] + wi;
        const uint8_t vl = valid[p];
        float gx = grad_screen[3 * p], gy = grad_screen[3 * p + 1], gz = grad_screen[3 * p + 2];
        if (grad_feat)
            for (int ch = 0; ch < C; ++ch) fs[ch] += grad_feat[(size_t)p * C + ch];
        if (!vl) continue;
        const float *m = M + 16 * n;
        const float *v = V + 16 * n;
        const float cx = x * m[0] + y * m[4] + z * m[8] + m[12];
        const float cy = x * m[1] + y * m[5] + z * m[9] + m[13];
        const float w = x * m[3] + y * m[7] + z * m[11] + m[15];
        const float iw = 1.0f / w;
        const float nx = cx * iw, ny = cy * iw;
        if (clip > 0.0f) {  // the per-point norm clip hook (rasterizer.py:667-673), same arithmetic as clip_grad_kernel
            const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
            const float sc = fminf(nrm, clip), den = fmaxf(nrm, 1e-12f);
            gx = gx / den * sc;
            gy = gy / den * sc;
            gz = gz / den * sc;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float jx = (m[i * 4 + 0] - nx * m[i * 4 + 3]) * iw;
            const float jy = (m[i * 4 + 1] - ny * m[i * 4 + 3]) * iw;
            const float tt = jx * gx + jy * gy + v[i * 4 + 2] * gz;
            if (i == 0) g0 += tt;
            else if (i == 1) g1 += tt;
            else g2 += tt;
        }
    }
    // sum over the point's eight lanes (two quad steps + the half-row mirror: plain VALU, fixed order)
    auto sum8 = [](float a) -> float {
        a += dpp_f32<0xB1>(a);                                       // quad_perm [1,0,3,2]: lanes 1 apart
        a += dpp_f32<0x4E>(a);                                       // quad_perm [2,3,0,1]: lanes 2 apart
        a += dpp_f32<0x141>(a);                                      // row_half_mirror: the other quad of the eight lanes
        return a;
    };
    g0 = sum8(g0); g1 = sum8(g1); g2 = sum8(g2);
    if (grad_feat)
        for (int ch = 0; ch < C; ++ch) fs[ch] = sum8(fs[ch]);
    if (live && c0 == 0) {
        grad_world[3 * wi] = g0; grad_world[3 * wi + 1] = g1; grad_world[3 * wi + 2] = g2;
        if (grad_feat)
            for (int ch = 0; ch < C; ++ch) grad_feat_world[(size_t)wi * C + ch] = fs[ch];
    }
}

static int project_backward_impl(const char *fn, const float *world, const float *M, const float *V, const int64_t *first_idx,
                                 const int64_t *num_pts, int N, int64_t Pw, int shared_cloud, const float *grad_screen,
                                 const uint8_t *valid, float clip, float *grad_world, const float *grad_feat, int C,
                                 float *grad_feat_world, void *stream)
{
    if (N <= 0 || Pw < 0) { set_error("%s: bad sizes", fn); return DSS_ERR_INVALID_ARGUMENT; }
    if (Pw == 0) return DSS_OK;
    if (!world || !M || !V || !first_idx || !num_pts || !grad_screen || !valid || !grad_world) {
        set_error("%s: NULL tensor pointer", fn);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (grad_feat && (!grad_feat_world || C < 1 || C > 8)) {
        set_error("%s: the feature-gradient reduction needs an output and 1 <= C <= 8 (C = %d)", fn, C);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    // (eight lanes per point while the launch is latency-bound; large clouds are bandwidth-bound and the one-thread kernel
    // with its eight cameras of loads in flight is faster there: 95 against 125 us at 8 x 1M points)
    if (shared_cloud && N >= 2 && Pw <= 262144)
        hipLaunchKernelGGL(project_backward_shared_kernel, dim3((unsigned)((Pw * 8 + 255) / 256)), dim3(256), 0, as_stream(stream),
                           world, M, V, first_idx, num_pts, N, Pw, grad_screen, valid, clip, grad_world, grad_feat, C,
                           grad_feat_world);
    else
        hipLaunchKernelGGL(project_backward_kernel, dim3((unsigned)((Pw + 255) / 256)), dim3(256), 0, as_stream(stream),
                           world, M, V, first_idx, num_pts, N, Pw, shared_cloud, grad_screen, valid, clip, grad_world, grad_feat, C,
                           grad_feat_world);
    return check_launch(fn);
}

extern "C" int dss_project_backward(const float *world, const float *M, const float *V, const int64_t *first_idx,
                                    const int64_t *num_pts, int N, int64_t Pw, int shared_cloud,
                                    const float *grad_screen, const uint8_t *valid, float clip, float *grad_world,
                                    void *stream)
{
    return project_backward_impl("dss_project_backward", world, M, V, first_idx, num_pts, N, Pw, shared_cloud, grad_screen, valid,
                                 clip, grad_world, nullptr, 0, nullptr, stream);
}

extern "C" int dss_project_backward_features(const float *world, const float *M, const float *V, const int64_t *first_idx,
                                             const int64_t *num_pts, int N, int64_t Pw, int shared_cloud,
                                             const float *grad_screen, const uint8_t *valid, float clip, float *grad_world,
                                             const float *grad_feat, int C, float *grad_feat_world, void *stream)
{
    return project_backward_impl("dss_project_backward_features", world, M, V, first_idx, num_pts, N, Pw, shared_cloud, grad_screen,
                                 valid, clip, grad_world, grad_feat, C, grad_feat_world, stream);
}
