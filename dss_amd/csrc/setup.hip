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
#include "common.h"

namespace dss {

__device__ __forceinline__ float eps_denom_py(float d)  // DSS/utils/mathHelper.py:10-14
{
    const float s = (float)((d > 0) - (d < 0)) + (d == 0.0f ? 1.0f : 0.0f);
    return s * fmaxf(fabsf(d), 1e-17f);
}
__device__ __forceinline__ float eps_sqrt_py(float d) { return fmaxf(fabsf(d), 1e-17f); }  // mathHelper.py:16-21

struct SetupArgs {
    const float *world, *normals;     // (Pw,3)
    const float *h_point;             // (Pw,) or nullptr
    const float *h_cloud;             // (N,) or nullptr
    const float *M, *V;               // (N,4,4) row-vector convention
    const float *znear, *zfar;        // (N,)
    const int64_t *first_idx, *num_pts;
    int N;
    int64_t P;
    int shared;                       // 1: every cloud reads world[p - first_idx[n]]
    int backface;
    int S;
    float cutoffC, sigma;
    float *screen, *ellipse, *radii, *scaler, *cutoff;
    uint8_t *valid;
};

__global__ __launch_bounds__(256) void point_setup_kernel(const SetupArgs A)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    const int n = find_cloud(p, A.first_idx, A.num_pts, A.N);
    float sx = 0.f, sy = 0.f, sz = -1.0f, ea = 1.f, eb = 0.f, ec = 1.f, rx = 0.f, ry = 0.f, sc = 0.f;
    uint8_t ok = 0;
    if (n >= 0) {
        const int64_t wi = A.shared ? (p - A.first_idx[n]) : p;
        const float *m = A.M + 16 * n;
        const float *v = A.V + 16 * n;
        const float ph0 = A.world[3 * wi], ph1 = A.world[3 * wi + 1], ph2 = A.world[3 * wi + 2], ph3 = 1.0f;
        const float n0 = A.normals[3 * wi], n1 = A.normals[3 * wi + 1], n2 = A.normals[3 * wi + 2];
        const float zview = ph0 * v[2] + ph1 * v[6] + ph2 * v[10] + ph3 * v[14];
        // _filter_points_with_invalid_depth, rasterizer.py:183-217
        ok = (zview >= A.znear[n]) && (zview <= A.zfar[n]);
        if (A.backface) {
            // _filter_backface_points, rasterizer.py:148-181: keep view-space normal z < 0.
            // transform_normals uses the inverse-transpose of the rotation block; R is orthonormal.
            const float nz = n0 * v[2] + n1 * v[6] + n2 * v[10];
            ok = ok && (nz < 0);
        }
        if (ok) {
            float clip[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) clip[j] = ph0 * m[j] + ph1 * m[4 + j] + ph2 * m[8 + j] + ph3 * m[12 + j];
            const float w = clip[3];
            sx = clip[0] / w;
            sy = clip[1] / w;
            sz = zview;
            const float dw = eps_denom_py(w), dw2 = eps_denom_py(w * w);
            float WJ[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    WJ[i][j] = m[i * 4 + j] * (1.0f / dw) + m[i * 4 + 3] * (-1.0f / dw2 * clip[j]);
            const float hh = A.h_point ? A.h_point[wi] : A.h_cloud[n];
            // Sk^T Sk = I - n^ n^^T with the NORMALISED normal (rasterizer.py:337-341); zero normal -> 0
            const float nlen = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
            const float nden = nlen > 1e-12f ? nlen : 1e-12f;
            const float nn[3] = {n0 / nden, n1 / nden, n2 / nden};
            const float hv = nlen > 1e-12f ? hh : 0.0f;
            float Vr[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Vr[i][j] = hv * ((i == j ? 1.0f : 0.0f) - nn[i] * nn[j]);
            float T[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) T[i][j] = Vr[i][0] * WJ[0][j] + Vr[i][1] * WJ[1][j] + Vr[i][2] * WJ[2][j];
            float Vk[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Vk[i][j] = WJ[0][i] * T[0][j] + WJ[1][i] * T[1][j] + WJ[2][i] * T[2][j];
            const float detVk = Vk[0][0] * Vk[1][1] - Vk[0][1] * Vk[1][0];
            const float absdetMk = sqrtf(detVk > 0.0f ? detVk : 0.0f) / hh;
            const float pixel = 2.0f / (float)A.S;
            const float G00 = Vk[0][0] + A.sigma * (pixel * pixel), G11 = Vk[1][1] + A.sigma * (pixel * pixel);
            const float G01 = Vk[0][1], G10 = Vk[1][0];
            const float detG = G00 * G11 - G01 * G10;
            ea = G11 / detG;
            ec = G00 / detG;
            eb = (-G01 / detG) + (-G10 / detG);
            const float den = eps_denom_py(4.0f * ea * ec - eb * eb);
            rx = sqrtf(eps_sqrt_py(4.0f * ec * A.cutoffC / den));
            ry = sqrtf(eps_sqrt_py(4.0f * ea * A.cutoffC / den));
            const float pi = 3.14159265358979323846f;
            const float s2 = sqrtf(eps_sqrt_py(detG * 4.0f * pi * pi));
            sc = absdetMk / eps_denom_py(s2);
        }
    }
    A.screen[3 * p] = sx; A.screen[3 * p + 1] = sy; A.screen[3 * p + 2] = sz;
    A.ellipse[3 * p] = ea; A.ellipse[3 * p + 1] = eb; A.ellipse[3 * p + 2] = ec;
    A.radii[2 * p] = rx; A.radii[2 * p + 1] = ry;
    A.scaler[p] = sc;
    A.cutoff[p] = A.cutoffC;
    A.valid[p] = ok;
}

// Backward of the projection (the autograd of pytorch3d's transform, rasterizer.py:614):
//   d ndc_x / d world = (M[:3,0] - ndc_x M[:3,3]) / w   (same for y),   d z / d world = V[:3,2]
// One thread per WORLD point; for a shared cloud the N cameras are summed in a fixed order
// (deterministic, no atomics).
__global__ __launch_bounds__(256) void project_backward_kernel(
    const float *__restrict__ world, const float *__restrict__ M, const float *__restrict__ V,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t Pw, int shared,
    const float *__restrict__ grad_screen, const uint8_t *__restrict__ valid, float *__restrict__ grad_world)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= Pw) return;
    const float x = world[3 * wi], y = world[3 * wi + 1], z = world[3 * wi + 2];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    const int n_lo = shared ? 0 : find_cloud(wi, first_idx, num_pts, N);
    const int n_hi = shared ? N : n_lo + 1;
    for (int n = max(n_lo, 0); n < n_hi && n_lo >= 0; ++n) {
        int64_t p = wi;
        if (shared) {
            if (wi >= num_pts[n]) continue;
            p = first_idx[n] + wi;
        }
        if (!valid[p]) continue;
        const float *m = M + 16 * n;
        const float *v = V + 16 * n;
        const float cx = x * m[0] + y * m[4] + z * m[8] + m[12];
        const float cy = x * m[1] + y * m[5] + z * m[9] + m[13];
        const float w = x * m[3] + y * m[7] + z * m[11] + m[15];
        const float iw = 1.0f / w;
        const float nx = cx * iw, ny = cy * iw;
        const float gx = grad_screen[3 * p], gy = grad_screen[3 * p + 1], gz = grad_screen[3 * p + 2];
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
    grad_world[3 * wi] = g0; grad_world[3 * wi + 1] = g1; grad_world[3 * wi + 2] = g2;
}

}  // namespace dss

using namespace dss;

extern "C" int dss_point_setup(const float *world, const float *normals, const float *h_point, const float *h_cloud,
                               const float *M, const float *V, const float *znear, const float *zfar,
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
    if (!world || !normals || (!h_point && !h_cloud) || !M || !V || !znear || !zfar || !first_idx || !num_pts ||
        !pts_screen || !ellipse || !radii || !scaler || !cutoff || !valid) {
        set_error("dss_point_setup: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    SetupArgs A;
    A.world = world; A.normals = normals; A.h_point = h_point; A.h_cloud = h_cloud; A.M = M; A.V = V;
    A.znear = znear; A.zfar = zfar; A.first_idx = first_idx; A.num_pts = num_pts; A.N = N; A.P = P;
    A.shared = shared_cloud; A.backface = backface_culling; A.S = S; A.cutoffC = cutoff_threshold;
    A.sigma = antialiasing_sigma; A.screen = pts_screen; A.ellipse = ellipse; A.radii = radii; A.scaler = scaler;
    A.cutoff = cutoff; A.valid = valid;
    hipLaunchKernelGGL(point_setup_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), A);
    return check_launch("dss_point_setup");
}

extern "C" int dss_project_backward(const float *world, const float *M, const float *V, const int64_t *first_idx,
                                    const int64_t *num_pts, int N, int64_t Pw, int shared_cloud,
                                    const float *grad_screen, const uint8_t *valid, float *grad_world, void *stream)
{
    if (N <= 0 || Pw < 0) { set_error("dss_project_backward: bad sizes"); return DSS_ERR_INVALID_ARGUMENT; }
    if (Pw == 0) return DSS_OK;
    if (!world || !M || !V || !first_idx || !num_pts || !grad_screen || !valid || !grad_world) {
        set_error("dss_project_backward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(project_backward_kernel, dim3((unsigned)((Pw + 255) / 256)), dim3(256), 0, as_stream(stream),
                       world, M, V, first_idx, num_pts, N, Pw, shared_cloud, grad_screen, valid, grad_world);
    return check_launch("dss_project_backward");
}
