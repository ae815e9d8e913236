// Phong shading of the points: the reference's LightingTexture
// (DSS/core/texture.py:65-125) = apply_lighting (:26-63) with the diffuse / specular terms of
// DSS/core/lighting.py:10-77 and :80-172 for PointLights (:239-302, direction = location - point) or
// DirectionalLights (:175-236), L lights per cloud:
//     n^ = normalize(normal), d^ = normalize(direction)           (F.normalize, eps 1e-6)
//     diffuse  = sum_l kd_l * relu(n^ . d^)
//     specular = sum_l ks_l * (relu(v^ . (-d^ + 2 (n^ . d^) n^)) * [n^ . d^ > 0]) ^ shininess,  v^ = normalize(camera - x)
//     out      = rgb * (ambient + diffuse) + specular
// This is where the RGB loss reaches the NORMALS (and, through point lights and the view direction, the
// positions): the EWA terms are constants for autograd (rasterizer.py:562-565).  One thread per world point; a
// cloud shared by N cameras is looped in camera order (deterministic sums, no atomics).
#include "common.h"

namespace dss {

struct PhongArgs {
    const float *world, *normals, *rgb;       // (Pw,3), (Pw,3), (P,3)
    const int64_t *first_idx, *num_pts;
    int N, shared, L, point_lights;
    int64_t Pw;
    const float *ambient, *kd, *ks, *lvec;    // (N,3), (N,L,3), (N,L,3), (N,L,3) location or direction
    const float *cam;                         // (N,3) camera centres
    float shininess;
};

__device__ __forceinline__ float safe_norm(float x, float y, float z) { return fmaxf(sqrtf(x * x + y * y + z * z), 1e-6f); }

// d/du of u / max(|u|, eps) applied to an upstream gradient g
__device__ __forceinline__ void normalize_backward(const float u[3], const float g[3], float out[3])
{
    const float raw = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (raw > 1e-6f) {
        const float inv = 1.0f / raw;
        const float h[3] = {u[0] * inv, u[1] * inv, u[2] * inv};
        const float dot = h[0] * g[0] + h[1] * g[1] + h[2] * g[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = (g[i] - h[i] * dot) * inv;
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = g[i] * 1e6f;  // clamped denominator: a constant scale
    }
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void phong_kernel(const PhongArgs A, const float *__restrict__ grad_out,
                                                    float *__restrict__ out, float *__restrict__ grad_world,
                                                    float *__restrict__ grad_normals, float *__restrict__ grad_rgb)
{
    const int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= A.Pw) return;
    const float x[3] = {A.world[3 * wi], A.world[3 * wi + 1], A.world[3 * wi + 2]};
    const float m[3] = {A.normals[3 * wi], A.normals[3 * wi + 1], A.normals[3 * wi + 2]};
    const float mn = safe_norm(m[0], m[1], m[2]);
    const float nh[3] = {m[0] / mn, m[1] / mn, m[2] / mn};
    float gx[3] = {0.f, 0.f, 0.f}, gm[3] = {0.f, 0.f, 0.f};
    const int n_lo = A.shared ? 0 : find_cloud(wi, A.first_idx, A.num_pts, A.N);
    const int n_hi = A.shared ? A.N : n_lo + 1;
    for (int n = max(n_lo, 0); n < n_hi && n_lo >= 0; ++n) {
        int64_t p = wi;
        if (A.shared) {
            if (wi >= A.num_pts[n]) continue;
            p = A.first_idx[n] + wi;
        }
        const float c[3] = {A.rgb[3 * p], A.rgb[3 * p + 1], A.rgb[3 * p + 2]};
        float g[3] = {0.f, 0.f, 0.f};
        if (BACKWARD) { g[0] = grad_out[3 * p]; g[1] = grad_out[3 * p + 1]; g[2] = grad_out[3 * p + 2]; }
        const float w[3] = {A.cam[3 * n] - x[0], A.cam[3 * n + 1] - x[1], A.cam[3 * n + 2] - x[2]};
        const float wn = safe_norm(w[0], w[1], w[2]);
        const float v[3] = {w[0] / wn, w[1] / wn, w[2] / wn};
        float dif[3] = {0.f, 0.f, 0.f}, spec[3] = {0.f, 0.f, 0.f};
        float gn[3] = {0.f, 0.f, 0.f}, gv[3] = {0.f, 0.f, 0.f};  // d loss / d n^, d v^ (this camera)
        for (int l = 0; l < A.L; ++l) {
            const float *lv = A.lvec + ((size_t)n * A.L + l) * 3;
            const float *kd = A.kd + ((size_t)n * A.L + l) * 3;
            const float *ks = A.ks + ((size_t)n * A.L + l) * 3;
            float u[3] = {lv[0], lv[1], lv[2]};
            if (A.point_lights) { u[0] -= x[0]; u[1] -= x[1]; u[2] -= x[2]; }
            const float un = safe_norm(u[0], u[1], u[2]);
            const float d[3] = {u[0] / un, u[1] / un, u[2] / un};
            const float ca = nh[0] * d[0] + nh[1] * d[1] + nh[2] * d[2];
            const float r[3] = {-d[0] + 2.0f * (ca * nh[0]), -d[1] + 2.0f * (ca * nh[1]), -d[2] + 2.0f * (ca * nh[2])};
            const float a0 = v[0] * r[0] + v[1] * r[1] + v[2] * r[2];
            const bool lit = ca > 0.0f;
            const float alpha = lit ? fmaxf(a0, 0.0f) : 0.0f;
            const float D = fmaxf(ca, 0.0f);
            const float S = powf(alpha, A.shininess);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                dif[ch] += kd[ch] * D;
                spec[ch] += ks[ch] * S;
            }
            if (BACKWARD) {
                const float gd = g[0] * c[0] * kd[0] + g[1] * c[1] * kd[1] + g[2] * c[2] * kd[2];   // d loss / d D
                const float gs = g[0] * ks[0] + g[1] * ks[1] + g[2] * ks[2];                        // d loss / d S
                float gca = lit ? gd : 0.0f;
                const float ga0 = (lit && a0 > 0.0f) ? gs * A.shininess * powf(alpha, A.shininess - 1.0f) : 0.0f;
                float gdv[3];  // d loss / d d^
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    gv[i] += ga0 * r[i];
                    gdv[i] = -ga0 * v[i];                       // r = -d^ + ...
                }
                const float gr_n = ga0 * (v[0] * nh[0] + v[1] * nh[1] + v[2] * nh[2]);
                gca += 2.0f * gr_n;                             // r = ... + 2 ca n^
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    gn[i] += 2.0f * ca * ga0 * v[i] + gca * d[i];
                    gdv[i] += gca * nh[i];
                }
                if (A.point_lights) {                           // u = location - x
                    float gu[3];
                    normalize_backward(u, gdv, gu);
#pragma unroll
                    for (int i = 0; i < 3; ++i) gx[i] -= gu[i];
                }
            }
        }
        const float *amb = A.ambient + 3 * n;
        if (!BACKWARD) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) out[3 * p + ch] = c[ch] * (amb[ch] + dif[ch]) + spec[ch];
        } else {
            if (grad_rgb) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) grad_rgb[3 * p + ch] = g[ch] * (amb[ch] + dif[ch]);
            }
            float gw[3], gmi[3];
            normalize_backward(w, gv, gw);                      // w = camera - x
            normalize_backward(m, gn, gmi);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gx[i] -= gw[i];
                gm[i] += gmi[i];
            }
        }
    }
    if (BACKWARD) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (grad_world) grad_world[3 * wi + i] = gx[i];
            if (grad_normals) grad_normals[3 * wi + i] = gm[i];
        }
    }
}

}  // namespace dss

using namespace dss;

static int phong_args(const char *who, PhongArgs &A, const float *world, const float *normals, const float *rgb,
                      const int64_t *first_idx, const int64_t *num_pts, int N, int64_t Pw, int shared_cloud,
                      const float *ambient, const float *diffuse_color, const float *specular_color,
                      const float *light_vec, int L, int point_lights, const float *cam_center, float shininess)
{
    if (N <= 0 || Pw < 0 || L < 0) {
        set_error("%s: bad sizes N=%d Pw=%lld L=%d", who, N, (long long)Pw, L);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (Pw > 0 && (!world || !normals || !rgb || !first_idx || !num_pts || !ambient || !cam_center ||
                   (L > 0 && (!diffuse_color || !specular_color || !light_vec)))) {
        set_error("%s: NULL tensor pointer", who);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    A.world = world; A.normals = normals; A.rgb = rgb; A.first_idx = first_idx; A.num_pts = num_pts; A.N = N;
    A.shared = shared_cloud; A.L = L; A.point_lights = point_lights; A.Pw = Pw; A.ambient = ambient; A.kd = diffuse_color;
    A.ks = specular_color; A.lvec = light_vec; A.cam = cam_center; A.shininess = shininess;
    return DSS_OK;
}

extern "C" int dss_phong_forward(const float *world, const float *normals, const float *rgb, const int64_t *first_idx,
                                 const int64_t *num_pts, int N, int64_t Pw, int shared_cloud, const float *ambient,
                                 const float *diffuse_color, const float *specular_color, const float *light_vec, int L,
                                 int point_lights, const float *cam_center, float shininess, float *out, void *stream)
{
    PhongArgs A;
    int rc = phong_args("dss_phong_forward", A, world, normals, rgb, first_idx, num_pts, N, Pw, shared_cloud, ambient,
                        diffuse_color, specular_color, light_vec, L, point_lights, cam_center, shininess);
    if (rc) return rc;
    if (Pw == 0) return DSS_OK;
    if (!out) { set_error("dss_phong_forward: NULL output"); return DSS_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(phong_kernel<false>, dim3((unsigned)((Pw + 255) / 256)), dim3(256), 0, as_stream(stream), A, nullptr,
                       out, nullptr, nullptr, nullptr);
    return check_launch("dss_phong_forward");
}

extern "C" int dss_phong_backward(const float *grad_out, const float *world, const float *normals, const float *rgb,
                                  const int64_t *first_idx, const int64_t *num_pts, int N, int64_t Pw, int shared_cloud,
                                  const float *ambient, const float *diffuse_color, const float *specular_color,
                                  const float *light_vec, int L, int point_lights, const float *cam_center,
                                  float shininess, float *grad_world, float *grad_normals, float *grad_rgb, void *stream)
{
    PhongArgs A;
    int rc = phong_args("dss_phong_backward", A, world, normals, rgb, first_idx, num_pts, N, Pw, shared_cloud, ambient,
                        diffuse_color, specular_color, light_vec, L, point_lights, cam_center, shininess);
    if (rc) return rc;
    if (Pw == 0) return DSS_OK;
    if (!grad_out) { set_error("dss_phong_backward: NULL grad_out"); return DSS_ERR_INVALID_ARGUMENT; }
    hipLaunchKernelGGL(phong_kernel<true>, dim3((unsigned)((Pw + 255) / 256)), dim3(256), 0, as_stream(stream), A, grad_out,
                       nullptr, grad_world, grad_normals, grad_rgb);
    return check_launch("dss_phong_backward");
}
