// Per-point setup body shared by point_setup_kernel (setup.hip) and the fused setup+bin_count kernel
// (raster_forward.hip).  See setup.hip for the reference mapping.
#pragma once
#include "common.h"

namespace dss {

__device__ __forceinline__ float eps_sqrt_py(float d) { return fmaxf(fabsf(d), 1e-17f); }  // mathHelper.py:16-21

struct SetupArgs {
    const float *world, *normals;     // (Pw,3)
    const float *h_point;             // (Pw,) or nullptr
    const float *h_cloud;             // (N,) or nullptr
    const float *vr6, *frame_n;       // anisotropic mode: (Pw,6) Vrk xx,xy,xz,yy,yz,zz + (Pw,3) PCA normal; or nullptr
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
    // optional packed splat records for the fine pass of the fused forward (nullptr: not written): 64 bytes per point,
    // {px,py,rx,ry} {a,b,c,cutoff} {scaler,f0,f1,f2} {pz,-,-,-}; feat = the (P,3) features copied into them
    float4 *rec;
    const float *feat;
};

// Setup of packed point p of cloud n (n < 0: unowned point -> culled).  Writes every output array and
// returns the screen-space record the binning pass needs (px, py, pz, rx, ry).
__device__ __forceinline__ void setup_point(const SetupArgs &A, int64_t p, int n, float &o_px, float &o_py, float &o_pz,
                                            float &o_rx, float &o_ry)
{
    float sx = 0.f, sy = 0.f, sz = -1.0f, ea = 1.f, eb = 0.f, ec = 1.f, rx = 0.f, ry = 0.f, sc = 0.f;
    uint8_t ok = 0;
    // the features that ride in the packed record: requested FIRST (their address only depends on p).  Left next to the
    // record stores at the end of the function the load sat behind every output store (the compiler may not move it above
    // stores that could alias) and added one cold memory round trip to the binning kernel's dependent chain (-0.6 us per step).
    float fr0 = 0.f, fr1 = 0.f, fr2 = 0.f;
    if (A.rec) {
        const float *f = A.feat + 3 * (size_t)p;
        fr0 = f[0]; fr1 = f[1]; fr2 = f[2];
    }
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
            // anisotropic mode (rasterizer.py:256-291): Vrk is an input and the tangent frame of det(Sk WJk) is
            // the PCA frame, whose normal replaces the cloud normal below (culling above used the cloud normal)
            const bool aniso = A.vr6 != nullptr;
            const float f0 = aniso ? A.frame_n[3 * wi] : n0, f1 = aniso ? A.frame_n[3 * wi + 1] : n1,
                        f2 = aniso ? A.frame_n[3 * wi + 2] : n2;
            const float hh = aniso ? 0.0f : (A.h_point ? A.h_point[wi] : A.h_cloud[n]);
            // Sk^T Sk = I - n^ n^^T with the NORMALISED normal (rasterizer.py:337-341); zero normal -> 0
            const float nlen = sqrtf(f0 * f0 + f1 * f1 + f2 * f2);
            const float nden = nlen > 1e-12f ? nlen : 1e-12f;
            const float nn[3] = {f0 / nden, f1 / nden, f2 / nden};
            const float hv = nlen > 1e-12f ? hh : 0.0f;
            float Vr[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) Vr[i][j] = hv * ((i == j ? 1.0f : 0.0f) - nn[i] * nn[j]);
            if (aniso) {
                const float *q = A.vr6 + 6 * wi;
                Vr[0][0] = q[0]; Vr[0][1] = Vr[1][0] = q[1]; Vr[0][2] = Vr[2][0] = q[2];
                Vr[1][1] = q[3]; Vr[1][2] = Vr[2][1] = q[4]; Vr[2][2] = q[5];
            }
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
            // |det(Sk WJk)| = |n^ . (w0 x w1)| (Binet-Cauchy), see oracle_point_setup: well conditioned
            // for edge-on splats where sqrt(det Vk)/h cancels catastrophically
            const float cx0 = WJ[1][0] * WJ[2][1] - WJ[2][0] * WJ[1][1];
            const float cx1 = WJ[2][0] * WJ[0][1] - WJ[0][0] * WJ[2][1];
            const float cx2 = WJ[0][0] * WJ[1][1] - WJ[1][0] * WJ[0][1];
            const float absdetMk = nlen > 1e-12f ? fabsf(nn[0] * cx0 + nn[1] * cx1 + nn[2] * cx2) : 0.0f;
            (void)detVk;
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
    if (A.rec) {
        float4 *R = A.rec + 4 * (size_t)p;
        R[0] = make_float4(sx, sy, rx, ry);
        R[1] = make_float4(ea, eb, ec, A.cutoffC);
        R[2] = make_float4(sc, fr0, fr1, fr2);
        R[3] = make_float4(sz, 0.0f, 0.0f, 0.0f);
    }
    o_px = sx; o_py = sy; o_pz = sz; o_rx = rx; o_ry = ry;
}


}  // namespace dss
