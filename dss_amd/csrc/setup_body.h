// Per-point setup body shared by point_setup_kernel (setup.hip) and the fused setup+bin_count kernel
// (raster_forward.hip).  See setup.hip for the reference mapping.
#pragma once
#include "common.h"

#ifndef DSS_SETUP_MARK
#define DSS_SETUP_MARK(slot)   // (timing builds of raster_forward.hip stamp the setup's phases: tools/setup_timing.py)
#endif

namespace dss {

__device__ __forceinline__ float eps_sqrt_py(float d) { return fmaxf(fabsf(d), 1e-17f); }  // mathHelper.py:16-21

struct SetupArgs {
    const float *world, *normals;     // (Pw,3)
    const float *h_point;             // (Pw,) or nullptr; with h_cloud also given: (P,) per packed point
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

// Everything the setup of one point produces (the values of the output arrays at index p).
struct SetupVals {
    float sx, sy, sz, ea, eb, ec, rx, ry, sc, fr0, fr1, fr2;
    uint8_t ok;
};

// Setup of packed point p of cloud n (n < 0: unowned point -> culled): the arithmetic alone.
__device__ __forceinline__ SetupVals setup_point_compute(const SetupArgs &A, int64_t p, int n)
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
#ifdef DSS_FINE_TIMING
        asm volatile("" ::"v"(ph0), "v"(ph2), "v"(n0), "v"(n2));   // (the inputs have arrived)
        DSS_SETUP_MARK(7);
#endif
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
            // h_point AND h_cloud given: h_point holds one value per PACKED point (a shared cloud whose cameras cull differently:
            // the reference evaluates the isotropic scale on each camera's filtered cloud, rasterizer.py:344-402)
            const float hh = aniso ? 0.0f : (A.h_point ? A.h_point[A.h_cloud ? p : wi] : A.h_cloud[n]);
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
    SetupVals v;
    v.sx = sx; v.sy = sy; v.sz = sz; v.ea = ea; v.eb = eb; v.ec = ec; v.rx = rx; v.ry = ry; v.sc = sc;
    v.fr0 = fr0; v.fr1 = fr1; v.fr2 = fr2; v.ok = ok;
    return v;
}

// one thread writes the outputs of its point.  full = false (DSS_WS_BAND_OUTPUTS, a splat that cannot reach the rank's row
// band): only what the backward of OTHER bands' visible points needs -- screen position, radii, validity; the ellipse, the
// scaler, the cutoff and the packed record (84 of the ~105 bytes a point writes) are read for binned splats only.
__device__ __forceinline__ void setup_point_store(const SetupArgs &A, int64_t p, const SetupVals &v, bool full = true)
{
    A.screen[3 * p] = v.sx; A.screen[3 * p + 1] = v.sy; A.screen[3 * p + 2] = v.sz;
    A.radii[2 * p] = v.rx; A.radii[2 * p + 1] = v.ry;
    A.valid[p] = v.ok;
    if (!full) return;
    A.ellipse[3 * p] = v.ea; A.ellipse[3 * p + 1] = v.eb; A.ellipse[3 * p + 2] = v.ec;
    A.scaler[p] = v.sc;
    A.cutoff[p] = A.cutoffC;
    if (A.rec) {
        float4 *R = A.rec + 4 * (size_t)p;
        R[0] = make_float4(v.sx, v.sy, v.rx, v.ry);
        R[1] = make_float4(v.ea, v.eb, v.ec, A.cutoffC);
        R[2] = make_float4(v.sc, v.fr0, v.fr1, v.fr2);
        R[3] = make_float4(v.sz, 0.0f, 0.0f, 0.0f);
    }
}

// Writes every output array and returns the screen-space record the binning pass needs (px, py, pz, rx, ry).
__device__ __forceinline__ void setup_point(const SetupArgs &A, int64_t p, int n, float &o_px, float &o_py, float &o_pz,
                                            float &o_rx, float &o_ry)
{
    const SetupVals v = setup_point_compute(A, p, n);
    setup_point_store(A, p, v);
    o_px = v.sx; o_py = v.sy; o_pz = v.sz; o_rx = v.rx; o_ry = v.ry;
}

// The same stores for the 64 CONSECUTIVE points [p0, p0 + 64) of a wavefront (lane l holds point p0 + l), transposed through
// 4 KB of LDS owned by the wavefront so that every store instruction writes one contiguous run: the (P,3) arrays as 48
// float4 (instead of three dword stores at a 12-byte lane stride), the 64-byte records as four runs of 1 KB (instead of four
// float4 stores at a 64-byte lane stride, every lane in a cache line of its own).  Needs 16-byte aligned screen / ellipse /
// record arrays and an 8-byte aligned radii array (`wide`, wave-uniform; the caller falls back to setup_point_store).
__device__ __forceinline__ bool setup_wide_ok(const SetupArgs &A)
{
    return ((((uintptr_t)A.screen | (uintptr_t)A.ellipse | (uintptr_t)A.rec) & 15u) | ((uintptr_t)A.radii & 7u)) == 0;
}
__device__ __forceinline__ void setup_wave_store(const SetupArgs &A, int64_t p0, const SetupVals &v, float4 *lds /* [256] */)
{
    const int lane = threadIdx.x & 63;
    float *l1 = reinterpret_cast<float *>(lds);
    l1[3 * lane] = v.sx; l1[3 * lane + 1] = v.sy; l1[3 * lane + 2] = v.sz;
    l1[192 + 3 * lane] = v.ea; l1[192 + 3 * lane + 1] = v.eb; l1[192 + 3 * lane + 2] = v.ec;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 48) {
        const float4 a = lds[lane], b = lds[48 + lane];
        reinterpret_cast<float4 *>(A.screen + 3 * p0)[lane] = a;
        reinterpret_cast<float4 *>(A.ellipse + 3 * p0)[lane] = b;
    }
    const int64_t p = p0 + lane;
    reinterpret_cast<float2 *>(A.radii)[p] = make_float2(v.rx, v.ry);
    A.scaler[p] = v.sc;
    A.cutoff[p] = A.cutoffC;
    A.valid[p] = v.ok;
    if (A.rec) {
        __builtin_amdgcn_wave_barrier();
        // record quarter k of lane l at float4 slot 4 l + k, rotated by l / 4 within the record so that the 16 lanes of a
        // quarter-wave spread over the banks
        const int rot = (lane >> 2) & 3;
        lds[4 * lane + ((0 + rot) & 3)] = make_float4(v.sx, v.sy, v.rx, v.ry);
        lds[4 * lane + ((1 + rot) & 3)] = make_float4(v.ea, v.eb, v.ec, A.cutoffC);
        lds[4 * lane + ((2 + rot) & 3)] = make_float4(v.sc, v.fr0, v.fr1, v.fr2);
        lds[4 * lane + ((3 + rot) & 3)] = make_float4(v.sz, 0.0f, 0.0f, 0.0f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 *R = A.rec + 4 * (size_t)p0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // float4 slot s = 64 k + lane of the wave's 4 KB belongs to point s / 4, quarter s % 4
            const int s = 64 * k + lane, pt = s >> 2, q = s & 3;
            R[s] = lds[4 * pt + ((q + ((pt >> 2) & 3)) & 3)];
        }
        __builtin_amdgcn_wave_barrier();
    }
}


}  // namespace dss
