// Point-cloud regularisers of the training iteration: the projection and repulsion terms of
// DSS/training/losses.py:145-459 (Trainer builds both with knn_k = 12, trainer.py:134-137; configs/dss.yml:30
// weights the projection term with 0.01, so it runs every iteration right after the render), on the PACKED
// neighbour lists of dss_knn_points (entry 0 is the point itself and is skipped, losses.py:177-179).
//
//     phi_k      = max(0, 1 - d_k / (4 mean_k d_k))^4                                     get_phi        :262-278
//     mollified  = sum_k phi_k n_j / eps_denom(sum_k phi_k)   (kept for visible & in-mask) _denoise_normals :181-222
//     normal_w_k = exp(-|nm^_j - nm^_i|^2 / sigma^2)          (F.normalize, eps 1e-12)     get_normal_w   :224-246
//   projection   w_k = phi_k normal_w_k (visible_j ? 1 : 0.1);  sdf_k = (x_j - p_i) . nm_j
//                loss_i = sum_k w_k sdf_k^2 / eps_denom(sum_k w_k)                                          :296-392
//   repulsion    s_k = exp(-|x_j - p_i|^2 N_n / diag_n^2 filter_scale);  w_k = s_k normal_w_k                :248-260
//                r = sum_k w_k (I - nm_j nm_j^T)(p_i - x_j) / eps_denom(sum_k w_k) (1 + sum_k s_k)
//                loss_i,c = exp(-|r_c|)                                                                     :395-492
// The reference evaluates these as ~40 padded (N, Pmax, K[, 3]) torch tensors per call (gathers, exps, masked
// writes) and mollifies the normals twice.  Here: one thread per point, the K-1 neighbours walked in registers, the
// mollified normals computed once and shared.  Every weight is a constant for autograd in the reference
// (torch.no_grad blocks, .detach() on the neighbour positions), so d loss_i / d p_i is closed form and is
// recomputed from the inputs in the backward call instead of being stored: no saved tensors besides the inputs.
// Neighbour gathers are random 12-byte reads that stay in L2 (a 100k-point cloud is 1.2 MB).
#include "common.h"

namespace dss {

struct Nbr {  // neighbourhood cursor of packed point p
    const float *d2;
    const int64_t *idx;
    int64_t first;
};

__device__ __forceinline__ bool open_neighbourhood(int64_t p, const float *knn_d2, const int64_t *knn_idx,
                                                   const int64_t *first_idx, const int64_t *num_pts, int N, int K, Nbr &nb)
{
    const int n = find_cloud(p, first_idx, num_pts, N);
    if (n < 0) return false;
    nb.d2 = knn_d2 ? knn_d2 + p * K : nullptr;
    nb.idx = knn_idx + p * K;
    nb.first = first_idx[n];
    return true;
}

__device__ __forceinline__ float support_radius(const float *d2, int K)  // h = 4 * mean of the K-1 squared distances
{
    float s = 0.f;
    for (int k = 1; k < K; ++k) s += d2[k];
    return (s / (float)(K - 1)) * 4.0f;
}

__device__ __forceinline__ float phi_weight(float d2, float h)
{
    float w = fmaxf(1.0f - d2 / h, 0.0f);  // NaN (h == 0: coincident neighbourhood) propagates like the reference
    w = (d2 / h != d2 / h) ? d2 / h : w;
    w *= w;
    return w * w;
}

__device__ __forceinline__ void load3(const float *a, int64_t i, float v[3])
{
    v[0] = a[3 * i]; v[1] = a[3 * i + 1]; v[2] = a[3 * i + 2];
}

__device__ __forceinline__ void unit3(const float v[3], float u[3])  // F.normalize(dim=-1), eps 1e-12
{
    const float inv = 1.0f / fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    u[0] = v[0] * inv; u[1] = v[1] * inv; u[2] = v[2] * inv;
}

__device__ __forceinline__ float sqdiff3(const float a[3], const float b[3])
{
    const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return x * x + y * y + z * z;
}

__global__ __launch_bounds__(256) void mollify_normals_kernel(const float *__restrict__ normals,
                                                              const float *__restrict__ knn_d2,
                                                              const int64_t *__restrict__ knn_idx,
                                                              const uint8_t *__restrict__ keep,
                                                              const int64_t *__restrict__ first_idx,
                                                              const int64_t *__restrict__ num_pts, int N, int64_t P, int K,
                                                              float *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float own[3];
    load3(normals, p, own);
    Nbr nb;
    if ((keep && keep[p]) || !open_neighbourhood(p, knn_d2, knn_idx, first_idx, num_pts, N, K, nb)) {
        out[3 * p] = own[0]; out[3 * p + 1] = own[1]; out[3 * p + 2] = own[2];
        return;
    }
    const float h = support_radius(nb.d2, K);
    float acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f;
    for (int k = 1; k < K; ++k) {
        const float w = phi_weight(nb.d2[k], h);
        float nj[3];
        load3(normals, nb.first + nb.idx[k], nj);
        acc[0] += w * nj[0]; acc[1] += w * nj[1]; acc[2] += w * nj[2];
        wsum += w;
    }
    const float den = eps_denom_py(wsum);
    out[3 * p] = acc[0] / den; out[3 * p + 1] = acc[1] / den; out[3 * p + 2] = acc[2] / den;
}

__global__ __launch_bounds__(256) void projection_loss_kernel(const float *__restrict__ points,
                                                              const float *__restrict__ mollified,
                                                              const float *__restrict__ knn_d2,
                                                              const int64_t *__restrict__ knn_idx,
                                                              const uint8_t *__restrict__ visible,
                                                              const int64_t *__restrict__ first_idx,
                                                              const int64_t *__restrict__ num_pts, int N, int64_t P, int K,
                                                              float inv_sigma2, const float *__restrict__ grad_loss,
                                                              float *__restrict__ loss, float *__restrict__ grad_points)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    Nbr nb;
    float num = 0.f, den = 0.f, g[3] = {0.f, 0.f, 0.f};
    if (open_neighbourhood(p, knn_d2, knn_idx, first_idx, num_pts, N, K, nb)) {
        float x[3], mi[3], ni[3];
        load3(points, p, x);
        load3(mollified, p, mi);
        unit3(mi, ni);
        const float h = support_radius(nb.d2, K);
        for (int k = 1; k < K; ++k) {
            const int64_t j = nb.first + nb.idx[k];
            float xj[3], mj[3], nj[3];
            load3(points, j, xj);
            load3(mollified, j, mj);
            unit3(mj, nj);
            const float vis_w = (!visible || visible[j]) ? 1.0f : 0.1f;
            const float w = phi_weight(nb.d2[k], h) * expf(-sqdiff3(nj, ni) * inv_sigma2) * vis_w;
            const float sdf = (xj[0] - x[0]) * mj[0] + (xj[1] - x[1]) * mj[1] + (xj[2] - x[2]) * mj[2];
            num += w * sdf * sdf;
            den += w;
            const float c = -2.0f * w * sdf;
            g[0] += c * mj[0]; g[1] += c * mj[1]; g[2] += c * mj[2];
        }
    }
    den = eps_denom_py(den);
    if (loss) loss[p] = num / den;
    if (grad_points) {
        const float s = (grad_loss ? grad_loss[p] : 1.0f) / den;
        grad_points[3 * p] = g[0] * s; grad_points[3 * p + 1] = g[1] * s; grad_points[3 * p + 2] = g[2] * s;
    }
}

__global__ __launch_bounds__(256) void repulsion_loss_kernel(const float *__restrict__ points,
                                                             const float *__restrict__ mollified,
                                                             const int64_t *__restrict__ knn_idx,
                                                             const int *__restrict__ bbox, float filter_scale,
                                                             const int64_t *__restrict__ first_idx,
                                                             const int64_t *__restrict__ num_pts, int N, int64_t P, int K,
                                                             float inv_sigma2, const float *__restrict__ grad_loss,
                                                             float *__restrict__ loss, float *__restrict__ grad_points)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    float acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f, ssum = 0.f;
    float a00 = 0.f, a01 = 0.f, a02 = 0.f, a11 = 0.f, a12 = 0.f, a22 = 0.f;  // sum_k w_k (I - nm nm^T), symmetric
    if (n >= 0) {
        // get_spatial_w: inv_sigma_spatial = num_points / |bbox diagonal|^2, then * filter_scale (losses.py:252-258)
        const float ex = ord2f(bbox[6 * n + 3]) - ord2f(bbox[6 * n]), ey = ord2f(bbox[6 * n + 4]) - ord2f(bbox[6 * n + 1]),
                    ez = ord2f(bbox[6 * n + 5]) - ord2f(bbox[6 * n + 2]);
        const float inv_sigma = (float)num_pts[n] / (ex * ex + ey * ey + ez * ez);
        const int64_t first = first_idx[n];
        const int64_t *idx = knn_idx + p * K;
        float x[3], mi[3], ni[3];
        load3(points, p, x);
        load3(mollified, p, mi);
        unit3(mi, ni);
        for (int k = 1; k < K; ++k) {
            const int64_t j = first + idx[k];
            float xj[3], mj[3], nj[3];
            load3(points, j, xj);
            load3(mollified, j, mj);
            unit3(mj, nj);
            const float df[3] = {x[0] - xj[0], x[1] - xj[1], x[2] - xj[2]};
            const float d2 = df[0] * df[0] + df[1] * df[1] + df[2] * df[2];
            const float s = expf(-d2 * inv_sigma * filter_scale);
            const float w = s * expf(-sqdiff3(nj, ni) * inv_sigma2);
            const float dot = df[0] * mj[0] + df[1] * mj[1] + df[2] * mj[2];
            ssum += s;
            wsum += w;
            acc[0] += w * (df[0] - dot * mj[0]); acc[1] += w * (df[1] - dot * mj[1]); acc[2] += w * (df[2] - dot * mj[2]);
            a00 += w * (1.0f - mj[0] * mj[0]); a11 += w * (1.0f - mj[1] * mj[1]); a22 += w * (1.0f - mj[2] * mj[2]);
            a01 -= w * mj[0] * mj[1]; a02 -= w * mj[0] * mj[2]; a12 -= w * mj[1] * mj[2];
        }
    }
    const float scale = (ssum + 1.0f) / eps_denom_py(wsum);  // density_w / sum of weights
    float dl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float r = acc[c] * scale;
        const float l = expf(-fabsf(r));
        if (loss) loss[3 * p + c] = l;
        dl[c] = (grad_loss ? grad_loss[3 * p + c] : 1.0f) * (r > 0.f ? -l : (r < 0.f ? l : 0.f)) * scale;
    }
    if (grad_points) {
        grad_points[3 * p] = dl[0] * a00 + dl[1] * a01 + dl[2] * a02;
        grad_points[3 * p + 1] = dl[0] * a01 + dl[1] * a11 + dl[2] * a12;
        grad_points[3 * p + 2] = dl[0] * a02 + dl[1] * a12 + dl[2] * a22;
    }
}

// In-mask filter of the regularisers (DSS/models/point_modeling.py:183-208): a point is "in mask" if, in ANY view, the
// target mask sampled at its projection is non-zero -- F.grid_sample(mask, -ndc_xy clamped to [-1,1], bilinear,
// padding_mode='reflection', align_corners=False) (utils/__init__.py:266-317) -- and it is visible.  The reference
// re-projects all points with pytorch3d (transform_points) and runs grid_sample, any() and & as separate passes over
// (N, P) tensors; here one thread per point loops over the cameras.  `M` is the full projection matrix of
// dss_point_setup (row vectors: p_h @ M).  With the sample position inside [-1,1] the reflection is the identity and
// the unnormalised coordinate is only clipped to [0, size-1]; the bilinear value is non-zero iff a tap with a
// non-zero weight lies on a non-zero mask pixel.
__global__ __launch_bounds__(256) void points_inmask_kernel(const float *__restrict__ points, const float *__restrict__ M,
                                                            const float *__restrict__ mask, const uint8_t *__restrict__ visible,
                                                            int N, int64_t P, int H, int W, uint8_t *__restrict__ inmask)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (visible && !visible[p]) {
        inmask[p] = 0;
        return;
    }
    const float x = points[3 * p], y = points[3 * p + 1], z = points[3 * p + 2];
    bool in = false;
    for (int n = 0; n < N && !in; ++n) {
        const float *m = M + 16 * n;
        const float X = ((x * m[0] + y * m[4]) + z * m[8]) + m[12];
        const float Y = ((x * m[1] + y * m[5]) + z * m[9]) + m[13];
        const float Wc = ((x * m[3] + y * m[7]) + z * m[11]) + m[15];
        const float gx = fminf(fmaxf(-(X / Wc), -1.0f), 1.0f), gy = fminf(fmaxf(-(Y / Wc), -1.0f), 1.0f);
        const float ix = fminf(fmaxf(((gx + 1.0f) * (float)W - 1.0f) / 2.0f, 0.0f), (float)(W - 1));
        const float iy = fminf(fmaxf(((gy + 1.0f) * (float)H - 1.0f) / 2.0f, 0.0f), (float)(H - 1));
        if (!(ix == ix && iy == iy)) continue;  // NaN projection (w == 0): never in mask
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
        const float fx = ix - (float)x0, fy = iy - (float)y0;  // weight of the +1 taps; 1 - f of the others
        const float *img = mask + (size_t)n * H * W;
        float v = img[(size_t)y0 * W + x0] * ((1.0f - fx) * (1.0f - fy));
        if (x0 + 1 < W) v += img[(size_t)y0 * W + x0 + 1] * (fx * (1.0f - fy));
        if (y0 + 1 < H) v += img[(size_t)(y0 + 1) * W + x0] * ((1.0f - fx) * fy);
        if (x0 + 1 < W && y0 + 1 < H) v += img[(size_t)(y0 + 1) * W + x0 + 1] * (fx * fy);
        in = v != 0.0f;
    }
    inmask[p] = in ? 1 : 0;
}

static int check_common(const char *who, int N, int64_t P, int K, const void *a, const void *b, const void *c, const void *d,
                        const void *e)
{
    if (N <= 0 || P < 0 || K < 2 || K > 40) {
        set_error("%s: bad sizes N=%d P=%lld K=%d (2 <= K <= 40, the self entry included)", who, N, (long long)P, K);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P > 0 && (!a || !b || !c || !d || !e)) {
        set_error("%s: NULL tensor pointer", who);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    return DSS_OK;
}

}  // namespace dss

using namespace dss;

extern "C" int dss_mollify_normals(const float *normals, const float *knn_d2, const int64_t *knn_idx, const uint8_t *keep,
                                   const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int K,
                                   float *normals_out, void *stream)
{
    if (int rc = check_common("dss_mollify_normals", N, P, K, normals, knn_d2, knn_idx, first_idx, num_pts)) return rc;
    if (P == 0) return DSS_OK;
    if (!normals_out || normals_out == normals) {
        set_error("dss_mollify_normals: normals_out must be a separate buffer (neighbours read the input normals)");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(mollify_normals_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), normals,
                       knn_d2, knn_idx, keep, first_idx, num_pts, N, P, K, normals_out);
    return check_launch("dss_mollify_normals");
}

extern "C" int dss_projection_loss(const float *points, const float *mollified, const float *knn_d2, const int64_t *knn_idx,
                                   const uint8_t *visible, const int64_t *first_idx, const int64_t *num_pts, int N,
                                   int64_t P, int K, float sharpness_sigma, const float *grad_loss, float *loss,
                                   float *grad_points, void *stream)
{
    if (int rc = check_common("dss_projection_loss", N, P, K, points, mollified, knn_d2, knn_idx, first_idx)) return rc;
    if (P == 0) return DSS_OK;
    if (!num_pts || (!loss && !grad_points) || !(sharpness_sigma > 0.f)) {
        set_error("dss_projection_loss: needs num_pts, loss and/or grad_points, sharpness_sigma > 0");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(projection_loss_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), points,
                       mollified, knn_d2, knn_idx, visible, first_idx, num_pts, N, P, K,
                       1.0f / (sharpness_sigma * sharpness_sigma), grad_loss, loss, grad_points);
    return check_launch("dss_projection_loss");
}

extern "C" int dss_repulsion_loss(const float *points, const float *mollified, const int64_t *knn_idx,
                                  const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int K,
                                  float sharpness_sigma, float filter_scale, const float *grad_loss, float *loss,
                                  float *grad_points, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_common("dss_repulsion_loss", N, P, K, points, mollified, knn_idx, first_idx, num_pts)) return rc;
    if (P == 0) return DSS_OK;
    if ((!loss && !grad_points) || !(sharpness_sigma > 0.f)) {
        set_error("dss_repulsion_loss: needs loss and/or grad_points, sharpness_sigma > 0");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || workspace_bytes < (size_t)N * 6 * sizeof(int)) {
        set_error("dss_repulsion_loss: workspace of %zu bytes needed (24 per cloud)", (size_t)N * 6 * sizeof(int));
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    int *bbox = reinterpret_cast<int *>(workspace);
    if (int rc = launch_cloud_bbox(points, first_idx, num_pts, N, P, bbox, st)) return rc;
    hipLaunchKernelGGL(repulsion_loss_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, points, mollified, knn_idx,
                       bbox, filter_scale, first_idx, num_pts, N, P, K, 1.0f / (sharpness_sigma * sharpness_sigma), grad_loss,
                       loss, grad_points);
    return check_launch("dss_repulsion_loss");
}

extern "C" int dss_points_inmask(const float *points, const float *M, const float *mask, const uint8_t *visible, int N,
                                 int64_t P, int H, int W, uint8_t *inmask, void *stream)
{
    if (N <= 0 || P < 0 || H <= 0 || W <= 0) {
        set_error("dss_points_inmask: bad sizes N=%d P=%lld H=%d W=%d", N, (long long)P, H, W);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!points || !M || !mask || !inmask) {
        set_error("dss_points_inmask: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(points_inmask_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), points, M,
                       mask, visible, N, P, H, W, inmask);
    return check_launch("dss_points_inmask");
}
