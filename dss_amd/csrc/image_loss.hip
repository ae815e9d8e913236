// Image loss of the training iteration and its gradient with respect to the rendered RGBA image:
// Trainer.calc_dr_loss (DSS/training/trainer.py:332-372) with the loss objects of Trainer.__init__ (:138-141):
//     inside = mask != 0 and alpha != 0                       (mask_img.bool() & mask_img_pred.bool(), :351)
//     rgb    = sum_inside sum_c |img_c - pred_c| / #inside    (L1Loss losses.py:127-135 + BaseLoss :54-61; 0 if none)
//     sil    = mean |mask - alpha| + 0.01 mean_n (1 - I_n / eps_denom(U_n))          (:361-367, IouLoss :498-513)
//     total  = lambda_dr_rgb rgb + lambda_dr_silhouette sil
// This sits between the render forward and the render backward of every iteration.  As ~20 elementwise / reduction
// torch kernels (forward + autograd) over (N,H,W,4) images it costs more than the whole 90 us render step at 512^2;
// here: one reduction pass over the images (per-block partial sums in a fixed order -> deterministic), a one-workgroup
// finalize (per-image sums in double, the four loss scalars), and ONE pass that writes the gradient image that
// dss_render_backward consumes.  HBM-bound: 32 B read per pixel per pass, 16 B written by the gradient pass.
#include "common.h"

namespace dss {

#define IMG_LOSS_MAX_BLOCKS 64   // reduction blocks per image
#define IMG_LOSS_TERMS 5         // inside count, sum |rgb diff| inside, sum |mask - alpha|, intersection, union

struct TargetView {  // target colours with arbitrary element strides: (N,H,W,3) or a permuted (N,3,H,W) view
    const float *p;
    int64_t sn, sh, sw, sc;
};

__device__ __forceinline__ float sign_of(float v) { return (float)((v > 0.f) - (v < 0.f)); }

#define IMG_LOSS_REDUCE_THREADS 1024
__global__ __launch_bounds__(IMG_LOSS_REDUCE_THREADS) void image_loss_reduce_kernel(
    const float4 *__restrict__ rgba, const TargetView img, const float *__restrict__ mask, int64_t mask_sn, int H, int W,
    double *__restrict__ part /* (N, blocks, 5) */, int64_t rgba_sn4 = 0, int64_t rgba_sh4 = 0 /* float4 strides; 0: dense */)
{
    constexpr int T = IMG_LOSS_REDUCE_THREADS, WAVES = T / 64;
    __shared__ float wave_part[WAVES][IMG_LOSS_TERMS];
    const int n = blockIdx.y, b = blockIdx.x, nb = gridDim.x;
    const unsigned HW = (unsigned)H * (unsigned)W;   // (pixels of ONE image: below 2^31; 32-bit divides for row / column)
    const bool dense = rgba_sh4 == 0;
    const int64_t base = dense ? (int64_t)n * HW : (int64_t)n * rgba_sn4;
    float cnt = 0.f, srgb = 0.f, smask = 0.f, inter = 0.f, uni = 0.f;  // <= a few hundred terms per thread: exact counts
    // 16 wavefronts per block (at most 64 blocks per image: a quarter of the CUs, so each of them gets a full CU's worth of
    // loads in flight), four pixels per trip with all their loads issued before the first is used; the terms are added in
    // pixel order
    const unsigned stride = (unsigned)nb * T;
    for (unsigned i0 = (unsigned)b * T + threadIdx.x; i0 < HW; i0 += 4u * stride) {
        float4 px[4];
        float t[4], c0[4], c1[4], c2[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned i = i0 + (unsigned)u * stride;
            ok[u] = i < HW && i >= i0;
            const unsigned ic = ok[u] ? i : i0;
            const unsigned y = ic / (unsigned)W, x = ic - y * (unsigned)W;
            px[u] = rgba[dense ? base + ic : base + (int64_t)y * rgba_sh4 + x];
            t[u] = mask[(int64_t)n * mask_sn + ic];  // mask_sn = H*W, or the full image's stride for a row band
            const float *tp = img.p + n * img.sn + (int64_t)y * img.sh + (int64_t)x * img.sw;
            c0[u] = tp[0];
            c1[u] = tp[img.sc];
            c2[u] = tp[2 * img.sc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok[u]) continue;
            const float d = fabsf(c0[u] - px[u].x) + fabsf(c1[u] - px[u].y) + fabsf(c2[u] - px[u].z);
            const bool inside = t[u] != 0.f && px[u].w != 0.f;
            cnt += inside ? 1.f : 0.f;
            srgb += inside ? d : 0.f;
            smask += fabsf(t[u] - px[u].w);
            inter += px[u].w * t[u];
            uni += px[u].w + t[u] - px[u].w * t[u];
        }
    }
    const float v[IMG_LOSS_TERMS] = {wave_sum(cnt), wave_sum(srgb), wave_sum(smask), wave_sum(inter), wave_sum(uni)};
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < IMG_LOSS_TERMS; ++k) wave_part[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < IMG_LOSS_TERMS) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) a += (double)wave_part[w][threadIdx.x];
        part[((size_t)n * nb + b) * IMG_LOSS_TERMS + threadIdx.x] = a;
    }
}

// sums (N+1, 5): rows 0..N-1 per image, row N the totals over the batch.  losses (4): total, weighted rgb term,
// weighted silhouette term, IoU term (mean_n 1 - I/U).  One workgroup: wavefront w reduces the block partials of
// images w, w+4, ... (lane b holds block b, fixed shuffle tree -> deterministic), then one thread adds up the images.
// (The first version walked all N * blocks partials with five threads: 65 us of dependent loads at 8 x 512^2.)
__device__ __forceinline__ double shfl_down_f64(double v, int delta)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_down(lo, delta);
    hi = __shfl_down(hi, delta);
    return __hiloint2double(hi, lo);
}

// mode 0: block partials -> per-image sums -> batch totals + losses (one GPU).  Row bands on several GPUs: mode 1 =
// block partials -> per-image sums of the band only; the caller all-reduces them; mode 2 = (all-reduced) per-image
// sums -> totals + losses.  pix_per_image = pixels of the FULL image (the mean of the silhouette term runs over it).
__global__ __launch_bounds__(256) void image_loss_finalize_kernel(const double *__restrict__ part, int N, int nb,
                                                                  double pix_per_image, float lambda_rgb, float lambda_sil,
                                                                  double *__restrict__ sums, float *__restrict__ losses,
                                                                  int mode)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (mode != 2)
        for (int n = wave; n < N; n += 4) {
            double v[IMG_LOSS_TERMS];
#pragma unroll
            for (int k = 0; k < IMG_LOSS_TERMS; ++k)
                v[k] = lane < nb ? part[((size_t)n * nb + lane) * IMG_LOSS_TERMS + k] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int k = 0; k < IMG_LOSS_TERMS; ++k) v[k] += shfl_down_f64(v[k], o);
            if (lane == 0)
#pragma unroll
                for (int k = 0; k < IMG_LOSS_TERMS; ++k) sums[(size_t)n * IMG_LOSS_TERMS + k] = v[k];
        }
    if (mode == 1) return;
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[IMG_LOSS_TERMS] = {0.0, 0.0, 0.0, 0.0, 0.0}, iou = 0.0;
        for (int n = 0; n < N; ++n) {
            const double *sn = sums + (size_t)n * IMG_LOSS_TERMS;
#pragma unroll
            for (int k = 0; k < IMG_LOSS_TERMS; ++k) tot[k] += sn[k];
            const double u = sn[4];
            const double den = (u < 0 ? -1.0 : 1.0) * fmax(fabs(u), 1e-17);  // eps_denom, mathHelper.py:10-14
            iou += 1.0 - sn[3] / den;
        }
#pragma unroll
        for (int k = 0; k < IMG_LOSS_TERMS; ++k) sums[(size_t)N * IMG_LOSS_TERMS + k] = tot[k];
        iou /= N;
        const double rgb = tot[0] > 0 ? tot[1] / tot[0] : 0.0;  // `if mask_pred.sum() > 0`, trainer.py:352
        const double sil = tot[2] / ((double)N * pix_per_image) + 0.01 * iou;
        losses[0] = (float)(lambda_rgb * rgb + lambda_sil * sil);
        losses[1] = (float)(lambda_rgb * rgb);
        losses[2] = (float)(lambda_sil * sil);
        losses[3] = (float)iou;
    }
}

__global__ __launch_bounds__(256) void image_loss_grad_kernel(const float4 *__restrict__ rgba, const TargetView img,
                                                              const float *__restrict__ mask, int64_t mask_sn, int N, int H,
                                                              int W, double pix_per_image, float lambda_rgb,
                                                              float lambda_sil, const double *__restrict__ sums,
                                                              const float *__restrict__ grad_total,
                                                              float4 *__restrict__ grad_rgba)
{
    // (N H W pixels below 2^32 -- checked on the host: 32-bit divides for image / row / column)
    const unsigned HW = (unsigned)H * (unsigned)W;
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)q >= (uint64_t)N * HW) return;
    const unsigned n = q / HW;
    const unsigned i = q - n * HW;
    const unsigned y = i / (unsigned)W, x = i - y * (unsigned)W;
    const float up = grad_total ? grad_total[0] : 1.0f;
    const double cnt = sums[(size_t)N * IMG_LOSS_TERMS];
    const double I = sums[(size_t)n * IMG_LOSS_TERMS + 3], U = sums[(size_t)n * IMG_LOSS_TERMS + 4];
    const float w_rgb = cnt > 0 ? (float)((double)lambda_rgb / cnt) : 0.f;
    const float w_l1 = (float)((double)lambda_sil / ((double)N * pix_per_image));
    const float4 px = rgba[q];
    const float t = mask[(int64_t)n * mask_sn + i];
    const float *tp = img.p + (int64_t)n * img.sn + (int64_t)y * img.sh + (int64_t)x * img.sw;
    const bool inside = t != 0.f && px.w != 0.f;
    float4 g;
    g.x = inside ? w_rgb * sign_of(px.x - tp[0]) * up : 0.f;
    g.y = inside ? w_rgb * sign_of(px.y - tp[img.sc]) * up : 0.f;
    g.z = inside ? w_rgb * sign_of(px.z - tp[2 * img.sc]) * up : 0.f;
    // d(1 - I/U)/d alpha = -(t U - I (1 - t)) / U^2; with the denominator clamped (U == 0) only -t / eps is left
    const double Ue = (U < 0 ? -1.0 : 1.0) * fmax(fabs(U), 1e-17);
    const double diou = fabs(U) > 1e-17 ? -((double)t * Ue - I * (1.0 - (double)t)) / (Ue * Ue) : -(double)t / Ue;
    g.w = (w_l1 * sign_of(px.w - t) + (float)((double)lambda_sil * 0.01 * diou / N)) * up;
    grad_rgba[q] = g;
}

// Row bands, two launches per step: the block partials themselves are what the ranks all-reduce ((N, 64, 5) doubles: 20 KB
// at 8 cameras, a latency-bound collective either way), and the gradient kernel adds them up in its prologue -- every block
// in the same fixed order, so every rank derives the same bits.  Replaces reduce | finalize | [all-reduce] | finalize |
// gradient (three of whose five launches were ~6 us single-workgroup kernels).
#define IMG_LOSS_BAND_BLOCKS 64
__global__ __launch_bounds__(256) void image_loss_band_grad_kernel(const float4 *__restrict__ rgba, const TargetView img,
                                                                   const float *__restrict__ mask, int64_t mask_sn, int N, int H,
                                                                   int W, double pix_per_image, float lambda_rgb,
                                                                   float lambda_sil, const double *__restrict__ part,
                                                                   const float *__restrict__ grad_total,
                                                                   float4 *__restrict__ grad_rgba, float *__restrict__ losses,
                                                                   double *__restrict__ sums_out, int64_t rgba_sn4,
                                                                   int64_t rgba_sh4, float *__restrict__ alpha_out,
                                                                   int64_t alpha_sn, int64_t alpha_sh)
{
    __shared__ double s_sum[64][IMG_LOSS_TERMS];   // per image (N <= 64)
    // thread (n, k) adds the 64 block partials of term k of image n in block order
    for (int e = threadIdx.x; e < N * IMG_LOSS_TERMS; e += 256) {
        const int n = e / IMG_LOSS_TERMS, k = e - n * IMG_LOSS_TERMS;
        const double *pp = part + (size_t)n * IMG_LOSS_BAND_BLOCKS * IMG_LOSS_TERMS + k;
        double a = 0.0;
#pragma unroll 16
        for (int b = 0; b < IMG_LOSS_BAND_BLOCKS; ++b) a += pp[(size_t)b * IMG_LOSS_TERMS];
        s_sum[n][k] = a;
    }
    __syncthreads();
    double cnt = 0.0;
    for (int n = 0; n < N; ++n) cnt += s_sum[n][0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && (losses || sums_out)) {
        double tot[IMG_LOSS_TERMS] = {0.0, 0.0, 0.0, 0.0, 0.0}, iou = 0.0;
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int k = 0; k < IMG_LOSS_TERMS; ++k) {
                tot[k] += s_sum[n][k];
                if (sums_out) sums_out[(size_t)n * IMG_LOSS_TERMS + k] = s_sum[n][k];
            }
            const double u = s_sum[n][4];
            const double den = (u < 0 ? -1.0 : 1.0) * fmax(fabs(u), 1e-17);  // eps_denom, mathHelper.py:10-14
            iou += 1.0 - s_sum[n][3] / den;
        }
        if (sums_out)
#pragma unroll
            for (int k = 0; k < IMG_LOSS_TERMS; ++k) sums_out[(size_t)N * IMG_LOSS_TERMS + k] = tot[k];
        if (losses) {
            iou /= N;
            const double rgb = tot[0] > 0 ? tot[1] / tot[0] : 0.0;  // `if mask_pred.sum() > 0`, trainer.py:352
            const double sil = tot[2] / ((double)N * pix_per_image) + 0.01 * iou;
            losses[0] = (float)(lambda_rgb * rgb + lambda_sil * sil);
            losses[1] = (float)(lambda_rgb * rgb);
            losses[2] = (float)(lambda_sil * sil);
            losses[3] = (float)iou;
        }
    }
    // (N H W pixels below 2^32 -- checked on the host: 32-bit divides for image / row / column)
    const unsigned HW = (unsigned)H * (unsigned)W;
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
    if ((uint64_t)q >= (uint64_t)N * HW) return;
    const unsigned n = q / HW;
    const unsigned i = q - n * HW;
    const unsigned y = i / (unsigned)W, x = i - y * (unsigned)W;
    const float up = grad_total ? grad_total[0] : 1.0f;
    const double I = s_sum[n][3], U = s_sum[n][4];
    const float w_rgb = cnt > 0 ? (float)((double)lambda_rgb / cnt) : 0.f;
    const float w_l1 = (float)((double)lambda_sil / ((double)N * pix_per_image));
    const float4 px = rgba[rgba_sh4 == 0 ? (int64_t)q : (int64_t)n * rgba_sn4 + (int64_t)y * rgba_sh4 + x];
    const float t = mask[(int64_t)n * mask_sn + i];
    const float *tp = img.p + (int64_t)n * img.sn + (int64_t)y * img.sh + (int64_t)x * img.sw;
    const bool inside = t != 0.f && px.w != 0.f;
    float4 g;
    g.x = inside ? w_rgb * sign_of(px.x - tp[0]) * up : 0.f;
    g.y = inside ? w_rgb * sign_of(px.y - tp[img.sc]) * up : 0.f;
    g.z = inside ? w_rgb * sign_of(px.z - tp[2 * img.sc]) * up : 0.f;
    // (same expressions as image_loss_grad_kernel)
    const double Ue = (U < 0 ? -1.0 : 1.0) * fmax(fabs(U), 1e-17);
    const double diou = fabs(U) > 1e-17 ? -((double)t * Ue - I * (1.0 - (double)t)) / (Ue * Ue) : -(double)t / Ue;
    g.w = (w_l1 * sign_of(px.w - t) + (float)((double)lambda_sil * 0.01 * diou / N)) * up;
    grad_rgba[q] = g;
    // the occupancy gradient once more, where the owner form's exchange sends it from ((row, camera, col) send buffer of
    // dss_amd.distributed.AlphaPlaneExchange): saves the strided copy that would extract it
    if (alpha_out) alpha_out[(int64_t)n * alpha_sn + (int64_t)y * alpha_sh + x] = g.w;
}

static int image_blocks(int H, int W)
{
    const int64_t b = ((int64_t)H * W + 4 * IMG_LOSS_REDUCE_THREADS - 1) / (4 * IMG_LOSS_REDUCE_THREADS);
    return (int)(b < 1 ? 1 : (b > IMG_LOSS_MAX_BLOCKS ? IMG_LOSS_MAX_BLOCKS : b));
}

static int check_image_args(const char *who, const void *rgba, const void *img, const void *mask, int N, int H, int W)
{
    if (N <= 0 || H <= 0 || W <= 0 || (int64_t)N * H * W >= (1ll << 32)) {
        set_error("%s: bad sizes N=%d H=%d W=%d (at most 2^32 pixels)", who, N, H, W);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!rgba || !img || !mask) {
        set_error("%s: NULL tensor pointer", who);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if ((reinterpret_cast<uintptr_t>(rgba) & 15) != 0) {
        set_error("%s: the RGBA image must be 16-byte aligned", who);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    return DSS_OK;
}

}  // namespace dss

using namespace dss;

extern "C" size_t dss_image_loss_workspace(int N, int H, int W)
{
    return (size_t)(N > 0 ? N : 1) * image_blocks(H, W) * IMG_LOSS_TERMS * sizeof(double);
}

extern "C" int dss_image_loss_forward(const float *rgba, const float *target_rgb, int64_t t_stride_n, int64_t t_stride_h,
                                      int64_t t_stride_w, int64_t t_stride_c, const float *target_mask, int N, int H, int W,
                                      float lambda_rgb, float lambda_silhouette, double *sums, float *losses,
                                      void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_image_args("dss_image_loss_forward", rgba, target_rgb, target_mask, N, H, W)) return rc;
    if (!sums || !losses) {
        set_error("dss_image_loss_forward: NULL output pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || workspace_bytes < dss_image_loss_workspace(N, H, W)) {
        set_error("dss_image_loss_forward: workspace too small");
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const int nb = image_blocks(H, W);
    const TargetView tv = {target_rgb, t_stride_n, t_stride_h, t_stride_w, t_stride_c};
    double *part = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(image_loss_reduce_kernel, dim3(nb, N), dim3(IMG_LOSS_REDUCE_THREADS), 0, st, reinterpret_cast<const float4 *>(rgba), tv,
                       target_mask, (int64_t)H * W, H, W, part);
    hipLaunchKernelGGL(image_loss_finalize_kernel, dim3(1), dim3(256), 0, st, part, N, nb, (double)H * (double)W, lambda_rgb,
                       lambda_silhouette, sums, losses, 0);
    return check_launch("dss_image_loss_forward");
}

extern "C" int dss_image_loss_backward(const float *rgba, const float *target_rgb, int64_t t_stride_n, int64_t t_stride_h,
                                       int64_t t_stride_w, int64_t t_stride_c, const float *target_mask, int N, int H,
                                       int W, float lambda_rgb, float lambda_silhouette, const double *sums,
                                       const float *grad_total, float *grad_rgba, void *stream)
{
    if (int rc = check_image_args("dss_image_loss_backward", rgba, target_rgb, target_mask, N, H, W)) return rc;
    if (!sums || !grad_rgba || (reinterpret_cast<uintptr_t>(grad_rgba) & 15) != 0) {
        set_error("dss_image_loss_backward: sums and a 16-byte aligned grad_rgba are required");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const TargetView tv = {target_rgb, t_stride_n, t_stride_h, t_stride_w, t_stride_c};
    const int64_t total = (int64_t)N * H * W;
    hipLaunchKernelGGL(image_loss_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(rgba), tv, target_mask, (int64_t)H * W, N, H, W, (double)H * (double)W,
                       lambda_rgb, lambda_silhouette, sums, grad_total, reinterpret_cast<float4 *>(grad_rgba));
    return check_launch("dss_image_loss_backward");
}

// ---- row bands (multi-GPU): band sums -> [all-reduce by the caller] -> losses from the sums -> band gradient ----------
extern "C" int dss_image_loss_band_sums(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                        int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c, const float *target_mask,
                                        int64_t mask_stride_n, int N, int rows, int W, double *sums, void *workspace,
                                        size_t workspace_bytes, void *stream)
{
    if (int rc = check_image_args("dss_image_loss_band_sums", rgba_band, target_rgb, target_mask, N, rows, W)) return rc;
    if (!sums || !workspace || workspace_bytes < dss_image_loss_workspace(N, rows, W)) {
        set_error("dss_image_loss_band_sums: sums and a workspace of dss_image_loss_workspace(N, rows, W) bytes are required");
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const int nb = image_blocks(rows, W);
    const TargetView tv = {target_rgb, t_stride_n, t_stride_h, t_stride_w, t_stride_c};
    double *part = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(image_loss_reduce_kernel, dim3(nb, N), dim3(IMG_LOSS_REDUCE_THREADS), 0, st, reinterpret_cast<const float4 *>(rgba_band),
                       tv, target_mask, mask_stride_n, rows, W, part);
    hipLaunchKernelGGL(image_loss_finalize_kernel, dim3(1), dim3(256), 0, st, part, N, nb, 0.0, 0.f, 0.f, sums, nullptr, 1);
    return check_launch("dss_image_loss_band_sums");
}

extern "C" int dss_image_loss_from_sums(double *sums, int N, int H, int W, float lambda_rgb, float lambda_silhouette,
                                        float *losses, void *stream)
{
    if (N <= 0 || H <= 0 || W <= 0 || !sums || !losses) {
        set_error("dss_image_loss_from_sums: bad arguments");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(image_loss_finalize_kernel, dim3(1), dim3(256), 0, as_stream(stream), nullptr, N, 0,
                       (double)H * (double)W, lambda_rgb, lambda_silhouette, sums, losses, 2);
    return check_launch("dss_image_loss_from_sums");
}

extern "C" int dss_image_loss_band_backward(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                            int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                            const float *target_mask, int64_t mask_stride_n, int N, int rows, int W, int H,
                                            float lambda_rgb, float lambda_silhouette, const double *sums,
                                            const float *grad_total, float *grad_band, void *stream)
{
    if (int rc = check_image_args("dss_image_loss_band_backward", rgba_band, target_rgb, target_mask, N, rows, W)) return rc;
    if (H < rows || !sums || !grad_band || (reinterpret_cast<uintptr_t>(grad_band) & 15) != 0) {
        set_error("dss_image_loss_band_backward: needs H >= rows, sums and a 16-byte aligned grad_band");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const TargetView tv = {target_rgb, t_stride_n, t_stride_h, t_stride_w, t_stride_c};
    const int64_t total = (int64_t)N * rows * W;
    hipLaunchKernelGGL(image_loss_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(rgba_band), tv, target_mask, mask_stride_n, N, rows, W,
                       (double)H * (double)W, lambda_rgb, lambda_silhouette, sums, grad_total,
                       reinterpret_cast<float4 *>(grad_band));
    return check_launch("dss_image_loss_band_backward");
}

// ---- row bands, fused form: block partials -> [all-reduce of the partials by the caller] -> band gradient + losses -------
extern "C" size_t dss_image_loss_band_partials_count(int N) { return (size_t)(N > 0 ? N : 1) * IMG_LOSS_BAND_BLOCKS * IMG_LOSS_TERMS; }

extern "C" int dss_image_loss_band_partials(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                            int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                            const float *target_mask, int64_t mask_stride_n, int N, int rows, int W,
                                            int64_t rgba_stride_n, int64_t rgba_stride_h, double *partials, void *stream)
{
    if ((rgba_stride_h != 0 && (rgba_stride_h % 4 || rgba_stride_n % 4 || rgba_stride_h < (int64_t)W * 4))) {
        set_error("dss_image_loss_band_partials: rgba strides must be multiples of 4 floats, rows of W pixels contiguous");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (N <= 0 || N > 64 || rows < 0 || W <= 0 || (int64_t)N * rows * W >= (1ll << 32) || !partials || (rows > 0 && (!rgba_band || !target_rgb || !target_mask)) ||
        (reinterpret_cast<uintptr_t>(rgba_band) & 15) != 0) {
        set_error("dss_image_loss_band_partials: bad arguments (1 <= N <= 64, 16-byte aligned band)");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const TargetView tv = {target_rgb, t_stride_n, t_stride_h, t_stride_w, t_stride_c};
    // (an empty band -- a rank without rows -- contributes zeros: the loop of the kernel does not run)
    hipLaunchKernelGGL(image_loss_reduce_kernel, dim3(IMG_LOSS_BAND_BLOCKS, N), dim3(IMG_LOSS_REDUCE_THREADS), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(rgba_band), tv, target_mask, mask_stride_n, rows, W, partials,
                       rgba_stride_n / 4, rgba_stride_h / 4);
    return check_launch("dss_image_loss_band_partials");
}

extern "C" int dss_image_loss_band_backward_partials(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                                     int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                                     const float *target_mask, int64_t mask_stride_n, int N, int rows, int W,
                                                     int H, float lambda_rgb, float lambda_silhouette, const double *partials,
                                                     const float *grad_total, float *grad_band, float *losses, double *sums,
                                                     int64_t rgba_stride_n, int64_t rgba_stride_h, float *alpha_out,
                                                     int64_t alpha_stride_n, int64_t alpha_stride_h, void *stream)
{
    if ((rgba_stride_h != 0 && (rgba_stride_h % 4 || rgba_stride_n % 4 || rgba_stride_h < (int64_t)W * 4))) {
        set_error("dss_image_loss_band_backward_partials: rgba strides must be multiples of 4 floats, rows of W pixels contiguous");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (N <= 0 || N > 64 || rows < 0 || W <= 0 || H < rows || (int64_t)N * rows * W >= (1ll << 32) || !partials || (rows > 0 && (!rgba_band || !target_rgb || !target_mask || !grad_band)) ||
        ((reinterpret_cast<uintptr_t>(grad_band) | reinterpret_cast<uintptr_t>(rgba_band)) & 15) != 0) {
        set_error("dss_image_loss_band_backward_partials: bad arguments (1 <= N <= 64, H >= rows, 16-byte aligned band tensors)");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const TargetView tv = {target_rgb, t_stride_n, t_stride_h, t_stride_w, t_stride_c};
    const int64_t total = (int64_t)N * rows * W;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(image_loss_band_grad_kernel, dim3(blocks > 0 ? blocks : 1u), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4 *>(rgba_band), tv, target_mask, mask_stride_n, N, rows, W,
                       (double)H * (double)W, lambda_rgb, lambda_silhouette, partials, grad_total,
                       reinterpret_cast<float4 *>(grad_band), losses, sums, rgba_stride_n / 4, rgba_stride_h / 4, alpha_out,
                       alpha_stride_n, alpha_stride_h);
    return check_launch("dss_image_loss_band_backward_partials");
}
