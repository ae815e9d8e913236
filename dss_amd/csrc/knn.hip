// K-th nearest-neighbour squared distance on a uniform 3-D grid, gfx950.
//
// The source-space variance scale h of the EWA kernel is a kNN statistic
// (DSS/core/rasterizer.py:310-326 global, :366-388 per point):
//     sq_dist = knn(points, points, K=7)[:, :, 1:] ;  h_k = 0.5 * max(sq_dist)
// i.e. half the 7th-smallest squared distance from each point to the cloud INCLUDING itself.  The
// reference delegates the search to third-party CUDA (FRNN frnn_grid_points / pytorch3d knn_points);
// this is the "next" row of SURVEY 8f.  Exact search: counting-sort the points of every cloud into a
// res^3 grid, then one thread per point visits Chebyshev rings of cells until the K-th distance found
// is provably final (<= distance to the unvisited region).
//
//   knn_bbox      per-cloud bounding box (ordered-int atomics)
//   knn_count     cell of every point, per-cell counts
//   knn_scan      exclusive scan of the cell counts (one workgroup per cloud)
//   knn_fill      counting sort: points grouped by cell
//   knn_query     ring search, K smallest squared distances in registers
//   cloud_mean    deterministic per-cloud mean * scale, clamped (the global-h statistic)
#include "common.h"

namespace dss {

#define KNN_MAX_K 16
#define KNN_MAX_RES 64
#define KNN_STRIDE ((size_t)KNN_MAX_RES * KNN_MAX_RES * KNN_MAX_RES + 1)  // cells per cloud + end sentinel

__device__ __forceinline__ int f2ord(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

struct KnnGrid {  // per cloud, device resident (8 floats)
    float minx, miny, minz, inv_cell, cell;
    int res;
    int pad0, pad1;
};

// Bounding box per cloud.  Grid (G, N): workgroups of cloud n stride over its points, reduce min/max in
// registers -> wave (shuffles) -> one atomic set per WAVE.  (One atomic set per POINT, the first version,
// serialised P same-address atomics: 2.2 ms at 80k points.)
__global__ __launch_bounds__(256) void knn_bbox_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                       const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                       int *__restrict__ bbox /* (N,6) ordered ints */)
{
    const int n = blockIdx.y;
    const int64_t f = first_idx[n], cnt = num_pts[n];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = f + i;
        if (p >= P) break;
        const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
        if (!(x == x && y == y && z == z)) continue;
        const int ox = f2ord(x), oy = f2ord(y), oz = f2ord(z);
        lo[0] = min(lo[0], ox); lo[1] = min(lo[1], oy); lo[2] = min(lo[2], oz);
        hi[0] = max(hi[0], ox); hi[1] = max(hi[1], oy); hi[2] = max(hi[2], oz);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = min(lo[d], __shfl_xor(lo[d], o));
            hi[d] = max(hi[d], __shfl_xor(hi[d], o));
        }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (lo[d] != 0x7fffffff) atomicMin(&bbox[6 * n + d], lo[d]);
            if (hi[d] != (int)0x80000000) atomicMax(&bbox[6 * n + 3 + d], hi[d]);
        }
    }
}

__global__ void knn_init_kernel(int N, int *__restrict__ bbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * N) bbox[i] = (i % 6 < 3) ? 0x7fffffff : (int)0x80000000;  // +inf / -inf in ordered-int space
}

__global__ void knn_grid_kernel(const int *__restrict__ bbox, const int64_t *__restrict__ num_pts, int N,
                                KnnGrid *__restrict__ grids)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    KnnGrid g;
    const float x0 = ord2f(bbox[6 * n]), y0 = ord2f(bbox[6 * n + 1]), z0 = ord2f(bbox[6 * n + 2]);
    const float x1 = ord2f(bbox[6 * n + 3]), y1 = ord2f(bbox[6 * n + 4]), z1 = ord2f(bbox[6 * n + 5]);
    const float ext = fmaxf(fmaxf(x1 - x0, y1 - y0), fmaxf(z1 - z0, 1e-12f));
    // surface-like clouds occupy ~3 res^2 cells: aim at ~8 points per occupied cell
    int res = (int)ceilf(sqrtf((float)num_pts[n] / 24.0f));
    res = max(1, min(KNN_MAX_RES, res));
    g.minx = x0; g.miny = y0; g.minz = z0;
    g.cell = ext / (float)res * 1.0001f;
    g.inv_cell = 1.0f / g.cell;
    g.res = res;
    g.pad0 = g.pad1 = 0;
    grids[n] = g;
}

__device__ __forceinline__ int cell_coord(float v, float mn, float inv_cell, int res)
{
    const int c = (int)floorf((v - mn) * inv_cell);
    return min(max(c, 0), res - 1);
}

__global__ __launch_bounds__(256) void knn_count_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                        const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                        const KnnGrid *__restrict__ grids, uint32_t *__restrict__ counts,
                                                        int32_t *__restrict__ cell_of)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    if (n < 0) { cell_of[p] = -1; return; }
    const KnnGrid g = grids[n];
    const int cx = cell_coord(pts[3 * p], g.minx, g.inv_cell, g.res);
    const int cy = cell_coord(pts[3 * p + 1], g.miny, g.inv_cell, g.res);
    const int cz = cell_coord(pts[3 * p + 2], g.minz, g.inv_cell, g.res);
    const int c = (cz * g.res + cy) * g.res + cx;
    cell_of[p] = c;
    atomicAdd(&counts[(size_t)n * KNN_STRIDE + c], 1u);
}

// one workgroup per cloud; scans res^3 cells
__global__ __launch_bounds__(1024) void knn_scan_kernel(const uint32_t *__restrict__ counts,
                                                        const KnnGrid *__restrict__ grids,
                                                        uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const size_t base_n = (size_t)n * KNN_STRIDE;
    const int res = grids[n].res;
    const int cells = res * res * res;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < cells; base += 1024) {
        const int i = base + tid;
        const uint32_t v = (i < cells) ? counts[base_n + i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wave_tot[wid] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wid; ++w) woff += wave_tot[w];
        const uint32_t excl = carry_s + woff + x - v;
        if (i < cells) {
            offsets[base_n + i] = excl;
            cursor[base_n + i] = excl;
        }
        __syncthreads();
        if (tid == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (tid == 0) offsets[base_n + cells] = carry_s;
}

__global__ __launch_bounds__(256) void knn_fill_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                       const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                       const int32_t *__restrict__ cell_of, uint32_t *__restrict__ cursor,
                                                       float4 *__restrict__ sorted /* (P) xyz + id, grouped by cell */)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int c = cell_of[p];
    if (c < 0) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    const uint32_t pos = atomicAdd(&cursor[(size_t)n * KNN_STRIDE + c], 1u);
    sorted[first_idx[n] + pos] = make_float4(pts[3 * p], pts[3 * p + 1], pts[3 * p + 2], __int_as_float((int)p));
}

// FULL = false: K-th squared distance only (kth_sqdist (P,)).  FULL = true: the whole neighbour list, ascending in
// (distance, id): dists (P,Krt) squared distances and idx (P,Krt) cloud-local ids, zero-padded when the cloud has
// fewer than Krt points (the layout pytorch3d.ops.knn_points returns for a self query, losses.py:157-180).
template <int K, bool FULL>
__global__ __launch_bounds__(256) void knn_query_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                        const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                        const KnnGrid *__restrict__ grids, const uint32_t *__restrict__ offsets,
                                                        const float4 *__restrict__ sorted, int Krt,
                                                        float *__restrict__ kth_sqdist, float *__restrict__ dists,
                                                        int64_t *__restrict__ idx)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    if (n < 0) {
        if (FULL) {
            for (int k = 0; k < Krt; ++k) { dists[p * Krt + k] = 0.0f; idx[p * Krt + k] = 0; }
        } else {
            kth_sqdist[p] = 0.0f;
        }
        return;
    }
    const KnnGrid g = grids[n];
    const int64_t f0 = first_idx[n];
    const int64_t cnt_n = num_pts[n];
    const uint32_t *off = offsets + (size_t)n * KNN_STRIDE;
    const float qx = pts[3 * p], qy = pts[3 * p + 1], qz = pts[3 * p + 2];
    const int cx = cell_coord(qx, g.minx, g.inv_cell, g.res);
    const int cy = cell_coord(qy, g.miny, g.inv_cell, g.res);
    const int cz = cell_coord(qz, g.minz, g.inv_cell, g.res);
    float best[K];
    int bid[FULL ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) best[k] = __builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < (FULL ? K : 1); ++k) bid[k] = 0x7fffffff;
    const int kk = (int)min((int64_t)Krt, cnt_n);  // fewer points than K: k-th = farthest available
    if (kk <= 0) {
        if (!FULL) kth_sqdist[p] = 0.0f;
        return;
    }
    for (int ring = 0; ring < g.res; ++ring) {
        const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.res - 1);
        const int y0 = max(cy - ring, 0), y1 = min(cy + ring, g.res - 1);
        const int z0 = max(cz - ring, 0), z1 = min(cz + ring, g.res - 1);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool shell_zy = (z == cz - ring) || (z == cz + ring) || (y == cy - ring) || (y == cy + ring);
                for (int x = x0; x <= x1; ++x) {
                    // only the new shell of this ring
                    if (!shell_zy && x != cx - ring && x != cx + ring) continue;
                    const int c = (z * g.res + y) * g.res + x;
                    const uint32_t s = off[c], e = off[c + 1];
                    for (uint32_t j = s; j < e; ++j) {
                        const float4 q = sorted[f0 + j];
                        const float dx = q.x - qx, dy = q.y - qy, dz = q.z - qz;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        if (FULL) {
                            // total order (distance, id): deterministic lists whatever the cell order
                            const int id = __float_as_int(q.w);
                            if (d2 < best[K - 1] || (d2 == best[K - 1] && id < bid[K - 1])) {
                                bool lt[K];
#pragma unroll
                                for (int k = 0; k < K; ++k) lt[k] = d2 < best[k] || (d2 == best[k] && id < bid[k]);
#pragma unroll
                                for (int k = K - 1; k >= 1; --k) {
                                    best[k] = lt[k - 1] ? best[k - 1] : (lt[k] ? d2 : best[k]);
                                    bid[k] = lt[k - 1] ? bid[k - 1] : (lt[k] ? id : bid[k]);
                                }
                                best[0] = lt[0] ? d2 : best[0];
                                bid[0] = lt[0] ? id : bid[0];
                            }
                        } else if (d2 < best[K - 1]) {
#pragma unroll
                            for (int k = K - 1; k >= 1; --k) {
                                const bool sh = d2 < best[k - 1];
                                best[k] = sh ? best[k - 1] : (d2 < best[k] ? d2 : best[k]);
                            }
                            best[0] = d2 < best[0] ? d2 : best[0];
                        }
                    }
                }
            }
        // distance from the query to the boundary of the visited block (exact lower bound for unvisited points);
        // faces that coincide with the grid boundary have nothing behind them
        float bound = __builtin_huge_valf();
        if (cx - ring > 0) bound = fminf(bound, qx - (g.minx + (float)(cx - ring) * g.cell));
        if (cx + ring < g.res - 1) bound = fminf(bound, (g.minx + (float)(cx + ring + 1) * g.cell) - qx);
        if (cy - ring > 0) bound = fminf(bound, qy - (g.miny + (float)(cy - ring) * g.cell));
        if (cy + ring < g.res - 1) bound = fminf(bound, (g.miny + (float)(cy + ring + 1) * g.cell) - qy);
        if (cz - ring > 0) bound = fminf(bound, qz - (g.minz + (float)(cz - ring) * g.cell));
        if (cz + ring < g.res - 1) bound = fminf(bound, (g.minz + (float)(cz + ring + 1) * g.cell) - qz);
        // slack for the fp32 cell-boundary arithmetic
        bound = bound - 1e-6f * fmaxf(fabsf(bound), g.cell);
        float kth = best[0];
#pragma unroll
        for (int k = 1; k < K; ++k) kth = (k < kk) ? best[k] : kth;
        if (bound == __builtin_huge_valf() || (bound > 0.0f && kth <= bound * bound)) break;
    }
    if (FULL) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (k < Krt) {
                dists[p * Krt + k] = k < kk ? best[k] : 0.0f;
                idx[p * Krt + k] = k < kk ? (int64_t)bid[k] - f0 : 0;
            }
        return;
    }
    float kth = best[0];
#pragma unroll
    for (int k = 1; k < K; ++k) kth = (k < kk) ? best[k] : kth;
    kth_sqdist[p] = kth;
}

// deterministic per-cloud mean of values*scale clamped to [lo,hi]: one workgroup per cloud, fixed order
__global__ __launch_bounds__(1024) void cloud_mean_kernel(const float *__restrict__ vals, const int64_t *__restrict__ first_idx,
                                                          const int64_t *__restrict__ num_pts, float scale, float lo,
                                                          float hi, float fallback, int min_points,
                                                          float *__restrict__ out)
{
    __shared__ double part[1024];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int64_t f0 = first_idx[n], cnt = num_pts[n];
    double acc = 0.0;
    for (int64_t i = tid; i < cnt; i += 1024) acc += (double)(vals[f0 + i] * scale);
    part[tid] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) part[tid] += part[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        float m = (cnt >= min_points && cnt > 0) ? (float)(part[0] / (double)cnt) : fallback;
        out[n] = fminf(fmaxf(m, lo), hi);
    }
}

}  // namespace dss

using namespace dss;

static size_t knn_cells(int N) { return (size_t)(N > 0 ? N : 1) * KNN_STRIDE; }

extern "C" size_t dss_knn_workspace(int N, int64_t P)
{
    const size_t n = N > 0 ? N : 1, p = P > 0 ? P : 1;
    return align_up(n * 6 * 4, 256) + align_up(n * sizeof(KnnGrid), 256) + align_up((knn_cells(N) + 1) * 4, 256) * 3 +
           align_up(p * 4, 256) + align_up(p * 16, 256);
}

#define KNN_FULL_MAX_K 40

// grid build + query; exactly one of (kth_sqdist) / (dists, idx) is written
static int knn_run(const char *who, const float *points, const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P,
                   int K, float *kth_sqdist, float *dists, int64_t *idx, void *workspace, size_t workspace_bytes,
                   void *stream)
{
    const bool full = dists != nullptr;
    if (N <= 0 || P < 0 || K < 1 || K > (full ? KNN_FULL_MAX_K : KNN_MAX_K)) {
        set_error("%s: bad sizes N=%d P=%lld K=%d (K <= %d)", who, N, (long long)P, K, full ? KNN_FULL_MAX_K : KNN_MAX_K);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (P == 0) return DSS_OK;
    if (!points || !first_idx || !num_pts || (!full && !kth_sqdist) || (full && !idx)) {
        set_error("%s: NULL tensor pointer", who);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || workspace_bytes < dss_knn_workspace(N, P)) {
        set_error("%s: workspace too small", who);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    char *w = reinterpret_cast<char *>(workspace);
    size_t off = 0;
    int *bbox = reinterpret_cast<int *>(w + off);                 off += align_up((size_t)N * 6 * 4, 256);
    KnnGrid *grids = reinterpret_cast<KnnGrid *>(w + off);        off += align_up((size_t)N * sizeof(KnnGrid), 256);
    const size_t cbytes = align_up((knn_cells(N) + 1) * 4, 256);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + off);     off += cbytes;
    uint32_t *offsets = reinterpret_cast<uint32_t *>(w + off);    off += cbytes;
    uint32_t *cursor = reinterpret_cast<uint32_t *>(w + off);     off += cbytes;
    int32_t *cell_of = reinterpret_cast<int32_t *>(w + off);      off += align_up((size_t)P * 4, 256);
    float4 *sorted = reinterpret_cast<float4 *>(w + off);
    hipLaunchKernelGGL(knn_init_kernel, dim3((6 * N + 63) / 64), dim3(64), 0, st, N, bbox);
    if (hipMemsetAsync(counts, 0, cbytes, st) != hipSuccess) return check_launch("knn memset");
    const unsigned pb = (unsigned)((P + 255) / 256);
    const unsigned bb = (unsigned)((P / N + 2047) / 2048 > 64 ? 64 : (P / N + 2047) / 2048);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(bb ? bb : 1, N), dim3(256), 0, st, points, first_idx, num_pts, N, P, bbox);
    hipLaunchKernelGGL(knn_grid_kernel, dim3((N + 63) / 64), dim3(64), 0, st, bbox, num_pts, N, grids);
    hipLaunchKernelGGL(knn_count_kernel, dim3(pb), dim3(256), 0, st, points, first_idx, num_pts, N, P, grids, counts,
                       cell_of);
    hipLaunchKernelGGL(knn_scan_kernel, dim3(N), dim3(1024), 0, st, counts, grids, offsets, cursor);
    hipLaunchKernelGGL(knn_fill_kernel, dim3(pb), dim3(256), 0, st, points, first_idx, num_pts, N, P, cell_of, cursor,
                       sorted);
#define KNN_LAUNCH(KK, FF)                                                                                          \
    hipLaunchKernelGGL((knn_query_kernel<KK, FF>), dim3(pb), dim3(256), 0, st, points, first_idx, num_pts, N, P, grids,  \
                       offsets, sorted, K, kth_sqdist, dists, idx)
    if (full) {
        if (K <= 8) KNN_LAUNCH(8, true);
        else if (K <= 16) KNN_LAUNCH(16, true);
        else KNN_LAUNCH(KNN_FULL_MAX_K, true);
    } else {
        if (K <= 8) KNN_LAUNCH(8, false);
        else KNN_LAUNCH(KNN_MAX_K, false);
    }
#undef KNN_LAUNCH
    return check_launch(who);
}

extern "C" int dss_knn_kth_sqdist(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N,
                                  int64_t P, int K, float *kth_sqdist, void *workspace, size_t workspace_bytes,
                                  void *stream)
{
    return knn_run("dss_knn_kth_sqdist", points, first_idx, num_pts, N, P, K, kth_sqdist, nullptr, nullptr, workspace,
                   workspace_bytes, stream);
}

extern "C" int dss_knn_points(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P,
                              int K, float *dists, int64_t *idx, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!dists) {
        set_error("dss_knn_points: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    return knn_run("dss_knn_points", points, first_idx, num_pts, N, P, K, nullptr, dists, idx, workspace, workspace_bytes,
                   stream);
}

extern "C" int dss_cloud_mean_clamp(const float *values, const int64_t *first_idx, const int64_t *num_pts, int N,
                                    float scale, float lo, float hi, float fallback, int min_points, float *out,
                                    void *stream)
{
    if (N <= 0 || !values || !first_idx || !num_pts || !out) {
        set_error("dss_cloud_mean_clamp: bad arguments");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    hipLaunchKernelGGL(cloud_mean_kernel, dim3(N), dim3(1024), 0, as_stream(stream), values, first_idx, num_pts, scale,
                       lo, hi, fallback, min_points, out);
    return check_launch("dss_cloud_mean_clamp");
}
