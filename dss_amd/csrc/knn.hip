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
//   knn_scan      exclusive scan of the cell counts: 1024-cell blocks scanned locally, then offset by the block prefix
//   knn_fill      counting sort: points grouped by cell
//   knn_query     ring search, K smallest squared distances in registers; one thread per point IN CELL ORDER, so the
//                 lanes of a wavefront walk the same few cells (same trip counts, broadcast loads)
//   cloud_mean    deterministic per-cloud mean * scale, clamped (the global-h statistic)
#include "common.h"

namespace dss {

#define KNN_MAX_K 16
#define KNN_MAX_RES 128
#define KNN_SCAN_BLOCK 1024  // cells per workgroup of the two-level scan

struct KnnGrid {  // per cloud, device resident (8 floats)
    float minx, miny, minz, inv_cell, cell;
    int res;
    int pad0, pad1;
};

// Bounding box per cloud.  Grid (G, N): workgroups of cloud n stride over its points, reduce min/max in
// registers -> wave (shuffles) -> workgroup (LDS) -> one atomic per bound and WORKGROUP.  (One atomic set per POINT,
// the first version, serialised P same-address atomics: 2.2 ms at 80k points.)
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
    __shared__ int part[4][6];
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { part[wid][d] = lo[d]; part[wid][3 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {  // one atomic per workgroup and bound
        const int d = threadIdx.x;
        int v = part[0][d];
#pragma unroll
        for (int w = 1; w < 4; ++w) v = d < 3 ? min(v, part[w][d]) : max(v, part[w][d]);
        if (d < 3 && v != 0x7fffffff) atomicMin(&bbox[6 * n + d], v);
        if (d >= 3 && v != (int)0x80000000) atomicMax(&bbox[6 * n + d], v);
    }
}

__global__ void knn_init_kernel(int N, int *__restrict__ bbox)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * N) bbox[i] = (i % 6 < 3) ? 0x7fffffff : (int)0x80000000;  // +inf / -inf in ordered-int space
}

__global__ void knn_grid_kernel(const int *__restrict__ bbox, const int64_t *__restrict__ num_pts, int N, int res_cap,
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
    res = max(1, min(res_cap, res));  // res_cap: the resolution the workspace was sized for (knn_res_cap)
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
                                                        const KnnGrid *__restrict__ grids, size_t stride,
                                                        uint32_t *__restrict__ counts, int32_t *__restrict__ cell_of,
                                                        uint32_t *__restrict__ rank_of /* the point's arrival number in its cell */)
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
    // the counting atomic's return value is the point's slot inside the cell: the fill pass needs no second atomic (on a
    // clustered cloud both passes are bound by the fullest cell: 3,900 same-address atomics, 86 us each way)
    rank_of[p] = atomicAdd(&counts[(size_t)n * stride + c], 1u);
}

// Two-level exclusive scan of the cell counts.  Grid (blocks, N); block b of cloud n owns cells [1024 b, 1024 b + 1024).
// Pass 1: local exclusive scan -> offsets, block total -> blk_tot.  Pass 2: every block sums the totals of the blocks
// before it (at most 2048 values) and adds that prefix; the block holding the last cell also writes the end sentinel.
// (The first version scanned all res^3 cells with ONE workgroup per cloud: 288 us at res = 64.)
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *wave_tot /* LDS [4] */, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t t = wave_tot[w];
        before += (w < wid) ? t : 0u;
        total += t;
    }
    return before + incl - v;
}

__global__ __launch_bounds__(256) void knn_scan_local_kernel(const uint32_t *__restrict__ counts,
                                                             const KnnGrid *__restrict__ grids, size_t stride, int nblk_max,
                                                             uint32_t *__restrict__ offsets, uint32_t *__restrict__ blk_tot)
{
    __shared__ uint32_t wave_tot[4];
    const int n = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const int res = grids[n].res;
    const int cells = res * res * res;
    if (b * KNN_SCAN_BLOCK >= cells) return;
    const size_t base = (size_t)n * stride;
    const int c0 = b * KNN_SCAN_BLOCK + 4 * tid;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (c0 + i < cells) ? counts[base + c0 + i] : 0u;
    uint32_t total;
    uint32_t run = block_excl_scan_256(v[0] + v[1] + v[2] + v[3], wave_tot, total);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (c0 + i < cells) offsets[base + c0 + i] = run;
        run += v[i];
    }
    if (tid == 0) blk_tot[(size_t)n * nblk_max + b] = total;
}

__global__ __launch_bounds__(256) void knn_scan_add_kernel(const KnnGrid *__restrict__ grids, size_t stride, int nblk_max,
                                                           const uint32_t *__restrict__ blk_tot, uint32_t *__restrict__ offsets,
                                                           uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t wave_tot[4];
    const int n = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const int res = grids[n].res;
    const int cells = res * res * res;
    if (b * KNN_SCAN_BLOCK >= cells) return;
    const size_t base = (size_t)n * stride;
    uint32_t part = 0;
    for (int i = tid; i < b; i += 256) part += blk_tot[(size_t)n * nblk_max + i];
    uint32_t prefix;
    block_excl_scan_256(part, wave_tot, prefix);  // only the total is needed: blocks before b
    const int c0 = b * KNN_SCAN_BLOCK + 4 * tid;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (c0 + i < cells) {
            const uint32_t o = offsets[base + c0 + i] + prefix;
            offsets[base + c0 + i] = o;
            cursor[base + c0 + i] = o;
        }
    if (tid == 0 && (b + 1) * KNN_SCAN_BLOCK >= cells) offsets[base + cells] = prefix + blk_tot[(size_t)n * nblk_max + b];
}

// Per-camera depth culling inside the search (dss_knn_kth_sqdist_view): the reference runs its neighbour search AFTER
// filter_renderable has dropped, per camera, the points outside [znear, zfar] (rasterizer.py:599, 183-217, 310-326) -- a point's
// neighbours are then the ones the SAME camera keeps.  mode 1: one cloud shared by the cameras, blockIdx.y = camera, results at
// kth_sqdist[camera * P + p]; mode 2: cloud n belongs to camera n.  A query point its camera drops gets 0 (never read).
struct KnnView {
    const float *V;        // (cameras, 4, 4) world -> view, row-vector convention
    const float *znear, *zfar;
    int mode;              // 0: off
    uint32_t *culls;       // (cameras) != 0: the camera drops at least one point of its cloud (written by knn_fill_kernel)
    int n_cams;
};
// The unmasked search of the whole cloud runs FIRST (one launch: statistic and K-th distance of every point); a camera's
// masked search is only needed for a query that has a dropped point among the neighbours that count, and a dropped point
// lies on the far side of one of the camera's two depth planes: the distance from a KEPT query to it is at least the
// query's distance to that plane.  So whenever both planes are farther from the query than rho -- the K-th distance of the
// unmasked search, or the search radius r if that is smaller (neighbours beyond r do not count) -- the camera's result is
// the unmasked one and is copied.  Likewise for a camera that drops nothing at all.  (With the near plane where the
// reference's data sets put it, znear = 0.1, no camera drops anything: one search, N copies.  With znear = 1.0 cutting
// through the cloud, only the queries within their neighbourhood's reach of the cut search again: 8 cameras x 99,790
// points of the trained cloud of tools/clustered_timing.py took 1.26 ms as eight masked searches.)
// Exactness: every dropped point is farther than rho; with rho = the K-th distance the K nearest are all kept, so the
// masked list is the unmasked list; with rho = r < the K-th distance, the entries within r are the same in both lists and
// the others do not count.  The margin allows for the rounding of the view depth (1e-5 (1 + |z|)) and for a view matrix
// whose depth axis is not a unit vector.  plain_dk = inf (fewer than K points in the cloud) never takes the shortcut.
__device__ __forceinline__ bool knn_view_shortcut(uint32_t camera_drops, float x, float y, float z, float v2, float v6, float v10,
                                                  float v14, float zn, float zf, float plain_dk, float r2)
{
    if (camera_drops == 0u) return true;
    const float zview = x * v2 + y * v6 + z * v10 + 1.0f * v14;
    const float margin = fminf(zview - zn, zf - zview) - 1e-5f * (1.0f + fabsf(zview));
    const float rho2 = r2 > 0.0f ? fminf(plain_dk, r2) : plain_dk;
    return margin > 0.0f && margin * margin > rho2 * (v2 * v2 + v6 * v6 + v10 * v10) * 1.0002f;
}
// mode 1 of dss_knn_kth_sqdist_view (one cloud, several cameras): the row of a camera that drops nothing is the unmasked search
// itself -- a coalesced copy; the query launches skip such cameras
__global__ __launch_bounds__(256) void knn_view_rows_kernel(float *__restrict__ kth, const float *__restrict__ plain, int64_t P,
                                                            const uint32_t *__restrict__ culls)
{
    const int cam = blockIdx.y;
    if (culls[cam] != 0u) return;
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    float *row = kth + (size_t)cam * (size_t)P;
    if (i + 3 < P && ((((uintptr_t)row) | ((uintptr_t)plain)) & 15u) == 0) {
        reinterpret_cast<float4 *>(row)[i >> 2] = reinterpret_cast<const float4 *>(plain)[i >> 2];
    } else {
        for (int64_t j = i; j < min(i + 4, P); ++j) row[j] = plain[j];
    }
}
__device__ __forceinline__ bool knn_kept(float x, float y, float z, float v2, float v6, float v10, float v14, float zn, float zf)
{
    const float zview = x * v2 + y * v6 + z * v10 + 1.0f * v14;   // the expression of setup_point_compute
    return (zview >= zn) && (zview <= zf);
}

__global__ __launch_bounds__(256) void knn_fill_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                       const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                       const int32_t *__restrict__ cell_of, size_t stride,
                                                       uint32_t *__restrict__ cursor,
                                                       float4 *__restrict__ sorted /* (P) xyz + id, grouped by cell */,
                                                       const KnnView view = KnnView(),
                                                       const uint32_t *__restrict__ rank_of = nullptr /* given: `cursor` = the cell offsets, read only */)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int c = cell_of[p];
    if (c < 0) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    const float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
    const uint32_t pos = rank_of ? cursor[(size_t)n * stride + c] + rank_of[p] : atomicAdd(&cursor[(size_t)n * stride + c], 1u);
    sorted[first_idx[n] + pos] = make_float4(x, y, z, __int_as_float((int)p));
    if (view.mode != 0) {
        // which cameras drop a point of their cloud (dss_knn_kth_sqdist_view): one atomic per wavefront and camera that does
        const int c0 = view.mode == 1 ? 0 : n, c1 = view.mode == 1 ? view.n_cams : n + 1;
        for (int cam = c0; cam < c1; ++cam) {
            const float *vm = view.V + 16 * cam;
            const bool drop = !knn_kept(x, y, z, vm[2], vm[6], vm[10], vm[14], view.znear[cam], view.zfar[cam]);
            const unsigned long long m = __ballot(drop);
            // (one atomic per wavefront, and none once the flag is seen up: 12,000 atomics on eight words took 100 us at 8 x 100k)
            if (drop && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m) &&
                __hip_atomic_load(&view.culls[cam], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                atomicOr(&view.culls[cam], 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Clustered clouds.  The grid is uniform, sized for ~8 points per occupied cell of an evenly sampled surface.  The model of
// the reference's training loop does not stay that way: after ~900 iterations at BASELINE configs[2] a few outliers have
// stretched the bounding box to +-2 while nine tenths of the points sit within 0.55 of the origin -- a typical point shares
// its cell with 650 others (3,900 in the fullest cell), every query scans ~4,000 candidates instead of ~200, and the two
// searches of an iteration go from 0.13 / 0.16 ms to 1.7 / 2.3 ms (profiles/r6_b_train_mvr_ref_kernel_stats.csv, before this).  No
// single cell size serves a cloud whose neighbour distances span two orders of magnitude.  So, from KNN_SKIP_MIN_P points
// on, the cell-sorted array gets a skip structure:
//   * knn_dense_list_kernel lists the DENSE cells (more than KNN_DENSE_CELL points) and counts the points they hold;
//     knn_subsort_kernel orders the points of each along a Morton curve of their position inside the cell, so that
//     consecutive slots are close in space, and raises `dense_flag` when those cells hold a real share of the cloud;
//   * knn_block_box_kernel records the bounding box of every KNN_BLOCK consecutive slots;
//   * a query (both kernels) first looks at its own sorted neighbourhood (own cell dense: the KNN_SEED slots around the
//     query's slot) -- which gives a K-th distance close to the final one at once -- and then walks long candidate runs
//     block by block, skipping every block whose box lies farther than the current K-th distance.  The cooperative kernel
//     tests 32 boxes per trip (two per lane) and reads up to four surviving blocks together; clouds with `dense_flag` up go
//     to it at every size (a second launch next to the one-thread kernel's where that one is the choice for even clouds:
//     each leaves at once when the cloud is not its kind).
// Same candidates rule, same (distance, id) order: results are identical; only points that cannot enter the list are skipped.
// On the trained cloud: 3,999 -> 200 candidates + 250 box tests per query.  A cloud without dense cells pays four launches
// that find nothing to do (~13 us at 100k-200k points; `dense_flag` stays 0, the boxes are neither built nor read and the
// queries run as before).
// ---------------------------------------------------------------------------------------------------------------
#define KNN_SKIP_MIN_P 65536     // per call: below, the grid build is a chain of launch latencies and clouds are small
#define KNN_DENSE 64             // slots in a candidate run (and points in the query's own cell, for the seed) from which the blocks are walked
#define KNN_DENSE_CELL 64        // points in a cell from which it is listed and sub-sorted
#define KNN_DENSE_SHARE 8        // `dense_flag` goes up when at least 1 / KNN_DENSE_SHARE of the points live in such cells (a 1M-point
                                 // evenly sampled cloud at the resolution cap has a few of them: the cooperative walk it would
                                 // select is 2.5x slower there than the one-thread kernel, 2.21 against 0.88 ms for K = 12 lists)
#define KNN_BLOCK 16             // slots per box
#define KNN_SEED 16              // slots on either side of the query's own slot looked at first
#define KNN_SUBSORT_MAX 4096     // largest cell that is sub-sorted (larger ones stay in arrival order: correct, loose boxes)
#define KNN_SUBSORT_WGS 1024     // workgroups of the (persistent) sub-sort launch
#define KNN_SUBSORT_THREADS 1024  // (the launch lasts as long as its largest cell: ~80 bitonic stages of 4096 keys)

__device__ __forceinline__ uint32_t knn_spread5(uint32_t v)   // bit i of v (i < 5) -> bit 3 i
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}

// Dense cells of all clouds -> list[] (cloud, cell), *n_list, and the number of points they hold.
// One thread per cell, one atomic per wavefront that found any.
__global__ __launch_bounds__(256) void knn_dense_list_kernel(const KnnGrid *__restrict__ grids, size_t stride,
                                                             const uint32_t *__restrict__ offsets, uint2 *__restrict__ list,
                                                             uint32_t *__restrict__ n_list, uint32_t *__restrict__ n_dense_pts)
{
    const int n = blockIdx.y;
    const int res = grids[n].res, cells = res * res * res;
    const int c = blockIdx.x * 256 + threadIdx.x;
    const uint32_t *off = offsets + (size_t)n * stride;
    const uint32_t cnt = c < cells ? off[c + 1] - off[c] : 0u;
    const bool dense = cnt > (uint32_t)KNN_DENSE_CELL;
    const unsigned long long m = __ballot(dense);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(m);
    uint32_t pts_here = dense ? cnt : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pts_here += (uint32_t)__shfl_xor((int)pts_here, o);
    uint32_t base = 0;
    if (lane == leader) {
        base = atomicAdd(n_list, (uint32_t)__popcll(m));
        atomicAdd(n_dense_pts, pts_here);
    }
    base = (uint32_t)__shfl((int)base, leader);
    if (dense) list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)n, (uint32_t)c);
}

// Persistent grid over the list: one workgroup sorts one dense cell at a time in LDS -- keys = (15-bit Morton code of the
// position inside the cell) << 12 | local slot, bitonic network over the next power of two; the points travel through
// registers (read in sorted order, barrier, written back in place).
__global__ __launch_bounds__(KNN_SUBSORT_THREADS) void knn_subsort_kernel(const KnnGrid *__restrict__ grids, size_t stride,
                                                          const uint32_t *__restrict__ offsets,
                                                          const int64_t *__restrict__ first_idx, float4 *__restrict__ sorted,
                                                          const uint2 *__restrict__ list, const uint32_t *__restrict__ n_list,
                                                          const uint32_t *__restrict__ n_dense_pts, int64_t P,
                                                          uint32_t *__restrict__ dense_flag)
{
    __shared__ uint32_t s_key[KNN_SUBSORT_MAX];
    const int tid = threadIdx.x;
    const uint32_t total = *n_list;
    // the later launches walk the blocks only if the dense cells hold a real share of the points (KNN_DENSE_SHARE)
    if (blockIdx.x == 0 && tid == 0) *dense_flag = (unsigned long long)*n_dense_pts * KNN_DENSE_SHARE >= (unsigned long long)P ? 1u : 0u;
    for (uint32_t it = blockIdx.x; it < total; it += gridDim.x) {
        const uint2 e = list[it];
        const int n = (int)e.x, cell = (int)e.y;
        const KnnGrid g = grids[n];
        const uint32_t *off = offsets + (size_t)n * stride;
        const int64_t f0 = first_idx[n];
        const uint32_t s = off[cell], cnt = off[cell + 1] - s;
        if (cnt > (uint32_t)KNN_SUBSORT_MAX) continue;   // (workgroup-uniform)
        uint32_t m = 128;
        while (m < cnt) m <<= 1;
        const int cx = cell % g.res, cy = (cell / g.res) % g.res, cz = cell / (g.res * g.res);
        const float bx = g.minx + (float)cx * g.cell, by = g.miny + (float)cy * g.cell, bz = g.minz + (float)cz * g.cell;
        for (uint32_t i = tid; i < m; i += KNN_SUBSORT_THREADS) {
            uint32_t key = 0xffffffffu;
            if (i < cnt) {
                const float4 q = sorted[f0 + s + i];
                const uint32_t ux = (uint32_t)min(31, max(0, (int)((q.x - bx) * g.inv_cell * 32.0f)));
                const uint32_t uy = (uint32_t)min(31, max(0, (int)((q.y - by) * g.inv_cell * 32.0f)));
                const uint32_t uz = (uint32_t)min(31, max(0, (int)((q.z - bz) * g.inv_cell * 32.0f)));
                key = ((knn_spread5(ux) | (knn_spread5(uy) << 1) | (knn_spread5(uz) << 2)) << 12) | i;
            }
            s_key[i] = key;
        }
        __syncthreads();
        for (uint32_t k = 2; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < (m >> 1); t += KNN_SUBSORT_THREADS) {
                    const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;   // the pair (lo, lo ^ j)
                    const uint32_t a = s_key[lo], b = s_key[hi];
                    const bool up = (lo & k) == 0;
                    if ((a > b) == up) { s_key[lo] = b; s_key[hi] = a; }
                }
                __syncthreads();
            }
        float4 hold[KNN_SUBSORT_MAX / KNN_SUBSORT_THREADS];
#pragma unroll
        for (int u = 0; u < KNN_SUBSORT_MAX / KNN_SUBSORT_THREADS; ++u) {
            const uint32_t i = (uint32_t)u * KNN_SUBSORT_THREADS + tid;
            if (i < cnt) hold[u] = sorted[f0 + s + (s_key[i] & 0xfffu)];
        }
        __builtin_amdgcn_s_waitcnt(0);   // every read of the cell's slots has returned ...
        __syncthreads();                 // ... in every wavefront, before the first slot is overwritten
#pragma unroll
        for (int u = 0; u < KNN_SUBSORT_MAX / KNN_SUBSORT_THREADS; ++u) {
            const uint32_t i = (uint32_t)u * KNN_SUBSORT_THREADS + tid;
            if (i < cnt) sorted[f0 + s + i] = hold[u];
        }
        __syncthreads();   // (s_key is rewritten by the next cell)
    }
}

// boxes[2 b], boxes[2 b + 1] = min / max corner of the packed slots [16 b, 16 b + 16) of the cell-sorted array.  Slots that
// belong to no cloud hold stale bytes: a box can only grow by them (NaN is ignored by fminf / fmaxf), never lose a point.
__global__ __launch_bounds__(256) void knn_block_box_kernel(const float4 *__restrict__ sorted, int64_t P,
                                                            const uint32_t *__restrict__ dense_flag, float4 *__restrict__ boxes)
{
    if (*dense_flag == 0u) return;
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b * KNN_BLOCK >= P) return;
    float lo[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()};
    float hi[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
#pragma unroll
    for (int i = 0; i < KNN_BLOCK; ++i) {
        const int64_t j = b * KNN_BLOCK + i;
        if (j < P) {
            const float4 q = sorted[j];
            lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z);
            hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z);
        }
    }
    boxes[2 * b] = make_float4(lo[0], lo[1], lo[2], 0.f);
    boxes[2 * b + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

// ---------------------------------------------------------------------------------------------------------------
// Small inputs (P <= KNN_SMALL_P): the grid build is a chain of launch latencies (ten launches, ~45 us at 32k points for
// a few us of work).  Three of them disappear here: [bbox partials + zeroed counts] -> [grid + count] -> [scan, one
// launch] -> [fill] instead of memset, init, bbox, grid, count, scan x 2, fill.  (A whole-build-in-one-workgroup kernel was
// tried first: 180 us -- 32 points per thread through dependent load -> returning atomic -> store chains on ONE CU.)
// ---------------------------------------------------------------------------------------------------------------
#define KNN_SMALL_P 131072
#define KNN_BB_WGS 32           // bounding-box workgroups per cloud
// grid (KNN_BB_WGS + zero workgroups, N): the first KNN_BB_WGS workgroups of cloud n reduce their share of its points to
// one partial box each (plain stores: no init, no atomics); the others zero the cell counts of cloud n
__global__ __launch_bounds__(256) void knn_bbox_partial_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                               const int64_t *__restrict__ num_pts, int64_t P,
                                                               int *__restrict__ partial /* (N, KNN_BB_WGS, 6) */,
                                                               uint32_t *__restrict__ counts, size_t stride)
{
    const int n = blockIdx.y;
    if (blockIdx.x >= KNN_BB_WGS) {
        const size_t zw = gridDim.x - KNN_BB_WGS;
        uint32_t *c = counts + (size_t)n * stride;
        for (size_t i = (size_t)(blockIdx.x - KNN_BB_WGS) * 256 + threadIdx.x; i < stride; i += zw * 256) c[i] = 0u;
        return;
    }
    const int64_t f = first_idx[n], cnt = num_pts[n];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)KNN_BB_WGS * 256) {
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
    __shared__ int part[4][6];
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { part[wid][d] = lo[d]; part[wid][3 + d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        int v = part[0][d];
#pragma unroll
        for (int w = 1; w < 4; ++w) v = d < 3 ? min(v, part[w][d]) : max(v, part[w][d]);
        partial[((size_t)n * KNN_BB_WGS + blockIdx.x) * 6 + d] = v;
    }
}

// grid parameters of cloud n from the partial boxes (the arithmetic of knn_grid_kernel)
__device__ __forceinline__ KnnGrid knn_grid_from_partials(const int *__restrict__ partial, int n, int64_t npts, int res_cap, int lane)
{
    int v[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        v[d] = lane < KNN_BB_WGS ? partial[((size_t)n * KNN_BB_WGS + lane) * 6 + d] : (d < 3 ? 0x7fffffff : (int)0x80000000);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int t = __shfl_xor(v[d], o);
            v[d] = d < 3 ? min(v[d], t) : max(v[d], t);
        }
    }
    KnnGrid g;
    const float x0 = ord2f(v[0]), y0 = ord2f(v[1]), z0 = ord2f(v[2]);
    const float x1 = ord2f(v[3]), y1 = ord2f(v[4]), z1 = ord2f(v[5]);
    const float ext = fmaxf(fmaxf(x1 - x0, y1 - y0), fmaxf(z1 - z0, 1e-12f));
    int res = (int)ceilf(sqrtf((float)npts / 24.0f));
    res = max(1, min(res_cap, res));
    g.minx = x0; g.miny = y0; g.minz = z0;
    g.cell = ext / (float)res * 1.0001f;
    g.inv_cell = 1.0f / g.cell;
    g.res = res;
    g.pad0 = g.pad1 = 0;
    return g;
}

// count with the grid derived in place: one workgroup handles 256 consecutive packed points, its first wavefront reduces
// the partial boxes of the (at most KNN_GRID_LDS) clouds it touches into LDS; workgroup 0 also publishes grids[] for the
// later launches.  Clouds beyond KNN_GRID_LDS fall back to the separate grid kernel (host side).
#define KNN_GRID_LDS 16
__global__ __launch_bounds__(256) void knn_count_grid_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                             const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                             const int *__restrict__ partial, int res_cap,
                                                             KnnGrid *__restrict__ grids, size_t stride,
                                                             uint32_t *__restrict__ counts, int32_t *__restrict__ cell_of)
{
    __shared__ KnnGrid s_g[KNN_GRID_LDS];
    if (threadIdx.x < 64) {
        for (int n = 0; n < N; ++n) {
            const KnnGrid g = knn_grid_from_partials(partial, n, num_pts[n], res_cap, threadIdx.x);
            if (threadIdx.x == 0) {
                s_g[n] = g;
                if (blockIdx.x == 0) grids[n] = g;
            }
        }
    }
    __syncthreads();
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int n = find_cloud(p, first_idx, num_pts, N);
    if (n < 0) { cell_of[p] = -1; return; }
    const KnnGrid g = s_g[n];
    const int cx = cell_coord(pts[3 * p], g.minx, g.inv_cell, g.res);
    const int cy = cell_coord(pts[3 * p + 1], g.miny, g.inv_cell, g.res);
    const int cz = cell_coord(pts[3 * p + 2], g.minz, g.inv_cell, g.res);
    const int c = (cz * g.res + cy) * g.res + cx;
    cell_of[p] = c;
    atomicAdd(&counts[(size_t)n * stride + c], 1u);
}

// exclusive scan of the cell counts in ONE launch (at most KNN_SCAN1_BLOCKS blocks of 1024 cells per cloud): block b first
// sums the counts of all the cells in front of it (coalesced, eight independent loads in flight per thread: with 256
// threads and one load at a time the last block of 50 took 28 us), then scans its own 1024 cells, one per thread
#define KNN_SCAN1_BLOCKS 160
__global__ __launch_bounds__(1024) void knn_scan_single_kernel(const uint32_t *__restrict__ counts,
                                                               const KnnGrid *__restrict__ grids, size_t stride,
                                                               uint32_t *__restrict__ offsets, uint32_t *__restrict__ cursor)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t wave_pre[16];
    const int n = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int res = grids[n].res;
    const int cells = res * res * res;
    if (b * KNN_SCAN_BLOCK >= cells) return;
    const uint32_t *c = counts + (size_t)n * stride;
    const int lim = b * KNN_SCAN_BLOCK;
    uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, p5 = 0, p6 = 0, p7 = 0;
    int i = tid;
    for (; i + 7 * 1024 < lim; i += 8 * 1024) {
        p0 += c[i]; p1 += c[i + 1024]; p2 += c[i + 2048]; p3 += c[i + 3072];
        p4 += c[i + 4096]; p5 += c[i + 5120]; p6 += c[i + 6144]; p7 += c[i + 7168];
    }
    for (; i < lim; i += 1024) p0 += c[i];
    uint32_t part = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const int c0 = lim + tid;
    const uint32_t v = c0 < cells ? c[c0] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wid] = incl;
    if (lane == 0) wave_pre[wid] = part;
    __syncthreads();
    uint32_t before = 0, total = 0, prefix = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t t = wave_tot[w];
        before += (w < wid) ? t : 0u;
        total += t;
        prefix += wave_pre[w];
    }
    const uint32_t run = prefix + before + incl - v;
    if (c0 < cells) {
        offsets[(size_t)n * stride + c0] = run;
        cursor[(size_t)n * stride + c0] = run;
    }
    if (tid == 0 && (b + 1) * KNN_SCAN_BLOCK >= cells) offsets[(size_t)n * stride + cells] = prefix + total;
}

// FULL = false: K-th squared distance only (kth_sqdist (P,)).  FULL = true: the whole neighbour list, ascending in
// (distance, id): dists (P,Krt) squared distances and idx (P,Krt) cloud-local ids, zero-padded when the cloud has
// fewer than Krt points (the layout pytorch3d.ops.knn_points returns for a self query, losses.py:157-180).
template <int K, bool FULL, bool VIEW = false>
__global__ __launch_bounds__(256) void knn_query_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                        const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                        const KnnGrid *__restrict__ grids, size_t stride,
                                                        const uint32_t *__restrict__ offsets,
                                                        const float4 *__restrict__ sorted, int Krt,
                                                        float *__restrict__ kth_sqdist, float *__restrict__ dists,
                                                        int64_t *__restrict__ idx, float r2 /* > 0: FRNN semantics, see dss_knn_kth_sqdist_radius */,
    const KnnView view = KnnView(), const float4 *__restrict__ boxes = nullptr /* skip structure, see knn_subsort_kernel */,
    const uint32_t *__restrict__ dense_flag = nullptr, const int role = 0 /* 1: only clouds without dense cells, 2: only clouds with */,
    float *__restrict__ dk_out = nullptr /* !VIEW: the K-th distance itself (inf: fewer than K points), see knn_view_shortcut */,
    const float *__restrict__ plain_stat = nullptr, const float *__restrict__ plain_dk = nullptr /* VIEW: the unmasked search */)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= P) return;
    const bool skip = boxes != nullptr && *dense_flag != 0u;
    if ((role == 1 && skip) || (role == 2 && !skip)) return;   // (the launch next to this one takes the cloud)
    const int n = find_cloud(slot, first_idx, num_pts, N);
    if (n < 0) {  // packed slot outside every cloud
        if (FULL) {
            for (int k = 0; k < Krt; ++k) { dists[slot * Krt + k] = 0.0f; idx[slot * Krt + k] = 0; }
        } else {
            kth_sqdist[slot] = 0.0f;
        }
        return;
    }
    const KnnGrid g = grids[n];
    const int64_t f0 = first_idx[n];
    const int64_t cnt_n = num_pts[n];
    const uint32_t *off = offsets + (size_t)n * stride;
    // the query of this thread is the slot-th point of the cell-sorted order (every slot of a cloud's range holds
    // exactly one of its points): neighbouring lanes query neighbouring points
    const float4 self = sorted[slot];
    const int64_t p = (int64_t)__float_as_int(self.w);
    const float qx = self.x, qy = self.y, qz = self.z;
    float v2 = 0.f, v6 = 0.f, v10 = 0.f, v14 = 0.f, zn = 0.f, zf = 0.f;
    if (VIEW) {
        const int cam = view.mode == 1 ? (int)blockIdx.y : n;
        if (view.mode == 1 && view.culls[cam] == 0u) return;   // (the row is a copy: knn_view_rows_kernel)
        const float *vm = view.V + 16 * cam;
        v2 = vm[2]; v6 = vm[6]; v10 = vm[10]; v14 = vm[14]; zn = view.znear[cam]; zf = view.zfar[cam];
        if (view.mode == 1) kth_sqdist += (size_t)cam * (size_t)P;
        if (!knn_kept(qx, qy, qz, v2, v6, v10, v14, zn, zf)) { kth_sqdist[p] = 0.0f; return; }
        if (knn_view_shortcut(view.culls[cam], qx, qy, qz, v2, v6, v10, v14, zn, zf, plain_dk[p], r2)) { kth_sqdist[p] = plain_stat[p]; return; }
    }
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
    // One candidate: keep the K best in (distance, id) order (FULL) or by distance (kth only).
    auto consider = [&](const float4 q) __attribute__((always_inline)) {
        const float dx = q.x - qx, dy = q.y - qy, dz = q.z - qz;
        float d2 = dx * dx + dy * dy + dz * dz;
        if (VIEW && !knn_kept(q.x, q.y, q.z, v2, v6, v10, v14, zn, zf)) d2 = __builtin_huge_valf();   // the camera drops it
        if (FULL) {
            // total order (distance, id): deterministic lists whatever the cell order
            const int id = __float_as_int(q.w);
            // bitwise | and &: the short-circuit forms compile to two nested exec-mask branches per comparison
            if ((d2 < best[K - 1]) | ((d2 == best[K - 1]) & (id < bid[FULL ? K - 1 : 0]))) {
                bool lt[K];
#pragma unroll
                for (int k = 0; k < K; ++k) lt[k] = (d2 < best[k]) | ((d2 == best[k]) & (id < bid[FULL ? k : 0]));
#pragma unroll
                for (int k = K - 1; k >= 1; --k) {
                    // the empty asm keeps the operands values: left alone, the compiler rewrites select(c, bid[k-1],
                    // bid[k]) as a LOAD from select(c, &bid[k-1], &bid[k]), which pins bid[] in scratch memory and
                    // puts scratch loads/stores into this loop
                    float pb = best[k - 1];
                    int pi = bid[k - 1];
                    asm volatile("" : "+v"(pb), "+v"(pi));
                    best[k] = lt[k - 1] ? pb : (lt[k] ? d2 : best[k]);
                    bid[k] = lt[k - 1] ? pi : (lt[k] ? id : bid[k]);
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
    };
    // Candidates [s, e) of the cell-sorted array: cells that are consecutive along x are consecutive in memory, so a
    // whole (z, y) row of the search block is ONE contiguous range.  Four candidates are requested per memory round
    // trip (clamped, unconditional loads): the walk is latency-bound -- ~100 dependent 16-byte loads per query at
    // 2 wavefronts per CU took 210 us for 32k points when issued one at a time.
    // skip structure (clustered clouds, see knn_subsort_kernel): the slots [seed_lo, seed_hi) around the query's own slot
    // have been looked at before the rings start and are passed over afterwards; runs longer than KNN_DENSE slots are
    // walked block by block, a block whose box is farther than the current K-th distance is not read.  The box distance
    // uses the expression of `consider` on the clamped offsets, which are never larger in magnitude than those of a point
    // inside the box: rounding is monotone, so box distance <= point distance in fp32 as well -- and the comparison is
    // strict, so a candidate that ties with the K-th entry (and may win on its id) is still read.
    uint32_t seed_lo = 0, seed_hi = 0;
    auto visit = [&](uint32_t s, uint32_t e) __attribute__((always_inline)) {
        if (skip && e - s > (uint32_t)KNN_DENSE) {
            const uint64_t a0 = (uint64_t)f0 + s, a1 = (uint64_t)f0 + e;   // packed slots
            for (uint64_t b = a0 / KNN_BLOCK; b * KNN_BLOCK < a1; ++b) {
                const float4 mn = boxes[2 * b], mx = boxes[2 * b + 1];
                const float dx = fmaxf(fmaxf(mn.x - qx, qx - mx.x), 0.0f), dy = fmaxf(fmaxf(mn.y - qy, qy - mx.y), 0.0f),
                            dz = fmaxf(fmaxf(mn.z - qz, qz - mx.z), 0.0f);
                if (dx * dx + dy * dy + dz * dz > best[K - 1]) continue;
                const uint32_t j0 = (uint32_t)(max(b * KNN_BLOCK, a0) - (uint64_t)f0);
                const uint32_t j1 = (uint32_t)(min(b * KNN_BLOCK + KNN_BLOCK, a1) - (uint64_t)f0);
                for (uint32_t j = j0; j < j1; j += 4) {
                    const float4 q0 = sorted[f0 + j], q1 = sorted[f0 + min(j + 1, j1 - 1)], q2 = sorted[f0 + min(j + 2, j1 - 1)],
                                 q3 = sorted[f0 + min(j + 3, j1 - 1)];
                    const int cnt = (int)min(j1 - j, 4u);
#pragma nounroll
                    for (int i = 0; i < cnt; ++i)
                        if (j + i < seed_lo || j + i >= seed_hi) consider(i == 0 ? q0 : (i == 1 ? q1 : (i == 2 ? q2 : q3)));
                }
            }
            return;
        }
        for (uint32_t j = s; j < e; j += 4) {
            const float4 q0 = sorted[f0 + j], q1 = sorted[f0 + min(j + 1, e - 1)], q2 = sorted[f0 + min(j + 2, e - 1)],
                         q3 = sorted[f0 + min(j + 3, e - 1)];
            const int cnt = (int)min(e - j, 4u);
#pragma nounroll
            for (int i = 0; i < cnt; ++i) consider(i == 0 ? q0 : (i == 1 ? q1 : (i == 2 ? q2 : q3)));  // ONE copy of the insert
        }
    };
    if (skip) {
        // own cell dense (its points lie along a Morton curve): the neighbours of the query's slot are neighbours in space
        const int c_own = (cz * g.res + cy) * g.res + cx;
        const uint32_t cs = off[c_own], ce = off[c_own + 1];
        if (ce - cs > (uint32_t)KNN_DENSE) {
            const uint32_t me = (uint32_t)(slot - f0);
            seed_lo = me > cs + KNN_SEED ? me - KNN_SEED : cs;
            seed_hi = min(ce, me + KNN_SEED);
            for (uint32_t j = seed_lo; j < seed_hi; ++j) consider(sorted[f0 + j]);
        }
    }
    // The search starts with the 3x3x3 block (the own cell alone almost never proves K >= 7 neighbours final): the
    // offsets of its nine rows are requested together and parked in LDS (a register array indexed by a rolled loop
    // would go to scratch, nine unrolled copies of the walk cost 246 VGPRs).  Further rings, rarely needed, add their
    // shell: full rows on the y/z faces, the two end cells of the other rows.
    __shared__ uint32_t row_lo[9][256], row_hi[9][256];
    {
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.res - 1);
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
            const bool in = z >= 0 && z < g.res && y >= 0 && y < g.res;
            const int c = in ? (z * g.res + y) * g.res : 0;
            const uint32_t s = off[c + x0], e = off[c + x1 + 1];
            row_lo[r][threadIdx.x] = in ? s : 0u;
            row_hi[r][threadIdx.x] = in ? e : 0u;
        }
    }
    for (int ring = 1; ring <= g.res; ++ring) {
        const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.res - 1);
        const int y0 = max(cy - ring, 0), y1 = min(cy + ring, g.res - 1);
        const int z0 = max(cz - ring, 0), z1 = min(cz + ring, g.res - 1);
        const int ny = y1 - y0 + 1;
        const int walks = ring == 1 ? 9 : 2 * ny * (z1 - z0 + 1);  // rows of the block, or (row, end) pairs of the shell
#pragma nounroll
        for (int t = 0; t < walks; ++t) {
            uint32_t s, e;
            if (ring == 1) {
                s = row_lo[t][threadIdx.x];
                e = row_hi[t][threadIdx.x];
            } else {
                const int row = t >> 1, z = z0 + row / ny, y = y0 + row % ny;
                const int c = (z * g.res + y) * g.res;
                const bool full = z == cz - ring || z == cz + ring || y == cy - ring || y == cy + ring;
                const int xa = full ? x0 : ((t & 1) ? cx + ring : cx - ring);
                const int xb = full ? x1 : xa;
                if ((full && (t & 1)) || xa < 0 || xb >= g.res) continue;  // a full row is one walk
                s = off[c + xa];
                e = off[c + xb + 1];
            }
            visit(s, e);
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
        // fixed-radius statistic: nothing beyond r counts, and everything not yet visited is farther than `bound`
        if (!FULL && r2 > 0.0f && bound > 0.0f && bound * bound > r2) break;
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
    if (!VIEW && dk_out != nullptr) dk_out[p] = kk < Krt ? __builtin_huge_valf() : kth;
    if (r2 > 0.0f) {
        // fixed-radius semantics of the reference's default neighbour search (frnn_grid_points(K, r), rasterizer.py:317-326):
        // neighbours beyond r come back as -1 and the statistic is the MAX over the K - 1 returned distances -- the farthest
        // neighbour found within r, or -1 for a point without any
        kth = -1.0f;
#pragma unroll
        for (int k = 1; k < K; ++k) kth = (k < kk && best[k] <= r2) ? best[k] : kth;
    }
    kth_sqdist[p] = kth;
}

// ---------------------------------------------------------------------------------------------------------------
// Cooperative query: KNN_LPQ = 16 lanes (one DPP row) per query point.  The one-thread-per-point kernel above is a chain
// of ~100 dependent 16-byte loads per query with two wavefronts per CU at DSS sizes (32k points = 512 wavefronts): 58 us,
// the slowest kernel of a training iteration.  Here the 16 lanes of a group take every 16th
// candidate of each row range -- one coalesced 256-byte request per trip, ~18 dependent trips per query, 16 x the
// wavefronts to hide them behind --, keep their own sorted K-lists and merge them with four DPP rounds (quad_perm x 2,
// row_half_mirror, row_mirror: plain VALU moves inside the row, no LDS round trip), after which every lane holds the
// group's K best.  Same candidates, same (distance, id) order, same ring termination rule: identical results.
// Further rings (rare) start from the merged list in lane 0 and empty lists elsewhere, so nothing is counted twice.
// ---------------------------------------------------------------------------------------------------------------
#define KNN_LPQ 16
template <int CTRL>
__device__ __forceinline__ float knn_dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int knn_dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

// (distance, id) total order; ids are unique within a cloud, the padding entries (inf, 0x7fffffff) compare equal
__device__ __forceinline__ bool knn_less(float da, int ia, float db, int ib) { return (da < db) | ((da == db) & (ia < ib)); }

// this lane's ascending K-list merged with the list of the lane CTRL maps to: min(A[i], B[K-1-i]) picks the K smallest of
// the union as a bitonic sequence, an odd-even transposition network sorts it (see merge_round in raster_forward.hip)
template <int K, bool FULL, int CTRL>
__device__ __forceinline__ void knn_merge_round(float (&best)[K], int (&bid)[FULL ? K : 1])
{
    float ob[K];
    int oi[FULL ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ob[k] = knn_dpp_f<CTRL>(best[k]);
        if (FULL) oi[FULL ? k : 0] = knn_dpp_i<CTRL>(bid[FULL ? k : 0]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const bool lt = FULL ? knn_less(ob[K - 1 - k], oi[FULL ? K - 1 - k : 0], best[k], bid[FULL ? k : 0]) : ob[K - 1 - k] < best[k];
        best[k] = lt ? ob[K - 1 - k] : best[k];
        if (FULL) bid[FULL ? k : 0] = lt ? oi[FULL ? K - 1 - k : 0] : bid[FULL ? k : 0];
    }
#pragma unroll
    for (int round = 0; round < K; ++round) {
#pragma unroll
        for (int k = round & 1; k + 1 < K; k += 2) {
            const bool sw = FULL ? knn_less(best[k + 1], bid[FULL ? k + 1 : 0], best[k], bid[FULL ? k : 0]) : best[k + 1] < best[k];
            const float a = best[k], b = best[k + 1];
            best[k] = sw ? b : a;
            best[k + 1] = sw ? a : b;
            if (FULL) {
                const int x = bid[FULL ? k : 0], y = bid[FULL ? k + 1 : 0];
                bid[FULL ? k : 0] = sw ? y : x;
                bid[FULL ? k + 1 : 0] = sw ? x : y;
            }
        }
    }
}

template <int K, bool FULL, bool VIEW = false, bool SKIP = false /* compiled with the skip structure (P >= KNN_SKIP_MIN_P): +20 VGPRs */>
__global__ __launch_bounds__(256) void knn_query_coop_kernel(const float *__restrict__ pts, const int64_t *__restrict__ first_idx,
                                                             const int64_t *__restrict__ num_pts, int N, int64_t P,
                                                             const KnnGrid *__restrict__ grids, size_t stride,
                                                             const uint32_t *__restrict__ offsets,
                                                             const float4 *__restrict__ sorted, int Krt,
                                                             float *__restrict__ kth_sqdist, float *__restrict__ dists,
                                                             int64_t *__restrict__ idx, float r2 /* > 0: FRNN semantics, see dss_knn_kth_sqdist_radius */,
    const KnnView view = KnnView(), const uint32_t chunks = 0 /* VIEW: query chunks per camera (the grid is persistent) */,
    const float4 *__restrict__ boxes = nullptr /* skip structure, see knn_subsort_kernel */,
    const uint32_t *__restrict__ dense_flag = nullptr, const int role = 0 /* 1: only clouds without dense cells, 2: only clouds with */,
    float *__restrict__ dk_out = nullptr /* !VIEW: the K-th distance itself (inf: fewer than K points), see knn_view_shortcut */,
    const float *__restrict__ plain_stat = nullptr, const float *__restrict__ plain_dk = nullptr /* VIEW: the unmasked search */)
{
    const bool skip = SKIP && boxes != nullptr && *dense_flag != 0u;
    if ((role == 1 && skip) || (role == 2 && !skip)) return;
    constexpr int GPB = 256 / KNN_LPQ;   // query groups per workgroup
    const int grp = threadIdx.x / KNN_LPQ, sub = threadIdx.x % KNN_LPQ;
    __shared__ uint32_t row_lo[KNN_LPQ][GPB], row_hi[KNN_LPQ][GPB];
    float *const kth_base = kth_sqdist;
    // VIEW, one cloud seen by several cameras: a grid row covers the query chunks of ONE camera (at most 16,384 workgroups, the
    // rest by stride), there are two rows, and every workgroup takes its chunks for each camera of its row's share that drops
    // points, one camera after the other (two rows: half the serial depth when every camera drops points) -- a
    // camera that drops nothing costs a scalar load (its row is a copy, knn_view_rows_kernel), where a grid row per camera cost
    // a workgroup dispatch per chunk.  Otherwise: one pass, chunk = this workgroup.
    const bool walk = VIEW && view.mode == 1;
    for (int cam_y = walk ? (int)blockIdx.y : 0; cam_y < (walk ? view.n_cams : 1); cam_y += walk ? (int)gridDim.y : 1) {
    if (walk && view.culls[cam_y] == 0u) continue;
    for (uint32_t bx = blockIdx.x; bx < (walk ? chunks : blockIdx.x + 1u); bx += gridDim.x) {
    kth_sqdist = kth_base;
    const int64_t slot = (int64_t)bx * GPB + grp;
    if (slot >= P) continue;   // (whole DPP rows leave together)
    const int n = find_cloud(slot, first_idx, num_pts, N);
    if (n < 0) {  // packed slot outside every cloud
        if (sub == 0) {
            if (FULL) {
                for (int k = 0; k < Krt; ++k) { dists[slot * Krt + k] = 0.0f; idx[slot * Krt + k] = 0; }
            } else {
                kth_sqdist[slot] = 0.0f;
            }
        }
        continue;
    }
    const KnnGrid g = grids[n];
    const int64_t f0 = first_idx[n];
    const int64_t cnt_n = num_pts[n];
    const uint32_t *off = offsets + (size_t)n * stride;
    const float4 self = sorted[slot];   // the slot-th point of the cell-sorted order
    const int64_t p = (int64_t)__float_as_int(self.w);
    const float qx = self.x, qy = self.y, qz = self.z;
    float v2 = 0.f, v6 = 0.f, v10 = 0.f, v14 = 0.f, zn = 0.f, zf = 0.f;
    if (VIEW) {
        const int cam = view.mode == 1 ? cam_y : n;
        const float *vm = view.V + 16 * cam;
        v2 = vm[2]; v6 = vm[6]; v10 = vm[10]; v14 = vm[14]; zn = view.znear[cam]; zf = view.zfar[cam];
        if (view.mode == 1) kth_sqdist = kth_base + (size_t)cam * (size_t)P;
        if (!knn_kept(qx, qy, qz, v2, v6, v10, v14, zn, zf)) {   // (the whole 16-lane group shares the query: leaves together)
            if (sub == 0) kth_sqdist[p] = 0.0f;
            continue;
        }
        if (knn_view_shortcut(view.culls[cam], qx, qy, qz, v2, v6, v10, v14, zn, zf, plain_dk[p], r2)) {
            if (sub == 0) kth_sqdist[p] = plain_stat[p];
            continue;
        }
    }
    const int cx = cell_coord(qx, g.minx, g.inv_cell, g.res);
    const int cy = cell_coord(qy, g.miny, g.inv_cell, g.res);
    const int cz = cell_coord(qz, g.minz, g.inv_cell, g.res);
    float best[K];
    int bid[FULL ? K : 1];
#pragma unroll
    for (int k = 0; k < K; ++k) best[k] = __builtin_huge_valf();
#pragma unroll
    for (int k = 0; k < (FULL ? K : 1); ++k) bid[k] = 0x7fffffff;
    const int kk = (int)min((int64_t)Krt, cnt_n);
    if (kk <= 0) {
        if (!FULL && sub == 0) kth_sqdist[p] = 0.0f;
        continue;
    }
    auto consider = [&](const float4 q, bool on) __attribute__((always_inline)) {
        const float dx = q.x - qx, dy = q.y - qy, dz = q.z - qz;
        if (VIEW) on = on && knn_kept(q.x, q.y, q.z, v2, v6, v10, v14, zn, zf);   // the camera drops it
        const float d2 = on ? dx * dx + dy * dy + dz * dz : __builtin_huge_valf();
        if (FULL) {
            const int id = on ? __float_as_int(q.w) : 0x7fffffff;
            if (knn_less(d2, id, best[K - 1], bid[FULL ? K - 1 : 0])) {
                bool lt[K];
#pragma unroll
                for (int k = 0; k < K; ++k) lt[k] = knn_less(d2, id, best[k], bid[FULL ? k : 0]);
#pragma unroll
                for (int k = K - 1; k >= 1; --k) {
                    float pb = best[k - 1];
                    int pi = bid[FULL ? k - 1 : 0];
                    asm volatile("" : "+v"(pb), "+v"(pi));   // (keeps bid[] out of scratch, see knn_query_kernel)
                    best[k] = lt[k - 1] ? pb : (lt[k] ? d2 : best[k]);
                    bid[FULL ? k : 0] = lt[k - 1] ? pi : (lt[k] ? id : bid[FULL ? k : 0]);
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
    };
    // candidates [s, e) of the cell-sorted array, every KNN_LPQ-th one for this lane: the group's 16 loads of a trip are
    // one contiguous 256-byte run; two trips are requested together
    // skip structure (clustered clouds, see knn_subsort_kernel; the reasoning about exactness is in knn_query_kernel): `kb` is
    // a K-th distance the GROUP has established (after the seed, after every ring, and after a long run walked without one);
    // a lane prunes with the smaller of it and its own list's K-th entry.  Lane i tests the box of block i of sixteen; the
    // group then reads the surviving blocks one after the other, a slot per lane.
    float kb = __builtin_huge_valf();
    uint32_t seed_lo = 0, seed_hi = 0;
    const int gsh = (int)(threadIdx.x & 48u);   // bit position of the group's lanes in the wavefront's ballot
    auto merge_all = [&]() __attribute__((always_inline)) {
        knn_merge_round<K, FULL, 0xB1>(best, bid);
        knn_merge_round<K, FULL, 0x4E>(best, bid);
        knn_merge_round<K, FULL, 0x141>(best, bid);
        knn_merge_round<K, FULL, 0x140>(best, bid);
    };
    auto keep_lane0 = [&]() __attribute__((always_inline)) {   // lane 0 carries the merged list, the others start empty (nothing is counted twice)
        const bool drop = sub != 0;   // (selects on values: a branch around stores to bid[] sends the array to scratch memory)
#pragma unroll
        for (int k = 0; k < K; ++k) best[k] = drop ? __builtin_huge_valf() : best[k];
#pragma unroll
        for (int k = 0; k < (FULL ? K : 1); ++k) bid[k] = drop ? 0x7fffffff : bid[k];
    };
    auto visit = [&](uint32_t s, uint32_t e) __attribute__((always_inline)) {
        if (skip && e - s > (uint32_t)KNN_DENSE) {
            const uint64_t a0 = (uint64_t)f0 + s, a1 = (uint64_t)f0 + e;   // packed slots
            const uint64_t b_last = (a1 - 1) / KNN_BLOCK;
            for (uint64_t bb = a0 / KNN_BLOCK; bb <= b_last; bb += 2 * KNN_LPQ) {   // (group-uniform) 32 blocks per trip
                // two boxes per lane, their four loads in flight together: the walk is a chain of dependent round trips (boxes,
                // then the survivors' slots) at ~6 wavefronts per SIMD -- the fewer trips, the faster
                const uint64_t bA = bb + (uint64_t)sub, bB = bA + KNN_LPQ;
                const bool inA = bA <= b_last, inB = bB <= b_last;
                const float4 mnA = boxes[2 * (inA ? bA : b_last)], mxA = boxes[2 * (inA ? bA : b_last) + 1];
                const float4 mnB = boxes[2 * (inB ? bB : b_last)], mxB = boxes[2 * (inB ? bB : b_last) + 1];
                const float lim = fminf(kb, best[K - 1]);
                const float ax = fmaxf(fmaxf(mnA.x - qx, qx - mxA.x), 0.0f), ay = fmaxf(fmaxf(mnA.y - qy, qy - mxA.y), 0.0f),
                            az = fmaxf(fmaxf(mnA.z - qz, qz - mxA.z), 0.0f);
                const float cx2 = fmaxf(fmaxf(mnB.x - qx, qx - mxB.x), 0.0f), cy2 = fmaxf(fmaxf(mnB.y - qy, qy - mxB.y), 0.0f),
                            cz2 = fmaxf(fmaxf(mnB.z - qz, qz - mxB.z), 0.0f);
                const bool passA = inA && !(ax * ax + ay * ay + az * az > lim);
                const bool passB = inB && !(cx2 * cx2 + cy2 * cy2 + cz2 * cz2 > lim);
                unsigned gm = ((unsigned)(__ballot(passA) >> gsh) & 0xffffu) | (((unsigned)(__ballot(passB) >> gsh) & 0xffffu) << 16);
                while (gm != 0u) {
                    // up to four surviving blocks per trip, their loads in flight together
                    uint64_t j[4];
                    bool on[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool have = gm != 0u;
                        j[u] = (bb + (uint64_t)(have ? __builtin_ctz(gm) : 0)) * KNN_BLOCK + (uint64_t)sub;
                        gm &= gm - (have ? 1u : 0u);
                        const uint32_t jl = (uint32_t)(j[u] - (uint64_t)f0);
                        on[u] = have && j[u] >= a0 && j[u] < a1 && (jl < seed_lo || jl >= seed_hi);
                    }
                    const float4 q0 = sorted[on[0] ? j[0] : a0], q1 = sorted[on[1] ? j[1] : a0], q2 = sorted[on[2] ? j[2] : a0],
                                 q3 = sorted[on[3] ? j[3] : a0];
                    consider(q0, on[0]);
                    consider(q1, on[1]);
                    consider(q2, on[2]);
                    consider(q3, on[3]);
                }
            }
            if (kb == __builtin_huge_valf()) {   // a long run without a group bound: establish one now
                merge_all();
                kb = best[K - 1];
                keep_lane0();
            }
            return;
        }
        for (uint32_t base = s; base < e; base += 2 * KNN_LPQ) {   // (group-uniform trip count)
            const uint32_t j0 = base + (uint32_t)sub, j1 = j0 + KNN_LPQ;
            const bool on0 = j0 < e, on1 = j1 < e;
            const float4 q0 = sorted[f0 + (on0 ? j0 : s)], q1 = sorted[f0 + (on1 ? j1 : s)];   // clamped, unconditional
            consider(q0, on0);
            consider(q1, on1);
        }
    };
    if (skip) {
        // own cell dense (its points lie along a Morton curve): the neighbours of the query's slot are neighbours in space
        const int c_own = (cz * g.res + cy) * g.res + cx;
        const uint32_t cs = off[c_own], ce = off[c_own + 1];
        if (ce - cs > (uint32_t)KNN_DENSE) {
            const uint32_t me = (uint32_t)(slot - f0);
            seed_lo = me > cs + KNN_SEED ? me - KNN_SEED : cs;
            seed_hi = min(ce, me + KNN_SEED);
            for (uint32_t j = seed_lo + (uint32_t)sub; j < seed_lo + 2 * KNN_SEED; j += KNN_LPQ) {   // (two trips, group-uniform)
                const bool on = j < seed_hi;
                consider(sorted[f0 + (on ? j : seed_lo)], on);
            }
            merge_all();
            kb = best[K - 1];
            keep_lane0();
        }
    }
    uint32_t s1 = 0, e1 = 0;
    if (sub < 9) {   // lane r of the group fetches the offsets of row r of the 3 x 3 x 3 block
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.res - 1);
        const int z = cz + sub / 3 - 1, y = cy + sub % 3 - 1;
        const bool in = z >= 0 && z < g.res && y >= 0 && y < g.res;
        const int c = in ? (z * g.res + y) * g.res : 0;
        const uint32_t s = off[c + x0], e = off[c + x1 + 1];
        s1 = in ? s : 0u;
        e1 = in ? e : 0u;
        if (!SKIP) {
            row_lo[sub][grp] = s1;
            row_hi[sub][grp] = e1;
        }
    }
    if (!SKIP) __builtin_amdgcn_wave_barrier();   // same wavefront wrote and reads the row table (LDS operations of a wave complete in order)
    for (int ring = 1; ring <= g.res; ++ring) {
        const int x0 = max(cx - ring, 0), x1 = min(cx + ring, g.res - 1);
        const int y0 = max(cy - ring, 0), y1 = min(cy + ring, g.res - 1);
        const int z0 = max(cz - ring, 0), z1 = min(cz + ring, g.res - 1);
        const int ny = y1 - y0 + 1;
        const int walks = ring == 1 ? 9 : 2 * ny * (z1 - z0 + 1);
        if constexpr (!SKIP) {
            // (small clouds, P < KNN_SKIP_MIN_P: one walk after the other -- the batched form below costs this instantiation ten
            // VGPRs, i.e. the eighth wavefront per SIMD that the latency-bound launch at 32k points lives on)
#pragma nounroll
            for (int t = 0; t < walks; ++t) {
                uint32_t s, e;
                if (ring == 1) {
                    s = row_lo[t][grp];
                    e = row_hi[t][grp];
                } else {
                    const int row = t >> 1, z = z0 + row / ny, y = y0 + row % ny;
                    const int c = (z * g.res + y) * g.res;
                    const bool full = z == cz - ring || z == cz + ring || y == cy - ring || y == cy + ring;
                    const int xa = full ? x0 : ((t & 1) ? cx + ring : cx - ring);
                    const int xb = full ? x1 : xa;
                    if ((full && (t & 1)) || xa < 0 || xb >= g.res) continue;
                    s = off[c + xa];
                    e = off[c + xb + 1];
                }
                visit(s, e);
            }
        } else {
        // The sixteen lanes fetch the candidate ranges of sixteen walks together and the group visits the non-empty ones (ring 1:
        // the nine rows requested above).  One walk after the other is a dependent round trip per walk, most of them for an
        // EMPTY range: a stray point needs four or five rings -- ~600 walks --, and the launch lasts as long as its slowest
        // query (0.55 ms for the trained cloud of tools/clustered_timing.py whatever the rest of the kernel did).
#pragma nounroll
        for (int t0 = 0; t0 < walks; t0 += KNN_LPQ) {
            uint32_t s = s1, e = e1;
            if (ring > 1) {
                const int t = t0 + sub;
                s = e = 0u;
                if (t < walks) {
                    const int row = t >> 1, z = z0 + row / ny, y = y0 + row % ny;
                    const int c = (z * g.res + y) * g.res;
                    const bool full = z == cz - ring || z == cz + ring || y == cy - ring || y == cy + ring;
                    const int xa = full ? x0 : ((t & 1) ? cx + ring : cx - ring);
                    const int xb = full ? x1 : xa;
                    if (!((full && (t & 1)) || xa < 0 || xb >= g.res)) {   // (a full row is one walk)
                        s = off[c + xa];
                        e = off[c + xb + 1];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            row_lo[sub][grp] = s;
            row_hi[sub][grp] = e;
            __builtin_amdgcn_wave_barrier();   // (same wavefront writes and reads: LDS operations of a wave complete in order)
            unsigned ne = (unsigned)(__ballot(e > s) >> gsh) & 0xffffu;
            while (ne != 0u) {
                const int i = __builtin_ctz(ne);
                ne &= ne - 1u;
                visit(row_lo[i][grp], row_hi[i][grp]);
            }
        }
        }   // (SKIP)
        // the group's K best: four merge rounds inside the DPP row, every lane ends with the same list
        knn_merge_round<K, FULL, 0xB1>(best, bid);    // quad_perm [1,0,3,2]
        knn_merge_round<K, FULL, 0x4E>(best, bid);    // quad_perm [2,3,0,1]
        knn_merge_round<K, FULL, 0x141>(best, bid);   // row_half_mirror
        knn_merge_round<K, FULL, 0x140>(best, bid);   // row_mirror
        kb = best[K - 1];
        float bound = __builtin_huge_valf();
        if (cx - ring > 0) bound = fminf(bound, qx - (g.minx + (float)(cx - ring) * g.cell));
        if (cx + ring < g.res - 1) bound = fminf(bound, (g.minx + (float)(cx + ring + 1) * g.cell) - qx);
        if (cy - ring > 0) bound = fminf(bound, qy - (g.miny + (float)(cy - ring) * g.cell));
        if (cy + ring < g.res - 1) bound = fminf(bound, (g.miny + (float)(cy + ring + 1) * g.cell) - qy);
        if (cz - ring > 0) bound = fminf(bound, qz - (g.minz + (float)(cz - ring) * g.cell));
        if (cz + ring < g.res - 1) bound = fminf(bound, (g.minz + (float)(cz + ring + 1) * g.cell) - qz);
        bound = bound - 1e-6f * fmaxf(fabsf(bound), g.cell);
        float kth = best[0];
#pragma unroll
        for (int k = 1; k < K; ++k) kth = (k < kk) ? best[k] : kth;
        if (bound == __builtin_huge_valf() || (bound > 0.0f && kth <= bound * bound)) break;
        // fixed-radius statistic: nothing beyond r counts, and everything not yet visited is farther than `bound`
        if (!FULL && r2 > 0.0f && bound > 0.0f && bound * bound > r2) break;
        if (sub != 0) {   // next ring: lane 0 carries the merged list, the others start empty (nothing is counted twice)
#pragma unroll
            for (int k = 0; k < K; ++k) best[k] = __builtin_huge_valf();
#pragma unroll
            for (int k = 0; k < (FULL ? K : 1); ++k) bid[k] = 0x7fffffff;
        }
    }
    if (sub != 0) continue;
    if (!FULL && !VIEW && dk_out != nullptr) {
        float dk = best[0];
#pragma unroll
        for (int k = 1; k < K; ++k) dk = (k < kk) ? best[k] : dk;
        dk_out[p] = kk < Krt ? __builtin_huge_valf() : dk;
    }
    if (FULL) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (k < Krt) {
                dists[p * Krt + k] = k < kk ? best[k] : 0.0f;
                idx[p * Krt + k] = k < kk ? (int64_t)bid[FULL ? k : 0] - f0 : 0;
            }
        continue;
    }
    float kth = best[0];
#pragma unroll
    for (int k = 1; k < K; ++k) kth = (k < kk) ? best[k] : kth;
    if (r2 > 0.0f) {
        // fixed-radius semantics of the reference's default neighbour search (frnn_grid_points(K, r), rasterizer.py:317-326):
        // neighbours beyond r come back as -1 and the statistic is the MAX over the K - 1 returned distances -- the farthest
        // neighbour found within r, or -1 for a point without any
        kth = -1.0f;
#pragma unroll
        for (int k = 1; k < K; ++k) kth = (k < kk && best[k] <= r2) ? best[k] : kth;
    }
    kth_sqdist[p] = kth;
    }   // (chunks)
    }   // (cameras)
}

// deterministic per-cloud mean of values*scale clamped to [lo,hi]: one workgroup per cloud, fixed order (four independent
// partial sums per thread, wave reduction on shuffles, 16 wave totals: the 10-step LDS tree of round 1 took 19 us)
__global__ __launch_bounds__(1024) void cloud_mean_kernel(const float *__restrict__ vals, const int64_t *__restrict__ first_idx,
                                                          const int64_t *__restrict__ num_pts, float scale, float lo,
                                                          float hi, float fallback, int min_points,
                                                          float *__restrict__ out)
{
    __shared__ double part[16];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int64_t f0 = first_idx[n], cnt = num_pts[n];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int64_t i = tid;
    for (; i + 3 * 1024 < cnt; i += 4 * 1024) {
        const float v0 = vals[f0 + i], v1 = vals[f0 + i + 1024], v2 = vals[f0 + i + 2048], v3 = vals[f0 + i + 3072];
        a0 += (double)(v0 * scale); a1 += (double)(v1 * scale); a2 += (double)(v2 * scale); a3 += (double)(v3 * scale);
    }
    for (; i < cnt; i += 1024) a0 += (double)(vals[f0 + i] * scale);
    double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += part[w];
        float m = (cnt >= min_points && cnt > 0) ? (float)(t / (double)cnt) : fallback;
        out[n] = fminf(fmaxf(m, lo), hi);
    }
}

// The reference's Vrk_invariant statistic under culling (rasterizer.py:236-240, 183-217, 320-326): the cloud is extended
// to the N cameras, every camera DROPS the points outside its depth range, and h_n = clamp(mean over the PADDED cloud n of
// 0.5 max kNN-7 d^2) -- `h_k.mean(dim=1)` runs over the padded length P_max = the largest kept count of the batch, the padding
// contributing zeros.  Here the points are masked, not dropped: phase 1 sums values * scale over the points camera n keeps
// (the depth test of setup_point_compute, same expression) and counts them; phase 2 divides by the largest count.
// (values = K-th neighbour distances within the WHOLE cloud: a kept point whose 7 nearest include a dropped point sees a
// slightly smaller distance than the reference's search among the kept points -- second order, documented.)
#define RENDERABLE_BLOCKS 32   // workgroups per camera (one per camera left seven eighths of a 22 us launch to 8 CUs)
__global__ __launch_bounds__(1024) void renderable_sum_kernel(const float *__restrict__ vals, const float *__restrict__ world,
                                                              const float *__restrict__ V, const float *__restrict__ znear,
                                                              const float *__restrict__ zfar, const int64_t *__restrict__ first_idx,
                                                              const int64_t *__restrict__ num_pts, int shared, float scale,
                                                              double *__restrict__ part /* (N, RENDERABLE_BLOCKS, 2): sum, count */,
                                                              int64_t vals_cam_stride /* 0: one value per world point */)
{
    __shared__ double psum[16], pcnt[16];
    const int n = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const int64_t f0 = shared ? 0 : first_idx[n], cnt = num_pts[n];
    const float *v = V + 16 * n;
    const float v2 = v[2], v6 = v[6], v10 = v[10], v14 = v[14], zn = znear[n], zf = zfar[n];
    double a = 0.0;
    float c = 0.f;
    // block b of a camera takes the points b * 1024 + tid + k * (RENDERABLE_BLOCKS * 1024): a fixed assignment, so the sums
    // are reproducible; four points per trip with their loads in flight together
    const int64_t stride = (int64_t)RENDERABLE_BLOCKS * 1024;
    for (int64_t i0 = (int64_t)b * 1024 + tid; i0 < cnt; i0 += 4 * stride) {
        float x[4], y[4], z[4], q[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t i = i0 + u * stride;
            in[u] = i < cnt;
            const int64_t wi = f0 + (in[u] ? i : i0);
            x[u] = world[3 * wi]; y[u] = world[3 * wi + 1]; z[u] = world[3 * wi + 2];
            q[u] = vals[(size_t)n * (size_t)vals_cam_stride + wi];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float zview = x[u] * v2 + y[u] * v6 + z[u] * v10 + 1.0f * v14;   // the expression of setup_point_compute
            const bool ok = in[u] && (zview >= zn) && (zview <= zf);
            a += ok ? (double)(q[u] * scale) : 0.0;
            c += ok ? 1.f : 0.f;
        }
    }
    double cc = (double)c;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); cc += __shfl_xor(cc, o); }
    if ((tid & 63) == 0) { psum[tid >> 6] = a; pcnt[tid >> 6] = cc; }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0, tc = 0.0;
        for (int w = 0; w < 16; ++w) { t += psum[w]; tc += pcnt[w]; }
        part[((size_t)n * RENDERABLE_BLOCKS + b) * 2] = t;
        part[((size_t)n * RENDERABLE_BLOCKS + b) * 2 + 1] = tc;
    }
}
__global__ __launch_bounds__(64) void renderable_mean_kernel(double *__restrict__ part, int N, float lo, float hi,
                                                             float fallback, int min_points, float *__restrict__ out)
{
    // ONE wavefront.  Pass 1: lane l adds up the block partials of cameras l, l + 64, ... in block order (fixed) and parks
    // the totals in the camera's first slot; the largest kept count of the batch is a wave maximum.  Pass 2: the means.
    double cmax = 0.0;
    for (int n = threadIdx.x; n < N; n += 64) {
        double s = 0.0, c = 0.0;
#pragma unroll 8
        for (int b = 0; b < RENDERABLE_BLOCKS; ++b) {
            s += part[((size_t)n * RENDERABLE_BLOCKS + b) * 2];
            c += part[((size_t)n * RENDERABLE_BLOCKS + b) * 2 + 1];
        }
        part[(size_t)n * RENDERABLE_BLOCKS * 2] = s;
        part[(size_t)n * RENDERABLE_BLOCKS * 2 + 1] = c;
        cmax = fmax(cmax, c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cmax = fmax(cmax, __shfl_xor(cmax, o));
    for (int n = threadIdx.x; n < N; n += 64) {
        const double s = part[(size_t)n * RENDERABLE_BLOCKS * 2], cnt = part[(size_t)n * RENDERABLE_BLOCKS * 2 + 1];
        // `sq_dist[num_points_per_cloud < 7] = 1e-3` fills the whole padded row of a small cloud (rasterizer.py:322)
        const float m = (cnt >= (double)min_points && cmax > 0.0) ? (float)(s / cmax) : fallback;
        out[n] = fminf(fmaxf(m, lo), hi);
    }
}

}  // namespace dss

using namespace dss;

// Largest grid resolution a cloud of at most P points can ask for (knn_grid_kernel: ~8 points per occupied cell of a
// surface-like cloud), and the per-cloud stride of the cell arrays it implies (cells + end sentinel).
static int knn_res_cap(int64_t P)
{
    int res = (int)ceil(sqrt((double)(P > 0 ? P : 1) / 24.0));
    return res < 1 ? 1 : (res > KNN_MAX_RES ? KNN_MAX_RES : res);
}
static size_t knn_stride(int64_t P)
{
    const size_t r = (size_t)knn_res_cap(P);
    return r * r * r + 1;
}
static size_t knn_cells(int N, int64_t P) { return (size_t)(N > 0 ? N : 1) * knn_stride(P); }
static int knn_blocks(int64_t P) { return (int)((knn_stride(P) - 1 + KNN_SCAN_BLOCK - 1) / KNN_SCAN_BLOCK); }

#define KNN_MAX_VIEW_CAMS 1024
extern "C" size_t dss_knn_workspace(int N, int64_t P)
{
    const size_t n = N > 0 ? N : 1, p = P > 0 ? P : 1;
    return align_up(n * 6 * 4 * KNN_BB_WGS, 256) + align_up(n * sizeof(KnnGrid), 256) + align_up((knn_cells(N, P) + 1) * 4, 256) * 3 +
           align_up(n * (size_t)knn_blocks(P) * 4, 256) + align_up(p * 4, 256) + align_up(p * 16, 256) +
           (KNN_MAX_VIEW_CAMS + 64) * 4 +   // dss_knn_kth_sqdist_view: one "drops points" flag per camera; the "dense cells" flag
           align_up((p / KNN_BLOCK + 1) * 32, 256) +  // the block boxes of the skip structure (knn_subsort_kernel)
           align_up((p / KNN_DENSE_CELL + 1) * 8, 256) +  // and its list of dense cells
           align_up(p * 4, 256) * 3;                   // arrival numbers (knn_count_kernel); dss_knn_kth_sqdist_view: the unmasked search's two rows
}

#define KNN_FULL_MAX_K 40
#define KNN_COOP_KTH_MAX_P 120000   // K-th distance: crossover of the two query kernels between 65k (cooperative +24 %) and 131k points (tie), profiles/r4_d_knn_sweep.json

// Per-cloud bounding boxes as ordered ints (decode with ord2f), for callers outside this file (regularizers.hip).
int dss::launch_cloud_bbox(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P,
                           int *bbox, hipStream_t st)
{
    hipLaunchKernelGGL(knn_init_kernel, dim3((6 * N + 63) / 64), dim3(64), 0, st, N, bbox);
    const unsigned bb = (unsigned)((P / N + 2047) / 2048 > 64 ? 64 : (P / N + 2047) / 2048);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(bb ? bb : 1, N), dim3(256), 0, st, points, first_idx, num_pts, N, P, bbox);
    return check_launch("cloud bbox");
}

// grid build + query; exactly one of (kth_sqdist) / (dists, idx) is written
static int knn_run(const char *who, const float *points, const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P,
                   int K, float *kth_sqdist, float *dists, int64_t *idx, void *workspace, size_t workspace_bytes,
                   void *stream, float r2 = -1.0f, KnnView view = KnnView(), int n_cams = 1)
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
    int *bbox = reinterpret_cast<int *>(w + off);                 off += align_up((size_t)N * 6 * 4 * KNN_BB_WGS, 256);  // box, or the partial boxes
    KnnGrid *grids = reinterpret_cast<KnnGrid *>(w + off);        off += align_up((size_t)N * sizeof(KnnGrid), 256);
    const size_t cbytes = align_up((knn_cells(N, P) + 1) * 4, 256);
    const size_t stride = knn_stride(P);
    const int nblk = knn_blocks(P);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + off);     off += cbytes;
    uint32_t *offsets = reinterpret_cast<uint32_t *>(w + off);    off += cbytes;
    uint32_t *cursor = reinterpret_cast<uint32_t *>(w + off);     off += cbytes;
    uint32_t *blk_tot = reinterpret_cast<uint32_t *>(w + off);    off += align_up((size_t)N * nblk * 4, 256);
    int32_t *cell_of = reinterpret_cast<int32_t *>(w + off);      off += align_up((size_t)P * 4, 256);
    float4 *sorted = reinterpret_cast<float4 *>(w + off);                    off += align_up((size_t)P * 16, 256);
    uint32_t *flags = reinterpret_cast<uint32_t *>(w + off);         off += (KNN_MAX_VIEW_CAMS + 64) * 4;   // [cameras | dense]
    float4 *boxes = reinterpret_cast<float4 *>(w + off);            off += align_up(((size_t)P / KNN_BLOCK + 1) * 32, 256);
    uint2 *dense_list = reinterpret_cast<uint2 *>(w + off);         off += align_up(((size_t)P / KNN_DENSE_CELL + 1) * 8, 256);
    uint32_t *rank_of = reinterpret_cast<uint32_t *>(w + off);      off += align_up((size_t)P * 4, 256);
    float *plain_stat = reinterpret_cast<float *>(w + off);         off += align_up((size_t)P * 4, 256);
    float *plain_dk = reinterpret_cast<float *>(w + off);           off += align_up((size_t)P * 4, 256);
    uint32_t *dense_flag = flags + KNN_MAX_VIEW_CAMS, *n_dense = dense_flag + 1, *n_dense_pts = dense_flag + 2;
    const bool skip = P >= KNN_SKIP_MIN_P && option(DSS_OPT_KNN_QUERY) != 3;   // (3: the uniform-grid walk whatever the cloud, for A/B)
    if (view.mode != 0) {
        if (n_cams > KNN_MAX_VIEW_CAMS) { set_error("%s: at most %d cameras", who, KNN_MAX_VIEW_CAMS); return DSS_ERR_UNSUPPORTED; }
        view.culls = flags;
        view.n_cams = n_cams;
        if (hipMemsetAsync(view.culls, 0, (size_t)n_cams * 4, st) != hipSuccess) return check_launch("knn view memset");
    }
    if (skip && hipMemsetAsync(dense_flag, 0, 12, st) != hipSuccess) return check_launch("knn flag memset");
    const unsigned pb_s = (unsigned)((P + 255) / 256);
    if (P <= KNN_SMALL_P && N <= KNN_GRID_LDS && nblk <= KNN_SCAN1_BLOCKS) {
        // small inputs: four launches instead of eight (see knn_bbox_partial_kernel)
        int *partial = bbox;   // (N, KNN_BB_WGS, 6)
        const unsigned zero_wgs = (unsigned)((stride + 4095) / 4096);
        hipLaunchKernelGGL(knn_bbox_partial_kernel, dim3(KNN_BB_WGS + (zero_wgs ? zero_wgs : 1), N), dim3(256), 0, st, points,
                           first_idx, num_pts, P, partial, counts, stride);
        hipLaunchKernelGGL(knn_count_grid_kernel, dim3(pb_s), dim3(256), 0, st, points, first_idx, num_pts, N, P, partial,
                           knn_res_cap(P), grids, stride, counts, cell_of);
        hipLaunchKernelGGL(knn_scan_single_kernel, dim3(nblk, N), dim3(1024), 0, st, counts, grids, stride, offsets, cursor);
        hipLaunchKernelGGL(knn_fill_kernel, dim3(pb_s), dim3(256), 0, st, points, first_idx, num_pts, N, P, cell_of, stride,
                           cursor, sorted, view);
    } else {
        if (hipMemsetAsync(counts, 0, cbytes, st) != hipSuccess) return check_launch("knn memset");
        const unsigned pb = (unsigned)((P + 255) / 256);
        if (int rc = launch_cloud_bbox(points, first_idx, num_pts, N, P, bbox, st)) return rc;
        hipLaunchKernelGGL(knn_grid_kernel, dim3((N + 63) / 64), dim3(64), 0, st, bbox, num_pts, N, knn_res_cap(P), grids);
        hipLaunchKernelGGL(knn_count_kernel, dim3(pb), dim3(256), 0, st, points, first_idx, num_pts, N, P, grids, stride,
                           counts, cell_of, rank_of);
        hipLaunchKernelGGL(knn_scan_local_kernel, dim3(nblk, N), dim3(256), 0, st, counts, grids, stride, nblk, offsets,
                           blk_tot);
        hipLaunchKernelGGL(knn_scan_add_kernel, dim3(nblk, N), dim3(256), 0, st, grids, stride, nblk, blk_tot, offsets, cursor);
        hipLaunchKernelGGL(knn_fill_kernel, dim3(pb), dim3(256), 0, st, points, first_idx, num_pts, N, P, cell_of, stride,
                           offsets, sorted, view, rank_of);
    }
    if (skip) {
        hipLaunchKernelGGL(knn_dense_list_kernel, dim3((unsigned)((stride + 254) / 256), N), dim3(256), 0, st, grids, stride, offsets,
                           dense_list, n_dense, n_dense_pts);
        hipLaunchKernelGGL(knn_subsort_kernel, dim3(KNN_SUBSORT_WGS), dim3(KNN_SUBSORT_THREADS), 0, st, grids, stride, offsets, first_idx, sorted,
                           dense_list, n_dense, n_dense_pts, P, dense_flag);
        hipLaunchKernelGGL(knn_block_box_kernel, dim3((unsigned)((P / KNN_BLOCK + 256) / 256)), dim3(256), 0, st, sorted, P,
                           dense_flag, boxes);
    }
    const float4 *bx = skip ? boxes : nullptr;
    const uint32_t *df = skip ? dense_flag : nullptr;
    // small inputs: one wavefront per workgroup, so that the few hundred wavefronts spread over all 256 CUs
    const unsigned qt = P <= 131072 ? 64u : 256u;
    const unsigned qb = (unsigned)((P + qt - 1) / qt);
    const unsigned gy = view.mode == 1 ? (unsigned)n_cams : 1u;
    // ROLE 0: the launch takes every cloud (and uses the skip structure where `dense_flag` is up); 2: it runs only for clouds
    // with dense cells, next to a cooperative launch that leaves those alone
#define KNN_LAUNCH_R(KK, FF, ROLE)                                                                                  \
    hipLaunchKernelGGL((knn_query_kernel<KK, FF>), dim3(qb), dim3(qt), 0, st, points, first_idx, num_pts, N, P, grids,  \
                       stride, offsets, sorted, K, kth_sqdist, dists, idx, r2, KnnView(), bx, df, ROLE)
#define KNN_LAUNCH(KK, FF) KNN_LAUNCH_R(KK, FF, 0)
    // cooperative kernel: 16 lanes per query (K <= 16); the one-thread-per-query kernel keeps the deep lists
    const unsigned cb = (unsigned)((P + (256 / KNN_LPQ) - 1) / (256 / KNN_LPQ));
#define KNN_LAUNCH_COOP_R(KK, FF, ROLE)                                                                             \
    do {                                                                                                            \
        if (skip)                                                                                                   \
            hipLaunchKernelGGL((knn_query_coop_kernel<KK, FF, false, true>), dim3(cb), dim3(256), 0, st, points, first_idx, num_pts, \
                               N, P, grids, stride, offsets, sorted, K, kth_sqdist, dists, idx, r2, KnnView(), 0u, bx, df, ROLE); \
        else                                                                                                        \
            hipLaunchKernelGGL((knn_query_coop_kernel<KK, FF>), dim3(cb), dim3(256), 0, st, points, first_idx, num_pts, N, P, \
                               grids, stride, offsets, sorted, K, kth_sqdist, dists, idx, r2, KnnView(), 0u, bx, df, ROLE); \
    } while (0)
#define KNN_LAUNCH_COOP(KK, FF) KNN_LAUNCH_COOP_R(KK, FF, 0)
    // a size at which the one-thread kernel is the choice for an evenly sampled cloud, K <= 16: clouds with dense cells go to
    // the cooperative kernel all the same (sixteen boxes tested per trip), as a second launch; each launch leaves at once
    // when the cloud is not its kind
#define KNN_LAUNCH_BOTH(KK, FF)                                                                                     \
    do {                                                                                                            \
        if (skip) { KNN_LAUNCH_R(KK, FF, 1); KNN_LAUNCH_COOP_R(KK, FF, 2); } else KNN_LAUNCH(KK, FF);                 \
    } while (0)
    const int qopt = option(DSS_OPT_KNN_QUERY) == 3 ? 0 : option(DSS_OPT_KNN_QUERY);   // 0: by size, 1: cooperative, 2: one thread per query (3: by size, no skip structure)
    if (view.mode != 0) {
        // K-th distance under per-camera culling (K <= 8: the variance-scale statistic): one grid row per camera
        if (full || K > 8) { set_error("%s: the per-camera search is built for the K-th distance with K <= 8", who); return DSS_ERR_UNSUPPORTED; }
        const bool coop = qopt == 1 || (qopt == 0 && P <= KNN_COOP_KTH_MAX_P);
        // (1) the unmasked search of every point: statistic + K-th distance (see knn_view_shortcut)
        if (coop) {
            if (skip)
                hipLaunchKernelGGL((knn_query_coop_kernel<8, false, false, true>), dim3(cb), dim3(256), 0, st, points, first_idx,
                                   num_pts, N, P, grids, stride, offsets, sorted, K, plain_stat, dists, idx, r2, KnnView(), 0u, bx, df, 0,
                                   plain_dk);
            else
                hipLaunchKernelGGL((knn_query_coop_kernel<8, false>), dim3(cb), dim3(256), 0, st, points, first_idx, num_pts, N, P,
                                   grids, stride, offsets, sorted, K, plain_stat, dists, idx, r2, KnnView(), 0u, bx, df, 0, plain_dk);
        } else if (skip) {
            // (clouds with dense cells go to the cooperative kernel at every size, see KNN_LAUNCH_BOTH)
            hipLaunchKernelGGL((knn_query_kernel<8, false>), dim3(qb), dim3(qt), 0, st, points, first_idx, num_pts, N, P, grids,
                               stride, offsets, sorted, K, plain_stat, dists, idx, r2, KnnView(), bx, df, 1, plain_dk);
            hipLaunchKernelGGL((knn_query_coop_kernel<8, false, false, true>), dim3(cb), dim3(256), 0, st, points, first_idx,
                               num_pts, N, P, grids, stride, offsets, sorted, K, plain_stat, dists, idx, r2, KnnView(), 0u, bx, df, 2,
                               plain_dk);
        } else {
            hipLaunchKernelGGL((knn_query_kernel<8, false>), dim3(qb), dim3(qt), 0, st, points, first_idx, num_pts, N, P, grids,
                               stride, offsets, sorted, K, plain_stat, dists, idx, r2, KnnView(), bx, df, 0, plain_dk);
        }
        // (2) per camera: copy, or search again among the points the camera keeps
        if (coop) {
            // one cloud, several cameras: the grid covers one camera's chunks (see knn_query_coop_kernel)
            const dim3 grid = view.mode == 1 ? dim3(cb < 16384u ? cb : 16384u, n_cams > 1 ? 2u : 1u) : dim3(cb);
            if (skip)
                hipLaunchKernelGGL((knn_query_coop_kernel<8, false, true, true>), grid, dim3(256), 0, st, points, first_idx,
                                   num_pts, N, P, grids, stride, offsets, sorted, K, kth_sqdist, dists, idx, r2, view, cb, bx, df, 0,
                                   nullptr, plain_stat, plain_dk);
            else
                hipLaunchKernelGGL((knn_query_coop_kernel<8, false, true>), grid, dim3(256), 0, st, points, first_idx, num_pts,
                                   N, P, grids, stride, offsets, sorted, K, kth_sqdist, dists, idx, r2, view, cb, bx, df, 0, nullptr,
                                   plain_stat, plain_dk);
        } else {
            hipLaunchKernelGGL((knn_query_kernel<8, false, true>), dim3(qb, gy), dim3(qt), 0, st, points, first_idx, num_pts, N, P,
                               grids, stride, offsets, sorted, K, kth_sqdist, dists, idx, r2, view, bx, df, 0, nullptr, plain_stat,
                               plain_dk);
        }
        if (view.mode == 1)   // the rows of the cameras that drop nothing = the unmasked search
            hipLaunchKernelGGL(knn_view_rows_kernel, dim3((unsigned)((P + 1023) / 1024), (unsigned)n_cams), dim3(256), 0, st, kth_sqdist,
                               plain_stat, P, view.culls);
        return check_launch(who);
    }
    if (full) {
        // full lists: the cooperative kernel wins while the launch is latency-bound (32k points, K = 12: 58 us against
        // ~100); at 100k points of an evenly sampled cloud the merges of (distance, id) lists cost more than the shorter
        // chains save (182 vs 155 us) -- but the one-thread kernel falls off a cliff as soon as cells fill up (30-60 points
        // per cell, the training loop on its way to the clustered state: 1.0 ms), the cooperative one does not
        const bool coop = qopt == 1 || (qopt == 0 && P <= KNN_COOP_KTH_MAX_P);
        if (K <= 8) { if (coop) KNN_LAUNCH_COOP(8, true); else KNN_LAUNCH_BOTH(8, true); }
        else if (K <= 12) { if (coop) KNN_LAUNCH_COOP(12, true); else KNN_LAUNCH_BOTH(12, true); }  // the regularisers' knn_k (trainer.py:134-137)
        else if (K <= 16) { if (coop) KNN_LAUNCH_COOP(16, true); else KNN_LAUNCH_BOTH(16, true); }
        else KNN_LAUNCH(KNN_FULL_MAX_K, true);
    } else {
        // K-th distance only: cooperative up to KNN_COOP_KTH_MAX_P points (tools/knn_sweep.py, profiles/r4_d_knn_sweep.json)
        const bool coop = qopt == 1 || (qopt == 0 && P <= KNN_COOP_KTH_MAX_P);
        if (K <= 8) { if (coop) KNN_LAUNCH_COOP(8, false); else KNN_LAUNCH_BOTH(8, false); }
        else { if (coop) KNN_LAUNCH_COOP(KNN_MAX_K, false); else KNN_LAUNCH_BOTH(KNN_MAX_K, false); }
    }
#undef KNN_LAUNCH_BOTH
#undef KNN_LAUNCH_COOP
#undef KNN_LAUNCH_COOP_R
#undef KNN_LAUNCH
#undef KNN_LAUNCH_R
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
    if (P == 0 && N > 0 && K >= 1 && K <= KNN_FULL_MAX_K) return DSS_OK;  // empty input: nothing to write
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

// dss_cloud_mean_clamp under the reference's depth culling, for clouds whose culled points are masked instead of dropped
// (see renderable_sum_kernel).  values (Pw,) per WORLD point; shared_cloud = 1: one cloud of num_pts[0] points seen by N
// cameras; workspace: 16 N bytes.
extern "C" int dss_renderable_mean_clamp(const float *values, const float *world, const float *V, const float *znear,
                                         const float *zfar, const int64_t *first_idx, const int64_t *num_pts, int N,
                                         int shared_cloud, float scale, float lo, float hi, float fallback, int min_points,
                                         int64_t values_cam_stride, float *out, void *workspace, size_t workspace_bytes,
                                         void *stream)
{
    if (N <= 0 || !values || !world || !V || !znear || !zfar || !first_idx || !num_pts || !out || !workspace ||
        workspace_bytes < (size_t)N * 16 * RENDERABLE_BLOCKS) {
        set_error("dss_renderable_mean_clamp: bad arguments (workspace of 512 N bytes)");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    double *sums = reinterpret_cast<double *>(workspace);
    hipLaunchKernelGGL(renderable_sum_kernel, dim3(RENDERABLE_BLOCKS, N), dim3(1024), 0, as_stream(stream), values, world, V, znear, zfar,
                       first_idx, num_pts, shared_cloud, scale, sums, values_cam_stride);
    hipLaunchKernelGGL(renderable_mean_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, N, lo, hi, fallback, min_points, out);
    return check_launch("dss_renderable_mean_clamp");
}

// dss_knn_kth_sqdist with the fixed-radius semantics of the reference's DEFAULT neighbour search: SurfaceSplatting is built
// with frnn_radius = 0.2 (rasterizer.py:110) and then calls frnn.frnn_grid_points(K = 7, r = frnn_radius) (:317, :373), which
// reports neighbours beyond r as -1 [third party, lxxue/FRNN, not vendored: behaviour as restated by tests/ref_loop/launcher.py];
// the statistic 0.5 * max over the K - 1 returned distances is then the farthest neighbour FOUND within r -- or -0.5 for an
// isolated point, which pulls the cloud's mean h down once points have drifted away from the surface.  radius <= 0: the plain
// K-th distance (the reference's knn_points branch, frnn_radius <= 0).
extern "C" int dss_knn_kth_sqdist_radius(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N,
                                         int64_t P, int K, float radius, float *kth_sqdist, void *workspace,
                                         size_t workspace_bytes, void *stream)
{
    return knn_run("dss_knn_kth_sqdist_radius", points, first_idx, num_pts, N, P, K, kth_sqdist, nullptr, nullptr, workspace,
                   workspace_bytes, stream, radius > 0.0f ? radius * radius : -1.0f);
}

// dss_knn_kth_sqdist[_radius] in the reference's ORDER under depth culling: filter_renderable extends the cloud to the
// cameras and drops, per camera, the points outside [znear, zfar] BEFORE the neighbour search (rasterizer.py:599, 236-240,
// 183-217, 310-326), so a point's neighbours are the ones the same camera keeps.  shared_cloud = 1: ONE cloud (N must be 1)
// seen by n_cams cameras; kth_sqdist (n_cams, P): row c = the statistic among the points camera c keeps (0 for the points it
// drops).  shared_cloud = 0: cloud n is seen by camera n (n_cams == N); kth_sqdist (P,).  K <= 8.  One grid build, one query
// launch with a grid row per camera.
extern "C" int dss_knn_kth_sqdist_view(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N,
                                       int64_t P, int K, float radius, const float *V, const float *znear, const float *zfar,
                                       int n_cams, int shared_cloud, float *kth_sqdist, void *workspace, size_t workspace_bytes,
                                       void *stream)
{
    if (!V || !znear || !zfar || n_cams <= 0 || (shared_cloud ? N != 1 : n_cams != N)) {
        set_error("dss_knn_kth_sqdist_view: cameras missing, or the cloud / camera counts do not fit (shared: one cloud; else one camera per cloud)");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    KnnView view = {V, znear, zfar, shared_cloud ? 1 : 2, nullptr, n_cams};
    return knn_run("dss_knn_kth_sqdist_view", points, first_idx, num_pts, N, P, K, kth_sqdist, nullptr, nullptr, workspace,
                   workspace_bytes, stream, radius > 0.0f ? radius * radius : -1.0f, view, n_cams);
}
