// Forward EWA splat rasterizer for gfx950 (MI355X).
//
// Replaces the reference's naive / coarse / fine CUDA kernels
// (DSS/csrc/rasterize_points.cu:131-212, 293-432, 506-597) with a different decomposition:
//
//   bin         one thread per splat: exact pixel rect -> 8x8 screen-tile rect; the splat id is
//               appended straight into fixed-capacity per-tile sub-lists (one returning atomic per
//               (splat, tile) pair; no count/scan/fill passes, no single-workgroup scan).  A tile whose
//               sub-list overflows is rasterized from the whole cloud instead (exact, just slower).
//   fine        one 256-thread workgroup per tile = four wavefronts, one 4x4 pixel footprint
//               each, four candidate slices per pixel.  Candidates are staged through LDS in
//               chunks of 256; every wavefront culls the chunk against its footprint with one
//               ballot per 64 candidates and compacts the survivors (mbcnt prefix); every lane keeps
//               the K nearest hits of its (pixel, slice) sorted in registers; the four slices are
//               merged with xor-shuffles; results leave through an LDS transpose so that each image
//               row of the tile is written as one contiguous run.
//
// The per-pair arithmetic (dx, dy, Q, comparisons) is written exactly like the reference
// (rasterize_points.cu:64-124) and compiled with -ffp-contract=off, so fragments are bit-identical
// to the reference CPU/CUDA naive path; the K-set is defined by the total order (z, idx).
#include "setup_body.h"

namespace dss {

// Each tile owns DSS_SUB counters / sub-lists, selected by the low bits of the splat id.
// Same-address global atomics serialise (~50 ns each on MI355X: 523 splats on the hottest 16x16 tile of
// the bunny scene cost ~30 us); splitting cuts the depth of every hot address by DSS_SUB.
// Sub-lists have a fixed capacity `cap` (workspace layout: counts (N*tiles*SUB) uint32, then lists
// (N*tiles*SUB*cap) int32), sized ~32x the mean load (see bin_capacity); a count above `cap` marks the
// tile as overflowed.
#define DSS_SUB 8

// Heavy-first dispatch.  The fine kernel's duration is set by its slowest workgroups (the densest tiles run
// 2-3x longer than the mean, tools/fine_timing.py) and a launch needs two occupancy rounds at 512^2, so a
// dense tile that is dispatched late ends the kernel late.  While binning, the thread that brings sub-list 0
// of a tile to DSS_HEAVY_SUB0 entries (~8x that in the whole tile) appends the tile to a short queue and
// flags it; the fine kernel's first DSS_HEAVY_MAX workgroups serve the queue, the others skip flagged
// tiles.  Queue counter and flags live in the memset region of the tile counters.
#define DSS_HEAVY_MAX 2048
#define DSS_HEAVY_SUB0 4
#define DSS_HEAVY_QUEUES 32  // independent queue heads: ~1100 appends per launch on ONE counter serialise (+6 us)
struct HeavyQ {
    uint32_t *count;  // DSS_HEAVY_QUEUES words, zeroed with the tile counters (only the binning pass uses them)
    int32_t *list;    // DSS_HEAVY_MAX slots holding tile id + 1, 0 = empty; zeroed with the tile counters
    uint8_t *flag;    // one byte per tile, zeroed with the tile counters
};

struct TileGrid {
    int S;        // image side
    int row0;     // first image row of the band
    int rows;     // rows in the band
    int tiles_x;  // tiles per band row
    int tiles_y;  // tile rows in the band
};

// ---------------------------------------------------------------------------------------------
// Splat -> tile rectangle (band-local tile coordinates).  Image column c <-> NDC index S-1-c.
// The rectangle is exact: it is the set of tiles containing at least one pixel whose centre
// passes both axis tests |dx|<=rx and |dy|<=ry (the Q test can only remove pixels).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool splat_tile_rect(float px, float py, float pz, float rx, float ry,
                                                const TileGrid g, int &tx0, int &tx1, int &ty0, int &ty1)
{
    if (pz < 0) return false;  // rasterize_points.cu:79-80
    int xlo, xhi, ylo, yhi;
    if (!ndc_index_range(px, rx, g.S, xlo, xhi)) return false;
    if (!ndc_index_range(py, ry, g.S, ylo, yhi)) return false;
    // tighten with the exact per-pixel predicate (monotone in the pixel index).  The loops run 1-3 times; they
    // must stay rolled and free of the IEEE divide of the non-power-of-two pixel map (unrolled eightfold with the
    // divide inlined they were 1100 instructions and 8.5k of the binning kernel's 30k cycles per wavefront).
    const NdcMap ndc(g.S);
    // (the empty asm keeps the optimiser from turning each search into an eightfold-unrolled batch evaluation)
#define DSS_TIGHTEN(NDC_EXPR)                                                              \
    while (xlo <= xhi && fabsf(NDC_EXPR(xlo) - px) > rx) { ++xlo; asm volatile("" : "+v"(xlo)); } \
    while (xhi >= xlo && fabsf(NDC_EXPR(xhi) - px) > rx) { --xhi; asm volatile("" : "+v"(xhi)); } \
    while (ylo <= yhi && fabsf(NDC_EXPR(ylo) - py) > ry) { ++ylo; asm volatile("" : "+v"(ylo)); } \
    while (yhi >= ylo && fabsf(NDC_EXPR(yhi) - py) > ry) { --yhi; asm volatile("" : "+v"(yhi)); }
    if (ndc.pow2) {  // uniform
        const float inv = ndc.invS;
#define DSS_NDC_POW2(i) (-1 + (2 * (i) + 1.0f) * inv)
        DSS_TIGHTEN(DSS_NDC_POW2)
#undef DSS_NDC_POW2
    } else {
#define DSS_NDC_DIV(i) pix_to_ndc((i), g.S)
        DSS_TIGHTEN(DSS_NDC_DIV)
#undef DSS_NDC_DIV
    }
#undef DSS_TIGHTEN
    if (xlo > xhi || ylo > yhi) return false;
    const int c0 = g.S - 1 - xhi, c1 = g.S - 1 - xlo;
    int r0 = g.S - 1 - yhi, r1 = g.S - 1 - ylo;
    r0 = max(r0, g.row0);
    r1 = min(r1, g.row0 + g.rows - 1);
    if (r0 > r1) return false;
    tx0 = c0 / DSS_TILE;
    tx1 = c1 / DSS_TILE;
    ty0 = (r0 - g.row0) / DSS_TILE;
    ty1 = (r1 - g.row0) / DSS_TILE;
    return true;
}

// exactly one thread per tile gets here (the one whose atomic brought sub-list 0 to the threshold)
__device__ __forceinline__ void mark_heavy(const HeavyQ hq, int tile)
{
    if (!hq.count) return;
    // queue q holds list slots q, q + QUEUES, q + 2 QUEUES, ...: workgroup b of the fine kernel reads slot b,
    // so the queues drain interleaved
    const uint32_t q = (uint32_t)tile & (DSS_HEAVY_QUEUES - 1);
    const uint32_t pos = atomicAdd(&hq.count[q], 1u);
    const uint32_t slot = pos * DSS_HEAVY_QUEUES + q;
    if (slot < DSS_HEAVY_MAX) {
        hq.list[slot] = tile + 1;  // 0 = empty slot
        hq.flag[tile] = 1;  // only queued tiles are flagged: a full queue leaves the rest to the normal workgroups
    }
}

// append splat p to the sub-list (p mod SUB) of every tile of its rectangle
__device__ __forceinline__ void bin_point(int64_t p, int n, float px, float py, float pz, float rx, float ry,
                                          const TileGrid g, uint32_t *__restrict__ counts,
                                          int32_t *__restrict__ lists, uint32_t cap, const HeavyQ hq)
{
    if (n < 0) return;
    int tx0, tx1, ty0, ty1;
    if (!splat_tile_rect(px, py, pz, rx, ry, g, tx0, tx1, ty0, ty1)) return;
    const size_t sub0 = ((size_t)n * g.tiles_x * g.tiles_y) * DSS_SUB + ((unsigned)p & (DSS_SUB - 1));
    if (tx1 - tx0 <= 1 && ty1 - ty0 <= 1) {
        // common case (splat overlaps at most 2x2 tiles): the returning atomics are independent, issue
        // them back to back so their latencies overlap instead of chaining
        const size_t t00 = sub0 + (size_t)(ty0 * g.tiles_x + tx0) * DSS_SUB;
        const size_t t01 = t00 + DSS_SUB, t10 = t00 + (size_t)g.tiles_x * DSS_SUB, t11 = t10 + DSS_SUB;
        const bool hx = tx1 > tx0, hy = ty1 > ty0;
        uint32_t p0, p1 = 0, p2 = 0, p3 = 0;
        p0 = atomicAdd(&counts[t00], 1u);
        if (hx) p1 = atomicAdd(&counts[t01], 1u);
        if (hy) p2 = atomicAdd(&counts[t10], 1u);
        if (hx && hy) p3 = atomicAdd(&counts[t11], 1u);
        if (p0 < cap) lists[t00 * cap + p0] = (int32_t)p;
        if (hx && p1 < cap) lists[t01 * cap + p1] = (int32_t)p;
        if (hy && p2 < cap) lists[t10 * cap + p2] = (int32_t)p;
        if (hx && hy && p3 < cap) lists[t11 * cap + p3] = (int32_t)p;
        if (((unsigned)p & (DSS_SUB - 1)) == 0) {  // sub-list 0 decides
            if (p0 == DSS_HEAVY_SUB0 - 1) mark_heavy(hq, (int)(t00 / DSS_SUB));
            if (hx && p1 == DSS_HEAVY_SUB0 - 1) mark_heavy(hq, (int)(t01 / DSS_SUB));
            if (hy && p2 == DSS_HEAVY_SUB0 - 1) mark_heavy(hq, (int)(t10 / DSS_SUB));
            if (hx && hy && p3 == DSS_HEAVY_SUB0 - 1) mark_heavy(hq, (int)(t11 / DSS_SUB));
        }
        return;
    }
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) {
            const size_t t = sub0 + (size_t)(ty * g.tiles_x + tx) * DSS_SUB;
            const uint32_t pos = atomicAdd(&counts[t], 1u);
            if (pos < cap) lists[t * cap + pos] = (int32_t)p;
            if (((unsigned)p & (DSS_SUB - 1)) == 0 && pos == DSS_HEAVY_SUB0 - 1) mark_heavy(hq, (int)(t / DSS_SUB));
        }
}

__global__ __launch_bounds__(256) void bin_kernel(
    const float *__restrict__ points, const float *__restrict__ radii,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t P,
    TileGrid g, uint32_t *__restrict__ counts, int32_t *__restrict__ lists, uint32_t cap, HeavyQ hq,
    uint8_t *__restrict__ visible_to_clear /* (P) or nullptr */)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (visible_to_clear) visible_to_clear[p] = 0;  // saves a separate memset launch
    const int n = find_cloud(p, first_idx, num_pts, N);
    bin_point(p, n, points[3 * p], points[3 * p + 1], points[3 * p + 2], radii[2 * p], radii[2 * p + 1], g, counts,
              lists, cap, hq);
}

// dss_render_forward: per-point setup (culling + projection + EWA terms) fused with the binning --
// the screen record goes from registers straight into the tile lists.
__global__ __launch_bounds__(256) void setup_bin_kernel(const SetupArgs A, TileGrid g, uint32_t *__restrict__ counts,
                                                        int32_t *__restrict__ lists, uint32_t cap, HeavyQ hq,
                                                        uint8_t *__restrict__ visible_to_clear)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    if (visible_to_clear) visible_to_clear[p] = 0;
    const int n = find_cloud(p, A.first_idx, A.num_pts, A.N);
    float px, py, pz, rx, ry;
    setup_point(A, p, n, px, py, pz, rx, ry);
    bin_point(p, n, px, py, pz, rx, ry, g, counts, lists, cap, hq);
}

// ---------------------------------------------------------------------------------------------
// Fine pass.
// ---------------------------------------------------------------------------------------------

struct FineArgs {
    const float *points, *ellipse, *cutoff, *radii;
    const int64_t *first_idx, *num_pts;
    const uint32_t *counts;    // (N*tiles*DSS_SUB) sub-list fill counts, or nullptr (naive mode)
    const int32_t *lists;      // (N*tiles*DSS_SUB*cap)
    uint32_t cap;              // sub-list capacity
    HeavyQ heavy;              // heavy-first queue (count == nullptr: identity order)
    uint32_t *clean_counts;    // DSS_WS_CLEAN: == counts, every owner resets what it has read; else nullptr
    int32_t *idx;
    float *zbuf, *qv, *occ;
    uint8_t *visible;
    TileGrid g;
    int N, K;
    float thr;
    // optional fused blend (dss_render_forward): image (N,rows,S,C+1) and wsum (N,rows,S)
    const float *scaler, *feat;
    float *image, *wsum;
    int C;
    // element strides of `image` over (camera, band row); a pixel's C+1 channels are contiguous and pixels of a
    // row are contiguous.  Dense (N,rows,S,C+1): rows*S*(C+1) and S*(C+1).  The multi-GPU send buffer is laid out
    // (row, camera, col, ch) so that the all-gathered bands ARE the full image, no reassembly copy.
    long long img_sn, img_sr;
};

// block id -> tile id.  Identity on purpose.  Consecutive workgroup ids are dealt round-robin to the 8
// XCDs (each with its own L2); an XCD-contiguous mapping (XCD x owns tiles [x*T/8, (x+1)*T/8)) was
// measured with an A/B build (round 1, not kept) and is SLOWER (512^2 bunny: 37.3 vs 33.4 us; 8 x 1024^2, 1M points: 1.36
// vs 1.31 ms): the dense screen region then sits on one XCD, and the write traffic is already at the
// algorithmic minimum (PMC WRITE_SIZE 22.7 MB vs 22.5 MB) so there is nothing for the shared L2 to merge.
// Round-robin placement doubles as load balancing here.
__device__ __forceinline__ int xcd_tile(unsigned b, int total) { return (int)b < total ? (int)b : -1; }

// Candidate source of one tile: its DSS_SUB fixed-capacity sub-lists (binned mode) or the whole cloud
// (naive mode, or a tile whose sub-list overflowed).  `at(i)` maps the i-th candidate to a splat id.
struct TileSource {
    bool use_list;
    int64_t first;          // cloud scan: first packed index
    int64_t count;
    const int32_t *base;    // binned: &lists[tile*SUB*cap]
    uint32_t cap;
    uint32_t ps[DSS_SUB + 1];  // prefix sums of the sub-list lengths
    __device__ __forceinline__ void init(const FineArgs &A, int n, int tile_id)
    {
        use_list = false;
        if (A.counts != nullptr) {
            const uint4 *c4 = reinterpret_cast<const uint4 *>(A.counts + (size_t)tile_id * DSS_SUB);
            uint32_t c[DSS_SUB];
#pragma unroll
            for (int q = 0; q < DSS_SUB / 4; ++q) {
                const uint4 u = c4[q];
                c[4 * q] = u.x; c[4 * q + 1] = u.y; c[4 * q + 2] = u.z; c[4 * q + 3] = u.w;
            }
            bool ok = true;
            ps[0] = 0;
#pragma unroll
            for (int q = 0; q < DSS_SUB; ++q) {
                ok = ok && (c[q] <= A.cap);
                ps[q + 1] = ps[q] + c[q];
            }
            use_list = ok;
            count = ps[DSS_SUB];
            cap = A.cap;
            base = A.lists + (size_t)tile_id * DSS_SUB * A.cap;
        }
        if (!use_list) {
            first = A.first_idx[n];
            count = A.num_pts[n];
        }
    }
    __device__ __forceinline__ int64_t at(int64_t i) const
    {
        if (!use_list) return first + i;
        // ps[] is only ever indexed with compile-time constants: a dynamic ps[sub] sends the whole array to
        // scratch memory (measured: +60 MB of HBM writes per launch at 512^2)
        uint32_t sub = 0, start = 0;
#pragma unroll
        for (int q = 1; q < DSS_SUB; ++q) {
            const bool ge = (uint32_t)i >= ps[q];
            sub = ge ? (uint32_t)q : sub;
            start = ge ? ps[q] : start;
        }
        return (int64_t)base[(size_t)sub * cap + ((uint32_t)i - start)];
    }
};

// K-nearest bookkeeping: one 64-bit key per slot, (z bits << 32) | idx.  Hits have z >= 0
// (pz < 0 is culled, rasterize_points.cu:79-80), for which the IEEE bit pattern is monotone, so
// one unsigned 64-bit compare implements the strict total order (z, idx) of
// rasterize_points_cpu.cpp:85 / oracle frag_less.  Empty slots hold ~0.
#define KEY_EMPTY 0xffffffffffffffffull

// Optional per-workgroup phase timestamps (tools/fine_timing.py builds a -DDSS_FINE_TIMING copy of the
// library; never compiled into the shipped libdss_hip.so).
#ifdef DSS_FINE_TIMING
__device__ long long *g_fine_timing = nullptr;  // (blocks, 12) int64
#define FT_MARK(slot)                                                                   \
    do {                                                                                \
        if (g_fine_timing && threadIdx.x == 0)                                          \
            g_fine_timing[(size_t)blockIdx.x * 12 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
#define FT_VAL(slot, v)                                                                 \
    do {                                                                                \
        if (g_fine_timing && threadIdx.x == 0) g_fine_timing[(size_t)blockIdx.x * 12 + (slot)] = (long long)(v); \
    } while (0)
#else
#define FT_MARK(slot)
#define FT_VAL(slot, v)
#endif

// sorted insertion of (ekey, eq) into an ascending K-list held in registers; branch-free.
template <int KMAX>
__device__ __forceinline__ void klist_insert(unsigned long long (&key)[KMAX], float (&kq)[KMAX],
                                             unsigned long long ekey, float eq)
{
    bool lt[KMAX];  // e < slot[k] on the old list (monotone in k)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) lt[k] = ekey < key[k];
#pragma unroll
    for (int k = KMAX - 1; k >= 1; --k) {
        key[k] = lt[k - 1] ? key[k - 1] : (lt[k] ? ekey : key[k]);
        kq[k] = lt[k - 1] ? kq[k - 1] : (lt[k] ? eq : kq[k]);
    }
    key[0] = lt[0] ? ekey : key[0];
    kq[0] = lt[0] ? eq : kq[0];
}

// Work decomposition of one 8x8 tile (one 256-thread workgroup = 4 wavefronts):
//   wavefront w  -> 4x4 pixel footprint (w%2, w/2) of the tile
//   lane l       -> pixel (l%16) of the footprint, candidate slice l/16 (4 slices)
// A pixel's candidates are split 4 ways across lanes 16 apart; each lane keeps its own K-list and
// the four lists are merged at the end with two xor-shuffle rounds.  At DSS sizes this kernel is
// bound by instruction issue on the CUs that host the densest screen regions (tools/fine_timing.py),
// not by bandwidth: small tiles spread a dense region over several CUs, and the candidate slices cut
// the serial per-pixel chain 4x.
#define FOOT 4
#define FOOT_PER_ROW (DSS_TILE / FOOT)
#define FINE_WAVES (FOOT_PER_ROW * FOOT_PER_ROW)
#define FINE_THREADS (FINE_WAVES * 64)
#define CHUNK FINE_THREADS

template <int KMAX>
__device__ __forceinline__ void fine_tile(const FineArgs &A, const int tile_id)
{
    // candidate chunk: three records per splat -> 2x ds_read_b128 + 1x ds_read_b64 per test
    __shared__ float4 s_geo[CHUNK];   // px, py, rx, ry
    __shared__ float4 s_ell[CHUNK];   // a, b, c, cutoff
    __shared__ float2 s_zid[CHUNK];   // pz, idx (bits)
    __shared__ unsigned short s_surv[FINE_WAVES][CHUNK];  // per-wavefront compacted survivor slots
    constexpr int PLANES = (KMAX <= 8) ? 3 : 1;   // idx / zbuf / qvalue staged together when they fit
    __shared__ int s_out[PLANES][DSS_TILE_PIX * KMAX];

    FT_MARK(0);
    FT_VAL(8, __builtin_amdgcn_s_memrealtime());
    const TileGrid g = A.g;
    const int tiles = g.tiles_x * g.tiles_y;
    const int n = tile_id / tiles;
    const int t = tile_id - n * tiles;
    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fx = wid % FOOT_PER_ROW, fy = wid / FOOT_PER_ROW;  // footprint inside the tile
    const int pl = lane & 15, slice = lane >> 4;      // pixel inside the footprint, candidate slice
    const int tr = fy * FOOT + (pl >> 2), tc = fx * FOOT + (pl & 3);
    const int r = g.row0 + ty * DSS_TILE + tr;        // image row
    const int c = tx * DSS_TILE + tc;                 // image col
    const int S = g.S;
    const NdcMap ndc(S);  // same values as pix_to_ndc; one multiply instead of an IEEE divide when S = 2^k
    const float xf = ndc(S - 1 - c);
    const float yf = ndc(S - 1 - r);
    // NDC extent of this wavefront's footprint (pixel centres).  NDC decreases with the image index.
    const int fc0 = tx * DSS_TILE + fx * FOOT, fr0 = g.row0 + ty * DSS_TILE + fy * FOOT;
    const float f_xmax = ndc(S - 1 - fc0), f_xmin = ndc(S - 1 - (fc0 + 3));
    const float f_ymax = ndc(S - 1 - fr0), f_ymin = ndc(S - 1 - (fr0 + 3));

    // candidate source: tile sub-lists (binned) or the whole cloud (naive / overflowed tile)
    TileSource src;
    src.init(A, n, tile_id);
    const int64_t count = src.count;

    // rows of the tile are contiguous runs of 16*K dwords in the (N,rows,S,K) tensors
    const int K = A.K;
    const int run = DSS_TILE * K;
    const int c0 = tx * DSS_TILE;
    const int valid_cols = min(DSS_TILE, S - c0) * K;
    const int valid_rows = min(DSS_TILE, g.rows - ty * DSS_TILE);
    const size_t tile_base = (((size_t)n * g.rows + (size_t)ty * DSS_TILE) * S + c0) * K;

    FT_MARK(1);
    FT_VAL(10, count);
    if (count <= 0) {
        // empty tile (most of the screen): stream the fill values, no LDS, no barriers
        for (int rr = wid; rr < valid_rows; rr += FINE_WAVES) {
            const size_t rb = tile_base + (size_t)rr * S * K;
            for (int cc = lane; cc < valid_cols; cc += 64) {
                A.idx[rb + cc] = -1;
                A.zbuf[rb + cc] = -1.0f;
                A.qv[rb + cc] = -1.0f;
            }
            if (lane < min(DSS_TILE, S - c0)) {
                const size_t pix = ((size_t)n * g.rows + (size_t)ty * DSS_TILE + rr) * S + c0 + lane;
                A.occ[pix] = 0.0f;
                if (A.image) {  // fused blend of an empty pixel: zeros, weight sum clamped to kEpsilon
                    float *o = A.image + (size_t)n * A.img_sn + (size_t)(ty * DSS_TILE + rr) * A.img_sr +
                               (size_t)(c0 + lane) * (A.C + 1);
                    if (A.C == 3) {
                        *reinterpret_cast<float4 *>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        for (int ch = 0; ch <= A.C; ++ch) o[ch] = 0.0f;
                    }
                    A.wsum[pix] = 1e-4f;
                }
            }
        }
        FT_MARK(7);
        FT_VAL(9, __builtin_amdgcn_s_memrealtime());
        return;
    }

    unsigned long long key[KMAX];
    float kq[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        key[k] = KEY_EMPTY;
        kq[k] = -1.0f;
    }
    unsigned short *surv = s_surv[wid];

    for (int64_t base = 0; base < count; base += CHUNK) {
        const int m = (int)min((int64_t)CHUNK, count - base);
        __syncthreads();  // previous chunk fully consumed
        // DSS_WS_CLEAN: every thread has read the tile's counters (src.init) before this barrier
        if (base == 0 && A.clean_counts && tid < DSS_SUB) A.clean_counts[(size_t)tile_id * DSS_SUB + tid] = 0;
        if (tid < m) {
            const int64_t p = src.at(base + tid);
            const float px = A.points[3 * p], py = A.points[3 * p + 1], pz = A.points[3 * p + 2];
            const float2 rr = reinterpret_cast<const float2 *>(A.radii)[p];
            s_geo[tid] = make_float4(px, py, rr.x, rr.y);
            s_ell[tid] = make_float4(A.ellipse[3 * p], A.ellipse[3 * p + 1], A.ellipse[3 * p + 2], A.cutoff[p]);
            s_zid[tid] = make_float2(pz, __int_as_float((int)p));
        }
        __syncthreads();
        if (base == 0) FT_MARK(2);
        // ---- cull + compact: one candidate per lane, ballot, prefix popcount -> survivor list ----
        int nsurv = 0;
        for (int sub = 0; sub < m; sub += 64) {
            const int j = sub + lane;
            bool keep = false;
            if (j < m) {
                const float4 ge = s_geo[j];
                // conservative, rounding-monotone rejection against the footprint (see splat_tile_rect)
                const bool out = (s_zid[j].x < 0) || ((f_xmax - ge.x) < -ge.z) || ((f_xmin - ge.x) > ge.z) ||
                                 ((f_ymax - ge.y) < -ge.w) || ((f_ymin - ge.y) > ge.w);
                keep = !out;
            }
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                surv[nsurv + rank] = (unsigned short)j;
            }
            nsurv += __popcll(mask);
        }
        // same wavefront wrote and now reads `surv`: LDS ops of one wave complete in order; the wave
        // barrier only stops the compiler from moving the reads above the writes
        __builtin_amdgcn_wave_barrier();
        if (base == 0) FT_MARK(3);
        // ---- test + insert: lane (pixel, slice) takes survivors slice, slice+4, ... ----
        const int trips = (nsurv + 3) >> 2;
        for (int it = 0; it < trips; ++it) {
            const int si = it * 4 + slice;
            const bool live = si < nsurv;
            const int jj = live ? (int)surv[si] : 0;
            const float4 ge = s_geo[jj], el = s_ell[jj];
            const float2 zi = s_zid[jj];
            const float dx = xf - ge.x;
            const float dy = yf - ge.y;
            // rasterize_points.cu:92-101, same expression order (no FMA contraction)
            const float qval = el.x * dx * dx + el.y * dx * dy + el.z * dy * dy;
            const bool hit = live && !(fabsf(dx) > ge.z || fabsf(dy) > ge.w) && !(qval > el.w);
            const unsigned long long ekey =
                hit ? (((unsigned long long)__float_as_uint(zi.x + 0.0f) << 32) |
                       (unsigned long long)(unsigned)__float_as_int(zi.y))
                    : KEY_EMPTY;
            // A hit can only reach the output if it beats this lane's current K-th entry AND lies within
            // the depth-merge threshold of the nearest entry seen so far (the final nearest is never
            // farther, so dropping it now is exact: rasterize_points.cu:586-595 would drop it later).
            const float znear_now = __uint_as_float((unsigned)(key[0] >> 32));
            const bool useful = (ekey < key[KMAX - 1]) && !(key[0] != KEY_EMPTY && (zi.x - znear_now > A.thr));
            if (__ballot(useful) != 0ull) klist_insert<KMAX>(key, kq, useful ? ekey : KEY_EMPTY, qval);
        }
    }

    FT_MARK(4);
    // ---- merge the four candidate slices of every pixel (lanes 16 and 32 apart) ----
    // Two ascending K-lists A, B -> the K smallest of their union: min(A[i], B[K-1-i]) over i picks exactly
    // those K (as a bitonic sequence), an odd-even transposition network sorts them.  K(K-1)/2 + K compare-
    // exchanges of 7 VALU each (90 at K=5) instead of K branch-free insertions of ~12K each (300): the merge
    // was a third of the kernel's VALU instructions.  Keys are unique across slices (disjoint candidates)
    // except KEY_EMPTY, whose payload is the same everywhere.
#pragma unroll
    for (int xo = 16; xo <= 32; xo <<= 1) {
        unsigned long long okey[KMAX];
        float oq[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned lo = __shfl_xor((unsigned)key[k], xo, 64);
            const unsigned hi = __shfl_xor((unsigned)(key[k] >> 32), xo, 64);
            okey[k] = ((unsigned long long)hi << 32) | lo;
            oq[k] = __shfl_xor(kq[k], xo, 64);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const bool lt = okey[KMAX - 1 - k] < key[k];
            key[k] = lt ? okey[KMAX - 1 - k] : key[k];
            kq[k] = lt ? oq[KMAX - 1 - k] : kq[k];
        }
#pragma unroll
        for (int round = 0; round < KMAX; ++round) {
#pragma unroll
            for (int k = round & 1; k + 1 < KMAX; k += 2) {
                const bool sw = key[k + 1] < key[k];
                const unsigned long long ka = key[k], kb = key[k + 1];
                const float qa = kq[k], qb = kq[k + 1];
                key[k] = sw ? kb : ka;
                key[k + 1] = sw ? ka : kb;
                kq[k] = sw ? qb : qa;
                kq[k + 1] = sw ? qa : qb;
            }
        }
    }

    FT_MARK(5);
    // ---- epilogue (slice 0 lanes own the pixel): depth merge, occupancy, visibility, stores ----
    const bool owner = slice == 0;
    const bool in_img = owner && (c < S) && (r < g.row0 + g.rows);
    float kz[KMAX];
    int ki[KMAX];
    const float z0 = __uint_as_float((unsigned)(key[0] >> 32));
    const bool any = key[0] != KEY_EMPTY;
    bool alive = any;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const float z = __uint_as_float((unsigned)(key[k] >> 32));
        // rasterize_points.cu:586-595: stop at the first k with z[k]-z[0] > thr
        alive = alive && (key[k] != KEY_EMPTY) && !(z - z0 > A.thr);
        ki[k] = alive ? (int)(unsigned)(key[k] & 0xffffffffull) : -1;
        kz[k] = alive ? z : -1.0f;
        kq[k] = alive ? kq[k] : -1.0f;
    }
    if (in_img) {
        const size_t pix = ((size_t)n * g.rows + (r - g.row0)) * S + c;
        A.occ[pix] = any ? 1.0f : 0.0f;
        if (A.visible) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K && ki[k] >= 0) A.visible[ki[k]] = 1;
        }
        if (A.image) {
            // fused blend (same arithmetic and order as blend_forward_kernel): w = exp(-q/2)*scaler,
            // img = sum f*w/cum, alpha = occupancy
            float wk[KMAX];
            float cum = 0.0f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                wk[k] = 0.0f;
                if (k < K && ki[k] >= 0) {
                    wk[k] = expf(-0.5f * kq[k]) * A.scaler[ki[k]];
                    cum += wk[k];
                }
            }
            if (cum < 1e-4f) cum = 1e-4f;
            A.wsum[pix] = cum;
            // normalised weights once per fragment (K IEEE divides per pixel, not K*C): img = sum f * (w / cum)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) wk[k] = wk[k] / cum;
            float *o = A.image + (size_t)n * A.img_sn + (size_t)(r - g.row0) * A.img_sr + (size_t)c * (A.C + 1);
            if (A.C == 3) {
                // RGBA as ONE 16-byte store per pixel (four dword stores at a 16-byte stride quadruple the
                // write requests the memory side sees)
                float acc3[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K && ki[k] >= 0) {
                        const float *f = A.feat + (size_t)ki[k] * 3;
                        acc3[0] += f[0] * wk[k];
                        acc3[1] += f[1] * wk[k];
                        acc3[2] += f[2] * wk[k];
                    }
                *reinterpret_cast<float4 *>(o) = make_float4(acc3[0], acc3[1], acc3[2], any ? 1.0f : 0.0f);
            } else {
                for (int ch = 0; ch < A.C; ++ch) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
                        if (k < K && ki[k] >= 0) acc += A.feat[(size_t)ki[k] * A.C + ch] * wk[k];
                    o[ch] = acc;
                }
                o[A.C] = any ? 1.0f : 0.0f;
            }
        }
    }

    // stage the tile through LDS and let each wavefront stream one full image row per plane
    const int lds_pix = (tr * DSS_TILE + tc) * K;
    if (PLANES == 3) {
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    s_out[0][lds_pix + k] = ki[k];
                    s_out[PLANES - 1 > 0 ? 1 : 0][lds_pix + k] = __float_as_int(kz[k]);
                    s_out[PLANES - 1][lds_pix + k] = __float_as_int(kq[k]);
                }
        }
        __syncthreads();
        for (int rr = wid; rr < valid_rows; rr += FINE_WAVES) {
            const size_t rb = tile_base + (size_t)rr * S * K;
            for (int cc = lane; cc < valid_cols; cc += 64) {
                A.idx[rb + cc] = s_out[0][rr * run + cc];
                reinterpret_cast<int *>(A.zbuf)[rb + cc] = s_out[PLANES - 1 > 0 ? 1 : 0][rr * run + cc];
                reinterpret_cast<int *>(A.qv)[rb + cc] = s_out[PLANES - 1][rr * run + cc];
            }
        }
    } else {
#define DSS_STORE_PLANE(REGS, DST, CAST)                                                          \
    __syncthreads();                                                                              \
    if (owner) {                                                                                  \
        _Pragma("unroll") for (int k = 0; k < KMAX; ++k) if (k < K) s_out[0][lds_pix + k] = CAST(REGS[k]); \
    }                                                                                             \
    __syncthreads();                                                                              \
    for (int rr = wid; rr < valid_rows; rr += FINE_WAVES) {                                       \
        for (int cc = lane; cc < valid_cols; cc += 64)                                            \
            reinterpret_cast<int *>(DST)[tile_base + (size_t)rr * S * K + cc] = s_out[0][rr * run + cc]; \
    }
        DSS_STORE_PLANE(ki, A.idx, (int))
        DSS_STORE_PLANE(kz, A.zbuf, __float_as_int)
        DSS_STORE_PLANE(kq, A.qv, __float_as_int)
#undef DSS_STORE_PLANE
    }
    FT_MARK(7);
    FT_VAL(9, __builtin_amdgcn_s_memrealtime());
}

// Binned mode: grid = DSS_HEAVY_MAX + tiles.  The first workgroups (dispatched first) take the queued dense
// tiles, the others their own tile unless it is flagged as queued.  One workgroup per tile on purpose: the
// hardware dispatcher is the dynamic scheduler.  Both persistent variants were measured and are slower at
// 512^2 (36 us -> 54 us with a static snake schedule: a workgroup that draws two dense tiles ends the kernel;
// +38 us with an atomic work counter: ~4600 same-address atomics).
template <int KMAX>
__global__ __launch_bounds__(FINE_THREADS) void fine_kernel(const FineArgs A)
{
    const int total = A.N * A.g.tiles_x * A.g.tiles_y;
    const bool clean = A.clean_counts != nullptr;
    int tile_id;
    if (A.heavy.count != nullptr) {
        if (clean && blockIdx.x == 0 && threadIdx.x < DSS_HEAVY_QUEUES) A.heavy.count[threadIdx.x] = 0;  // binning-only state
        // the resets happen after a barrier: every thread of the workgroup must have read the value first
        if (blockIdx.x < DSS_HEAVY_MAX) {
            const int e = A.heavy.list[blockIdx.x];
            if (clean) {
                __syncthreads();
                if (threadIdx.x == 0) A.heavy.list[blockIdx.x] = 0;  // this workgroup is the slot's only reader
            }
            if (e == 0) return;
            tile_id = e - 1;
        } else {
            tile_id = xcd_tile(blockIdx.x - DSS_HEAVY_MAX, total);
            if (tile_id < 0) return;
            const uint8_t queued = A.heavy.flag[tile_id];
            if (clean) {
                __syncthreads();
                if (queued && threadIdx.x == 0) A.heavy.flag[tile_id] = 0;  // ... and the flag's only reader
            }
            if (queued) return;
        }
    } else {
        tile_id = xcd_tile(blockIdx.x, total);
        if (tile_id < 0) return;
    }
    fine_tile<KMAX>(A, tile_id);
}

// ---------------------------------------------------------------------------------------------
// Generic-K path for DSS_MAX_K_FAST < K <= kMaxPointsPerPixel (=150, rasterization_utils.cuh:18):
// one wavefront per 8x8 tile, one lane per pixel, the K-list lives in scratch memory (like the
// reference's thread-local `Pix q[150]`, rasterize_points.cu:177) and is kept sorted by binary
// insertion.  Rare configuration: correctness over speed.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fine_generic_kernel(const FineArgs A)
{
    const TileGrid g = A.g;
    const int tiles = g.tiles_x * g.tiles_y;
    const int tile_id = xcd_tile(blockIdx.x, A.N * tiles);
    if (tile_id < 0) return;
    const int n = tile_id / tiles;
    const int t = tile_id - n * tiles;
    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
    const int lane = threadIdx.x;
    const int r = g.row0 + ty * DSS_TILE + (lane >> 3);
    const int c = tx * DSS_TILE + (lane & 7);
    const int S = g.S, K = A.K;
    const float xf = pix_to_ndc(S - 1 - c, S);
    const float yf = pix_to_ndc(S - 1 - r, S);
    TileSource src;
    src.init(A, n, tile_id);
    const int64_t count = src.count;
    unsigned long long key[DSS_MAX_K];
    float kq[DSS_MAX_K];
    int cnt = 0;
    for (int64_t j = 0; j < count; ++j) {
        const int64_t p = src.at(j);  // wave-uniform
        const float pz = A.points[3 * p + 2];
        if (pz < 0) continue;
        const float dx = xf - A.points[3 * p];
        const float dy = yf - A.points[3 * p + 1];
        if (fabsf(dx) > A.radii[2 * p] || fabsf(dy) > A.radii[2 * p + 1]) continue;
        const float qval = A.ellipse[3 * p] * dx * dx + A.ellipse[3 * p + 1] * dx * dy + A.ellipse[3 * p + 2] * dy * dy;
        if (qval > A.cutoff[p]) continue;
        const unsigned long long ekey = ((unsigned long long)__float_as_uint(pz + 0.0f) << 32) | (unsigned long long)(unsigned)p;
        if (cnt == K && !(ekey < key[K - 1])) continue;
        int pos = (cnt < K) ? cnt : K - 1;
        while (pos > 0 && ekey < key[pos - 1]) {
            key[pos] = key[pos - 1];
            kq[pos] = kq[pos - 1];
            --pos;
        }
        key[pos] = ekey;
        kq[pos] = qval;
        if (cnt < K) ++cnt;
    }
    if (c >= S || r >= g.row0 + g.rows) return;
    const size_t pix = ((size_t)n * g.rows + (r - g.row0)) * S + c;
    A.occ[pix] = cnt > 0 ? 1.0f : 0.0f;
    const float z0 = cnt > 0 ? __uint_as_float((unsigned)(key[0] >> 32)) : 0.0f;
    bool alive = cnt > 0;
    for (int k = 0; k < K; ++k) {
        float z = -1.0f, qv = -1.0f;
        int id = -1;
        if (alive && k < cnt) {
            z = __uint_as_float((unsigned)(key[k] >> 32));
            if (z - z0 > A.thr) {
                alive = false;
                z = -1.0f;
            } else {
                id = (int)(unsigned)(key[k] & 0xffffffffull);
                qv = kq[k];
                if (A.visible) A.visible[id] = 1;
            }
        }
        A.idx[pix * K + k] = id;
        A.zbuf[pix * K + k] = z;
        A.qv[pix * K + k] = qv;
    }
}

template <int KMAX>
static void launch_fine(const FineArgs &A, int blocks, hipStream_t st)
{
    const int grid = blocks + (A.heavy.count ? DSS_HEAVY_MAX : 0);
    hipLaunchKernelGGL(fine_kernel<KMAX>, dim3(grid), dim3(FINE_THREADS), 0, st, A);
}

static bool dispatch_fine(const FineArgs &A, int blocks, hipStream_t st)
{
    const int K = A.K;
    switch (K) {
        case 1: launch_fine<1>(A, blocks, st); return true;
        case 2: launch_fine<2>(A, blocks, st); return true;
        case 3: launch_fine<3>(A, blocks, st); return true;
        case 4: launch_fine<4>(A, blocks, st); return true;
        case 5: launch_fine<5>(A, blocks, st); return true;
        case 6: launch_fine<6>(A, blocks, st); return true;
        case 7: launch_fine<7>(A, blocks, st); return true;
        case 8: launch_fine<8>(A, blocks, st); return true;
        default: break;
    }
    if (K <= 12) { launch_fine<12>(A, blocks, st); return true; }
    if (K <= 16) { launch_fine<16>(A, blocks, st); return true; }
    if (K <= 24) { launch_fine<24>(A, blocks, st); return true; }
    if (K <= 32) { launch_fine<32>(A, blocks, st); return true; }
    if (K <= DSS_MAX_K) {
        hipLaunchKernelGGL(fine_generic_kernel, dim3(blocks), dim3(64), 0, st, A);
        return true;
    }
    return false;
}

// workspace layout (binned mode): counts | lists
struct FwdWorkspace {
    uint32_t *counts;  // N*tiles*SUB
    int32_t *lists;    // N*tiles*SUB*cap
    uint32_t cap;
    HeavyQ heavy;
    size_t count_bytes;  // bytes to zero before binning: tile counters + heavy flags + queue counter
    size_t bytes;
};

// Sub-list capacity: ~32x the mean number of (splat, tile) pairs per sub-list (2 tiles per splat assumed),
// a power of two in [32, 16384].  Depends only on (N, P, S) so the size query and the launch agree.
static uint32_t bin_capacity(int N, int64_t P, int S)
{
    const double tiles = (double)((S + DSS_TILE - 1) / DSS_TILE) * ((S + DSS_TILE - 1) / DSS_TILE);
    const double mean_sub = 2.0 * ((double)P / (N > 0 ? N : 1)) / (tiles * DSS_SUB);
    uint32_t cap = 32;
    while (cap < 16384 && (double)cap < 32.0 * mean_sub) cap <<= 1;
    return cap;
}

static FwdWorkspace carve_fwd(void *ws, int N, int64_t P, int S)
{
    FwdWorkspace w;
    const size_t tiles_max = (size_t)N * ((S + DSS_TILE - 1) / DSS_TILE) * ((S + DSS_TILE - 1) / DSS_TILE);
    char *p = reinterpret_cast<char *>(ws);
    w.cap = bin_capacity(N, P, S);
    const size_t cbytes = align_up(tiles_max * DSS_SUB * 4, 256), fbytes = align_up(tiles_max, 256);
    const size_t qbytes = align_up((size_t)DSS_HEAVY_MAX * 4, 256);
    w.count_bytes = cbytes + fbytes + 256 + qbytes;
    w.counts = reinterpret_cast<uint32_t *>(p);
    w.heavy.flag = reinterpret_cast<uint8_t *>(p + cbytes);
    w.heavy.count = reinterpret_cast<uint32_t *>(p + cbytes + fbytes);
    w.heavy.list = reinterpret_cast<int32_t *>(p + cbytes + fbytes + 256);
    const size_t lists_off = w.count_bytes;
    w.lists = reinterpret_cast<int32_t *>(p + lists_off);
    w.bytes = lists_off + align_up(tiles_max * DSS_SUB * (size_t)w.cap * 4, 256);
    return w;
}

}  // namespace dss

using namespace dss;

extern "C" size_t dss_splat_forward_workspace(int N, int64_t P, int S, int K, int bin_size)
{
    (void)K;
    if (bin_size == 0 || N <= 0 || P <= 0 || S <= 0) return 256;
    return carve_fwd(nullptr, N, P, S).bytes;
}

static int validate_fwd(const char *fn, int N, int64_t P, int S, int K, int row0, int row1)
{
    if (N <= 0 || P < 0 || S <= 0 || K <= 0) {
        set_error("%s: N=%d P=%lld S=%d K=%d must be positive", fn, N, (long long)P, S, K);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (row0 < 0 || row1 > S || row0 >= row1) {
        set_error("%s: row band [%d,%d) outside image of side %d", fn, row0, row1, S);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (K > DSS_MAX_K) {
        set_error("%s: points_per_pixel %d exceeds kMaxPointsPerPixel=%d", fn, K, DSS_MAX_K);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (S > 65535 * DSS_TILE || P > 0x7ffffff0ll) {
        set_error("%s: S=%d or P=%lld too large", fn, S, (long long)P);
        return DSS_ERR_UNSUPPORTED;
    }
    return DSS_OK;
}

static TileGrid make_grid(int S, int row0, int row1)
{
    TileGrid g;
    g.S = S;
    g.row0 = row0;
    g.rows = row1 - row0;
    g.tiles_x = (S + DSS_TILE - 1) / DSS_TILE;
    g.tiles_y = (g.rows + DSS_TILE - 1) / DSS_TILE;
    return g;
}

static int splat_bin_impl(const float *points, const float *radii, const int64_t *first_idx,
                          const int64_t *num_pts, int N, int64_t P, int S, int row0, int row1,
                          void *workspace, size_t workspace_bytes, uint8_t *visible_to_clear, void *stream)
{
    int rc = validate_fwd("dss_splat_bin", N, P, S, 1, row0, row1);
    if (rc) return rc;
    if (P == 0) return DSS_OK;
    if (!points || !radii || !first_idx || !num_pts) {
        set_error("dss_splat_bin: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t need = dss_splat_forward_workspace(N, P, S, 1, 1);
    if (!workspace || workspace_bytes < need) {
        set_error("dss_splat_bin: workspace %zu bytes < required %zu", workspace_bytes, need);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const TileGrid g = make_grid(S, row0, row1);
    const int tiles = g.tiles_x * g.tiles_y;
    if ((long long)N * tiles > 0x7fffffffll) { set_error("dss_splat_bin: too many tiles"); return DSS_ERR_UNSUPPORTED; }
    FwdWorkspace w = carve_fwd(workspace, N, P, S);
    if (hipMemsetAsync(w.counts, 0, w.count_bytes, st) != hipSuccess) return check_launch("memset tile counts");
    const int pb = (int)((P + 255) / 256);
    hipLaunchKernelGGL(bin_kernel, dim3(pb), dim3(256), 0, st, points, radii, first_idx, num_pts, N, P, g, w.counts,
                       w.lists, w.cap, w.heavy, visible_to_clear);
    return check_launch("dss_splat_bin");
}

extern "C" int dss_splat_bin(const float *points, const float *radii, const int64_t *first_idx,
                             const int64_t *num_pts, int N, int64_t P, int S, int row0, int row1,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    return splat_bin_impl(points, radii, first_idx, num_pts, N, P, S, row0, row1, workspace, workspace_bytes, nullptr,
                          stream);
}

static int splat_fine_impl(const float *points, const float *ellipse, const float *cutoff, const float *radii,
                           const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, float merge_thr,
                           int S, int K, int row0, int row1, int32_t *idx, float *zbuf, float *qvalue, float *occ,
                           uint8_t *visible, const float *scaler, const float *feat, int C, float *image, float *wsum,
                           const void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = validate_fwd("dss_splat_fine", N, P, S, K, row0, row1);
    if (rc) return rc;
    if (!idx || !zbuf || !qvalue || !occ || !first_idx || !num_pts ||
        (P > 0 && (!points || !ellipse || !cutoff || !radii))) {
        set_error("dss_splat_fine: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const TileGrid g = make_grid(S, row0, row1);
    const long long blocks_ll = (long long)N * g.tiles_x * g.tiles_y;
    if (blocks_ll > 0x7fffffffll) { set_error("dss_splat_fine: too many tiles"); return DSS_ERR_UNSUPPORTED; }
    FineArgs A;
    A.points = points; A.ellipse = ellipse; A.cutoff = cutoff; A.radii = radii;
    A.first_idx = first_idx; A.num_pts = num_pts;
    A.counts = nullptr; A.lists = nullptr; A.cap = 0;
    A.heavy.count = nullptr; A.heavy.list = nullptr; A.heavy.flag = nullptr; A.clean_counts = nullptr;
    A.idx = idx; A.zbuf = zbuf; A.qv = qvalue; A.occ = occ; A.visible = visible;
    A.g = g; A.N = N; A.K = K; A.thr = merge_thr;
    A.scaler = scaler; A.feat = feat; A.image = image; A.wsum = wsum; A.C = C;
    A.img_sn = (long long)g.rows * S * (C + 1); A.img_sr = (long long)S * (C + 1);
    if (workspace && P > 0) {
        if (workspace_bytes < dss_splat_forward_workspace(N, P, S, K, 1)) {
            set_error("dss_splat_fine: workspace too small");
            return DSS_ERR_WORKSPACE;
        }
        FwdWorkspace w = carve_fwd(const_cast<void *>(workspace), N, P, S);
        A.counts = w.counts; A.lists = w.lists; A.cap = w.cap; A.heavy = w.heavy;
    }
    if (!dispatch_fine(A, (int)blocks_ll, as_stream(stream))) {
        set_error("dss_splat_fine: no kernel for K=%d", K);
        return DSS_ERR_UNSUPPORTED;
    }
    return check_launch("dss_splat_fine");
}

extern "C" int dss_splat_fine(const float *points, const float *ellipse, const float *cutoff, const float *radii,
                              const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, float merge_thr,
                              int S, int K, int row0, int row1, int32_t *idx, float *zbuf, float *qvalue, float *occ,
                              uint8_t *visible, const void *workspace, size_t workspace_bytes, void *stream)
{
    return splat_fine_impl(points, ellipse, cutoff, radii, first_idx, num_pts, N, P, merge_thr, S, K, row0, row1, idx,
                           zbuf, qvalue, occ, visible, nullptr, nullptr, 0, nullptr, nullptr, workspace, workspace_bytes,
                           stream);
}

extern "C" int dss_splat_fine_blend(const float *points, const float *ellipse, const float *cutoff, const float *radii,
                                    const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P,
                                    float merge_thr, int S, int K, int row0, int row1, int32_t *idx, float *zbuf,
                                    float *qvalue, float *occ, uint8_t *visible, const float *scaler, const float *feat,
                                    int C, float *image, float *wsum, const void *workspace, size_t workspace_bytes,
                                    void *stream)
{
    if (K > DSS_MAX_K_FAST || C < 1 || C > 8 || !scaler || !feat || !image || !wsum) {
        set_error("dss_splat_fine_blend: needs K <= %d, 1 <= C <= 8 and non-NULL blend tensors", DSS_MAX_K_FAST);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    return splat_fine_impl(points, ellipse, cutoff, radii, first_idx, num_pts, N, P, merge_thr, S, K, row0, row1, idx,
                           zbuf, qvalue, occ, visible, scaler, feat, C, image, wsum, workspace, workspace_bytes, stream);
}

extern "C" int dss_splat_forward(const float *points, const float *ellipse, const float *cutoff,
                                 const float *radii, const int64_t *first_idx, const int64_t *num_pts,
                                 int N, int64_t P, float merge_thr, int S, int K, int bin_size,
                                 int row0, int row1, int32_t *idx, float *zbuf, float *qvalue, float *occ,
                                 uint8_t *visible, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = validate_fwd("dss_splat_forward", N, P, S, K, row0, row1);
    if (rc) return rc;
    const bool binned = (bin_size != 0 && P > 0);
    if (binned) {
        // the binning pass also clears `visible` (one launch fewer than a separate memset)
        rc = splat_bin_impl(points, radii, first_idx, num_pts, N, P, S, row0, row1, workspace, workspace_bytes, visible,
                            stream);
        if (rc) return rc;
    } else if (visible && P > 0) {
        if (hipMemsetAsync(visible, 0, (size_t)P, as_stream(stream)) != hipSuccess) return check_launch("memset visible");
    }
    return dss_splat_fine(points, ellipse, cutoff, radii, first_idx, num_pts, N, P, merge_thr, S, K, row0, row1, idx,
                          zbuf, qvalue, occ, visible, binned ? workspace : nullptr, workspace_bytes, stream);
}

#ifdef DSS_FINE_TIMING
extern "C" __attribute__((visibility("default"))) int dss_debug_set_fine_timing(long long *buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(dss::g_fine_timing), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

// ---------------------------------------------------------------------------------------------
// Fused single-call forward: [setup + binning] -> [fine + blend]: two kernel launches plus the counter
// memset, and neither the screen records nor the fragment lists are re-read by a separate pass.
// ---------------------------------------------------------------------------------------------
extern "C" size_t dss_render_forward_workspace(int N, int64_t P, int S, int K)
{
    return dss_splat_forward_workspace(N, P, S, K, 1);
}

extern "C" int dss_render_forward(const float *world, const float *normals, const float *h_point, const float *h_cloud,
                                  const float *vr6, const float *frame_normals, const float *M, const float *V, const float *znear, const float *zfar,
                                  const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int shared_cloud,
                                  int backface_culling, int S, int K, float cutoff_threshold, float antialiasing_sigma,
                                  float merge_thr, int row0, int row1, const float *feat, int C,
                                  float *pts_screen, float *ellipse, float *radii, float *scaler, float *cutoff,
                                  uint8_t *valid, int32_t *idx, float *zbuf, float *qvalue, float *occ,
                                  uint8_t *visible, float *image, int64_t image_cam_stride, int64_t image_row_stride,
                                  float *wsum, void *workspace, size_t workspace_bytes, int workspace_state,
                                  void *stream)
{
    int rc = validate_fwd("dss_render_forward", N, P, S, K, row0, row1);
    if (rc) return rc;
    if (K > DSS_MAX_K_FAST) {
        set_error("dss_render_forward: fused path needs points_per_pixel <= %d (use the separate entry points)",
                  DSS_MAX_K_FAST);
        return DSS_ERR_UNSUPPORTED;
    }
    if (P == 0 || C < 1 || C > 8) {
        set_error("dss_render_forward: needs P > 0 and 1 <= C <= 8 (P=%lld C=%d)", (long long)P, C);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (!world || !normals || (!h_point && !h_cloud && !vr6) || (vr6 && !frame_normals) || !M || !V || !znear || !zfar ||
        !first_idx || !num_pts || !feat ||
        !pts_screen || !ellipse || !radii || !scaler || !cutoff || !valid || !idx || !zbuf || !qvalue || !occ ||
        !visible || !image || !wsum) {
        set_error("dss_render_forward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t need = dss_splat_forward_workspace(N, P, S, K, 1);
    if (!workspace || workspace_bytes < need) {
        set_error("dss_render_forward: workspace %zu bytes < required %zu", workspace_bytes, need);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const TileGrid g = make_grid(S, row0, row1);
    const int tiles = g.tiles_x * g.tiles_y;
    if ((long long)N * tiles > 0x7fffffffll) { set_error("dss_render_forward: too many tiles"); return DSS_ERR_UNSUPPORTED; }
    FwdWorkspace w = carve_fwd(workspace, N, P, S);
    const bool clean = workspace_state == DSS_WS_CLEAN;
    if (!clean && hipMemsetAsync(w.counts, 0, w.count_bytes, st) != hipSuccess) return check_launch("memset tile counts");
    SetupArgs SA;
    SA.world = world; SA.normals = normals; SA.h_point = h_point; SA.h_cloud = h_cloud; SA.M = M; SA.V = V;
    SA.vr6 = vr6; SA.frame_n = frame_normals;
    SA.znear = znear; SA.zfar = zfar; SA.first_idx = first_idx; SA.num_pts = num_pts; SA.N = N; SA.P = P;
    SA.shared = shared_cloud; SA.backface = backface_culling; SA.S = S; SA.cutoffC = cutoff_threshold;
    SA.sigma = antialiasing_sigma; SA.screen = pts_screen; SA.ellipse = ellipse; SA.radii = radii; SA.scaler = scaler;
    SA.cutoff = cutoff; SA.valid = valid;
    // one wave per workgroup: at DSS sizes (tens of thousands of points) 256-thread groups would occupy only
    // half of the CUs with one wave per SIMD, and this kernel is a chain of dependent latencies
    const int pb = (int)((P + 63) / 64);
    hipLaunchKernelGGL(setup_bin_kernel, dim3(pb), dim3(64), 0, st, SA, g, w.counts, w.lists, w.cap, w.heavy, visible);
    FineArgs A;
    A.points = pts_screen; A.ellipse = ellipse; A.cutoff = cutoff; A.radii = radii;
    A.first_idx = first_idx; A.num_pts = num_pts;
    A.counts = w.counts; A.lists = w.lists; A.cap = w.cap; A.heavy = w.heavy;
    A.clean_counts = clean ? w.counts : nullptr;
    A.idx = idx; A.zbuf = zbuf; A.qv = qvalue; A.occ = occ; A.visible = visible;
    A.g = g; A.N = N; A.K = K; A.thr = merge_thr;
    A.scaler = scaler; A.feat = feat; A.image = image; A.wsum = wsum; A.C = C;
    A.img_sn = image_cam_stride > 0 ? image_cam_stride : (long long)g.rows * S * (C + 1);
    A.img_sr = image_row_stride > 0 ? image_row_stride : (long long)S * (C + 1);
    if (!dispatch_fine(A, N * tiles, st)) { set_error("dss_render_forward: no kernel for K=%d", K); return DSS_ERR_UNSUPPORTED; }
    return check_launch("dss_render_forward");
}
