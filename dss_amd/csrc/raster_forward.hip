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
#ifdef DSS_FINE_TIMING
#include <hip/hip_runtime.h>
namespace dss { extern __device__ long long *g_fine_timing; }
#define DSS_SETUP_MARK(slot)                                                                                                  \
    do {                                                                                                                      \
        if (dss::g_fine_timing && threadIdx.x == 0)                                                                           \
            dss::g_fine_timing[((size_t)blockIdx.x + 8192) * 12 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime();       \
    } while (0)
#endif
#include "setup_body.h"
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <vector>

namespace dss {

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
#define FT_DECL(var) long long var = 0
#define FT_ACC(var, v) var += (v)
// the binning launch stamps into rows 8192.. of the same buffer (tools/setup_timing.py; the compiler may move arithmetic
// across a stamp: read them as "issued by", not "finished by")
#define FT_MARK_S(slot)                                                                 \
    do {                                                                                \
        if (g_fine_timing && threadIdx.x == 0)                                          \
            g_fine_timing[((size_t)blockIdx.x + 8192) * 12 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define FT_MARK_S(slot)
#define FT_MARK(slot)
#define FT_VAL(slot, v)
#define FT_DECL(var)
#define FT_ACC(var, v)
#endif

// Each tile owns DSS_SUB counters / sub-lists, selected by the low bits of the splat id.
// Same-address global atomics serialise (~50 ns each on MI355X: 523 splats on the hottest 16x16 tile of
// the bunny scene cost ~30 us); splitting cuts the depth of every hot address by DSS_SUB.
// Sub-lists have a fixed capacity `cap` >= 32 (workspace layout: counts (N*tiles*SUB) uint32, ..., lists
// (N*tiles*SUB*cap) int32), see bin_capacity; a count above `cap` marks the tile as overflowed.
#ifndef DSS_SUB     // (-DDSS_SUB=16 -DSPEC=16: development A/B build, profiles/r6_a_setup_bin_ab.txt)
#define DSS_SUB 8
#endif

// Queue of OCCUPIED tiles.  The thread whose append is the first of a sub-list (returned position 0) claims the
// tile with one atomicExch on its flag word; the winner appends the tile to one of DSS_QUEUES queues.  The fine
// kernel then launches workgroups for queue slots (occupied tiles only) plus a few fat workgroups that stream the
// fill values of the EMPTY tiles, 16 tiles each: round 1 launched one 256-thread workgroup per tile + 2048 queue
// workgroups, and tools/fine_timing.py showed the launch bound by workgroup SLOT turnover (6144 workgroups through
// 2048 resident slots, 3000 of them only to write 5 KB of fill values after one dependent load: median start time
// 14 us into a 27 us kernel).
// Queue of a tile = XCD of its 32x32-pixel super-block + 8 * (parity of the tile
// coordinates): workgroup b serves queue b % 32 and is placed on XCD b % 8 by the dispatcher (observed, used for
// speed only), so the tiles of one super-block -- which share most of their splat records -- run on ONE XCD's L2:
// round-robin tiles made every XCD fetch the whole record set (FETCH 7.5x the unique bytes, r1 profiles), an
// XCD-contiguous split put the dense screen region on one XCD (slower, round 1).
#define DSS_QUEUES 32
#define DSS_SB_SHIFT 2   // super-block = 4x4 tiles (32x32 pixels)
struct TileQueue {
    uint32_t *tail;   // DSS_QUEUES append counters (binning only; zero when binning starts)
    int32_t *list;    // DSS_QUEUES x capq slots holding tile id + 1, 0 = empty (zero when binning starts)
    uint32_t *flag;   // one word per tile: 1 = occupied and queued (zero when binning starts)
    uint32_t capq;    // slots per queue
};

// Spill path of the fixed-capacity sub-lists.  A splat that finds a sub-list full (returned position >= cap) keeps the
// count going and records WHICH of its (up to 2 x 2) tiles were full in one mask byte of its own.  After the binning
// launch a second (always launched, normally empty: it exits on one scalar load) launch revisits the marked splats,
// recomputes their tile rectangle and moves the dropped entries into a shared pool, contiguous per sub-list: the first
// thread to arrive at a sub-list reserves count - cap pool entries (the count is final by then) and publishes the offset,
// the others pick it up.  The fine pass reads slots < cap of a sub-list from its primary list and the rest from the
// pool.  Round 1 rasterized a tile with an overflowed sub-list from its WHOLE cloud (exact but O(64 P) per tile: a far
// camera put every occupied tile over capacity) and sized the capacity at 32x the mean load to make that rare (1 GB of
// lists at 4M points); with the spill path the capacity is 4x the mean load.  (A log of (sub-list, id) pairs instead of
// the mask was tried first: 8 more bytes per point, and its append counter(s) serialised -- 0.9 ms for 83k entries.)
// A splat larger than 2 x 2 tiles (radius above 8 pixels) records its full tiles in a 64-bit mask of its own -- bit 8 dy + dx
// of `big[p]`, valid when bit 7 of its mask byte is set -- for rectangles of up to 8 x 8 tiles; a wider one (radius above 28
// pixels: a point close to the camera) appends one such mask per 8 x 8 block of its rectangle to the `giant` records.  Only
// when those run out does a splat raise the flag that sends every spilled tile back to whole-cloud scans.  (Until round 6
// every splat wider than 2 x 2 tiles raised it, "rare" being the assumption: the reference's own training loop at configs[2]
// lives there -- h sits at its upper clamp 1e-3, splats are ~10 pixels wide, hundreds overlap per pixel -- and ~1,900 tiles
// per call scanned all 99,790 points of their cloud: 1.3 of the fine pass's 1.56 ms, tools/fine_timing.py trained.)
// The cell-ordered binning of more than 2M splats (bin_sorted_kernel) records them the same way.
struct Spill {
    uint32_t *cursor;   // (N*tiles*SUB) arrival counters of the pool pass                  (zero when binning starts)
    uint32_t *offset;   // (N*tiles*SUB) 1 + first pool entry of an overflowed sub-list      (zero when binning starts)
    uint8_t *mask;      // (P) bit 0..3: tile (tx0,ty0), (tx1,ty0), (tx0,ty1), (tx1,ty1) was full (zero when binning starts)
    uint32_t *ctrl;     // [0] some mask is set, [1] pool entries handed out                 (zero when binning starts)
    uint32_t *fail;     // fail[0] == fail[1]: a wide splat found no record for its full tiles / a waiter gave up -> spilled tiles fall back to
                        // whole-cloud scans.  fail[1] = the epoch of the binning launch (written by its first thread),
                        // fail[0] = that epoch, written by whoever fails.  Never reset: every racing writer of a word
                        // stores the same value (two different values from two XCDs would leave the outcome to the
                        // write-back order of their L2s), and a stale or uninitialised fail[0] can only equal the new
                        // epoch by a 2^-32 accident, which costs speed, not exactness.  Outside the zero region.
    int32_t *pool;
    unsigned long long *big;  // (P) full tiles of a splat larger than 2 x 2 tiles (valid when bit 7 of its mask byte is set)
    uint4 *giant;             // (8, giant_cap) records of splats wider than 8 x 8 tiles, one per 8 x 8 block of their rectangle that
                              // met a full sub-list: {splat (or its position), sub-list << 28 | block origin ty << 14 | tx, mask lo, mask hi}; segment = entry mod 8,
                              // its record count in ctrl[2 + segment] (eight counters: one serialises at ~11 ns per append)
    uint32_t giant_cap;       // records per segment
    uint32_t cap_entries;  // pool capacity in entries
    uint32_t epoch;        // unique per binning launch of this process (host counter)
};

// first thread of a binning launch: the launch's epoch, and -- for a pool pass that runs INSIDE the fine launch -- the value
// of the pass's completion counter before this call (fail[2] counts finished pool-pass workgroups and is never reset;
// fail[3] is what it was when the call started: the fine workgroups of spilled tiles wait for fail[2] - fail[3] to reach the
// number of pool-pass workgroups.  Both live behind the zero region like fail[0..1]: any initial value works.)
__device__ __forceinline__ void spill_begin(const Spill sp)
{
    sp.fail[1] = sp.epoch;
    sp.fail[3] = sp.fail[2];
}

struct TileGrid {
    int S;        // image side
    int row0;     // first image row of the band
    int rows;     // rows in the band (band-local rows: the band tensors have this many rows)
    int tiles_x;  // tiles per band row
    int tiles_y;  // tile rows in the band
    int tshift;   // log2 of the image-row distance of two consecutive band tile rows: 3 for a contiguous band; 3 + log2(c)
                  // for a tile-row-CYCLIC band (multi-GPU: a rank owns every c-th 8-row tile row starting at row0, so that
                  // every rank gets the same mix of dense and empty screen regions).  Band-local row l <-> image row
                  // row0 + ((l >> 3) << tshift) + (l & 7).
};
// first image row of band tile row ty
__host__ __device__ __forceinline__ int tile_row0(const TileGrid &g, int ty) { return g.row0 + (ty << g.tshift); }

// Super-block (i, j) of camera n goes to XCD (i + 3 j + 5 n) mod 8: neighbours in a row differ by 1, in a column by 3, so
// any compact screen region is spread evenly over the XCDs (the first version dealt the 64x64-pixel super-blocks of a
// 512^2 image by COLUMN: a centred object then ran on three of the eight XCDs).
__host__ __device__ __forceinline__ int sb_cols(const TileGrid &g) { return (g.tiles_x + (1 << DSS_SB_SHIFT) - 1) >> DSS_SB_SHIFT; }
__host__ __device__ __forceinline__ int sb_rows(const TileGrid &g) { return (g.tiles_y + (1 << DSS_SB_SHIFT) - 1) >> DSS_SB_SHIFT; }
// slots per queue: in every super-block row each XCD gets at most ceil(cols / 8) super-blocks, and a (super-block,
// parity) pair holds at most (4 x 4) / 4 tiles
static inline uint32_t queue_capacity(int N, const TileGrid &g)
{
    const long long per_cam = (long long)((sb_cols(g) + 7) / 8) * sb_rows(g);
    return (uint32_t)(per_cam * N * ((1 << DSS_SB_SHIFT) * (1 << DSS_SB_SHIFT) / 4));
}
__device__ __forceinline__ int queue_of(int n, int tx, int ty, const TileGrid &g)
{
    const int xcd = ((tx >> DSS_SB_SHIFT) + 3 * (ty >> DSS_SB_SHIFT) + 5 * n) & 7;
    return xcd + 8 * (((ty & 1) << 1) | (tx & 1));
}

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
    // band tile rows whose 8 image rows [R, R + 7], R = row0 + (ty << tshift), meet [r0, r1] (contiguous band, tshift = 3:
    // ty = (r - row0) / 8 as before; the last tile row of a band may be short: rows beyond g.rows are never stored)
    r0 = max(r0, g.row0);
    r1 = min(r1, tile_row0(g, g.tiles_y - 1) + (g.rows - 1 - (g.tiles_y - 1) * DSS_TILE));   // last image row of the band
    if (r0 > r1) return false;
    tx0 = c0 / DSS_TILE;
    tx1 = c1 / DSS_TILE;
    const int step = 1 << g.tshift;
    ty0 = (r0 - g.row0 - (DSS_TILE - 1) + step - 1) >> g.tshift;   // ceil((r0 - row0 - 7) / step), numerator + step - 1 >= 0
    ty0 = max(ty0, 0);
    ty1 = min((r1 - g.row0) >> g.tshift, g.tiles_y - 1);
    return ty0 <= ty1;
}

// First entry of one of the tile's sub-lists: claim the tile (at most DSS_SUB threads per tile get here, exactly one
// wins the exchange) and append it to its queue.
__device__ __forceinline__ void claim_tile(const TileQueue tq, int n, int tx, int ty, const TileGrid g)
{
    const int tile = (n * g.tiles_y + ty) * g.tiles_x + tx;
    if (atomicExch(&tq.flag[tile], 1u) != 0u) return;
    const int q = queue_of(n, tx, ty, g);
    const uint32_t pos = atomicAdd(&tq.tail[q], 1u);
    if (pos < tq.capq) tq.list[(size_t)q * tq.capq + pos] = tile + 1;  // (always true: queue_capacity)
}

// append splat p to the sub-list (p mod SUB) of every tile of its (non-empty) tile rectangle [tx0, tx1] x [ty0, ty1]
__device__ __forceinline__ void bin_rect(int64_t p, int n, int tx0, int tx1, int ty0, int ty1,
                                         const TileGrid g, uint32_t *__restrict__ counts,
                                         int32_t *__restrict__ lists, uint32_t cap, const TileQueue tq,
                                         const Spill sp)
{
#ifdef DSS_FINE_TIMING
    asm volatile("" ::"v"(tx0), "v"(tx1), "v"(ty0), "v"(ty1));   // (the tile rectangle is known)
    FT_MARK_S(8);
#endif
    const size_t sub0 = ((size_t)n * g.tiles_x * g.tiles_y) * DSS_SUB + ((unsigned)p & (DSS_SUB - 1));
    if (tx1 - tx0 <= 1 && ty1 - ty0 <= 1) {
        // common case (splat overlaps at most 2x2 tiles): the returning atomics are independent, issue
        // them back to back so their latencies overlap instead of chaining
        const size_t t00 = sub0 + (size_t)(ty0 * g.tiles_x + tx0) * DSS_SUB;
        const size_t t01 = t00 + DSS_SUB, t10 = t00 + (size_t)g.tiles_x * DSS_SUB, t11 = t10 + DSS_SUB;
        const bool hx = tx1 > tx0, hy = ty1 > ty0;
        uint32_t p0, p1 = 1, p2 = 1, p3 = 1;
#ifdef DSS_FINE_TIMING
        asm volatile("" ::"v"(t00), "s"(cap));   // (addresses and the list capacity -- a late kernel argument -- are there)
        FT_MARK_S(9);
#endif
        p0 = atomicAdd(&counts[t00], 1u);
        if (hx) p1 = atomicAdd(&counts[t01], 1u);
        if (hy) p2 = atomicAdd(&counts[t10], 1u);
        if (hx && hy) p3 = atomicAdd(&counts[t11], 1u);
        FT_MARK_S(3);
        if (p0 < cap) lists[t00 * cap + p0] = (int32_t)p;
#ifdef DSS_FINE_TIMING
        if (g_fine_timing && threadIdx.x == 0 && (p0 | p1 | p2 | p3) != 0xffffffffu) FT_MARK_S(4);   // (atomics returned)
#endif
        if (hx && p1 < cap) lists[t01 * cap + p1] = (int32_t)p;
        if (hy && p2 < cap) lists[t10 * cap + p2] = (int32_t)p;
        if (hx && hy && p3 < cap) lists[t11 * cap + p3] = (int32_t)p;
        if (sp.ctrl) {
            const unsigned full = (p0 >= cap ? 1u : 0u) | ((hx && p1 >= cap) ? 2u : 0u) | ((hy && p2 >= cap) ? 4u : 0u) |
                                  ((hx && hy && p3 >= cap) ? 8u : 0u);
            if (full) {
                sp.mask[p] = (uint8_t)full;
                sp.ctrl[0] = 1u;
            }
        }
        if (tq.flag) {
            if (p0 == 0) claim_tile(tq, n, tx0, ty0, g);
            if (p1 == 0) claim_tile(tq, n, tx1, ty0, g);
            if (p2 == 0) claim_tile(tq, n, tx0, ty1, g);
            if (p3 == 0) claim_tile(tq, n, tx1, ty1, g);
        }
        return;
    }
    const bool one_block = tx1 - tx0 < 8 && ty1 - ty0 < 8;
    for (int by = ty0; by <= ty1; by += 8)
        for (int bx = tx0; bx <= tx1; bx += 8) {   // (one trip unless the rectangle is wider than 8 x 8 tiles)
            unsigned long long full = 0ull;
            const int ye = min(by + 7, ty1), xe = min(bx + 7, tx1);
            for (int ty = by; ty <= ye; ++ty)
                for (int tx = bx; tx <= xe; ++tx) {
                    const size_t t = sub0 + (size_t)(ty * g.tiles_x + tx) * DSS_SUB;
                    const uint32_t pos = atomicAdd(&counts[t], 1u);
                    if (pos < cap) lists[t * cap + pos] = (int32_t)p;
                    else full |= 1ull << (8 * (ty - by) + (tx - bx));
                    if (tq.flag && pos == 0) claim_tile(tq, n, tx, ty, g);
                }
            if (full == 0ull || !sp.ctrl) continue;
            if (one_block && sp.big != nullptr) {
                sp.big[p] = full;
                sp.mask[p] = (uint8_t)0x80u;
            } else if (sp.giant != nullptr) {
                const unsigned seg = (unsigned)p & 7u;
                const uint32_t r = atomicAdd(&sp.ctrl[2 + seg], 1u);
                if (r < sp.giant_cap)
                    sp.giant[(size_t)seg * sp.giant_cap + r] = make_uint4((uint32_t)p, (seg << 28) | ((uint32_t)by << 14) | (uint32_t)bx,
                                                                          (uint32_t)full, (uint32_t)(full >> 32));
                else sp.fail[0] = sp.epoch;   // (out of records)
            } else {
                sp.fail[0] = sp.epoch;        // (no mask for it)
            }
            sp.ctrl[0] = 1u;
        }
}

// append splat p to the sub-list (p mod SUB) of every tile of its rectangle
__device__ __forceinline__ void bin_point(int64_t p, int n, float px, float py, float pz, float rx, float ry,
                                          const TileGrid g, uint32_t *__restrict__ counts,
                                          int32_t *__restrict__ lists, uint32_t cap, const TileQueue tq,
                                          const Spill sp)
{
    if (n < 0) return;
    int tx0, tx1, ty0, ty1;
    if (!splat_tile_rect(px, py, pz, rx, ry, g, tx0, tx1, ty0, ty1)) return;
    bin_rect(p, n, tx0, tx1, ty0, ty1, g, counts, lists, cap, tq, sp);
}

__global__ __launch_bounds__(256) void bin_kernel(
    const float *__restrict__ points, const float *__restrict__ radii,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, int64_t P,
    TileGrid g, uint32_t *__restrict__ counts, int32_t *__restrict__ lists, uint32_t cap, TileQueue tq,
    Spill sp, uint8_t *__restrict__ visible_to_clear /* (P) or nullptr */)
{
    if (sp.ctrl && blockIdx.x == 0 && threadIdx.x == 0) spill_begin(sp);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (visible_to_clear) visible_to_clear[p] = 0;  // saves a separate memset launch
    const int n = find_cloud(p, first_idx, num_pts, N);
    bin_point(p, n, points[3 * p], points[3 * p + 1], points[3 * p + 2], radii[2 * p], radii[2 * p + 1], g, counts,
              lists, cap, tq, sp);
}

// dss_render_forward: per-point setup (culling + projection + EWA terms) fused with the binning --
// the screen record goes from registers straight into the tile lists.
// BAND_ONLY (DSS_WS_BAND_OUTPUTS, multi-GPU): a splat whose tile rectangle misses the rank's band writes its screen position,
// radii and validity only (see setup_point_store); a kernel of its own so that the whole-image launch keeps its schedule.
template <bool BAND_ONLY>
__global__ __launch_bounds__(256) void setup_bin_kernel(const SetupArgs A, TileGrid g, uint32_t *__restrict__ counts,
                                                        int32_t *__restrict__ lists, uint32_t cap, TileQueue tq,
                                                        Spill sp, uint8_t *__restrict__ visible_to_clear)
{
    if (sp.ctrl && blockIdx.x == 0 && threadIdx.x == 0) spill_begin(sp);
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= A.P) return;
    if (visible_to_clear) visible_to_clear[p] = 0;
    FT_MARK_S(0);
    const int n = find_cloud(p, A.first_idx, A.num_pts, A.N);
    FT_MARK_S(1);
    float px, py, pz, rx, ry;
    if (BAND_ONLY) {
        const SetupVals v = setup_point_compute(A, p, n);
        int tx0, tx1, ty0, ty1;
        const bool reach = n >= 0 && splat_tile_rect(v.sx, v.sy, v.sz, v.rx, v.ry, g, tx0, tx1, ty0, ty1);
        setup_point_store(A, p, v, reach);
        FT_MARK_S(2);
        if (reach) bin_rect(p, n, tx0, tx1, ty0, ty1, g, counts, lists, cap, tq, sp);   // (the rectangle is computed once)
        FT_MARK_S(5);
        return;
    } else {
#ifdef DSS_FINE_TIMING
        const SetupVals v = setup_point_compute(A, p, n);
        asm volatile("" ::"v"(v.rx), "v"(v.ry), "v"(v.sc), "v"(v.ea));   // (the arithmetic has finished)
        FT_MARK_S(6);
        setup_point_store(A, p, v);
        px = v.sx; py = v.sy; pz = v.sz; rx = v.rx; ry = v.ry;
#else
        setup_point(A, p, n, px, py, pz, rx, ry);
#endif
    }
    FT_MARK_S(2);
    bin_point(p, n, px, py, pz, rx, ry, g, counts, lists, cap, tq, sp);
    FT_MARK_S(5);
}

// Pool pass (see struct Spill): workgroup `block` of `nblocks`, 256 threads, grid-stride over the splats.  Unless some splat
// was marked every workgroup returns after one scalar load.  Runs either as the first workgroups of the fine launch
// (dss_render_forward: fine_kernel, FineArgs::spill_wgs) or as a launch of its own behind the binning (dss_splat_bin).
// Returns false when nothing was marked.
__device__ __forceinline__ bool spill_pass(
    const float *__restrict__ points, const float *__restrict__ radii, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, int N, int64_t P, const TileGrid g, const uint32_t *__restrict__ counts, uint32_t cap,
    const Spill sp, int sorted /* 1: the lists were filled by bin_sorted_kernel (sub-list in bits 4..6 of the mask byte) */,
    unsigned block, unsigned nblocks,
    const float4 *__restrict__ crec = nullptr /* sorted path: mask bytes, list and pool entries are POSITIONS of the order; */,
    const int32_t *__restrict__ s_id = nullptr /* the splat's geometry comes from its candidate record, its id from the order */)
{
    if (__builtin_amdgcn_readfirstlane((int)sp.ctrl[0]) == 0) return false;
    // The mask bytes are scanned SIXTEEN per load (the array is 256-byte aligned and padded to 256 bytes with zeros).  One
    // byte per thread and trip was a chain of P / (workgroups * 256) dependent loads -- 61 trips, ~120 us, at 8M splats on
    // 512 workgroups: invisible inside the 0.75 ms fine launch of the whole image, but the tiles that wait for the pool
    // (the sphere's limb) ended the launch, and on a row band of the multi-GPU step the pass was half of the fine launch.
    // one dropped (splat, sub-list) entry -> its place in the pool
    auto pool_put = [&](const size_t t, const int64_t p) __attribute__((always_inline)) {
        const uint32_t extra = counts[t] - cap;   // entries of this sub-list that live in the pool (final)
        const uint32_t pos = atomicAdd(&sp.cursor[t], 1u);
        uint32_t off1 = 0;
        // the publisher's branch comes first in program order: a waiter of the same wavefront never spins ahead of it
        if (pos == 0) {
            off1 = atomicAdd(&sp.ctrl[1], extra) + 1u;
            __hip_atomic_store(&sp.offset[t], off1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (pos != 0) {
            // the thread that drew position 0 has executed its atomic before ours and publishes without waiting for
            // anyone; the offset is its own tag (non-zero once written)
            for (int spin = 0; spin < (1 << 22) && off1 == 0; ++spin) {
                off1 = __hip_atomic_load(&sp.offset[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (off1 == 0) __builtin_amdgcn_s_sleep(2);
            }
            if (off1 == 0) sp.fail[0] = sp.epoch;  // gave up: the fine pass falls back to whole-cloud scans
        }
        if (off1 != 0 && (unsigned long long)(off1 - 1u) + pos < sp.cap_entries) sp.pool[(size_t)(off1 - 1u) + pos] = (int32_t)p;
    };
    const int64_t nvec = (P + 15) / 16;
    for (int64_t v = (int64_t)block * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)nblocks * blockDim.x) {
    const uint4 m4 = reinterpret_cast<const uint4 *>(sp.mask)[v];
    if ((m4.x | m4.y | m4.z | m4.w) == 0u) continue;
#pragma unroll 1
    for (int mb = 0; mb < 16; ++mb) {
    const uint32_t mword = mb < 8 ? (mb < 4 ? m4.x : m4.y) : (mb < 12 ? m4.z : m4.w);   // (selects: no indexed register array)
    const unsigned full = (mword >> (8 * (mb & 3))) & 0xffu;
    const int64_t p = 16 * v + mb;
    if (full == 0 || p >= P) continue;
    sp.mask[p] = 0;  // (this thread is the byte's only reader: the DSS_WS_CLEAN state is restored here)
    const int n = find_cloud(crec ? (int64_t)s_id[p] : p, first_idx, num_pts, N);
    int tx0, tx1, ty0, ty1;
    float gx, gy, gz, grx, gry;
    if (crec) {
        const float4 c0 = crec[2 * (size_t)p], c1 = crec[2 * (size_t)p + 1];
        gx = c0.x; gy = c0.y; gz = c1.w; grx = c0.z; gry = c0.w;
    } else {
        gx = points[3 * p]; gy = points[3 * p + 1]; gz = points[3 * p + 2]; grx = radii[2 * p]; gry = radii[2 * p + 1];
    }
    if (n < 0 || !splat_tile_rect(gx, gy, gz, grx, gry, g, tx0, tx1, ty0, ty1)) continue;
    const size_t sub0 = ((size_t)n * g.tiles_x * g.tiles_y) * DSS_SUB + (sorted ? ((full >> 4) & (DSS_SUB - 1)) : ((unsigned)p & (DSS_SUB - 1)));
    // the splat's full tiles: bits 0..3 of the mask byte (a rectangle of at most 2 x 2 tiles), or the 64-bit mask of a larger one
    const bool is_big = (full & 0x80u) != 0u;
    unsigned long long todo = is_big ? sp.big[p] : (unsigned long long)(full & 0xfu);
#pragma unroll 1
    while (todo != 0ull) {
        const int k = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int tx = is_big ? tx0 + (k & 7) : ((k & 1) ? tx1 : tx0), ty = is_big ? ty0 + (k >> 3) : ((k & 2) ? ty1 : ty0);
        pool_put(sub0 + (size_t)(ty * g.tiles_x + tx) * DSS_SUB, p);
    }
    }
    }
    // splats wider than 8 x 8 tiles: one record per 8 x 8 block of their rectangle that met a full sub-list (bin_rect)
    if (sp.giant != nullptr) {
#pragma unroll 1
        for (unsigned seg = 0; seg < 8u; ++seg) {
            const uint32_t nrec = min((uint32_t)__builtin_amdgcn_readfirstlane((int)sp.ctrl[2 + seg]), sp.giant_cap);
            for (uint32_t r = block * blockDim.x + threadIdx.x; r < nrec; r += nblocks * blockDim.x) {
                const uint4 rec = sp.giant[(size_t)seg * sp.giant_cap + r];
                const int64_t p = (int64_t)rec.x;   // (sorted path: a position of the order, like every list entry)
                const int n = find_cloud(crec ? (int64_t)s_id[p] : p, first_idx, num_pts, N);
                if (n < 0) continue;
                const int bx = (int)(rec.y & 0x3fffu), by = (int)((rec.y >> 14) & 0x3fffu);
                const size_t sub0 = ((size_t)n * g.tiles_x * g.tiles_y) * DSS_SUB + (rec.y >> 28);
                unsigned long long todo = ((unsigned long long)rec.w << 32) | rec.z;
#pragma unroll 1
                while (todo != 0ull) {
                    const int k = __builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    pool_put(sub0 + (size_t)((by + (k >> 3)) * g.tiles_x + bx + (k & 7)) * DSS_SUB, p);
                }
            }
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void spill_kernel(
    const float *__restrict__ points, const float *__restrict__ radii, const int64_t *__restrict__ first_idx,
    const int64_t *__restrict__ num_pts, int N, int64_t P, TileGrid g, const uint32_t *__restrict__ counts, uint32_t cap,
    Spill sp, int sorted)
{
    // grid-stride over the splats: the grid is bounded (spill_grid), so that the usual empty launch costs at most 2048
    // workgroup dispatches whatever P is
    spill_pass(points, radii, first_idx, num_pts, N, P, g, counts, cap, sp, sorted, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Cell-ordered binning for large inputs (P > SORT_MIN_P).
//
// setup_bin_kernel on a randomly ordered cloud of 8M splats costs 1.0 ms: 0.35 ms of per-point setup traffic, 0.24 ms of
// 4-byte list appends that each dirty their own 64-byte sector (1.7 GB written for 0.9 GB of output) and 0.4 ms of
// returning global atomics (~2.2 per splat at ~16 G/s) -- measured by removing each in turn.  A cloud in Morton order pays
// 0.65 ms for the same work, because neighbouring threads then append to the same few lists.  Here the renderer creates
// that order itself, without global atomics:
//   setup_cell_kernel   per-point setup (same body) + the 32 x 32-pixel screen cell of every splat's centre + a histogram
//                       of the workgroup's 16,384 splats in LDS -> block_hist[block][cell]
//   sort_block_scan / sort_seg_scan / sort_cell_scan   exclusive prefix of every cell over the blocks (two levels), of the
//                       cells over the image
//   sort_scatter_kernel every workgroup ranks its splats again in LDS and writes (px, py, rx, ry | id) in cell order
//   bin_sorted_kernel   1024 consecutive sorted splats per workgroup -- one or two cells -- : the (tile, sub-list) keys of
//                       their pairs are counted in an LDS hash table, ONE global atomicAdd per key reserves the run, the
//                       ids are written behind it: ~8x fewer global atomics, list appends in contiguous runs
//   queue_build_kernel  the occupied-tile queues of the fine pass from the finished counters (no per-tile claims)
// Measured (same-run A/B against the direct binning, tools/ab_libs.py): 8 x 1M splats @1024^2 setup 0.40 + scans 0.05 +
// scatter 0.27 + binning 0.18 ms against 1.01 ms, step +2 %; 4M splats @2048^2 step +4 %; 8 x 99,790 splats -5 % (hence
// SORT_MIN_P).  The order costs what the atomics cost: the scatter of 20 bytes per splat to random positions is the
// same partial-sector traffic the list appends were.
// The tile lists, counters, flags, queues and the spill path are the ones of the direct binning: the fine pass does not
// know which of the two filled them (the K-set of a pixel does not depend on the order of its candidates).
// ---------------------------------------------------------------------------------------------
#ifndef SORT_MIN_P               // (-DSORT_MIN_P=...: development A/B builds, tools/ab_bench.py)
#define SORT_MIN_P 2000000      // below ~2M splats the direct binning wins (8 x 99,790 points: 0.69 vs 0.73 ms per step)
#endif
#define SORT_THREADS 1024
#define SORT_PER_THREAD 16                            // at most: 16,384 splats per workgroup of the histogram pass
#define SORT_CELL_MAX 16384                           // 64 KB of LDS counters
#define SORT_NO_CELL 0xffffffffu
struct SortGrid {
    int shift;     // cell side = 1 << shift pixels
    int cx, cy;    // cells per image row / column
    int total;     // N * cx * cy + 1 (<= SORT_CELL_MAX)
};
static SortGrid make_sort_grid(int N, int S)
{
    SortGrid c;
    c.shift = 5;
    for (;;) {
        c.cx = ((S - 1) >> c.shift) + 1;
        c.cy = c.cx;
        c.total = N * c.cx * c.cy + 1;   // + the cell of the culled splats (last; only filled when the order is saved)
        if (c.total <= SORT_CELL_MAX || c.shift >= 14) break;
        ++c.shift;
    }
    return c;
}
// splats per thread of the histogram pass: at least ~2 workgroups per CU (49 workgroups of 16,384 splats left four fifths of
// the chip idle at 8 x 99,790 points), at most SORT_PER_THREAD
static inline int sort_per_thread(int64_t P)
{
    int64_t per = P / ((int64_t)SORT_THREADS * 512);
    return (int)(per < 1 ? 1 : (per > SORT_PER_THREAD ? SORT_PER_THREAD : per));
}
static inline size_t sort_blocks(int64_t P)
{
    const int64_t chunk = (int64_t)SORT_THREADS * sort_per_thread(P);
    return (size_t)((P + chunk - 1) / chunk);
}
// MODE 0: the order of this call only (culled splats are left out).  MODE 1 (DSS_WS_ORDER_SAVE): culled splats are sorted
// too, into the last cell, so that the order is a permutation of ALL points and stays usable when the culling changes.
// MODE 2 (DSS_WS_ORDER_REUSE): no histogram at all -- the setup alone, plus the 16-byte binning record (px, py, rx, ry) of
// every point in NATURAL order (rx = -1: culled) that bin_sorted_kernel<true> gathers through the saved order.
template <int MODE>
__global__ __launch_bounds__(SORT_THREADS) void setup_cell_kernel(const SetupArgs A, SortGrid sg, int per_thread,
                                                                   uint32_t *__restrict__ cell_of,
                                                                   uint32_t *__restrict__ block_hist, Spill sp,
                                                                   uint8_t *__restrict__ visible_to_clear,
                                                                   float4 *__restrict__ crec /* (P,2) candidate records by position */,
                                                                   const int32_t *__restrict__ inv /* MODE 2: position of every splat */,
                                                                   TileGrid g, int band_only, uint8_t *__restrict__ live = nullptr)
{
    // band_only (DSS_WS_BAND_OUTPUTS, multi-GPU): splats whose tile rectangle misses the rank's band store their screen
    // position, radii and validity only; MODE 0 leaves them out of the sort, MODE 2 marks them in the binning record (rx = -1,
    // like a culled splat) -- every rank still evaluates every splat (the medians of the backward need all radii), but of the
    // ~140 bytes a splat writes here seven eighths of the cloud keep 38
    extern __shared__ uint32_t s_hist[];
    if (sp.ctrl && blockIdx.x == 0 && threadIdx.x == 0) spill_begin(sp);
    if (MODE != 2) {
        for (int c = threadIdx.x; c < sg.total; c += SORT_THREADS) s_hist[c] = 0u;
        __syncthreads();
    }
    const int64_t b0 = (int64_t)blockIdx.x * (MODE == 2 ? (int)blockDim.x : SORT_THREADS) * per_thread;
    if (MODE == 2) {
        // this kernel is nothing but streaming stores (~140 bytes per point): full wavefronts write them as contiguous runs
        // (setup_wave_store; the dynamic LDS of this instantiation is 4 KB per wavefront)
        float4 *wave_lds = reinterpret_cast<float4 *>(s_hist) + (threadIdx.x >> 6) * 256;
        const bool wide = setup_wide_ok(A);
#pragma unroll 1
        for (int u = 0; u < per_thread; ++u) {
            const int64_t p0 = b0 + (int64_t)u * blockDim.x + (threadIdx.x & ~63);   // first point of the wavefront
            if (p0 >= A.P) break;
            const int64_t p = p0 + (threadIdx.x & 63);
            const bool full = wide && p0 + 64 <= A.P;
            if (!full && p >= A.P) continue;
            if (visible_to_clear) visible_to_clear[p] = 0;
            const int n = find_cloud(p, A.first_idx, A.num_pts, A.N);
            const SetupVals v = setup_point_compute(A, p, n);
            if (band_only) {
                int tx0, tx1, ty0, ty1;
                const bool reach = n >= 0 && splat_tile_rect(v.sx, v.sy, v.sz, v.rx, v.ry, g, tx0, tx1, ty0, ty1);
                if (full) {
                    // what every point writes, as contiguous runs (the (P,3) positions transposed through LDS like
                    // setup_wave_store); the rest per lane, for the lanes whose splat meets the band
                    float *l1 = reinterpret_cast<float *>(wave_lds);
                    const int lane = threadIdx.x & 63;
                    __builtin_amdgcn_wave_barrier();
                    l1[3 * lane] = v.sx; l1[3 * lane + 1] = v.sy; l1[3 * lane + 2] = v.sz;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 48) reinterpret_cast<float4 *>(A.screen + 3 * p0)[lane] = wave_lds[lane];
                    __builtin_amdgcn_wave_barrier();
                    reinterpret_cast<float2 *>(A.radii)[p] = make_float2(v.rx, v.ry);
                    A.valid[p] = v.ok;
                    if (reach) {
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
                } else {
                    setup_point_store(A, p, v, reach);
                }
                // only the splats that meet the band store a candidate record (one scattered 32-byte store each: for all 8M
                // splats of configs[3] that was 100 us of a rank's 240 us setup) and raise the byte of their position; the
                // binning reads the bytes in position order, takes the raised ones and lowers them again, so that the stale
                // record of a splat that has left the band is never looked at
                if (reach) {
                    const size_t pos = (size_t)min((uint32_t)inv[p], (uint32_t)(A.P - 1));   // (clamped: a lost order cannot fault)
                    crec[2 * pos] = make_float4(v.sx, v.sy, v.rx, v.ry);
                    crec[2 * pos + 1] = make_float4(v.ea, v.eb, v.ec, v.sz);
                    live[pos] = 1;
                }
                continue;
            }
            if (full) setup_wave_store(A, p0, v, wave_lds);
            else setup_point_store(A, p, v);
            const bool live = n >= 0 && !(v.sz < 0);   // culled splats (pz = -1) never reach a tile list
            {
                // the candidate record goes to the splat's POSITION in the saved order (two 16-byte stores into one 32-byte
                // slot: neighbours on the screen are neighbours there)
                const size_t pos = (size_t)min((uint32_t)inv[p], (uint32_t)(A.P - 1));
                crec[2 * pos] = make_float4(v.sx, v.sy, live ? v.rx : -1.0f, v.ry);
                crec[2 * pos + 1] = make_float4(v.ea, v.eb, v.ec, v.sz);
            }
        }
        return;
    }
#pragma unroll 1
    for (int u = 0; u < per_thread; ++u) {
        const int64_t p = b0 + (int64_t)u * SORT_THREADS + threadIdx.x;
        if (p >= A.P) break;
        if (visible_to_clear) visible_to_clear[p] = 0;
        const int n = find_cloud(p, A.first_idx, A.num_pts, A.N);
        float px, py, pz, rx, ry;
        bool reach = true;
        if (band_only) {
            const SetupVals v = setup_point_compute(A, p, n);
            int tx0, tx1, ty0, ty1;
            reach = n >= 0 && splat_tile_rect(v.sx, v.sy, v.sz, v.rx, v.ry, g, tx0, tx1, ty0, ty1);
            setup_point_store(A, p, v, reach);
            px = v.sx; py = v.sy; pz = v.sz; rx = v.rx; ry = v.ry;
        } else {
            setup_point(A, p, n, px, py, pz, rx, ry);
        }
        // culled splats (pz = -1) never reach a tile list; band_only, MODE 0: neither do the splats that miss the band
        const bool live = n >= 0 && !(pz < 0) && (MODE == 1 || reach);
        uint32_t key = SORT_NO_CELL;
        if (live) {
            // pixel column / row of the centre (any monotone map of NDC does: only neighbourhood matters); NaN -> cell 0
            const float fx = (1.0f - px) * 0.5f * (float)A.S, fy = (1.0f - py) * 0.5f * (float)A.S;
            const int ix = min(max((int)fx, 0), A.S - 1) >> sg.shift, iy = min(max((int)fy, 0), A.S - 1) >> sg.shift;
            key = (uint32_t)((n * sg.cy + iy) * sg.cx + ix);
        } else if (MODE == 1) {
            key = (uint32_t)(sg.total - 1);
        }
        if (key != SORT_NO_CELL) atomicAdd(&s_hist[key], 1u);
        cell_of[p] = key;
    }
    __syncthreads();
    uint32_t *row = block_hist + (size_t)blockIdx.x * sg.total;
    for (int c = threadIdx.x; c < sg.total; c += SORT_THREADS) row[c] = s_hist[c];
}
// Prefix of every cell over the blocks, two levels (one thread per cell walking 500+ blocks was 36-39 us of load latency):
// sort_block_scan_kernel, grid (cells / 256, segments of SORT_SEG blocks): block_hist[b][c] <- the cell's splats in the
// earlier blocks of b's segment, seg_tot[seg][c] <- the segment's total; sort_seg_scan_kernel, one thread per cell:
// seg_tot[seg][c] <- the cell's splats in earlier segments, cell_total[c].
#define SORT_SEG 8
__global__ __launch_bounds__(256) void sort_block_scan_kernel(uint32_t nb, int total, uint32_t *__restrict__ block_hist,
                                                              uint32_t *__restrict__ seg_tot)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    const uint32_t b0 = blockIdx.y * SORT_SEG;
    uint32_t v[SORT_SEG];
#pragma unroll
    for (int u = 0; u < SORT_SEG; ++u) v[u] = (b0 + u < nb) ? block_hist[(size_t)(b0 + u) * total + c] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int u = 0; u < SORT_SEG; ++u) {
        if (b0 + u < nb) block_hist[(size_t)(b0 + u) * total + c] = run;
        run += v[u];
    }
    seg_tot[(size_t)blockIdx.y * total + c] = run;
}
__global__ __launch_bounds__(256) void sort_seg_scan_kernel(uint32_t nseg, int total, uint32_t *__restrict__ seg_tot,
                                                            uint32_t *__restrict__ cell_total)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= total) return;
    uint32_t run = 0, sg = 0;
    for (; sg + 8 <= nseg; sg += 8) {   // eight independent loads in flight per trip
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = seg_tot[(size_t)(sg + u) * total + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            seg_tot[(size_t)(sg + u) * total + c] = run;
            run += v[u];
        }
    }
    for (; sg < nseg; ++sg) {
        const uint32_t v = seg_tot[(size_t)sg * total + c];
        seg_tot[(size_t)sg * total + c] = run;
        run += v;
    }
    cell_total[c] = run;
}
// one workgroup: exclusive scan of the cell totals -> cell_start; the number of sorted splats -> *sorted_count
__global__ __launch_bounds__(1024) void sort_cell_scan_kernel(const uint32_t *__restrict__ cell_total, uint32_t *__restrict__ cell_start,
                                                              int total, uint32_t *__restrict__ sorted_count)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (total + 1023) / 1024;
    const int b = tid * per, e = min(b + per, total);
    uint32_t sum = 0;
    for (int i = b; i < e; ++i) sum += cell_total[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = tid >= o ? s_part[tid - o] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (int i = b; i < e; ++i) {
        cell_start[i] = run;
        run += cell_total[i];
    }
    if (tid == 1023) *sorted_count = s_part[1023];
}
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(
    const float *__restrict__ points, const float *__restrict__ radii, const float *__restrict__ ellipse, int64_t P, SortGrid sg,
    int per_thread, const uint32_t *__restrict__ cell_of,
    const uint32_t *__restrict__ cell_start, const uint32_t *__restrict__ block_base, const uint32_t *__restrict__ seg_base,
    float4 *__restrict__ crec, int32_t *__restrict__ s_id, int32_t *__restrict__ inv /* saved order only, else nullptr */)
{
    extern __shared__ uint32_t s_hist[];
    // (merging two or four histogram blocks per scatter workgroup -- longer runs per cell -- measured 2-4 % slower)
    const uint32_t *base = block_base + (size_t)blockIdx.x * sg.total;
    const uint32_t *sbase = seg_base + (size_t)(blockIdx.x / SORT_SEG) * sg.total;
    for (int c = threadIdx.x; c < sg.total; c += SORT_THREADS) s_hist[c] = cell_start[c] + sbase[c] + base[c];
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * SORT_THREADS * per_thread;
#pragma unroll 4
    for (int u = 0; u < per_thread; ++u) {
        const int64_t p = b0 + (int64_t)u * SORT_THREADS + threadIdx.x;
        if (p >= P) break;
        const uint32_t key = cell_of[p];
        if (key == SORT_NO_CELL) continue;
        const float2 rr = reinterpret_cast<const float2 *>(radii)[p];
        // (the last cell only receives splats when the order is saved: the culled ones, rx = -1 for the binning)
        const float4 ge = make_float4(points[3 * p], points[3 * p + 1], key == (uint32_t)(sg.total - 1) ? -1.0f : rr.x, rr.y);
        // (a splat that wrote position / radii only -- DSS_WS_BAND_OUTPUTS, saved order -- has no ellipse: it is never binned)
        const float4 el = make_float4(ellipse[3 * p], ellipse[3 * p + 1], ellipse[3 * p + 2], points[3 * p + 2]);
        const uint32_t pos = atomicAdd(&s_hist[key], 1u);
        crec[2 * (size_t)pos] = ge;
        crec[2 * (size_t)pos + 1] = el;
        s_id[pos] = (int32_t)p;
        if (inv) inv[p] = (int32_t)pos;
    }
}

// Binning in cell order.  Workgroup b handles the sorted splats [1024 b, 1024 b + 1024), four per thread.  Splat j of thread t
// goes to sub-list sub(t, j) = (j + 4 (t mod 2)) of each of its tiles: the (at most 2 x 2) pairs of a splat share the sub-list -- the
// spill pass finds it in bits 4..6 of the splat's mask byte --, a workgroup spreads a tile's entries over all eight
// sub-lists (four per workgroup concentrated a dense tile's entries and made the pool pass do real work).  Keys (tile, sub) are counted in an open-addressing LDS table
// (SORT_HASH slots >= the 4096 pairs a workgroup can have), one global atomicAdd per key reserves its run.
#define SORT_BIN_THREADS 256
#define SORT_BIN_PER_THREAD 4
#define SORT_BIN_CHUNK (SORT_BIN_THREADS * SORT_BIN_PER_THREAD)   // sorted splats per workgroup
#define SORT_HASH 4096
#define SORT_EMPTY 0xffffffffu
// (256 threads x 4 splats: a 1024-thread workgroup with one splat per thread ran one workgroup per CU -- four barrier-
// separated phases of dependent latencies with nothing else resident to hide them: 508 us at 8M splats)
// GATHER (DSS_WS_ORDER_REUSE): s_id is the order an earlier call saved -- a permutation of all P points --, s_geo the
// records of THIS call in natural order: entry i bins point s_id[i] with s_geo[s_id[i]] (one 16-byte sector per splat).
// Any permutation gives the same tile lists up to the order of their entries, which the fine pass does not depend on; a
// stale order only costs locality.  (Ids are clamped to [0, P): a workspace that lost its order cannot fault.)
template <bool GATHER>
__global__ __launch_bounds__(SORT_BIN_THREADS) void bin_sorted_kernel(
    const float4 *__restrict__ crec, const int32_t *__restrict__ s_id, const uint32_t *__restrict__ sorted_count,
    const int64_t *__restrict__ first_idx, const int64_t *__restrict__ num_pts, int N, TileGrid g,
    uint32_t *__restrict__ counts, int32_t *__restrict__ lists, uint32_t cap, TileQueue tq, Spill sp, uint32_t P,
    uint8_t *__restrict__ live_pos = nullptr /* GATHER, DSS_WS_BAND_OUTPUTS: the positions whose record this call's setup stored */)
{
    __shared__ uint32_t h_key[SORT_HASH];
    __shared__ uint32_t h_cnt[SORT_HASH];   // count of the key, then (after the reservation) the run's first list position
    const uint32_t count = GATHER ? P : *sorted_count;
    const uint32_t b0 = blockIdx.x * SORT_BIN_CHUNK;
    if (b0 >= count) return;
    const int tiles = g.tiles_x * g.tiles_y;
    // the four splats of this thread: entries b0 + j * 256 + tid; all loads first
    // (round 5: the candidate records are stored by POSITION in both forms -- the saved order's records were gathered through
    // the order before, one 16-byte sector of a random 128-byte line per splat -- and the lists / mask bytes hold positions)
    int pid[SORT_BIN_PER_THREAD];       // point id (cloud lookup only)
    uint32_t posn[SORT_BIN_PER_THREAD]; // position in the order = list entry
    float4 ge[SORT_BIN_PER_THREAD];
#pragma unroll
    for (int j = 0; j < SORT_BIN_PER_THREAD; ++j) {
        const uint32_t i = b0 + (uint32_t)j * SORT_BIN_THREADS + threadIdx.x;
        const uint32_t ic = i < count ? i : count - 1u;
        posn[j] = ic;
        bool fresh = i < count;
        if (GATHER && live_pos != nullptr) {
            fresh = fresh && live_pos[ic] != 0;
            if (fresh) live_pos[ic] = 0;   // (this thread is the byte's only reader)
        }
        pid[j] = fresh ? s_id[ic] : -1;
        ge[j] = fresh ? crec[2 * (size_t)ic] : make_float4(0.0f, 0.0f, -1.0f, 0.0f);
        if (GATHER && pid[j] >= 0) pid[j] = (int)min((uint32_t)pid[j], P - 1u);   // (a lost order cannot fault)
    }
    if (GATHER) {
        // a chunk without a single live record leaves here: on a rank of the multi-GPU step (DSS_WS_BAND_OUTPUTS) seven of
        // eight chunks hold only splats outside the rank's rows (rx = -1), and the hash table's reset, scan and three
        // barriers were most of what such a chunk cost
        bool live = false;
#pragma unroll
        for (int j = 0; j < SORT_BIN_PER_THREAD; ++j) live = live || (pid[j] >= 0 && !(ge[j].z < 0));
        if (!__syncthreads_or(live)) return;
    }
    for (int k = threadIdx.x; k < SORT_HASH; k += SORT_BIN_THREADS) { h_key[k] = SORT_EMPTY; h_cnt[k] = 0u; }
    __syncthreads();
    int cl[SORT_BIN_PER_THREAD], rx0[SORT_BIN_PER_THREAD], rx1[SORT_BIN_PER_THREAD], ry0[SORT_BIN_PER_THREAD], ry1[SORT_BIN_PER_THREAD];
    uint32_t slot[SORT_BIN_PER_THREAD][4], rank[SORT_BIN_PER_THREAD][4];
    unsigned onm[SORT_BIN_PER_THREAD];   // bit k: pair k of the splat is aggregated; bit 4: larger than 2 x 2 tiles
#pragma unroll
    for (int j = 0; j < SORT_BIN_PER_THREAD; ++j) {
        const uint32_t sub = ((uint32_t)j + 4u * (threadIdx.x & 1u)) & (DSS_SUB - 1);
        int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
        cl[j] = pid[j] >= 0 ? find_cloud(pid[j], first_idx, num_pts, N) : -1;
        bool any = cl[j] >= 0 && !(ge[j].z < 0) /* culled splat of a saved order */ &&
                   splat_tile_rect(ge[j].x, ge[j].y, 0.0f, ge[j].z, ge[j].w, g, tx0, tx1, ty0, ty1);
        const bool small = any && tx1 - tx0 <= 1 && ty1 - ty0 <= 1;
        rx0[j] = tx0; rx1[j] = tx1; ry0[j] = ty0; ry1[j] = ty1;
        onm[j] = (any && !small) ? 16u : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tx = (k & 1) ? tx1 : tx0, ty = (k & 2) ? ty1 : ty0;
            const bool on = small && (!(k & 1) || tx1 > tx0) && (!(k & 2) || ty1 > ty0);
            slot[j][k] = 0; rank[j][k] = 0;
            if (on) {
                onm[j] |= 1u << k;
                const uint32_t key = (uint32_t)((cl[j] * tiles + ty * g.tiles_x + tx) * DSS_SUB) + sub;
                uint32_t h = (key * 2654435761u) >> 20;   // 12 bits
                for (;;) {
                    const uint32_t old = atomicCAS(&h_key[h], SORT_EMPTY, key);
                    if (old == SORT_EMPTY || old == key) break;
                    h = (h + 1u) & (SORT_HASH - 1);
                }
                slot[j][k] = h;
                rank[j][k] = atomicAdd(&h_cnt[h], 1u);
            }
        }
    }
    __syncthreads();
    // one reservation per key: the run [base, base + n) of its sub-list.  The returning atomics of a thread's slots are
    // issued together.  (No tile is claimed here: the direct binning appends a tile to its queue when a sub-list receives
    // its first entry -- ~60k appends to 32 queue tails, hidden inside a 1 ms kernel; inside this 0.15 ms kernel the same
    // appends serialised on the tails for 0.36 ms.)
    {
        constexpr int SLOTS = SORT_HASH / SORT_BIN_THREADS;
        uint32_t kk[SLOTS], bb[SLOTS];
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int k = threadIdx.x + u * SORT_BIN_THREADS;
            kk[u] = h_key[k];
            bb[u] = 1u;
            if (kk[u] != SORT_EMPTY) bb[u] = atomicAdd(&counts[kk[u]], h_cnt[k]);
        }
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            if (kk[u] == SORT_EMPTY) continue;
            h_cnt[threadIdx.x + u * SORT_BIN_THREADS] = bb[u];   // (the occupied-tile queues are built afterwards: queue_build_kernel)
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SORT_BIN_PER_THREAD; ++j) {
        const uint32_t sub = ((uint32_t)j + 4u * (threadIdx.x & 1u)) & (DSS_SUB - 1);
        unsigned full = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(onm[j] & (1u << k))) continue;
            const uint32_t key = h_key[slot[j][k]];
            const uint32_t pos = h_cnt[slot[j][k]] + rank[j][k];
            if (pos < cap) lists[(size_t)key * cap + pos] = (int32_t)posn[j];
            else full |= 1u << k;
        }
        if (full && sp.ctrl) {
            sp.mask[posn[j]] = (uint8_t)(full | (sub << 4));
            sp.ctrl[0] = 1u;
        }
        if (onm[j] & 16u) {
            // a splat larger than 2 x 2 tiles (radius above 8 pixels): one returning atomic per tile, like the direct binning
            // (full tiles: a 64-bit mask per 8 x 8 block of the rectangle, as in bin_rect)
            const bool one_block = rx1[j] - rx0[j] < 8 && ry1[j] - ry0[j] < 8;
            for (int by = ry0[j]; by <= ry1[j]; by += 8)
                for (int bx = rx0[j]; bx <= rx1[j]; bx += 8) {
                    unsigned long long fullm = 0ull;
                    const int ye = min(by + 7, (int)ry1[j]), xe = min(bx + 7, (int)rx1[j]);
                    for (int ty = by; ty <= ye; ++ty)
                        for (int tx = bx; tx <= xe; ++tx) {
                            const size_t t = (size_t)(cl[j] * tiles + ty * g.tiles_x + tx) * DSS_SUB + sub;
                            const uint32_t pos = atomicAdd(&counts[t], 1u);
                            if (pos < cap) lists[t * cap + pos] = (int32_t)posn[j];
                            else fullm |= 1ull << (8 * (ty - by) + (tx - bx));
                        }
                    if (fullm == 0ull || !sp.ctrl) continue;
                    if (one_block && sp.big != nullptr) {
                        sp.big[posn[j]] = fullm;
                        sp.mask[posn[j]] = (uint8_t)(0x80u | (sub << 4));
                    } else if (sp.giant != nullptr) {
                        const unsigned seg = (unsigned)posn[j] & 7u;
                        const uint32_t r = atomicAdd(&sp.ctrl[2 + seg], 1u);
                        if (r < sp.giant_cap)
                            sp.giant[(size_t)seg * sp.giant_cap + r] = make_uint4((uint32_t)posn[j], (sub << 28) | ((uint32_t)by << 14) | (uint32_t)bx,
                                                                                  (uint32_t)fullm, (uint32_t)(fullm >> 32));
                        else sp.fail[0] = sp.epoch;
                    } else {
                        sp.fail[0] = sp.epoch;
                    }
                    sp.ctrl[0] = 1u;
                }
        }
    }
}

// Occupied-tile queues of the cell-ordered path: one thread per tile looks at its eight sub-list counters; the occupied
// tiles of a workgroup are ranked per queue with ballots and appended with ONE atomicAdd per (workgroup, queue).
__global__ __launch_bounds__(1024) void queue_build_kernel(const uint32_t *__restrict__ counts, int total_tiles, TileGrid g,
                                                           TileQueue tq)
{
    __shared__ uint32_t s_cnt[DSS_QUEUES];    // occupied tiles of this workgroup per queue, then the reserved base
    __shared__ uint32_t s_wave[16][DSS_QUEUES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile = blockIdx.x * 1024 + tid;
    bool occ = false;
    int q = 0;
    if (tile < total_tiles) {
        const uint4 *c4 = reinterpret_cast<const uint4 *>(counts + (size_t)tile * DSS_SUB);
        const uint4 a = c4[0], b = c4[1];
        occ = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) != 0u;
        const int tiles = g.tiles_x * g.tiles_y;
        const int n = tile / tiles, t = tile - n * tiles;
        q = queue_of(n, t % g.tiles_x, t / g.tiles_x, g);
    }
    // rank of this tile among the wave's occupied tiles of the same queue; per-wave totals per queue
    uint32_t my_rank = 0;
    for (int qq = 0; qq < DSS_QUEUES; ++qq) {
        const unsigned long long m = __ballot(occ && q == qq);
        if (occ && q == qq) my_rank = (uint32_t)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (lane == 0) s_wave[wid][qq] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (tid < DSS_QUEUES) {
        uint32_t tot = 0;
        for (int w = 0; w < 16; ++w) { const uint32_t c = s_wave[w][tid]; s_wave[w][tid] = tot; tot += c; }   // exclusive over the waves
        s_cnt[tid] = tot ? atomicAdd(&tq.tail[tid], tot) : 0u;
    }
    __syncthreads();
    if (occ) {
        tq.flag[tile] = 1u;
        const uint32_t pos = s_cnt[q] + s_wave[wid][q] + my_rank;
        if (pos < tq.capq) tq.list[(size_t)q * tq.capq + pos] = tile + 1;
    }
}

// ---------------------------------------------------------------------------------------------
// Fine pass.
// ---------------------------------------------------------------------------------------------

struct FineArgs {
    const float *points, *ellipse, *cutoff, *radii;
    const float4 *rec;         // packed 64-byte splat records (fused forward, see SetupArgs::rec) or nullptr
    // cell-ordered path (round 5): list / pool entries are POSITIONS of the point order; crec (P,2) = the 32-byte candidate
    // record {px,py,rx,ry} {a,b,c,pz} of every position, sort_id (P) = the splat id there; cutoffC = the call's Q threshold
    // (constant in the fused forward).  nullptr: list entries are splat ids (direct binning)
    const float4 *crec;
    const int32_t *sort_id;
    float cutoffC;
    const int64_t *first_idx, *num_pts;
    const uint32_t *counts;    // (N*tiles*DSS_SUB) sub-list fill counts, or nullptr (naive mode)
    const int32_t *lists;      // (N*tiles*DSS_SUB*cap)
    uint32_t cap;              // sub-list capacity
    Spill spill;               // overflowed sub-lists re-binned into a pool (pool == nullptr: whole-cloud fallback)
    TileQueue queue;           // occupied tiles (list == nullptr: one workgroup per tile, identity order)
    int prio;                  // 1: raise the wave priority of heavy tiles (latency-bound launches)
    uint32_t queue_wgs;        // workgroups serving queue slots (DSS_QUEUES x slots per queue); fill workgroups follow
    uint32_t *clean_counts;    // DSS_WS_CLEAN: == counts, every owner resets what it has read; else nullptr
    uint32_t spill_wgs;        // > 0: the first spill_wgs workgroups of the launch are the pool pass (spill_pass) over the P
    int spill_sorted;          //      splats below; spilled tiles wait for them.  0: a launch of its own did it (or nobody)
    int64_t P;
    int32_t *idx;
    float *zbuf;               // may be nullptr: the depth plane is not written (the fused backward never reads it)
    float *qv, *occ;
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

// K-nearest bookkeeping: one 64-bit key per slot, (z bits << 32) | idx.  Hits have z >= 0
// (pz < 0 is culled, rasterize_points.cu:79-80), for which the IEEE bit pattern is monotone, so
// one unsigned 64-bit compare implements the strict total order (z, idx) of
// rasterize_points_cpu.cpp:85 / oracle frag_less.  Empty slots hold ~0.
#define KEY_EMPTY 0xffffffffffffffffull

// sorted insertion of ekey into an ascending K-list held in registers; branch-free.  The list holds the 64-bit keys only:
// the Q value of a fragment is recomputed from its record in the epilogue, for the K survivors of a pixel instead of being
// carried through every insertion and both merge rounds (two of the six v_cndmask per slot and insertion; the kernel is
// bound by VALU issue at scale).
template <int KMAX>
__device__ __forceinline__ void klist_insert(unsigned long long (&key)[KMAX], unsigned long long ekey, bool useful)
{
    // `useful` implies ekey < key[KMAX - 1]; lanes without it keep their list (the lane masks are combined on the scalar unit)
    bool lt[KMAX];  // e < slot[k] on the old list (monotone in k)
#pragma unroll
    for (int k = 0; k < KMAX - 1; ++k) lt[k] = (int)useful & (int)(ekey < key[k]);
    lt[KMAX - 1] = useful;
#pragma unroll
    for (int k = KMAX - 1; k >= 1; --k) {
        key[k] = lt[k - 1] ? key[k - 1] : (lt[k] ? ekey : key[k]);
        asm volatile("" : "+v"(key[k]));   // slot by slot, in place: no renamed copy of the list for the skip path to mirror
    }
    key[0] = lt[0] ? ekey : key[0];
    asm volatile("" : "+v"(key[0]));
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// Comparator networks that sort every UNIMODAL sequence -- ascending, then
// descending -- of K keys: the minimum-size networks found by exhaustive search over the thresholded inputs 0^a 1^b 0^c
// (0-1 principle on a class closed under monotone maps).  5 compare-exchanges at K = 5, where the odd-even transposition
// network that sorts ANY input needs 10.
template <int I, int J, int K>
__device__ __forceinline__ void compare_exchange(unsigned long long (&key)[K])   // smaller key to slot I
{
    const bool sw = key[J] < key[I];
    const unsigned long long ka = key[I], kb = key[J];
    key[I] = sw ? kb : ka;
    key[J] = sw ? ka : kb;
}
template <int K>
__device__ __forceinline__ void sort_unimodal(unsigned long long (&key)[K])
{
    static_assert(K >= 1 && K <= 6, "networks found for K <= 6");
    if constexpr (K == 2) {
        compare_exchange<0, 1>(key);
    } else if constexpr (K == 3) {
        compare_exchange<0, 2>(key); compare_exchange<1, 2>(key);
    } else if constexpr (K == 4) {
        compare_exchange<0, 2>(key); compare_exchange<0, 3>(key); compare_exchange<1, 3>(key); compare_exchange<2, 3>(key);
    } else if constexpr (K == 5) {
        compare_exchange<0, 4>(key); compare_exchange<1, 3>(key); compare_exchange<1, 4>(key); compare_exchange<2, 4>(key);
        compare_exchange<3, 4>(key);
    } else if constexpr (K == 6) {
        compare_exchange<0, 4>(key); compare_exchange<0, 5>(key); compare_exchange<1, 5>(key); compare_exchange<2, 4>(key);
        compare_exchange<3, 5>(key); compare_exchange<2, 3>(key); compare_exchange<4, 5>(key);
    }
}

// One merge round of the candidate slices: this lane's ascending K-list with the list of the lane CTRL maps to
// (quad_perm within the pixel's quad: plain VALU moves, no LDS round trip like ds_bpermute / __shfl_xor).
// Two ascending K-lists A, B -> the K smallest of their union: min(A[i], B[K-1-i]) over i picks exactly those K, as an
// ascending-then-descending sequence (A[i] wins while it is below B[K-1-i] and never again after); the network above
// sorts it (K <= 6; an odd-even transposition network beyond).  Keys are unique across slices (disjoint candidates)
// except KEY_EMPTY, whose payload is the same everywhere.
template <int KMAX, int CTRL>
__device__ __forceinline__ void merge_round(unsigned long long (&key)[KMAX])
{
    unsigned long long okey[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const unsigned lo = dpp_u32<CTRL>((unsigned)key[k]);
        const unsigned hi = dpp_u32<CTRL>((unsigned)(key[k] >> 32));
        okey[k] = ((unsigned long long)hi << 32) | lo;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const bool lt = okey[KMAX - 1 - k] < key[k];
        key[k] = lt ? okey[KMAX - 1 - k] : key[k];
    }
    if constexpr (KMAX <= 6) {
        sort_unimodal<KMAX>(key);
    } else {
#pragma unroll
        for (int round = 0; round < KMAX; ++round) {
#pragma unroll
            for (int k = round & 1; k + 1 < KMAX; k += 2) {
                const bool sw = key[k + 1] < key[k];
                const unsigned long long ka = key[k], kb = key[k + 1];
                key[k] = sw ? kb : ka;
                key[k + 1] = sw ? ka : kb;
            }
        }
    }
}

// Work decomposition of one 8x8 tile (one 256-thread workgroup = 4 wavefronts):
//   wavefront w  -> 4x4 pixel footprint (w%2, w/2) of the tile
//   lane l       -> pixel l/4 of the footprint, candidate slice l%4: a pixel's four slices are one lane QUAD
// A pixel's candidates are split 4 ways; each lane keeps its own K-list and the four lists are merged at the end with
// two quad_perm DPP rounds.  At DSS sizes this kernel is bound by latency on the CUs that host the densest screen
// regions, not by bandwidth: small tiles spread a dense region over several CUs, and the candidate slices cut the
// serial per-pixel chain 4x.
#define FOOT 4
#define FOOT_PER_ROW (DSS_TILE / FOOT)
#define FINE_WAVES (FOOT_PER_ROW * FOOT_PER_ROW)
#define FINE_THREADS (FINE_WAVES * 64)
#define CHUNK FINE_THREADS
#ifndef SPEC
#define SPEC 32   // list entries per sub-list and chunk: DSS_SUB * SPEC == CHUNK
#endif
static_assert(DSS_SUB * SPEC == CHUNK, "one list slot per thread and chunk");

// fill values of the rows [row_begin, valid rows) step row_step of one EMPTY tile, written by one wavefront
__device__ __forceinline__ void fill_tile_rows(const FineArgs &A, int tile_id, int lane, int row_begin, int row_step)
{
    const TileGrid g = A.g;
    const int tiles = g.tiles_x * g.tiles_y;
    const int n = tile_id / tiles;
    const int t = tile_id - n * tiles;
    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
    const int S = g.S, K = A.K;
    const int c0 = tx * DSS_TILE;
    const int cols = min(DSS_TILE, S - c0);
    const int valid_cols = cols * K;
    const int valid_rows = min(DSS_TILE, g.rows - ty * DSS_TILE);
    const size_t tile_base = (((size_t)n * g.rows + (size_t)ty * DSS_TILE) * S + c0) * K;
    for (int rr = row_begin; rr < valid_rows; rr += row_step) {
        const size_t rb = tile_base + (size_t)rr * S * K;
        for (int cc = lane; cc < valid_cols; cc += 64) {
            A.idx[rb + cc] = -1;
            if (A.zbuf) A.zbuf[rb + cc] = -1.0f;
            A.qv[rb + cc] = -1.0f;
        }
        if (lane < cols) {
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
}

#define FILL_TILES 32   // tiles per fill workgroup (<= 32: one flag bit each)
// fill values of up to FILL_TILES consecutive EMPTY tiles (bit i of `empty`: tile t0 + i), written by one 256-thread workgroup in
// 16-byte pieces: thread -> (tile, piece of a tile row), then down the 8 rows.  Needs S % 8 == 0 (every tile full width, every
// piece 16-byte aligned) and, with the fused blend, RGBA output; the caller falls back to fill_tile_rows otherwise.
// (The per-wavefront row loop of fill_tile_rows issues 48 partly filled store instructions per tile; a fill workgroup
// lived 13 us at 512^2 and kept its slot from the occupied tiles of a crowded XCD: tools/fine_timing.py.)
__device__ __forceinline__ void fill_tiles(const FineArgs &A, int t0, unsigned empty, int tid)
{
    const TileGrid g = A.g;
    const int tiles = g.tiles_x * g.tiles_y;
    const int S = g.S, K = A.K;
    const int q4 = 2 * K;   // 16-byte pieces of one tile row in a (.., K) plane
    const int4 m1i = make_int4(-1, -1, -1, -1);
    const float4 m1f = make_float4(-1.f, -1.f, -1.f, -1.f);
    for (int j = tid; j < FILL_TILES * q4; j += FINE_THREADS) {
        const int i = j / q4, e = j - i * q4;
        if (!((empty >> i) & 1u)) continue;
        const int tile_id = t0 + i;
        const int n = tile_id / tiles;
        const int t = tile_id - n * tiles;
        const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
        const int valid_rows = min(DSS_TILE, g.rows - ty * DSS_TILE);
        size_t o = (((size_t)n * g.rows + (size_t)ty * DSS_TILE) * S + (size_t)tx * DSS_TILE) * K + (size_t)e * 4;
        for (int r = 0; r < valid_rows; ++r, o += (size_t)S * K) {
            *reinterpret_cast<int4 *>(A.idx + o) = m1i;
            if (A.zbuf) *reinterpret_cast<float4 *>(A.zbuf + o) = m1f;
            *reinterpret_cast<float4 *>(A.qv + o) = m1f;
        }
    }
    // per-pixel planes: occupancy (2 pieces per tile row) and, with the fused blend, weight sum (2) + RGBA (8)
    const int per = A.image ? 12 : 2;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 w4 = make_float4(1e-4f, 1e-4f, 1e-4f, 1e-4f);   // weight sum clamped to kEpsilon
    for (int j = tid; j < FILL_TILES * per; j += FINE_THREADS) {
        const int i = j / per, e = j - i * per;
        if (!((empty >> i) & 1u)) continue;
        const int tile_id = t0 + i;
        const int n = tile_id / tiles;
        const int t = tile_id - n * tiles;
        const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
        const int valid_rows = min(DSS_TILE, g.rows - ty * DSS_TILE);
        const size_t pix = ((size_t)n * g.rows + (size_t)ty * DSS_TILE) * S + (size_t)tx * DSS_TILE;
        if (e < 2) {
            float *o = A.occ + pix + 4 * e;
            for (int r = 0; r < valid_rows; ++r, o += S) *reinterpret_cast<float4 *>(o) = z4;
        } else if (e < 4) {
            float *o = A.wsum + pix + 4 * (e - 2);
            for (int r = 0; r < valid_rows; ++r, o += S) *reinterpret_cast<float4 *>(o) = w4;
        } else {
            float *o = A.image + (size_t)n * A.img_sn + (size_t)(ty * DSS_TILE) * A.img_sr + (size_t)(tx * DSS_TILE + (e - 4)) * 4;
            for (int r = 0; r < valid_rows; ++r, o += A.img_sr) *reinterpret_cast<float4 *>(o) = z4;
        }
    }
}

// The fine kernel's (only) by-value argument, re-read from the kernarg segment (offset 0): same bytes as the parameter,
// but loaded where they are used instead of living in SGPRs from the kernel's entry on.
// REQUIRES: every kernel that inlines fine_tile takes `const FineArgs` as its FIRST by-value parameter (kernarg offset 0):
// fine_kernel below is the only one.  A kernel with another leading argument would read garbage here.
__device__ __forceinline__ const FineArgs &reloaded_args()
{
    auto kp = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const FineArgs *)(const char *)kp;
}

template <int KMAX, bool PACKED>
__device__ __forceinline__ void fine_tile(const FineArgs &A_entry, const int tile_id, int32_t *slot_to_clear)
{
    const FineArgs &A = A_entry;
    // candidate chunk: three records per splat -> 2x ds_read_b128 + 1x ds_read_b64 per test
    // slot CHUNK of each array is a sentinel that no pixel hits (negative radii): it pads the survivor lists to whole passes
    __shared__ float4 s_geo[CHUNK + 1];   // px, py, rx, ry
    __shared__ float4 s_ell[CHUNK + 1];   // a, b, c, cutoff
    __shared__ float2 s_zid[CHUNK + 1];   // idx (bits), pz + 0.0f: read as ONE 64-bit word this is the fragment's sort key
    __shared__ unsigned short s_surv[FINE_WAVES][4][CHUNK / 4];  // per wavefront and slice: compacted survivor slots
    constexpr int PLANES = (KMAX <= 8) ? 3 : 1;   // idx / zbuf / qvalue staged together when they fit
    __shared__ int s_out[PLANES][DSS_TILE_PIX * KMAX];

    FT_MARK(0);
    FT_VAL(8, __builtin_amdgcn_s_memrealtime());
    const TileGrid g = A.g;
    const int tiles = g.tiles_x * g.tiles_y;
    const int n = tile_id / tiles;
    const int t = tile_id - n * tiles;
    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;

    // the wavefront index through readfirstlane: everything derived from it (footprint bounds, survivor-list base) lives in SGPRs
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fx = wid % FOOT_PER_ROW, fy = wid / FOOT_PER_ROW;  // footprint inside the tile
    const int pl = lane >> 2, slice = lane & 3;       // pixel inside the footprint, candidate slice
    const int tr = fy * FOOT + (pl >> 2), tc = fx * FOOT + (pl & 3);
    const int r = tile_row0(g, ty) + tr;              // image row
    const int c = tx * DSS_TILE + tc;                 // image col
    const int S = g.S;
    const NdcMap ndc(S);  // same values as pix_to_ndc; one multiply instead of an IEEE divide when S = 2^k
    const float xf = ndc(S - 1 - c);
    const float yf = ndc(S - 1 - r);
    // NDC extent of this wavefront's footprint (pixel centres).  NDC decreases with the image index.
    const int fc0 = tx * DSS_TILE + fx * FOOT, fr0 = tile_row0(g, ty) + fy * FOOT;
    const float f_xmax = ndc(S - 1 - fc0), f_xmin = ndc(S - 1 - (fc0 + 3));
    const float f_ymax = ndc(S - 1 - fr0), f_ymin = ndc(S - 1 - (fr0 + 3));

    // ---- candidate source: the tile's DSS_SUB sub-lists (thread -> sub-list tid/32, slot tid%32 of every chunk) or
    // the whole cloud (naive mode, or a tile with an overflowed sub-list).  The first chunk's list entries are
    // requested SPECULATIVELY, together with the counters: slot < 32 <= cap is always a legal address, the counters
    // only decide afterwards which of the entries exist (one dependent round trip less per tile).
    const uint32_t my_sub = (uint32_t)tid / SPEC, my_slot = (uint32_t)tid % SPEC;
    bool use_list = false;
    uint32_t c_mine = 0, cmax = 0;
    uint32_t cs[DSS_SUB];  // the tile's sub-list counts (wave-uniform)
    const int32_t *lbase = nullptr, *pbase = nullptr;  // this thread's sub-list: primary slots [0, cap), pool beyond
    int32_t id_next = 0;
    if (A.counts != nullptr) {
        lbase = A.lists + ((size_t)tile_id * DSS_SUB + my_sub) * A.cap;
        id_next = lbase[my_slot];
        const uint4 *c4 = reinterpret_cast<const uint4 *>(A.counts + (size_t)tile_id * DSS_SUB);
#pragma unroll
        for (int q = 0; q < DSS_SUB / 4; ++q) {
            const uint4 u = c4[q];  // the same address in every lane: wave-uniform values -> SGPRs
            cs[4 * q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.x);
            cs[4 * q + 1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.y);
            cs[4 * q + 2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.z);
            cs[4 * q + 3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)u.w);
        }
        use_list = true;
        bool spilled = false;
#pragma unroll
        for (int q = 0; q < DSS_SUB; ++q) {
            spilled = spilled || (cs[q] > A.cap);
            cmax = max(cmax, cs[q]);
            c_mine = (my_sub == (uint32_t)q) ? cs[q] : c_mine;  // static indices only: a dynamic cs[sub] goes to scratch
        }
        if (spilled) {
            // rare: some of the tile's sub-lists continue in the spill pool.  Usable if logging and the pool pass completed
            // and every overflowed sub-list got its whole range inside the pool; otherwise scan the whole cloud (exact, slow)
            use_list = A.spill.pool != nullptr;
            if (use_list && A.spill_wgs != 0u) {
                // The pool pass runs in THIS launch (its first spill_wgs workgroups, dispatched before any tile): if it was
                // needed at all -- "some mask set", read BEFORE the counter: the pass's last workgroup resets the word once
                // the counter is complete -- wait until every one of its workgroups has finished; then make their plain
                // stores (pool entries; released by each of them with an agent-scope fence: the XCDs' L2s are not coherent)
                // visible to this workgroup's loads.
                const uint32_t flagged = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&A.spill.ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                const uint32_t done0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.spill.fail[3]);
                uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(&A.spill.fail[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - done0;
                if (flagged != 0u) {
                    for (int spin = 0; spin < (1 << 24) && done < A.spill_wgs; ++spin) {
                        __builtin_amdgcn_s_sleep(8);
                        done = (uint32_t)__builtin_amdgcn_readfirstlane(
                            (int)__hip_atomic_load(&A.spill.fail[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - done0;
                    }
                }
                use_list = done >= A.spill_wgs;   // (not flagged, or gave up: whole-cloud scan -- exact, slow)
                if (use_list) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            use_list = use_list && __builtin_amdgcn_readfirstlane((int)A.spill.fail[0]) !=
                                       __builtin_amdgcn_readfirstlane((int)A.spill.fail[1]);
            if (use_list) {
#pragma unroll
                for (int q = 0; q < DSS_SUB; ++q) {
                    if (cs[q] > A.cap) {
                        const uint32_t o1 = (uint32_t)__builtin_amdgcn_readfirstlane(
                            (int)A.spill.offset[(size_t)tile_id * DSS_SUB + q]);
                        use_list = use_list && o1 != 0u &&
                                   (unsigned long long)(o1 - 1u) + (cs[q] - A.cap) <= A.spill.cap_entries;
                    }
                }
            }
            if (use_list && c_mine > A.cap)
                pbase = A.spill.pool + (A.spill.offset[(size_t)tile_id * DSS_SUB + my_sub] - 1u);
        }
    } else {
#pragma unroll
        for (int q = 0; q < DSS_SUB; ++q) cs[q] = 0;
    }
    if (A.prio) {
        // Latency-bound launches (every occupied tile resident at once, the launch lasts as long as its slowest tile:
        // 20 us against a 10 us mean at 512^2): waves of heavy tiles get issue priority over the light tiles and the fill
        // workgroups they share a SIMD with (fine pass 23.9 -> 21.7 us; thresholds 100..280 all measure the same).
        // Throughput-bound launches lose 2 % with it, so the host only sets the flag for <= 4096 tiles.
        uint32_t tot = 0;
#pragma unroll
        for (int q = 0; q < DSS_SUB; ++q) tot += cs[q];
        if (tot > 220u) __builtin_amdgcn_s_setprio(3);
        else if (tot > 140u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
    }
    int64_t first = 0, count = cmax;
    if (!use_list) {
        first = A.first_idx[n];
        count = A.num_pts[n];
    }
    const int step = use_list ? SPEC : CHUNK;

    // rows of the tile are contiguous runs of 8*K dwords in the (N,rows,S,K) tensors
    const int K = A.K;
    const int run = DSS_TILE * K;
    const int c0 = tx * DSS_TILE;
    const int valid_cols = min(DSS_TILE, S - c0) * K;
    const int valid_rows = min(DSS_TILE, g.rows - ty * DSS_TILE);
    const size_t tile_base = (((size_t)n * g.rows + (size_t)ty * DSS_TILE) * S + c0) * K;

    FT_MARK(1);
#ifdef DSS_FINE_TIMING
    {
        long long ft_total = count;   // candidates of the tile: all sub-lists together
        if (use_list) {
            ft_total = 0;
            for (int q = 0; q < DSS_SUB; ++q) ft_total += cs[q];
        }
        FT_VAL(10, ft_total);
    }
#endif
    if (count <= 0) {
        // empty tile in identity order (naive mode with an empty cloud): stream the fill values, no LDS, no barriers
        fill_tile_rows(A, tile_id, lane, wid, FINE_WAVES);
        FT_MARK(7);
        FT_VAL(9, __builtin_amdgcn_s_memrealtime());
        return;
    }

    unsigned long long key[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) key[k] = KEY_EMPTY;
    unsigned short *surv = &s_surv[wid][0][0];
    FT_DECL(ft_surv);
    if (tid == 0) {   // visible to every wavefront after the first barrier of the chunk loop
        s_geo[CHUNK] = make_float4(0.f, 0.f, -1.f, -1.f);
        s_ell[CHUNK] = make_float4(0.f, 0.f, 0.f, 0.f);
        s_zid[CHUNK] = make_float2(0.f, 0.f);
    }

    // Depth cut of this wavefront's footprint (wave-uniform, refreshed after every chunk that is not the last): a candidate
    // deeper than EVERY pixel's current K-th entry, or farther than the depth-merge threshold behind EVERY pixel's nearest
    // entry so far, fails `useful` below at each of the 16 pixels whatever slice it lands in -- it is dropped in the cull, before
    // its 16 ellipse tests.  (A pixel's bound is the smallest over its four slices: a full slice holds K fragments in front
    // of the candidate; the footprint's bound is the largest over its pixels.  Bit patterns of z >= 0 order like the
    // values; an empty slot reads 0xffffffff = no bound.)  Clustered clouds -- the reference's training loop at
    // configs[2] ends with hundreds of overlapping splats per pixel -- spent their time testing occluded candidates.
    unsigned cut_k = 0xffffffffu, cut_0 = 0xffffffffu;
    for (int64_t base = 0; base < count; base += step) {
        // this thread's candidate of the chunk and its slot in the (dense) LDS staging area
        bool have;
        int64_t p;
        int dst, m;
        if (use_list) {
            have = (uint32_t)base + my_slot < c_mine;
            p = id_next;
            // the next chunk's entry is requested now and arrives while this chunk is processed
            const uint32_t nxt = (uint32_t)base + SPEC + my_slot;
            if (nxt < c_mine) id_next = nxt < A.cap ? lbase[nxt] : pbase[nxt - A.cap];
            // sub-list q contributes clamp(count - base, 0, 32) entries to this chunk; they are packed in sub-list order
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < DSS_SUB; ++q) {
                const uint32_t mq = cs[q] > (uint32_t)base ? min(cs[q] - (uint32_t)base, (uint32_t)SPEC) : 0u;  // scalar
                before += ((uint32_t)q < my_sub) ? mq : 0u;
                total += mq;
            }
            dst = (int)(before + my_slot);
            m = (int)total;
        } else {
            have = base + tid < count;
            p = first + base + tid;
            dst = tid;
            m = (int)min((int64_t)CHUNK, count - base);
        }
        __syncthreads();  // previous chunk fully consumed
        // (cell-ordered path without packed records -- more than 8 fragments per pixel, other channel counts --: the entry is a
        // position, the per-point arrays are indexed by the splat id)
        if (!PACKED && have && use_list && A.crec != nullptr) p = A.sort_id[p];
        if (have) {
            if (PACKED && use_list && A.crec != nullptr) {
                // cell-ordered path: the list entry is a POSITION of the point order; its 32-byte candidate record and its id
                // sit next to those of the tile's other candidates (4 records per 128-byte line, shared with the
                // neighbouring tiles on the same XCD)
                const float4 *C = A.crec + 2 * p;
                const float4 c1 = C[1];
                s_geo[dst] = C[0];
                s_ell[dst] = make_float4(c1.x, c1.y, c1.z, A.cutoffC);
                s_zid[dst] = make_float2(__int_as_float(A.sort_id[p]), c1.w + 0.0f);
            } else if (PACKED) {
                // one 64-byte record = one half cache line per candidate: three loads, one memory transaction (the four
                // separate arrays cost four transactions, and every XCD ended up fetching every line of all of them:
                // point ids are spatially random, so each 128-byte line held a point of every screen region)
                const float4 *R = A.rec + 4 * p;
                s_geo[dst] = R[0];
                s_ell[dst] = R[1];
                s_zid[dst] = make_float2(__int_as_float((int)p), R[3].x + 0.0f);
            } else {
                const float px = A.points[3 * p], py = A.points[3 * p + 1], pz = A.points[3 * p + 2];
                const float2 rr = reinterpret_cast<const float2 *>(A.radii)[p];
                s_geo[dst] = make_float4(px, py, rr.x, rr.y);
                s_ell[dst] = make_float4(A.ellipse[3 * p], A.ellipse[3 * p + 1], A.ellipse[3 * p + 2], A.cutoff[p]);
                s_zid[dst] = make_float2(__int_as_float((int)p), pz + 0.0f);
            }
        }
        __syncthreads();
        if (base == 0) FT_MARK(2);
        // ---- cull + compact: one candidate per lane, ballot, prefix popcount -> per-slice survivor lists ----
        int nsurv = 0;
        for (int sub64 = 0; sub64 < m; sub64 += 64) {
            const int j = sub64 + lane;
            bool keep = false;
            if (j < m) {
                const float4 ge = s_geo[j];
                // conservative, rounding-monotone rejection against the footprint (see splat_tile_rect); bitwise on
                // purpose: short-circuit forms compile to nested exec-mask branches
                const float ez = s_zid[j].y;
                const bool out = (int)(ez < 0) | (int)((f_xmax - ge.x) < -ge.z) | (int)((f_xmin - ge.x) > ge.z) |
                                 (int)((f_ymax - ge.y) < -ge.w) | (int)((f_ymin - ge.y) > ge.w) |
                                 (int)(__float_as_uint(ez) > cut_k) | (int)(ez - __uint_as_float(cut_0) > A.thr);
                keep = !out;
            }
            const unsigned long long mask = __ballot(keep);
            if (keep) {
                const int rank = nsurv + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                // survivor `rank` -> slice rank%4; stored as the byte offset of its 16-byte records
                surv[(rank & 3) * (CHUNK / 4) + (rank >> 2)] = (unsigned short)(j * 16);
            }
            nsurv += __popcll(mask);
        }
        {   // pad to a whole number of two-survivor passes of all four slices with the sentinel slot
            const int rank = nsurv + lane;
            if (lane < 8 && rank < ((nsurv + 7) & ~7)) surv[(rank & 3) * (CHUNK / 4) + (rank >> 2)] = (unsigned short)(CHUNK * 16);
        }
        // same wavefront wrote and now reads `surv`: LDS ops of one wave complete in order; the wave
        // barrier only stops the compiler from moving the reads above the writes
        __builtin_amdgcn_wave_barrier();
        if (base == 0) FT_MARK(3);
        FT_ACC(ft_surv, nsurv);
        // ---- test + insert: lane (pixel, slice) takes survivors slice, slice+4, ...; two per pass so that the LDS
        // reads of both are in flight together ----
        const int trips = (nsurv + 3) >> 2;
        const unsigned short *sv = surv + slice * (CHUNK / 4);
        unsigned pair = *reinterpret_cast<const unsigned *>(sv);  // entries it, it+1 (it is even)
        for (int it = 0; it < trips; it += 2) {
            // all seven LDS reads of the pass are issued before the first test: the records of both survivors and the next
            // pass's pair of offsets (past the end of the list it is a harmless read inside the workgroup's LDS).  The empty
            // asm keeps the second survivor's reads above the first survivor's insertion branch (the compiler sinks them
            // below it otherwise: three dependent LDS round trips per pass instead of one).
            const unsigned off0 = pair & 0xffffu, off1 = pair >> 16;
            typedef float vec4 __attribute__((ext_vector_type(4)));   // a register quadruple the asm constraint can name
            vec4 ge[2], el[2];
            unsigned long long ek[2];
            ge[0] = *reinterpret_cast<const vec4 *>(reinterpret_cast<const char *>(s_geo) + off0);
            el[0] = *reinterpret_cast<const vec4 *>(reinterpret_cast<const char *>(s_ell) + off0);
            ek[0] = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(s_zid) + (off0 >> 1));
            ge[1] = *reinterpret_cast<const vec4 *>(reinterpret_cast<const char *>(s_geo) + off1);
            el[1] = *reinterpret_cast<const vec4 *>(reinterpret_cast<const char *>(s_ell) + off1);
            ek[1] = *reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(s_zid) + (off1 >> 1));
            pair = *reinterpret_cast<const unsigned *>(sv + it + 2);
            asm volatile("" : "+v"(ge[1]), "+v"(el[1]), "+v"(ek[1]), "+v"(pair));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned long long ekey = ek[u];
                const float ez = __uint_as_float((unsigned)(ekey >> 32));
                const float dx = xf - ge[u].x;
                const float dy = yf - ge[u].y;
                // rasterize_points.cu:92-101, same expression order (no FMA contraction)
                const float qval = el[u].x * dx * dx + el[u].y * dx * dy + el[u].z * dy * dy;
                const bool hit = (int)!(fabsf(dx) > ge[u].z) & (int)!(fabsf(dy) > ge[u].w) & (int)!(qval > el[u].w);
                // A hit can only reach the output if it beats this lane's current K-th entry AND lies within
                // the depth-merge threshold of the nearest entry seen so far (the final nearest is never
                // farther, so dropping it now is exact: rasterize_points.cu:586-595 would drop it later).  An empty
                // list has the NaN pattern in the place of the nearest depth: the comparison is false, as it has to be.
                const float znear_now = __uint_as_float((unsigned)(key[0] >> 32));
                const bool useful = (int)hit & (int)(ekey < key[KMAX - 1]) & (int)!(ez - znear_now > A.thr);
                if (__ballot(useful) != 0ull) klist_insert<KMAX>(key, ekey, useful);
            }
        }
        if (base + step < count) {   // (wave-uniform) more chunks follow: refresh the footprint's depth cut
            unsigned hk = (unsigned)(key[KMAX - 1] >> 32), h0 = (unsigned)(key[0] >> 32);
            hk = min(hk, dpp_u32<0xB1>(hk)); h0 = min(h0, dpp_u32<0xB1>(h0));     // the pixel's four slices (quad_perm)
            hk = min(hk, dpp_u32<0x4E>(hk)); h0 = min(h0, dpp_u32<0x4E>(h0));
            hk = max(hk, dpp_u32<0x124>(hk)); h0 = max(h0, dpp_u32<0x124>(h0));   // the row's four pixels (row_ror:4, :8)
            hk = max(hk, dpp_u32<0x128>(hk)); h0 = max(h0, dpp_u32<0x128>(h0));
            cut_k = max(max((unsigned)__builtin_amdgcn_readlane((int)hk, 0), (unsigned)__builtin_amdgcn_readlane((int)hk, 16)),
                        max((unsigned)__builtin_amdgcn_readlane((int)hk, 32), (unsigned)__builtin_amdgcn_readlane((int)hk, 48)));
            cut_0 = max(max((unsigned)__builtin_amdgcn_readlane((int)h0, 0), (unsigned)__builtin_amdgcn_readlane((int)h0, 16)),
                        max((unsigned)__builtin_amdgcn_readlane((int)h0, 32), (unsigned)__builtin_amdgcn_readlane((int)h0, 48)));
        }
    }

    // Everything below reads the kernel arguments again from the kernarg segment (scalar loads through a laundered
    // pointer) instead of keeping the ~25 output / blend pointers in SGPRs through the candidate loop: with them the kernel
    // needed 106 SGPRs (+16 spilled to VGPR lanes), which caps a SIMD at 6 wavefronts whatever the VGPR count says.
    const FineArgs &E = reloaded_args();
    {
        // DSS_WS_CLEAN: this workgroup is the only reader of the tile's counters and of its queue slot, and every thread has
        // read them by now (at least one barrier ago).  Done here, not inside the chunk loop: the three store addresses
        // were live through it (7 VGPRs).
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c));
        if (E.clean_counts && tid_c < DSS_SUB) {
            E.clean_counts[(size_t)tile_id * DSS_SUB + tid_c] = 0;
            if (E.spill.pool && cmax > E.cap) {  // (uniform) the tile used the spill structures: reset them as well
                E.spill.cursor[(size_t)tile_id * DSS_SUB + tid_c] = 0;
                E.spill.offset[(size_t)tile_id * DSS_SUB + tid_c] = 0;
            }
        }
        if (slot_to_clear && tid_c == DSS_SUB) *slot_to_clear = 0;
    }
    FT_MARK(4);
    FT_VAL(11, ft_surv);
    // ---- merge the four candidate slices of every pixel (the lanes of a quad) ----
    merge_round<KMAX, 0xB1>(key);  // quad_perm [1,0,3,2]
    merge_round<KMAX, 0x4E>(key);  // quad_perm [2,3,0,1]

    FT_MARK(5);
    // ---- epilogue: depth merge, occupancy, visibility, Q, blend, stores -- by ALL FOUR lanes of a pixel's quad ----
    // After the two merge rounds the four lanes of a quad hold the same K-list.  Round 3 let lane 0 of the quad do the whole
    // epilogue (K serial fragments: records, Q, weight, colour sums, staging stores) with the other three lanes masked off:
    // a quarter of the lanes active through ~25 % of the kernel's instructions.  Now lane j of the quad takes the fragments
    // k = j, j + 4, ... (two for K = 5..8), requests their Q AND blend records in ONE round trip, and the weight / colour sums
    // are reduced over the quad with two DPP steps: sum = (p0 + p1) + (p2 + p3) with p_j the lane's partial in ascending k --
    // the summation order blend_forward_kernel (blend.hip) uses too, so the fused and the stand-alone blend stay bit-identical.
    // Every thread-derived index of the epilogue is recomputed from a laundered thread id: left alone, the compiler
    // forms the 64-bit store addresses (occupancy, image, weight sum, two rows x three planes) at kernel entry and
    // keeps ~20 VGPRs alive through the candidate loop and the merge (110 VGPRs -> 4 waves per SIMD).
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wid_e = tid_e >> 6, pl_e = lane_e >> 2, j4 = lane_e & 3;
    const int tr_e = (wid_e / FOOT_PER_ROW) * FOOT + (pl_e >> 2), tc_e = (wid_e % FOOT_PER_ROW) * FOOT + (pl_e & 3);
    const int lr_e = ty * DSS_TILE + tr_e, c_e = tx * DSS_TILE + tc_e;   // band-local row, image column
    const bool owner = j4 == 0;
    const bool in_px = (c_e < S) && (lr_e < g.rows);   // the quad's pixel lies inside the image / band
    const bool in_img = owner && in_px;
    // depth merge (rasterize_points.cu:586-595: stop at the first k with z[k] - z[0] > thr), evaluated by every lane
    const float z0 = __uint_as_float((unsigned)(key[0] >> 32));
    const bool any = key[0] != KEY_EMPTY;
    unsigned alive_bits = 0;
    {
        bool alive = any;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const float z = __uint_as_float((unsigned)(key[k] >> 32));
            alive = alive && (key[k] != KEY_EMPTY) && !(z - z0 > E.thr) && (k < K);
            alive_bits |= (alive ? 1u : 0u) << k;
        }
    }
    // this lane's fragments: k = j4 + 4 m
    constexpr int MQ = (KMAX + 3) / 4;
    int mi[MQ];      // point id or -1
    float mz[MQ], mq[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        unsigned long long kk = KEY_EMPTY;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            if (4 * m + jj < KMAX) kk = (j4 == jj) ? key[4 * m + jj] : kk;
        const bool live = in_px && ((alive_bits >> (4 * m + j4)) & 1u) && (4 * m + j4 < KMAX);
        mi[m] = live ? (int)(unsigned)(kk & 0xffffffffull) : -1;
        mz[m] = live ? __uint_as_float((unsigned)(kk >> 32)) : -1.0f;
        mq[m] = -1.0f;
    }
    const size_t pix = ((size_t)n * g.rows + lr_e) * S + c_e;
    const bool blend = E.image != nullptr;
    const float xf_e = ndc(S - 1 - c_e), yf_e = ndc(S - 1 - (tile_row0(g, ty) + tr_e));
    float wk[MQ];
    float cum = 0.0f;
    if (PACKED) {
        // records of the lane's fragments: {px,py,..} {a,b,c,..} for Q and {scaler,f0,f1,f2} for the blend, one round trip
        float2 gp[MQ];
        float4 ge[MQ], br[MQ];
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            gp[m] = make_float2(0.f, 0.f);
            ge[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            br[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mi[m] >= 0) {
                const float4 *R = E.rec + 4 * (size_t)mi[m];
                gp[m] = *reinterpret_cast<const float2 *>(R);
                ge[m] = R[1];
                br[m] = R[2];
            }
        }
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            wk[m] = 0.0f;
            if (mi[m] >= 0) {
                const float dx = xf_e - gp[m].x;
                const float dy = yf_e - gp[m].y;
                mq[m] = ge[m].x * dx * dx + ge[m].y * dx * dy + ge[m].z * dy * dy;  // rasterize_points.cu:92-101
                if (E.visible) E.visible[mi[m]] = 1;
                wk[m] = ewa_weight(mq[m], br[m].x);
                cum += wk[m];
            }
        }
        // fused blend (same arithmetic and order as blend_forward_kernel): w = exp(-q/2) * scaler, img = sum f * (w / cum)
        cum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cum), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        cum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cum), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        if (cum < 1e-4f) cum = 1e-4f;
        const float inv_cum = fast_rcp(cum);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            if (mi[m] >= 0) {
                const float wn = wk[m] * inv_cum;
                a0 += br[m].y * wn;
                a1 += br[m].z * wn;
                a2 += br[m].w * wn;
            }
        }
        a0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a0), 0xB1, 0xf, 0xf, true));
        a1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a1), 0xB1, 0xf, 0xf, true));
        a2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), 0xB1, 0xf, 0xf, true));
        a0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a0), 0x4E, 0xf, 0xf, true));
        a1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a1), 0x4E, 0xf, 0xf, true));
        a2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), 0x4E, 0xf, 0xf, true));
        if (in_img) {
            E.occ[pix] = any ? 1.0f : 0.0f;
            E.wsum[pix] = cum;
            float *o = E.image + (size_t)n * E.img_sn + (size_t)lr_e * E.img_sr + (size_t)c_e * 4;
            // RGBA as ONE 16-byte store per pixel (four dword stores at a 16-byte stride quadruple the write requests)
            *reinterpret_cast<float4 *>(o) = make_float4(a0, a1, a2, any ? 1.0f : 0.0f);
        }
    } else {
        // separate per-point arrays (no packed records: another channel count, the lean workspace, or no fused blend)
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            wk[m] = 0.0f;
            if (mi[m] >= 0) {
                const size_t q = (size_t)mi[m];
                const float dx = xf_e - E.points[3 * q];
                const float dy = yf_e - E.points[3 * q + 1];
                mq[m] = E.ellipse[3 * q] * dx * dx + E.ellipse[3 * q + 1] * dx * dy + E.ellipse[3 * q + 2] * dy * dy;
                if (E.visible) E.visible[q] = 1;
                if (blend) {
                    wk[m] = ewa_weight(mq[m], E.scaler[q]);
                    cum += wk[m];
                }
            }
        }
        if (in_img) E.occ[pix] = any ? 1.0f : 0.0f;
        if (blend) {
            cum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cum), 0xB1, 0xf, 0xf, true));
            cum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cum), 0x4E, 0xf, 0xf, true));
            if (cum < 1e-4f) cum = 1e-4f;
            const float inv_cum = fast_rcp(cum);
            if (in_img) E.wsum[pix] = cum;
            float *o = E.image + (size_t)n * E.img_sn + (size_t)lr_e * E.img_sr + (size_t)c_e * (E.C + 1);
            for (int ch = 0; ch < E.C; ++ch) {
                float acc = 0.0f;
#pragma unroll
                for (int m = 0; m < MQ; ++m)
                    if (mi[m] >= 0) acc += E.feat[(size_t)mi[m] * E.C + ch] * (wk[m] * inv_cum);
                acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xf, 0xf, true));
                acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xf, 0xf, true));
                if (in_img) o[ch] = acc;
            }
            if (in_img) o[E.C] = any ? 1.0f : 0.0f;
        }
    }
    // stage the tile through LDS and let each wavefront stream full image rows per plane; every lane stores its own fragments
    const int lds_pix = (tr_e * DSS_TILE + tc_e) * K;
    if (PLANES == 3) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const int k = 4 * m + j4;
            if (k < K) {
                s_out[0][lds_pix + k] = mi[m];
                s_out[PLANES - 1 > 0 ? 1 : 0][lds_pix + k] = __float_as_int(mz[m]);
                s_out[PLANES - 1][lds_pix + k] = __float_as_int(mq[m]);
            }
        }
        __syncthreads();
        for (int rr = wid_e; rr < valid_rows; rr += FINE_WAVES) {
            const size_t rb = tile_base + (size_t)rr * S * K;
            for (int cc = lane_e; cc < valid_cols; cc += 64) {
                E.idx[rb + cc] = s_out[0][rr * run + cc];
                if (E.zbuf) reinterpret_cast<int *>(E.zbuf)[rb + cc] = s_out[PLANES - 1 > 0 ? 1 : 0][rr * run + cc];
                reinterpret_cast<int *>(E.qv)[rb + cc] = s_out[PLANES - 1][rr * run + cc];
            }
        }
    } else {
#define DSS_STORE_PLANE(REGS, DST, CAST)                                                          \
    __syncthreads();                                                                              \
    _Pragma("unroll") for (int m = 0; m < MQ; ++m) if (4 * m + j4 < K) s_out[0][lds_pix + 4 * m + j4] = CAST(REGS[m]); \
    __syncthreads();                                                                              \
    for (int rr = wid_e; rr < valid_rows; rr += FINE_WAVES) {                                       \
        for (int cc = lane_e; cc < valid_cols; cc += 64)                                            \
            reinterpret_cast<int *>(DST)[tile_base + (size_t)rr * S * K + cc] = s_out[0][rr * run + cc]; \
    }
        DSS_STORE_PLANE(mi, E.idx, (int))
        if (E.zbuf) { DSS_STORE_PLANE(mz, E.zbuf, __float_as_int) }
        DSS_STORE_PLANE(mq, E.qv, __float_as_int)
#undef DSS_STORE_PLANE
    }

    FT_MARK(7);
    FT_VAL(9, __builtin_amdgcn_s_memrealtime());
}

// Queue mode (binned): grid = ceil(N*tiles / FILL_TILES) + queue_wgs.  The first workgroups stream the fill values of the empty
// tiles, FILL_TILES tiles each; queue workgroup qb serves slot qb/32 of queue
// qb%32 (and exits at once when the slot is empty).  Identity mode (naive): one workgroup per tile.
// rounded up to a multiple of 8 workgroups so that queue workgroup qb still lands on XCD qb % 8
__host__ __device__ __forceinline__ uint32_t fill_workgroups(int total_tiles)
{
    return (((uint32_t)total_tiles + FILL_TILES - 1u) / FILL_TILES + 7u) & ~7u;
}

template <int KMAX, bool PACKED>
#ifdef DSS_FINE_OCC8   // (A/B builds: eight wavefronts per SIMD, i.e. at most 64 VGPRs -- the kernel needs 69, seven wavefronts.  Measured at
                       //  the end of round 6: 7 VGPR spills, the step +2 % at configs[1], +2 % at configs[3], +2 % at configs[4]: not taken)
__global__ __launch_bounds__(FINE_THREADS) __attribute__((amdgpu_num_sgpr(96))) __attribute__((amdgpu_waves_per_eu(8, 8))) void fine_kernel(const FineArgs A)
#else
__global__ __launch_bounds__(FINE_THREADS) __attribute__((amdgpu_num_sgpr(96))) void fine_kernel(const FineArgs A)
#endif
{
    const int total = A.N * A.g.tiles_x * A.g.tiles_y;
    const bool qmode = A.queue.list != nullptr;
    const bool clean = A.clean_counts != nullptr;
    // fill workgroups come FIRST in the grid: behind the queue workgroups they started 8-12 us into the kernel (the
    // dispatcher works through ~3000 workgroups whose slot turns out to be empty at ~3 ns each) and ended it
    const uint32_t fill_wgs = qmode ? fill_workgroups(total) : 0u;
    const uint32_t spill_wgs = A.spill_wgs;
    if (blockIdx.x < spill_wgs) {
        // Pool pass inside the fine launch (a launch of its own cost 4.7 us per step although it is empty in the common case:
        // every workgroup returns after one scalar load).  When it did run: release this workgroup's stores (pool entries)
        // to the other XCDs, then count it as finished; the last one to finish resets the two words only this pass and
        // the binning use (all of its workgroups have read ctrl[0] by then).
        const bool ran = spill_pass(A.points, A.radii, A.first_idx, A.num_pts, A.N, A.P, A.g, A.counts, A.cap, A.spill,
                                    A.spill_sorted, blockIdx.x, spill_wgs, A.crec, A.sort_id);
        if (!ran) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t before = __hip_atomic_fetch_add(&A.spill.fail[2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (before - A.spill.fail[3] == spill_wgs - 1u) {
                for (int k = 0; k < 10; ++k)   // "some mask set", pool top, the eight record counters
                    __hip_atomic_store(&A.spill.ctrl[k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (blockIdx.x - spill_wgs < fill_wgs) {
        const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
        const int t0 = (int)(blockIdx.x - spill_wgs) * FILL_TILES;
        FT_MARK(0);
        FT_VAL(8, __builtin_amdgcn_s_memrealtime());
        FT_VAL(10, -1);
        // the workgroup's flags in ONE round trip, by every wavefront (this workgroup is their only reader; wavefront 0 resets
        // them once all four have read)
        const bool mine = lane < FILL_TILES && t0 + lane < total;
        const uint32_t flag = mine ? A.queue.flag[t0 + lane] : 1u;
        const unsigned empty = (unsigned)__ballot(flag == 0u);   // (lanes >= FILL_TILES hold 1)
        if (clean) {
            __syncthreads();
            if (wid == 0 && mine && flag) A.queue.flag[t0 + lane] = 0;
        }
        if (empty == 0u) return;
        const uintptr_t bits = (uintptr_t)A.idx | (uintptr_t)A.zbuf | (uintptr_t)A.qv | (uintptr_t)A.occ | (uintptr_t)A.image |
                               (uintptr_t)A.wsum | (uintptr_t)((A.img_sn | A.img_sr) * 4);
        if ((A.g.S & 7) == 0 && (bits & 15) == 0 && (A.image == nullptr || A.C == 3)) {
            fill_tiles(A, t0, empty, tid);
        } else {
#pragma unroll 1
            for (int i = wid; i < FILL_TILES; i += FINE_WAVES)
                if (empty & (1u << i)) fill_tile_rows(A, t0 + i, lane, 0, 1);
        }
        FT_MARK(7);
        FT_VAL(9, __builtin_amdgcn_s_memrealtime());
        return;
    }
    const uint32_t qb = blockIdx.x - fill_wgs - spill_wgs;  // queue workgroup index (identity mode: tile id)
    if (!qmode && (int)qb >= total) return;
    if (qmode && clean && qb == 0 && threadIdx.x < DSS_QUEUES + (spill_wgs ? 0 : 10)) {
        // state that only binning and the pool pass use: queue tails, and -- unless the pool pass runs in this launch and
        // resets them itself -- "some mask set", pool top, the eight record counters of the wide splats (contiguous words)
        A.queue.tail[threadIdx.x] = 0;
    }
    // one workgroup per queue slot (qb -> queue qb%32, slot qb/32): a loop over several slots per workgroup was tried and
    // costs 60 VGPRs (values hoisted out of the tile loop), and most slots of a large render hold a tile anyway
    int tile_id = (int)qb;
    int32_t *slot = nullptr;
    if (qmode) {
        slot = A.queue.list + (size_t)(qb % DSS_QUEUES) * A.queue.capq + qb / DSS_QUEUES;
        // uniform value, but a vector load (the kernel also writes the slot): tell the compiler, or the tile id and
        // everything derived from it lives in VGPRs.  Reset (DSS_WS_CLEAN) inside fine_tile once every thread read it.
        tile_id = __builtin_amdgcn_readfirstlane(*slot) - 1;
    }
    if (tile_id < 0 || tile_id >= total) return;
    fine_tile<KMAX, PACKED>(A, tile_id, (qmode && clean) ? slot : nullptr);
}

// ---------------------------------------------------------------------------------------------
// Generic-K path for DSS_MAX_K_FAST < K <= kMaxPointsPerPixel (=150, rasterization_utils.cuh:18):
// one wavefront per 8x8 tile, one lane per pixel, the K-list lives in scratch memory (like the
// reference's thread-local `Pix q[150]`, rasterize_points.cu:177) and is kept sorted by binary
// insertion.  Rare configuration: correctness over speed; identity order, every tile scans what it is given.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fine_generic_kernel(const FineArgs A)
{
    const TileGrid g = A.g;
    const int tiles = g.tiles_x * g.tiles_y;
    const int tile_id = (int)blockIdx.x;
    if (tile_id >= A.N * tiles) return;
    const int n = tile_id / tiles;
    const int t = tile_id - n * tiles;
    const int ty = t / g.tiles_x, tx = t - ty * g.tiles_x;
    const int lane = threadIdx.x;
    const int lr = ty * DSS_TILE + (lane >> 3);   // band-local row
    const int r = tile_row0(g, ty) + (lane >> 3);
    const int c = tx * DSS_TILE + (lane & 7);
    const int S = g.S, K = A.K;
    const float xf = pix_to_ndc(S - 1 - c, S);
    const float yf = pix_to_ndc(S - 1 - r, S);
    // candidates: the tile's sub-lists when none overflowed, else the whole cloud
    uint32_t cs[DSS_SUB];
    bool use_list = A.counts != nullptr;
    if (use_list) {
        for (int q = 0; q < DSS_SUB; ++q) {
            cs[q] = A.counts[(size_t)tile_id * DSS_SUB + q];
            use_list = use_list && cs[q] <= A.cap;
        }
    }
    const int64_t first = A.first_idx[n], npts = A.num_pts[n];
    unsigned long long key[DSS_MAX_K];
    float kq[DSS_MAX_K];
    int cnt = 0;
    auto visit = [&](int64_t p) {  // p is wave-uniform
        const float pz = A.points[3 * p + 2];
        if (pz < 0) return;
        const float dx = xf - A.points[3 * p];
        const float dy = yf - A.points[3 * p + 1];
        if (fabsf(dx) > A.radii[2 * p] || fabsf(dy) > A.radii[2 * p + 1]) return;
        const float qval = A.ellipse[3 * p] * dx * dx + A.ellipse[3 * p + 1] * dx * dy + A.ellipse[3 * p + 2] * dy * dy;
        if (qval > A.cutoff[p]) return;
        const unsigned long long ekey = ((unsigned long long)__float_as_uint(pz + 0.0f) << 32) | (unsigned long long)(unsigned)p;
        if (cnt == K && !(ekey < key[K - 1])) return;
        int pos = (cnt < K) ? cnt : K - 1;
        while (pos > 0 && ekey < key[pos - 1]) {
            key[pos] = key[pos - 1];
            kq[pos] = kq[pos - 1];
            --pos;
        }
        key[pos] = ekey;
        kq[pos] = qval;
        if (cnt < K) ++cnt;
    };
    if (use_list) {
        for (int q = 0; q < DSS_SUB; ++q) {
            const int32_t *l = A.lists + ((size_t)tile_id * DSS_SUB + q) * A.cap;
            for (uint32_t j = 0; j < cs[q]; ++j) visit((int64_t)l[j]);
        }
    } else {
        for (int64_t j = 0; j < npts; ++j) visit(first + j);
    }
    if (c >= S || lr >= g.rows) return;
    const size_t pix = ((size_t)n * g.rows + lr) * S + c;
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
        if (A.zbuf) A.zbuf[pix * K + k] = z;
        A.qv[pix * K + k] = qv;
    }
}

static int fine_grid(const FineArgs &A, int blocks)
{
    return A.queue.list ? (int)(A.spill_wgs + A.queue_wgs + fill_workgroups(blocks)) : blocks;
}

template <int KMAX>
static void launch_fine(const FineArgs &A, int blocks, hipStream_t st)
{
    if (KMAX <= 8 && A.rec != nullptr && A.image != nullptr && A.C == 3)
        hipLaunchKernelGGL((fine_kernel<(KMAX <= 8 ? KMAX : 8), true>), dim3(fine_grid(A, blocks)), dim3(FINE_THREADS), 0, st, A);
    else
        hipLaunchKernelGGL((fine_kernel<KMAX, false>), dim3(fine_grid(A, blocks)), dim3(FINE_THREADS), 0, st, A);
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

// workspace layout (binned mode): counts | flags | queue tails | queue slots || lists
struct FwdWorkspace {
    uint32_t *counts;  // N*tiles*SUB
    int32_t *lists;    // N*tiles*SUB*cap
    uint32_t cap;
    TileQueue queue;
    Spill spill;
    float4 *rec;         // packed splat records (P x 64 bytes) behind the lists; only carved for the fused forward
    // cell-ordered binning (fused forward, P > SORT_MIN_P; nullptr otherwise): every region is fully rewritten per call
    uint32_t *sort_cell_of;      // (P) cell of every splat, SORT_NO_CELL for culled ones
    uint32_t *sort_block_hist;   // (blocks, cells) per-block histogram, then per-block offsets
    uint32_t *sort_cell_total, *sort_cell_start;   // (cells)
    uint32_t *sort_seg_tot;      // (segments of SORT_SEG blocks, cells)
    uint32_t *sort_count;        // number of sorted (= not culled) splats
    // Round 5: the cell-ordered path keeps a 32-byte CANDIDATE RECORD per position of the order -- {px, py, rx, ry} {a, b, c, pz};
    // rx = -1: culled / not binned -- and the tile lists hold POSITIONS: a tile's candidates are neighbours in memory (the 64-byte
    // records at random point ids pulled a 128-byte line each: 1.37 GB fetched for 0.42 GB of records at 8 x 1M points)
    float4 *sort_crec;           // (P, 2) candidate records in cell order
    int32_t *sort_id;            // (P) splat id in cell order (the saved point order)
    int32_t *sort_inv;           // (P) position of every splat in the saved order (DSS_WS_ORDER_SAVE writes it, _REUSE reads it)
    uint8_t *sort_live;          // (P) by position: 1 = this call's setup stored a record there (DSS_WS_BAND_OUTPUTS + _REUSE, see setup_cell_kernel)
    size_t count_bytes;  // bytes to zero before binning (the DSS_WS_CLEAN region): everything in front of the lists
    size_t bytes;
};

// Sub-list capacity: ~8x the mean number of (splat, tile) pairs per sub-list (2 tiles per splat assumed; round 2 used 4x:
// at 8 x 1M points @1024^2 the limb of the object then overflowed and the pool pass cost 0.11 ms, 3 % of the step), a power of two
// in [64, 16384] (>= SPEC for the speculative first reads; 64 leaves the benchmark scenes -- densest sub-list 42 entries
// at a mean of 2 -- on the primary lists).  Denser sub-lists go through the spill pool.  Depends only on (N, P, S) so the
// size query and the launch agree.
// dss_set_option(DSS_OPT_LEAN_WORKSPACE, 1) (read at every call; the size query and the launch must see the same value): half the sub-list
// capacity (more splats take the spill pass) and no packed records (the fine pass gathers from the five per-point arrays).
// Measured cost: +6 % and +6 % of the step at 8 x 1M points @1024^2 and 4M points @2048^2 (fine pass 0.98 -> 1.25 ms without
// records) for 177+256 -> 110 MB at the latter.  Off by default: 288 GB of HBM make the fast layout the right default.
static bool lean_workspace() { return option(DSS_OPT_LEAN_WORKSPACE) == 1; }
static uint32_t bin_capacity(int N, int64_t P, int S)
{
    const double tiles = (double)((S + DSS_TILE - 1) / DSS_TILE) * ((S + DSS_TILE - 1) / DSS_TILE);
    const double mean_sub = 2.0 * ((double)P / (N > 0 ? N : 1)) / (tiles * DSS_SUB);
    const bool lean = lean_workspace();
    uint32_t cap = lean ? 32 : 64;
    while (cap < 16384 && (double)cap < (lean ? 2.0 : 8.0) * mean_sub) cap <<= 1;
    return cap;
}
// spill pool entries: eight per point (at least 64k; two until round 6: the trained cloud of tools/clustered_timing.py puts
// ~3.5 over-capacity (splat, tile) pairs per splat into it): a scene with more than that falls back to whole-cloud scans for
// the tiles that did not fit (seen with 100k points on 25 tiles, 0.5 % of the screen).  Never zeroed, touched only when used.
static uint32_t spill_capacity(int64_t P)
{
    const int64_t per = lean_workspace() ? 2 : 8;   // (DSS_OPT_LEAN_WORKSPACE keeps round 5's pool and has no masks for wide splats)
    const int64_t c = per * P > 65536 ? per * P : 65536;
    return (uint32_t)(c < 0x7fffff00ll ? c : 0x7fffff00ll);
}

// band of image rows [row0, row1); row_cycle c > 1: only every c-th 8-row tile row of it, starting at row0 (c a power of two)
static TileGrid make_grid(int S, int row0, int row1, int row_cycle = 1)
{
    TileGrid g;
    g.S = S;
    g.row0 = row0;
    g.rows = dss_band_rows(row0, row1, row_cycle);
    g.tiles_x = (S + DSS_TILE - 1) / DSS_TILE;
    g.tiles_y = (g.rows + DSS_TILE - 1) / DSS_TILE;
    g.tshift = 3;
    for (int c = row_cycle; c > 1; c >>= 1) ++g.tshift;
    return g;
}

// (the layout is sized for the full image: a row band uses a prefix of every region)
static FwdWorkspace carve_fwd(void *ws, int N, int64_t P, int S, bool with_records = false)
{
    FwdWorkspace w;
    const TileGrid full = make_grid(S, 0, S);
    const size_t tiles_max = (size_t)N * full.tiles_x * full.tiles_y;
    char *p = reinterpret_cast<char *>(ws);
    w.cap = bin_capacity(N, P, S);
    const size_t cbytes = align_up(tiles_max * DSS_SUB * 4, 256), fbytes = align_up(tiles_max * 4, 256);
    w.queue.capq = queue_capacity(N, full);
    const size_t qbytes = align_up((size_t)DSS_QUEUES * w.queue.capq * 4, 256);
    const size_t mbytes = align_up((size_t)P, 256);
    w.count_bytes = 3 * cbytes + fbytes + 256 + qbytes + mbytes;
    w.counts = reinterpret_cast<uint32_t *>(p);
    w.spill.cursor = reinterpret_cast<uint32_t *>(p + cbytes);
    w.spill.offset = reinterpret_cast<uint32_t *>(p + 2 * cbytes);
    w.queue.flag = reinterpret_cast<uint32_t *>(p + 3 * cbytes);
    w.queue.tail = reinterpret_cast<uint32_t *>(p + 3 * cbytes + fbytes);   // 256-byte block: 32 queue tails ...
    w.spill.ctrl = w.queue.tail + DSS_QUEUES;                               // ... + the two spill control words
    w.queue.list = reinterpret_cast<int32_t *>(p + 3 * cbytes + fbytes + 256);
    w.spill.mask = reinterpret_cast<uint8_t *>(p + 3 * cbytes + fbytes + 256 + qbytes);
    const size_t lists_off = w.count_bytes;
    w.lists = reinterpret_cast<int32_t *>(p + lists_off);
    w.bytes = lists_off + align_up(tiles_max * DSS_SUB * (size_t)w.cap * 4, 256);
    w.spill.cap_entries = spill_capacity(P);
    w.spill.pool = reinterpret_cast<int32_t *>(p + w.bytes);
    w.bytes += align_up((size_t)w.spill.cap_entries * 4, 256);
    w.spill.big = nullptr;
    w.spill.giant = nullptr;
    w.spill.giant_cap = 0;
    if (!lean_workspace()) {
        w.spill.big = reinterpret_cast<unsigned long long *>(p + w.bytes);
        w.bytes += align_up((size_t)P * 8, 256);
        w.spill.giant_cap = (uint32_t)(P / 16 > 1024 ? P / 16 : 1024);   // eight segments: half a record per point (8 bytes)
        w.spill.giant = reinterpret_cast<uint4 *>(p + w.bytes);
        w.bytes += align_up((size_t)8 * w.spill.giant_cap * 16, 256);
    }
    w.spill.fail = reinterpret_cast<uint32_t *>(p + w.bytes);
    static std::atomic<uint32_t> epoch{1};
    w.spill.epoch = ws ? epoch.fetch_add(1, std::memory_order_relaxed) : 0u;
    w.bytes += 256;
    w.rec = nullptr;
    if (with_records && !lean_workspace()) {
        w.rec = reinterpret_cast<float4 *>(p + w.bytes);
        w.bytes += align_up((size_t)P * 64, 256);
    }
    w.sort_cell_of = nullptr;
    w.sort_crec = nullptr; w.sort_id = nullptr; w.sort_inv = nullptr; w.sort_live = nullptr;
    if (with_records && !lean_workspace() && P > SORT_MIN_P) {
        const SortGrid sg = make_sort_grid(N, S);
        w.sort_cell_of = reinterpret_cast<uint32_t *>(p + w.bytes);      w.bytes += align_up((size_t)P * 4, 256);
        w.sort_block_hist = reinterpret_cast<uint32_t *>(p + w.bytes);   w.bytes += align_up(sort_blocks(P) * (size_t)sg.total * 4, 256);
        w.sort_seg_tot = reinterpret_cast<uint32_t *>(p + w.bytes);
        w.bytes += align_up(((sort_blocks(P) + SORT_SEG - 1) / SORT_SEG) * (size_t)sg.total * 4, 256);
        w.sort_cell_total = reinterpret_cast<uint32_t *>(p + w.bytes);   w.bytes += align_up((size_t)sg.total * 4, 256);
        w.sort_cell_start = reinterpret_cast<uint32_t *>(p + w.bytes);   w.bytes += align_up((size_t)sg.total * 4, 256);
        w.sort_count = reinterpret_cast<uint32_t *>(p + w.bytes);        w.bytes += 256;
        w.sort_crec = reinterpret_cast<float4 *>(p + w.bytes);           w.bytes += align_up((size_t)P * 32, 256);
        w.sort_id = reinterpret_cast<int32_t *>(p + w.bytes);            w.bytes += align_up((size_t)P * 4, 256);
        w.sort_inv = reinterpret_cast<int32_t *>(p + w.bytes);           w.bytes += align_up((size_t)P * 4, 256);
        w.sort_live = reinterpret_cast<uint8_t *>(p + w.bytes);          w.bytes += align_up((size_t)P, 256);
    }
    return w;
}

// workgroups of the (normally empty) pool pass: at most 2048 x 256 threads, grid-stride beyond
static unsigned spill_grid(int64_t P)
{
    const int64_t wgs = (P + 255) / 256;
    return (unsigned)(wgs < 2048 ? (wgs > 0 ? wgs : 1) : 2048);
}

// pool-pass workgroups at the front of a fine launch: 1024 splats per 256-thread workgroup (grid-stride), at most 512 -- all
// resident at once, so the pass's internal waits (on a publisher of the same pass) cannot starve --, a multiple of 8 so that
// the queue workgroups keep their XCD (block b runs on XCD b mod 8)
static uint32_t spill_workgroups_in_fine(int64_t P)
{
    const int64_t wgs = (P + 1023) / 1024;
    const uint32_t n = (uint32_t)(wgs < 512 ? (wgs > 0 ? wgs : 1) : 512);
    return (n + 7u) & ~7u;
}

// queue-serving workgroups of a fine launch over `g` (band): one per slot of the band's queues
static uint32_t queue_workgroups(int N, const TileGrid &g)
{
    return (uint32_t)DSS_QUEUES * queue_capacity(N, g);
}

}  // namespace dss

using namespace dss;

extern "C" size_t dss_splat_forward_workspace(int N, int64_t P, int S, int K, int bin_size)
{
    (void)K;
    if (bin_size == 0 || N <= 0 || P <= 0 || S <= 0) return 256;
    return carve_fwd(nullptr, N, P, S).bytes;
}

extern "C" size_t dss_splat_forward_clean_bytes(int N, int64_t P, int S)
{
    if (N <= 0 || P <= 0 || S <= 0) return 0;
    return carve_fwd(nullptr, N, P, S).count_bytes;
}

static int validate_fwd(const char *fn, int N, int64_t P, int S, int K, int row0, int row1, int row_cycle = 1)
{
    if (row_cycle < 1 || (row_cycle & (row_cycle - 1)) || row_cycle > 4096) {
        set_error("%s: row_cycle %d must be a power of two (1 = contiguous band)", fn, row_cycle);
        return DSS_ERR_INVALID_ARGUMENT;
    }
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
                       w.lists, w.cap, w.queue, w.spill, visible_to_clear);
    hipLaunchKernelGGL(spill_kernel, dim3(spill_grid(P)), dim3(256), 0, st, points, radii, first_idx, num_pts, N, P, g, w.counts, w.cap,
                       w.spill, 0);
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
    if (!idx || !qvalue || !occ || !first_idx || !num_pts ||
        (P > 0 && (!points || !ellipse || !cutoff || !radii))) {  // zbuf may be NULL: depth plane not written
        set_error("dss_splat_fine: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const TileGrid g = make_grid(S, row0, row1);
    const long long blocks_ll = (long long)N * g.tiles_x * g.tiles_y;
    if (blocks_ll > 0x7fffffffll) { set_error("dss_splat_fine: too many tiles"); return DSS_ERR_UNSUPPORTED; }
    FineArgs A;
    A.points = points; A.ellipse = ellipse; A.cutoff = cutoff; A.radii = radii; A.rec = nullptr;
    A.crec = nullptr; A.sort_id = nullptr; A.cutoffC = 0.0f;
    A.first_idx = first_idx; A.num_pts = num_pts;
    A.counts = nullptr; A.lists = nullptr; A.cap = 0;
    A.queue.tail = nullptr; A.queue.list = nullptr; A.queue.flag = nullptr; A.queue.capq = 0; A.queue_wgs = 0; A.prio = 0;
    A.spill.cursor = nullptr; A.spill.offset = nullptr; A.spill.mask = nullptr; A.spill.ctrl = nullptr; A.spill.fail = nullptr;
    A.spill.pool = nullptr; A.spill.big = nullptr; A.spill.giant = nullptr; A.spill.giant_cap = 0; A.spill.cap_entries = 0;
    A.clean_counts = nullptr;
    A.spill_wgs = 0; A.spill_sorted = 0; A.P = P;   // (the pool pass ran behind the binning: dss_splat_bin)
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
        A.counts = w.counts; A.lists = w.lists; A.cap = w.cap; A.queue = w.queue; A.spill = w.spill;
        A.queue_wgs = queue_workgroups(N, g);
        A.prio = (long long)N * g.tiles_x * g.tiles_y <= 4096;
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
// Workspaces that hold a saved point order (DSS_WS_ORDER_SAVE), host-side bookkeeping at call time: the calls on a workspace
// are stream-ordered, so an order saved by an earlier call is complete when a later call's kernels read it.
// what: 1 = this call saves, 2 = this call reuses (error unless an order of the same N, P, S is registered), 0 = this call
// sorts for itself and overwrites whatever order the workspace held.
namespace dss {
struct SavedOrder { void *ws; int N; int64_t P; int S; };
static int saved_order_update(void *ws, int N, int64_t P, int S, int what)
{
    static std::mutex mu;
    static std::vector<SavedOrder> table;
    std::lock_guard<std::mutex> lock(mu);
    size_t at = table.size();
    for (size_t i = 0; i < table.size(); ++i)
        if (table[i].ws == ws) { at = i; break; }
    if (what == 2) {
        if (at == table.size() || table[at].N != N || table[at].P != P || table[at].S != S) {
            set_error("dss_render_forward: DSS_WS_ORDER_REUSE, but no call with DSS_WS_ORDER_SAVE and the same N, P, S has "
                      "run on this workspace (or a call without either flag has overwritten its order since)");
            return DSS_ERR_INVALID_ARGUMENT;
        }
        return DSS_OK;
    }
    if (what == 1) {
        const SavedOrder e = {ws, N, P, S};
        if (at == table.size()) {
            if (table.size() >= 64) table.erase(table.begin());
            table.push_back(e);
        } else {
            table[at] = e;
        }
    } else if (at != table.size()) {
        table.erase(table.begin() + (long)at);
    }
    return DSS_OK;
}
}  // namespace dss

extern "C" size_t dss_render_forward_workspace(int N, int64_t P, int S, int K)
{
    (void)K;
    if (N <= 0 || P <= 0 || S <= 0) return 256;
    return carve_fwd(nullptr, N, P, S, true).bytes;  // tile structures + one 64-byte record per point
}

extern "C" int dss_render_forward(const float *world, const float *normals, const float *h_point, const float *h_cloud,
                                  const float *vr6, const float *frame_normals, const float *M, const float *V, const float *znear, const float *zfar,
                                  const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int shared_cloud,
                                  int backface_culling, int S, int K, float cutoff_threshold, float antialiasing_sigma,
                                  float merge_thr, int row0, int row1, int row_cycle, const float *feat, int C,
                                  float *pts_screen, float *ellipse, float *radii, float *scaler, float *cutoff,
                                  uint8_t *valid, int32_t *idx, float *zbuf, float *qvalue, float *occ,
                                  uint8_t *visible, float *image, int64_t image_cam_stride, int64_t image_row_stride,
                                  float *wsum, void *workspace, size_t workspace_bytes, int workspace_state,
                                  void *stream)
{
    int rc = validate_fwd("dss_render_forward", N, P, S, K, row0, row1, row_cycle);
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
        !pts_screen || !ellipse || !radii || !scaler || !cutoff || !valid || !idx || !qvalue || !occ ||
        !visible || !image || !wsum) {
        set_error("dss_render_forward: NULL tensor pointer");
        return DSS_ERR_INVALID_ARGUMENT;
    }
    const size_t need = dss_render_forward_workspace(N, P, S, K);
    if (!workspace || workspace_bytes < need) {
        set_error("dss_render_forward: workspace %zu bytes < required %zu", workspace_bytes, need);
        return DSS_ERR_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const TileGrid g = make_grid(S, row0, row1, row_cycle);
    const int tiles = g.tiles_x * g.tiles_y;
    if ((long long)N * tiles > 0x7fffffffll) { set_error("dss_render_forward: too many tiles"); return DSS_ERR_UNSUPPORTED; }
    FwdWorkspace w = carve_fwd(workspace, N, P, S, true);
    const bool packed = C == 3 && !lean_workspace();  // the records carry three feature channels
    const int ws_flags = workspace_state & ~0xf;   // DSS_WS_ORDER_*
    workspace_state &= 0xf;
    if ((unsigned)workspace_state > DSS_WS_BINNED || (ws_flags & ~(DSS_WS_ORDER_SAVE | DSS_WS_ORDER_REUSE | DSS_WS_BAND_OUTPUTS))) {
        set_error("dss_render_forward: unknown workspace_state %d", workspace_state | ws_flags);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    // DSS_WS_BAND_OUTPUTS: only with a real row band (the whole image is reached by every splat that is rendered at all)
    const int band_only = ((ws_flags & DSS_WS_BAND_OUTPUTS) && (g.rows < S || row_cycle > 1)) ? 1 : 0;
    const bool clean = workspace_state == DSS_WS_CLEAN;
    const bool rerun = workspace_state == DSS_WS_BINNED;  // lists + records of this very input are in place: fine pass only
    if (!clean && !rerun && hipMemsetAsync(w.counts, 0, w.count_bytes, st) != hipSuccess)
        return check_launch("memset tile counts");
    SetupArgs SA;
    SA.world = world; SA.normals = normals; SA.h_point = h_point; SA.h_cloud = h_cloud; SA.M = M; SA.V = V;
    SA.vr6 = vr6; SA.frame_n = frame_normals;
    SA.znear = znear; SA.zfar = zfar; SA.first_idx = first_idx; SA.num_pts = num_pts; SA.N = N; SA.P = P;
    SA.shared = shared_cloud; SA.backface = backface_culling; SA.S = S; SA.cutoffC = cutoff_threshold;
    SA.sigma = antialiasing_sigma; SA.screen = pts_screen; SA.ellipse = ellipse; SA.radii = radii; SA.scaler = scaler;
    SA.cutoff = cutoff; SA.valid = valid; SA.rec = packed ? w.rec : nullptr; SA.feat = feat;
    // one wave per workgroup: at DSS sizes (tens of thousands of points) 256-thread groups would occupy only
    // half of the CUs with one wave per SIMD, and this kernel is a chain of dependent latencies
    const int tb = 64;   // (64 / 128 / 256 threads measure the same at 8 x 1M and 4M points: not dispatch-bound there either)
    const int pb = (int)((P + tb - 1) / tb);
    const bool sorted = w.sort_cell_of != nullptr;   // large inputs: cell-ordered binning (see setup_cell_kernel)
    // DSS_WS_ORDER_SAVE / DSS_WS_ORDER_REUSE (cell-ordered path only; no effect on the direct binning of smaller inputs)
    const bool save_order = sorted && !rerun && (ws_flags & DSS_WS_ORDER_SAVE);
    const bool reuse_order = sorted && !rerun && !save_order && (ws_flags & DSS_WS_ORDER_REUSE);
    if (sorted && !rerun) {
        const int rc_o = saved_order_update(workspace, N, P, S, save_order ? 1 : (reuse_order ? 2 : 0));
        if (rc_o) return rc_o;
    }
    if (!rerun && sorted) {
        const SortGrid sg = make_sort_grid(N, S);
        const unsigned sb = (unsigned)sort_blocks(P);
        const int per = sort_per_thread(P);
        const size_t lds = (size_t)sg.total * 4;
        const unsigned bin_wgs = (unsigned)((P + SORT_BIN_CHUNK - 1) / SORT_BIN_CHUNK);
        if (reuse_order) {
            // the point order an earlier call left in the workspace: setup in natural order + one gathered binning pass
            // (no histogram in this mode: 256-thread workgroups of four splats per thread -- a 1024-thread workgroup at ~100
            // VGPRs is ONE resident workgroup per CU, 16 of its 20 wavefront slots, each walking 15 splats one after the other)
            const int per2 = 4, tb2 = 256;
            const unsigned sb2 = (unsigned)((P + (int64_t)per2 * tb2 - 1) / ((int64_t)per2 * tb2));
            hipLaunchKernelGGL(setup_cell_kernel<2>, dim3(sb2), dim3(tb2), (tb2 / 64) * 4096, st, SA, sg, per2, w.sort_cell_of,
                               w.sort_block_hist, w.spill, visible, w.sort_crec, w.sort_inv, g, band_only, w.sort_live);
            hipLaunchKernelGGL(bin_sorted_kernel<true>, dim3(bin_wgs), dim3(SORT_BIN_THREADS), 0, st, w.sort_crec, w.sort_id,
                               w.sort_count, first_idx, num_pts, N, g, w.counts, w.lists, w.cap, w.queue, w.spill, (uint32_t)P,
                               band_only ? w.sort_live : (uint8_t *)nullptr);
        } else {
            // (the position bytes of the band ranks' binning start lowered with every saved order)
            if (save_order && hipMemsetAsync(w.sort_live, 0, (size_t)P, st) != hipSuccess) return check_launch("dss_render_forward (order bytes)");
            if (save_order)
                hipLaunchKernelGGL(setup_cell_kernel<1>, dim3(sb), dim3(SORT_THREADS), lds, st, SA, sg, per, w.sort_cell_of,
                                   w.sort_block_hist, w.spill, visible, w.sort_crec, w.sort_inv, g, band_only);
            else
                hipLaunchKernelGGL(setup_cell_kernel<0>, dim3(sb), dim3(SORT_THREADS), lds, st, SA, sg, per, w.sort_cell_of,
                                   w.sort_block_hist, w.spill, visible, w.sort_crec, w.sort_inv, g, band_only);
            const unsigned nseg = (sb + SORT_SEG - 1) / SORT_SEG;
            hipLaunchKernelGGL(sort_block_scan_kernel, dim3((unsigned)((sg.total + 255) / 256), nseg), dim3(256), 0, st, sb,
                               sg.total, w.sort_block_hist, w.sort_seg_tot);
            hipLaunchKernelGGL(sort_seg_scan_kernel, dim3((unsigned)((sg.total + 255) / 256)), dim3(256), 0, st, nseg, sg.total,
                               w.sort_seg_tot, w.sort_cell_total);
            hipLaunchKernelGGL(sort_cell_scan_kernel, dim3(1), dim3(1024), 0, st, w.sort_cell_total, w.sort_cell_start, sg.total,
                               w.sort_count);
            hipLaunchKernelGGL(sort_scatter_kernel, dim3(sb), dim3(SORT_THREADS), lds, st, pts_screen, radii, ellipse, P, sg, per,
                               w.sort_cell_of, w.sort_cell_start, w.sort_block_hist, w.sort_seg_tot, w.sort_crec, w.sort_id,
                               save_order ? w.sort_inv : (int32_t *)nullptr);
            hipLaunchKernelGGL(bin_sorted_kernel<false>, dim3(bin_wgs), dim3(SORT_BIN_THREADS), 0, st, w.sort_crec, w.sort_id,
                               w.sort_count, first_idx, num_pts, N, g, w.counts, w.lists, w.cap, w.queue, w.spill, (uint32_t)P);
        }
        hipLaunchKernelGGL(queue_build_kernel, dim3((unsigned)((N * tiles + 1023) / 1024)), dim3(1024), 0, st, w.counts,
                           N * tiles, g, w.queue);
    } else if (!rerun) {
        if (band_only)
            hipLaunchKernelGGL(setup_bin_kernel<true>, dim3(pb), dim3(tb), 0, st, SA, g, w.counts, w.lists, w.cap, w.queue, w.spill,
                               visible);
        else
            hipLaunchKernelGGL(setup_bin_kernel<false>, dim3(pb), dim3(tb), 0, st, SA, g, w.counts, w.lists, w.cap, w.queue,
                               w.spill, visible);
    }
    FineArgs A;
    A.points = pts_screen; A.ellipse = ellipse; A.cutoff = cutoff; A.radii = radii;
    A.rec = packed ? w.rec : nullptr;
    A.crec = sorted ? w.sort_crec : nullptr;   // cell-ordered path: list entries are positions of the point order
    A.sort_id = sorted ? w.sort_id : nullptr;
    A.cutoffC = cutoff_threshold;
    A.first_idx = first_idx; A.num_pts = num_pts;
    A.counts = w.counts; A.lists = w.lists; A.cap = w.cap; A.queue = w.queue; A.spill = w.spill;
    A.queue_wgs = queue_workgroups(N, g);
    A.prio = (long long)N * tiles <= 4096;
    A.clean_counts = clean ? w.counts : nullptr;
    // the pool pass of the sub-list overflow rides at the front of the fine launch (round 3 launched it on its own: 4.7 us
    // per step for a pass that is empty in the common case); a fine-only re-run finds the pool filled
    A.spill_wgs = rerun ? 0u : spill_workgroups_in_fine(P);
    A.spill_sorted = sorted ? 1 : 0;
    A.P = P;
    A.idx = idx; A.zbuf = zbuf; A.qv = qvalue; A.occ = occ; A.visible = visible;
    A.g = g; A.N = N; A.K = K; A.thr = merge_thr;
    A.scaler = scaler; A.feat = feat; A.image = image; A.wsum = wsum; A.C = C;
    A.img_sn = image_cam_stride > 0 ? image_cam_stride : (long long)g.rows * S * (C + 1);
    A.img_sr = image_row_stride > 0 ? image_row_stride : (long long)S * (C + 1);
    if (!dispatch_fine(A, N * tiles, st)) { set_error("dss_render_forward: no kernel for K=%d", K); return DSS_ERR_UNSUPPORTED; }
    return check_launch("dss_render_forward");
}
