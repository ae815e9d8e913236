// Developer tool (GPU box): calibration of the TCC FETCH_SIZE counter on the access patterns of this library's kernels.
// MI355X_MICROARCH.md calibrates "x2" only for wide coalesced streaming reads and calls other widths uncalibrated; the
// backward gather reads 4..20-byte pieces at indirected addresses, so a ratio "counter traffic / algorithmic bytes" quoted
// with the streaming correction may be off in either direction.  Every kernel here reads a KNOWN set of
// addresses from a buffer far larger than the L2s (no reuse), and the host prints, per kernel, the bytes requested and the
// bytes of the distinct 32 / 64 / 128-byte granules touched; tools/fetch_calibration.py runs this under
// `rocprofv3 --pmc FETCH_SIZE` and reports which of them the counter follows and with what factor.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o /tmp/fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unordered_set>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// every thread reads one 16-byte vector, consecutive (wide coalesced stream)
__global__ void cal_stream16(const float4 *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 v = in[i]; if (v.x == 123.456f) out[0] = v.y + v.z + v.w; }
}
// every thread reads one dword, consecutive
__global__ void cal_stream4(const float *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = in[i]; if (v == 123.456f) out[0] = v; }
}
// every thread reads one 20-byte fragment record (16 + 4 bytes, 4-byte aligned) at an indirected position: the K = 5 ids of
// a pixel in the backward gather
struct __attribute__((packed, aligned(4))) Frag4 { int a, b, c, d; };
__global__ void cal_gather20(const int *__restrict__ in, const uint32_t *__restrict__ rec, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const char *p = reinterpret_cast<const char *>(in) + (size_t)rec[i] * 20u;
    const Frag4 v = *reinterpret_cast<const Frag4 *>(p);
    const int w = *reinterpret_cast<const int *>(p + 16);
    if (v.a == 123456789) out[0] = (float)(v.b + v.c + v.d + w);
}
// 16 lanes read 16 consecutive dwords of an indirected row (the alpha-plane window rows of the gather)
__global__ void cal_rows16(const float *__restrict__ in, const uint32_t *__restrict__ row, float *__restrict__ out, size_t n_rows)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = t >> 4;
    if (r >= n_rows) return;
    const float v = in[(size_t)row[r] * 16u + (t & 15u)];
    if (v == 123.456f) out[0] = v;
}
// every thread reads one 64-byte splat record as four 16-byte loads at an indirected record (fine pass staging)
__global__ void cal_rec64(const float4 *__restrict__ in, const uint32_t *__restrict__ rec, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 *p = in + (size_t)rec[i] * 4u;
    const float4 a = p[0], b = p[1], c = p[3];
    if (a.x == 123.456f) out[0] = b.y + c.z;
}

static void granules(const std::vector<std::pair<uint64_t, uint32_t>> &acc, uint64_t out[3])
{
    const int sh[3] = {5, 6, 7};
    for (int g = 0; g < 3; ++g) {
        std::unordered_set<uint64_t> s;
        s.reserve(acc.size() * 2);
        for (auto &a : acc)
            for (uint64_t x = a.first >> sh[g]; x <= (a.first + a.second - 1) >> sh[g]; ++x) s.insert(x);
        out[g] = (uint64_t)s.size() << sh[g];
    }
}

int main()
{
    const size_t BUF = (size_t)2 << 30;   // 2 GiB: far beyond 8 x 4 MB of L2 + 256 MB of MALL
    char *buf; float *out;
    CHECK(hipMalloc(&buf, BUF)); CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(buf, 0, BUF));
    srand(7);
    auto rnd = [](uint64_t m) { return (uint64_t)(((uint64_t)rand() << 31) ^ (uint64_t)rand()) % m; };
    const size_t N = 4u << 20;            // accesses per kernel
    std::vector<uint32_t> h(N);
    uint32_t *d_idx; CHECK(hipMalloc(&d_idx, N * 4));
    uint64_t g[3];
    // 1, 2: streams (granules = requested)
    hipLaunchKernelGGL(cal_stream16, dim3((unsigned)(N / 256)), dim3(256), 0, 0, (const float4 *)buf, out, N);
    printf("cal_stream16 requested %llu g32 %llu g64 %llu g128 %llu\n", (unsigned long long)N * 16, (unsigned long long)N * 16, (unsigned long long)N * 16, (unsigned long long)N * 16);
    hipLaunchKernelGGL(cal_stream4, dim3((unsigned)(4 * N / 256)), dim3(256), 0, 0, (const float *)(buf + (BUF >> 1)), out, 4 * N);
    printf("cal_stream4 requested %llu g32 %llu g64 %llu g128 %llu\n", (unsigned long long)N * 16, (unsigned long long)N * 16, (unsigned long long)N * 16, (unsigned long long)N * 16);
    // 3: 20-byte records at random positions
    {
        const uint64_t recs = BUF / 20 - 1;
        std::vector<std::pair<uint64_t, uint32_t>> acc(N);
        for (size_t i = 0; i < N; ++i) { h[i] = (uint32_t)rnd(recs); acc[i] = {(uint64_t)h[i] * 20, 20u}; }
        CHECK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cal_gather20, dim3((unsigned)(N / 256)), dim3(256), 0, 0, (const int *)buf, d_idx, out, N);
        granules(acc, g);
        printf("cal_gather20 requested %llu g32 %llu g64 %llu g128 %llu\n", (unsigned long long)N * 20, (unsigned long long)g[0], (unsigned long long)g[1], (unsigned long long)g[2]);
    }
    // 4: rows of 16 dwords at random 64-byte-aligned positions
    {
        const size_t R = N / 4;
        const uint64_t rows = BUF / 64 - 1;
        std::vector<std::pair<uint64_t, uint32_t>> acc(R);
        for (size_t i = 0; i < R; ++i) { h[i] = (uint32_t)rnd(rows); acc[i] = {(uint64_t)h[i] * 64, 64u}; }
        CHECK(hipMemcpy(d_idx, h.data(), R * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cal_rows16, dim3((unsigned)(R * 16 / 256)), dim3(256), 0, 0, (const float *)buf, d_idx, out, R);
        granules(acc, g);
        printf("cal_rows16 requested %llu g32 %llu g64 %llu g128 %llu\n", (unsigned long long)R * 64, (unsigned long long)g[0], (unsigned long long)g[1], (unsigned long long)g[2]);
    }
    // 5: 64-byte records, 48 of their bytes read, random positions
    {
        const uint64_t recs = BUF / 64 - 1;
        std::vector<std::pair<uint64_t, uint32_t>> acc;
        acc.reserve(2 * N);
        for (size_t i = 0; i < N; ++i) { h[i] = (uint32_t)rnd(recs); acc.push_back({(uint64_t)h[i] * 64, 32u}); acc.push_back({(uint64_t)h[i] * 64 + 48, 16u}); }
        CHECK(hipMemcpy(d_idx, h.data(), N * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cal_rec64, dim3((unsigned)(N / 256)), dim3(256), 0, 0, (const float4 *)buf, d_idx, out, N);
        granules(acc, g);
        printf("cal_rec64 requested %llu g32 %llu g64 %llu g128 %llu\n", (unsigned long long)N * 48, (unsigned long long)g[0], (unsigned long long)g[1], (unsigned long long)g[2]);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
