// Multi-GPU row bands: putting all-gathered rows in image order (no counterpart in the reference, which has no distributed
// layer; the image it assembles with torch.cat at renderer.py:75-78 is here assembled across ranks).
#include "common.h"

namespace dss {

// dst[n][r][0..W) = src[row_pos[r]][n][0..W)  -- the all-gather of (band row, camera, ...) send buffers leaves the rows of
// rank g at positions g * band .. (dss_amd/distributed.py RowPartition.gather_index: unequal bands travel padded, a
// tile-row-cyclic partition interleaves the ranks' tile rows); the consumers (the loss, the owned windows of
// dss_render_backward_owned_plane) want dense (N, S, ...) images.  One 16-byte piece per thread and trip, rows of a camera
// contiguous on both sides: HBM-bound copy.
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T *__restrict__ src, const int32_t *__restrict__ row_pos,
                                                          int N, int rows, int W, T *__restrict__ dst)
{
    const int r = blockIdx.x, n = blockIdx.y;
    const T *__restrict__ s = src + ((size_t)row_pos[r] * N + n) * (size_t)W;
    T *__restrict__ d = dst + ((size_t)n * rows + r) * (size_t)W;
    for (int i = threadIdx.x; i < W; i += 256) d[i] = s[i];
}

}  // namespace dss

extern "C" int dss_gather_rows(const float *src, const int32_t *row_pos, int N, int rows, int row_floats, float *dst,
                               void *stream)
{
    using namespace dss;
    if (N <= 0 || rows < 0 || row_floats <= 0 || !src || !row_pos || !dst) {
        set_error("dss_gather_rows: bad arguments N=%d rows=%d row_floats=%d", N, rows, row_floats);
        return DSS_ERR_INVALID_ARGUMENT;
    }
    if (rows == 0) return DSS_OK;
    hipStream_t st = as_stream(stream);
    if (row_floats % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15u) == 0)
        hipLaunchKernelGGL(gather_rows_kernel<float4>, dim3((unsigned)rows, (unsigned)N), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(src), row_pos, N, rows, row_floats / 4, reinterpret_cast<float4 *>(dst));
    else
        hipLaunchKernelGGL(gather_rows_kernel<float>, dim3((unsigned)rows, (unsigned)N), dim3(256), 0, st, src, row_pos, N, rows,
                           row_floats, dst);
    return check_launch("dss_gather_rows");
}
