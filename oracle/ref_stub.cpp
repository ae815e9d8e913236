// TEST INFRASTRUCTURE ONLY (oracle/): not part of the shipped product path.
//
// Two-symbol stub that lets the *unmodified* reference CPU rasterizer
// (/root/reference/DSS/csrc/ext.cpp + rasterize_points_cpu.cpp) link without CUDA.
// rasterize_points.h:176-203 and :268-285 reference the *Cuda entry points without a
// WITH_CUDA guard, so the CPU-only build needs them defined; they are never reached
// with CPU tensors.
#include <torch/extension.h>
#include <tuple>

torch::Tensor RasterizePointsCoarseCuda(
    const torch::Tensor &, const torch::Tensor &, const torch::Tensor &,
    const torch::Tensor &, const int, const int, const int)
{
    AT_ERROR("oracle/_ref is a CPU-only build of the reference");
}

std::tuple<torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor>
RasterizePointsFineCuda(
    const torch::Tensor &, const torch::Tensor &, const torch::Tensor &,
    const torch::Tensor &, const torch::Tensor &, const float, const int,
    const int, const int)
{
    AT_ERROR("oracle/_ref is a CPU-only build of the reference");
}
