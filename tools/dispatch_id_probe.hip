#include <hip/hip_runtime.h>
#include <cstdio>
extern "C" __device__ unsigned long long dispatch_id(void) __asm("llvm.amdgcn.dispatch.id");
__global__ void k(unsigned long long *out)
{
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = dispatch_id();
        out[2 * blockIdx.x + 1] = (unsigned long long)__builtin_amdgcn_queue_ptr();
    }
}
int main()
{
    unsigned long long *d, h[8];
    hipMalloc(&d, sizeof(h));
    hipStream_t st; hipStreamCreate(&st);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, it == 2 ? st : 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("launch %d: ids %llu %llu %llu %llu queue %llx\n", it, h[0], h[2], h[4], h[6], h[1]);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, st, d);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int it = 0; it < 3; ++it) {
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("replay %d: ids %llu %llu queue %llx\n", it, h[0], h[2], h[1]);
    }
    return 0;
}
