// Developer tool (GPU box): issue rate of the VALU instructions the fine / gather kernels are made of.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o build_ab/valu_rate && build_ab/valu_rate
// One wavefront per SIMD (grid = CUs * 4 waves, 64-thread workgroups would not pin SIMDs, so 256-thread workgroups) and
// 8 per SIMD; every body is 16 independent instructions in an asm block, repeated ITER times.
// Measured (round 3): 4.1-4.2 cycles per wave-instruction per SIMD for compares, selects, integer, DPP, 64-bit and packed-f32
// operations; 2.3 for v_mul/add/fma_f32 with >= 2 waves per SIMD (4.5 with one).  A run of v_cndmask_b32 that read a VCC no
// VALU instruction wrote measures 23 cycles each here; rewriting the fine kernel's selects to SGPR-pair masks changed
// nothing in a same-run A/B, so treat that line as an artefact of this loop, not as a property of compiled code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096
#define BODY16(x) x x x x x x x x x x x x x x x x
#define BODY8(x) x x x x x x x x
#define BODY7(x) x x x x x x x

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned *out, unsigned seed)
{
    unsigned a = threadIdx.x * seed, b = a ^ 0x9e3779b9u, c = a + 17u, d = b + 3u, e = 0, f = 0;
    unsigned long long s = 0;
    for (int i = 0; i < ITER; ++i) {
        if (OP == 0) asm volatile(BODY16("v_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(e) : "v"(a), "v"(b) : "vcc");
        if (OP == 1) asm volatile(BODY16("v_cmp_lt_u64 vcc, %[p], %[q]\n") : : [p] "v"(((unsigned long long)a << 32) | b), [q] "v"(((unsigned long long)c << 32) | d) : "vcc");
        if (OP == 2) asm volatile(BODY16("v_cmp_lt_u32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
        if (OP == 3) asm volatile(BODY16("v_mul_f32 %0, %1, %2\n") : "+v"(e) : "v"(a), "v"(b));
        if (OP == 4) asm volatile(BODY16("v_pk_mul_f32 %0, %1, %2\n") : "+v"(s) : "v"(((unsigned long long)a << 32) | b), "v"(((unsigned long long)c << 32) | d));
        if (OP == 5) asm volatile(BODY16("v_min_u32 %0, %1, %2\n") : "+v"(e) : "v"(a), "v"(b));
        if (OP == 6) asm volatile(BODY16("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") : "+v"(e) : "v"(a));
        if (OP == 7) asm volatile(BODY16("v_cmp_lt_u64 %0, %[p], %[q]\n") : "=s"(s) : [p] "v"(((unsigned long long)a << 32) | b), [q] "v"(((unsigned long long)c << 32) | d));
        if (OP == 8) asm volatile(BODY16("v_cmp_gt_f32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
        if (OP == 9) asm volatile(BODY16("v_fma_f32 %0, %1, %2, %0\n") : "+v"(e) : "v"(a), "v"(b));
        if (OP == 10) asm volatile(BODY16("v_med3_f32 %0, %1, %2, %3\n") : "+v"(e) : "v"(a), "v"(b), "v"(c));
        if (OP == 11) asm volatile(BODY16("v_lshl_add_u64 %0, %1, 0, %2\n") : "+v"(s) : "v"(((unsigned long long)a << 32) | b), "v"(((unsigned long long)c << 32) | d));
        if (OP == 12) asm volatile(BODY16("v_max3_u32 %0, %1, %2, %3\n") : "+v"(e) : "v"(a), "v"(b), "v"(c));
        if (OP == 13) asm volatile(BODY16("v_add_f32 %0, %1, %2\n") : "+v"(e) : "v"(a), "v"(b));
        if (OP == 14) asm volatile(BODY16("v_cndmask_b32 %0, %1, %2, %3\n") : "+v"(e) : "v"(a), "v"(b), "s"(s));
        if (OP == 15) asm volatile(BODY16("v_cmp_class_f32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
        // compare + select pairs (16 instructions = 8 pairs): through VCC and through an SGPR pair
        if (OP == 16) asm volatile(BODY8("v_cmp_lt_u32 vcc, %1, %2\nv_cndmask_b32 %0, %1, %2, vcc\n") : "+v"(e) : "v"(a), "v"(b) : "vcc");
        if (OP == 17) asm volatile(BODY8("v_cmp_lt_u32 %1, %2, %3\nv_cndmask_b32 %0, %2, %3, %1\n") : "+v"(e), "+s"(s) : "v"(a), "v"(b));
        // one compare, seven selects on its result
        if (OP == 18) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n" BODY7("v_cndmask_b32 %0, %1, %2, vcc\n") "v_cmp_lt_u32 vcc, %2, %1\n" BODY7("v_cndmask_b32 %0, %2, %1, vcc\n") : "+v"(e) : "v"(a), "v"(b) : "vcc");
    }
    if (e == 0x12345u || f == 7u || s == 99ull) out[0] = e;
}

template <int OP>
static void run(const char *name, unsigned *out, int cus)
{
    for (int wps = 1; wps <= 8; wps *= 2) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int grid = cus * wps;   // 256-thread workgroups: 4 waves, one per SIMD
        rate_kernel<OP><<<grid, 256>>>(out, 3u);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) rate_kernel<OP><<<grid, 256>>>(out, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        const double instr_per_simd = (double)ITER * 16 * wps;
        printf("%-28s waves/SIMD %d  %.3f ms  %.2f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    unsigned *out;
    hipMalloc(&out, 64);
    const int cus = p.multiProcessorCount;
    printf("CUs %d clock %d kHz\n", cus, p.clockRate);
    run<0>("v_cndmask_b32 (vcc)", out, cus);
    run<14>("v_cndmask_b32 (sgpr pair)", out, cus);
    run<16>("8 x (v_cmp vcc; v_cndmask vcc)", out, cus);
    run<17>("8 x (v_cmp sgpr; v_cndmask sgpr)", out, cus);
    run<18>("2 x (v_cmp vcc; 7 v_cndmask vcc)", out, cus);
    run<1>("v_cmp_lt_u64 -> vcc", out, cus);
    run<7>("v_cmp_lt_u64 -> sgpr", out, cus);
    run<2>("v_cmp_lt_u32", out, cus);
    run<8>("v_cmp_gt_f32", out, cus);
    run<3>("v_mul_f32", out, cus);
    run<13>("v_add_f32", out, cus);
    run<9>("v_fma_f32", out, cus);
    run<4>("v_pk_mul_f32", out, cus);
    run<5>("v_min_u32", out, cus);
    run<12>("v_max3_u32", out, cus);
    run<10>("v_med3_f32", out, cus);
    run<6>("v_mov_b32_dpp quad_perm", out, cus);
    run<11>("v_lshl_add_u64", out, cus);
    return 0;
}
