// v_mfma_f32_16x16x4_f32 issue rate by operand placement (gfx950), one wave per SIMD, four accumulators in rotation:
//   acc VGPR / A VGPR / B VGPR      acc VGPR / A VGPR / B AGPR (what large_trsm_pipe issues)      acc AGPR / A VGPR / B VGPR
//   acc AGPR / A VGPR / B AGPR
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_operands.hip -o /tmp/mfma_ops && /tmp/mfma_ops
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

#define REP16(X) X X X X X X X X X X X X X X X X

template <int MODE, bool RANDOM = false> __global__ __launch_bounds__(256, 1) void k(float *out, unsigned long long *cyc, int iters)
{
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float a = 1.0f + threadIdx.x * 1e-6f, b = 1e-9f;
        asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a255");
        asm volatile("v_accvgpr_write_b32 a16, %0\n s_nop 4" ::"v"(b));
        if (MODE >= 2)
                asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %0\n v_accvgpr_write_b32 a2, %0\n v_accvgpr_write_b32 a3, %0\n"
                             "v_accvgpr_write_b32 a4, %0\n v_accvgpr_write_b32 a5, %0\n v_accvgpr_write_b32 a6, %0\n v_accvgpr_write_b32 a7, %0\n"
                             "v_accvgpr_write_b32 a8, %0\n v_accvgpr_write_b32 a9, %0\n v_accvgpr_write_b32 a10, %0\n v_accvgpr_write_b32 a11, %0\n"
                             "v_accvgpr_write_b32 a12, %0\n v_accvgpr_write_b32 a13, %0\n v_accvgpr_write_b32 a14, %0\n v_accvgpr_write_b32 a15, %0\n s_nop 4" ::"v"(0.0f));
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
        {
                if (RANDOM)
                {
                        // fresh mantissa bits in both operands every 64 MFMAs (power is data-dependent: constant operands read high)
                        const unsigned h = (unsigned)(i * 0x9E3779B1u) ^ (threadIdx.x * 0x85EBCA6Bu);
                        a = __int_as_float(0x3f800000 | (h & 0x007fffff));
                        const float bb = __int_as_float(0x3f000000 | ((h >> 3) & 0x007fffff));
                        asm volatile("v_accvgpr_write_b32 a16, %0\n s_nop 4" ::"v"(bb));
                }
                if (MODE == 0)
                        asm volatile(REP16("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                                           "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n")
                                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                                     : "v"(a), "v"(b));
                else if (MODE == 1)
                        asm volatile(REP16("v_mfma_f32_16x16x4_f32 %0, %4, a16, %0\n v_mfma_f32_16x16x4_f32 %1, %4, a16, %1\n"
                                           "v_mfma_f32_16x16x4_f32 %2, %4, a16, %2\n v_mfma_f32_16x16x4_f32 %3, %4, a16, %3\n")
                                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                                     : "v"(a));
                else if (MODE == 2)
                        asm volatile(REP16("v_mfma_f32_16x16x4_f32 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x4_f32 a[4:7], %0, %1, a[4:7]\n"
                                           "v_mfma_f32_16x16x4_f32 a[8:11], %0, %1, a[8:11]\n v_mfma_f32_16x16x4_f32 a[12:15], %0, %1, a[12:15]\n")
                                     :
                                     : "v"(a), "v"(b));
                else
                        asm volatile(REP16("v_mfma_f32_16x16x4_f32 a[0:3], %0, a16, a[0:3]\n v_mfma_f32_16x16x4_f32 a[4:7], %0, a16, a[4:7]\n"
                                           "v_mfma_f32_16x16x4_f32 a[8:11], %0, a16, a[8:11]\n v_mfma_f32_16x16x4_f32 a[12:15], %0, a16, a[12:15]\n")
                                     :
                                     : "v"(a));
        }
        asm volatile("s_nop 15");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = c0[0] + c1[1] + c2[2] + c3[3];
        if (MODE >= 2)
        {
                float x;
                asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(x));
                s += x;
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0)
                cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, bool RANDOM = false> void run(const char *name, int blocks, int iters = 2000)
{
        float *out;
        unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * 256);
        hipMalloc(&cyc, 8 * blocks * 4);
        hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[4];
        hipMemcpy(c, cyc, 32, hipMemcpyDeviceToHost);
        const double n = (double)iters * 64;
        std::printf("%-34s %4d blocks: %6.1f memtime ticks per MFMA per wave, %7.3f ms, %6.1f TFLOP/s chip-wide\n", name, blocks, (double)c[0] / n, ms,
                    (double)blocks * 4 * n * 2048 / (ms * 1e-3) / 1e12);
        hipFree(out);
        hipFree(cyc);
}

int main()
{
        // sustained load: the same loop 5x, 50x and 250x longer (1.7 ms -> 9 ms -> 90 ms -> 430 ms per launch), every CU busy: does the chip hold 2.4 GHz?
        for (int iters : {2000, 10000, 100000, 500000})
        {
                std::printf("iters %d: ", iters);
                run<3>("acc AGPR, A VGPR, B AGPR", 256, iters);
                std::printf("iters %d: ", iters);
                run<3, true>("same, random operand mantissas", 256, iters);
        }
        for (int blocks : {1, 256})
        {
                run<0>("acc VGPR, A VGPR, B VGPR", blocks);
                run<1>("acc VGPR, A VGPR, B AGPR", blocks);
                run<2>("acc AGPR, A VGPR, B VGPR", blocks);
                run<3>("acc AGPR, A VGPR, B AGPR", blocks);
        }
        return 0;
}
