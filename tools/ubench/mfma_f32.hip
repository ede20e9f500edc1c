// f32 MFMA issue-rate microbenchmark (gfx950): 16x16x4 and 32x32x2, plus an LDS-fed variant
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NACC> __global__ void k16(float *out, unsigned long long *cyc, int iters, float a0, float b0)
{
        f4 acc[NACC];
        for (int q = 0; q < NACC; ++q) acc[q] = (f4){0, 0, 0, 0};
        float a = a0 + threadIdx.x * 1e-6f, b = b0;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
        {
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC> __global__ void k32(float *out, unsigned long long *cyc, int iters, float a0, float b0)
{
        f16v acc[NACC];
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0;
        float a = a0 + threadIdx.x * 1e-6f, b = b0;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
        {
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <typename F> void run(const char *name, F launch, int blocks, int threads, int iters, int ops_per_iter, double flop_per_op)
{
        float *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, 8 * blocks * (threads / 64));
        launch(out, cyc); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); launch(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[64]; hipMemcpy(c, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
        double per = (double)c[0] / ((double)iters * ops_per_iter);
        double tf = (double)blocks * (threads / 64) * iters * ops_per_iter * flop_per_op / (ms * 1e-3) / 1e12;
        printf("%-44s blocks %4d thr %4d: %7.1f memtime ticks per op per wave, %.3f ms, %.1f TFLOP/s chip-wide\n", name, blocks, threads, per, ms, tf);
        hipFree(out); hipFree(cyc);
}
int main()
{
        const int it = 20000;
#define L(K, N) [&](float *o, unsigned long long *c) { hipLaunchKernelGGL(K<N>, dim3(B), dim3(T), 0, 0, o, c, it, 1.0000001f, 1e-9f); }
        int B, T;
        B = 1; T = 64;    run("mfma f32 16x16x4, 1 wave, 1 acc (dep chain)", L(k16, 1), B, T, it, 1, 2048);
        B = 1; T = 64;    run("mfma f32 16x16x4, 1 wave, 4 acc", L(k16, 4), B, T, it, 4, 2048);
        B = 1; T = 64;    run("mfma f32 16x16x4, 1 wave, 16 acc", L(k16, 16), B, T, it, 16, 2048);
        B = 256; T = 256; run("mfma f32 16x16x4, 256 x 4 waves, 16 acc", L(k16, 16), B, T, it, 16, 2048);
        B = 256; T = 768; run("mfma f32 16x16x4, 256 x 12 waves, 16 acc", L(k16, 16), B, T, it, 16, 2048);
        B = 1; T = 64;    run("mfma f32 32x32x2, 1 wave, 1 acc (dep chain)", L(k32, 1), B, T, it, 1, 4096);
        B = 1; T = 64;    run("mfma f32 32x32x2, 1 wave, 4 acc", L(k32, 4), B, T, it, 4, 4096);
        B = 256; T = 256; run("mfma f32 32x32x2, 256 x 4 waves, 4 acc", L(k32, 4), B, T, it, 4, 4096);
        B = 256; T = 768; run("mfma f32 32x32x2, 256 x 12 waves, 4 acc", L(k32, 4), B, T, it, 4, 4096);
        return 0;
}
