// f64 MFMA / f64 FMA issue-rate and latency microbenchmark (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_mfma(double *out, unsigned long long *cyc, int iters, double a0, double b0)
{
        d4 acc[NACC];
        for (int q = 0; q < NACC; ++q) acc[q] = (d4){0, 0, 0, 0};
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
        {
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
        for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
__global__ void k_fma(double *out, unsigned long long *cyc, int iters, double a0, double b0)
{
        double acc[NACC];
        for (int q = 0; q < NACC; ++q) acc[q] = q;
        double a = a0 + threadIdx.x * 1e-9, b = b0;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i)
        {
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = fma(a, acc[q], b);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
        for (int q = 0; q < NACC; ++q) s += acc[q];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <typename F> void run(const char *name, F launch, int blocks, int threads, int iters, int ops_per_iter, double flop_per_op)
{
        double *out; unsigned long long *cyc;
        hipMalloc(&out, sizeof(double) * blocks * threads); hipMalloc(&cyc, 8 * blocks * (threads / 64));
        launch(out, cyc); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); launch(out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[64]; hipMemcpy(c, cyc, 8 * (threads / 64), hipMemcpyDeviceToHost);
        double per = (double)c[0] / ((double)iters * ops_per_iter);
        double tf = (double)blocks * (threads / 64) * iters * ops_per_iter * flop_per_op / (ms * 1e-3) / 1e12;
        printf("%-44s blocks %4d thr %4d: %7.1f cycles per op per wave (wave 0), %.3f ms, %.1f TFLOP/s chip-wide\n", name, blocks, threads, per, ms, tf);
        hipFree(out); hipFree(cyc);
}
int main()
{
        const int it = 20000;
#define L(K, N) [&](double *o, unsigned long long *c) { hipLaunchKernelGGL(K<N>, dim3(B), dim3(T), 0, 0, o, c, it, 1.0000001, 1e-9); }
        int B, T;
        B = 1; T = 64;   run("mfma f64 16x16x4, 1 wave, 1 acc (dep chain)", L(k_mfma, 1), B, T, it, 1, 2048);
        B = 1; T = 64;   run("mfma f64 16x16x4, 1 wave, 2 acc", L(k_mfma, 2), B, T, it, 2, 2048);
        B = 1; T = 64;   run("mfma f64 16x16x4, 1 wave, 4 acc", L(k_mfma, 4), B, T, it, 4, 2048);
        B = 1; T = 64;   run("mfma f64 16x16x4, 1 wave, 8 acc", L(k_mfma, 8), B, T, it, 8, 2048);
        B = 1; T = 256;  run("mfma f64, 4 waves (1/SIMD), 4 acc", L(k_mfma, 4), B, T, it, 4, 2048);
        B = 1; T = 768;  run("mfma f64, 12 waves (3/SIMD), 1 acc", L(k_mfma, 1), B, T, it, 1, 2048);
        B = 1; T = 768;  run("mfma f64, 12 waves (3/SIMD), 4 acc", L(k_mfma, 4), B, T, it, 4, 2048);
        B = 256; T = 256; run("mfma f64, 256 blocks x 4 waves, 4 acc", L(k_mfma, 4), B, T, it, 4, 2048);
        B = 256; T = 768; run("mfma f64, 256 blocks x 12 waves, 4 acc", L(k_mfma, 4), B, T, it, 4, 2048);
        B = 1; T = 64;   run("v_fma_f64, 1 wave, 1 acc (dep chain)", L(k_fma, 1), B, T, it, 1, 128);
        B = 1; T = 64;   run("v_fma_f64, 1 wave, 8 acc", L(k_fma, 8), B, T, it, 8, 128);
        B = 256; T = 1024; run("v_fma_f64, 256 blocks x 16 waves, 8 acc", L(k_fma, 8), B, T, it, 8, 128);
        return 0;
}
