// fp32 x fp32 products on the bf16 matrix pipe: a = a1 + a2 + a3 (three bf16 pieces of 8 mantissa bits each, exact), and
// a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1): six v_mfma_f32_16x16x32_bf16 per 16x16x32 product (16 cycles each)
// against eight v_mfma_f32_16x16x4_f32 (32 cycles each): 2.7x the rate at ~1.2e-7 relative per product (the dropped terms).
// Checks the operand layout (A: row l & 15, k = 8 (l >> 4) .. + 7; B likewise; D: row 4 (l >> 4) + r, column l & 15) and the error
// against a double-precision product, then times both forms.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16x3.hip -o /tmp/mfma_bf16x3 && /tmp/mfma_bf16x3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float a, unsigned &h, unsigned &m, unsigned &l)
{
        // round to nearest at every level (truncation biases the dropped terms: see large_syrk_bf16x3)
        const unsigned u = (__float_as_uint(a) + 0x8000u) & 0xffff0000u;
        const float r1 = a - __uint_as_float(u);
        const unsigned u1 = (__float_as_uint(r1) + 0x8000u) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(u1);
        h = u >> 16, m = u1 >> 16, l = (__float_as_uint(r2) + 0x8000u) >> 16;
}

__device__ __forceinline__ void split8(const float *p, u4 &h, u4 &m, u4 &l)
{
        unsigned hh[8], mm[8], ll[8];
        for (int e = 0; e < 8; ++e)
                split3(p[e], hh[e], mm[e], ll[e]);
        for (int q = 0; q < 4; ++q)
        {
                h[q] = hh[2 * q] | (hh[2 * q + 1] << 16);
                m[q] = mm[2 * q] | (mm[2 * q + 1] << 16);
                l[q] = ll[2 * q] | (ll[2 * q + 1] << 16);
        }
}

__global__ void check(const float *A, const float *B, float *C) // A, B: [16][32]; C = A B^T [16][16]
{
        const int l = threadIdx.x, i = l & 15, g = l >> 4;
        u4 a1, a2, a3, b1, b2, b3;
        split8(A + i * 32 + 8 * g, a1, a2, a3);
        split8(B + i * 32 + 8 * g, b1, b2, b3);
        f4 c = {0, 0, 0, 0};
#define MM(x, y) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0)
        MM(a3, b1);
        MM(a2, b2);
        MM(a1, b3);
        MM(a2, b1);
        MM(a1, b2);
        MM(a1, b1);
        for (int r = 0; r < 4; ++r)
                C[(4 * g + r) * 16 + i] = c[r];
}

template <int MODE> __global__ __launch_bounds__(256) void rate(float *out, int iters)
{
        f4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        u4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
        const float af = 1.0f + threadIdx.x * 1e-6f, bf = 1e-9f;
        for (int it = 0; it < iters; ++it)
        {
                if (MODE == 0) // eight f32 MFMAs = one 16x16x32 product, four accumulators in rotation
                        for (int s = 0; s < 8; ++s)
                                for (int t = 0; t < 4; ++t)
                                        c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, c[t], 0, 0, 0);
                else // six bf16 MFMAs = the same product
                        for (int s = 0; s < 6; ++s)
                                for (int t = 0; t < 4; ++t)
                                        c[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c[t], 0, 0, 0);
        }
        out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
}

int main()
{
        std::mt19937 rng(3);
        std::normal_distribution<float> nd;
        std::vector<float> A(16 * 32), B(16 * 32), C(256);
        for (auto &x : A)
                x = nd(rng);
        for (auto &x : B)
                x = nd(rng) * 0.01f;
        float *dA, *dB, *dC;
        hipMalloc(&dA, 2048), hipMalloc(&dB, 2048), hipMalloc(&dC, 1024);
        hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double worst = 0, worst32 = 0;
        for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j)
                {
                        double s = 0, sabs = 0;
                        float s32 = 0;
                        for (int k = 0; k < 32; ++k)
                        {
                                s += (double)A[i * 32 + k] * (double)B[j * 32 + k];
                                sabs += std::fabs((double)A[i * 32 + k] * (double)B[j * 32 + k]);
                                s32 = fmaf(A[i * 32 + k], B[j * 32 + k], s32);
                        }
                        worst = std::fmax(worst, std::fabs(C[i * 16 + j] - s) / sabs);
                        worst32 = std::fmax(worst32, std::fabs((double)s32 - s) / sabs);
                }
        std::printf("bf16x3 (6 MFMAs): max |C - exact| / sum |a b| = %.2e   (fp32 fma chain: %.2e)\n", worst, worst32);
        float *out;
        hipMalloc(&out, 4 * 256 * 256);
        for (int mode = 0; mode < 2; ++mode)
        {
                hipEvent_t e0, e1;
                hipEventCreate(&e0), hipEventCreate(&e1);
                const int iters = 20000;
                if (mode == 0)
                        hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, 10);
                else
                        hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, 10);
                hipEventRecord(e0);
                if (mode == 0)
                        hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, iters);
                else
                        hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double prod = (double)iters * 4 * 4 * 256; // 16x16x32 products: iters x 4 accumulators x 4 waves x 256 blocks
                std::printf("%s: %.3f ms = %.1f T(fp32-equivalent FLOP)/s chip-wide\n", mode == 0 ? "8 x v_mfma_f32_16x16x4_f32  " : "6 x v_mfma_f32_16x16x32_bf16", ms,
                            prod * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12);
        }
        return 0;
}
