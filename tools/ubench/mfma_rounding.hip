// How do the MFMA accumulators round?  (round 3: the fp32 covariance error of the large path grows ~linearly over the first few hundred
// callbacks, which a zero-mean rounding error would not do.)
//
// For v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x4_f32:  D = C + sum_k a_k b_k  with C = +-1 and a single product p = f * 2^-24
// (ulp(1) = 2^-23 above 1, 2^-24 below), f = 0.25 .. 1.75: round-to-nearest-even gives 1 for f < 1, 1 + 2^-23 for f > 1; truncation gives 1
// for all f < 2.  Then the same with the product spread over several k (is the sum of products formed exactly before the one rounding?), and a
// statistical test: 4096 random accumulations, mean signed error in ulps against the exact sum (0 for round-to-nearest, -0.5 * sign for truncation).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rounding.hip -o /tmp/mfma_rounding && /tmp/mfma_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

static unsigned short to_bf16(float x) // exact for the values used here (<= 8 significant bits)
{
        unsigned u;
        memcpy(&u, &x, 4);
        return (unsigned short)(u >> 16);
}

// one wave; A, B: [16][32] bf16 (row, k), C, D: [16][16] float
__global__ void mm_bf16(const unsigned short *A, const unsigned short *B, const float *C, float *D)
{
        const int l = threadIdx.x, i = l & 15, g = l >> 4;
        bf8 a, b;
        for (int e = 0; e < 8; ++e)
        {
                a[e] = __builtin_bit_cast(__bf16, A[i * 32 + 8 * g + e]);
                b[e] = __builtin_bit_cast(__bf16, B[i * 32 + 8 * g + e]);
        }
        f4 c;
        for (int r = 0; r < 4; ++r)
                c[r] = C[(4 * g + r) * 16 + i];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r)
                D[(4 * g + r) * 16 + i] = c[r];
}

// A, B: [16][4] float
__global__ void mm_f32(const float *A, const float *B, const float *C, float *D)
{
        const int l = threadIdx.x, i = l & 15, g = l >> 4;
        f4 c;
        for (int r = 0; r < 4; ++r)
                c[r] = C[(4 * g + r) * 16 + i];
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + g], B[i * 4 + g], c, 0, 0, 0);
        for (int r = 0; r < 4; ++r)
                D[(4 * g + r) * 16 + i] = c[r];
}

template <typename T> T *dev(const std::vector<T> &h)
{
        T *p;
        hipMalloc(&p, h.size() * sizeof(T));
        hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
        return p;
}

int main()
{
        const float u = ldexpf(1.f, -24);
        // ---- (1) single product against C = +-1: element (row m, col n) of D uses A row m, B row n
        printf("single product p = f * 2^-24 added to C (RNE: 1 -> 1 + 2^-23 at f > 1;  -1 + p: ulp is 2^-24 there, exact for multiples of 1)\n");
        const float fs[] = {0.25f, 0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f, 2.0f, 2.5f, 3.0f};
        for (int sign = 1; sign >= -1; sign -= 2)
        {
                std::vector<unsigned short> A(16 * 32, 0), B(16 * 32, 0);
                std::vector<float> Af(16 * 4, 0.f), Bf(16 * 4, 0.f), C(256, (float)sign), D(256), Df(256);
                // row m of A: a_0 = fs[m] * 2^-12 * sign, B row n: b_0 = 2^-12  -> product fs[m] 2^-24 (sign as C: magnitude grows)
                for (int m = 0; m < 10; ++m)
                {
                        A[m * 32] = to_bf16(sign * fs[m] * ldexpf(1.f, -12));
                        Af[m * 4] = sign * fs[m] * ldexpf(1.f, -12);
                }
                for (int n = 0; n < 16; ++n)
                {
                        B[n * 32] = to_bf16(ldexpf(1.f, -12));
                        Bf[n * 4] = ldexpf(1.f, -12);
                }
                auto dA = dev(A), dB = dev(B);
                auto dAf = dev(Af), dBf = dev(Bf), dC = dev(C), dD = dev(D);
                mm_bf16<<<1, 64>>>(dA, dB, dC, dD);
                hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
                mm_f32<<<1, 64>>>(dAf, dBf, dC, dD);
                hipMemcpy(Df.data(), dD, 1024, hipMemcpyDeviceToHost);
                for (int m = 0; m < 10; ++m)
                        printf("  C = %+d, f = %4.2f: bf16 MFMA D - C = %5.2f x 2^-24   f32 MFMA D - C = %5.2f x 2^-24   (RNE: %5.2f)\n", sign, fs[m],
                               (double)(D[m * 16] - sign) / u * sign, (double)(Df[m * 16] - sign) / u * sign,
                               (double)((float)((double)sign + (double)sign * fs[m] * u) - sign) / u * sign);
        }
        // ---- (2) several sub-ulp products: 3 x 0.5 * 2^-24 spread over k = 0, 8, 16 (different lane groups) and k = 0, 1, 2 (same lane)
        {
                std::vector<unsigned short> A(16 * 32, 0), B(16 * 32, 0);
                std::vector<float> C(256, 1.f), D(256);
                const int ks[2][3] = {{0, 8, 16}, {0, 1, 2}};
                for (int m = 0; m < 2; ++m)
                        for (int q = 0; q < 3; ++q)
                                A[m * 32 + ks[m][q]] = to_bf16(0.5f * ldexpf(1.f, -12));
                for (int n = 0; n < 16; ++n)
                        for (int k = 0; k < 32; ++k)
                                B[n * 32 + k] = to_bf16(ldexpf(1.f, -12));
                auto dA = dev(A), dB = dev(B);
                auto dC = dev(C), dD = dev(D);
                mm_bf16<<<1, 64>>>(dA, dB, dC, dD);
                hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
                printf("three products of 0.5 x 2^-24 (exact sum 1.5 x 2^-24 -> RNE 2): across lane groups D - 1 = %.2f x 2^-24, inside one lane %.2f x 2^-24\n",
                       (double)(D[0] - 1.f) / u, (double)(D[16] - 1.f) / u);
        }
        // ---- (3) statistics: C random in [1, 2), 32 random bf16 products of magnitude ~2^-6 each (sum ~ 0.1): signed error in ulps of the result
        {
                std::mt19937 rng(7);
                std::uniform_real_distribution<float> uc(1.f, 2.f), ua(-1.f, 1.f);
                double sum_bf = 0, sum_f32 = 0, abs_bf = 0, abs_f32 = 0;
                int cnt = 0;
                for (int rep = 0; rep < 16; ++rep)
                {
                        std::vector<unsigned short> A(16 * 32), B(16 * 32);
                        std::vector<float> Aq(16 * 32), Bq(16 * 32), Af(16 * 4), Bf(16 * 4), C(256), D(256), Df(256);
                        for (int i = 0; i < 16 * 32; ++i)
                        {
                                A[i] = to_bf16(ua(rng) * 0.25f) , B[i] = to_bf16(ua(rng) * 0.25f);
                                unsigned x = (unsigned)A[i] << 16, y = (unsigned)B[i] << 16;
                                memcpy(&Aq[i], &x, 4), memcpy(&Bq[i], &y, 4);
                        }
                        for (int i = 0; i < 64; ++i)
                                Af[i] = ua(rng) * 0.5f, Bf[i] = ua(rng) * 0.5f;
                        for (auto &c : C)
                                c = uc(rng) * (rep & 1 ? -1.f : 1.f);
                        auto dA = dev(A), dB = dev(B);
                        auto dAf = dev(Af), dBf = dev(Bf), dC = dev(C), dD = dev(D);
                        mm_bf16<<<1, 64>>>(dA, dB, dC, dD);
                        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
                        mm_f32<<<1, 64>>>(dAf, dBf, dC, dD);
                        hipMemcpy(Df.data(), dD, 1024, hipMemcpyDeviceToHost);
                        for (int m = 0; m < 16; ++m)
                                for (int n = 0; n < 16; ++n)
                                {
                                        double e = C[m * 16 + n], ef = C[m * 16 + n];
                                        for (int k = 0; k < 32; ++k)
                                                e += (double)Aq[m * 32 + k] * (double)Bq[n * 32 + k];
                                        for (int k = 0; k < 4; ++k)
                                                ef += (double)Af[m * 4 + k] * (double)Bf[n * 4 + k];
                                        const double ulp = ldexp(1.0, ilogb(fabs(e)) - 23), ulpf = ldexp(1.0, ilogb(fabs(ef)) - 23);
                                        const double sg = e > 0 ? 1 : -1;
                                        sum_bf += sg * (D[m * 16 + n] - e) / ulp, abs_bf += fabs(D[m * 16 + n] - e) / ulp;
                                        sum_f32 += sg * (Df[m * 16 + n] - ef) / ulpf, abs_f32 += fabs(Df[m * 16 + n] - ef) / ulpf;
                                        ++cnt;
                                }
                }
                printf("random accumulations (%d): mean error TOWARDS LARGER MAGNITUDE in ulps / mean |error|:  bf16 16x16x32 %+.4f / %.4f    f32 16x16x4 %+.4f / %.4f\n",
                       cnt, sum_bf / cnt, abs_bf / cnt, sum_f32 / cnt, abs_f32 / cnt);
                printf("  (round to nearest: 0 / 0.25;  truncation of the sum: -0.5 / 0.5)\n");
        }
        return 0;
}
