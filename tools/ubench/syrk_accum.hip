// Accumulation schemes for P -= V V^T on the bf16 matrix pipe (round 3): error and BIAS of a K = 1024 product of binary32 rows, formed as six
// v_mfma_f32_16x16x32_bf16 per 32 columns (large_syrk_bf16x3), against the exact (binary64) product of the same binary32 inputs.
//   scheme 0   the six MFMAs of every slab accumulate into the one running fp32 accumulator           (round 2)
//   scheme 1   the six MFMAs of a slab accumulate into a zero-initialised temporary, acc += temp (VALU, fp32) once per slab
//   scheme 2   like 1, but the temporary collects F slabs (F = 2, 4) before it is added
//   scheme 3   like 1 with the slab sum added in binary64 (acc is a double)
//   scheme 4   eight v_mfma_f32_16x16x4_f32 per 32 columns into the running accumulator                (the fp32 pipe)
// Each wave computes a 16x16 tile A B^T with A = B (the tile's diagonal = sums of squares, its off-diagonal = mixed signs) from rows drawn
// N(0, sigma) with a decaying or flat column profile.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/syrk_accum.hip -o tools/ubench/syrk_accum && tools/ubench/syrk_accum
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
        const unsigned u = (__float_as_uint(a) + 0x7fffu + ((__float_as_uint(a) >> 16) & 1u)) & 0xffff0000u; // round to nearest even
        const float r1 = a - __uint_as_float(u);
        const unsigned u1 = (__float_as_uint(r1) + 0x7fffu + ((__float_as_uint(r1) >> 16) & 1u)) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(u1);
        h = u >> 16, m = u1 >> 16, l = (__float_as_uint(r2) + 0x7fffu + ((__float_as_uint(r2) >> 16) & 1u)) >> 16;
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

// V: [tiles][16][K]; out: [tiles][16][16] double
__global__ void tile(const float *V, double *out, int K, int scheme, int F)
{
        const int l = threadIdx.x, i = l & 15, g = l >> 4;
        const float *row = V + ((size_t)blockIdx.x * 16 + i) * K;
        f4 acc = {0, 0, 0, 0}, tmp = {0, 0, 0, 0};
        double acc64[4] = {0, 0, 0, 0};
        int pend = 0;
        for (int kc = 0; kc < K; kc += 32)
        {
                if (scheme == 4)
                {
                        for (int c = 0; c < 8; ++c)
                        {
                                const float a = row[kc + 4 * c + g];
                                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc, 0, 0, 0);
                        }
                        continue;
                }
                u4 a1, a2, a3;
                split8(row + kc + 8 * g, a1, a2, a3);
#define MM(c, x, y) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, x), __builtin_bit_cast(bf8, y), c, 0, 0, 0)
                if (scheme == 0)
                {
                        MM(acc, a1, a3);
                        MM(acc, a2, a2);
                        MM(acc, a3, a1);
                        MM(acc, a1, a2);
                        MM(acc, a2, a1);
                        MM(acc, a1, a1);
                }
                else
                {
                        MM(tmp, a1, a3);
                        MM(tmp, a2, a2);
                        MM(tmp, a3, a1);
                        MM(tmp, a1, a2);
                        MM(tmp, a2, a1);
                        MM(tmp, a1, a1);
                        if (++pend == F || kc + 32 >= K)
                        {
                                for (int r = 0; r < 4; ++r)
                                {
                                        if (scheme == 3)
                                                acc64[r] += (double)tmp[r];
                                        else
                                                acc[r] += tmp[r];
                                }
                                tmp = (f4){0, 0, 0, 0};
                                pend = 0;
                        }
                }
        }
        for (int r = 0; r < 4; ++r)
                out[((size_t)blockIdx.x * 16 + 4 * g + r) * 16 + i] = scheme == 3 ? acc64[r] : (double)acc[r];
}

int main()
{
        const int K = 1024, T = 512;
        for (int profile = 0; profile < 2; ++profile)
        {
                std::mt19937 rng(11);
                std::normal_distribution<float> nd(0.f, 0.05f);
                std::vector<float> V((size_t)T * 16 * K);
                for (size_t t = 0; t < (size_t)T * 16; ++t)
                        for (int k = 0; k < K; ++k)
                                V[t * K + k] = nd(rng) * (profile ? expf(-3.f * k / K) : 1.f);
                std::vector<double> ex((size_t)T * 256);
                for (int t = 0; t < T; ++t)
                        for (int m = 0; m < 16; ++m)
                                for (int n = 0; n < 16; ++n)
                                {
                                        double s = 0;
                                        for (int k = 0; k < K; ++k)
                                                s += (double)V[((size_t)t * 16 + m) * K + k] * (double)V[((size_t)t * 16 + n) * K + k];
                                        ex[((size_t)t * 16 + m) * 16 + n] = s;
                                }
                float *dV;
                double *dO;
                hipMalloc(&dV, V.size() * 4);
                hipMalloc(&dO, ex.size() * 8);
                hipMemcpy(dV, V.data(), V.size() * 4, hipMemcpyHostToDevice);
                printf("column profile: %s; errors relative to the mean |diagonal entry| of the product\n", profile ? "decaying e^(-3k/K)" : "flat");
                struct
                {
                        int scheme, F;
                        const char *name;
                } cases[] = {{0, 1, "0  running fp32 accumulator (round 2)"}, {1, 1, "1  per-slab temporary, fp32 add"}, {2, 2, "2  temporary over 2 slabs"},
                             {2, 4, "2  temporary over 4 slabs"},            {2, 8, "2  temporary over 8 slabs"},      {3, 1, "3  per-slab temporary, fp64 add"},
                             {4, 1, "4  fp32 MFMA 16x16x4 chain"}};
                for (auto &cs : cases)
                {
                        tile<<<T, 64>>>(dV, dO, K, cs.scheme, cs.F);
                        std::vector<double> o(ex.size());
                        hipMemcpy(o.data(), dO, o.size() * 8, hipMemcpyDeviceToHost);
                        double dscale = 0, bd = 0, ad = 0, bo = 0, ao = 0, mx = 0;
                        for (int t = 0; t < T; ++t)
                                for (int m = 0; m < 16; ++m)
                                        dscale += ex[((size_t)t * 16 + m) * 16 + m];
                        dscale /= T * 16;
                        for (size_t idx = 0; idx < ex.size(); ++idx)
                        {
                                const int m = (idx / 16) % 16, n = idx % 16;
                                const double e = (o[idx] - ex[idx]) / dscale;
                                if (m == n)
                                        bd += e, ad += fabs(e);
                                else
                                        bo += e * (ex[idx] > 0 ? 1 : -1), ao += fabs(e);
                                mx = fmax(mx, fabs(e));
                        }
                        const double nd_ = T * 16.0, no_ = T * 240.0;
                        printf("  %-40s diagonal: bias %+.2e  mean|e| %.2e   off-diagonal: bias %+.2e  mean|e| %.2e   max %.2e\n", cs.name, bd / nd_, ad / nd_, bo / no_,
                               ao / no_, mx);
                }
                hipFree(dV);
                hipFree(dO);
        }
        return 0;
}
