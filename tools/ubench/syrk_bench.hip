// syrk_bench.hip -- stand-alone correctness + timing harness for large_syrk_bf16x3 (P -= V V^T, ekf_large.h), round 4.
// Random V (binary32, n = 1027, columns n .. zero), P = 0: the kernel's P against a binary64 host product of the same V on sampled rows (the diagonal and
// the pose columns / rows are not the kernel's: large_x_update_rows forms those), exact symmetry, and timings of the diagnostic variants:
//   DIAG 1 = the K loop without the read-modify-write of P, 4 = without the split + LDS stash (stale LDS: timing only), 8 = without the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I awesomeslam_amd/csrc tools/ubench/syrk_bench.hip -o tools/ubench/syrk_bench && tools/ubench/syrk_bench [filters]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ekf_large.h"

using namespace aslam;

#define CK(x)                                                                                                          \
        do                                                                                                             \
        {                                                                                                              \
                hipError_t e_ = (x);                                                                                   \
                if (e_ != hipSuccess)                                                                                  \
                {                                                                                                      \
                        std::printf("%s: %s\n", #x, hipGetErrorString(e_));                                            \
                        std::exit(1);                                                                                  \
                }                                                                                                      \
        } while (0)

template <typename F> float time_ms(F launch, int reps)
{
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r)
                launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
}

/// binary32 rows -> three bf16 planes, columns permuted inside every 64-block (LPlanes / lplane_pos): what the closing blocks of large_trsm_bf16 store.
/// grid (NP / 16, B), 256 threads: 16 rows per workgroup, thread = (row tid >> 4, four columns 4 (tid & 15) of every block)
__global__ void split_v_planes(const float *G, LPlanes pl, int NP)
{
        const int b = blockIdx.y, r = 16 * blockIdx.x + (threadIdx.x >> 4), c = 4 * (threadIdx.x & 15);
        const float *src = G + ((size_t)b * NP + r) * NP;
        unsigned short *dst = pl.Lq(b, NP) + (size_t)r * NP;
        for (int j = 0; j < NP / 64; ++j)
        {
                const f4 v = *reinterpret_cast<const f4 *>(src + 64 * j + c);
                unsigned h0, m0, l0, h1, m1, l1;
                t16::split2(v[0], v[1], h0, m0, l0);
                t16::split2(v[2], v[3], h1, m1, l1);
                unsigned short *q = dst + 64 * j + lplane_pos(c);
                *reinterpret_cast<t16::u2v *>(q) = (t16::u2v){h0, h1};
                *reinterpret_cast<t16::u2v *>(q + (size_t)NP * NP) = (t16::u2v){m0, m1};
                *reinterpret_cast<t16::u2v *>(q + 2 * (size_t)NP * NP) = (t16::u2v){l0, l1};
        }
}

int main(int argc, char **argv)
{
        const int B = argc > 1 ? std::atoi(argv[1]) : 256;
        const int n = 1027, NP = 1088;
        const size_t M = (size_t)NP * NP;
        std::mt19937 rng(11);
        std::normal_distribution<float> nd;
        std::vector<float> V(M, 0.f);
        for (int i = 0; i <= n; ++i) // (row n = q rides along, as in the chain)
                for (int j = 0; j < n; ++j)
                        V[(size_t)i * NP + j] = nd(rng) * (1.0f + 0.01f * (float)(j % 7));
        float *dG;
        double *dP;
        int *dn, *dskip;
        CK(hipMalloc(&dG, sizeof(float) * M * B));
        CK(hipMalloc(&dP, sizeof(double) * M * B));
        CK(hipMalloc(&dn, sizeof(int) * B));
        CK(hipMalloc(&dskip, sizeof(int) * B));
        CK(hipMemset(dskip, 0, sizeof(int) * B));
        CK(hipMemset(dP, 0, sizeof(double) * M * B));
        for (int b = 0; b < B; ++b)
                CK(hipMemcpy(dG + M * b, V.data(), sizeof(float) * M, hipMemcpyHostToDevice));
        std::vector<int> nn(B, n);
        CK(hipMemcpy(dn, nn.data(), sizeof(int) * B, hipMemcpyHostToDevice));
        DevView d = {};
        d.B = B, d.NP = NP, d.n = dn;
        LargeView<float> lv = {};
        lv.NP = NP, lv.P = dP, lv.G = dG;
        const int ntile = (NP + 127) / 128;
        const dim3 grid(8 * (ntile * (ntile + 1) / 2) * ((B + 7) / 8));
        // ---- correctness: one launch on P = 0 -> P = -V V^T (lower + mirror), filters 0 and B - 1
        hipLaunchKernelGGL((large_syrk_bf16x3<0>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip);
        CK(hipDeviceSynchronize());
        std::vector<double> P(M), P0(M);
        CK(hipMemcpy(P0.data(), dP, sizeof(double) * M, hipMemcpyDeviceToHost));
        CK(hipMemcpy(P.data(), dP + M * (B - 1), sizeof(double) * M, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0, asym = 0;
        size_t differ = 0;
        for (size_t i = 0; i < M; ++i)
                differ += P[i] != P0[i];
        for (int i = 3; i < n; i += 13)
                for (int j = 3; j < n; ++j)
                {
                        if (i == j)
                                continue;
                        double s = 0, sa = 0;
                        for (int k = 0; k < n; ++k)
                        {
                                const double t = (double)V[(size_t)i * NP + k] * (double)V[(size_t)j * NP + k];
                                s += t, sa += std::fabs(t);
                        }
                        worst = std::fmax(worst, std::fabs(-s - P[(size_t)i * NP + j]) / sa);
                        scale = std::fmax(scale, sa);
                        asym = std::fmax(asym, std::fabs(P[(size_t)i * NP + j] - P[(size_t)j * NP + i]));
                }
        double untouched = 0;
        for (int i = 0; i < n; ++i)
                untouched = std::fmax(untouched, std::fmax(std::fabs(P[(size_t)i * NP + i]), std::fmax(std::fabs(P[(size_t)i * NP + std::min(i, 2)]), std::fabs(P[(size_t)std::min(i, 2) * NP + i]))));
        std::printf("syrk_bf16x3: max |P + V V^T| / sum |v v| = %.2e (rows 3, 16, ... of filter %d), asymmetry %.1e, entries of the pose columns / rows / diagonal written: %.1e, "
                    "filter %d differs from filter 0 in %zu entries\n", worst, B - 1, asym, untouched, B - 1, differ);
        const double sf = (ntile * (ntile + 1) / 2 - ntile * 0.25) * 2.0 * 128 * 128 * 1056 * B;
        const float m0 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<0>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
        std::printf("  syrk_bf16x3                                   %8.3f ms for %d filters = %6.1f T fp32-equivalent FLOP/s executed\n", m0, B, sf / (m0 * 1e-3) / 1e12);
        const float m1 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
        std::printf("  K loop only (no read-modify-write of P)       %8.3f ms\n", m1);
        const float m5 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<5>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
        std::printf("  K loop without the split + LDS stash          %8.3f ms\n", m5);
        const float m9 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<9>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
        std::printf("  K loop without the MFMAs                      %8.3f ms\n", m9);
        const float m13 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<13>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
        std::printf("  K loop: fetch + barriers only                 %8.3f ms\n", m13);
        {
                const float a = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1 | 4 | 16>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  K loop: fetch + barriers + MFMAs (no stash, no operand reads)   %8.3f ms\n", a);
                const float b2 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1 | 32>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  K loop without the barriers (racy)                             %8.3f ms\n", b2);
                const float c2 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1 | 4 | 32>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  K loop without stash and barriers                              %8.3f ms\n", c2);
                const float e2 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1 | 4 | 16 | 32>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  K loop: fetch + MFMAs only (no stash, reads, barriers)         %8.3f ms\n", e2);
        }
        // ---- V as bf16 planes (what large_trsm_bf16 stores in the chain), streamed by LDS-DMA
        LPlanes vpl = {};
        CK(hipMalloc(&vpl.base, sizeof(unsigned short) * LPlanes::per_filter(NP) * B));
        CK(hipMemset(vpl.base, 0, sizeof(unsigned short) * LPlanes::per_filter(NP) * B));
        hipLaunchKernelGGL(split_v_planes, dim3(NP / 16, B), dim3(256), 0, 0, dG, vpl, NP);
        CK(hipMemset(dP, 0, sizeof(double) * M * B));
        hipLaunchKernelGGL((large_syrk_bf16x3<0, 1>), grid, dim3(256), 0, 0, d, lv, vpl, B, dskip);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(P0.data(), dP, sizeof(double) * M, hipMemcpyDeviceToHost));
        CK(hipMemcpy(P.data(), dP + M * (B - 1), sizeof(double) * M, hipMemcpyDeviceToHost));
        worst = 0, asym = 0, differ = 0;
        for (size_t i = 0; i < M; ++i)
                differ += P[i] != P0[i];
        for (int i = 3; i < n; i += 13)
                for (int j = 3; j < n; ++j)
                {
                        if (i == j)
                                continue;
                        double s = 0, sa = 0;
                        for (int k = 0; k < n; ++k)
                        {
                                const double t = (double)V[(size_t)i * NP + k] * (double)V[(size_t)j * NP + k];
                                s += t, sa += std::fabs(t);
                        }
                        worst = std::fmax(worst, std::fabs(-s - P[(size_t)i * NP + j]) / sa);
                        asym = std::fmax(asym, std::fabs(P[(size_t)i * NP + j] - P[(size_t)j * NP + i]));
                }
        std::printf("syrk_bf16x3<PL>: max |P + V V^T| / sum |v v| = %.2e, asymmetry %.1e, filter %d differs from filter 0 in %zu entries\n", worst, asym, B - 1, differ);
        const float p0 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<0, 1>), grid, dim3(256), 0, 0, d, lv, vpl, B, dskip); }, 5);
        std::printf("  syrk_bf16x3<PL>                               %8.3f ms for %d filters = %6.1f T fp32-equivalent FLOP/s executed\n", p0, B, sf / (p0 * 1e-3) / 1e12);
        const float p1 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1, 1>), grid, dim3(256), 0, 0, d, lv, vpl, B, dskip); }, 5);
        std::printf("  <PL> K loop only                              %8.3f ms\n", p1);
        const float p9 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<9, 1>), grid, dim3(256), 0, 0, d, lv, vpl, B, dskip); }, 5);
        std::printf("  <PL> K loop without the MFMAs                 %8.3f ms\n", p9);
        return 0;
}
