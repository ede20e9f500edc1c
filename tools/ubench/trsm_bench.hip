// trsm_bench.hip -- stand-alone correctness + timing harness for the kernels of the large-state EKF's binary32 chain
// (ekf_large_trsm.h, ekf_large_chol.h, large_syrk_bf16x3 in ekf_large.h), round 2.
// Random SPD S -> host Cholesky (double) -> L and the inverses of its 64x64 diagonal blocks in binary32; random G.  Checks:
// large_trsm_pipe's V against a host triangular solve, large_chol_resident's L and Linv against the host factor.  Timings: the product
// kernels at several batch sizes, and diagnostic variants of large_trsm_pipe with one part of the block pipeline removed, stamped inside
// the kernel with s_memtime / s_memrealtime (cycles per MFMA and wave).  DIAG bits of large_trsm_pipe: 1 = no global fetch of the L
// blocks (stale LDS), 2 = no LDS stash and no synchronisation, 16 = no synchronisation (racy), 32 = every fetch reads one block (L1 hits:
// separates the issue cost of the fetch from its latency), 8 = stamps.  large_chol_resident<17, 1>: phase stamps.  large_syrk_bf16x3<1>:
// the K loop without the read-modify-write of P.  The numbers quoted in DESIGN.md and profiles/r02_experiments.md come from this program.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I awesomeslam_amd/csrc tools/ubench/trsm_bench.hip -o /tmp/trsm_bench && /tmp/trsm_bench [filters]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "ekf_large.h"
#include "trsm16_right_looking.h"

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

int main(int argc, char **argv)
{
        const int B = argc > 1 ? std::atoi(argv[1]) : 128;
        const int n = 1027, NP = 1088, NB = NP / LB;
        const size_t M = (size_t)NP * NP;
        std::mt19937 rng(7);
        std::normal_distribution<double> nd;
        // one filter's data on the host, replicated on the device
        std::vector<double> S(M, 0.0), L(M, 0.0);
        {
                std::vector<double> A((size_t)n * 48);
                for (auto &x : A)
                        x = nd(rng) * 0.05;
                for (int i = 0; i < n; ++i)
                        for (int j = 0; j <= i; ++j)
                        {
                                double s = (i == j) ? 0.2 : 0.0;
                                for (int k = 0; k < 48; ++k)
                                        s += A[(size_t)i * 48 + k] * A[(size_t)j * 48 + k];
                                S[(size_t)i * NP + j] = s;
                        }
                for (int i = n; i < NP; ++i)
                        S[(size_t)i * NP + i] = 1.0;
                for (int j = 0; j < NP; ++j)
                {
                        double d = S[(size_t)j * NP + j];
                        for (int k = 0; k < j; ++k)
                                d -= L[(size_t)j * NP + k] * L[(size_t)j * NP + k];
                        d = std::sqrt(d);
                        L[(size_t)j * NP + j] = d;
                        for (int i = j + 1; i < NP; ++i)
                        {
                                double s = S[(size_t)i * NP + j];
                                for (int k = 0; k < j; ++k)
                                        s -= L[(size_t)i * NP + k] * L[(size_t)j * NP + k];
                                L[(size_t)i * NP + j] = s / d;
                        }
                }
        }
        std::vector<float> Lf(M), Linv((size_t)NB * LB * LB, 0.f), G(M);
        for (size_t i = 0; i < M; ++i)
                Lf[i] = (float)L[i];
        for (int k = 0; k < NB; ++k)
        {
                // inverse of the lower-triangular diagonal block (double, forward substitution on the identity)
                std::vector<double> X((size_t)LB * LB, 0.0);
                for (int c = 0; c < LB; ++c)
                        for (int i = c; i < LB; ++i)
                        {
                                double s = (i == c) ? 1.0 : 0.0;
                                for (int q = c; q < i; ++q)
                                        s -= L[(size_t)(64 * k + i) * NP + 64 * k + q] * X[(size_t)q * LB + c];
                                X[(size_t)i * LB + c] = s / L[(size_t)(64 * k + i) * NP + 64 * k + i];
                        }
                for (int i = 0; i < LB * LB; ++i)
                        Linv[(size_t)k * LB * LB + i] = (float)X[i];
        }
        for (auto &x : G)
                x = (float)nd(rng);
        // device buffers
        float *dS, *dG, *dG0, *dLinv;
        double *dP;
        int *dn, *dskip;
        CK(hipMalloc(&dS, sizeof(float) * M * B));
        CK(hipMalloc(&dG, sizeof(float) * M * B));
        CK(hipMalloc(&dG0, sizeof(float) * M));
        CK(hipMalloc(&dP, sizeof(double) * M * B));
        CK(hipMalloc(&dLinv, sizeof(float) * NB * LB * LB * B));
        CK(hipMalloc(&dn, sizeof(int) * B));
        CK(hipMalloc(&dskip, sizeof(int) * B));
        CK(hipMemset(dskip, 0, sizeof(int) * B));
        CK(hipMemset(dP, 0, sizeof(double) * M * B));
        CK(hipMemcpy(dG0, G.data(), sizeof(float) * M, hipMemcpyHostToDevice));
        std::vector<int> nn(B, n);
        CK(hipMemcpy(dn, nn.data(), sizeof(int) * B, hipMemcpyHostToDevice));
        for (int b = 0; b < B; ++b)
        {
                CK(hipMemcpy(dS + M * b, Lf.data(), sizeof(float) * M, hipMemcpyHostToDevice));
                CK(hipMemcpy(dLinv + (size_t)NB * LB * LB * b, Linv.data(), sizeof(float) * NB * LB * LB, hipMemcpyHostToDevice));
        }
        DevView d = {};
        d.B = B;
        d.NP = NP;
        d.n = dn;
        LargeView<float> lv = {};
        lv.NP = NP;
        lv.P = dP;
        lv.G = dG;
        lv.S = dS;
        lv.Linv = dLinv;
        auto reset_G = [&]() {
                for (int b = 0; b < B; ++b)
                        CK(hipMemcpyAsync(dG + M * b, dG0, sizeof(float) * M, hipMemcpyDeviceToDevice, 0));
        };
        // ---- correctness of the product kernel on the last filter
        reset_G();
        hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 0>), dim3(8 * ((B + 7) / 8) * NB), dim3(256), 0, 0, d, lv, B, dskip);
        CK(hipDeviceSynchronize());
        std::vector<float> V(M);
        CK(hipMemcpy(V.data(), dG + M * (B - 1), sizeof(float) * M, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int i = 0; i < NP; i += 37)
        {
                // row i of V = G L^-T in double from the float L
                std::vector<double> v(NP);
                for (int c = 0; c < NP; ++c)
                {
                        double s = G[(size_t)i * NP + c];
                        for (int q = 0; q < c; ++q)
                                s -= v[q] * (double)Lf[(size_t)c * NP + q];
                        v[c] = s / (double)Lf[(size_t)c * NP + c];
                }
                for (int c = 0; c < NP; ++c)
                {
                        worst = std::fmax(worst, std::fabs(v[c] - (double)V[(size_t)i * NP + c]));
                        scale = std::fmax(scale, std::fabs(v[c]));
                }
        }
        std::printf("large_trsm_pipe: max |V - host| / max |V| = %.2e (rows 0, 37, ... of filter %d)\n", worst / scale, B - 1);
        // ---- the bf16-pipe sweep (ekf_large_trsm16.h): planes of L from large_split_planes, then the same check
        const char *only = std::getenv("ONLY"); // diagnostics: ONLY=trsm16 stops after this section, ONLY=chol16 skips it
        t16::Planes pl = {};
        if (!(only && std::string(only) == "chol16"))
        {
                CK(hipMalloc(&pl.base, sizeof(unsigned short) * t16::Planes::per_filter(NP) * B));
                CK(hipMemset(pl.base, 0, sizeof(unsigned short) * t16::Planes::per_filter(NP) * B));
                const float msp = time_ms([&]() { hipLaunchKernelGGL(large_split_planes<0>, dim3(NB + 1, B), dim3(256), 0, 0, d, lv, pl, dskip); }, 3);
                reset_G();
                const bool rl = std::getenv("RIGHT") != nullptr; // RIGHT=1: the right-looking experiment (trsm16_right_looking.h) in this section instead of the library's kernel
                auto k16 = rl ? large_trsm_bf16r<LARGE_NB_MAX, 0> : large_trsm_bf16<LARGE_NB_MAX, 0>;
                auto k16s = rl ? large_trsm_bf16r<LARGE_NB_MAX, 1> : large_trsm_bf16<LARGE_NB_MAX, 1>;
                std::printf("bf16 sweep: %s\n", rl ? "right-looking (large_trsm_bf16r)" : "left-looking (large_trsm_bf16)");
                hipLaunchKernelGGL(k16, dim3(8 * ((B + 7) / 8) * NB), dim3(256), 0, 0, d, lv, pl, B, dskip);
                CK(hipDeviceSynchronize());
                std::vector<float> V2(M);
                CK(hipMemcpy(V2.data(), dG + M * (B - 1), sizeof(float) * M, hipMemcpyDeviceToHost));
                double w2 = 0, s2 = 0, wr = 0;
                int wi = -1, wc = -1;
                for (int i = 0; i < NP; i += 37)
                {
                        std::vector<double> v(NP);
                        for (int c = 0; c < NP; ++c)
                        {
                                double s = G[(size_t)i * NP + c];
                                for (int q = 0; q < c; ++q)
                                        s -= v[q] * (double)Lf[(size_t)c * NP + q];
                                v[c] = s / (double)Lf[(size_t)c * NP + c];
                        }
                        for (int c = 0; c < NP; ++c)
                        {
                                const double e = std::fabs(v[c] - (double)V2[(size_t)i * NP + c]);
                                if (e > w2)
                                        w2 = e, wi = i, wc = c;
                                s2 = std::fmax(s2, std::fabs(v[c]));
                                wr = std::fmax(wr, std::fabs((double)V[(size_t)i * NP + c] - (double)V2[(size_t)i * NP + c]));
                        }
                }
                std::printf("large_trsm_bf16: max |V - host| / max |V| = %.2e at (%d, %d); against large_trsm_pipe %.2e; large_split_planes %.3f ms for %d filters\n", w2 / s2, wi, wc,
                            wr / s2, msp, B);
                for (int bb : {15, 60, B})
                {
                        if (bb > B)
                                break;
                        const float m0 = time_ms([&]() { hipLaunchKernelGGL(k16, dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lv, pl, bb, dskip); }, 5);
                        std::printf("  trsm_bf16 %3d filters: %7.3f ms\n", bb, m0);
                }
                {
                        double *dY;
                        CK(hipMalloc(&dY, sizeof(double) * 16));
                        LargeView<float> lw = lv;
                        lw.Y = dY;
                        auto stamps = [&](const char *name, auto kern) {
                                hipLaunchKernelGGL(kern, dim3(8 * ((15 + 7) / 8) * NB), dim3(256), 0, 0, d, lw, pl, 15, dskip);
                                CK(hipDeviceSynchronize());
                                double y[13];
                                CK(hipMemcpy(y, dY, sizeof(y), hipMemcpyDeviceToHost));
                                std::printf("  trsm_bf16 %-34s workgroup 0 of 255, shader cycles per block: first-half region %.0f, mid %.0f, second-half region %.0f, barrier %.0f; closing block %.0f; total %.0f\n",
                                            name, y[0] / 136, y[1] / 136, y[2] / 136, y[3] / 136, (y[4] + y[5] + y[6] + y[7] + y[8] + y[9] + y[10] + y[11] + y[12]) / 17, y[0] + y[1] + y[2] + y[3] + y[4] + y[5] + y[6] + y[7] + y[8] + y[9] + y[10] + y[11] + y[12]);
                                std::printf("      closing block: wait for the slice %.0f, its LDS reads + issue of the next %.0f, C + split %.0f, first half %.0f, second half %.0f, stores %.0f, strip write %.0f, vmcnt wait %.0f, barrier %.0f\n", y[9] / 17, y[10] / 17, y[5] / 17, y[6] / 17, y[7] / 17, y[11] / 17, y[8] / 17, y[12] / 17, y[4] / 17);
                        };
                        stamps("product", k16s);
                        if (!rl)
                        {
                        stamps("second half: no DMA pieces", large_trsm_bf16<LARGE_NB_MAX, 5>);
                        stamps("second half: no VALU", large_trsm_bf16<LARGE_NB_MAX, 2>);
                        stamps("second half: no LDS reads", large_trsm_bf16<LARGE_NB_MAX, 3>);
                        stamps("second half: MFMAs only", large_trsm_bf16<LARGE_NB_MAX, 4>);
                        }
                }
        }
        if (only && std::string(only) == "trsm16")
                return 0;
        // ---- timings (the result does not matter: G is solved again in place).  DIAG bits: 1 = no global fetch of the L blocks (stale
        // LDS), 2 = no LDS stash and no barrier, 16 = no barrier (racy), 8 = in-kernel stamps
        const double mfma_per_wave = 64.0 * 136 + 40.0 * 17;
        const double fl = mfma_per_wave * 2048.0 * 68 * B; // x 2048 flop x 68 waves per filter
        for (int bb : {15, 30, 60, 120, B})
        {
                if (bb > B)
                        break;
                const float m0 = time_ms([&]() { hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 0>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lv, bb, dskip); }, 5);
                std::printf("  %3d filters = %4d workgroups (%.2f per CU): %7.3f ms = %6.1f TFLOP/s executed (%4.1f %% of 157.3)\n", bb, 17 * bb, 17.0 * bb / 256, m0,
                            fl * bb / B / (m0 * 1e-3) / 1e12, fl * bb / B / (m0 * 1e-3) / 1e12 / 157.3 * 100);
        }
        // ---- clock and cycles per MFMA inside the kernel: s_memtime (shader clock) and s_memrealtime (100 MHz) around the sweep of every workgroup
        {
                double *dY;
                const int bb = 15; // one workgroup per CU
                CK(hipMalloc(&dY, sizeof(double) * 2 * NB * (B + 8)));
                LargeView<float> lw = lv;
                lw.Y = dY;
                auto report = [&](const char *name) {
                        CK(hipDeviceSynchronize());
                        std::vector<double> y(2 * NB * 8 * ((bb + 7) / 8));
                        CK(hipMemcpy(y.data(), dY, sizeof(double) * y.size(), hipMemcpyDeviceToHost));
                        double cyc = 0, real = 0;
                        int cnt = 0;
                        for (size_t i = 0; i < y.size() / 2; ++i)
                                if (y[2 * i] > 0)
                                        cyc += y[2 * i], real += y[2 * i + 1], ++cnt;
                        cyc /= cnt, real /= cnt;
                        std::printf("  %-28s per workgroup: %9.0f shader cycles, %7.2f us => %5.0f MHz, %5.1f cycles per MFMA per wave\n", name, cyc, real / 100.0,
                                    cyc / (real / 100.0), cyc / mfma_per_wave);
                };
                CK(hipMemset(dY, 0, sizeof(double) * 2 * NB * (B + 8)));
                hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 8>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lw, bb, dskip);
                report("product");
                hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 9>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lw, bb, dskip);
                report("no fetch");
                hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 9 + 16>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lw, bb, dskip);
                report("no fetch, no barrier");
                hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 11>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lw, bb, dskip);
                report("no fetch/stash/barrier");
                hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 8 + 16>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lw, bb, dskip);
                report("no barrier");
                hipLaunchKernelGGL((large_trsm_pipe<LARGE_NB_MAX, 8 + 32>), dim3(8 * ((bb + 7) / 8) * NB), dim3(256), 0, 0, d, lw, bb, dskip);
                report("every fetch from one block (L1)");
                CK(hipFree(dY));
        }
        // ---- large_chol_resident: S (binary32 copy of the SPD matrix) -> L, Linv against the host factor; then its time
        {
                std::vector<float> Sf(M);
                for (int i = 0; i < NP; ++i)
                        for (int j = 0; j < NP; ++j)
                                Sf[(size_t)i * NP + j] = (float)(i >= j ? S[(size_t)i * NP + j] : S[(size_t)j * NP + i]);
                uint32_t *dstatus;
                CK(hipMalloc(&dstatus, sizeof(uint32_t) * B));
                CK(hipMemset(dstatus, 0, sizeof(uint32_t) * B));
                DevView dc = d;
                dc.status = dstatus;
                auto reset_S = [&]() {
                        for (int b = 0; b < B; ++b)
                                CK(hipMemcpyAsync(dS + M * b, Sf.data(), sizeof(float) * M, hipMemcpyHostToDevice, 0));
                        CK(hipMemsetAsync(dLinv, 0xff, sizeof(float) * NB * LB * LB * B, 0)); // NaNs: nothing may be read before it is written
                };
                reset_S();
                {
                        // the bf16 planes large_chol_resident writes beside L must be, bit for bit, what large_split_planes makes of that L
                        LargeView<float> lp = lv;
                        LPlanes pa = {}, pb = {};
                        const size_t pe = LPlanes::per_filter(NP) * B;
                        CK(hipMalloc(&pa.base, sizeof(unsigned short) * pe));
                        CK(hipMalloc(&pb.base, sizeof(unsigned short) * pe));
                        CK(hipMemset(pa.base, 0, sizeof(unsigned short) * pe));
                        CK(hipMemset(pb.base, 0, sizeof(unsigned short) * pe));
                        lp.Lpl = pa.base;
                        hipLaunchKernelGGL((large_chol_resident<LARGE_NB_MAX>), dim3(B), dim3(256), 0, 0, dc, lp, dskip);
                        hipLaunchKernelGGL(large_split_planes<0>, dim3(NB + 1, B), dim3(256), 0, 0, d, lv, pb, dskip);
                        CK(hipDeviceSynchronize());
                        std::vector<unsigned short> ha(LPlanes::per_filter(NP)), hb(LPlanes::per_filter(NP));
                        CK(hipMemcpy(ha.data(), pa.Lq(B - 1, NP), sizeof(unsigned short) * ha.size(), hipMemcpyDeviceToHost));
                        CK(hipMemcpy(hb.data(), pb.Lq(B - 1, NP), sizeof(unsigned short) * hb.size(), hipMemcpyDeviceToHost));
                        size_t diff = 0, first = 0;
                        for (size_t i = 0; i < ha.size(); ++i)
                                if (ha[i] != hb[i] && !diff++)
                                        first = i;
                        std::printf("large_chol_resident's bf16 planes against large_split_planes of its L: %zu of %zu elements differ%s\n", diff, ha.size(),
                                    diff ? "" : " (bit-identical)");
                        if (diff)
                                std::printf("   first difference at element %zu (plane %zu, row %zu, position %zu)\n", first, first / M, (first % M) / NP, first % NP);
                        const float mc = time_ms([&]() { hipLaunchKernelGGL((large_chol_resident<LARGE_NB_MAX>), dim3(B), dim3(256), 0, 0, dc, lp, dskip); }, 3);
                        std::printf("  chol_resident writing the planes too, %d filters: %7.3f ms (the factor of a factor: timing only)\n", B, mc);
                        CK(hipFree(pa.base));
                        CK(hipFree(pb.base));
                        reset_S();
                }
                {
                        // ---- large_chol_bf16: the same factorisation on the bf16 pipe, writing L, Linv and their planes
                        LPlanes pa = {}, pb = {};
                        const size_t pe = LPlanes::per_filter(NP) * B;
                        CK(hipMalloc(&pa.base, sizeof(unsigned short) * pe));
                        CK(hipMalloc(&pb.base, sizeof(unsigned short) * pe));
                        CK(hipMemset(pa.base, 0, sizeof(unsigned short) * pe));
                        CK(hipMemset(pb.base, 0, sizeof(unsigned short) * pe));
                        CK(hipMemset(dstatus, 0, sizeof(uint32_t) * B));
                        hipLaunchKernelGGL((large_chol_bf16<LARGE_NB_MAX>), dim3(B), dim3(256), 0, 0, dc, lv, pa, dskip);
                        hipLaunchKernelGGL(large_split_planes<0>, dim3(NB + 1, B), dim3(256), 0, 0, d, lv, pb, dskip);
                        CK(hipDeviceSynchronize());
                        std::vector<float> L2(M), Li2((size_t)NB * LB * LB);
                        std::vector<uint32_t> st2(B);
                        CK(hipMemcpy(L2.data(), dS + M * (B - 1), sizeof(float) * M, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(Li2.data(), dLinv + (size_t)NB * LB * LB * (B - 1), sizeof(float) * NB * LB * LB, hipMemcpyDeviceToHost));
                        CK(hipMemcpy(st2.data(), dstatus, sizeof(uint32_t) * B, hipMemcpyDeviceToHost));
                        double eL = 0, sL = 0;
                        for (int i = 0; i < NP; ++i)
                                for (int j = 0; j <= i; ++j)
                                {
                                        eL = std::fmax(eL, std::fabs((double)L2[(size_t)i * NP + j] - L[(size_t)i * NP + j]));
                                        sL = std::fmax(sL, std::fabs(L[(size_t)i * NP + j]));
                                }
                        std::vector<unsigned short> ha(LPlanes::per_filter(NP)), hb(LPlanes::per_filter(NP));
                        CK(hipMemcpy(ha.data(), pa.Lq(B - 1, NP), sizeof(unsigned short) * ha.size(), hipMemcpyDeviceToHost));
                        CK(hipMemcpy(hb.data(), pb.Lq(B - 1, NP), sizeof(unsigned short) * hb.size(), hipMemcpyDeviceToHost));
                        size_t diff = 0;
                        for (size_t i = 0; i < ha.size(); ++i)
                                diff += ha[i] != hb[i];
                        uint32_t anyst = 0;
                        for (auto v : st2)
                                anyst |= v;
                        std::printf("large_chol_bf16: max |L - host| / max |L| = %.2e, status bits %u; its planes against large_split_planes of its L: %zu of %zu differ\n", eL / sL,
                                    anyst, diff, ha.size());
                        {
                                // every filter has the same input: every filter's L must be bit-identical to filter 0's -- a race detector
                                std::vector<float> L0(M), Lb(M);
                                CK(hipMemcpy(L0.data(), dS, sizeof(float) * M, hipMemcpyDeviceToHost));
                                int bad = 0, firstb = -1;
                                size_t firsti = 0, nbad0 = 0;
                                for (int b = 1; b < B; ++b)
                                {
                                        CK(hipMemcpy(Lb.data(), dS + M * b, sizeof(float) * M, hipMemcpyDeviceToHost));
                                        size_t nb_ = 0, fi = 0;
                                        for (int i = 0; i < NP; ++i)
                                                for (int j = 0; j <= i; ++j)
                                                        if (std::memcmp(&Lb[(size_t)i * NP + j], &L0[(size_t)i * NP + j], 4) && !nb_++)
                                                                fi = (size_t)i * NP + j;
                                        if (nb_)
                                        {
                                                if (!bad++)
                                                        firstb = b, firsti = fi, nbad0 = nb_;
                                        }
                                }
                                std::printf("large_chol_bf16: %d of %d filters differ from filter 0 in L (identical inputs)", bad, B - 1);
                                if (bad)
                                        std::printf("; first: filter %d, %zu entries, first at row %zu column %zu (block %zu, %zu)", firstb, nbad0, firsti / NP, firsti % NP, firsti / NP / 64,
                                                    firsti % NP / 64);
                                std::printf("\n");
                        }
                        for (int bb : {15, 64, B})
                        {
                                if (bb > B)
                                        break;
                                reset_S();
                                CK(hipDeviceSynchronize());
                                hipEvent_t e0, e1;
                                CK(hipEventCreate(&e0));
                                CK(hipEventCreate(&e1));
                                CK(hipEventRecord(e0));
                                hipLaunchKernelGGL((large_chol_bf16<LARGE_NB_MAX>), dim3(bb), dim3(256), 0, 0, dc, lv, pa, dskip);
                                CK(hipEventRecord(e1));
                                CK(hipEventSynchronize(e1));
                                float ms = 0;
                                CK(hipEventElapsedTime(&ms, e0, e1));
                                std::printf("  chol_bf16 %3d filters: %7.3f ms\n", bb, ms);
                        }
                        // and the bf16 TRSM on ITS planes
                        reset_G();
                        const float mt = time_ms([&]() { hipLaunchKernelGGL((large_trsm_bf16<LARGE_NB_MAX>), dim3(8 * ((B + 7) / 8) * NB), dim3(256), 0, 0, d, lv, pa, B, dskip); }, 3);
                        std::printf("  trsm_bf16 on the planes large_chol_bf16 wrote, %d filters: %7.3f ms\n", B, mt);
                        CK(hipFree(pa.base));
                        CK(hipFree(pb.base));
                        CK(hipMemset(dstatus, 0, sizeof(uint32_t) * B));
                        reset_S();
                }
                hipLaunchKernelGGL((large_chol_resident<LARGE_NB_MAX>), dim3(B), dim3(256), 0, 0, dc, lv, dskip);
                CK(hipDeviceSynchronize());
                std::vector<float> Lg(M), Lig((size_t)NB * LB * LB);
                std::vector<uint32_t> st(B);
                CK(hipMemcpy(Lg.data(), dS + M * (B - 1), sizeof(float) * M, hipMemcpyDeviceToHost));
                CK(hipMemcpy(Lig.data(), dLinv + (size_t)NB * LB * LB * (B - 1), sizeof(float) * Lig.size(), hipMemcpyDeviceToHost));
                CK(hipMemcpy(st.data(), dstatus, sizeof(uint32_t) * B, hipMemcpyDeviceToHost));
                double eL = 0, sL = 0, eI = 0, sI = 0;
                for (int i = 0; i < NP; ++i)
                        for (int j = 0; j <= i; ++j)
                        {
                                eL = std::fmax(eL, std::fabs((double)Lg[(size_t)i * NP + j] - L[(size_t)i * NP + j]));
                                sL = std::fmax(sL, std::fabs(L[(size_t)i * NP + j]));
                        }
                for (size_t i = 0; i < Lig.size(); ++i)
                {
                        eI = std::fmax(eI, std::fabs((double)Lig[i] - (double)Linv[i]));
                        sI = std::fmax(sI, std::fabs((double)Linv[i]));
                }
                uint32_t anyst = 0;
                for (uint32_t x : st)
                        anyst |= x;
                std::printf("large_chol_resident: max |L - host| / max |L| = %.2e, max |Linv - host| / max |Linv| = %.2e, status bits %u (filter %d)\n", eL / sL,
                            eI / sI, anyst, B - 1);
                for (int bb : {15, 64, 128, B})
                {
                        if (bb > B)
                                break;
                        reset_S();
                        CK(hipDeviceSynchronize());
                        hipEvent_t e0, e1;
                        CK(hipEventCreate(&e0));
                        CK(hipEventCreate(&e1));
                        CK(hipEventRecord(e0));
                        hipLaunchKernelGGL((large_chol_resident<LARGE_NB_MAX>), dim3(bb), dim3(256), 0, 0, dc, lv, dskip);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms = 0;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        std::printf("  chol_resident %3d filters: %7.3f ms\n", bb, ms);
                }
                {
                        double *dY;
                        CK(hipMalloc(&dY, sizeof(double) * 4));
                        LargeView<float> lw = lv;
                        lw.Y = dY;
                        reset_S();
                        hipLaunchKernelGGL((large_chol_resident<LARGE_NB_MAX, 1>), dim3(B), dim3(256), 0, 0, dc, lw, dskip);
                        CK(hipDeviceSynchronize());
                        double y[4];
                        CK(hipMemcpy(y, dY, sizeof(y), hipMemcpyDeviceToHost));
                        std::printf("  chol_resident, workgroup 0, shader cycles by phase over the 17 block rows: sweeps %.0f, C -> tiles %.0f, 64x64 factor + inverse %.0f, "
                                    "stores + drain %.0f (per block row: %.0f / %.0f / %.0f / %.0f)\n",
                                    y[0], y[1], y[2], y[3], y[0] / 17, y[1] / 17, y[2] / 17, y[3] / 17);
                        CK(hipFree(dY));
                }
                // restore the factor for the kernels timed below
                for (int b = 0; b < B; ++b)
                {
                        CK(hipMemcpy(dS + M * b, Lf.data(), sizeof(float) * M, hipMemcpyHostToDevice));
                        CK(hipMemcpy(dLinv + (size_t)NB * LB * LB * b, Linv.data(), sizeof(float) * NB * LB * LB, hipMemcpyHostToDevice));
                }
                CK(hipFree(dstatus));
        }
        // ---- syrk
        {
                const int ntile = (NP + 127) / 128;
                const dim3 grid(8 * (ntile * (ntile + 1) / 2) * ((B + 7) / 8));
                const double sf = (ntile * (ntile + 1) / 2 - ntile * 0.25) * 2.0 * 128 * 128 * 1056 * B;
                // (round 2 also timed large_syrk_f32p64 here, the fp32-MFMA form: 3.06 ms per 256 filters, K loop 2.7 ms; removed in round 3)
                const float mb = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<0>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  syrk_bf16x3      %8.3f ms for %d filters = %6.1f T fp32-equivalent FLOP/s executed\n", mb, B, sf / (mb * 1e-3) / 1e12);
                const float mbk = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<1>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  syrk_bf16x3 without the read-modify-write of P      %8.3f ms\n", mbk);
                const float mb2 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<2>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                const float mbk2 = time_ms([&]() { hipLaunchKernelGGL((large_syrk_bf16x3<3>), grid, dim3(256), 0, 0, d, lv, LPlanes{nullptr}, B, dskip); }, 5);
                std::printf("  syrk_bf16x3 with round 2's running accumulator (no per-slab temporaries)  %8.3f ms, without the read-modify-write %8.3f ms\n", mb2, mbk2);
        }
        return 0;
}
