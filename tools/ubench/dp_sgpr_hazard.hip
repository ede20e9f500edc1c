// Hazard probe for gfx950 (round 2, UKF chol(P) investigation): instruction sequences lifted from the one compiled
// instance of factor_diag_tile_fast that gives 1e-6-class errors on the GPU although its instruction stream is
// logically right (tools/isa_emul.py).  Each test runs the sequence as written by hipcc, under the same EXEC mask
// (lanes 0..15), and compares with the exact result; then again with wait states inserted.
//   T1  v_mul_f64 x2 reading an SGPR pair, immediately followed by v_readlane_b32 overwriting that pair (WAR)
//   T2  v_readlane_b32 -> SGPR pair, two instructions, v_cndmask_b32 using the pair as its mask (RAW, 2 wait states)
//   T3  v_readfirstlane_b32 pair -> (s_and_saveexec) -> v_mul_f64 reading the pair (RAW across an EXEC change)
// hipcc --offload-arch=gfx950 -O2 dp_sgpr_hazard.hip -o dp_sgpr_hazard && ./dp_sgpr_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#define GAP0 ""
#define GAP1 "s_nop 0\n"
#define GAP4 "s_nop 3\n"
#define GAP16 "s_nop 7\ns_nop 7\n"

// T1: r1 = x1 * m, r2 = x2 * m with m in s[20:21]; then s20/s21 are overwritten from lanes 11/12 of `junk`
#define T1_BODY(GAP)                                                                                                   \
        asm volatile("v_readfirstlane_b32 s20, %[mlo]\n"                                                               \
                     "v_readfirstlane_b32 s21, %[mhi]\n"                                                               \
                     "s_nop 7\n"                                                                                       \
                     "s_mov_b64 s[22:23], exec\n"                                                                      \
                     "s_mov_b64 exec, %[mask]\n"                                                                       \
                     "s_nop 7\n"                                                                                       \
                     "v_mul_f64 %[r1], %[x1], s[20:21]\n"                                                              \
                     "v_mul_f64 %[r2], %[x2], s[20:21]\n" GAP "v_readlane_b32 s20, %[junk], 11\n"                      \
                     "v_lshlrev_b32 %[t], 7, %[t]\n"                                                                   \
                     "v_readlane_b32 s21, %[junk], 12\n"                                                               \
                     "s_nop 7\n"                                                                                       \
                     "s_mov_b64 exec, s[22:23]\n"                                                                      \
                     : [r1] "=&v"(r1), [r2] "=&v"(r2), [t] "+v"(t)                                                      \
                     : [x1] "v"(x1), [x2] "v"(x2), [mlo] "v"(mlo), [mhi] "v"(mhi), [junk] "v"(junk), [mask] "s"(mask)  \
                     : "s20", "s21", "s22", "s23", "memory")

template <int G> __global__ void t1(const double *x, double m, unsigned long long mask, double *out)
{
        const int lane = threadIdx.x;
        double x1 = x[lane], x2 = x[64 + lane], r1 = 0, r2 = 0;
        unsigned long long mb;
        memcpy(&mb, &m, 8);
        int mlo = (int)(mb & 0xffffffffu), mhi = (int)(mb >> 32), junk = 0x00010001 * (lane + 1), t = lane;
        for (int rep = 0; rep < 64; ++rep)
        {
                if (G == 0)
                        T1_BODY(GAP0);
                else if (G == 1)
                        T1_BODY(GAP1);
                else if (G == 4)
                        T1_BODY(GAP4);
                else
                        T1_BODY(GAP16);
                x1 = r1 * (1.0 / m) + x1 * 0.0 + x[lane]; // keep the loop alive without changing the operands much
                x1 = x[lane];
        }
        out[lane] = r1;
        out[64 + lane] = r2;
        out[128 + lane] = (double)t;
}

// T2: mask pair restored by v_readlane_b32 from a spill VGPR, used by v_cndmask_b32 two instructions later
#define T2_BODY(GAP)                                                                                                   \
        asm volatile("s_mov_b64 s[22:23], exec\n"                                                                      \
                     "s_mov_b64 exec, %[mask]\n"                                                                       \
                     "s_nop 7\n"                                                                                       \
                     "v_readlane_b32 s20, %[spill], 11\n"                                                              \
                     "v_lshlrev_b32 %[t], 7, %[t]\n"                                                                   \
                     "v_readlane_b32 s21, %[spill], 12\n"                                                              \
                     "v_sub_u32 %[t], %[t], %[u]\n"                                                                    \
                     "v_add_u32 %[u], 1, %[u]\n" GAP "v_cndmask_b32_e64 %[r], %[a], 0, s[20:21]\n"                     \
                     "s_nop 7\n"                                                                                       \
                     "s_mov_b64 exec, s[22:23]\n"                                                                      \
                     : [r] "=&v"(r), [t] "+v"(t), [u] "+v"(u)                                                           \
                     : [a] "v"(a), [spill] "v"(spill), [mask] "s"(mask)                                                \
                     : "s20", "s21", "s22", "s23", "memory")

template <int G> __global__ void t2(unsigned long long sel, unsigned long long mask, int *out)
{
        const int lane = threadIdx.x;
        int a = 1000 + lane, r = -1, t = lane, u = 3;
        // the spill VGPR: lane 11 holds the low word of the selection mask, lane 12 the high word
        int spill = lane == 11 ? (int)(sel & 0xffffffffu) : lane == 12 ? (int)(sel >> 32) : 0x5a5a5a5a;
        for (int rep = 0; rep < 64; ++rep)
        {
                // poison s20/s21 first so that a stale read is visible
                asm volatile("s_mov_b32 s20, -1\ns_mov_b32 s21, -1\ns_nop 7" ::: "s20", "s21");
                if (G == 0)
                        T2_BODY(GAP0);
                else if (G == 1)
                        T2_BODY(GAP1);
                else if (G == 4)
                        T2_BODY(GAP4);
                else
                        T2_BODY(GAP16);
        }
        out[lane] = r;
        out[64 + lane] = t + u;
}

// T3: the pivot multiplier: v_fmac_f64 -> v_readfirstlane_b32 x2 -> v_cmp/s_and_saveexec -> v_mul_f64 x2 with the pair
#define T3_BODY(GAP)                                                                                                   \
        asm volatile("v_fma_f64 %[p], %[y], %[e], %[y]\n"                                                              \
                     "s_nop 0\n"                                                                                       \
                     "v_readfirstlane_b32 s21, %[phi]\n"                                                               \
                     "v_readfirstlane_b32 s20, %[plo]\n"                                                               \
                     "v_cmp_gt_u32_e64 s[24:25], 16, %[lane]\n"                                                        \
                     "s_and_saveexec_b64 s[22:23], s[24:25]\n" GAP "v_mul_f64 %[r1], %[x1], s[20:21]\n"                \
                     "v_mul_f64 %[r2], %[x2], s[20:21]\n"                                                              \
                     "s_nop 7\n"                                                                                       \
                     "s_mov_b64 exec, s[22:23]\n"                                                                      \
                     : [r1] "=&v"(r1), [r2] "=&v"(r2), [p] "=&v"(p)                                                     \
                     : [x1] "v"(x1), [x2] "v"(x2), [y] "v"(y), [e] "v"(e), [lane] "v"(lane),                           \
                       [plo] "v"(((int *)&p)[0]), [phi] "v"(((int *)&p)[1])                                            \
                     : "s20", "s21", "s22", "s23", "s24", "s25", "memory")

template <int G> __global__ void t3(const double *x, double yy, double ee, double *out)
{
        const int lane = threadIdx.x;
        double x1 = x[lane], x2 = x[64 + lane], r1 = 0, r2 = 0, y = yy, e = ee;
        // p = y*e + y computed by plain code first so that plo/phi operands exist; the asm recomputes it
        double p = fma(y, e, y);
        for (int rep = 0; rep < 64; ++rep)
        {
                asm volatile("s_mov_b32 s20, -1\ns_mov_b32 s21, -1\ns_nop 7" ::: "s20", "s21");
                if (G == 0)
                        T3_BODY(GAP0);
                else if (G == 1)
                        T3_BODY(GAP1);
                else if (G == 4)
                        T3_BODY(GAP4);
                else
                        T3_BODY(GAP16);
        }
        out[lane] = r1;
        out[64 + lane] = r2;
}

template <typename F> int check_f64(const char *name, F launch, const std::vector<double> &x, double m, int nlanes)
{
        double *dx, *dout;
        hipMalloc(&dx, 128 * 8);
        hipMalloc(&dout, 192 * 8);
        hipMemcpy(dx, x.data(), 128 * 8, hipMemcpyHostToDevice);
        hipMemset(dout, 0, 192 * 8);
        launch(dx, dout);
        hipDeviceSynchronize();
        std::vector<double> out(192);
        hipMemcpy(out.data(), dout, 192 * 8, hipMemcpyDeviceToHost);
        int bad = 0;
        double worst = 0;
        for (int q = 0; q < 2; ++q)
                for (int l = 0; l < nlanes; ++l)
                {
                        const double want = x[64 * q + l] * m, got = out[64 * q + l];
                        if (want != got)
                        {
                                ++bad;
                                const double rel = fabs(got - want) / fabs(want);
                                worst = rel > worst ? rel : worst;
                        }
                }
        printf("%-58s %s (%d of %d values differ, worst rel %.2e)\n", name, bad ? "WRONG" : "exact", bad, 2 * nlanes, worst);
        hipFree(dx);
        hipFree(dout);
        return bad;
}

int main()
{
        std::vector<double> x(128);
        for (int i = 0; i < 128; ++i)
                x[i] = 0.37 + 0.013 * i + 1e-9 * i * i;
        const double m = 1.2345678901234567;
        const unsigned long long lanes16 = 0xffffull, all = ~0ull;
        int bad = 0;
#define RUN_T1(G, MASK, NL, LABEL)                                                                                     \
        bad += check_f64(LABEL, [&](double *dx, double *dout) { hipLaunchKernelGGL(t1<G>, dim3(256), dim3(64), 0, 0, dx, m, MASK, dout); }, x, m, NL)
        RUN_T1(0, lanes16, 16, "T1 WAR  v_mul_f64 s[20:21]; v_readlane s20   exec=0xffff gap 0");
        RUN_T1(1, lanes16, 16, "T1 WAR                                        exec=0xffff gap 1");
        RUN_T1(4, lanes16, 16, "T1 WAR                                        exec=0xffff gap 4");
        RUN_T1(16, lanes16, 16, "T1 WAR                                        exec=0xffff gap 16");
        RUN_T1(0, all, 64, "T1 WAR                                        exec=all    gap 0");
        RUN_T1(16, all, 64, "T1 WAR                                        exec=all    gap 16");
        // T2
        {
                const unsigned long long sel = 0x00010001000100010ull >> 4; // lanes 0, 16, 32, 48: the (row == 0) mask of the real code
                for (int variant = 0; variant < 8; ++variant)
                {
                        const int G = (variant & 3) == 0 ? 0 : (variant & 3) == 1 ? 1 : (variant & 3) == 2 ? 4 : 16;
                        const unsigned long long mask = variant < 4 ? lanes16 : all;
                        int *dout;
                        hipMalloc(&dout, 128 * 4);
                        hipMemset(dout, 0, 128 * 4);
                        if (G == 0)
                                hipLaunchKernelGGL(t2<0>, dim3(256), dim3(64), 0, 0, sel, mask, dout);
                        else if (G == 1)
                                hipLaunchKernelGGL(t2<1>, dim3(256), dim3(64), 0, 0, sel, mask, dout);
                        else if (G == 4)
                                hipLaunchKernelGGL(t2<4>, dim3(256), dim3(64), 0, 0, sel, mask, dout);
                        else
                                hipLaunchKernelGGL(t2<16>, dim3(256), dim3(64), 0, 0, sel, mask, dout);
                        hipDeviceSynchronize();
                        int out[128];
                        hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
                        int nb = 0;
                        const int nl = variant < 4 ? 16 : 64;
                        for (int l = 0; l < nl; ++l)
                        {
                                const int want = ((sel >> l) & 1) ? 0 : 1000 + l;
                                nb += out[l] != want;
                        }
                        printf("T2 RAW  v_readlane s[20:21]; 2 instr; v_cndmask   exec=%s gap %-2d        %s (%d of %d lanes differ)\n",
                               variant < 4 ? "0xffff" : "all   ", G, nb ? "WRONG" : "exact", nb, nl);
                        bad += nb;
                        hipFree(dout);
                }
        }
        // T3
        {
                const double y = 0.9, e = 3.0e-9, p = fma(y, e, y);
#define RUN_T3(G, LABEL)                                                                                               \
        bad += check_f64(LABEL, [&](double *dx, double *dout) { hipLaunchKernelGGL(t3<G>, dim3(256), dim3(64), 0, 0, dx, y, e, dout); }, x, p, 16)
                RUN_T3(0, "T3 RAW  readfirstlane pair; saveexec; v_mul_f64 x2      gap 0");
                RUN_T3(1, "T3 RAW                                                  gap 1");
                RUN_T3(4, "T3 RAW                                                  gap 4");
                RUN_T3(16, "T3 RAW                                                  gap 16");
        }
        printf("dp_sgpr_hazard: %s\n", bad ? "a sequence gave a wrong result" : "all sequences exact");
        return 0;
}
