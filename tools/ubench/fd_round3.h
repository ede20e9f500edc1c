__device__ __forceinline__ bool factor_diag_old(double *T, double *Ti, int lane)
{
        double a[16], s[16];
        const int row = lane & 15;
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
                a[c] = T[row * TLD + c];
                s[c] = (row == c) ? 1.0 : 0.0;
        }
        // Hand-scheduled: left to itself hipcc sinks the rank-1 updates into lazy dot-product chains (one dependent
        // 32-cycle v_fma_f64 per earlier column in front of every pivot) and spills the broadcast multipliers.  Here
        // column j's updates are issued eagerly, the critical one (row/column j+1) first, and the remaining ones fill
        // the latency gaps of the NEXT pivot's rsqrt chain; sched_barrier pins that order.
#define ASLAM_UPD(c)                                                                                                   \
        if ((c) < 16)                                                                                                  \
        {                                                                                                              \
                const double lc_ = readlane_f64(lij, ((c) < 16) ? (c) : 15);                                           \
                a[((c) < 16) ? (c) : 15] = fma(-lij, lc_, a[((c) < 16) ? (c) : 15]);                                   \
                s[((c) < 16) ? (c) : 15] = fma(-lc_, xj, s[((c) < 16) ? (c) : 15]);                                    \
                asm volatile("" : "+v"(a[((c) < 16) ? (c) : 15]), "+v"(s[((c) < 16) ? (c) : 15])); /* no sinking */   \
        }
        const double d0 = readlane_f64(a[0], 0);
        bool ok = d0 > 0.0;
        double inv = readfirstlane_f64(rsqrt_newton(d0));
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
                const double lij = a[j] * inv; // L(i,j) for i >= j
                const double xj = s[j] * inv;  // (L^-1)(j, lane)
                a[j] = lij;
                s[j] = xj;
                double d = 1.0, y = 1.0, t = 0.0, e = 0.0, ye = 0.0, pp = 0.0, invn = 1.0;
                __builtin_amdgcn_sched_barrier(0);
                ASLAM_UPD(j + 1);
                if (j + 1 < 16)
                {
                        d = readlane_f64(a[(j + 1 < 16) ? j + 1 : 15], (j + 1 < 16) ? j + 1 : 15);
                        ok = ok && (d > 0.0);
                        y = __builtin_amdgcn_rsq(d);
                }
                __builtin_amdgcn_sched_barrier(0);
                t = d * y;
                ASLAM_UPD(j + 2);
                ASLAM_UPD(j + 3);
                ASLAM_UPD(j + 4);
                __builtin_amdgcn_sched_barrier(0);
                e = fma(-t, y, 1.0);
                ASLAM_UPD(j + 5);
                ASLAM_UPD(j + 6);
                ASLAM_UPD(j + 7);
                __builtin_amdgcn_sched_barrier(0);
                ye = y * e;
                pp = fma(e, 0.375, 0.5);
                ASLAM_UPD(j + 8);
                ASLAM_UPD(j + 9);
                ASLAM_UPD(j + 10);
                __builtin_amdgcn_sched_barrier(0);
                invn = fma(ye, pp, y);
                ASLAM_UPD(j + 11);
                ASLAM_UPD(j + 12);
                ASLAM_UPD(j + 13);
                ASLAM_UPD(j + 14);
                ASLAM_UPD(j + 15);
                __builtin_amdgcn_sched_barrier(0);
                inv = readfirstlane_f64(invn);
        }
#undef ASLAM_UPD
        if (lane < 16)
        {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                {
                        T[row * TLD + c] = (c <= row) ? a[c] : 0.0;
                        Ti[c * TLD + row] = s[c]; // Linv(c, row): zero above the diagonal by construction
                }
        }
        return ok;
}

