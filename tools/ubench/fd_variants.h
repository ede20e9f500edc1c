// variants of the diagonal tile factorisation: MERGE = a / s in two lane groups of one array; CHAIN: 0 = old chain (pivot read back from the updated tile),
// 1 = short chain (d from q, dd, rinv), 2 = short chain with q and dd pinned where they are formed
template <bool MERGE, int CHAIN, bool GROUP = false, int SKIP = 0> __device__ __forceinline__ bool fd_variant(double *T, double *Ti, int lane)
{
        double x[16], s[16];
        const int row = lane & 15;
        const bool inv_part = MERGE && (lane & 16) != 0;
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
                const double av = T[row * TLD + c];
                x[c] = inv_part ? ((row == c) ? 1.0 : 0.0) : av;
                s[c] = (row == c) ? 1.0 : 0.0;
        }
#define UPD(c)                                                                                                         \
        if ((c) < 16 && !((SKIP & 1) && (c) > j + 1))                                                                                                  \
        {                                                                                                              \
                const double lc_ = readlane_f64(v, ((c) < 16) ? (c) : 15);                                             \
                x[((c) < 16) ? (c) : 15] = fma(-v, lc_, x[((c) < 16) ? (c) : 15]);                                     \
                asm volatile("" : "+v"(x[((c) < 16) ? (c) : 15]));                                                     \
                if (!MERGE)                                                                                            \
                {                                                                                                      \
                        s[((c) < 16) ? (c) : 15] = fma(-lc_, xj, s[((c) < 16) ? (c) : 15]);                            \
                        asm volatile("" : "+v"(s[((c) < 16) ? (c) : 15]));                                             \
                }                                                                                                      \
        }
#define CL(c) (((c) < 16) ? (c) : 15)
#define UPDG3(c0, c1, c2)                                                                                              \
        if (GROUP)                                                                                                     \
        {                                                                                                              \
                const double l0_ = readlane_f64(v, CL(c0)), l1_ = readlane_f64(v, CL(c1)), l2_ = readlane_f64(v, CL(c2)); \
                asm volatile("" ::"s"(l0_), "s"(l1_), "s"(l2_));                                                       \
                if ((c0) < 16) { x[CL(c0)] = fma(-v, l0_, x[CL(c0)]); if (!MERGE) s[CL(c0)] = fma(-l0_, xj, s[CL(c0)]); } \
                if ((c1) < 16) { x[CL(c1)] = fma(-v, l1_, x[CL(c1)]); if (!MERGE) s[CL(c1)] = fma(-l1_, xj, s[CL(c1)]); } \
                if ((c2) < 16) { x[CL(c2)] = fma(-v, l2_, x[CL(c2)]); if (!MERGE) s[CL(c2)] = fma(-l2_, xj, s[CL(c2)]); } \
                asm volatile("" : "+v"(x[CL(c0)]), "+v"(x[CL(c1)]), "+v"(x[CL(c2)]));                                  \
                if (!MERGE) asm volatile("" : "+v"(s[CL(c0)]), "+v"(s[CL(c1)]), "+v"(s[CL(c2)]));                      \
        }                                                                                                              \
        else                                                                                                           \
        {                                                                                                              \
                UPD(c0);                                                                                               \
                UPD(c1);                                                                                               \
                UPD(c2);                                                                                               \
        }
        double d = readlane_f64(x[0], 0);
        bool ok = d > 0.0;
        double inv, rinv;
        {
                const double y = __builtin_amdgcn_rsq(d), t = d * y, y2 = y * y;
                const double e = fma(-t, y, 1.0);
                inv = fma(y * e, fma(e, 0.375, 0.5), y);
                rinv = fma(y2 * e, 1.0 + e, y2);
        }
        double q = 0.0, dd = 1.0;
        if (CHAIN)
        {
                const double u = readlane_f64(x[0], 1);
                q = u * u;
                dd = readlane_f64(x[1], 1);
        }
#pragma unroll
        for (int j = 0; j < ((SKIP & 4) ? 0 : 16); ++j)
        {
                double y = 1.0, t = 0.0, y2 = 1.0, e = 0.0, invn = 1.0, rinvn = 1.0;
                if (CHAIN && j + 1 < 16)
                {
                        d = fma(-q, rinv, dd);
                        ok = ok && (d > 0.0);
                        y = __builtin_amdgcn_rsq(d);
                }
                const double v = x[j] * inv;
                const double xj = s[j] * inv;
                x[j] = v;
                s[j] = xj;
                __builtin_amdgcn_sched_barrier(0);
                UPD(j + 1);
                if (!CHAIN && j + 1 < 16)
                {
                        d = readlane_f64(x[(j + 1 < 16) ? j + 1 : 15], (j + 1 < 16) ? j + 1 : 15);
                        ok = ok && (d > 0.0);
                        y = (SKIP & 2) ? d : __builtin_amdgcn_rsq(d);
                }
                if (CHAIN)
                        UPD(j + 2);
                __builtin_amdgcn_sched_barrier(0);
                t = d * y;
                if (CHAIN)
                        y2 = y * y;
                if (CHAIN && j + 2 < 16)
                {
                        const double u = readlane_f64(x[(j + 1 < 16) ? j + 1 : 15], (j + 2 < 16) ? j + 2 : 15);
                        q = u * u;
                        dd = readlane_f64(x[(j + 2 < 16) ? j + 2 : 15], (j + 2 < 16) ? j + 2 : 15);
                        if (CHAIN == 2)
                                asm volatile("" : "+v"(q), "+v"(dd));
                }
                if (!CHAIN && GROUP)
                {
                        UPDG3(j + 2, j + 3, j + 4);
                        __builtin_amdgcn_sched_barrier(0);
                        e = fma(-t, y, 1.0);
                        UPDG3(j + 5, j + 6, j + 7);
                        __builtin_amdgcn_sched_barrier(0);
                        const double ye = y * e, pp = fma(e, 0.375, 0.5);
                        UPDG3(j + 8, j + 9, j + 10);
                        __builtin_amdgcn_sched_barrier(0);
                        invn = fma(ye, pp, y);
                        UPDG3(j + 11, j + 12, j + 13);
                        UPD(j + 14);
                        UPD(j + 15);
                        __builtin_amdgcn_sched_barrier(0);
                }
                else
                {
                if (!CHAIN)
                        UPD(j + 2);
                UPD(j + 3);
                UPD(j + 4);
                if (CHAIN)
                        UPD(j + 5);
                __builtin_amdgcn_sched_barrier(0);
                e = fma(-t, y, 1.0);
                if (!CHAIN)
                        UPD(j + 5);
                UPD(j + 6);
                UPD(j + 7);
                if (CHAIN)
                {
                        UPD(j + 8);
                        UPD(j + 9);
                }
                __builtin_amdgcn_sched_barrier(0);
                const double ye = y * e, pp = fma(e, 0.375, 0.5);
                double y2e = 0.0, ep = 0.0;
                if (CHAIN)
                        y2e = y2 * e, ep = 1.0 + e;
                if (!CHAIN)
                {
                        UPD(j + 8);
                        UPD(j + 9);
                }
                UPD(j + 10);
                if (CHAIN)
                {
                        UPD(j + 11);
                        UPD(j + 12);
                }
                __builtin_amdgcn_sched_barrier(0);
                invn = fma(ye, pp, y);
                if (CHAIN)
                        rinvn = fma(y2e, ep, y2);
                if (!CHAIN)
                {
                        UPD(j + 11);
                        UPD(j + 12);
                }
                UPD(j + 13);
                UPD(j + 14);
                UPD(j + 15);
                __builtin_amdgcn_sched_barrier(0);
                }
                inv = (SKIP & 8) ? inv : (CHAIN ? invn : readfirstlane_f64(invn));
                rinv = rinvn;
        }
#undef UPD
        if (lane < (MERGE ? 32 : 16))
        {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                {
                        if (!inv_part)
                                T[row * TLD + c] = (c <= row) ? x[c] : 0.0;
                        if (!MERGE)
                                Ti[c * TLD + row] = s[c];
                        else if (inv_part)
                                Ti[c * TLD + row] = x[c];
                }
        }
        return ok;
}

// DPP variant: the multiplier L(c, j) reaches every lane of its row of 16 through row_newbcast on the fmac itself (no v_readlane, no SGPR)
#define DPP_FMAC(acc, bsrc, other, c)                                                                                  \
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #c " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(other))
template <int J> struct DppCol
{
        // updates of columns J+1 .. 15 with column J (lij, xj), in the order FIRST, then the rest
        template <int C> static __device__ __forceinline__ void upd(double (&a)[16], double (&s)[16], const double &lij, const double &xj)
        {
                if constexpr (C < 16)
                {
#define CASE_(cc)                                                                                                      \
        if constexpr (C == cc)                                                                                         \
        {                                                                                                              \
                DPP_FMAC(a[cc], lij, lij, cc);                                                                         \
                DPP_FMAC(s[cc], lij, xj, cc);                                                                          \
        }
                        CASE_(1) CASE_(2) CASE_(3) CASE_(4) CASE_(5) CASE_(6) CASE_(7) CASE_(8) CASE_(9) CASE_(10) CASE_(11) CASE_(12) CASE_(13) CASE_(14) CASE_(15)
#undef CASE_
                }
        }
};
template <int J> __device__ __forceinline__ void fd_dpp_col(double (&a)[16], double (&s)[16], double &inv, bool &ok)
{
        const double lij = a[J] * inv, xj = s[J] * inv;
        a[J] = lij;
        s[J] = xj;
        double lq = lij, xq = xj;
        asm volatile("s_nop 1" : "+v"(lq), "+v"(xq)); // VALU write -> DPP read of the same VGPR: 2 wait states
        double d = 1.0, y = 1.0;
        DppCol<J>::template upd<J + 1>(a, s, lq, xq);
        if constexpr (J + 1 < 16)
        {
                d = readlane_f64(a[J + 1], J + 1);
                ok = ok && (d > 0.0);
                y = __builtin_amdgcn_rsq(d);
        }
        __builtin_amdgcn_sched_barrier(0);
        const double t = d * y;
        DppCol<J>::template upd<J + 2>(a, s, lq, xq);
        DppCol<J>::template upd<J + 3>(a, s, lq, xq);
        DppCol<J>::template upd<J + 4>(a, s, lq, xq);
        __builtin_amdgcn_sched_barrier(0);
        const double e = fma(-t, y, 1.0);
        DppCol<J>::template upd<J + 5>(a, s, lq, xq);
        DppCol<J>::template upd<J + 6>(a, s, lq, xq);
        DppCol<J>::template upd<J + 7>(a, s, lq, xq);
        __builtin_amdgcn_sched_barrier(0);
        const double ye = y * e, pp = fma(e, 0.375, 0.5);
        DppCol<J>::template upd<J + 8>(a, s, lq, xq);
        DppCol<J>::template upd<J + 9>(a, s, lq, xq);
        DppCol<J>::template upd<J + 10>(a, s, lq, xq);
        __builtin_amdgcn_sched_barrier(0);
        const double invn = fma(ye, pp, y);
        DppCol<J>::template upd<J + 11>(a, s, lq, xq);
        DppCol<J>::template upd<J + 12>(a, s, lq, xq);
        DppCol<J>::template upd<J + 13>(a, s, lq, xq);
        DppCol<J>::template upd<J + 14>(a, s, lq, xq);
        DppCol<J>::template upd<J + 15>(a, s, lq, xq);
        __builtin_amdgcn_sched_barrier(0);
        inv = readfirstlane_f64(invn);
        if constexpr (J + 1 < 16)
                fd_dpp_col<J + 1>(a, s, inv, ok);
}
__device__ __forceinline__ bool fd_dpp(double *T, double *Ti, int lane)
{
        double a[16], s[16];
        const int row = lane & 15;
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
                a[c] = T[row * TLD + c];
                s[c] = (row == c) ? 1.0 : 0.0;
        }
        const double d0 = readlane_f64(a[0], 0);
        bool ok = d0 > 0.0;
        double inv = readfirstlane_f64(rsqrt_newton(d0));
        fd_dpp_col<0>(a, s, inv, ok);
        if (lane < 16)
        {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                {
                        T[row * TLD + c] = (c <= row) ? a[c] : 0.0;
                        Ti[c * TLD + row] = s[c];
                }
        }
        return ok;
}
