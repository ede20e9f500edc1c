// trsm16_right_looking.h -- EXPERIMENT (round 3, not part of the library): the bf16-pipe sweep of ekf_large_trsm16.h turned right-looking.
// Measured (tools/ubench/trsm_bench.hip, MI355X): a history block drops from ~ 1750 to ~ 1320 shader cycles, but with every CU busy the kernel
// only goes from 2.31 to 2.13 ms per 256 filters -- both sweeps stream 54 MB of planes per filter from L2 (17 strips x the lower triangle), 13.8 GB
// per launch = 6.5 TB/s at 2.1 ms: the L2 -> LDS roofline of the 64-row strip -- and its error against the host solve is 1.1e-6 of max |V|
// (left-looking: 1.5e-7): the MFMAs accumulate 192 times into the strip value itself and truncate every addend to ITS exponent
// (tools/ubench/mfma_rounding.hip).  Kept as the record of that measurement; include after ekf_large_trsm16.h.
#pragma once
#include <type_traits>

namespace aslam
{
// RIGHT-LOOKING sweep (large_trsm_bf16r).  The left-looking sweep above re-reads and re-splits strip block j for every later block column
// (16 v_accvgpr_read + 88 VALU + 16 additions per 64x64 block), and with one wave per SIMD nothing overlaps: a wave issues in order and the VALU
// instructions ADD to the MFMAs' time (tools/ubench/trsm_bench.hip: second-half region 444 cycles with the MFMAs alone, 690 with the split).
// Right-looking, a solved block column X_j is split ONCE, at its closing block, kept (negated) in 24 VGPRs, and applied to every later block column
// while it is there:  strip(:, K) -= X_j L(K, j)^T  for K = j + 1 .. nb - 1 -- the MFMAs accumulate straight into the strip's AGPR tiles (accumulator
// operand and destination in AGPRs), so a history block is 48 MFMAs, 24 operand-row reads and 6 DMA pieces and NOTHING ELSE.  The strip starts as
// the rows themselves (G), loaded once; block column j is final when its turn comes:  X_j = strip(:, j) Linv_j^T.
// The blocks of L stream COLUMN by column:  Linv_0, L(1,0) .. L(nb-1,0), Linv_1, L(2,1) ..  (same planes, same five-buffer pipeline).
// 17 block columns but 256 AGPRs = 16 tiles of 16: block column 16 takes the registers of block column 0, which is consumed by the first closing
// block (its rows wait in VGPRs until then).
namespace t16
{
#define ASLAM_T16_QA_OUT "=&{v[160:175]}"(R.QA0), "=&{v[176:191]}"(R.QA1), "=&{v[192:207]}"(R.QA2)
#define ASLAM_T16_PA_OUT "=&{v[96:111]}"(R.PA0), "=&{v[112:127]}"(R.PA1), "=&{v[128:143]}"(R.PA2)

/// strip registers a[16 K .. 16 K + 15] -> x (accumulator layout)
template <int K> __device__ __forceinline__ void strip_read16(f4 (&x)[4])
{
        float lo[8], hi[8];
        strip_read8<16 * K>(lo);
        strip_read8<16 * K + 8>(hi);
#pragma unroll
        for (int r = 0; r < 4; ++r)
                x[0][r] = lo[r], x[1][r] = lo[4 + r], x[2][r] = hi[r], x[3][r] = hi[4 + r];
}

/// cursor over the column-major block sequence (j, j), (j + 1, j), .., (nb - 1, j), (j + 1, j + 1), ..  on the planes
struct SeqR
{
        int j, K, nb, NP, wave;
        __amdgpu_buffer_rsrc_t rs;
        unsigned v0, v1, v2, v3, v4, v5;
        __device__ __forceinline__ SeqR(const Planes &pl, int b, int nb_, int NP_, int tid)
            : j(0), K(0), nb(nb_), NP(NP_), wave(__builtin_amdgcn_readfirstlane(tid >> 6)),
              rs(__builtin_amdgcn_make_buffer_rsrc(uniform_ptr(pl.Lq(b, NP_)), 0, (int)(Planes::per_filter(NP_) * 2), 0x00020000))
        {
                const int l = tid & 63, r = l >> 3, lc = (l & 7) ^ r;
                const unsigned ps = (unsigned)(NP_ * NP_ * 2), r8 = (unsigned)(8 * NP_ * 2);
                v0 = (unsigned)((r * NP_ + 8 * lc) * 2) + (unsigned)(tid >> 6) * r8;
                v1 = v0 + 4u * r8, v2 = v0 + ps, v3 = v1 + ps, v4 = v2 + ps, v5 = v3 + ps;
        }
        __device__ __forceinline__ Dma next()
        {
                Dma dm;
                dm.rsrc = rs;
                dm.v0 = v0, dm.v1 = v1, dm.v2 = v2, dm.v3 = v3, dm.v4 = v4, dm.v5 = v5;
                dm.so = (unsigned)(((LB * K) * NP + LB * j) * 2);
                const bool down = K + 1 < nb, right = !down && j + 1 < nb; // (past the end: the last block again)
                j += right ? 1 : 0;
                K = down ? K + 1 : (right ? j : K);
                return dm;
        }
        __device__ __forceinline__ void issue(unsigned short *dst)
        {
                typedef __attribute__((address_space(3))) unsigned short lds_us;
                const Dma dm = next();
                const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)dst + (unsigned)wave * 1024u));
                const unsigned vo[6] = {dm.v0, dm.v1, dm.v2, dm.v3, dm.v4, dm.v5};
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)dm.so);
#pragma unroll
                for (int i = 0; i < 6; ++i)
                        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %3 offen lds"
                                     :
                                     : "s"(ldsw), "n"((i >> 1) * 8192 + (i & 1) * 4096), "v"(vo[i]), "s"(so), "s"(dm.rsrc)
                                     : "m0", "scc", "memory");
        }
};

/// history block (K, j): strip(:, K) -= X_j L(K, j)^T, the pieces of -X_j in the B registers of sets P (columns 0 .. 31 of block j) and Q (32 .. 63).
/// On entry set P holds the first-half operand rows of this block, on exit those of the next block of the sequence.
template <int K, int STAMP> __device__ __forceinline__ void rblock(Regs &R, Pipe &pp, SeqR &seq, int a_h0, int a_h1, bool after_closing)
{
        typedef __attribute__((address_space(3))) unsigned short lds_us;
        constexpr int R0 = 16 * (K & 15); // (block column 16 lives in the registers of block column 0)
        const Dma dm = seq.next();
        const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)pp.b4 + (unsigned)seq.wave * 1024u));
        const unsigned a_cur = (unsigned)(uintptr_t)(lds_us *)(pp.b0 + a_h1);
        asm volatile(ASLAM_T16_R0 : ASLAM_T16_QA_OUT : ASLAM_T16_P_IN, ASLAM_T16_COMMON(a_cur, R0, 0) : "m0", "scc", "memory");
        pp.template stamp<STAMP>(0);
        const unsigned a_nxt = (unsigned)(uintptr_t)(lds_us *)(pp.b1 + a_h0);
        asm volatile(ASLAM_T16_R1 : ASLAM_T16_PA_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, R0, 3) : "m0", "scc", "memory");
        pp.template stamp<STAMP>(2);
        pp.end_dyn(after_closing);
        pp.template stamp<STAMP>(3);
}

template <int K, int STAMP> __device__ __forceinline__ void rchain(Regs &R, int j, int nb, Pipe &pp, SeqR &seq, int a_h0, int a_h1)
{
        if (K < nb)
        {
                if (K > j)
                        rblock<K, STAMP>(R, pp, seq, a_h0, a_h1, K == j + 1);
                if constexpr (K + 1 < LARGE_NB_MAX)
                        rchain<K + 1, STAMP>(R, j, nb, pp, seq, a_h0, a_h1);
        }
}

/// The right-looking sweep of one 16-row strip per wave: rows (row stride NP floats, this lane's row + 4 lg at `rowp`) -> X = rows L^-T, in place.
template <int STAMP> __device__ __forceinline__ void sweep16r(Regs &R, Pipe &pp, unsigned short (*lds)[BLK], const Planes &pl, int b, int nb, int NP, float *rowp, int tid)
{
        typedef __attribute__((address_space(3))) unsigned short lds_us;
        const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int a_h0 = li * PLD + 8 * (lg ^ (li & 7)), a_h1 = li * PLD + 8 * ((4 + lg) ^ (li & 7));
        SeqR seq(pl, b, nb, NP, tid);
        pp.b0 = lds[0], pp.b1 = lds[1], pp.b2 = lds[2], pp.b3 = lds[3], pp.b4 = lds[4];
        seq.issue(pp.b0);
        seq.issue(pp.b1);
        seq.issue(pp.b2);
        seq.issue(pp.b3);
        // the strip = the rows
        f4 g16[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        {
                auto put = [&](auto kc) {
                        constexpr int KK = decltype(kc)::value;
                        if (KK < nb)
                        {
                                f4 x[4];
#pragma unroll
                                for (int t = 0; t < 4; ++t)
                                        x[t] = *reinterpret_cast<const f4 *>(rowp + LB * KK + 16 * t);
                                if constexpr (KK < 16)
                                        strip_write16<KK>(x);
                                else
                                {
#pragma unroll
                                        for (int t = 0; t < 4; ++t)
                                                g16[t] = x[t];
                                }
                        }
                };
                put(std::integral_constant<int, 0>());
                put(std::integral_constant<int, 1>());
                put(std::integral_constant<int, 2>());
                put(std::integral_constant<int, 3>());
                put(std::integral_constant<int, 4>());
                put(std::integral_constant<int, 5>());
                put(std::integral_constant<int, 6>());
                put(std::integral_constant<int, 7>());
                put(std::integral_constant<int, 8>());
                put(std::integral_constant<int, 9>());
                put(std::integral_constant<int, 10>());
                put(std::integral_constant<int, 11>());
                put(std::integral_constant<int, 12>());
                put(std::integral_constant<int, 13>());
                put(std::integral_constant<int, 14>());
                put(std::integral_constant<int, 15>());
                put(std::integral_constant<int, 16>());
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        load_planes(R.PA0, R.PA1, R.PA2, pp.b0, a_h0);
        if constexpr (STAMP)
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pp.tlast)::"memory");
#pragma unroll 1
        for (int j = 0; j < nb; ++j)
        {
                // ---- the closing block of column j: X = strip(:, j) Linv_j^T.  Set P holds the first-half rows of Linv_j.
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); // (the last MFMAs into the strip -> v_accvgpr_read)
                f4 x[4];
                switch (j)
                {
#define ASLAM_T16_TAKE(J)                                                                                              \
        case J:                                                                                                        \
                strip_read16<(J) & 15>(x);                                                                             \
                break;
                        ASLAM_T16_TAKE(0)
                        ASLAM_T16_TAKE(1)
                        ASLAM_T16_TAKE(2)
                        ASLAM_T16_TAKE(3)
                        ASLAM_T16_TAKE(4)
                        ASLAM_T16_TAKE(5)
                        ASLAM_T16_TAKE(6)
                        ASLAM_T16_TAKE(7)
                        ASLAM_T16_TAKE(8)
                        ASLAM_T16_TAKE(9)
                        ASLAM_T16_TAKE(10)
                        ASLAM_T16_TAKE(11)
                        ASLAM_T16_TAKE(12)
                        ASLAM_T16_TAKE(13)
                        ASLAM_T16_TAKE(14)
                        ASLAM_T16_TAKE(15)
                default:
                        strip_read16<0>(x); // block column 16
                        break;
#undef ASLAM_T16_TAKE
                }
                pp.template stamp<STAMP>(9);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                R.run[4 * t + r] = x[t][r];
                {
                        const float c01[8] = {R.run[0], R.run[1], R.run[2], R.run[3], R.run[4], R.run[5], R.run[6], R.run[7]};
                        split8(c01, R.Pbh, R.Pbm, R.Pbl);
                }
                pp.template stamp<STAMP>(5);
                {
                        const Dma dm = seq.next();
                        const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)pp.b4 + (unsigned)seq.wave * 1024u));
                        const unsigned a_cur = (unsigned)(uintptr_t)(lds_us *)(pp.b0 + a_h1), a_nxt = (unsigned)(uintptr_t)(lds_us *)(pp.b1 + a_h0);
                        asm volatile(ASLAM_T16_C0 : "=&{v[32:47]}"(R.e), "+{v[64:79]}"(R.run), ASLAM_T16_Q_OUT : ASLAM_T16_P_IN, ASLAM_T16_COMMON(a_cur, 0, 0) : ASLAM_T16_SCRATCH);
                        pp.template stamp<STAMP>(6);
                        asm volatile(ASLAM_T16_C1R : "+{v[32:47]}"(R.e), ASLAM_T16_PA_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, 0, 3) : ASLAM_T16_SCRATCH);
                }
                pp.template stamp<STAMP>(7);
                asm volatile("s_nop 15" : "+v"(R.e)); // MFMA results -> VALU / stores
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        x[t] = tile4(R.e, t);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        *reinterpret_cast<f4 *>(rowp + LB * j + 16 * t) = x[t];
                pp.template stamp<STAMP>(11);
                {
                        // the pieces of -X_j: the B operands of this column's history blocks
                        const float n0[8] = {-x[0][0], -x[0][1], -x[0][2], -x[0][3], -x[1][0], -x[1][1], -x[1][2], -x[1][3]};
                        const float n1[8] = {-x[2][0], -x[2][1], -x[2][2], -x[2][3], -x[3][0], -x[3][1], -x[3][2], -x[3][3]};
                        split8(n0, R.Pbh, R.Pbm, R.Pbl);
                        split8(n1, R.Qbh, R.Qbm, R.Qbl);
                }
                if (j == 0 && nb == LARGE_NB_MAX)
                        strip_write16<0>(g16); // block column 16 moves into the registers block column 0 has just left
                pp.template stamp<STAMP>(8);
                pp.template end<16>();
                pp.template stamp<STAMP>(4);
                rchain<1, STAMP>(R, j, nb, pp, seq, a_h0, a_h1);
        }
}
} // namespace t16

/// V = G L^-T on the bf16 pipe, right-looking (see above).  Same grid and workgroup -> (filter, row block) map as large_trsm_bf16.
template <int NBMAX, int STAMP = 0>
__global__ __launch_bounds__(256, 1) void large_trsm_bf16r(DevView d, LargeView<float> lv, t16::Planes pl, int nfilters, const int *skipped)
{
        using namespace t16;
        static_assert(NBMAX == 17, "the chain lists 17 block columns");
        __shared__ __attribute__((aligned(1024))) unsigned short lds[NBUF][BLK];
        const int NP = lv.NP, nblk = NP / LB;
        const int slot = blockIdx.x >> 3;
        const int b = (slot / nblk) * 8 + (blockIdx.x & 7), rb = slot % nblk;
        if (b >= nfilters || skipped[b])
                return;
        const int n = d.n[b];
        const int nb = large_blocks(n);
        if (rb >= nb)
                return;
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        float *rowp = lv.G + (size_t)b * NP * NP + (size_t)(LB * rb + 16 * wave + li) * NP + 4 * lg;
        asm volatile("" ::: "a0", "a255"); // the strip
        Pipe pp;
        Regs R;
        asm volatile("; ASLAM_STRIP_LIVE_BEGIN" ::: "memory");
        sweep16r<STAMP>(R, pp, lds, pl, b, nb, NP, rowp, tid);
        asm volatile("; ASLAM_STRIP_LIVE_END" ::: "memory");
        if constexpr (STAMP)
                if (tid == 0 && blockIdx.x == 0)
                        for (int i = 0; i < 13; ++i)
                                lv.Y[i] = (double)pp.ph[i];
}

} // namespace aslam
