// ekf_large_trsm16.h -- V = G L^-T (and the block rows of the Cholesky factor of S) with the binary32 products formed on the BF16 matrix
// pipe (round 3).  Same recurrence and the same register-resident strip as ekf_large_trsm.h:
//      V(i, k) = ( G(i, k) - sum_{j<k} V(i, j) L(k, j)^T ) Linv_k^T
// a wave owns 16 rows and keeps their solved strip, TRANSPOSED, as binary32 accumulator tiles in AGPRs a0 .. a255 (tile T = a[4T .. 4T+3],
// register r of lane l = V[row l & 15][16 T + 4 (l >> 4) + r]).  What changes is the product: every float is the sum of three bf16 pieces
// (a = a1 + a2 + a3, round to nearest at every level) and a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1): six
// v_mfma_f32_16x16x32_bf16 (16 cycles each) per 16x16x32 product in place of eight v_mfma_f32_16x16x4_f32 (32 cycles each) -- a 64x64
// history block costs 48 MFMAs = 768 cycles of matrix pipe instead of 64 = 2048, at a smaller error (tools/ubench/mfma_bf16x3.hip).
//
//   * The blocks of L arrive ALREADY SPLIT: the kernels that produce L (large_chol_bf16, or large_split_planes for the multi-workgroup chain)
//     write three bf16 planes per block with the columns permuted so that the A operand of an MFMA -- row l & 15 of the block, the eight
//     contraction slots of lane group l >> 4 -- is ONE 16-byte LDS read: position 32 h + 8 g + 4 w + r holds column 32 h + 16 w + 4 g + r
//     (h = half of the block, g = lane group, w = which of the two strip tiles of the half, r = register).
//   * The B operand is the strip: the eight contraction slots of a lane are its OWN four registers of two consecutive strip tiles, read from the
//     AGPRs and split by the VALU (8 v_accvgpr_read + 44 VALU per 16x32 operand, serving the 24 MFMAs of a half block): no value crosses lanes.
//   * Accumulation: a block sums its 48 MFMAs in accumulators that start at zero and the VALU adds the block's sums to the column's running sum
//     -- an MFMA truncates small addends it adds to a large accumulator (tools/ubench/mfma_rounding.hip); running sums over a whole block
//     column were the source of the round-2 bias.
//   * The inner loop is hand-scheduled: a half block is ONE asm statement (tools/gen_trsm16_regions.py -> ekf_large_trsm16_regions.inc) whose
//     operands are bound to fixed register tuples; the blocks of L arrive by LDS-DMA into five buffers, their pieces issued inside those
//     statements and counted with s_waitcnt vmcnt (Pipe).  A wave is alone on its SIMD and issues in order, so every VALU instruction of the split
//     adds to the MFMAs' time: DESIGN.md section 6 and profiles/r03_experiments.md section 3 have the cycle budget of a block.
// The translation unit that includes this header for its kernels (aslam_large16.hip) is compiled with -mllvm -amdgpu-mfma-vgpr-form: with the AGPRs
// reserved for the strip hipcc would otherwise pick the AGPR form of every MFMA builtin.
#pragma once

namespace aslam
{
namespace t16
{
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

// A staged block in LDS: three planes of 64 unpadded 128-byte rows (the blocks arrive by LDS-DMA, which writes 1 KiB = eight whole rows per
// wave-instruction, lane l at byte 16 l: rows cannot be padded).  The eight 16-byte chunks of row r are XOR-swizzled with r & 7 -- applied on the
// SOURCE side of the DMA (lane l fetches the chunk that belongs at its place) and in the operand reads -- so that the 16 rows of an operand read
// spread over all eight chunk positions (2-way conflicts instead of 16-way).
constexpr int PLD = LB;           // bf16 per LDS row
constexpr int PLANE = LB * PLD;   // elements of one plane of a staged block
constexpr int BLK = 3 * PLANE;    // a staged block: three planes, 24 576 bytes
constexpr int NBUF = 5;           // LDS buffers: block i multiplied, i + 1 complete (its first-half rows are read during block i), i + 2 landing, i + 3 / i + 4 in flight

typedef LPlanes Planes; // (ekf_large.h)
__host__ __device__ __forceinline__ int perm_pos(int c)
{
        return lplane_pos(c);
}

/// two floats -> the three bf16 pieces of each, packed (lo = first): round to nearest even at every level
__device__ __forceinline__ void split2(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
        const f2v v = {a, b};
        const bf2 hh = __builtin_convertvector(v, bf2);
        h = __builtin_bit_cast(unsigned, hh);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const bf2 mm = __builtin_convertvector((f2v){ra, rb}, bf2);
        m = __builtin_bit_cast(unsigned, mm);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        const bf2 ll = __builtin_convertvector((f2v){sa, sb}, bf2);
        l = __builtin_bit_cast(unsigned, ll);
}

/// eight floats (the contraction slots of this lane, in slot order) -> three packed operands
__device__ __forceinline__ void split8(const float (&x)[8], u4v &h, u4v &m, u4v &l)
{
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
                unsigned hh, mm, ll;
                split2(x[2 * e], x[2 * e + 1], hh, mm, ll);
                h[e] = hh, m[e] = mm, l[e] = ll;
        }
}

/// strip registers a[R0 .. R0+7] -> x (two consecutive tiles: the eight contraction slots of a half block)
template <int R0> __device__ __forceinline__ void strip_read8(float (&x)[8])
{
        static_assert(R0 >= 0 && R0 + 7 < 256, "strip register");
        asm volatile("v_accvgpr_read_b32 %0, a%c8\n\tv_accvgpr_read_b32 %1, a%c9\n\tv_accvgpr_read_b32 %2, a%c10\n\tv_accvgpr_read_b32 %3, a%c11\n\t"
                     "v_accvgpr_read_b32 %4, a%c12\n\tv_accvgpr_read_b32 %5, a%c13\n\tv_accvgpr_read_b32 %6, a%c14\n\tv_accvgpr_read_b32 %7, a%c15\n\t"
                     "s_nop 1"
                     : "=v"(x[0]), "=v"(x[1]), "=v"(x[2]), "=v"(x[3]), "=v"(x[4]), "=v"(x[5]), "=v"(x[6]), "=v"(x[7])
                     : "n"(R0), "n"(R0 + 1), "n"(R0 + 2), "n"(R0 + 3), "n"(R0 + 4), "n"(R0 + 5), "n"(R0 + 6), "n"(R0 + 7));
}

/// x (four tiles of a solved block column, accumulator layout) -> strip registers a[16 K .. 16 K + 15]
template <int K> __device__ __forceinline__ void strip_write16(const f4 (&x)[4])
{
        asm volatile("v_accvgpr_write_b32 a%c8, %0\n\tv_accvgpr_write_b32 a%c9, %1\n\tv_accvgpr_write_b32 a%c10, %2\n\tv_accvgpr_write_b32 a%c11, %3\n\t"
                     "v_accvgpr_write_b32 a%c12, %4\n\tv_accvgpr_write_b32 a%c13, %5\n\tv_accvgpr_write_b32 a%c14, %6\n\tv_accvgpr_write_b32 a%c15, %7"
                     :
                     : "v"(x[0][0]), "v"(x[0][1]), "v"(x[0][2]), "v"(x[0][3]), "v"(x[1][0]), "v"(x[1][1]), "v"(x[1][2]), "v"(x[1][3]),
                       "n"(16 * K), "n"(16 * K + 1), "n"(16 * K + 2), "n"(16 * K + 3), "n"(16 * K + 4), "n"(16 * K + 5), "n"(16 * K + 6),
                       "n"(16 * K + 7));
        asm volatile("v_accvgpr_write_b32 a%c8, %0\n\tv_accvgpr_write_b32 a%c9, %1\n\tv_accvgpr_write_b32 a%c10, %2\n\tv_accvgpr_write_b32 a%c11, %3\n\t"
                     "v_accvgpr_write_b32 a%c12, %4\n\tv_accvgpr_write_b32 a%c13, %5\n\tv_accvgpr_write_b32 a%c14, %6\n\tv_accvgpr_write_b32 a%c15, %7\n\t"
                     "s_nop 3"
                     :
                     : "v"(x[2][0]), "v"(x[2][1]), "v"(x[2][2]), "v"(x[2][3]), "v"(x[3][0]), "v"(x[3][1]), "v"(x[3][2]), "v"(x[3][3]),
                       "n"(16 * K + 8), "n"(16 * K + 9), "n"(16 * K + 10), "n"(16 * K + 11), "n"(16 * K + 12), "n"(16 * K + 13),
                       "n"(16 * K + 14), "n"(16 * K + 15));
}

/// this lane's operand rows of half h: row 16 t + li, chunk (4 h + lg) ^ (li & 7), planes 0 .. 2.  a_row = li * PLD (elements)
template <int T0 = 0> __device__ __forceinline__ void load_frags(u4v (&A)[4][3], const unsigned short *buf, int a_h)
{
#pragma unroll
        for (int t = T0; t < 4; ++t)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                        A[t][p] = *reinterpret_cast<const u4v *>(buf + a_h + 16 * t * PLD + p * PLANE);
}

/// a wave-uniform pointer the compiler computed with vector instructions (64-bit multiplies) -> scalar registers
template <typename T> __device__ __forceinline__ T *uniform_ptr(T *p)
{
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (T *)(((unsigned long long)hi << 32) | lo);
}

/// One block's LDS-DMA descriptor: the regions issue its six one-KiB pieces per wave (piece i: plane i >> 1, rows 8 (4 (i & 1) + wave) .. + 7 of the
/// block; lane l = row l >> 3 of the piece, the 16 bytes that belong at chunk position l & 7 of that row: logical chunk (l & 7) ^ (l >> 3)).  Every
/// block of the planes has the same strides (LPlanes), so a lane's six byte offsets inside a block are constants of the kernel and a block is ONE
/// scalar: the byte offset of its first element.
struct Dma
{
        __amdgpu_buffer_rsrc_t rsrc; // the filter's planes
        unsigned so;                 // byte offset of the block
        unsigned v0, v1, v2, v3, v4, v5; // this lane's byte offsets of the six pieces inside a block
};

/// cursor over the block sequence  Linv_0; L(1,0), Linv_1; L(2,0), L(2,1), Linv_2; ... (as TrsmSeq) on the planes = blocks (k, 0 .. k), k = 0 .. nb - 1.
/// Scalar state only (and the six lane offsets).
struct Seq
{
        int k, j, nb, NP, wave;
        int diag = 0; // Cholesky, diagonal block column: only the history blocks L(k, 0 .. k-1) exist -- the inverse of block k is written AFTER this
                      // sweep, and a prefetch of it now would leave a stale copy in the CU's L1 for the next block row to hit
        __amdgpu_buffer_rsrc_t rs; // the filter's planes
        unsigned v0, v1, v2, v3, v4, v5;
        __device__ __forceinline__ Seq(const Planes &pl, int b, int k0, int nb_, int NP_, int tid)
            : k(k0), j(0), nb(nb_), NP(NP_), wave(__builtin_amdgcn_readfirstlane(tid >> 6)),
              rs(__builtin_amdgcn_make_buffer_rsrc(uniform_ptr(pl.Lq(b, NP_)), 0, (int)(Planes::per_filter(NP_) * 2), 0x00020000))
        {
                const int l = tid & 63, r = l >> 3, lc = (l & 7) ^ r;
                const unsigned ps = (unsigned)(NP_ * NP_ * 2), r8 = (unsigned)(8 * NP_ * 2);
                v0 = (unsigned)((r * NP_ + 8 * lc) * 2) + (unsigned)(tid >> 6) * r8;
                v1 = v0 + 4u * r8, v2 = v0 + ps, v3 = v1 + ps, v4 = v2 + ps, v5 = v3 + ps;
        }
        /// the next block of the sequence
        __device__ __forceinline__ Dma next()
        {
                if (diag && j >= k)
                        j = max(k - 1, 0); // (past the end of the diagonal column's history: the last block again)
                Dma dm;
                dm.rsrc = rs;
                dm.v0 = v0, dm.v1 = v1, dm.v2 = v2, dm.v3 = v3, dm.v4 = v4, dm.v5 = v5;
                dm.so = (unsigned)(((LB * k) * NP + LB * j) * 2);
                const bool hist = j < k, adv = !hist && k + 1 < nb;
                j = hist ? j + 1 : (adv ? 0 : j);
                k += adv ? 1 : 0;
                return dm;
        }
        /// outside the regions (the start of the pipeline): all six pieces of the next block -> LDS buffer `dst`
        __device__ __forceinline__ void issue(unsigned short *dst)
        {
                typedef __attribute__((address_space(3))) unsigned short lds_us;
                const Dma dm = next();
                const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)dst + (unsigned)wave * 1024u));
                const unsigned vo[6] = {dm.v0, dm.v1, dm.v2, dm.v3, dm.v4, dm.v5};
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)dm.so); // (wave-uniform, but hipcc may have formed it in a vector register)
#pragma unroll
                for (int i = 0; i < 6; ++i)
                        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, %3 offen lds"
                                     :
                                     : "s"(ldsw), "n"((i >> 1) * 8192 + (i & 1) * 4096), "v"(vo[i]), "s"(so), "s"(dm.rsrc)
                                     : "m0", "scc", "memory");
        }
};

/// Block pipeline over FIVE LDS buffers fed by LDS-DMA: block i is multiplied out of b0; block i + 1 must be COMPLETE in b1 throughout block i (its
/// first-half operand rows are read behind the last MFMAs of block i); so at the end of block i every wave waits for its pieces of block i + 2 --
/// at most the twelve youngest of its DMA pieces outstanding: blocks i + 3 and i + 4 (loads complete in order; a stricter count never hurts) -- and
/// the barrier publishes block i + 2.  The six pieces a wave moves of block i + 4 are issued INSIDE the regions of block i, between their MFMAs
/// (issued together the 24 KB of a block keep the CU's one address unit busy for ~ 400 - 600 cycles), into the buffer block i - 1 left at the last
/// barrier.  (Round 3's first version had four buffers and the same counts: the first-half rows of the next block could be read before they had
/// landed -- right on an idle chip, wrong with every CU busy; tools/ubench/trsm_bench.hip now compares all filters of a batch of identical inputs
/// bit for bit.)
struct Pipe
{
        unsigned short *b0, *b1, *b2, *b3, *b4;
        // diagnostic builds (STAMP): shader cycles by phase -- 0 first-half region, 1 between the halves, 2 second-half region, 3 end (barrier), 4 closing block
        unsigned long long ph[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0; // 5 .. 8: inside the closing block (C + split, first half, second half, strip write); 9 .. 12: the wait for the slice of the rows, its LDS reads + the issue of the next slice, the stores, the vmcnt wait at the end
        template <int STAMP> __device__ __forceinline__ void stamp(int i)
        {
                if constexpr (STAMP)
                {
                        unsigned long long t;
                        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                        ph[i] += t - tlast;
                        tlast = t;
                }
        }
        /// LATER: vector-memory instructions this wave has issued after its pieces of block i + 2 -- two blocks' pieces = 12, plus, around a closing
        /// block, the four loads of the next slice of G and the closing block's S stores (4 of V; the Cholesky's 6 plane stores on top): loads and
        /// stores retire in issue order on this counter, so counting the stores keeps the wait from covering pieces issued only one block ago
        /// (with 16 the closing block stood ~ 770 cycles at this wait: tools/ubench/trsm_bench.hip, phase stamps).
        template <int LATER = 12> __device__ __forceinline__ void end()
        {
                static_assert(LATER >= 12 && LATER <= 63, "vmcnt count");
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LATER) : "memory");
                rotate();
        }
        /// the same with the count chosen at run time (wave-uniform): 16 in the block that follows a closing block (its four stores), else 12
        __device__ __forceinline__ void end_dyn(bool after_closing)
        {
                if (after_closing)
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else
                        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                rotate();
        }
        __device__ __forceinline__ void rotate()
        {
                unsigned short *t = b0;
                b0 = b1;
                b1 = b2;
                b2 = b3;
                b3 = b4;
                b4 = t;
        }
};

#include "ekf_large_trsm16_regions.inc"

typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u16v __attribute__((ext_vector_type(16)));

/// Everything the hand-scheduled regions (tools/gen_trsm16_regions.py) keep in fixed physical registers.  P / Q: the two operand sets (a region
/// consumes one and fills the other: A = operand rows of the staged block, one 16-register tuple per plane, row tile t in registers 4 t .. 4 t + 3;
/// bh / bm / bl = the pieces of the B operand); e / o: the block-local sums of the even / odd history blocks; run: the column's running sum.
struct Regs
{
        u16v PA0, PA1, PA2, QA0, QA1, QA2;
        u4v Pbh, Pbm, Pbl, Qbh, Qbm, Qbl;
        f16v e, o, run;
};

#define ASLAM_T16_SCRATCH "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "m0", "scc", "memory"
#define ASLAM_T16_Q_OUT "=&{v[160:175]}"(R.QA0), "=&{v[176:191]}"(R.QA1), "=&{v[192:207]}"(R.QA2), "=&{v[208:211]}"(R.Qbh), "=&{v[212:215]}"(R.Qbm), "=&{v[216:219]}"(R.Qbl)
#define ASLAM_T16_P_OUT "=&{v[96:111]}"(R.PA0), "=&{v[112:127]}"(R.PA1), "=&{v[128:143]}"(R.PA2), "=&{v[144:147]}"(R.Pbh), "=&{v[148:151]}"(R.Pbm), "=&{v[152:155]}"(R.Pbl)
#define ASLAM_T16_P_IN "{v[96:111]}"(R.PA0), "{v[112:127]}"(R.PA1), "{v[128:143]}"(R.PA2), "{v[144:147]}"(R.Pbh), "{v[148:151]}"(R.Pbm), "{v[152:155]}"(R.Pbl)
#define ASLAM_T16_Q_IN "{v[160:175]}"(R.QA0), "{v[176:191]}"(R.QA1), "{v[192:207]}"(R.QA2), "{v[208:211]}"(R.Qbh), "{v[212:215]}"(R.Qbm), "{v[216:219]}"(R.Qbl)
/// operands every region takes: the LDS address of the operand rows it reads, the bf16 mask, the first strip register of the B operand it prepares,
/// and the LDS-DMA pieces it issues (pieces H .. H + 2 of descriptor dm into the buffer at LDS address ldsw)
#define ASLAM_T16_VO0 [vo0] "v"(dm.v0), [vo1] "v"(dm.v1), [vo2] "v"(dm.v2)
#define ASLAM_T16_VO3 [vo0] "v"(dm.v3), [vo1] "v"(dm.v4), [vo2] "v"(dm.v5)
#define ASLAM_T16_COMMON(addr, R0, H) [lds] "v"(addr), [msk] "s"(0xffff0000u), [r0] "n"(R0), [ldsw] "s"(ldsw), [rsrc] "s"(dm.rsrc), [so] "s"(dm.so), ASLAM_T16_VO##H

/// One history block J (even J: sums in R.e, odd J: in R.o), set P holding the operands of its first half on entry and of the NEXT block's first half
/// on exit.  Inside the MFMA gaps of the first half: the strip registers of the second half are read and split, its operand rows are read from LDS
/// (-> set Q), and the finished sums of the previous block are added to the column's running sum; inside those of the second half: the same for the
/// first half of the next block in the stream (history block J + 1 of this column, whose operand rows are in b1 -- or the closing block, whose B
/// operand is not a strip tile: that split is wasted).  Both halves carry three of the six DMA pieces of the block three ahead.
template <int J, int STAMP, int NST> __device__ __forceinline__ void history_block(Regs &R, Pipe &pp, Seq &seq, int a_h0, int a_h1, int tid)
{
        typedef __attribute__((address_space(3))) unsigned short lds_us;
        const Dma dm = seq.next();
        const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)pp.b4 + (unsigned)seq.wave * 1024u));
        const unsigned a_cur = (unsigned)(uintptr_t)(lds_us *)(pp.b0 + a_h1); // the second half of this block
        if constexpr (J == 0)
                asm volatile(ASLAM_T16_H0_E : "=&{v[32:47]}"(R.e), ASLAM_T16_Q_OUT : ASLAM_T16_P_IN, ASLAM_T16_COMMON(a_cur, 16 * J + 8, 0) : ASLAM_T16_SCRATCH);
        else if constexpr (J % 2 == 0)
                asm volatile(ASLAM_T16_H0_E_ADD
                             : "=&{v[32:47]}"(R.e), "+{v[64:79]}"(R.run), ASLAM_T16_Q_OUT
                             : ASLAM_T16_P_IN, "{v[48:63]}"(R.o), ASLAM_T16_COMMON(a_cur, 16 * J + 8, 0)
                             : ASLAM_T16_SCRATCH);
        else
                asm volatile(ASLAM_T16_H0_O_ADD
                             : "=&{v[48:63]}"(R.o), "+{v[64:79]}"(R.run), ASLAM_T16_Q_OUT
                             : ASLAM_T16_P_IN, "{v[32:47]}"(R.e), ASLAM_T16_COMMON(a_cur, 16 * J + 8, 0)
                             : ASLAM_T16_SCRATCH);
        pp.template stamp<STAMP>(0);
        const unsigned a_nxt = (unsigned)(uintptr_t)(lds_us *)(pp.b1 + a_h0); // the first half of the next block
        constexpr int RN = (J + 1 < LARGE_NB_MAX - 1) ? 16 * (J + 1) : 0;
        pp.template stamp<STAMP>(1);
        if constexpr (STAMP == 2)
                asm volatile(ASLAM_T16_H1_E_NOVALU : "+{v[32:47]}"(R.e), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, RN, 3) : ASLAM_T16_SCRATCH);
        else if constexpr (STAMP == 3)
                asm volatile(ASLAM_T16_H1_E_NODS : "+{v[32:47]}"(R.e), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, RN, 3) : ASLAM_T16_SCRATCH);
        else if constexpr (STAMP == 5) // (the block's pieces 3 .. 5 are never fetched: timing only)
                asm volatile(ASLAM_T16_H1_E_NODMA : "+{v[32:47]}"(R.e), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, RN, 3) : ASLAM_T16_SCRATCH);
        else if constexpr (STAMP == 4)
                asm volatile(ASLAM_T16_H1_E_BARE : "+{v[32:47]}"(R.e), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, RN, 3) : ASLAM_T16_SCRATCH);
        else if constexpr (J % 2 == 0)
                asm volatile(ASLAM_T16_H1_E : "+{v[32:47]}"(R.e), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, RN, 3) : ASLAM_T16_SCRATCH);
        else
                asm volatile(ASLAM_T16_H1_O : "+{v[48:63]}"(R.o), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, RN, 3) : ASLAM_T16_SCRATCH);
        pp.template stamp<STAMP>(2);
        pp.template end<(J == 0) ? 16 + NST : (J == 1) ? 12 + NST : 12>(); // (NST: the stores of the closing block one / two blocks back)
        pp.template stamp<STAMP>(3);
}

template <int J, int STAMP, int NST> __device__ __forceinline__ void chain(Regs &R, int k, Pipe &pp, Seq &seq, int a_h0, int a_h1, int tid)
{
        if (J < k)
        {
                history_block<J, STAMP, NST>(R, pp, seq, a_h0, a_h1, tid);
                if constexpr (J + 1 < LARGE_NB_MAX - 1)
                        chain<J + 1, STAMP, NST>(R, k, pp, seq, a_h0, a_h1, tid);
        }
}

/// tile t of a 16-register tuple
template <typename V16> __device__ __forceinline__ auto tile4(const V16 &v, int t)
{
        typedef decltype(v[0] + v[0]) E;
        typedef E v4 __attribute__((ext_vector_type(4)));
        return t == 0   ? (v4){v[0], v[1], v[2], v[3]}
               : t == 1 ? (v4){v[4], v[5], v[6], v[7]}
               : t == 2 ? (v4){v[8], v[9], v[10], v[11]}
                        : (v4){v[12], v[13], v[14], v[15]};
}

__device__ __forceinline__ void load_planes(u16v &A0, u16v &A1, u16v &A2, const unsigned short *buf, int a_h)
{
        u4v A[4][3];
        load_frags(A, buf, a_h);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                        A0[4 * t + r] = A[t][0][r], A1[4 * t + r] = A[t][1][r], A2[4 * t + r] = A[t][2][r];
}

} // namespace t16

namespace t16
{
/// what a sweep needs to know about the rows it solves: where they come from (a buffer resource + the byte offset of the wave's first row: the 16 x 64
/// slices arrive by LDS-DMA), where the solved columns go in binary32 (this lane's row, + 4 lg), and -- Cholesky only -- where they go as bf16 planes
struct Rows
{
        __amdgpu_buffer_rsrc_t rsrc; // the matrix the rows live in (G, or S for the Cholesky)
        unsigned so0;                // byte offset of row 0 of this wave's 16 rows
        float *out;                  // this lane's row, + 4 lg: the binary32 stores
        __amdgpu_buffer_rsrc_t rq;   // CHOL: the filter's planes of L
        unsigned vq;                 // CHOL: this lane's byte offset inside a plane (row, + 8 lg elements)
        unsigned qplane;             // CHOL: bytes per plane
};

/// The sweep of one 16-row strip per wave over block columns 0 .. nbk - 1:  X(:, k) = (rows(:, k) - sum_{j<k} X(:, j) L(k, j)^T) Linv_k^T, kept in the
/// strip and written out.  CHOL (large_chol_bf16: the rows are block row nbk of S itself): block column nbk follows -- its history blocks are
/// L(nbk, j) = the X(:, j) this very sweep has just stored as planes, so every store is drained and the block pipeline restarted in front of them --
/// and the sweep returns  c = S(nbk, nbk) - sum_j X(:, j) X(:, j)^T  (this wave's 16 rows: c[t][r] = column 16 t + 4 lg + r of row li) for the
/// caller to factor.  `lds`: four block buffers; `gl`: this wave's 16 x 64 floats for the slices of the rows.
/// F32OUT (CHOL only): also store the solved columns in binary32 (the L of large_trsm_pipe and of the diagnostics); the default chain reads L through its
/// planes only, and the Cholesky is HBM-bound: 2.1 MB of stores per filter less.
template <int STAMP, bool CHOL, bool F32OUT = true>
__device__ __forceinline__ void sweep16(f4 (&cdiag)[4], Regs &R, Pipe &pp, unsigned short (*lds)[BLK], float *gl, const Planes &pl, int b, int nbk, int nb_rows, int NP,
                                        const Rows &rows, int tid)
{
        typedef __attribute__((address_space(3))) float lds_f;
        typedef __attribute__((address_space(3))) unsigned short lds_us;
        const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
        // The slices of the rows arrive by LDS-DMA like the blocks of L (four one-KiB pieces per wave and block column: its 16 rows x 64 columns,
        // piece i = rows 4 i .. 4 i + 3, lane l = row l >> 4, 16-byte chunk l & 15), so that every load of the loop is counted by the same
        // s_waitcnt vmcnt arithmetic: a load the compiler issues makes hipcc wait on ITS count of outstanding loads, which knows nothing of the
        // DMA pieces in flight and drains them.
        const unsigned g_vo = (unsigned)(((lane >> 4) * NP + 4 * (lane & 15)) * 4);
        const unsigned g_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_f *)gl);
        const unsigned g_rs4 = (unsigned)(4 * NP * 4);
        auto g_issue = [&](int kcol) { // this wave's 16 x 64 slice of block column kcol -> gl
#pragma unroll
                for (int i = 0; i < 4; ++i)
                        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                                     :
                                     : "s"(g_lds), "n"(i * 1024), "v"(g_vo), "s"(rows.rsrc), "s"(rows.so0 + (unsigned)i * g_rs4 + (unsigned)(LB * kcol * 4))
                                     : "m0", "scc", "memory");
        };
        const int a_h0 = li * PLD + 8 * (lg ^ (li & 7)), a_h1 = li * PLD + 8 * ((4 + lg) ^ (li & 7)); // this lane's operand rows, half 0 / 1 (swizzled chunk)
        Seq seq(pl, b, 0, nbk, NP, tid);
        auto start = [&]() { // (re)start of the block pipeline at the block `seq` points at: blocks i .. i + 3 -> LDS; the caller waits and synchronises
                pp.b0 = lds[0], pp.b1 = lds[1], pp.b2 = lds[2], pp.b3 = lds[3], pp.b4 = lds[4];
                seq.issue(pp.b0);
                seq.issue(pp.b1);
                seq.issue(pp.b2);
                seq.issue(pp.b3);
        };
        if (!CHOL || nbk > 0)
        {
                start();
                g_issue(0);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                load_planes(R.PA0, R.PA1, R.PA2, pp.b0, a_h0);
                R.Pbh = R.Pbm = R.Pbl = (u4v){0u, 0u, 0u, 0u};
        }
        if constexpr (STAMP)
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pp.tlast)::"memory");
        const int klast = CHOL ? nbk : nbk - 1;
        static_assert(CHOL || F32OUT, "V is stored in binary32");
        constexpr int NST = (F32OUT ? 4 : 0) + (CHOL ? 6 : 0); // vector-memory stores of a closing block per wave (binary32: 4; the planes of L: 6)
#pragma unroll 1
        for (int k = 0; k <= klast; ++k)
        {
                if (CHOL && k == nbk)
                {
                        // the history blocks of the diagonal block column are this workgroup's own output: every store has to have left the CU's memory
                        // pipeline, and every wave has to be done with the LDS buffers, before the pipeline restarts on them
                        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                        seq.k = nbk, seq.j = 0, seq.nb = nbk + 1, seq.diag = 1; // L(nbk, 0), ..., L(nbk, nbk - 1), then the last one again
                        if (nbk > 0)
                                start();
                        g_issue(nbk);
                        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                        if (nbk > 0)
                        {
                                load_planes(R.PA0, R.PA1, R.PA2, pp.b0, a_h0);
                                float s8[8];
                                strip_read8<0>(s8);
                                split8(s8, R.Pbh, R.Pbm, R.Pbl);
                        }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                        R.run[i] = R.e[i] = R.o[i] = 0.f;
                chain<0, STAMP, NST>(R, k, pp, seq, a_h0, a_h1, tid);
                // ---- the closing block of column k: C = rows - history (running sum + the sums of the last history block, which no later block has
                // added), X = C Linv_k^T (Linv is lower triangular in tiles: output tile t needs c tiles <= t).  Set P holds the first-half rows of
                // Linv_k.  This column's slice of the rows was issued one block column ago.
                // (k >= 1: issued at the top of closing block k - 1 -- its DMA pieces, its stores and the pieces of >= 1 history block are younger;
                // k = 0 and the Cholesky's diagonal column: everything was waited for at the (re)start)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + NST) : "memory");
                pp.template stamp<STAMP>(9);
                f4 g0[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        g0[t] = *reinterpret_cast<const f4 *>(&gl[li * LB + 16 * t + 4 * lg]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(CHOL && k == nbk))
                        g_issue(CHOL ? k + 1 : min(k + 1, nbk - 1)); // the next slice (ahead of this block's pieces of L: Pipe::end counts on that order)
                pp.template stamp<STAMP>(10);
                asm volatile("s_nop 15" : "+v"(R.e), "+v"(R.o)); // MFMA results -> VALU
                {
                        const bool last_even = (k & 1) != 0; // history block k - 1
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                        {
                                const f4 last = last_even ? tile4(R.e, t) : tile4(R.o, t);
                                const f4 ct = k > 0 ? g0[t] - (tile4(R.run, t) + last) : g0[t]; // (the select keeps hipcc from hoisting the sums: without it the Cholesky parks registers in the strip's AGPRs)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                        R.run[4 * t + r] = ct[r]; // C takes the place of the running sum
                        }
                }
                if (CHOL && k == nbk)
                {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                                cdiag[t] = tile4(R.run, t); // (only here: a value written in every column would stay live through the regions)
                        break;
                }
                {
                        const float c01[8] = {R.run[0], R.run[1], R.run[2], R.run[3], R.run[4], R.run[5], R.run[6], R.run[7]};
                        split8(c01, R.Pbh, R.Pbm, R.Pbl);
                }
                pp.template stamp<STAMP>(5);
                {
                        const Dma dm = seq.next();
                        const unsigned ldsw = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)pp.b4 + (unsigned)seq.wave * 1024u));
                        const unsigned a_cur = (unsigned)(uintptr_t)(lds_us *)(pp.b0 + a_h1), a_nxt = (unsigned)(uintptr_t)(lds_us *)(pp.b1 + a_h0);
                        // first half; behind it C tiles 2, 3 are split (-> set Q) and the second-half rows of Linv_k are read
                        asm volatile(ASLAM_T16_C0 : "=&{v[32:47]}"(R.e), "+{v[64:79]}"(R.run), ASLAM_T16_Q_OUT : ASLAM_T16_P_IN, ASLAM_T16_COMMON(a_cur, 0, 0) : ASLAM_T16_SCRATCH);
                        pp.template stamp<STAMP>(6);
                        // second half (row tiles 2, 3); behind it the first operand of the next block column (history block 0: strip tiles 0, 1 -- for
                        // k = 0 they are produced right here and read again below)
                        asm volatile(ASLAM_T16_C1 : "+{v[32:47]}"(R.e), ASLAM_T16_P_OUT : ASLAM_T16_Q_IN, ASLAM_T16_COMMON(a_nxt, 0, 3) : ASLAM_T16_SCRATCH);
                }
                pp.template stamp<STAMP>(7);
                asm volatile("s_nop 15" : "+v"(R.e)); // MFMA results -> VALU / stores
                f4 x[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        x[t] = tile4(R.e, t);
                if constexpr (F32OUT)
                {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                                *reinterpret_cast<f4 *>(rows.out + LB * k + 16 * t) = x[t];
                }
                if constexpr (CHOL)
                {
                        // the planes of L(nbk, k): row tiles 2 u and 2 u + 1 of a lane are eight consecutive positions of a permuted plane row
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                        {
                                u2x h0, m0, l0, h1, m1, l1;
                                split_bf16x3(x[2 * u], h0, m0, l0);
                                split_bf16x3(x[2 * u + 1], h1, m1, l1);
                                const int so = (LB * k + 32 * u) * 2;
                                __builtin_amdgcn_raw_buffer_store_b128((u4v){h0[0], h0[1], h1[0], h1[1]}, rows.rq, (int)rows.vq, so, 0);
                                __builtin_amdgcn_raw_buffer_store_b128((u4v){m0[0], m0[1], m1[0], m1[1]}, rows.rq, (int)rows.vq, so + (int)rows.qplane, 0);
                                __builtin_amdgcn_raw_buffer_store_b128((u4v){l0[0], l0[1], l1[0], l1[1]}, rows.rq, (int)rows.vq, so + 2 * (int)rows.qplane, 0);
                        }
                }
                pp.template stamp<STAMP>(11);
                if (k < 16) // (the last block column is never a history block)
                {
                        // x -> strip registers a[16 k ..]: a computed jump (tools/gen_trsm16_regions.py); x is still in the accumulator registers
                        const int ks = __builtin_amdgcn_readfirstlane(k);
                        asm volatile(ASLAM_T16_STRIP_WRITE_DYN : : [k] "s"(ks), "{v[32:47]}"(R.e) : "s96", "s97", "s98", "scc");
                }
                if (k == 0)
                {
                        float s8[8];
                        strip_read8<0>(s8);
                        split8(s8, R.Pbh, R.Pbm, R.Pbl);
                }
                pp.template stamp<STAMP>(8);
                if constexpr (STAMP)
                {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 + NST) : "memory");
                        pp.template stamp<STAMP>(12);
                }
                pp.template end<16 + NST>();
                pp.template stamp<STAMP>(4);
        }
        (void)nb_rows;
}
} // namespace t16

/// V = G L^-T on the bf16 pipe.  grid (8 * ceil(B / 8) * NP / 64), 256 threads; wave w of a workgroup owns 16 rows of G; in place: G -> V.
/// Same workgroup -> (filter, row block) map as large_trsm_pipe (a filter's workgroups share one XCD).
template <int NBMAX, int STAMP = 0>
__global__ __launch_bounds__(256, 1) void large_trsm_bf16(DevView d, LargeView<float> lv, t16::Planes pl, int nfilters, const int *skipped)
{
        using namespace t16;
        static_assert(NBMAX == 17, "the chain lists 17 block columns");
        __shared__ __attribute__((aligned(1024))) unsigned short lds[NBUF][BLK];
        __shared__ __attribute__((aligned(1024))) float gl[4][16 * LB];
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
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        Rows rows;
        rows.rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(lv.G + (size_t)b * NP * NP), 0, NP * NP * 4, 0x00020000);
        rows.so0 = (unsigned)((LB * rb + 16 * wv) * NP * 4);
        rows.out = lv.G + (size_t)b * NP * NP + (size_t)(LB * rb + 16 * wave + li) * NP + 4 * lg;
        rows.rq = rows.rsrc, rows.vq = 0, rows.qplane = 0;
        asm volatile("" ::: "a0", "a255"); // the strip
        Pipe pp;
        Regs R;
        f4 c[4];
        asm volatile("; ASLAM_STRIP_LIVE_BEGIN vmem: global_store_dwordx4=4" ::: "memory"); // (tools/check_vmcnt_protocol.py: what Pipe's counts assume)
        sweep16<STAMP, false>(c, R, pp, lds, gl[wv], pl, b, nb, nb, NP, rows, tid);
        asm volatile("; ASLAM_STRIP_LIVE_END" ::: "memory");
        if constexpr (STAMP)
                if (tid == 0 && blockIdx.x == 0)
                        for (int i = 0; i < 13; ++i)
                                lv.Y[i] = (double)pp.ph[i];
}

/// S = L L^T on the bf16 pipe: one workgroup per filter, block row after block row (large_chol_resident's structure, ekf_large_chol.h), every block
/// row a sweep16<CHOL> over the planes of the block rows above it -- which this kernel writes as it goes (binary32 L and Linv are written too) --
/// closed by the 64x64 factorisation of the diagonal block in binary64 tiles.  grid (B), 256 threads.  Status bit 4 (ASLAM_ST_NOT_PD) on a
/// non-positive pivot.
/// F32OUT = false (the default chain, whose TRSM streams the planes): the off-diagonal blocks of L are not stored in binary32 at all.
template <int NBMAX, bool F32OUT = true>
__global__ __launch_bounds__(256, 1) void large_chol_bf16(DevView d, LargeView<float> lv, t16::Planes pl, const int *skipped)
{
        using namespace t16;
        static_assert(NBMAX == 17, "the chain lists 17 block columns");
        constexpr int PIPE_BYTES = NBUF * BLK * 2, TILE_BYTES = chol64::TILES * TSZ * (int)sizeof(double);
        static_assert(TILE_BYTES <= PIPE_BYTES, "the tiles of the diagonal factorisation alias the block buffers");
        __shared__ __attribute__((aligned(1024))) unsigned short lds[NBUF][BLK];
        __shared__ __attribute__((aligned(1024))) float gl[4][16 * LB];
        double *tiles = reinterpret_cast<double *>(&lds[0][0]);
        const int b = blockIdx.x;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n);
        const int tid0 = threadIdx.x;
        const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
        float *Sb = lv.S + (size_t)b * NP * NP;
        float *Linv = lv.Linv + (size_t)b * LARGE_NB_MAX * LB * LB;
        asm volatile("" ::: "a0", "a255"); // the strip
        Rows rows;
        rows.rsrc = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Sb), 0, NP * NP * 4, 0x00020000);
        rows.rq = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(pl.Lq(b, NP)), 0, 3 * NP * NP * 2, 0x00020000);
        rows.qplane = (unsigned)(NP * NP * 2);
        bool ok = true;
        Pipe pp;
        Regs R;
#pragma unroll 1
        for (int I = 0; I < nb; ++I)
        {
                // one opaque re-definition of the thread index per block row: hipcc otherwise hoists every tid-derived address of the loop body out
                // of the loop, runs out of VGPRs and parks the overflow in AGPRs -- the strip's (tools/check_agpr_strip.py)
                int tid = tid0;
                asm volatile("" : "+v"(tid));
                const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
                rows.so0 = (unsigned)((LB * I + 16 * wv) * NP * 4);
                rows.out = Sb + (size_t)(LB * I + 16 * wave + li) * NP + 4 * lg;
                rows.vq = (unsigned)(((LB * I + 16 * wave + li) * NP + 8 * lg) * 2);
                f4 c[4];
                if constexpr (F32OUT)
                        asm volatile("; ASLAM_STRIP_LIVE_BEGIN vmem: global_store_dwordx4=4 buffer_store_dwordx4=6" ::: "memory"); // (tools/check_vmcnt_protocol.py)
                else
                        asm volatile("; ASLAM_STRIP_LIVE_BEGIN vmem: buffer_store_dwordx4=6" ::: "memory");
                sweep16<0, true, F32OUT>(c, R, pp, lds, gl[wv], pl, b, I, nb, NP, rows, tid);
                asm volatile("; ASLAM_STRIP_LIVE_END" ::: "memory");
                // every DMA piece still in flight targets the buffers the tiles are about to take
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                // C (this wave's 16 rows: tile row `wave`) -> binary64 tiles, lower block triangle
#pragma unroll
                for (int t = 0; t < 4; ++t)
                {
                        if (t <= wave)
                        {
                                double *T = chol64::Lt(tiles, wave, t) + li * TLD + 4 * lg;
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                        T[r] = (double)c[t][r];
                        }
                }
                ok = chol64::factor_and_invert<true>(tiles, tid) && ok;
                chol64::store_block(tiles, Sb + ((size_t)LB * I) * NP + (size_t)LB * I, NP, Linv + (size_t)I * LB * LB, tid, pl.block(b, NP, I, I));
                // the next block row reads the planes back through the block pipeline
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
        }
        if (!ok && tid0 == 0)
                atomicOr(&d.status[b], 4u); // ASLAM_ST_NOT_PD
}

/// L (lower block triangle of S after the factorisation) and Linv -> their bf16 planes (the form large_trsm_bf16 streams).  For the chains whose
/// Cholesky kernels still write binary32 (the multi-workgroup chain of small batches; the fp32-MFMA resident Cholesky).  grid (NP / 64 + 1, B),
/// 256 threads: workgroup x < NP / 64 splits block row x of L (blocks j < x), workgroup NP / 64 the inverses of the diagonal blocks.
template <int UNUSED = 0> // (a template: the header is included by both translation units of the library)
__global__ __launch_bounds__(256) void large_split_planes(DevView d, LargeView<float> lv, t16::Planes pl, const int *skipped)
{
        using namespace t16;
        const int b = blockIdx.y;
        if (skipped[b])
                return;
        const int NP = lv.NP, nblk = NP / LB, nb = large_blocks(d.n[b]);
        const int tid = threadIdx.x;
        auto put = [&](const float *src, unsigned short *dst, size_t pstride) {
                // src: 4 consecutive floats (columns c .. c+3 of a block, c = 4 (tid & 15)); dst: the permuted position of c in plane 0
                const f4 v = *reinterpret_cast<const f4 *>(src);
                unsigned h0, m0, l0, h1, m1, l1;
                split2(v[0], v[1], h0, m0, l0);
                split2(v[2], v[3], h1, m1, l1);
                *reinterpret_cast<u2v *>(dst) = (u2v){h0, h1};
                *reinterpret_cast<u2v *>(dst + pstride) = (u2v){m0, m1};
                *reinterpret_cast<u2v *>(dst + 2 * pstride) = (u2v){l0, l1};
        };
        const int c = 4 * (tid & 15), r0 = tid >> 4;
        if ((int)blockIdx.x < nblk)
        {
                const int I = blockIdx.x;
                if (I >= nb)
                        return;
                const float *S = lv.S + (size_t)b * NP * NP + (size_t)LB * I * NP;
                unsigned short *Lq = pl.Lq(b, NP) + (size_t)LB * I * NP;
                for (int j = 0; j < I; ++j)
                        for (int r = r0; r < LB; r += 16)
                                put(S + (size_t)r * NP + LB * j + c, Lq + (size_t)r * NP + LB * j + perm_pos(c), (size_t)NP * NP);
        }
        else
        {
                const float *Li = lv.Linv + (size_t)b * LARGE_NB_MAX * LB * LB;
                for (int k = 0; k < nb; ++k)
                        for (int r = r0; r < LB; r += 16)
                                put(Li + (size_t)k * LB * LB + r * LB + c, pl.block(b, NP, k, k) + (size_t)r * NP + perm_pos(c), (size_t)NP * NP);
        }
}
} // namespace aslam
