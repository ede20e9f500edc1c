// ekf_large_trsm.h -- V = G L^-T for the large-state EKF in binary32 (K = P H^T S^-1 = V L^-1, ekf.cpp:301), with the
// solved block columns kept in REGISTERS.
//
// Once S = L L^T is factored (blocked Cholesky, 64-wide block columns, inverses of the diagonal blocks in lv.Linv) every row of
// G is independent:  V(i, k) = ( G(i, k) - sum_{j<k} V(i, j) L(k, j)^T ) Linv_k^T.   A left-looking sweep that keeps V in HBM
// re-reads the solved columns once per block column (38 MB per filter and callback at n = 1027, the largest item of the
// fabric traffic of round 1).  Here a wave owns 16 rows of G and holds their whole solved row strip -- 17 block columns x 64
// = 1088 columns -- as 68 MFMA accumulator tiles (272 registers): G is read once, V is written once, and only L streams by.
//
// The strip is held TRANSPOSED: tile q of a wave is W = V^T restricted to rows [16 q, 16 q + 16) x the wave's 16 columns,
// in the 16x16x4 accumulator layout  reg r of lane l = W[16 q + 4 (l >> 4) + r][l & 15].  In that layout register r of a
// finished tile IS the B operand (k = l >> 4, n = l & 15) of a later product that contracts over W's row index, for the four
// rows k' = 4 (l >> 4) + r; the matching A operand (i = l & 15, k = l >> 4) is L[i][16 q + 4 (l >> 4) + r]: four consecutive
// floats of a row of L, i.e. ONE 16-byte LDS read feeds four MFMAs, and no accumulator is ever moved, transposed or
// re-read.  (The contraction index is visited in a permuted order; both operands use the same permutation.)
//
// The 64x64 blocks of L -- Linv_0; L(1,0), Linv_1; L(2,0), L(2,1), Linv_2; ... -- are shared by the four waves of a workgroup and
// stream through three LDS buffers (large_trsm_pipe below).  Register indices must be static, and with ONE wave per SIMD nothing hides
// a taken branch: the history blocks of a block column are a fall-through chain (template recursion over the block index) whose only
// taken branch is its exit -- 14 KB of code that stays in the instruction cache.  One workgroup per CU (the strip needs most of the
// register file), 64 MFMAs per wave and block.  n^3 flops like any triangular solve with n right-hand sides, plus the 64-deep Linv
// products.  What the earlier forms cost (profiles/r02_experiments.md section 3, tools/ubench/trsm_bench.hip; cycles per MFMA and
// wave, 32 = the pipe's rate): 153 blocks unrolled 55 (instruction-cache misses), loops + switch 51, chain 46, this file 44.
#pragma once

namespace aslam
{
constexpr int TRSM_LDT = LB + 8; // LDS row stride in floats: 18 sixteen-byte slots -> conflict-free ds_read_b128 operand reads

/// registers -> LDS buffer [64][TRSM_LDT]
__device__ __forceinline__ void trsm_stash(float *buf, const f4 (&pf)[4], int tid)
{
        const int r0 = tid >> 4, c4 = (tid & 15) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f4 *>(buf + (r0 + 16 * q) * TRSM_LDT + c4) = pf[q];
}

// ---- the strip lives in AGPRs a0 .. a255 that the compiler never sees: every MFMA of this kernel is inline assembly with
// VGPR ("v") accumulators and A operands, so no value of the compiler's is ever placed in an AGPR; the strip tiles are named
// by number (tile T = registers a[4T .. 4T+3]) as B operands and written with v_accvgpr_write.  (Left to the compiler -- a
// 64-tile array indexed statically inside the switch cases -- the 256-value phi web around the two loops ended in 1 800
// v_accvgpr copies and 520 spilled registers.)  hipcc inserts no wait states inside or around inline assembly: the blocks end
// in the s_nop a following VALU / store read of an MFMA result needs (16x16x4 f32: 8 passes).

// ---- MFMA groups that carry the LDS traffic of the block pipeline INSIDE their instruction stream.  A wave that is alone on its SIMD
// pays ~10 cycles of idle matrix pipe for every instruction that sits between two MFMA groups (60 instructions per block were 9 of 41
// cycles per MFMA), while LDS, scalar and vector-memory instructions placed between the MFMAs of a group issue in its shadow.  A group
// of 16 MFMAs (strip tile T = 4 k-steps x 4 row tiles) is two statements: the first half (k-steps 0, 1) starts with the four 16-byte
// fragment reads a LATER group needs -- row tiles t = 0 .. 3 of column tile Q at LDS byte address `rd` -- and ends with the wait for
// them (eight MFMAs = 256 cycles later: they have arrived), so that the compiler never sees a register with a load in flight; the second
// half (k-steps 2, 3) is MFMAs only.

/// first half, plain
template <int T, int Q>
__device__ __forceinline__ void trsm_half_a(f4 &c0, f4 &c1, f4 &c2, f4 &c3, const f4 &a0, const f4 &a1, const f4 &a2, const f4 &a3, f4 (&o)[4], unsigned rd)
{
        static_assert(T >= 0 && T < 64 && TRSM_LDT * 16 * 4 == 4608, "strip tile; ds offsets");
        asm volatile("ds_read_b128 %4, %18 offset:0+%c19\n\t"
                     "ds_read_b128 %5, %18 offset:4608+%c19\n\t"
                     "ds_read_b128 %6, %18 offset:9216+%c19\n\t"
                     "ds_read_b128 %7, %18 offset:13824+%c19\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %8, a%c16, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %10, a%c16, %1\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %12, a%c16, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %14, a%c16, %3\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %9, a%c17, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %11, a%c17, %1\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %13, a%c17, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %15, a%c17, %3\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                     : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(a2[0]), "v"(a2[1]), "v"(a3[0]), "v"(a3[1]), "n"(4 * T), "n"(4 * T + 1), "v"(rd), "n"(64 * Q)
                     : "memory");
}

/// first half + the read of the synchronisation counter (TrsmPipe): `seen` is a scalar when the statement ends
template <int T, int Q>
__device__ __forceinline__ void trsm_half_a_peek(f4 &c0, f4 &c1, f4 &c2, f4 &c3, const f4 &a0, const f4 &a1, const f4 &a2, const f4 &a3, f4 (&o)[4], unsigned rd,
                                                 unsigned ctr_lds, unsigned &seen)
{
        static_assert(T >= 0 && T < 64, "strip tile");
        unsigned vtmp, sval;
        asm volatile("ds_read_b32 %8, %22\n\t"
                     "ds_read_b128 %4, %20 offset:0+%c21\n\t"
                     "ds_read_b128 %5, %20 offset:4608+%c21\n\t"
                     "ds_read_b128 %6, %20 offset:9216+%c21\n\t"
                     "ds_read_b128 %7, %20 offset:13824+%c21\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %10, a%c18, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %12, a%c18, %1\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %14, a%c18, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %16, a%c18, %3\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %11, a%c19, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %13, a%c19, %1\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %15, a%c19, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %17, a%c19, %3\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "v_readfirstlane_b32 %9, %8"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(vtmp), "=s"(sval)
                     : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(a2[0]), "v"(a2[1]), "v"(a3[0]), "v"(a3[1]), "n"(4 * T), "n"(4 * T + 1), "v"(rd), "n"(64 * Q), "v"(ctr_lds)
                     : "memory");
        seen = sval;
}

/// first half + the stash of the prefetched block (four ds_write_b128 per thread: rows r0 + 16 q of an LDS buffer, byte address `lds` for
/// q = 0) + the signal of TrsmPipe: one lane adds 1 to the counter, queued behind this wave's reads of the block and its four ds_writes.
/// The compiler sees pf as plain inputs and waits for their global loads in front of the statement.
template <int T, int Q>
__device__ __forceinline__ void trsm_half_a_stash(f4 &c0, f4 &c1, f4 &c2, f4 &c3, const f4 &a0, const f4 &a1, const f4 &a2, const f4 &a3, f4 (&o)[4], unsigned rd,
                                                  const f4 (&pf)[4], unsigned lds, unsigned ctr_lds)
{
        static_assert(T >= 0 && T < 64, "strip tile");
        asm volatile("ds_read_b128 %4, %18 offset:0+%c19\n\t"
                     "ds_read_b128 %5, %18 offset:4608+%c19\n\t"
                     "ds_read_b128 %6, %18 offset:9216+%c19\n\t"
                     "ds_read_b128 %7, %18 offset:13824+%c19\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %8, a%c16, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %10, a%c16, %1\n\t"
                     "ds_write_b128 %24, %20\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %12, a%c16, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %14, a%c16, %3\n\t"
                     "ds_write_b128 %24, %21 offset:4608\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %9, a%c17, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %11, a%c17, %1\n\t"
                     "ds_write_b128 %24, %22 offset:9216\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %13, a%c17, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %15, a%c17, %3\n\t"
                     "ds_write_b128 %24, %23 offset:13824\n\t"
                     "s_mov_b64 exec, 1\n\t"
                     "ds_add_u32 %25, %26\n\t"
                     "s_mov_b64 exec, -1\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
                     : "v"(a0[0]), "v"(a0[1]), "v"(a1[0]), "v"(a1[1]), "v"(a2[0]), "v"(a2[1]), "v"(a3[0]), "v"(a3[1]), "n"(4 * T), "n"(4 * T + 1), "v"(rd), "n"(64 * Q), "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(lds), "v"(ctr_lds), "v"(1u)
                     : "memory");
}

/// second half: k-steps 2, 3
template <int T> __device__ __forceinline__ void trsm_half_b(f4 &c0, f4 &c1, f4 &c2, f4 &c3, const f4 &a0, const f4 &a1, const f4 &a2, const f4 &a3)
{
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, a%c12, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %6, a%c12, %1\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %8, a%c12, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %10, a%c12, %3\n\t"
                     "v_mfma_f32_16x16x4_f32 %0, %5, a%c13, %0\n\t"
                     "v_mfma_f32_16x16x4_f32 %1, %7, a%c13, %1\n\t"
                     "v_mfma_f32_16x16x4_f32 %2, %9, a%c13, %2\n\t"
                     "v_mfma_f32_16x16x4_f32 %3, %11, a%c13, %3"
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
                     : "v"(a0[2]), "v"(a0[3]), "v"(a1[2]), "v"(a1[3]), "v"(a2[2]), "v"(a2[3]), "v"(a3[2]), "v"(a3[3]), "n"(4 * T + 2), "n"(4 * T + 3));
}

/// the four 16-byte fragments (row tiles t = 0 .. 3) of column tile q of a staged 64x64 block
__device__ __forceinline__ void trsm_frags(f4 (&a)[4], const float *buf, int a_off, int q)
{
#pragma unroll
        for (int t = 0; t < 4; ++t)
                a[t] = *reinterpret_cast<const f4 *>(buf + a_off + 16 * t * TRSM_LDT + 16 * q);
}

/// x[tp] += Linv(tile tp, tile t) * C(tile t) for tp = t .. 3 (Linv is lower triangular in tiles): B operand = c in VGPRs
template <int NTP>
__device__ __forceinline__ void trsm_mfma_x(f4 &x0, f4 &x1, f4 &x2, f4 &x3, const f4 &a0, const f4 &a1, const f4 &a2, const f4 &a3, const f4 &c)
{
        // NTP = number of target tiles (4 - t); targets are the LAST NTP of x0..x3, fragments a0..a(NTP-1) belong to them in order
        if constexpr (NTP == 4)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %20, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %20, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %12, %20, %2\n\tv_mfma_f32_16x16x4_f32 %3, %16, %20, %3\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %5, %21, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %21, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %13, %21, %2\n\tv_mfma_f32_16x16x4_f32 %3, %17, %21, %3\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %6, %22, %0\n\tv_mfma_f32_16x16x4_f32 %1, %10, %22, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %14, %22, %2\n\tv_mfma_f32_16x16x4_f32 %3, %18, %22, %3\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %7, %23, %0\n\tv_mfma_f32_16x16x4_f32 %1, %11, %23, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %15, %23, %2\n\tv_mfma_f32_16x16x4_f32 %3, %19, %23, %3"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
                             : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(a2[0]),
                               "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(a3[0]), "v"(a3[1]), "v"(a3[2]), "v"(a3[3]), "v"(c[0]), "v"(c[1]),
                               "v"(c[2]), "v"(c[3]));
        else if constexpr (NTP == 3)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %3, %15, %0\n\tv_mfma_f32_16x16x4_f32 %1, %7, %15, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %11, %15, %2\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %4, %16, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %16, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %12, %16, %2\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %5, %17, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %17, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %13, %17, %2\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %6, %18, %0\n\tv_mfma_f32_16x16x4_f32 %1, %10, %18, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %2, %14, %18, %2"
                             : "+v"(x1), "+v"(x2), "+v"(x3)
                             : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(a2[0]),
                               "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
        else if constexpr (NTP == 2)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %10, %0\n\tv_mfma_f32_16x16x4_f32 %1, %6, %10, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %3, %11, %0\n\tv_mfma_f32_16x16x4_f32 %1, %7, %11, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %4, %12, %0\n\tv_mfma_f32_16x16x4_f32 %1, %8, %12, %1\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %5, %13, %0\n\tv_mfma_f32_16x16x4_f32 %1, %9, %13, %1"
                             : "+v"(x2), "+v"(x3)
                             : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]), "v"(c[0]),
                               "v"(c[1]), "v"(c[2]), "v"(c[3]));
        else
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %5, %0\n\tv_mfma_f32_16x16x4_f32 %0, %2, %6, %0\n\t"
                             "v_mfma_f32_16x16x4_f32 %0, %3, %7, %0\n\tv_mfma_f32_16x16x4_f32 %0, %4, %8, %0"
                             : "+v"(x3)
                             : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
}

/// strip tiles 4 K .. 4 K + 3 <- x
template <int K> __device__ __forceinline__ void trsm_keep(const f4 (&x)[4])
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

// ---------------------------------------------------------------------------------------------------------------------------------
// large_trsm_pipe: the sweep, software-pipelined ACROSS the per-block barrier.  With two LDS buffers and the barrier at the end of a
// block, every block starts with eight LDS reads whose latency nothing hides (one wave per SIMD, and the four waves of the workgroup
// issue their reads in the same cycles: ~300 of a block's ~2 350 cycles).  Here the blocks go through THREE LDS buffers: block i + 2
// is written while block i is multiplied, so block i + 1 is already visible and its first two fragment sets are read behind the last
// MFMAs of block i; the barrier sits 16 MFMAs behind the LDS traffic it waits for.

/// cursor over the block sequence  Linv_0; L(1,0), Linv_1; L(2,0), L(2,1), Linv_2; ...  up to Linv_{nb-1} (which is then fetched
/// again and again: the pipeline runs two blocks past the end).  Scalar state only, advanced with selects, and the loads are BUFFER
/// loads (resource + scalar byte offset of the block row + one 32-bit lane offset): with a single wave per SIMD every instruction
/// between two MFMA groups is a bubble; two taken branches per block, or 64-bit vector address arithmetic per block (~300 of a
/// block's ~2 800 cycles: trsm_bench, "no fetch" against "every fetch from one block"), cost more than the data.
struct TrsmSeq
{
        int k, j, nb, NP;
        unsigned offh, offl;       // byte offsets of L(k, 0) inside S and of Linv_k inside Linv (this filter's)
        unsigned voh, vol;         // this thread's byte offset inside a block of L (row stride NP) / of Linv (row stride 64): row tid >> 4, columns 4 (tid & 15) ..
        __amdgpu_buffer_rsrc_t rs, rl;
        int pin = 0; // diagnostic (trsm_bench, DIAG & 32): every fetch reads block 0 of S -- L1 hits instead of L2 / HBM
        __device__ __forceinline__ TrsmSeq(const float *Sb, const float *Linv, int k0, int nb_, int NP_, int tid)
            : k(k0), j(0), nb(nb_), NP(NP_), offh((unsigned)(LB * k0) * NP_ * 4u), offl((unsigned)k0 * LB * LB * 4u),
              voh((unsigned)(((tid >> 4) * NP_ + (tid & 15) * 4) * 4)), vol((unsigned)(((tid >> 4) * LB + (tid & 15) * 4) * 4)),
              rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Sb), 0, NP_ * NP_ * 4, 0x00020000)),
              rl(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Linv), 0, LARGE_NB_MAX * LB * LB * 4, 0x00020000))
        {
        }
        /// four 16-byte loads per thread: rows (tid >> 4) + 16 q of the block
        __device__ __forceinline__ void fetch(f4 (&pf)[4])
        {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const bool hist = j < k;
                const __amdgpu_buffer_rsrc_t r = hist ? rs : rl;
                const unsigned so = pin ? 0u : hist ? offh + (unsigned)(LB * 4 * j) : offl;
                const unsigned ldb = (unsigned)(hist ? NP : LB) * 64u; // bytes between rows r and r + 16
                const unsigned vo = hist ? voh : vol;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, (int)(so + q * ldb), 0);
                        pf[q] = (f4){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
                }
                const bool adv = !hist && k + 1 < nb;
                j = hist ? j + 1 : (adv ? 0 : j);
                k += adv ? 1 : 0;
                offh += adv ? (unsigned)LB * NP * 4u : 0u;
                offl += adv ? (unsigned)(LB * LB * 4) : 0u;
        }
};

struct TrsmPipe
{
        float *cur, *nxt, *far; // LDS buffers of block i, i + 1, i + 2
        // The per-block synchronisation of the four waves is a COUNTER in LDS, not s_barrier: a wave adds 1 behind its stash ("my reads of
        // this block's buffer and my part of block i + 2 are in the LDS queue": the LDS executes a wave's instructions in order, so when
        // the add lands they have landed) and checks, one block later and ahead of its next stash, that all four waves have done so.
        // Signal and check are 48 MFMAs apart, so skew between the waves is absorbed instead of stalling every wave at every block
        // (s_barrier cost 3 - 5 of 43 cycles per MFMA here: trsm_bench, "no barrier").
        unsigned *ctr;   // LDS word, zero at kernel start
        unsigned nsig;   // signals this wave has given
        unsigned seen;   // last value read
        unsigned lost = 0; // set when a wait gave up (bounded spin): the kernel then raises ASLAM_ST_INTERNAL for the filter
        __device__ __forceinline__ unsigned ctr_lds() const
        {
                typedef __attribute__((address_space(3))) unsigned lds_uint;
                return (unsigned)(uintptr_t)(lds_uint *)ctr;
        }
        __device__ __forceinline__ void rotate()
        {
                float *t = cur;
                cur = nxt;
                nxt = far;
                far = t;
        }
        __device__ __forceinline__ void signal(int lane)
        {
                if (lane == 0)
                        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                ++nsig;
        }
        /// issue the read of the counter (its latency hides behind the MFMAs that follow)
        __device__ __forceinline__ void peek()
        {
                seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
        /// all four waves have given `nsig` signals; the spin is bounded (a lost signal would otherwise hang the GPU).  Giving up leaves
        /// the LDS buffers racy, i.e. the filter's results invalid: `lost` is set and the kernels report it as the sticky status bit
        /// ASLAM_ST_INTERNAL.  Wave-uniform by construction (scalar compares on readfirstlane values): every call site runs with EXEC == -1,
        /// which the s_mov_b64 exec pairs of trsm_half_a_stash rely on.
        __device__ __forceinline__ void wait()
        {
                if (__builtin_expect(seen < 4u * nsig, 0)) // wave-uniform; the usual case falls through
                {
                        unsigned v = seen;
                        for (int spin = 0; v < 4u * nsig && spin < (1 << 22); ++spin)
                                v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (v < 4u * nsig)
                                lost = 1u;
                }
                asm volatile("" ::: "memory");
        }
};

/// one history block with fragments a0 (column tile 0) and a1 (column tile 1) of it already in registers; leaves a0 / a1 of the NEXT block
template <int J, int DIAG>
__device__ __forceinline__ void trsm_history_pipe(f4 (&c)[4], f4 (&a0)[4], f4 (&a1)[4], f4 (&pf)[4], TrsmPipe &pp, TrsmSeq &seq, int a_off, int tid)
{
        typedef __attribute__((address_space(3))) float lds_float;
        const unsigned rd_cur = (unsigned)(uintptr_t)(lds_float *)(pp.cur + a_off), rd_nxt = (unsigned)(uintptr_t)(lds_float *)(pp.nxt + a_off);
        f4 a2[4], a3[4]; // column tiles 2 and 3 of this block; a0 / a1 leave as column tiles 0 / 1 of the next one
        // group 0 (column tile 0), reading column tile 2 of this block
        trsm_half_a<4 * J + 0, 2>(c[0], c[1], c[2], c[3], a0[0], a0[1], a0[2], a0[3], a2, rd_cur);
        trsm_half_b<4 * J + 0>(c[0], c[1], c[2], c[3], a0[0], a0[1], a0[2], a0[3]);
        // group 1 (column tile 1), reading column tile 3 -- the last read of buffer `cur` -- and the synchronisation counter
        if constexpr (!(DIAG & 2) && !(DIAG & 16))
                trsm_half_a_peek<4 * J + 1, 3>(c[0], c[1], c[2], c[3], a1[0], a1[1], a1[2], a1[3], a3, rd_cur, pp.ctr_lds(), pp.seen);
        else
                trsm_half_a<4 * J + 1, 3>(c[0], c[1], c[2], c[3], a1[0], a1[1], a1[2], a1[3], a3, rd_cur);
        trsm_half_b<4 * J + 1>(c[0], c[1], c[2], c[3], a1[0], a1[1], a1[2], a1[3]);
        // The synchronisation point of this block (TrsmPipe).  It orders (a) the stash of block i + 1 (third group of the PREVIOUS block) before
        // the first read of that data (in the third group of this block) and (b) every wave's reads of the previous block's buffer before
        // the stash below overwrites it -- events a whole block apart.
        if constexpr (!(DIAG & 2) && !(DIAG & 16)) // (16: timing experiment without the synchronisation -- racy)
                pp.wait();
        else
                asm volatile("" ::: "memory");
        // group 2 (column tile 2): block i + 2 -> LDS (fetched while block i - 1 was multiplied) and the signal between its MFMAs, reading
        // column tile 0 of the NEXT block; block i + 3 -> registers behind it
        if constexpr (!(DIAG & 2))
        {
                const unsigned lds = (unsigned)(uintptr_t)(lds_float *)(pp.far + (tid >> 4) * TRSM_LDT + (tid & 15) * 4);
                trsm_half_a_stash<4 * J + 2, 0>(c[0], c[1], c[2], c[3], a2[0], a2[1], a2[2], a2[3], a0, rd_nxt, pf, lds, pp.ctr_lds());
                if constexpr (!(DIAG & 16))
                        ++pp.nsig;
        }
        else
                trsm_half_a<4 * J + 2, 0>(c[0], c[1], c[2], c[3], a2[0], a2[1], a2[2], a2[3], a0, rd_nxt);
        trsm_half_b<4 * J + 2>(c[0], c[1], c[2], c[3], a2[0], a2[1], a2[2], a2[3]);
        if constexpr (!(DIAG & 1))
                seq.fetch(pf);
        // group 3 (column tile 3), reading column tile 1 of the next block
        trsm_half_a<4 * J + 3, 1>(c[0], c[1], c[2], c[3], a3[0], a3[1], a3[2], a3[3], a1, rd_nxt);
        trsm_half_b<4 * J + 3>(c[0], c[1], c[2], c[3], a3[0], a3[1], a3[2], a3[3]);
        pp.rotate();
}

template <int J, int DIAG>
__device__ __forceinline__ void trsm_chain_pipe(f4 (&c)[4], f4 (&a0)[4], f4 (&a1)[4], f4 (&pf)[4], int k, TrsmPipe &pp, TrsmSeq &seq, int a_off, int tid)
{
        if (J < k)
        {
                trsm_history_pipe<J, DIAG>(c, a0, a1, pf, pp, seq, a_off, tid);
                if constexpr (J + 1 < LARGE_NB_MAX - 1)
                        trsm_chain_pipe<J + 1, DIAG>(c, a0, a1, pf, k, pp, seq, a_off, tid);
        }
}

/// (re)start of the block pipeline at the block `seq` points at: blocks i, i + 1 -> LDS, block i + 2 -> registers, the first two fragment
/// sets of block i -> a0 / a1.  The caller guarantees that no wave still reads the three buffers (a barrier since the last read).
__device__ __forceinline__ void trsm_pipe_start(f4 (&pf)[4], f4 (&a0)[4], f4 (&a1)[4], TrsmPipe &pp, TrsmSeq &seq, int a_off, int tid)
{
        // all three blocks are requested before the first is waited for: one exposed memory latency per start, not three (a0 / a1 are free here)
        seq.fetch(a0);
        seq.fetch(a1);
        seq.fetch(pf); // block i + 2: written to LDS during block i
        trsm_stash(pp.cur, a0, tid);
        trsm_stash(pp.nxt, a1, tid);
        __syncthreads();
        trsm_frags(a0, pp.cur, a_off, 0);
        trsm_frags(a1, pp.cur, a_off, 1);
}

/// The sweep of one 16-row strip per wave over block columns 0 .. nbk - 1:  X(:, k) = (G(:, k) - sum_{j<k} X(:, j) L(k, j)^T) Linv_k^T,
/// written over G and kept in the strip.  `rg`: buffer resource of the matrix the rows live in, `vg`: byte offset of this lane's row (+ 4 lg
/// floats) in it -- buffer loads / stores with a scalar column offset: no 64-bit vector address arithmetic between MFMA groups.
/// CHOL (large_chol_resident: the rows are block row nbk of S itself): block column nbk follows, closed differently -- its history
/// blocks are L(nbk, j) = the X(:, j) this very sweep has just stored, so the pipeline is drained and restarted in front of them, and
/// the sweep returns  c = S(nbk, nbk) - sum_j X(:, j) X(:, j)^T  (this wave's 16 rows: c[t][r] = column 16 t + 4 lg + r of row li)
/// for the caller to factor.  `seq` walks Linv_0; L(1,0), Linv_1; ... (nb = nbk), `seq_diag` (CHOL only) starts at L(nbk, 0).
/// `rq`, `vq`, `qplane` (CHOL only, qplane != 0): the solved blocks L(nbk, k) are ALSO written as three bf16 planes with permuted columns, the form
/// large_trsm_bf16 streams (ekf_large_trsm16.h: position 32 (t >> 1) + 8 lg + 4 (t & 1) + r holds column 16 t + 4 lg + r, so row tiles 2 u and
/// 2 u + 1 of a lane make one 16-byte store) -- buffer resource of the filter's planes, this lane's byte offset inside a plane (row, + 8 lg
/// elements), bytes per plane.
template <int DIAG, bool CHOL>
__device__ __forceinline__ void trsm_sweep(f4 (&c)[4], __amdgpu_buffer_rsrc_t rg, unsigned vg, int nbk, TrsmSeq &seq, const TrsmSeq &seq_diag, TrsmPipe &pp,
                                           int a_off, int tid, __amdgpu_buffer_rsrc_t rq, unsigned vq, unsigned qplane)
{
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        auto gload = [&](int col) { // four floats of this lane's row at column `col` (wave-uniform) + 4 lg
                const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rg, (int)vg, col * 4, 0);
                return (f4){__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        };
        f4 pf[4], a0[4], a1[4], g0[4];
        // From here to the end of the sweep the strip registers a0 .. a255 are live WITHOUT the compiler knowing: tools/check_agpr_strip.py
        // (run by the build on the assembly of this very compilation) rejects any compiler-generated AGPR use between the two markers.
        asm volatile("; ASLAM_STRIP_LIVE_BEGIN" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t)
        {
                c[t] = (f4){0.f, 0.f, 0.f, 0.f};
                g0[t] = gload(16 * t); // G[row][16 t + 4 lg .. +3]
        }
        if (!CHOL || nbk > 0)
                trsm_pipe_start(pf, a0, a1, pp, seq, a_off, tid);
        const int klast = CHOL ? nbk : nbk - 1;
#pragma unroll 1
        for (int k = 0; k <= klast; ++k)
        {
                if (CHOL && k == nbk)
                {
                        // the history blocks of the diagonal block column are this workgroup's own output: every store has to have left
                        // the CU's memory pipeline, and every wave has to be done with the LDS buffers, before the pipeline restarts
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                        seq = seq_diag; // L(nbk, 0), ..., L(nbk, nbk - 1)
                        trsm_pipe_start(pf, a0, a1, pp, seq, a_off, tid);
                }
                trsm_chain_pipe<0, DIAG>(c, a0, a1, pf, k, pp, seq, a_off, tid);
                // ---- the closing block of column k: C = G - history, X = Linv_k C (a0 = tiles (t, 0), a1 = tiles (t, 1) of Linv_k)
                asm volatile("s_nop 15" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])); // MFMA result -> VALU read
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        c[t] = g0[t] - c[t];
                if (CHOL && k == nbk)
                        break;
                asm volatile("s_nop 4" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])); // VALU result -> MFMA operand
                f4 x[4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        x[t] = (f4){0.f, 0.f, 0.f, 0.f};
                trsm_mfma_x<4>(x[0], x[1], x[2], x[3], a0[0], a0[1], a0[2], a0[3], c[0]);
                if constexpr (!(DIAG & 2) && !(DIAG & 16))
                        pp.peek();
                // fragments (2,2), (3,2), (3,3) of Linv_k
                a0[2] = *reinterpret_cast<const f4 *>(pp.cur + a_off + 32 * TRSM_LDT + 32);
                a0[3] = *reinterpret_cast<const f4 *>(pp.cur + a_off + 48 * TRSM_LDT + 32);
                a0[0] = *reinterpret_cast<const f4 *>(pp.cur + a_off + 48 * TRSM_LDT + 48);
                // the synchronisation point of this block (see trsm_history_pipe): the three reads above stay in flight
                if constexpr (!(DIAG & 2) && !(DIAG & 16))
                        pp.wait();
                else
                        asm volatile("" ::: "memory");
                if constexpr (!(DIAG & 2))
                {
                        trsm_stash(pp.far, pf, tid);
                        if constexpr (!(DIAG & 16))
                                pp.signal(tid & 63);
                }
                if constexpr (!(DIAG & 1))
                        seq.fetch(pf);
                trsm_mfma_x<3>(x[0], x[1], x[2], x[3], a1[1], a1[2], a1[3], a1[3], c[1]);
                trsm_mfma_x<2>(x[0], x[1], x[2], x[3], a0[2], a0[3], a0[3], a0[3], c[2]);
                trsm_frags(a1, pp.nxt, a_off, 1);
                asm volatile("" ::: "memory");
                trsm_mfma_x<1>(x[0], x[1], x[2], x[3], a0[0], a0[0], a0[0], a0[0], c[3]);
                // the first fragments of the next block (history block 0 of column k + 1)
                trsm_frags(a0, pp.nxt, a_off, 0);
                asm volatile("s_nop 15" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); // MFMA result -> store data
#pragma unroll
                for (int t = 0; t < 4; ++t)
                        __builtin_amdgcn_raw_buffer_store_b128((u4){__float_as_uint(x[t][0]), __float_as_uint(x[t][1]), __float_as_uint(x[t][2]), __float_as_uint(x[t][3])}, rg,
                                                               (int)vg, (LB * k + 16 * t) * 4, 0);
                if constexpr (CHOL)
                {
                        if (qplane) // wave-uniform
                        {
#pragma unroll
                                for (int u = 0; u < 2; ++u)
                                {
                                        u2x h0, m0, l0, h1, m1, l1;
                                        split_bf16x3(x[2 * u], h0, m0, l0);
                                        split_bf16x3(x[2 * u + 1], h1, m1, l1);
                                        const int so = (LB * k + 32 * u) * 2;
                                        __builtin_amdgcn_raw_buffer_store_b128((u4){h0[0], h0[1], h1[0], h1[1]}, rq, (int)vq, so, 0);
                                        __builtin_amdgcn_raw_buffer_store_b128((u4){m0[0], m0[1], m1[0], m1[1]}, rq, (int)vq, so + (int)qplane, 0);
                                        __builtin_amdgcn_raw_buffer_store_b128((u4){l0[0], l0[1], l1[0], l1[1]}, rq, (int)vq, so + 2 * (int)qplane, 0);
                                }
                        }
                }
                switch (k)
                {
#define ASLAM_TRSM_KEEP(K)                                                                                             \
        case K:                                                                                                        \
                trsm_keep<K>(x);                                                                                       \
                break;
                        ASLAM_TRSM_KEEP(0)
                        ASLAM_TRSM_KEEP(1)
                        ASLAM_TRSM_KEEP(2)
                        ASLAM_TRSM_KEEP(3)
                        ASLAM_TRSM_KEEP(4)
                        ASLAM_TRSM_KEEP(5)
                        ASLAM_TRSM_KEEP(6)
                        ASLAM_TRSM_KEEP(7)
                        ASLAM_TRSM_KEEP(8)
                        ASLAM_TRSM_KEEP(9)
                        ASLAM_TRSM_KEEP(10)
                        ASLAM_TRSM_KEEP(11)
                        ASLAM_TRSM_KEEP(12)
                        ASLAM_TRSM_KEEP(13)
                        ASLAM_TRSM_KEEP(14)
                        ASLAM_TRSM_KEEP(15)
                default:
                        break; // the last block column is never a history block
#undef ASLAM_TRSM_KEEP
                }
                // next block column: fresh accumulators, its slice of G (consumed k + 1 blocks from now)
                const int kn = min(k + 1, klast);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                {
                        c[t] = (f4){0.f, 0.f, 0.f, 0.f};
                        g0[t] = gload(LB * kn + 16 * t);
                }
                pp.rotate();
        }
        asm volatile("; ASLAM_STRIP_LIVE_END" ::: "memory");
}

/// V = G L^-T.  grid (8 * ceil(B / 8) * NP / 64), 256 threads; wave w of a workgroup owns 16 rows of G.  In place: G -> V.
/// The 17 workgroups of a filter stream the same blocks of L: they are dealt to ONE XCD (workgroup i runs on XCD i % 8), next to each other
/// in its dispatch order, so that a block comes from HBM once and from that XCD's L2 sixteen times (PMC: 23.6 -> MB per filter and callback).
template <int NBMAX, int DIAG = 0>
__global__ __launch_bounds__(256, 1) void large_trsm_pipe(DevView d, LargeView<float> lv, int nfilters, const int *skipped)
{
        static_assert(NBMAX == 17, "trsm_sweep lists 17 block columns");
        __shared__ __attribute__((aligned(16))) float lds[3][LB * TRSM_LDT];
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
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(lv.G + (size_t)b * NP * NP, 0, NP * NP * 4, 0x00020000);
        const unsigned vg = (unsigned)(((LB * rb + 16 * wave + li) * NP + 4 * lg) * 4); // this lane's row of G
        const int a_off = li * TRSM_LDT + 4 * lg;
        asm volatile("" ::: "a0", "a255"); // the strip (see above)
        unsigned long long t0_ = 0, r0_ = 0;
        if constexpr (DIAG & 8)
        {
                t0_ = __builtin_amdgcn_s_memtime();
                r0_ = __builtin_amdgcn_s_memrealtime();
        }
        TrsmSeq seq(lv.S + (size_t)b * NP * NP, lv.Linv + (size_t)b * LARGE_NB_MAX * LB * LB, 0, nb, NP, tid);
        if constexpr (DIAG & 32)
                seq.pin = 1;
        __shared__ unsigned sync_ctr;
        if (tid == 0)
                sync_ctr = 0; // (the first barrier of the sweep publishes it)
        TrsmPipe pp = {lds[0], lds[1], lds[2], &sync_ctr, 0u, 0u};
        f4 c[4];
        trsm_sweep<DIAG, false>(c, rg, vg, nb, seq, seq, pp, a_off, tid, rg, 0u, 0u);
        if (pp.lost && lane == 0)
                atomicOr(&d.status[b], 16u); // ASLAM_ST_INTERNAL
        if constexpr (DIAG & 8)
        {
                if (tid == 0)
                {
                        const size_t wg = blockIdx.x;
                        lv.Y[2 * wg] = (double)(__builtin_amdgcn_s_memtime() - t0_);
                        lv.Y[2 * wg + 1] = (double)(__builtin_amdgcn_s_memrealtime() - r0_);
                }
        }
}
} // namespace aslam
