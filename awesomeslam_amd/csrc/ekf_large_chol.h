// ekf_large_chol.h -- S = L L^T for the large-state EKF in binary32 (the factor behind K = P H^T S^-1, ekf.cpp:301), one workgroup
// per filter, the whole factorisation in ONE launch.
//
// Row block I of L obeys the very recurrence large_trsm_pipe runs on the rows of G:
//      L(I, k) = ( S(I, k) - sum_{j<k} L(I, j) L(k, j)^T ) Linv_k^T            k < I
//      L(I, I) L(I, I)^T = S(I, I) - sum_{j<I} L(I, j) L(I, j)^T               (a 64x64 Cholesky)
// so the factorisation is 17 such strip sweeps, block row after block row, each needing all earlier block rows complete.  The
// round-2 chain before this kernel spent 33 launches on it (17 x {diagonal block, left-looking panel}): 1.6 ms per 128 filters, with
// grids that shrink to one workgroup per filter -- the serial spine of every stream group.  Here a filter stays on one CU: no
// inter-workgroup dependency exists, the filters of a batch are the parallelism (256 filters = one workgroup on every CU), and the
// 64x64 diagonal factorisations run in the same workgroup between two sweeps.  Every block of L is re-read once per later block row
// (13 MB per filter from HBM, against 38 MB for the old panel), at the rate large_trsm_pipe already streams them.
//
// The diagonal block: C = S(I,I) - sum (binary32, from the MFMA accumulators) goes to LDS as binary64 16x16 tiles; four waves factor it
// (16x16 diagonal tiles: one wave, in binary32 like the data; panel / trailing updates: fp64 MFMA on the tiles), build L(I,I)^-1 tile
// by tile (Linv_ij = -Linv_ii sum_k L_ik Linv_kj), and write both back in binary32.
#pragma once

namespace aslam
{
namespace chol64
{
// LDS tile map (binary64 tiles of TSZ doubles, rows of TLD): L (lower 4x4: 10 tiles) | inverses of the diagonal tiles (4) |
// R(i,j) = Linv(i,j)^T (lower: 10) | one transposed product per wave (4)
constexpr int TILES = 28;
__device__ __forceinline__ int lower(int i, int j) { return i * (i + 1) / 2 + j; }
__device__ __forceinline__ double *Lt(double *t, int i, int j) { return t + lower(i, j) * TSZ; }
__device__ __forceinline__ double *Ti(double *t, int k) { return t + (10 + k) * TSZ; }
__device__ __forceinline__ double *Rt(double *t, int i, int j) { return t + (14 + lower(i, j)) * TSZ; }
__device__ __forceinline__ double *Wt(double *t, int w) { return t + (24 + w) * TSZ; }

/// acc += X Y^T for two row-major 16x16 tiles; acc[r] of lane (li, lg) = element (lg + 4 r, li)
__device__ __forceinline__ d4 xyt(const double *X, const double *Y, d4 acc, double sign, int li, int lg)
{
        double xa[4], yb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
                xa[s] = sign * X[li * TLD + lg + 4 * s];
                yb[s] = Y[li * TLD + lg + 4 * s];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
                acc = mfma_f64(xa[s], yb[s], acc);
        return acc;
}

/// Cholesky factor of the 16x16 diagonal tile `T` (binary64 in LDS, lower triangle valid) in place and the inverse of the factor -> `Ti`,
/// computed in BINARY32: the data of this path is binary32 (S, L and every product), and the 16 dependent pivot steps are the serial spine of
/// the diagonal blocks.  One wave, all 64 lanes active; lane i < 16 owns row i (lanes 16-63 mirror), the inverse rides on the same multipliers
/// (outer-product form, as factor_diag_tile_fast), and like there (round 4) the multiplier L(c, j) reaches the lanes through the DPP operand of
/// the fmac (row_newbcast:c) instead of a v_readlane into an SGPR per update.  Returns false on a non-positive pivot.
#define ASLAM_DPP_FMAC32(acc, bsrc, other, c)                                                                          \
        asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:" #c " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(other))
template <int C> __device__ __forceinline__ void factor_diag_update_f32(float (&a)[16], float (&s)[16], const float &lij, const float &xj)
{
        if constexpr (C < 16)
        {
#define ASLAM_CASE(cc)                                                                                                 \
        if constexpr (C == cc)                                                                                         \
        {                                                                                                              \
                ASLAM_DPP_FMAC32(a[cc], lij, lij, cc);                                                                 \
                ASLAM_DPP_FMAC32(s[cc], lij, xj, cc);                                                                  \
        }
                ASLAM_CASE(1) ASLAM_CASE(2) ASLAM_CASE(3) ASLAM_CASE(4) ASLAM_CASE(5) ASLAM_CASE(6) ASLAM_CASE(7) ASLAM_CASE(8) ASLAM_CASE(9)
                ASLAM_CASE(10) ASLAM_CASE(11) ASLAM_CASE(12) ASLAM_CASE(13) ASLAM_CASE(14) ASLAM_CASE(15)
#undef ASLAM_CASE
        }
}
template <int J> __device__ __forceinline__ void factor_diag_column_f32(float (&a)[16], float (&s)[16], bool &ok)
{
        const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[J]), J));
        ok = ok && (d > 0.f);
        const float y = __builtin_amdgcn_rsqf(d);
        const float inv = y * fmaf(-0.5f * d * y, y, 1.5f); // one Newton step on the hardware seed
        float lij = a[J] * inv;                             // L(i, J) for i >= J
        float xj = s[J] * inv;                              // (L^-1)(J, lane)
        a[J] = lij;
        s[J] = xj;
        asm volatile("s_nop 1" : "+v"(lij), "+v"(xj)); // a VALU result read through DPP: two wait states
        factor_diag_update_f32<J + 1>(a, s, lij, xj);
        factor_diag_update_f32<J + 2>(a, s, lij, xj);
        factor_diag_update_f32<J + 3>(a, s, lij, xj);
        factor_diag_update_f32<J + 4>(a, s, lij, xj);
        factor_diag_update_f32<J + 5>(a, s, lij, xj);
        factor_diag_update_f32<J + 6>(a, s, lij, xj);
        factor_diag_update_f32<J + 7>(a, s, lij, xj);
        factor_diag_update_f32<J + 8>(a, s, lij, xj);
        factor_diag_update_f32<J + 9>(a, s, lij, xj);
        factor_diag_update_f32<J + 10>(a, s, lij, xj);
        factor_diag_update_f32<J + 11>(a, s, lij, xj);
        factor_diag_update_f32<J + 12>(a, s, lij, xj);
        factor_diag_update_f32<J + 13>(a, s, lij, xj);
        factor_diag_update_f32<J + 14>(a, s, lij, xj);
        factor_diag_update_f32<J + 15>(a, s, lij, xj);
        if constexpr (J + 1 < 16)
                factor_diag_column_f32<J + 1>(a, s, ok);
}
__device__ __forceinline__ bool factor_diag_tile_f32(double *T, double *Ti, int lane)
{
        float a[16], s[16];
        const int row = lane & 15;
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
                a[c] = (float)T[row * TLD + c];
                s[c] = (row == c) ? 1.f : 0.f;
        }
        bool ok = true;
        factor_diag_column_f32<0>(a, s, ok);
        // one unmasked store per lane and entry: lanes 16-63 hold copies of rows 0-15 and write the same values to the same addresses
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
                T[row * TLD + c] = (c <= row) ? (double)a[c] : 0.0;
                Ti[c * TLD + row] = (double)s[c]; // Linv(c, row): zero above the diagonal by construction
        }
        return ok;
}
#undef ASLAM_DPP_FMAC32

/// In: the lower 4x4 tiles of a symmetric positive definite 64x64 matrix in Lt.  Out: its Cholesky factor in Lt (zeros above the
/// diagonal of the diagonal tiles) and R = (L^-1)^T tiles.  256 threads; starts and ends with a barrier.  Returns false (on wave 0)
/// on a non-positive pivot.
template <bool F32_DIAG> __device__ __forceinline__ bool factor_and_invert(double *t, int tid)
{
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 15, lg = lane >> 4;
        bool ok = true;
        __syncthreads();
#pragma unroll 1
        for (int kb = 0; kb < 4; ++kb)
        {
                if (wave == 0)
                        ok = (F32_DIAG ? factor_diag_tile_f32(Lt(t, kb, kb), Ti(t, kb), lane) : factor_diag_tile_fast(Lt(t, kb, kb), Ti(t, kb), lane)) && ok;
                __syncthreads();
                // panel: L(ib, kb) = S(ib, kb) Linv_kb^T, one tile per wave; wave 3 (never has one) transposes Linv_kb into R(kb, kb)
                const int ib = kb + 1 + wave;
                if (ib < 4)
                {
                        double *S = Lt(t, ib, kb);
                        const d4 p = xyt(S, Ti(t, kb), (d4){0.0, 0.0, 0.0, 0.0}, 1.0, li, lg);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                S[(lg + 4 * r) * TLD + li] = p[r];
                }
                if (wave == 3)
                {
                        const double *src = Ti(t, kb);
                        double *dst = Rt(t, kb, kb);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                        {
                                const int e = lane + 64 * q;
                                dst[(e >> 4) * TLD + (e & 15)] = src[(e & 15) * TLD + (e >> 4)];
                        }
                }
                __syncthreads();
                // trailing update: S(ib, jb) -= L(ib, kb) L(jb, kb)^T for kb < jb <= ib: up to six tiles over four waves
                const int m = 3 - kb, cnt = m * (m + 1) / 2;
                for (int q = wave; q < cnt; q += 4)
                {
                        const int i = (q >= 3) ? 2 : (q >= 1) ? 1 : 0, j = q - i * (i + 1) / 2;
                        double *S = Lt(t, kb + 1 + i, kb + 1 + j);
                        d4 acc;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                acc[r] = S[(lg + 4 * r) * TLD + li];
                        acc = xyt(Lt(t, kb + 1 + i, kb), Lt(t, kb + 1 + j, kb), acc, -1.0, li, lg);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                S[(lg + 4 * r) * TLD + li] = acc[r];
                }
                __syncthreads();
        }
        // the off-diagonal tiles of the inverse, diagonal by diagonal:  W = sum_{k=j}^{i-1} L(i,k) Linv(k,j),  Linv(i,j) = -Linv(i,i) W;
        // stored transposed: R(i,j) = Linv(i,j)^T = -W^T Linv(i,i)^T
#pragma unroll 1
        for (int dgl = 1; dgl < 4; ++dgl)
        {
                const int j = wave, i = j + dgl;
                if (i < 4)
                {
                        d4 w = {0.0, 0.0, 0.0, 0.0};
                        for (int k = j; k < i; ++k)
                                w = xyt(Lt(t, i, k), Rt(t, k, j), w, 1.0, li, lg); // L(i,k) (R(k,j))^T = L(i,k) Linv(k,j)
                        double *W = Wt(t, wave);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                W[li * TLD + lg + 4 * r] = w[r]; // transposed
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier(); // the same wave reads W back through LDS
                        const d4 x = xyt(W, Ti(t, i), (d4){0.0, 0.0, 0.0, 0.0}, -1.0, li, lg); // -W^T Linv(i,i)^T
                        double *R = Rt(t, i, j);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                R[(lg + 4 * r) * TLD + li] = x[r];
                }
                __syncthreads();
        }
        return ok;
}

/// L (zeros above the diagonal) -> the 64x64 block at `Sdiag` (row stride NP), Linv = R^T -> `Linv_out` [64][64]: thread = row r, 16-column
/// segment q.  T = float or double.
/// `Liq` (binary32 only, may be null): the inverse ALSO as three bf16 planes with permuted columns, at the diagonal block's place in the planes of L
/// (LPlanes, ekf_large.h: `Liq` = plane 0 of that block, row stride NP, plane stride NP * NP): the thread's
/// columns 16 q + 4 g .. + 3 go to positions 32 (q >> 1) + 8 g + 4 (q & 1) .. + 3.
template <typename T> __device__ __forceinline__ void store_block(const double *t, T *Sdiag, int NP, T *Linv_out, int tid, unsigned short *Liq = nullptr)
{
        const int r = tid >> 2, q = tid & 3, ti = r >> 4, a = r & 15;
        T *ls = Sdiag + (size_t)r * NP + 16 * q;
        T *io = Linv_out + r * LB + 16 * q;
        const double *Ltile = Lt(const_cast<double *>(t), ti, min(q, ti)) + a * TLD;
        const double *Rtile = Rt(const_cast<double *>(t), ti, min(q, ti)) + a;
        T inv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
        {
                ls[e] = (q <= ti) ? (T)Ltile[e] : (T)0;
                inv[e] = (q <= ti) ? (T)Rtile[e * TLD] : (T)0; // Linv(a, b) = R(b, a)
                io[e] = inv[e];
        }
        if constexpr (sizeof(T) == 4)
        {
                if (Liq)
                {
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                        {
                                u2x h, m, l;
                                split_bf16x3((f4){(float)inv[4 * g], (float)inv[4 * g + 1], (float)inv[4 * g + 2], (float)inv[4 * g + 3]}, h, m, l);
                                unsigned short *dst = Liq + (size_t)r * NP + 32 * (q >> 1) + 8 * g + 4 * (q & 1);
                                *reinterpret_cast<u2x *>(dst) = h;
                                *reinterpret_cast<u2x *>(dst + (size_t)NP * NP) = m;
                                *reinterpret_cast<u2x *>(dst + 2 * (size_t)NP * NP) = l;
                        }
                }
        }
}
} // namespace chol64

/// Diagonal block k of the multi-workgroup chain (few filters per launch: batch 1, configs[4]): Cholesky factor and its inverse.
/// grid (B), 256 threads.  Round 1 did this with one wave holding the block in registers (lane = row, `v_readlane` broadcasts):
/// 41 us per launch at batch 1, 17 launches = 42 % of a callback; the four-wave tile form of large_chol_resident takes its place.
template <typename T> __global__ __launch_bounds__(256) void large_potrf_inv_tiles(DevView d, LargeView<T> lv, int k, const int *skipped)
{
        __shared__ __attribute__((aligned(16))) double tiles[chol64::TILES * TSZ];
        const int b = blockIdx.x;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        if (k >= large_blocks(n))
                return;
        T *S = lv.S + (size_t)b * NP * NP + (size_t)k * LB * NP + k * LB;
        T *Li = lv.Linv + ((size_t)b * LARGE_NB_MAX + k) * LB * LB;
        const int tid = threadIdx.x;
        {
                const int r = tid >> 2, q = tid & 3, ti = r >> 4, a = r & 15;
                if (q <= ti)
                {
                        const T *src = S + (size_t)r * NP + 16 * q;
                        double *dst = chol64::Lt(tiles, ti, q) + a * TLD;
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                                dst[e] = (double)src[e];
                }
        }
        const bool ok = chol64::factor_and_invert<sizeof(T) == 4>(tiles, tid); // binary32 mode: binary32 diagonal tiles
        chol64::store_block(tiles, S, NP, Li, tid);
        if (!ok && tid == 0)
                atomicOr(&d.status[b], 4u); // ASLAM_ST_NOT_PD
}

/// Few filters per launch (batch < 32: batch 1 is configs[4]), binary32: ONE launch per block column k does everything that depends on the
/// factor of diagonal block k -- RIGHT-looking, so that no launch is deeper than three 64x64x64 products and a 64x64 factorisation, and a
/// filter's work spreads over as many workgroups as it has tiles (the chip is empty at this batch: redundant work is free, launches and serial
/// depth are what cost).  With Linv_k = L(k,k)^-1 (from the previous launch) a workgroup takes one 64x64 tile:
///   S tile (i, j), k < j <= i:   X_i = S(i,k) Linv_k^T,  X_j = S(j,k) Linv_k^T  (both recomputed by every workgroup that needs them: L(i,k) is
///                                never stored -- nothing reads it afterwards),  S(i,j) -= X_i X_j^T.
///                                Tile (k+1, k+1) is then the next diagonal block, complete: the same workgroup factors and inverts it
///                                (chol64::factor_and_invert, as large_potrf_inv_tiles) -- the 17 diagonal launches of the left-looking chain are gone;
///   G tile (g, j), j > k:        G(g,j) -= (G(g,k) Linv_k^T) X_j^T  -- the rows of G are solved by the same launches (no TRSM launch);
///   V panel g:                   V(g,k) = G(g,k) Linv_k^T -> lv.Vw: final.  (Not over G(g,k): the G tiles of this launch read it.)
/// Measured at batch 1, n = 1027 (profiles/r03_batch1_breakdown.txt): the left-looking chain spent 836 us per callback in 33 launches of the
/// Cholesky and 162 in the TRSM.  grid (tiles(k), 1, B) with tiles(k) = M (M + 1) / 2 + NB M + NB, M = NB - k - 1; 256 threads.
template <int UNUSED = 0> // (a template: the header is included by both translation units of the library)
__global__ __launch_bounds__(256) void large_right_step(DevView d, LargeView<float> lv, int k, const int *skipped)
{
        typedef Mfma<float> MM;
        constexpr int LD = LB + 8; // 18 sixteen-byte slots per row: conflict-free ds_read_b128 operand reads (as large_update_panel)
        constexpr int PROD_BYTES = 3 * LB * LD * (int)sizeof(float), TILE_BYTES = chol64::TILES * TSZ * (int)sizeof(double);
        __shared__ __attribute__((aligned(16))) unsigned char raw[PROD_BYTES > TILE_BYTES ? PROD_BYTES : TILE_BYTES];
        float(*As)[LD] = reinterpret_cast<float(*)[LD]>(raw);
        float(*Bs)[LD] = As + LB;
        float(*Is)[LD] = Bs + LB;
        double *tiles = reinterpret_cast<double *>(raw); // the diagonal factorisation, once the products are done with the operands
        const int b = blockIdx.z;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n);
        if (k >= nb)
                return;
        const int m = nb - k - 1, nS = m * (m + 1) / 2, nG = nb * m;
        const int idx = blockIdx.x;
        int type, ri, cj = k;
        if (idx < nS)
        {
                int ii = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
                while ((ii + 1) * (ii + 2) / 2 <= idx)
                        ++ii;
                while (ii * (ii + 1) / 2 > idx)
                        --ii;
                type = 0, ri = k + 1 + ii, cj = k + 1 + (idx - ii * (ii + 1) / 2);
        }
        else if (idx < nS + nG)
                type = 1, ri = (idx - nS) / m, cj = k + 1 + (idx - nS) % m;
        else if (idx < nS + nG + nb)
                type = 2, ri = idx - nS - nG;
        else
                return;
        float *S = lv.S + (size_t)b * NP * NP, *G = lv.G + (size_t)b * NP * NP;
        const float *Ablk = (type == 0 ? S : G) + (size_t)(LB * ri) * NP + LB * k;
        const float *Bblk = S + (size_t)(LB * cj) * NP + LB * k;
        const float *Li = lv.Linv + ((size_t)b * LARGE_NB_MAX + k) * LB * LB;
        const bool same = (type == 0 && ri == cj), two = (type != 2 && !same);
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
                const int e = tid + 256 * q, r = e >> 4, c4 = (e & 15) * 4;
                *reinterpret_cast<f4 *>(&As[r][c4]) = *reinterpret_cast<const f4 *>(Ablk + (size_t)r * NP + c4);
                *reinterpret_cast<f4 *>(&Is[r][c4]) = *reinterpret_cast<const f4 *>(Li + r * LB + c4);
                if (two)
                        *reinterpret_cast<f4 *>(&Bs[r][c4]) = *reinterpret_cast<const f4 *>(Bblk + (size_t)r * NP + c4);
        }
        __syncthreads();
        // acc[v] = rows 16 wave .. + 15 of A times rows 16 v .. + 15 of B, transposed: sum_t A(r, t) B(c, t); one 16-byte LDS read per operand row
        // feeds four MFMA steps (lane (li, lg) holds t = 16 c + 4 lg + r for step r: a permuted walk over the contraction index, the same for both)
        auto prod = [&](const float(*A)[LD], const float(*Bm)[LD], f4(&acc)[4]) {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                        acc[v] = MM::zero();
#pragma unroll
                for (int c = 0; c < LB / 16; ++c)
                {
                        const f4 av = *reinterpret_cast<const f4 *>(&A[16 * wave + li][16 * c + 4 * lg]);
                        f4 bv[4];
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                                bv[v] = *reinterpret_cast<const f4 *>(&Bm[16 * v + li][16 * c + 4 * lg]);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                        acc[v] = MM::mma(av[r], bv[v][r], acc[v]);
                }
        };
        f4 xa[4], xb[4];
        prod(As, Is, xa);
        if (type == 2)
        {
                float *V = lv.Vw + (size_t)b * NP * NP + (size_t)(LB * ri + 16 * wave) * NP + LB * k;
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                V[(size_t)MM::row(lane, r) * NP + 16 * v + li] = xa[v][r];
                return;
        }
        if (two)
                prod(Bs, Is, xb);
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                        As[16 * wave + MM::row(lane, r)][16 * v + li] = xa[v][r];
                        if (two)
                                Bs[16 * wave + MM::row(lane, r)][16 * v + li] = xb[v][r];
                }
        __syncthreads();
        f4 c[4];
        prod(As, two ? Bs : As, c);
        float *C = (type == 0 ? S : G) + (size_t)(LB * ri + 16 * wave) * NP + LB * cj;
        if (same && ri == k + 1)
        {
                // the next diagonal block is complete with this update: factor it here
                __syncthreads(); // (the tiles alias the operands)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                        if (v <= wave)
                        {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                        chol64::Lt(tiles, wave, v)[MM::row(lane, r) * TLD + li] = (double)(C[(size_t)MM::row(lane, r) * NP + 16 * v + li] - c[v][r]);
                        }
                const bool ok = chol64::factor_and_invert<true>(tiles, tid);
                chol64::store_block(tiles, S + (size_t)(LB * ri) * NP + LB * ri, NP, lv.Linv + ((size_t)b * LARGE_NB_MAX + ri) * LB * LB, tid);
                if (!ok && tid == 0)
                        atomicOr(&d.status[b], 4u); // ASLAM_ST_NOT_PD
                return;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                        C[(size_t)MM::row(lane, r) * NP + 16 * v + li] -= c[v][r];
}

/// grid (B), 256 threads: the Cholesky factor of S (lower block triangle, in place) and the inverses of its diagonal blocks (lv.Linv)
/// for filter blockIdx.x.  Status bit 4 (ASLAM_ST_NOT_PD) on a non-positive pivot.
template <int NBMAX, int STAMP = 0>
__global__ __launch_bounds__(256, 1) void large_chol_resident(DevView d, LargeView<float> lv, const int *skipped)
{
        static_assert(NBMAX == 17, "trsm_sweep lists 17 block columns");
        constexpr int PIPE_BYTES = 3 * LB * TRSM_LDT * (int)sizeof(float), TILE_BYTES = chol64::TILES * TSZ * (int)sizeof(double);
        // the block pipeline of a sweep and the tiles of the diagonal factorisation never live at the same time
        __shared__ __attribute__((aligned(16))) unsigned char raw[PIPE_BYTES > TILE_BYTES ? PIPE_BYTES : TILE_BYTES];
        float *pipe = reinterpret_cast<float *>(raw);
        double *tiles = reinterpret_cast<double *>(raw);
        const int b = blockIdx.x;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n);
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int a_off = li * TRSM_LDT + 4 * lg;
        float *Sb = lv.S + (size_t)b * NP * NP;
        float *Linv = lv.Linv + (size_t)b * LARGE_NB_MAX * LB * LB;
        asm volatile("" ::: "a0", "a255"); // the strip (ekf_large_trsm.h)
        __shared__ unsigned sync_ctr; // the sweeps' wave synchronisation (TrsmPipe): counts on through all block rows
        if (tid == 0)
                sync_ctr = 0;
        unsigned nsig = 0, lost = 0;
        const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(Sb, 0, NP * NP * 4, 0x00020000);
        // bf16 planes of L and of the inverses for large_trsm_bf16 (lv.Lpl != nullptr)
        const LPlanes lpl = {lv.Lpl};
        const unsigned qplane = lv.Lpl ? (unsigned)(NP * NP * 2) : 0u;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(lv.Lpl ? lpl.Lq(b, NP) : reinterpret_cast<unsigned short *>(Sb), 0, 3 * NP * NP * 2, 0x00020000);
        const unsigned vq0 = (unsigned)(((16 * wave + li) * NP + 8 * lg) * 2); // this lane's row inside a block row, + 8 lg elements
        bool ok = true;
        // STAMP (diagnostic build, tools/ubench/trsm_bench.hip): shader cycles of workgroup 0 by phase -> lv.Y[0 .. 3]: sweeps, conversion, diagonal
        // factorisation, stores + drain
        unsigned long long tph[4] = {0, 0, 0, 0}, tm_ = STAMP ? __builtin_amdgcn_s_memtime() : 0, tn_ = 0;
#define ASLAM_PH(i)                                                                                                    \
        if constexpr (STAMP)                                                                                           \
        {                                                                                                              \
                tn_ = __builtin_amdgcn_s_memtime();                                                                    \
                tph[i] += tn_ - tm_;                                                                                   \
                tm_ = tn_;                                                                                             \
        }
#pragma unroll 1
        for (int I = 0; I < nb; ++I)
        {
                const unsigned vs = (unsigned)(((LB * I + 16 * wave + li) * NP + 4 * lg) * 4); // this lane's row of block row I (+ 4 lg floats), bytes
                TrsmSeq seq(Sb, Linv, 0, I, NP, tid);
                const TrsmSeq seq_diag(Sb, Linv, I, I + 1, NP, tid); // the history blocks of the diagonal block: L(I, 0 .. I-1), this sweep's own output
                TrsmPipe pp = {pipe, pipe + LB * TRSM_LDT, pipe + 2 * LB * TRSM_LDT, &sync_ctr, nsig, 0u};
                f4 c[4];
                trsm_sweep<0, true>(c, rsb, vs, I, seq, seq_diag, pp, a_off, tid, rq, vq0 + (unsigned)(LB * I * NP * 2), qplane);
                nsig = pp.nsig;
                lost |= pp.lost;
                __syncthreads(); // every wave is done with the pipeline buffers: the tiles take their place
                ASLAM_PH(0)
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
                ASLAM_PH(1)
                ok = chol64::factor_and_invert<true>(tiles, tid) && ok;
                ASLAM_PH(2)
                chol64::store_block(tiles, Sb + ((size_t)LB * I) * NP + (size_t)LB * I, NP, Linv + (size_t)I * LB * LB, tid, lv.Lpl ? lpl.block(b, NP, I, I) : nullptr);
                // the next block row reads them back through the block pipeline
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                ASLAM_PH(3)
        }
#undef ASLAM_PH
        if constexpr (STAMP)
                if (tid == 0 && b == 0)
                        for (int i = 0; i < 4; ++i)
                                lv.Y[i] = (double)tph[i];
        if (!ok && tid == 0)
                atomicOr(&d.status[b], 4u); // ASLAM_ST_NOT_PD
        if (lost && lane == 0)
                atomicOr(&d.status[b], 16u); // ASLAM_ST_INTERNAL: a wave gave up waiting for the others (TrsmPipe::wait)
}
} // namespace aslam
