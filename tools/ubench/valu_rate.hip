// Issue rate of the VALU instructions the bf16 split of ekf_large_trsm16.h is made of (one wave, 512 independent instructions between two
// s_memtime reads), and whether the residual of the split can be formed by v_dot2_f32_bf16 (x - h as h * (-1) + 0 * h' + x: one instruction per
// element instead of a shift/mask + subtract) EXACTLY: the three-piece split must stay bit-identical to split_bf16x3's.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate && tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP> __global__ void rate(unsigned long long *out, float *sink)
{
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        float b0 = 1.5f, b1 = 2.5f;
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)");
        for (int it = 0; it < 8; ++it)
        {
                if constexpr (OP == 0)
                        asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
                if constexpr (OP == 1)
                        asm volatile(REP8("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8\n")
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
                if constexpr (OP == 2)
                        asm volatile(REP8("v_dot2_f32_bf16 %0, %0, %8, %0\n v_dot2_f32_bf16 %1, %1, %8, %1\n v_dot2_f32_bf16 %2, %2, %8, %2\n v_dot2_f32_bf16 %3, %3, %8, %3\n v_dot2_f32_bf16 %4, %4, %8, %4\n v_dot2_f32_bf16 %5, %5, %8, %5\n v_dot2_f32_bf16 %6, %6, %8, %6\n v_dot2_f32_bf16 %7, %7, %8, %7\n")
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
                if constexpr (OP == 3)
                        asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                                     : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&b0));
                if constexpr (OP == 4)
                        asm volatile(REP8("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7\n")
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
                if constexpr (OP == 5)
                        asm volatile(REP8("v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 16, %1\n v_and_b32 %2, %2, %8\n v_lshlrev_b32 %3, 16, %3\n v_and_b32 %4, %4, %8\n v_lshlrev_b32 %5, 16, %5\n v_and_b32 %6, %6, %8\n v_lshlrev_b32 %7, 16, %7\n")
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));
        }
        asm volatile("s_nop 7\n s_nop 7");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0)
                out[OP] = t1 - t0;
        sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b1;
}

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

// reference split (split_bf16x3 of ekf_large.h) and the dot2 form, on pairs of floats
__global__ void split_check(const float *x, int npairs, unsigned *bad, unsigned *first, unsigned *bad_mid)
{
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= npairs)
                return;
        const f2 v = {x[2 * i], x[2 * i + 1]};
        const bf2 h = __builtin_convertvector(v, bf2);
        const f2 r1 = v - __builtin_convertvector(h, f2);
        const bf2 m = __builtin_convertvector(r1, bf2);
        const f2 r2 = r1 - __builtin_convertvector(m, f2);
        const bf2 l = __builtin_convertvector(r2, bf2);
        const unsigned H = __builtin_bit_cast(unsigned, h), M = __builtin_bit_cast(unsigned, m), Lq = __builtin_bit_cast(unsigned, l);
        // dot2 form: K0 = {-1.0 (0xbf80) in the low half}, K1 = {-1.0 in the high half}
        unsigned h2, m2, l2;
        float x0 = v[0], x1 = v[1];
        const unsigned K0 = 0x0000bf80u, K1 = 0xbf800000u;
        asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n"
                     "v_dot2_f32_bf16 %3, %0, %5, %3\n"
                     "v_dot2_f32_bf16 %4, %0, %6, %4\n"
                     "s_nop 3\n" // a DOT result needs three wait states before another VALU opcode reads it (no interlock: without them the
                                 // conversion below reads the OLD x -- measured; in the regions the four interleaved chains provide the distance)
                     "v_cvt_pk_bf16_f32 %1, %3, %4\n"
                     "v_dot2_f32_bf16 %3, %1, %5, %3\n"
                     "v_dot2_f32_bf16 %4, %1, %6, %4\n"
                     "s_nop 3\n"
                     "v_cvt_pk_bf16_f32 %2, %3, %4\n"
                     : "=&v"(h2), "=&v"(m2), "=&v"(l2), "+v"(x0), "+v"(x1)
                     : "s"(K0), "s"(K1));
        if (h2 != H || m2 != M || l2 != Lq)
        {
                if (fabsf(v[0]) < 1e30f && fabsf(v[1]) < 1e30f && (fabsf(v[0]) > 1e-30f || v[0] == 0.f) && (fabsf(v[1]) > 1e-30f || v[1] == 0.f))
                        atomicAdd(bad_mid, 1u);
                if (atomicAdd(bad, 1u) == 0)
                        first[0] = i, first[1] = H, first[2] = h2, first[3] = M, first[4] = m2, first[5] = Lq, first[6] = l2;
        }
}

int main()
{
        unsigned long long *out;
        float *sink;
        hipMalloc(&out, 64);
        hipMalloc(&sink, 256);
        hipMemset(out, 0, 64);
        rate<0><<<1, 64>>>(out, sink);
        rate<1><<<1, 64>>>(out, sink);
        rate<2><<<1, 64>>>(out, sink);
        rate<3><<<1, 64>>>(out, sink);
        rate<4><<<1, 64>>>(out, sink);
        rate<5><<<1, 64>>>(out, sink);
        unsigned long long h[8];
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        const char *names[] = {"v_add_f32", "v_cvt_pk_bf16_f32", "v_dot2_f32_bf16", "v_pk_add_f32", "v_accvgpr_read_b32", "v_and_b32 / v_lshlrev_b32"};
        for (int i = 0; i < 6; ++i)
                printf("%-28s %6.2f s_memtime ticks per instruction (512 instructions, one wave; 100 MHz ticks x 24 = shader cycles at 2.4 GHz)\n", names[i], (double)h[i] / 512.0);
        // exactness of the dot2 split
        const int np = 1 << 22;
        std::vector<float> x(2 * np);
        srand(7);
        for (int i = 0; i < 2 * np; ++i)
        {
                unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
                unsigned e = (u >> 23) & 0xff;
                if (i & 1)
                        u = (u & 0x807fffffu) | ((100u + e % 60u) << 23); // moderate exponents
                else if (e == 0xff)
                        u &= 0xff7fffffu & ~(1u << 30); // no Inf / NaN
                float f;
                memcpy(&f, &u, 4);
                x[i] = f;
        }
        x[0] = 0.f, x[1] = -0.f, x[2] = 1.0f, x[3] = 1.0f + 1.f / 256, x[4] = 1.0f + 3.f / 512, x[5] = 3.4e38f, x[6] = 1e-38f, x[7] = 1.17549435e-38f;
        float *dx;
        unsigned *bad, *first;
        hipMalloc(&dx, x.size() * 4);
        hipMalloc(&bad, 4);
        hipMalloc(&first, 32);
        hipMemset(first, 0, 32);
        hipMemset(bad, 0, 4);
        hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        split_check<<<(np + 255) / 256, 256>>>(dx, np, bad, first, first + 7);
        unsigned nb, f[8];
        hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost);
        hipMemcpy(f, first, 32, hipMemcpyDeviceToHost);
        printf("dot2 split against the shift/mask split: %u of %d pairs differ", nb, np);
        if (nb)
                printf(" (first: pair %u x = %g %g  h %08x / %08x  m %08x / %08x  l %08x / %08x)", f[0], x[2 * f[0]], x[2 * f[0] + 1], f[1], f[2], f[3], f[4], f[5], f[6]);
        printf("; %u of them with both |x| in [1e-30, 1e30] (the rest: a piece overflows to Inf and 0 x Inf poisons its neighbour, or denormal residuals are flushed)\n", f[7]);
        return 0;
}
