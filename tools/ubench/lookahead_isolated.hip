// isolates cholesky_lookahead<NT> / factor_diag_tile_fast: one workgroup, tiles in LDS, opaque or launch-time tid
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iawesomeslam_amd/csrc -Iinclude -o /tmp/lookahead tools/ubench/lookahead_isolated.hip
#include "small_common.h"
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
using namespace aslam;
template <int NT, bool OPAQUE> __global__ __launch_bounds__(SMALL_WG) void k(const double *in, double *outL, double *outI, int nt, int reps)
{
        __shared__ double Lt[NT * (NT + 1) / 2 * TSZ];
        __shared__ double Dinv[NT * TSZ];
        __shared__ uint32_t status;
        const int tid_launch = threadIdx.x;
        for (int rep = 0; rep < reps; ++rep)
        {
                int tid = tid_launch;
                if (OPAQUE)
                        asm volatile("" : "+v"(tid));
                const int ntl = nt * (nt + 1) / 2;
                for (int idx = tid; idx < ntl * 256; idx += SMALL_WG)
                        Lt[(idx >> 8) * TSZ + ((idx & 255) >> 4) * TLD + (idx & 15)] = in[idx];
                __syncthreads();
                cholesky_lookahead<NT>(Lt, Dinv, nt, tid, &status);
                for (int idx = tid; idx < ntl * 256; idx += SMALL_WG)
                        outL[idx] = Lt[(idx >> 8) * TSZ + ((idx & 255) >> 4) * TLD + (idx & 15)];
                for (int idx = tid; idx < nt * 256; idx += SMALL_WG)
                        outI[idx] = Dinv[(idx >> 8) * TSZ + ((idx & 255) >> 4) * TLD + (idx & 15)];
                __syncthreads();
        }
}
template <int NT, bool OPQ> double run(int nt, unsigned seed)
{
        const int n = 16 * nt, ntl = nt * (nt + 1) / 2;
        std::mt19937 g(seed); std::normal_distribution<double> N(0, 1);
        std::vector<double> A(n * n), S(n * n, 0.0);
        for (auto &v : A) v = N(g) * 0.05;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = (i == j) ? 0.2 : 0.0; for (int k = 0; k < n; ++k) s += A[i * n + k] * A[j * n + k]; S[i * n + j] = s; }
        std::vector<double> tiles(ntl * 256);
        for (int ib = 0; ib < nt; ++ib) for (int jb = 0; jb <= ib; ++jb) for (int e = 0; e < 256; ++e)
                tiles[(ib * (ib + 1) / 2 + jb) * 256 + e] = S[(16 * ib + (e >> 4)) * n + 16 * jb + (e & 15)];
        double *din, *dL, *dI; hipMalloc(&din, tiles.size() * 8); hipMalloc(&dL, tiles.size() * 8); hipMalloc(&dI, nt * 256 * 8);
        hipMemcpy(din, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<NT, OPQ>), dim3(1), dim3(SMALL_WG), 0, 0, din, dL, dI, nt, 3);
        std::vector<double> L(tiles.size()); hipMemcpy(L.data(), dL, L.size() * 8, hipMemcpyDeviceToHost);
        // reference Cholesky
        std::vector<double> R(n * n, 0.0);
        for (int j = 0; j < n; ++j) { double d = S[j * n + j]; for (int k = 0; k < j; ++k) d -= R[j * n + k] * R[j * n + k]; d = std::sqrt(d); R[j * n + j] = d;
                for (int i = j + 1; i < n; ++i) { double s = S[i * n + j]; for (int k = 0; k < j; ++k) s -= R[i * n + k] * R[j * n + k]; R[i * n + j] = s / d; } }
        double err = 0, mx = 0;
        for (int ib = 0; ib < nt; ++ib) for (int jb = 0; jb <= ib; ++jb) for (int e = 0; e < 256; ++e) {
                const int i = 16 * ib + (e >> 4), j = 16 * jb + (e & 15); if (j > i) continue;
                err = std::fmax(err, std::fabs(L[(ib * (ib + 1) / 2 + jb) * 256 + e] - R[i * n + j])); mx = std::fmax(mx, std::fabs(R[i * n + j])); }
        hipFree(din); hipFree(dL); hipFree(dI);
        return err / mx;
}
int main()
{
        for (unsigned seed = 1; seed <= 3; ++seed)
                printf("seed %u: NT=2 nt=1 launch %.2e opaque %.2e | NT=2 nt=2 launch %.2e opaque %.2e | NT=5 nt=3 launch %.2e opaque %.2e | NT=9 nt=9 launch %.2e opaque %.2e\n", seed,
                       run<2, false>(1, seed), run<2, true>(1, seed), run<2, false>(2, seed), run<2, true>(2, seed), run<5, false>(3, seed), run<5, true>(3, seed),
                       run<9, false>(9, seed), run<9, true>(9, seed));
        return 0;
}
