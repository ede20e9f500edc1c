// launch_cost.hip -- what a kernel launch costs on MI355X as a function of workgroup size and dynamic LDS (round 3: the large-state
// front end takes 330 us per launch by rocprofv3 but 72 us by its own s_memtime stamps).  hipcc --offload-arch=gfx950 -O3 -o launch_cost launch_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x)                                                                                                          \
        do                                                                                                             \
        {                                                                                                              \
                hipError_t e_ = (x);                                                                                   \
                if (e_ != hipSuccess)                                                                                  \
                {                                                                                                      \
                        std::printf("%s: %s\n", #x, hipGetErrorString(e_));                                          \
                        return 1;                                                                                      \
                }                                                                                                      \
        } while (0)

template <int WG> __global__ __launch_bounds__(WG) void touch(float *out, int spin)
{
        extern __shared__ float smem[];
        smem[threadIdx.x] = (float)threadIdx.x;
        __syncthreads();
        float v = smem[(threadIdx.x + 1) % WG];
        for (int i = 0; i < spin; ++i)
                v = v * 1.0001f + 0.5f;
        if (v == 12345.f)
                out[blockIdx.x] = v;
}

template <int WG> int run(const char *what, int grid, size_t lds, int spin, float *d)
{
        auto k = touch<WG>;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 20; ++w)
                hipLaunchKernelGGL(k, dim3(grid), dim3(WG), lds, 0, d, spin);
        CK(hipDeviceSynchronize());
        const int N = 200;
        CK(hipEventRecord(e0, 0));
        for (int w = 0; w < N; ++w)
                hipLaunchKernelGGL(k, dim3(grid), dim3(WG), lds, 0, d, spin);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("  %-40s grid %4d wg %4d lds %7zu spin %6d : %8.2f us per launch (back to back on one stream)\n", what, grid, WG, lds, spin, ms * 1e3 / N);
        return 0;
}

int main()
{
        float *d = nullptr;
        CK(hipMalloc(&d, 1 << 20));
        for (int grid : {8, 256})
        {
                run<256>("256 threads, no LDS", grid, 1024, 0, d);
                run<256>("256 threads, 60 KB", grid, 60 * 1024, 0, d);
                run<256>("256 threads, 122 KB", grid, 122352, 0, d);
                run<768>("768 threads, 4 KB", grid, 4096, 0, d);
                run<768>("768 threads, 60 KB", grid, 60 * 1024, 0, d);
                run<768>("768 threads, 122 KB", grid, 122352, 0, d);
                run<768>("768 threads, 150 KB", grid, 150 * 1024, 0, d);
                run<1024>("1024 threads, 122 KB", grid, 122352, 0, d);
                run<768>("768 threads, 122 KB, 20k-iteration body", grid, 122352, 20000, d);
                run<768>("768 threads, 4 KB, 20k-iteration body", grid, 4096, 20000, d);
        }
        return 0;
}
