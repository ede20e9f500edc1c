// aslam_large16.hip -- the second translation unit of libaslam_core.so: the bf16-pipe Cholesky and TRSM of the large-state EKF
// (ekf_large_trsm16.h).  Compiled apart from aslam_core.hip because these two kernels need `-mllvm -amdgpu-mfma-vgpr-form`: with the AGPR
// strip reserved by an asm clobber hipcc would otherwise select the AGPR form of every MFMA builtin and allocate a0 ... for their accumulators,
// on top of the strip (profiles/r03_experiments.md section 3); the flag is a backend option, not a function attribute.
#include "../../include/aslam_core.h"

#include <hip/hip_runtime.h>

#include "ekf_large.h"
#include "aslam_large16.h"

namespace aslam
{
void launch_chol_bf16(const DevView &dv, const LargeView<float> &lv, int nfilters, const int *skipped, hipStream_t st, bool f32out)
{
        const LPlanes pl = {lv.Lpl};
        if (f32out)
                hipLaunchKernelGGL((large_chol_bf16<LARGE_NB_MAX, true>), dim3(nfilters), dim3(256), 0, st, dv, lv, pl, skipped);
        else
                hipLaunchKernelGGL((large_chol_bf16<LARGE_NB_MAX, false>), dim3(nfilters), dim3(256), 0, st, dv, lv, pl, skipped);
}

void launch_trsm_bf16(const DevView &dv, const LargeView<float> &lv, int nfilters, const int *skipped, hipStream_t st)
{
        const LPlanes pl = {lv.Lpl};
        const int NB = lv.NP / LB;
        hipLaunchKernelGGL((large_trsm_bf16<LARGE_NB_MAX>), dim3(8 * ((nfilters + 7) / 8) * NB), dim3(256), 0, st, dv, lv, pl, nfilters, skipped);
}
} // namespace aslam
