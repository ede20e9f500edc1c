// aslam_large16.h -- launchers of the bf16-pipe kernels compiled in aslam_large16.hip (their own translation unit: see there)
#pragma once

namespace aslam
{
/// S = L L^T for `nfilters` filters (one workgroup each): L, Linv and their bf16 planes (lv.Lpl)
void launch_chol_bf16(const DevView &dv, const LargeView<float> &lv, int nfilters, const int *skipped, hipStream_t st);
/// V = G L^-T from the planes large_chol_bf16 wrote
void launch_trsm_bf16(const DevView &dv, const LargeView<float> &lv, int nfilters, const int *skipped, hipStream_t st);
} // namespace aslam
