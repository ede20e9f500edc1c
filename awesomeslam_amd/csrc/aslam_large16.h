// aslam_large16.h -- launchers of the bf16-pipe kernels compiled in aslam_large16.hip (their own translation unit: see there)
#pragma once

namespace aslam
{
/// S = L L^T for `nfilters` filters (one workgroup each): the bf16 planes of L and of the inverses of its diagonal blocks (lv.Lpl), the diagonal
/// blocks and their inverses in binary32, and -- f32out only: for large_trsm_pipe and the diagnostic read-back -- the rest of L in binary32
void launch_chol_bf16(const DevView &dv, const LargeView<float> &lv, int nfilters, const int *skipped, hipStream_t st, bool f32out);
/// V = G L^-T from the planes large_chol_bf16 wrote
void launch_trsm_bf16(const DevView &dv, const LargeView<float> &lv, int nfilters, const int *skipped, hipStream_t st);
} // namespace aslam
