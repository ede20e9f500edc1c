"""awesomeslam_amd -- MI355X-native EKF/UKF-SLAM predict/update core behind the awesome_slam ekf|ukf node surface.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C-ABI library + the C++ host mirror of
the reference nodes), the ctypes binding, the synthetic trace generator and the multi-GPU sharding helper.
The CPU oracle is NOT part of this package (oracle/ is test infrastructure).
"""
__version__ = "0.1.0"
