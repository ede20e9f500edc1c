// device_common.h -- shared device-side pieces of the MI355X (gfx950) EKF/UKF-SLAM core.
//
// Everything in the "bit-exact" section reproduces a binary32 rounding point of the reference
// (SURVEY.md F3); this translation unit is compiled with -ffp-contract=off so that none of those
// expressions is fused, and the hot loops ask for FMAs explicitly with fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aslam
{
typedef double d4 __attribute__((ext_vector_type(4)));

// ---- include/awesome_slam/config.h:39-65 -------------------------------------------------------------
constexpr float PI_F = 3.141592654f;         // const float PI = 3.141592654
constexpr float TWO_PI_F = 2.0f * PI_F;      // `2 * PI` (int * float)
constexpr float MIN_DIST_THRESH = 0.5f;      // config.h:43
constexpr unsigned MIN_LANDMARK_OCC = 10;    // config.h:44
constexpr float UKF_STD_A = 0.2f;            // config.h:54
constexpr float UKF_STD_YAW = 0.2f;          // config.h:55
constexpr float KP_ROBOT_POSE = 0.001f;      // EKF_KP_ROBOT_POSE == UKF_KP_ROBOT_POSE (config.h:56,63)
constexpr float KP_LANDMARK_POSE = 1.0f;     // UKF_KP_LANDMARK_POSE, used by BOTH nodes on growth (ekf.cpp:277)
constexpr float KR = 0.2f;                   // EKF_KR == UKF_KR (config.h:58,65)
constexpr float KQ = 0.001f;                 // EKF_KQ == UKF_KQ (config.h:59,66)

// flags of one filter
constexpr int FLAG_INIT_X = 1; // init_x (ekf.h:93)
constexpr int FLAG_INIT_Z = 2; // init_z (ekf.h:96)

// ---- bit-exact helpers ------------------------------------------------------------------------------
/// tools.h:44-50 (all binary32; fmodf is exact)
__device__ __forceinline__ float normalizeAngle(float theta)
{
        float ret = fmodf(theta, TWO_PI_F);
        ret = ret > PI_F ? ret - TWO_PI_F : ret;
        ret = ret < -PI_F ? ret + TWO_PI_F : ret;
        return ret;
}

/// tools.h:53-59 on two points whose coordinates are binary32 values held in doubles
__device__ __forceinline__ float eulerDistance(float ax, float ay, float bx, float by)
{
        float dx = (float)((double)ax - (double)bx);
        float dy = (float)((double)ay - (double)by);
        return sqrtf(dx * dx + dy * dy);
}

/// LaserData::toPoint, structures.h:104-111
__device__ __forceinline__ void toPoint(float range, float bearing, double z0, double z1, double z2, float &a, float &b)
{
        const double ang = z2 + (double)bearing;
        a = (float)(z0 + (double)range * cos(ang));
        b = (float)(z1 + (double)range * sin(ang));
}

/// stateTransitionFunction, common.h:46-75, on the three pose entries (landmarks pass through).
/// `aug` = the input vector is longer than N (UKF sigma point); noise_a = point(N).
__device__ __forceinline__ void stateTransition(double &p0, double &p1, double &p2, float vx, float az, float dt,
                                                bool aug, double noise_a)
{
        const double th = p2;
        if (fabsf(az) > 0.001)
        {
                const float r = vx / az;
                const double th2 = th + (double)(az * dt);
                p0 += (double)r * (-sin(th) + sin(th2));
                p1 += (double)r * (cos(th) - cos(th2));
        }
        else
        {
                const float vdt = vx * dt;
                p0 += (double)vdt * cos(th);
                p1 += (double)vdt * sin(th);
        }
        p2 += (double)(az * dt);
        if (aug)
        {
                const double h = 0.5 * (double)dt * (double)dt;
                p0 += h * noise_a * cos(th);
                p1 += h * noise_a * sin(th);
                p2 += h * (double)az;
        }
}

// ---- wave helpers -----------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
        union
        {
                double d;
                int i[2];
        } u;
        u.d = v;
        u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
        u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
        return u.d;
}

__device__ __forceinline__ double readfirstlane_f64(double v)
{
        union
        {
                double d;
                int i[2];
        } u;
        u.d = v;
        u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
        u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
        return u.d;
}

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c)
{
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
} // namespace aslam
