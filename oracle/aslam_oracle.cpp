/*
 * aslam_oracle.cpp -- CPU oracle: Eigen-free, ROS-free restatement of the reference filter nodes.
 *
 * TEST INFRASTRUCTURE ONLY (see aslam_oracle.h).  "parity unpinned" by the reference: it has no
 * tests or golden vectors and cannot be built here; this file is pinned by oracle/np_oracle.py and
 * tests/golden/.
 *
 * Every function names the reference lines it restates (paths relative to /root/reference/awesome_slam).
 * The arithmetic keeps the reference's float/double mixing: wherever the reference
 * stores into a `float` or passes a `const float &`, the value is rounded to binary32 here too.
 * Matrices are row-major std::vector<double>; Eigen's dense products are restated as plain
 * k-ascending dot products, MatrixXd::inverse() as partial-pivot LU + solve against the identity
 * (Eigen 3.3 dynamic-size inverse = PartialPivLU), llt().matrixL() as the textbook lower Cholesky.
 * Third-party arithmetic not present under /root/reference: Eigen 3 (un-vendored, version unpinned;
 * Ubuntu 20.04 ships 3.3.7), call sites ekf.cpp:271-278,297,300-301,309-310 and
 * ukf.cpp:238-242,280,287-288,303,315,353,374,378,389-391.
 *
 * Build: g++ -O2 -ffp-contract=off (no -march=native, no -ffast-math): the reference's catkin build
 * sets no optimisation or ISA flags (CMakeLists.txt:1-45), so its x86-64 code has no fused
 * multiply-adds.
 */
#include "aslam_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace
{
// ---------------------------------------------------------------- include/awesome_slam/config.h:39-65
const float PI = 3.141592654;
const float MIN_DIST_THRESH = 0.5;
const int MIN_LANDMARK_OCC = 10;
const float UKF_STD_A = 0.2;
const float UKF_STD_YAW = 0.2;
const float UKF_KP_ROBOT_POSE = 0.001;
const float UKF_KP_LANDMARK_POSE = 1.0;
const float UKF_KR = 0.2;
const float UKF_KQ = 0.001;
const float EKF_KP_ROBOT_POSE = 0.001;
const float EKF_KP_LANDMARK_POSE = 10000.0;
const float EKF_KR = 0.2;
const float EKF_KQ = 0.001;

// ---------------------------------------------------------------- minimal dense types (stand-in for Eigen)
typedef std::vector<double> Vec;

struct Mat
{
        int r, c;
        std::vector<double> a;
        Mat() : r(0), c(0)
        {
        }
        Mat(int r_, int c_, double v = 0.0) : r(r_), c(c_), a((size_t)r_ * c_, v)
        {
        }
        double &operator()(int i, int j)
        {
                return a[(size_t)i * c + j];
        }
        double operator()(int i, int j) const
        {
                return a[(size_t)i * c + j];
        }
        static Mat Identity(int n, double s = 1.0)
        {
                Mat m(n, n);
                for (int i = 0; i < n; ++i)
                        m(i, i) = s;
                return m;
        }
};

#ifdef ORC_WITH_EIGEN
// Compile-gated variant (oracle/Makefile target `eigen`, built only where <eigen3/Eigen/Dense> exists): the dense
// routines the reference takes from Eigen -- operator* (ekf.cpp:297,300-301,309-310; ukf.cpp:287-288,374,378,389-391),
// MatrixXd::inverse() (ekf.cpp:301, ukf.cpp:378) and llt().matrixL() (ukf.cpp:280) -- are evaluated BY EIGEN on the column-major
// MatrixXd the reference uses, so that tests/test_oracle_eigen.py can pin this file's hand-written versions of them (below) against
// the library the reference links.  Everything else in this file is shared by both builds.
} // namespace (re-opened below)
#include <eigen3/Eigen/Dense>
namespace
{
Eigen::MatrixXd to_eigen(const Mat &A)
{
        Eigen::MatrixXd E(A.r, A.c);
        for (int i = 0; i < A.r; ++i)
                for (int j = 0; j < A.c; ++j)
                        E(i, j) = A(i, j);
        return E;
}
Mat from_eigen(const Eigen::MatrixXd &E)
{
        Mat A((int)E.rows(), (int)E.cols());
        for (int i = 0; i < A.r; ++i)
                for (int j = 0; j < A.c; ++j)
                        A(i, j) = E(i, j);
        return A;
}
Mat mul(const Mat &A, const Mat &B)
{
        return from_eigen(to_eigen(A) * to_eigen(B));
}
Mat inverse(const Mat &S)
{
        return from_eigen(to_eigen(S).inverse());
}
Mat lltMatrixL(const Mat &A)
{
        const Eigen::MatrixXd L = to_eigen(A).llt().matrixL();
        return from_eigen(L);
}
#else
/// C = A * B, each coefficient a k-ascending dot product
Mat mul(const Mat &A, const Mat &B)
{
        Mat C(A.r, B.c);
        for (int i = 0; i < A.r; ++i)
        {
                double *ci = &C.a[(size_t)i * C.c];
                for (int k = 0; k < A.c; ++k)
                {
                        const double aik = A(i, k);
                        const double *bk = &B.a[(size_t)k * B.c];
                        for (int j = 0; j < B.c; ++j)
                                ci[j] += aik * bk[j];
                }
        }
        return C;
}

#endif // ORC_WITH_EIGEN

Mat transpose(const Mat &A)
{
        Mat T(A.c, A.r);
        for (int i = 0; i < A.r; ++i)
                for (int j = 0; j < A.c; ++j)
                        T(j, i) = A(i, j);
        return T;
}

Mat add(const Mat &A, const Mat &B)
{
        Mat C(A.r, A.c);
        for (size_t i = 0; i < A.a.size(); ++i)
                C.a[i] = A.a[i] + B.a[i];
        return C;
}

Mat sub(const Mat &A, const Mat &B)
{
        Mat C(A.r, A.c);
        for (size_t i = 0; i < A.a.size(); ++i)
                C.a[i] = A.a[i] - B.a[i];
        return C;
}

Vec mulv(const Mat &A, const Vec &x)
{
        Vec y(A.r, 0.0);
        for (int i = 0; i < A.r; ++i)
        {
                double s = 0.0;
                for (int k = 0; k < A.c; ++k)
                        s += A(i, k) * x[k];
                y[i] = s;
        }
        return y;
}

/// Eigen's conservativeResizeLike(other): result has other's shape, keeps the old top-left block,
/// takes every new coefficient from `other` (ekf.cpp:271-278, ukf.cpp:238-242).
void conservativeResizeLike(Mat &M, const Mat &other)
{
        Mat N = other;
        for (int i = 0; i < std::min(M.r, N.r); ++i)
                for (int j = 0; j < std::min(M.c, N.c); ++j)
                        N(i, j) = M(i, j);
        M = N;
}

void conservativeResizeLike(Vec &v, const Vec &other)
{
        Vec n = other;
        for (size_t i = 0; i < std::min(v.size(), n.size()); ++i)
                n[i] = v[i];
        v = n;
}

#ifndef ORC_WITH_EIGEN
/// MatrixXd::inverse() for a dynamic matrix = PartialPivLU(S).inverse() = solve(Identity)
Mat inverse(const Mat &S)
{
        const int n = S.r;
        Mat lu = S;
        std::vector<int> perm(n);
        for (int i = 0; i < n; ++i)
                perm[i] = i;
        for (int k = 0; k < n; ++k)
        {
                int piv = k;
                double best = std::fabs(lu(k, k));
                for (int i = k + 1; i < n; ++i)
                {
                        if (std::fabs(lu(i, k)) > best)
                        {
                                best = std::fabs(lu(i, k));
                                piv = i;
                        }
                }
                if (piv != k)
                {
                        for (int j = 0; j < n; ++j)
                                std::swap(lu(k, j), lu(piv, j));
                        std::swap(perm[k], perm[piv]);
                }
                const double d = lu(k, k);
                for (int i = k + 1; i < n; ++i)
                {
                        const double l = lu(i, k) / d;
                        lu(i, k) = l;
                        double *ri = &lu.a[(size_t)i * n];
                        const double *rk = &lu.a[(size_t)k * n];
                        for (int j = k + 1; j < n; ++j)
                                ri[j] -= l * rk[j];
                }
        }
        // X = P * I, then L y = X (unit lower), U x = y, all right-hand sides at once
        Mat X(n, n);
        for (int i = 0; i < n; ++i)
                X(i, perm[i]) = 1.0;
        for (int i = 0; i < n; ++i)
        {
                double *xi = &X.a[(size_t)i * n];
                for (int k = 0; k < i; ++k)
                {
                        const double l = lu(i, k);
                        const double *xk = &X.a[(size_t)k * n];
                        for (int j = 0; j < n; ++j)
                                xi[j] -= l * xk[j];
                }
        }
        for (int i = n - 1; i >= 0; --i)
        {
                double *xi = &X.a[(size_t)i * n];
                for (int k = i + 1; k < n; ++k)
                {
                        const double u = lu(i, k);
                        const double *xk = &X.a[(size_t)k * n];
                        for (int j = 0; j < n; ++j)
                                xi[j] -= u * xk[j];
                }
                const double d = lu(i, i);
                for (int j = 0; j < n; ++j)
                        xi[j] /= d;
        }
        return X;
}

/// Paug.llt().matrixL(): lower Cholesky factor; like Eigen, a non-positive pivot is not reported
/// (ukf.cpp:280 never checks info()), the square root of a negative number simply yields NaN.
Mat lltMatrixL(const Mat &A)
{
        const int n = A.r;
        Mat L(n, n);
        for (int j = 0; j < n; ++j)
        {
                double d = A(j, j);
                for (int k = 0; k < j; ++k)
                        d -= L(j, k) * L(j, k);
                d = std::sqrt(d);
                L(j, j) = d;
                for (int i = j + 1; i < n; ++i)
                {
                        double s = A(i, j);
                        for (int k = 0; k < j; ++k)
                                s -= L(i, k) * L(j, k);
                        L(i, j) = s / d;
                }
        }
        return L;
}
#endif // !ORC_WITH_EIGEN

// ---------------------------------------------------------------- include/awesome_slam/tools.h:44-66
/// tools.h:44-50 -- all binary32
inline float normalizeAngle(const float &theta)
{
        float ret = std::fmod(theta, 2 * PI);
        ret = ret > PI ? ret - 2 * PI : ret;
        ret = ret < -PI ? ret + 2 * PI : ret;
        return ret;
}

/// tools.h:53-59 -- the Vector2d coefficients are doubles, the differences are stored in floats
inline float eulerDistance(const double p1[2], const double p2[2])
{
        float dx = p1[0] - p2[0];
        float dy = p1[1] - p2[1];
        float ret = std::sqrt(dx * dx + dy * dy);
        return ret;
}

/// tools.h:62-66
inline float quat2euler(const float &w, const float &x, const float &y, const float &z)
{
        float yaw = std::atan2(2 * (w * z + x * y), 1 - 2 * (z * z + y * y));
        return yaw;
}

// ---------------------------------------------------------------- include/awesome_slam/structures.h:44-112
struct Point
{
        double point[2]; // Eigen::Vector2d
        Point()
        {
                point[0] = point[1] = 0.0;
        }
        Point(const float &a, const float &b)
        {
                point[0] = a;
                point[1] = b;
        }
        void assign(const float &a, const float &b)
        {
                point[0] = a;
                point[1] = b;
        }
        void assign(const Point &P)
        {
                point[0] = P.point[0];
                point[1] = P.point[1];
        }
        float distance(const Point &P) const
        {
                float ret = eulerDistance(point, P.point);
                return ret;
        }
};

struct LaserData
{
        float range;
        float bearing;
        void assign(const float &a, const float &b)
        {
                range = a;
                bearing = b;
        }
        /// structures.h:104-111
        Point toPoint(const Vec &Z) const
        {
                float a = Z[0] + range * std::cos(Z[2] + bearing);
                float b = Z[1] + range * std::sin(Z[2] + bearing);
                Point ret(a, b);
                return ret;
        }
};

// ---------------------------------------------------------------- include/awesome_slam/common.h:46-90
/// common.h:46-75
Vec stateTransitionFunction(const uint32_t N, const Vec &point, const float &vx, const float &az,
                            const float &delta_time)
{
        Vec P(point.begin(), point.begin() + N);

        if (std::fabs(az) > 0.001)
        {
                float r = vx / az;
                P[0] += r * (-std::sin(point[2]) + std::sin(point[2] + az * delta_time));
                P[1] += r * (std::cos(point[2]) - std::cos(point[2] + az * delta_time));
        }
        else
        {
                P[0] += vx * delta_time * std::cos(point[2]);
                P[1] += vx * delta_time * std::sin(point[2]);
        }

        P[2] += az * delta_time;

        if (point.size() > N)
        {
                P[0] += 0.5 * delta_time * delta_time * point[N] * std::cos(point[2]);
                P[1] += 0.5 * delta_time * delta_time * point[N] * std::sin(point[2]);
                P[2] += 0.5 * delta_time * delta_time * az;
        }

        return P;
}

/// common.h:78-90
Vec measurementFunction(const uint32_t N, const Vec &point)
{
        Vec P = point;
        for (uint32_t i = 0; i < N - 3; i += 2)
        {
                P[3 + i] = std::sqrt(std::pow(point[3 + i] - point[0], 2) + std::pow(point[4 + i] - point[1], 2));
                P[4 + i] = std::atan2(point[4 + i] - point[1], point[3 + i] - point[0]) - point[2];
        }
        return P;
}
} // namespace

// ==================================================================== the filter nodes
struct orc_filter
{
        int kind;
        int MAX_LANDMARK_COUNT; // config.h:45, run-time here

        // members of EKFSlam / UKFSlam (ekf.h:90-108, ukf.h:102-120)
        uint32_t N;
        bool init_z;
        bool init_x;
        std::vector<LaserData> sensor_landmark;
        std::vector<std::pair<LaserData, uint32_t>> new_landmark_wait;

        // Parameters (ekf.h:56-67, ukf.h:56-82)
        Vec X, Z, weights;
        Mat I, A, P, H, Q, R;
        float lambda;

        /// ukf.h:73-81
        void updateWeights(uint32_t n)
        {
                lambda = 3.0 - (n + 2);
                float weight = 0.5 / (lambda + n + 2);
                weights.assign(2 * n + 5, 1.0 * weight);
                weights[0] = lambda / (lambda + n + 2);
        }

        /// ekf.cpp:49-71 / ukf.cpp:49-67
        void initialize()
        {
                N = 3;
                init_x = true;
                init_z = true;
                sensor_landmark.clear();
                new_landmark_wait.clear();

                X.assign(N, 0.0);
                Z.assign(N, 0.0);
                Q = Mat(N, N);
                if (kind == ORC_EKF)
                {
                        A = Mat::Identity(N);
                        H = Mat::Identity(N);
                        I = Mat::Identity(N);
                        R = Mat::Identity(N, EKF_KR);
                        P = Mat::Identity(N, EKF_KP_ROBOT_POSE);
                        Q(0, 0) = EKF_KQ;
                        Q(1, 1) = EKF_KQ;
                        Q(2, 2) = EKF_KQ;
                        // bottomRightCorner(N - 3, N - 3) is empty at N = 3: EKF_KP_LANDMARK_POSE is never used
                        (void)EKF_KP_LANDMARK_POSE;
                }
                else
                {
                        updateWeights(3); // Parameters() constructor, ukf.h:68-71
                        R = Mat::Identity(N, UKF_KR);
                        P = Mat::Identity(N, UKF_KP_ROBOT_POSE);
                        Q(0, 0) = UKF_KQ;
                        Q(1, 1) = UKF_KQ;
                        Q(2, 2) = UKF_KQ;
                }
        }

        /// ekf.cpp:102-114 / ukf.cpp:98-110
        void cbSensorLandmark(int size, const double *x, const double *y)
        {
                init_z = false;
                sensor_landmark.clear();
                LaserData data;
                for (int i = 0; i < size; ++i)
                {
                        data.assign(x[i], y[i]);
                        sensor_landmark.push_back(data);
                }
        }

        /// ekf.cpp:117-134
        void updateH()
        {
                for (uint32_t i = 0; i < N - 3; i += 2)
                {
                        float hyp = std::pow(X[3 + i] - X[0], 2) + std::pow(X[4 + i] - X[1], 2);
                        float dist = std::sqrt(hyp);

                        H(3 + i, 0) = (-X[3 + i] + X[0]) / dist;
                        H(4 + i, 0) = -(-X[4 + i] + X[1]) / hyp;
                        H(3 + i, 1) = (-X[4 + i] + X[1]) / dist;
                        H(4 + i, 1) = (-X[3 + i] + X[0]) / hyp;
                        H(4 + i, 2) = -1;
                        H(3 + i, 3 + i) = -(-X[3 + i] + X[0]) / dist;
                        H(3 + i, 4 + i) = -(-X[4 + i] + X[1]) / dist;
                        H(4 + i, 3 + i) = (-X[4 + i] + X[1]) / hyp;
                        H(4 + i, 4 + i) = -(-X[3 + i] + X[0]) / hyp;
                }
        }

        /// ekf.cpp:217-253 / ukf.cpp:184-220
        void updateNewLandmarkWait(const LaserData &data)
        {
                if (new_landmark_wait.empty())
                {
                        new_landmark_wait.push_back({data, 1});
                        return;
                }

                uint32_t corr_id = 0;
                Point landmark_2 = data.toPoint(Z);
                Point landmark_1 = new_landmark_wait[0].first.toPoint(Z);
                float mindist = landmark_2.distance(landmark_1);
                uint32_t size = new_landmark_wait.size();
                for (uint32_t i = 1; i < size; ++i)
                {
                        landmark_1.assign(new_landmark_wait[i].first.toPoint(Z));
                        float dist = landmark_2.distance(landmark_1);
                        if (dist < mindist)
                        {
                                corr_id = i;
                                mindist = dist;
                        }
                }

                if (mindist < MIN_DIST_THRESH)
                {
                        new_landmark_wait[corr_id].second++;
                }
                else
                {
                        new_landmark_wait.push_back({data, 1});
                }
        }

        /// ekf.cpp:255-290 / ukf.cpp:222-257
        void updateNewLandmark(const std::vector<LaserData> &new_landmark)
        {
                uint32_t cacheN = N;
                uint32_t size = new_landmark.size();

                N += 2 * size;

                if (N >= (uint32_t)MAX_LANDMARK_COUNT)
                {
                        N = cacheN;
                        return;
                }

                conservativeResizeLike(X, Vec(N, 0.0));
                conservativeResizeLike(Z, Vec(N, 0.0));
                conservativeResizeLike(Q, Mat(N, N));
                if (kind == ORC_EKF)
                {
                        conservativeResizeLike(I, Mat::Identity(N));
                        conservativeResizeLike(A, Mat::Identity(N));
                        conservativeResizeLike(H, Mat::Identity(N));
                }
                // both nodes use the UKF constants here (ekf.cpp:277-278, ukf.cpp:241-242)
                conservativeResizeLike(P, Mat::Identity(N, 1.0 * UKF_KP_LANDMARK_POSE));
                conservativeResizeLike(R, Mat::Identity(N, 1.0 * UKF_KR));

                if (kind == ORC_UKF)
                        updateWeights(N);

                for (uint32_t i = 0; i < size * 2; i += 2)
                {
                        Z[cacheN + i] = new_landmark[i / 2].range;
                        Z[cacheN + i + 1] = new_landmark[i / 2].bearing;

                        X[cacheN + i] = Z[0] + Z[cacheN + i] * std::cos(Z[2] + Z[cacheN + i + 1]);
                        X[cacheN + i + 1] = Z[1] + Z[cacheN + i] * std::sin(Z[2] + Z[cacheN + i + 1]);
                }
        }

        /// ekf.cpp:137-213 (updateZandA) / ukf.cpp:113-180 (updateZ)
        void updateZ(double px, double py, double qw, double qx, double qy, double qz, double twist_vx,
                     double twist_wz, const float &delta_time)
        {
                Z[0] = px;
                Z[1] = py;
                Z[2] = quat2euler(qw, qx, qy, qz);

                std::vector<LaserData> new_landmark;

                for (LaserData &data : sensor_landmark)
                {
                        data.bearing = normalizeAngle(data.bearing);

                        if (N == 3)
                        {
                                updateNewLandmarkWait(data);
                                continue;
                        }

                        // (corr_id is `int` in ekf.cpp:160 and `uint32_t` in ukf.cpp:136; never negative)
                        uint32_t corr_id = 0;
                        Point landmark_1(X[3], X[4]);
                        Point landmark_2 = data.toPoint(Z);
                        float mindist = landmark_2.distance(landmark_1);
                        for (uint32_t j = 2; j < N - 3; j += 2)
                        {
                                landmark_1.assign(X[3 + j], X[4 + j]);
                                float dist = landmark_2.distance(landmark_1);
                                if (dist < mindist)
                                {
                                        corr_id = j;
                                        mindist = dist;
                                }
                        }

                        if (mindist < MIN_DIST_THRESH)
                        {
                                Z[3 + corr_id] = data.range;
                                Z[4 + corr_id] = data.bearing;
                                continue;
                        }
                        else
                        {
                                updateNewLandmarkWait(data);
                        }
                }

                for (auto &waitingData : new_landmark_wait)
                {
                        if (waitingData.second == (uint32_t)MIN_LANDMARK_OCC)
                        {
                                new_landmark.push_back(waitingData.first);
                                waitingData.second += 1;
                        }
                }

                if (new_landmark.size())
                {
                        updateNewLandmark(new_landmark);
                        new_landmark.clear();
                }

                // Update A (EKF only, ekf.cpp:206-212)
                if (kind == ORC_EKF && twist_vx && twist_wz)
                {
                        float delta_theta = twist_wz * delta_time;
                        float r = twist_vx / twist_wz;
                        A(0, 0) = r * (-std::cos(Z[2]) + std::cos(Z[2] + delta_theta));
                        A(1, 0) = r * (-std::sin(Z[2]) + std::sin(Z[2] + delta_theta));
                }
        }

        /// ekf.cpp:293-311
        void slamEKF(const float &vx, const float &az, const float &delta_time)
        {
                X = stateTransitionFunction(N, X, vx, az, delta_time);
                X[2] = normalizeAngle(X[2]);
                P = add(mul(mul(A, P), transpose(A)), Q);

                updateH();
                Mat S = add(mul(mul(H, P), transpose(H)), R);
                Mat K = mul(mul(P, transpose(H)), inverse(S));
                Vec hx = measurementFunction(N, X);
                Vec Y(N);
                for (uint32_t i = 0; i < N; ++i)
                        Y[i] = Z[i] - hx[i];

                for (uint32_t j = 0; j < N - 1; j += 2)
                {
                        Y[2 + j] = normalizeAngle(Y[2 + j]);
                }

                Vec KY = mulv(K, Y);
                for (uint32_t i = 0; i < N; ++i)
                        X[i] = X[i] + KY[i];
                P = mul(sub(I, mul(K, H)), P);
        }

        /// ukf.cpp:260-392
        void slamUKF(const float &vx, const float &az, const float &delta_time)
        {
                const uint32_t M = 2 * N + 5;
                Vec Xaug(N + 2);
                Mat Paug(N + 2, N + 2);

                for (uint32_t i = 0; i < N; ++i)
                        Xaug[i] = X[i];
                Xaug[N] = 0.0;
                Xaug[N + 1] = 0.0;

                for (uint32_t i = 0; i < N; ++i)
                        for (uint32_t j = 0; j < N; ++j)
                                Paug(i, j) = P(i, j);
                Paug(N, N) = UKF_STD_A * UKF_STD_A;
                Paug(N + 1, N + 1) = UKF_STD_YAW * UKF_STD_YAW;

                Mat L = lltMatrixL(Paug);

                // sigma points, one per column in the reference; stored one per row here
                std::vector<Vec> XsigAug(M, Vec(N + 2));
                XsigAug[0] = Xaug;
                float w = std::sqrt(lambda + N + 2);
                for (uint32_t i = 0; i < N + 2; ++i)
                {
                        for (uint32_t k = 0; k < N + 2; ++k)
                        {
                                XsigAug[i + 1][k] = Xaug[k] + w * L(k, i);
                                XsigAug[i + 3 + N][k] = Xaug[k] - w * L(k, i);
                        }
                }

                std::vector<Vec> XsigPred(M);
                for (uint32_t i = 0; i < M; ++i)
                {
                        XsigPred[i] = stateTransitionFunction(N, XsigAug[i], vx, az, delta_time);
                        XsigPred[i][2] = normalizeAngle(XsigPred[i][2]);
                }

                // predicted state mean
                std::fill(X.begin(), X.end(), 0.0);
                for (uint32_t i = 0; i < M; ++i)
                        for (uint32_t k = 0; k < N; ++k)
                                X[k] += weights[i] * XsigPred[i][k];

                // predicted state covariance
                Vec Xdiff(N);
                P = Mat(N, N);
                for (uint32_t i = 0; i < M; ++i)
                {
                        for (uint32_t k = 0; k < N; ++k)
                                Xdiff[k] = XsigPred[i][k] - X[k];
                        Xdiff[2] = normalizeAngle(Xdiff[2]);

                        // weights(i) * Xdiff * Xdiff.transpose(): (scalar * vector) first, then the outer product
                        for (uint32_t a = 0; a < N; ++a)
                        {
                                const double wa = weights[i] * Xdiff[a];
                                for (uint32_t b = 0; b < N; ++b)
                                        P(a, b) += wa * Xdiff[b];
                        }
                }
                P = add(P, Q);

                std::vector<Vec> Zsig(M);
                for (uint32_t i = 0; i < M; ++i)
                        Zsig[i] = measurementFunction(N, XsigPred[i]);

                Vec Zpred(N, 0.0);
                for (uint32_t i = 0; i < M; ++i)
                        for (uint32_t k = 0; k < N; ++k)
                                Zpred[k] += weights[i] * Zsig[i][k];

                for (uint32_t j = 0; j < N - 1; j += 2)
                        Zpred[2 + j] = normalizeAngle(Zpred[2 + j]);

                Vec Zdiff(N);
                Mat S(N, N);
                for (uint32_t i = 0; i < M; ++i)
                {
                        for (uint32_t k = 0; k < N; ++k)
                                Zdiff[k] = Zsig[i][k] - Zpred[k];
                        for (uint32_t j = 0; j < N - 1; j += 2)
                                Zdiff[2 + j] = normalizeAngle(Zdiff[2 + j]);

                        for (uint32_t a = 0; a < N; ++a)
                        {
                                const double wa = weights[i] * Zdiff[a];
                                for (uint32_t b = 0; b < N; ++b)
                                        S(a, b) += wa * Zdiff[b];
                        }
                }
                S = add(S, R);

                Mat Tc(N, N);
                for (uint32_t i = 0; i < M; ++i)
                {
                        for (uint32_t k = 0; k < N; ++k)
                                Zdiff[k] = Zsig[i][k] - Zpred[k];
                        for (uint32_t j = 0; j < N - 1; j += 2)
                                Zdiff[2 + j] = normalizeAngle(Zdiff[2 + j]);

                        for (uint32_t k = 0; k < N; ++k)
                                Xdiff[k] = XsigPred[i][k] - X[k];
                        Xdiff[2] = normalizeAngle(Xdiff[2]);

                        for (uint32_t a = 0; a < N; ++a)
                        {
                                const double wa = weights[i] * Xdiff[a];
                                for (uint32_t b = 0; b < N; ++b)
                                        Tc(a, b) += wa * Zdiff[b];
                        }
                }

                Mat K = mul(Tc, inverse(S));

                for (uint32_t k = 0; k < N; ++k)
                        Zdiff[k] = Z[k] - Zpred[k];
                for (uint32_t j = 0; j < N - 1; j += 2)
                        Zdiff[2 + j] = normalizeAngle(Zdiff[2 + j]);

                Vec KZ = mulv(K, Zdiff);
                for (uint32_t k = 0; k < N; ++k)
                        X[k] = X[k] + KZ[k];

                P = sub(P, mul(mul(K, S), transpose(K)));
        }

        void slam(const float &vx, const float &az, const float &delta_time)
        {
                if (kind == ORC_EKF)
                        slamEKF(vx, az, delta_time);
                else
                        slamUKF(vx, az, delta_time);
        }

        /// ekf.cpp:74-99 / ukf.cpp:70-95, delta_time supplied
        int cbOdom(double px, double py, double qw, double qx, double qy, double qz, double twist_vx, double twist_wz,
                   float delta_time)
        {
                if (init_z)
                        return 0;

                updateZ(px, py, qw, qx, qy, qz, twist_vx, twist_wz, delta_time);

                if (init_x)
                {
                        init_x = false;
                        X = Z;
                }

                slam(twist_vx, twist_wz, delta_time);
                return 1;
        }
};

// ==================================================================== C interface
extern "C" {

orc_filter *orc_create(int kind, int max_landmark_count)
{
        orc_filter *f = new orc_filter();
        f->kind = kind;
        f->MAX_LANDMARK_COUNT = max_landmark_count;
        f->lambda = 0.0f;
        f->initialize();
        return f;
}

void orc_destroy(orc_filter *f)
{
        delete f;
}

void orc_sensor(orc_filter *f, int n, const double *x, const double *y)
{
        f->cbSensorLandmark(n, x, y);
}

int orc_odom(orc_filter *f, double px, double py, double qw, double qx, double qy, double qz, double vx, double wz,
             float delta_time)
{
        return f->cbOdom(px, py, qw, qx, qy, qz, vx, wz, delta_time);
}

void orc_slam(orc_filter *f, float vx, float az, float delta_time)
{
        f->slam(vx, az, delta_time);
}

int orc_dim(const orc_filter *f)
{
        return (int)f->N;
}

void orc_get(const orc_filter *f, double *X, double *Z, double *P)
{
        const size_t n = f->N;
        if (X)
                std::memcpy(X, f->X.data(), n * sizeof(double));
        if (Z)
                std::memcpy(Z, f->Z.data(), n * sizeof(double));
        if (P)
                std::memcpy(P, f->P.a.data(), n * n * sizeof(double));
}

void orc_get_A(const orc_filter *f, double *a00, double *a10)
{
        *a00 = f->kind == ORC_EKF ? f->A(0, 0) : 1.0;
        *a10 = f->kind == ORC_EKF ? f->A(1, 0) : 0.0;
}

void orc_set(orc_filter *f, int N, const double *X, const double *Z, const double *P, double a00, double a10)
{
        f->initialize();
        f->init_z = false;
        f->init_x = false;
        // grow the constant matrices exactly as updateNewLandmark does, in one go
        f->N = N;
        conservativeResizeLike(f->Q, Mat(N, N));
        if (f->kind == ORC_EKF)
        {
                conservativeResizeLike(f->I, Mat::Identity(N));
                conservativeResizeLike(f->A, Mat::Identity(N));
                conservativeResizeLike(f->H, Mat::Identity(N));
                f->A(0, 0) = a00;
                f->A(1, 0) = a10;
        }
        conservativeResizeLike(f->R, Mat::Identity(N, 1.0 * UKF_KR));
        if (f->kind == ORC_UKF)
                f->updateWeights(N);
        f->X.assign(X, X + N);
        f->Z.assign(Z, Z + N);
        f->P = Mat(N, N);
        std::memcpy(f->P.a.data(), P, (size_t)N * N * sizeof(double));
}

int orc_wait_size(const orc_filter *f)
{
        return (int)f->new_landmark_wait.size();
}

void orc_get_wait(const orc_filter *f, float *range, float *bearing, uint32_t *count)
{
        for (size_t i = 0; i < f->new_landmark_wait.size(); ++i)
        {
                range[i] = f->new_landmark_wait[i].first.range;
                bearing[i] = f->new_landmark_wait[i].first.bearing;
                count[i] = f->new_landmark_wait[i].second;
        }
}

int orc_sensor_size(const orc_filter *f)
{
        return (int)f->sensor_landmark.size();
}

void orc_get_sensor(const orc_filter *f, float *range, float *bearing)
{
        for (size_t i = 0; i < f->sensor_landmark.size(); ++i)
        {
                range[i] = f->sensor_landmark[i].range;
                bearing[i] = f->sensor_landmark[i].bearing;
        }
}

void orc_get_weights(const orc_filter *f, double *w, float *lambda)
{
        std::memcpy(w, f->weights.data(), f->weights.size() * sizeof(double));
        *lambda = f->lambda;
}

int64_t orc_replay(orc_filter *f, int64_t T, const double *odom, const float *dt, const uint8_t *obs_new,
                   const int32_t *n_obs, const double *obs, int max_obs, double *poses_out, int32_t *dims_out)
{
        int64_t ran = 0;
        std::vector<double> xs(max_obs), ys(max_obs);
        for (int64_t t = 0; t < T; ++t)
        {
                if (obs_new[t])
                {
                        const double *o = obs + (size_t)t * max_obs * 2;
                        for (int i = 0; i < n_obs[t]; ++i)
                        {
                                xs[i] = o[2 * i];
                                ys[i] = o[2 * i + 1];
                        }
                        f->cbSensorLandmark(n_obs[t], xs.data(), ys.data());
                }
                const double *m = odom + (size_t)t * 8;
                int r = f->cbOdom(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], dt[t]);
                ran += r;
                if (poses_out)
                {
                        poses_out[3 * t + 0] = r ? f->X[0] : 0.0;
                        poses_out[3 * t + 1] = r ? f->X[1] : 0.0;
                        poses_out[3 * t + 2] = r ? f->X[2] : 0.0;
                }
                if (dims_out)
                        dims_out[t] = (int32_t)f->N;
        }
        return ran;
}

float orc_normalize_angle(float theta)
{
        return normalizeAngle(theta);
}

float orc_quat2euler(float w, float x, float y, float z)
{
        return quat2euler(w, x, y, z);
}

} // extern "C"
