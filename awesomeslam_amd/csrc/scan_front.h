// scan_front.h -- scan -> (range, bearing) landmarks (include/aslam_scan.h; sensor_landmark.cpp:59-298), one wavefront per scan.
//
// The reference walks the 360 beams once, growing a cluster while consecutive points are closer than MIN_DIST_THRESH and
// evaluating it when the chain breaks (:92-121).  As coded that is: clusters = maximal runs of "linked" beams
// (link(t) = dist(p(t-1), p(t)) < thresh) that END BEFORE beam 359; the beam that breaks a chain belongs to no cluster
// (`cluster.clear(); p1 = p2`, :113-120), except that beam 0 opens the first cluster (:84-85); a run that reaches beam 359
// is never evaluated.  Runs are independent, so: links and run ends by ballots, one lane per run for the classifier
// (:147-186) and the hyper fit (:192-298), outputs compacted in beam order.  binary32 where the reference computes in
// `float` (-ffp-contract=off); the 4x4 algebra of the fit in fp64.
//
// The fit: the reference takes the SVD Z = U S V^T of the n x 4 data matrix, Y = V S V^T, Q = Y Hinv Y, the eigenvector A*
// of Q for the smallest positive eigenvalue, and solves Y A = A* (:243-283).  Here V and S^2 come from the Jacobi
// eigen-decomposition of the 4x4 matrix Z^T Z (same V, S = sqrt of the eigenvalues), the rest is the same 4x4 algebra.
#pragma once

#include "device_common.h"

namespace aslam
{
constexpr int SCAN_BEAMS = 360;
constexpr float SCAN_STD = 0.4f, SCAN_MIN_MEAN = 1.5f, SCAN_MAX_MEAN = 3.0f; // config.h:49-51
constexpr int SCAN_MIN_CLUSTER_POINTS = 3;                                   // config.h:52

/// eigen-decomposition of a symmetric 4x4 matrix (cyclic Jacobi): A -> diagonal, V = eigenvectors in columns
__device__ __forceinline__ void jacobi4(double (&A)[4][4], double (&V)[4][4])
{
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                        V[i][j] = (i == j) ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 16; ++sweep)
        {
                double off = 0.0, diag = 0.0;
#pragma unroll
                for (int p = 0; p < 4; ++p)
                {
                        diag += A[p][p] * A[p][p];
#pragma unroll
                        for (int q = p + 1; q < 4; ++q)
                                off += A[p][q] * A[p][q];
                }
                if (!(off > 1e-60 * diag) || !(off > 0.0))
                        break;
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int q = p + 1; q < 4; ++q)
                        {
                                const double apq = A[p][q];
                                if (apq != 0.0)
                                {
                                        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                                        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                                        for (int k = 0; k < 4; ++k)
                                        {
                                                const double akp = A[k][p], akq = A[k][q];
                                                A[k][p] = c * akp - s * akq;
                                                A[k][q] = s * akp + c * akq;
                                        }
#pragma unroll
                                        for (int k = 0; k < 4; ++k)
                                        {
                                                const double apk = A[p][k], aqk = A[q][k];
                                                A[p][k] = c * apk - s * aqk;
                                                A[q][k] = s * apk + c * aqk;
                                        }
#pragma unroll
                                        for (int k = 0; k < 4; ++k)
                                        {
                                                const double vkp = V[k][p], vkq = V[k][q];
                                                V[k][p] = c * vkp - s * vkq;
                                                V[k][q] = s * vkp + c * vkq;
                                        }
                                }
                        }
        }
}

/// cluster member i of the run [s0, e] (consecutive beams)
struct ScanPts
{
        const float *px, *py;
        int s0;
        __device__ __forceinline__ float x(int i) const
        {
                return px[s0 + i];
        }
        __device__ __forceinline__ float y(int i) const
        {
                return py[s0 + i];
        }
};

/// circleClassification, sensor_landmark.cpp:147-186
__device__ __forceinline__ bool scan_classify(const ScanPts &c, int n)
{
        const float x1 = c.x(0), y1 = c.y(0), x2 = c.x(n - 1), y2 = c.y(n - 1);
        const float cc = eulerDistance(x1, y1, x2, y2);
        float sum = 0.0f;
        for (int i = 1; i < n - 1; ++i)
        {
                const float a = eulerDistance(x1, y1, c.x(i), c.y(i)), b = eulerDistance(x2, y2, c.x(i), c.y(i));
                sum += acosf((a * a + b * b - cc * cc) / (2.0f * a * b));
        }
        const float mean = sum / (float)(n - 2);
        float variance = 0.0f;
        for (int i = 1; i < n - 1; ++i)
        {
                // the angles again (the reference keeps them in a vector): pow(float, int) is evaluated in double and added into a float
                const float a = eulerDistance(x1, y1, c.x(i), c.y(i)), b = eulerDistance(x2, y2, c.x(i), c.y(i));
                const float ang = acosf((a * a + b * b - cc * cc) / (2.0f * a * b));
                const double dv = (double)(ang - mean);
                variance = (float)((double)variance + dv * dv);
        }
        const float sd = (float)sqrt((double)variance / 5.0); // `/ 5.0` whatever the cluster size, as coded (:178)
        return sd < SCAN_STD && mean > SCAN_MIN_MEAN && mean < SCAN_MAX_MEAN;
}

/// circleFitting (:192-298) + Circle::toLaserData (structures.h:124-131)
__device__ __forceinline__ void scan_fit(const ScanPts &c, int n, float &range, float &bearing)
{
        float xm = 0.0f, ym = 0.0f;
        for (int i = 0; i < n; ++i)
        {
                xm += c.x(i);
                ym += c.y(i);
        }
        xm = xm / (float)n;
        ym = ym / (float)n;
        float zm = 0.0f;
        double M[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                        M[i][j] = 0.0;
        for (int i = 0; i < n; ++i)
        {
                const float xf = c.x(i) - xm, yf = c.y(i) - ym;
                const float zf = xf * xf + yf * yf;
                zm += zf;
                const double row[4] = {(double)zf, (double)xf, (double)yf, 1.0};
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int q = p; q < 4; ++q)
                                M[p][q] += row[p] * row[q];
        }
        zm = zm / (float)n;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < p; ++q)
                        M[p][q] = M[q][p];
        double V[4][4];
        jacobi4(M, V); // M -> diag(S^2)
        double sv[4];
        double smin = 1e300;
        int imin = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
                sv[i] = sqrt(fmax(M[i][i], 0.0));
                if (sv[i] < smin)
                {
                        smin = sv[i];
                        imin = i;
                }
        }
        double A[4];
        if (smin > 10e-12)
        {
                // Y = V S V^T, Yinv = V S^-1 V^T
                double Y[4][4], Yi[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                        {
                                double a = 0.0, b = 0.0;
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                {
                                        a += V[i][k] * sv[k] * V[j][k];
                                        b += V[i][k] / sv[k] * V[j][k];
                                }
                                Y[i][j] = a;
                                Yi[i][j] = b;
                        }
                // Q = Y Hinv Y with Hinv = [[0,0,0,1/2],[0,1,0,0],[0,0,1,0],[1/2,0,0,-2 zm]] (:240-241)
                const double h33 = (double)(-2.0f * zm);
                double T[4][4], Q[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                        T[i][0] = 0.5 * Y[i][3];
                        T[i][1] = Y[i][1];
                        T[i][2] = Y[i][2];
                        T[i][3] = 0.5 * Y[i][0] + h33 * Y[i][3];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                        {
                                double a = 0.0;
#pragma unroll
                                for (int k = 0; k < 4; ++k)
                                        a += T[i][k] * Y[k][j];
                                Q[i][j] = a;
                        }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = i + 1; j < 4; ++j)
                                Q[i][j] = Q[j][i] = 0.5 * (Q[i][j] + Q[j][i]);
                double W[4][4];
                jacobi4(Q, W);
                // smallest positive eigenvalue (:268-275; the reference walks them in ascending order, any order gives the same pick
                // unless none is positive and below 99999, in which case it keeps index 0 = the smallest eigenvalue)
                int sid = -1, lowest = 0;
                double sev = 99999.0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                        if (Q[i][i] > 0.0 && Q[i][i] < sev)
                        {
                                sid = i;
                                sev = Q[i][i];
                        }
                        if (Q[i][i] < Q[lowest][lowest])
                                lowest = i;
                }
                if (sid < 0)
                        sid = lowest;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                        double a = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                                a += Yi[i][k] * W[k][sid];
                        A[i] = a;
                }
        }
        else
        {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                        A[i] = V[i][imin]; // V.col(3): the right singular vector of the smallest singular value (:285-288)
        }
        const float a = (float)((-A[1]) / (2.0 * A[0]));
        const float b = (float)((-A[2]) / (2.0 * A[0]));
        const double cx = (double)(a + xm), cy = (double)(b + ym);
        range = (float)sqrt(cx * cx + cy * cy);
        bearing = (float)atan2(cy, cx);
}

/// grid (count), 64 threads: one wavefront per scan
__global__ __launch_bounds__(64) void scan_landmarks_kernel(const float *ranges, const float *cos_map, const float *sin_map, int64_t count,
                                                            int max_out, float *range_out, float *bearing_out, int32_t *n_out,
                                                            uint32_t *status_out)
{
        __shared__ float px[SCAN_BEAMS], py[SCAN_BEAMS];
        __shared__ unsigned long long linkm[6];
        __shared__ short ends[SCAN_BEAMS / 2 + 2];
        const int64_t sc = blockIdx.x;
        if (sc >= count)
                return;
        const int lane = threadIdx.x;
        const float *r = ranges + sc * SCAN_BEAMS;
        // bearing2pose, :135-142
        for (int t = lane; t < SCAN_BEAMS; t += 64)
        {
                const float d = r[t];
                px[t] = d * cos_map[t];
                py[t] = d * sin_map[t];
        }
        __syncthreads();
        // links between consecutive beams and ends of runs that stop before beam 359
        unsigned long long lm[6], em[6];
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
                const int t = lane + 64 * k;
                const bool link = (t >= 1 && t < SCAN_BEAMS) && eulerDistance(px[t - 1], py[t - 1], px[t], py[t]) < MIN_DIST_THRESH;
                lm[k] = __ballot(link);
        }
        if (lane == 0)
        {
#pragma unroll
                for (int k = 0; k < 6; ++k)
                        linkm[k] = lm[k];
        }
        __syncthreads();
        auto linked = [&](int t) -> bool { return (linkm[t >> 6] >> (t & 63)) & 1ull; };
        uint32_t status = 0;
        int nout = 0;
        if (eulerDistance(px[0], py[0], px[SCAN_BEAMS - 1], py[SCAN_BEAMS - 1]) < MIN_DIST_THRESH)
                status = 1u; // ASLAM_SCAN_REF_ABORT: the reference's assert(theta < 359), :69-81
        else
        {
                int nruns = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k)
                {
                        const int t = lane + 64 * k;
                        const bool isend = t >= 1 && t + 1 < SCAN_BEAMS && linked(t) && !linked(t + 1);
                        em[k] = __ballot(isend);
                        if (isend)
                                ends[nruns + __popcll(em[k] & ((1ull << lane) - 1ull))] = (short)t;
                        nruns += __popcll(em[k]);
                }
                __syncthreads();
                for (int j0 = 0; j0 < nruns; j0 += 64)
                {
                        const int j = j0 + lane;
                        bool valid = false;
                        float rg = 0.0f, bg = 0.0f;
                        if (j < nruns)
                        {
                                const int e = ends[j];
                                int s = e;
                                while (s - 1 >= 1 && linked(s - 1))
                                        --s;
                                const int s0 = (s == 1) ? 0 : s; // beam 0 opens the first cluster (:84-85)
                                const int n = e - s0 + 1;
                                ScanPts c = {px, py, s0};
                                if (n > SCAN_MIN_CLUSTER_POINTS && scan_classify(c, n))
                                {
                                        scan_fit(c, n, rg, bg);
                                        valid = true;
                                }
                        }
                        const unsigned long long vm = __ballot(valid);
                        if (valid)
                        {
                                const int o = nout + __popcll(vm & ((1ull << lane) - 1ull));
                                if (o < max_out)
                                {
                                        range_out[sc * max_out + o] = rg;
                                        bearing_out[sc * max_out + o] = bg;
                                }
                        }
                        nout += __popcll(vm);
                }
                if (nout > max_out)
                {
                        nout = max_out;
                        status |= 2u; // ASLAM_SCAN_OVERFLOW
                }
        }
        if (lane == 0)
        {
                n_out[sc] = nout;
                status_out[sc] = status;
        }
}
} // namespace aslam
