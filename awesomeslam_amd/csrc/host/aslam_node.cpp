// aslam_node.cpp -- host mirror of aslam::EKFSlam / aslam::UKFSlam (see aslam_node.h).
//
// Build: g++ -O2 -ffp-contract=off (the binary32 rounding points of the reference must not be fused),
// linked against libaslam_core.so.
#include "aslam_node.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace aslam
{
namespace
{
// include/awesome_slam/config.h:39-65
const float PI = 3.141592654;
const float MIN_DIST_THRESH = 0.5;
const uint32_t MIN_LANDMARK_OCC = 10;

struct WorldPoint
{
        float x, y; // structures.h:44-47: a Point built from two floats
};

/// LaserData::toPoint (structures.h:104-111) from the odometry pose held in Z(0..2)
WorldPoint project(const LaserData &d, const std::vector<double> &Z)
{
        const double heading = Z[2] + d.bearing;
        WorldPoint p;
        p.x = Z[0] + d.range * std::cos(heading);
        p.y = Z[1] + d.range * std::sin(heading);
        return p;
}

/// Point::distance -> eulerDistance (structures.h:69-73, tools.h:53-59)
float separation(const WorldPoint &a, const WorldPoint &b)
{
        const float dx = (double)a.x - (double)b.x;
        const float dy = (double)a.y - (double)b.y;
        return std::sqrt(dx * dx + dy * dy);
}

void check(int rc, const char *what)
{
        if (rc != ASLAM_OK)
                throw std::runtime_error(std::string(what) + ": " + aslam_last_error());
}
} // namespace

float normalizeAngle(float theta)
{
        float wrapped = std::fmod(theta, 2 * PI);
        if (wrapped > PI)
                wrapped = wrapped - 2 * PI;
        if (wrapped < -PI)
                wrapped = wrapped + 2 * PI;
        return wrapped;
}

float quat2euler(float w, float x, float y, float z)
{
        return std::atan2(2 * (w * z + x * y), 1 - 2 * (z * z + y * y));
}

FilterNode::FilterNode(int filter_, int max_landmark_count, int device, double now_init)
    : filter(filter_), MAX_LANDMARK_COUNT(max_landmark_count), ctx(nullptr), N(3), init_z(true), init_x(true),
      last_time((float)now_init), // ekf.cpp:54: last_time = ros::Time::now().toSec(), a float member (ekf.h:98)
      growth_refused(false), param_X(3, 0.0), param_Z(3, 0.0), a00(1.0), a10(0.0)
{
        aslam_config cfg = {};
        cfg.filter = filter;
        cfg.dtype = ASLAM_F64;
        cfg.max_landmark_count = max_landmark_count;
        cfg.batch = 1;
        cfg.max_obs = 1; // the stored sensor message and the wait-list stay on the host at this seam
        cfg.max_wait = 1;
        cfg.device = device;
        check(aslam_create(&cfg, &ctx), "aslam_create");
}

FilterNode::~FilterNode()
{
        aslam_destroy(ctx);
}

void FilterNode::cbSensorLandmark(const Landmarks &msg)
{
        init_z = false;
        sensor_landmark.resize(msg.x.size());
        for (size_t i = 0; i < msg.x.size(); ++i)
        {
                sensor_landmark[i].range = msg.x[i]; // double -> float, LaserData::assign(const float &, const float &)
                sensor_landmark[i].bearing = msg.y[i];
        }
}

bool FilterNode::cbOdom(const Odometry &msg, double now)
{
        if (init_z)
                return false;
        // ekf.cpp:80-81: last_time is a float member
        float delta_time = std::min(now - last_time, 1.0);
        last_time = now;
        return cbOdomDt(msg, delta_time);
}

bool FilterNode::cbOdomDt(const Odometry &msg, float delta_time)
{
        if (init_z)
                return false;
        updateZ(msg, delta_time);
        if (init_x)
        {
                init_x = false;
                param_X = param_Z;
                check(aslam_set_state(ctx, 0, (int)N, param_X.data(), nullptr, nullptr), "aslam_set_state");
        }
        slam(msg.vx, msg.wz, delta_time);
        return true;
}

Landmarks FilterNode::landmarks() const
{
        Landmarks out;
        for (uint32_t i = 0; i + 3 < N; i += 2)
        {
                out.x.push_back(param_X[3 + i]);
                out.y.push_back(param_X[4 + i]);
        }
        return out;
}

/// updateZandA (ekf.cpp:137-213) / updateZ (ukf.cpp:113-180)
void FilterNode::updateZ(const Odometry &msg, float delta_time)
{
        param_Z[0] = msg.px;
        param_Z[1] = msg.py;
        param_Z[2] = quat2euler(msg.qw, msg.qx, msg.qy, msg.qz);

        const uint32_t mapped = (N - 3) / 2;
        for (LaserData &obs : sensor_landmark)
        {
                obs.bearing = normalizeAngle(obs.bearing);
                bool associated = false;
                if (mapped > 0)
                {
                        const WorldPoint seen = project(obs, param_Z);
                        uint32_t best = 0;
                        float best_d = 0.0f;
                        for (uint32_t k = 0; k < mapped; ++k)
                        {
                                const WorldPoint known = {(float)param_X[3 + 2 * k], (float)param_X[4 + 2 * k]};
                                const float dk = separation(seen, known);
                                if (k == 0 || dk < best_d)
                                {
                                        best = k;
                                        best_d = dk;
                                }
                        }
                        if (best_d < MIN_DIST_THRESH)
                        {
                                param_Z[3 + 2 * best] = obs.range;
                                param_Z[4 + 2 * best] = obs.bearing;
                                associated = true;
                        }
                }
                if (!associated)
                        updateNewLandmarkWait(obs);
        }

        std::vector<LaserData> promoted;
        for (auto &entry : new_landmark_wait)
        {
                if (entry.second == MIN_LANDMARK_OCC)
                {
                        promoted.push_back(entry.first);
                        entry.second += 1;
                }
        }
        if (!promoted.empty())
                updateNewLandmark(promoted);

        if (filter == ASLAM_EKF && msg.vx && msg.wz)
        {
                const float delta_theta = msg.wz * delta_time;
                const float r = msg.vx / msg.wz;
                a00 = r * (-std::cos(param_Z[2]) + std::cos(param_Z[2] + delta_theta));
                a10 = r * (-std::sin(param_Z[2]) + std::sin(param_Z[2] + delta_theta));
        }
}

/// ekf.cpp:217-253 / ukf.cpp:184-220
void FilterNode::updateNewLandmarkWait(const LaserData &data)
{
        if (!new_landmark_wait.empty())
        {
                const WorldPoint seen = project(data, param_Z);
                size_t best = 0;
                float best_d = 0.0f;
                for (size_t i = 0; i < new_landmark_wait.size(); ++i)
                {
                        const float di = separation(seen, project(new_landmark_wait[i].first, param_Z));
                        if (i == 0 || di < best_d)
                        {
                                best = i;
                                best_d = di;
                        }
                }
                if (best_d < MIN_DIST_THRESH)
                {
                        new_landmark_wait[best].second++;
                        return;
                }
        }
        new_landmark_wait.push_back({data, 1});
}

/// ekf.cpp:255-290 / ukf.cpp:222-257: the vectors grow here, the matrices grow on the device (aslam_grow)
void FilterNode::updateNewLandmark(const std::vector<LaserData> &new_landmark)
{
        const uint32_t cacheN = N;
        const uint32_t grown = N + 2 * (uint32_t)new_landmark.size();
        if (grown >= (uint32_t)MAX_LANDMARK_COUNT)
        {
                growth_refused = true; // "[WARN] MAXIMUM LANDMARK COUNT IS SET TO ..." in the reference
                return;
        }
        N = grown;
        param_X.resize(N, 0.0);
        param_Z.resize(N, 0.0);
        for (size_t k = 0; k < new_landmark.size(); ++k)
        {
                const uint32_t i = cacheN + 2 * (uint32_t)k;
                param_Z[i] = new_landmark[k].range;
                param_Z[i + 1] = new_landmark[k].bearing;
                param_X[i] = param_Z[0] + param_Z[i] * std::cos(param_Z[2] + param_Z[i + 1]);
                param_X[i + 1] = param_Z[1] + param_Z[i] * std::sin(param_Z[2] + param_Z[i + 1]);
        }
        check(aslam_grow(ctx, 0, (int)N, &param_X[cacheN], &param_Z[cacheN]), "aslam_grow");
}

void FilterNode::slam(float vx, float az, float delta_time)
{
        if (filter == ASLAM_EKF)
                check(aslam_ekf_step(ctx, 0, vx, az, delta_time, param_Z.data(), a00, a10, param_X.data(), nullptr),
                      "aslam_ekf_step");
        else
                check(aslam_ukf_step(ctx, 0, vx, az, delta_time, param_Z.data(), param_X.data(), nullptr), "aslam_ukf_step");
}
} // namespace aslam

// ---- C shim -----------------------------------------------------------------------------------------------
struct aslam_node
{
        aslam::FilterNode *impl;
};

namespace
{
thread_local std::string g_node_err;
}

extern "C" {

aslam_node *aslam_node_create(int filter, int max_landmark_count, int device)
{
        return aslam_node_create_at(filter, max_landmark_count, device, 0.0);
}

aslam_node *aslam_node_create_at(int filter, int max_landmark_count, int device, double now_init)
{
        try
        {
                aslam_node *n = new aslam_node();
                n->impl = new aslam::FilterNode(filter, max_landmark_count, device, now_init);
                return n;
        }
        catch (const std::exception &e)
        {
                g_node_err = e.what();
                return nullptr;
        }
}

void aslam_node_destroy(aslam_node *n)
{
        if (n)
        {
                delete n->impl;
                delete n;
        }
}

const char *aslam_node_error(void)
{
        return g_node_err.c_str();
}

int aslam_node_sensor(aslam_node *n, int count, const double *x, const double *y)
{
        aslam::Landmarks m;
        m.x.assign(x, x + count);
        m.y.assign(y, y + count);
        n->impl->cbSensorLandmark(m);
        return 0;
}

static aslam::Odometry to_msg(const double v[8])
{
        aslam::Odometry m;
        m.px = v[0];
        m.py = v[1];
        m.qw = v[2];
        m.qx = v[3];
        m.qy = v[4];
        m.qz = v[5];
        m.vx = v[6];
        m.wz = v[7];
        return m;
}

int aslam_node_odom(aslam_node *n, const double msg[8], float delta_time)
{
        try
        {
                return n->impl->cbOdomDt(to_msg(msg), delta_time) ? 1 : 0;
        }
        catch (const std::exception &e)
        {
                g_node_err = e.what();
                return -1;
        }
}

int aslam_node_odom_now(aslam_node *n, const double msg[8], double now)
{
        try
        {
                return n->impl->cbOdom(to_msg(msg), now) ? 1 : 0;
        }
        catch (const std::exception &e)
        {
                g_node_err = e.what();
                return -1;
        }
}

int aslam_node_dim(const aslam_node *n)
{
        return (int)n->impl->dim();
}

int aslam_node_get(const aslam_node *n, double *X, double *Z, double *a00, double *a10)
{
        const uint32_t N = n->impl->dim();
        if (X)
                std::copy(n->impl->X().begin(), n->impl->X().begin() + N, X);
        if (Z)
                std::copy(n->impl->Z().begin(), n->impl->Z().begin() + N, Z);
        if (a00)
                *a00 = n->impl->A00();
        if (a10)
                *a10 = n->impl->A10();
        return 0;
}

int aslam_node_wait(const aslam_node *n, float *range, float *bearing, uint32_t *count, int cap)
{
        const auto &w = n->impl->waitList();
        const int k = std::min<int>(cap, (int)w.size());
        for (int i = 0; i < k; ++i)
        {
                range[i] = w[i].first.range;
                bearing[i] = w[i].first.bearing;
                count[i] = w[i].second;
        }
        return (int)w.size();
}

aslam_ctx *aslam_node_core(const aslam_node *n)
{
        return n->impl->core();
}

void aslam_host_narrow_odom(int64_t count, const double *odom, double *pose, float *yaw, double *twist)
{
        for (int64_t i = 0; i < count; ++i)
        {
                const double *m = odom + 8 * i;
                pose[2 * i] = m[0];
                pose[2 * i + 1] = m[1];
                yaw[i] = aslam::quat2euler(m[2], m[3], m[4], m[5]);
                twist[2 * i] = m[6];
                twist[2 * i + 1] = m[7];
        }
}

} // extern "C"
