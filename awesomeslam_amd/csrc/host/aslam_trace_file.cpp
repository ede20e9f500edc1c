// aslam_trace_file.cpp -- reader / writer of the recorded-input file format (include/aslam_trace_file.h).
#include "aslam_trace_file.h"
#include "aslam_node.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace
{
thread_local std::string g_err;

struct Header
{
        char magic[8];
        int64_t batch, T;
        int32_t max_obs, L, warmup, reserved[7];
};
static_assert(sizeof(Header) == 64, "64-byte header");
const char MAGIC[9] = "ASLTRC01";

size_t pad64(size_t o)
{
        return (o + 63) & ~(size_t)63;
}

int fail(const std::string &m)
{
        g_err = m;
        return ASLAM_ERR_ARG;
}
} // namespace

struct aslam_trace_file
{
        Header h;
        std::vector<unsigned char> bytes; // the whole file
        size_t o_odom, o_dt, o_new, o_nobs, o_obs, o_lm, o_truth;
        // narrowed view, built on first use
        std::vector<double> pose, twist;
        std::vector<float> yaw;
};

extern "C" {

const char *aslam_trace_file_error(void)
{
        return g_err.c_str();
}

int aslam_trace_file_open(const char *path, aslam_trace_file **out)
{
        if (!path || !out)
                return fail("null argument");
        FILE *fp = std::fopen(path, "rb");
        if (!fp)
                return fail(std::string("cannot open ") + path);
        std::fseek(fp, 0, SEEK_END);
        const long len = std::ftell(fp);
        std::fseek(fp, 0, SEEK_SET);
        aslam_trace_file *f = new aslam_trace_file();
        f->bytes.resize(len > 0 ? (size_t)len : 0);
        const size_t got = f->bytes.empty() ? 0 : std::fread(f->bytes.data(), 1, f->bytes.size(), fp);
        std::fclose(fp);
        if (got != f->bytes.size() || got < sizeof(Header))
        {
                delete f;
                return fail(std::string(path) + ": short read");
        }
        std::memcpy(&f->h, f->bytes.data(), sizeof(Header));
        const Header &h = f->h;
        if (std::memcmp(h.magic, MAGIC, 8) != 0 || h.batch <= 0 || h.T <= 0 || h.max_obs < 0 || h.L < 0)
        {
                delete f;
                return fail(std::string(path) + ": not an ASLTRC01 trace file");
        }
        const size_t BT = (size_t)h.batch * (size_t)h.T;
        size_t o = pad64(sizeof(Header));
        f->o_odom = o, o = pad64(o + BT * 8 * sizeof(double));
        f->o_dt = o, o = pad64(o + BT * sizeof(float));
        f->o_new = o, o = pad64(o + BT);
        f->o_nobs = o, o = pad64(o + BT * sizeof(int32_t));
        f->o_obs = o, o = o + BT * (size_t)h.max_obs * 2 * sizeof(float);
        if (h.L > 0)
        {
                o = pad64(o);
                f->o_lm = o, o = pad64(o + (size_t)h.batch * (size_t)h.L * 2 * sizeof(double));
                f->o_truth = o, o = o + BT * 3 * sizeof(double);
        }
        if (o > f->bytes.size())
        {
                delete f;
                return fail(std::string(path) + ": truncated");
        }
        const int32_t *nobs = reinterpret_cast<const int32_t *>(f->bytes.data() + f->o_nobs);
        for (size_t i = 0; i < BT; ++i)
                if (nobs[i] < 0 || nobs[i] > h.max_obs)
                {
                        delete f;
                        return fail(std::string(path) + ": n_obs out of range");
                }
        *out = f;
        return ASLAM_OK;
}

void aslam_trace_file_close(aslam_trace_file *f)
{
        delete f;
}

int aslam_trace_file_dims(const aslam_trace_file *f, int64_t *batch, int64_t *T, int32_t *max_obs, int32_t *landmarks)
{
        if (!f)
                return fail("null trace file");
        if (batch)
                *batch = f->h.batch;
        if (T)
                *T = f->h.T;
        if (max_obs)
                *max_obs = f->h.max_obs;
        if (landmarks)
                *landmarks = f->h.L;
        return ASLAM_OK;
}

int aslam_trace_file_raw(const aslam_trace_file *f, const double **odom, const float **dt, const uint8_t **obs_new,
                         const int32_t **n_obs, const float **obs)
{
        if (!f)
                return fail("null trace file");
        const unsigned char *b = f->bytes.data();
        if (odom)
                *odom = reinterpret_cast<const double *>(b + f->o_odom);
        if (dt)
                *dt = reinterpret_cast<const float *>(b + f->o_dt);
        if (obs_new)
                *obs_new = b + f->o_new;
        if (n_obs)
                *n_obs = reinterpret_cast<const int32_t *>(b + f->o_nobs);
        if (obs)
                *obs = reinterpret_cast<const float *>(b + f->o_obs);
        return ASLAM_OK;
}

int aslam_trace_file_view(aslam_trace_file *f, aslam_trace *view)
{
        if (!f || !view)
                return fail("null argument");
        const size_t BT = (size_t)f->h.batch * (size_t)f->h.T;
        const unsigned char *b = f->bytes.data();
        if (f->yaw.empty())
        {
                // what cbOdom / updateZandA read from the message (ekf.cpp:139-142), quat2euler in binary32 (tools.h:62-66)
                f->pose.resize(2 * BT);
                f->twist.resize(2 * BT);
                f->yaw.resize(BT);
                aslam_host_narrow_odom((int64_t)BT, reinterpret_cast<const double *>(b + f->o_odom), f->pose.data(), f->yaw.data(),
                                       f->twist.data());
        }
        view->T = f->h.T;
        view->max_obs = f->h.max_obs;
        view->is_device = 0;
        view->pose = f->pose.data();
        view->yaw = f->yaw.data();
        view->twist = f->twist.data();
        view->dt = reinterpret_cast<const float *>(b + f->o_dt);
        view->obs_new = b + f->o_new;
        view->n_obs = reinterpret_cast<const int32_t *>(b + f->o_nobs);
        view->obs = reinterpret_cast<const float *>(b + f->o_obs);
        return ASLAM_OK;
}

int aslam_trace_file_write(const char *path, int64_t batch, int64_t T, int32_t max_obs, int32_t warmup, const double *odom,
                           const float *dt, const uint8_t *obs_new, const int32_t *n_obs, const float *obs, int32_t landmarks,
                           const double *landmark_xy, const double *truth)
{
        if (!path || batch <= 0 || T <= 0 || max_obs < 0 || !odom || !dt || !obs_new || !n_obs || (max_obs > 0 && !obs))
                return fail("bad argument");
        if (landmarks < 0 || (landmarks > 0 && (!landmark_xy || !truth)))
                return fail("ground truth needs both landmark_xy and truth");
        FILE *fp = std::fopen(path, "wb");
        if (!fp)
                return fail(std::string("cannot create ") + path);
        Header h;
        std::memset(&h, 0, sizeof(h));
        std::memcpy(h.magic, MAGIC, 8);
        h.batch = batch, h.T = T, h.max_obs = max_obs, h.L = landmarks, h.warmup = warmup;
        size_t pos = 0;
        bool ok = true;
        auto put = [&](const void *p, size_t nbytes) {
                static const unsigned char zeros[64] = {0};
                const size_t padded = pad64(pos);
                if (padded > pos)
                        ok = ok && std::fwrite(zeros, 1, padded - pos, fp) == padded - pos;
                pos = padded;
                if (nbytes)
                        ok = ok && std::fwrite(p, 1, nbytes, fp) == nbytes;
                pos += nbytes;
        };
        const size_t BT = (size_t)batch * (size_t)T;
        put(&h, sizeof(h));
        put(odom, BT * 8 * sizeof(double));
        put(dt, BT * sizeof(float));
        put(obs_new, BT);
        put(n_obs, BT * sizeof(int32_t));
        put(obs, BT * (size_t)max_obs * 2 * sizeof(float));
        if (landmarks > 0)
        {
                put(landmark_xy, (size_t)batch * (size_t)landmarks * 2 * sizeof(double));
                put(truth, BT * 3 * sizeof(double));
        }
        ok = (std::fclose(fp) == 0) && ok;
        return ok ? ASLAM_OK : fail(std::string(path) + ": write failed");
}

} // extern "C"
