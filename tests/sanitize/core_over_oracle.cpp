// TEST INFRASTRUCTURE ONLY (tests/sanitize): the per-callback seam of include/aslam_core.h -- the seven entry points the C++
// host mirror (awesomeslam_amd/csrc/host/aslam_node.cpp) calls -- implemented over the CPU oracle, so that the host mirror's
// own logic (association, wait-list, promotion, growth: its restatement of ekf.cpp:137-290 / ukf.cpp:113-257) can run on a
// box without a GPU, under AddressSanitizer + UndefinedBehaviorSanitizer, and be compared with the oracle's own replay.
// Never linked into the product: libaslam_node.so links libaslam_core.so (HIP) and nothing else.
#include "../../include/aslam_core.h"
#include "../../oracle/aslam_oracle.h"

#include <string>
#include <vector>

struct aslam_ctx
{
        orc_filter *f;
        int cap;
        uint32_t status;
};

static std::string g_err;

extern "C" {
const char *aslam_last_error(void)
{
        return g_err.c_str();
}

int aslam_create(const aslam_config *cfg, aslam_ctx **out)
{
        if (!cfg || !out || cfg->batch != 1 || cfg->dtype != ASLAM_F64)
        {
                g_err = "core_over_oracle: one fp64 filter only";
                return ASLAM_ERR_ARG;
        }
        aslam_ctx *c = new aslam_ctx();
        c->f = orc_create(cfg->filter == ASLAM_UKF ? ORC_UKF : ORC_EKF, cfg->max_landmark_count);
        c->cap = cfg->max_landmark_count;
        c->status = 0;
        *out = c;
        return ASLAM_OK;
}

int aslam_destroy(aslam_ctx *c)
{
        if (c)
        {
                orc_destroy(c->f);
                delete c;
        }
        return ASLAM_OK;
}

int aslam_set_state(aslam_ctx *c, int traj, int n, const double *X, const double *Z, const double *P)
{
        if (traj != 0)
                return ASLAM_ERR_ARG;
        const int n0 = orc_dim(c->f);
        std::vector<double> x(std::max(n, n0)), z(std::max(n, n0)), p((size_t)std::max(n, n0) * std::max(n, n0));
        orc_get(c->f, x.data(), z.data(), p.data());
        double a00, a10;
        orc_get_A(c->f, &a00, &a10);
        if (n != n0 && !(X && Z && P))
                return ASLAM_ERR_ARG;
        if (X)
                x.assign(X, X + n);
        if (Z)
                z.assign(Z, Z + n);
        if (P)
                p.assign(P, P + (size_t)n * n);
        orc_set(c->f, n, x.data(), z.data(), p.data(), a00, a10);
        return ASLAM_OK;
}

/* the matrix part of updateNewLandmark (ekf.cpp:271-278): old block kept, new diagonal UKF_KP_LANDMARK_POSE = 1.0 */
int aslam_grow(aslam_ctx *c, int traj, int n_new, const double *x_seed, const double *z_seed)
{
        if (traj != 0 || !x_seed || !z_seed)
                return ASLAM_ERR_ARG;
        const int n0 = orc_dim(c->f);
        if (n_new <= n0 || ((n_new - n0) & 1))
                return ASLAM_ERR_ARG;
        if (n_new >= c->cap)
        {
                c->status |= ASLAM_ST_GROWTH_REFUSED;
                return ASLAM_OK;
        }
        std::vector<double> x(n0), z(n0), p((size_t)n0 * n0);
        orc_get(c->f, x.data(), z.data(), p.data());
        double a00, a10;
        orc_get_A(c->f, &a00, &a10);
        std::vector<double> x1(n_new), z1(n_new), p1((size_t)n_new * n_new, 0.0);
        for (int i = 0; i < n0; ++i)
        {
                x1[i] = x[i];
                z1[i] = z[i];
                for (int j = 0; j < n0; ++j)
                        p1[(size_t)i * n_new + j] = p[(size_t)i * n0 + j];
        }
        for (int i = n0; i < n_new; ++i)
        {
                x1[i] = x_seed[i - n0];
                z1[i] = z_seed[i - n0];
                p1[(size_t)i * n_new + i] = (double)1.0f;
        }
        orc_set(c->f, n_new, x1.data(), z1.data(), p1.data(), a00, a10);
        return ASLAM_OK;
}

static int step(aslam_ctx *c, float vx, float az, float dt, const double *Z, const double *a, double *X_out)
{
        const int n = orc_dim(c->f);
        std::vector<double> x(n), z(n), p((size_t)n * n);
        orc_get(c->f, x.data(), z.data(), p.data());
        double a00, a10;
        orc_get_A(c->f, &a00, &a10);
        if (a)
                a00 = a[0], a10 = a[1];
        orc_set(c->f, n, x.data(), Z, p.data(), a00, a10);
        orc_slam(c->f, vx, az, dt);
        if (X_out)
        {
                orc_get(c->f, x.data(), z.data(), p.data());
                for (int i = 0; i < n; ++i)
                        X_out[i] = x[i];
        }
        return ASLAM_OK;
}

int aslam_ekf_step(aslam_ctx *c, int traj, float vx, float az, float dt, const double *Z, double a00, double a10, double *X_out, void *)
{
        if (traj != 0 || !Z)
                return ASLAM_ERR_ARG;
        const double a[2] = {a00, a10};
        return step(c, vx, az, dt, Z, a, X_out);
}

int aslam_ukf_step(aslam_ctx *c, int traj, float vx, float az, float dt, const double *Z, double *X_out, void *)
{
        if (traj != 0 || !Z)
                return ASLAM_ERR_ARG;
        return step(c, vx, az, dt, Z, nullptr, X_out);
}

int aslam_get_state(aslam_ctx *c, int traj, double *X, double *Z, double *P)
{
        if (traj != 0)
                return ASLAM_ERR_ARG;
        const int n = orc_dim(c->f);
        std::vector<double> x(n), z(n), p((size_t)n * n);
        orc_get(c->f, x.data(), z.data(), p.data());
        for (int i = 0; i < n; ++i)
        {
                if (X)
                        X[i] = x[i];
                if (Z)
                        Z[i] = z[i];
        }
        if (P)
                for (size_t i = 0; i < (size_t)n * n; ++i)
                        P[i] = p[i];
        return ASLAM_OK;
}
}
