// TEST INFRASTRUCTURE ONLY: sanitizer driver.  Reads a trace file (include/aslam_trace_file.h reader), drives every trajectory
// through (a) the C++ host mirror on the oracle-backed seam (core_over_oracle.cpp) and (b) the oracle's own replay, and checks
// that both agree: dimensions, Z and the wait-list bit for bit, X / P / pose stream to 1e-12.  Built with
// -fsanitize=address,undefined by tests/test_sanitize.py; any sanitizer report makes the run fail.
#include "../../awesomeslam_amd/csrc/host/aslam_node.h"
#include "../../include/aslam_trace_file.h"
#include "../../oracle/aslam_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

extern "C" int aslam_get_state(aslam_ctx *c, int traj, double *X, double *Z, double *P);

static double rel(const std::vector<double> &a, const std::vector<double> &b)
{
        double d = 0, s = 1e-300;
        for (size_t i = 0; i < a.size(); ++i)
        {
                d = std::fmax(d, std::fabs(a[i] - b[i]));
                s = std::fmax(s, std::fabs(b[i]));
        }
        return d / s;
}

int main(int argc, char **argv)
{
        if (argc < 4)
        {
                std::fprintf(stderr, "usage: san_driver <trace file> <ekf|ukf> <max_landmark_count>\n");
                return 2;
        }
        const int kind = std::strcmp(argv[2], "ukf") == 0 ? 1 : 0, cap = std::atoi(argv[3]);
        aslam_trace_file *tf = nullptr;
        if (aslam_trace_file_open(argv[1], &tf) != ASLAM_OK)
        {
                std::fprintf(stderr, "open: %s\n", aslam_trace_file_error());
                return 2;
        }
        // a truncated copy must be refused, not read past its end
        {
                aslam_trace_file *bad = nullptr;
                if (aslam_trace_file_open("/nonexistent/trace", &bad) == ASLAM_OK)
                        return 3;
        }
        int64_t B = 0, T = 0;
        int32_t mo = 0, L = 0;
        aslam_trace_file_dims(tf, &B, &T, &mo, &L);
        const double *odom;
        const float *dt;
        const uint8_t *obs_new;
        const int32_t *n_obs;
        const float *obs;
        aslam_trace_file_raw(tf, &odom, &dt, &obs_new, &n_obs, &obs);
        int rc = 0;
        for (int64_t b = 0; b < B; ++b)
        {
                aslam_node *node = aslam_node_create(kind, cap, 0);
                if (!node)
                {
                        std::fprintf(stderr, "node: %s\n", aslam_node_error());
                        return 2;
                }
                orc_filter *o = orc_create(kind, cap);
                std::vector<double> pn((size_t)T * 3, 0.0), po((size_t)T * 3, 0.0), obs64((size_t)T * mo * 2);
                std::vector<int32_t> dn(T), dorc(T);
                for (size_t i = 0; i < obs64.size(); ++i)
                        obs64[i] = (double)obs[(size_t)b * T * mo * 2 + i];
                for (int64_t t = 0; t < T; ++t)
                {
                        const size_t bt = (size_t)b * T + t;
                        if (obs_new[bt])
                        {
                                const int k = n_obs[bt];
                                std::vector<double> xs(k), ys(k);
                                for (int q = 0; q < k; ++q)
                                {
                                        xs[q] = obs64[((size_t)t * mo + q) * 2];
                                        ys[q] = obs64[((size_t)t * mo + q) * 2 + 1];
                                }
                                aslam_node_sensor(node, k, xs.data(), ys.data());
                        }
                        const int ran = aslam_node_odom(node, odom + bt * 8, dt[bt]);
                        if (ran < 0)
                        {
                                std::fprintf(stderr, "odom: %s\n", aslam_node_error());
                                return 2;
                        }
                        const int n = aslam_node_dim(node);
                        dn[t] = n;
                        if (ran)
                        {
                                std::vector<double> X(n), Z(n);
                                double a, c;
                                aslam_node_get(node, X.data(), Z.data(), &a, &c);
                                for (int q = 0; q < 3; ++q)
                                        pn[(size_t)t * 3 + q] = X[q];
                        }
                }
                orc_replay(o, T, odom + (size_t)b * T * 8, dt + (size_t)b * T, obs_new + (size_t)b * T, n_obs + (size_t)b * T, obs64.data(), mo,
                           po.data(), dorc.data());
                const int n = orc_dim(o);
                std::vector<double> Xo(n), Zo(n), Po((size_t)n * n), Xn(n), Zn(n), Pn((size_t)n * n);
                orc_get(o, Xo.data(), Zo.data(), Po.data());
                double a, c;
                aslam_node_get(node, Xn.data(), Zn.data(), &a, &c);
                aslam_get_state(aslam_node_core(node), 0, nullptr, nullptr, Pn.data());
                const bool dims_ok = dn == dorc && aslam_node_dim(node) == n;
                const bool z_ok = std::memcmp(Zn.data(), Zo.data(), sizeof(double) * n) == 0;
                const int wn = orc_wait_size(o);
                std::vector<float> wr(wn + 1), wb(wn + 1), wr2(wn + 1), wb2(wn + 1);
                std::vector<uint32_t> wc(wn + 1), wc2(wn + 1);
                orc_get_wait(o, wr.data(), wb.data(), wc.data());
                const int wn2 = aslam_node_wait(node, wr2.data(), wb2.data(), wc2.data(), wn + 1);
                const bool wait_ok = wn2 == wn && std::memcmp(wr.data(), wr2.data(), 4 * wn) == 0 && std::memcmp(wb.data(), wb2.data(), 4 * wn) == 0 &&
                                     std::memcmp(wc.data(), wc2.data(), 4 * wn) == 0;
                const double ex = rel(Xn, Xo), ep = rel(Pn, Po), epose = rel(pn, po);
                std::printf("trajectory %lld: N=%d dims %s Z %s wait-list(%d) %s rel err pose/X/P %.2e %.2e %.2e\n", (long long)b, n, dims_ok ? "exact" : "DIFFER",
                            z_ok ? "exact" : "DIFFER", wn, wait_ok ? "exact" : "DIFFER", epose, ex, ep);
                if (!(dims_ok && z_ok && wait_ok && ex < 1e-12 && ep < 1e-12 && epose < 1e-12))
                        rc = 1;
                orc_destroy(o);
                aslam_node_destroy(node);
        }
        aslam_trace_file_close(tf);
        return rc;
}
