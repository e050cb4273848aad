// host_logic_test.cpp -- CPU-only checks of the host logic (no GPU, no libmedgp_hip):
//   scg            : the resumable scg_machine against a direct loop-structured statement of the same algorithm
//                    (ref: util/c_optimizer_scg.cpp:25-284) on analytic objectives; evaluation budget semantics
//   hyp  cfg out   : dump c_experiment::get_global_hyp (srand/rand draws) as doubles
//   data cfg PAN out: dump one patient's (meta, t, y) as loaded + z-scored
//   cfg  cfg       : print every parsed field of an exp_setup.json + hyp_bound.txt as "key value" lines (tests feed it the
//                    files the REFERENCE's own writers produced: tests/golden/ref_cfg, medgpc/util/config.py:5-66)
//   mode cfg fold alg out: dump get_test_kernel_param / get_test_mode_param (ref: dataio/c_experiment.cpp:179-219)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <atomic>
#include <thread>
#include <vector>
#include <unistd.h>

#include "medgp_experiment.hpp"
#include "medgp_optimizer.hpp"
#include "medgp_workpool.hpp"
#include "medgp_loader.hpp"
using namespace medgp;
typedef std::vector<double> vec;
typedef std::function<bool(const vec &, double &, vec &)> objective_t;

static double dotp(const vec &a, const vec &b) { double s = 0; for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i]; return s; }

// direct statement with the reference's loop structure (for cross-checking the state machine)
static void scg_direct(int max_iteration, const vec &init, const objective_t &obj, double &opt_loss, vec &X, int &nev) {
    const double INT = 0.1, EXT = 3.0, MAX = 20, RATIO = 10, SIG = 0.1, RHO = SIG / 2.0;
    const int sb = std::signbit((double)max_iteration) ? 1 : 0;
    int i = 0; nev = 0;
    double f0, d0, x1 = 0, x2 = 0, x3, x4 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, f1 = 0, f2 = 0, f3 = 0, f4 = 0, F0, M;
    vec df0, df3, dF0, X0, s;
    bool flag = obj(init, f0, df0); nev++;
    i += sb;
    s.resize(df0.size());
    for (size_t j = 0; j < s.size(); j++) s[j] = -df0[j];
    d0 = -dotp(s, s);
    x3 = 1.0 / (1.0 - d0);
    opt_loss = f0; X = init;
    while (i < std::abs(max_iteration)) {
        i += sb + (max_iteration > 0 ? 1 : 0);
        X0 = X; F0 = opt_loss; dF0 = df0;
        M = (max_iteration > 0) ? MAX : std::min((int)MAX, std::abs(max_iteration) - i);
        while (1) {
            x2 = 0; f2 = opt_loss; d2 = d0; f3 = opt_loss; df3 = df0;
            bool success = false;
            while (!success && M > 0) {
                M -= 1; i += sb;
                vec np(X.size());
                for (size_t j = 0; j < X.size(); j++) np[j] = X[j] + x3 * s[j];
                double ft; vec gt;
                flag = obj(np, ft, gt); nev++;
                if (flag) { f3 = ft; df3 = gt; }
                if (!flag || std::isinf(f3) || std::isnan(f3)) x3 = (x2 + x3) / 2.0; else success = true;
            }
            if (f3 < F0) { for (size_t j = 0; j < X.size(); j++) X0[j] = X[j] + x3 * s[j]; F0 = f3; dF0 = df3; }
            d3 = dotp(df3, s);
            if (d3 > SIG * d0 || f3 > opt_loss + x3 * RHO * d0 || M == 0) break;
            x1 = x2; f1 = f2; d1 = d2; x2 = x3; f2 = f3; d2 = d3;
            double A = 6.0 * (f1 - f2) + 3.0 * (d2 + d1) * (x2 - x1), B = 3.0 * (f2 - f1) - (2.0 * d1 + d2) * (x2 - x1);
            double temp = B * B - A * d1 * (x2 - x1);
            if (temp < 0) x3 = x2 * EXT;
            else {
                x3 = x1 - (d1 * std::pow(x2 - x1, 2.0) / (B + std::sqrt(temp)));
                if (std::isnan(x3) || std::isinf(x3) || x3 < 0) x3 = x2 * EXT;
                else if (x3 > x2 * EXT) x3 = x2 * EXT;
                else if (x3 < x2 + INT * (x2 - x1)) x3 = x2 + INT * (x2 - x1);
            }
        }
        while ((std::fabs(d3) > -SIG * d0 || f3 > opt_loss + x3 * RHO * d0) && M > 0) {
            if (d3 > 0 || f3 > opt_loss + x3 * RHO * d0) { x4 = x3; f4 = f3; d4 = d3; } else { x2 = x3; f2 = f3; d2 = d3; }
            if (f4 > opt_loss) {
                x3 = x2 - (0.5 * d2 * std::pow(x4 - x2, 2.0)) / (f4 - f2 - d2 * (x4 - x2));
                if (std::isnan(x3) || std::isinf(x3)) x3 = (x2 + x4) / 2.0;
            } else {
                double A = 6.0 * (f2 - f4) / (x4 - x2) + 3.0 * (d4 + d2), B = 3.0 * (f4 - f2) - (2.0 * d2 + d4) * (x4 - x2);
                double disc = B * B - A * d2 * std::pow(x4 - x2, 2.0);
                if (disc < 0) x3 = (x2 + x4) / 2.0;
                else { x3 = x2 + (std::sqrt(disc) - B) / A; if (std::isnan(x3) || std::isinf(x3)) x3 = (x2 + x4) / 2.0; }
            }
            x3 = std::max(std::min(x3, x4 - INT * (x4 - x2)), x2 + INT * (x4 - x2));
            vec np(X.size());
            for (size_t j = 0; j < X.size(); j++) np[j] = X[j] + x3 * s[j];
            double ft; vec gt;
            flag = obj(np, ft, gt); nev++;
            if (flag) { f3 = ft; df3 = gt; }
            if (flag && f3 < F0) { for (size_t j = 0; j < X.size(); j++) X0[j] = X[j] + x3 * s[j]; F0 = f3; dF0 = df3; }
            M -= 1; i += sb;
            d3 = dotp(df3, s);
        }
        if (flag && std::fabs(d3) < -SIG * d0 && f3 < opt_loss + x3 * RHO * d0) {
            for (size_t j = 0; j < X.size(); j++) X[j] += x3 * s[j];
            opt_loss = f3;
            double a = dotp(df3, df3), b = dotp(df3, df0), c = dotp(df0, df0);
            for (size_t j = 0; j < s.size(); j++) s[j] = ((a - b) / c) * s[j] - df3[j];
            df0 = df3; d3 = d0; d0 = dotp(df0, s);
            if (d0 > 0) { for (size_t j = 0; j < s.size(); j++) s[j] = -df0[j]; d0 = -dotp(s, s); }
            x3 = x3 * std::min(RATIO, d3 / (d0 - std::pow(2.0, -52)));
        } else {
            X = X0; opt_loss = F0; df0 = dF0;
            for (size_t j = 0; j < s.size(); j++) s[j] = -df0[j];
            d0 = -dotp(s, s);
            x3 = 1.0 / (1.0 - d0);
        }
    }
}

static bool rosen(const vec &x, double &f, vec &g) {
    f = 0; g.assign(x.size(), 0.0);
    for (size_t i = 0; i + 1 < x.size(); i++) {
        double a = x[i + 1] - x[i] * x[i], b = 1 - x[i];
        f += 100 * a * a + b * b;
        g[i] += -400 * a * x[i] - 2 * b;
        g[i + 1] += 200 * a;
    }
    return true;
}
static bool quad(const vec &x, double &f, vec &g) {
    f = 0; g.assign(x.size(), 0.0);
    for (size_t i = 0; i < x.size(); i++) { double w = 1.0 + 3.0 * i; f += 0.5 * w * (x[i] - 0.3 * i) * (x[i] - 0.3 * i); g[i] = w * (x[i] - 0.3 * i); }
    return true;
}
// objective with a failure region and a NaN region (exercises the bisection branch, ref :125-131)
static bool hole(const vec &x, double &f, vec &g) {
    if (x[0] > 2.5) return false;
    bool ok = quad(x, f, g);
    if (x[1] > 4.0) f = NAN;
    return ok;
}

static int run_machine(int budget, const vec &init, const objective_t &obj, double &loss, vec &X) {
    scg_machine m;
    m.start(budget, init);
    int n = 0;
    while (!m.done()) {
        vec th = m.request(), g;
        double f = 0;
        bool ok = obj(th, f, g);
        n++;
        m.feed(ok, f, g);
    }
    loss = m.opt_loss; X = m.opt_parameter;
    return n;
}

static int test_scg() {
    int bad = 0;
    struct Case { const char *name; objective_t obj; vec init; int budget; };
    std::vector<Case> cases = {
        {"quad/-100", quad, vec(6, 2.0), -100}, {"quad/-7", quad, vec(6, 2.0), -7}, {"quad/+5", quad, vec(6, 2.0), 5},
        {"quad/-1", quad, vec(6, 2.0), -1}, {"quad/-2", quad, vec(6, 2.0), -2},
        {"rosen/-100", rosen, {-1.2, 1.0, -0.5, 0.8}, -100}, {"rosen/-1000", rosen, {-1.2, 1.0, -0.5, 0.8}, -1000},
        {"rosen/-30", rosen, {-1.2, 1.0, -0.5, 0.8}, -30}, {"hole/-60", hole, {2.4, 3.9, 0.0}, -60},
    };
    for (auto &c : cases) {
        double l1, l2, f0; vec X1, X2, g0;
        int n2 = 0;
        int n1 = run_machine(c.budget, c.init, c.obj, l1, X1);
        scg_direct(c.budget, c.init, c.obj, l2, X2, n2);
        c.obj(c.init, f0, g0);
        bool same = (n1 == n2) && (l1 == l2) && (X1 == X2);
        bool budget_ok = c.budget > 0 || n1 <= -c.budget;
        bool descent = !(l1 > f0);
        printf("%-12s evals %4d/%4d loss %.12g (start %.6g)  same=%d budget=%d descent=%d\n", c.name, n1, n2, l1, f0, same, budget_ok, descent);
        if (!same || !budget_ok || !descent) bad++;
    }
    // convergence
    double l; vec X;
    run_machine(-1000, {-1.2, 1.0, -0.5, 0.8}, rosen, l, X);
    if (!(l < 1e-10)) { printf("rosenbrock did not converge: %g\n", l); bad++; }
    run_machine(-100, vec(6, 2.0), quad, l, X);
    if (!(l < 1e-12)) { printf("quadratic did not converge: %g\n", l); bad++; }
    printf(bad ? "SCG_FAIL\n" : "SCG_PASS\n");
    return bad;
}

// ---- optdump: the product's state machines on analytic objectives, results dumped for the Python restatements of the
//      reference's optimisers (oracle/optimizer_oracle.py; tests/test_optimizer_oracle.py compares bit for bit).
//      Objectives use + - * / sqrt log only, so C++ and Python evaluate them identically.
static bool poly(const vec &x, double &f, vec &g) {   // coupled quartic: not separable, several line-search regimes
    const size_t n = x.size();
    f = 0; g.assign(n, 0.0);
    double sum = 0;
    for (size_t i = 0; i < n; i++) sum += x[i];
    for (size_t i = 0; i < n; i++) {
        const double w = 1.0 + 0.5 * (double)i, c = 0.25 * (double)i - 1.0, d = x[i] - c;
        f += 0.5 * w * d * d + 0.05 * d * d * d * d;
        g[i] = w * d + 0.2 * d * d * d;
    }
    f += 0.05 * sum * sum;
    for (size_t i = 0; i < n; i++) g[i] += 0.1 * sum;
    return true;
}
static void dump_vec(FILE *f, const vec &v) { fwrite(v.data(), 8, v.size(), f); }
static int opt_dump(const char *path) {
    FILE *fo = fopen(path, "wb");
    if (!fo) return 1;
    // SCG cases: [n_eval, loss, X...]
    struct Case { int id; objective_t obj; vec init; int budget; };
    std::vector<Case> cases = {
        {0, quad, vec(6, 2.0), -100}, {0, quad, vec(6, 2.0), -7}, {0, quad, vec(6, 2.0), 5},
        {1, rosen, {-1.2, 1.0, -0.5, 0.8}, -100}, {1, rosen, {-1.2, 1.0, -0.5, 0.8}, -37},
        {2, hole, {2.4, 3.9, 0.0}, -60}, {3, poly, {1.5, -2.0, 0.7, 3.0, -0.4}, -80},
    };
    for (auto &c : cases) {
        double l; vec X;
        const int n = run_machine(c.budget, c.init, c.obj, l, X);
        vec head = {(double)c.id, (double)c.budget, (double)n, l};
        dump_vec(fo, head); dump_vec(fo, X);
    }
    // variational EM: Q = 2, D = 2, R = 2; theta = [lik (D) | A (QDR) | mu, v (2Q) | kappa (QD)]; objective = poly + the linked
    // Normal priors of the A entries as the reference's objective applies them (ref: inference/c_inference_prior.cpp:60-150,
    // prior/c_prior.cpp:383-421: lp = -(a - m)^2 / (2 v) - log(2 pi v) / 2, nlml -= lp, grad += (a - m) / v; clamp: grad = 0)
    for (int variant = 0; variant < 2; variant++) {
        const int Q = 2, D = 2, R = 2, nlik = D, H = D + Q * (D * R + 2 + D);
        c_prior prior(Q * (D * R + 2 + D), 0, D);
        prior.setup_param(7, {Q, D, R}, 2, {variant ? 0.3f : 0.01f, 0.01f});
        vec init(H);
        for (int h = 0; h < H; h++) init[h] = 0.3 * (double)((h * 7) % 5) - 0.6;
        if (variant) { init[nlik + 1] = 0.0; init[nlik + 6] = 0.0; }   // exactly-zero A entries: psi == 0 -> clamped (ref :151-154)
        varem_machine m;
        m.start(-6, init, &prior, {Q, D, R}, nlik, 15, false);
        int nev = 0;
        const double PI = 3.14159265;
        while (!m.done()) {
            vec th = m.request(), g;
            double f = 0;
            poly(th, f, g);
            if (variant) { g[nlik + 1] = 0.0; g[nlik + 6] = 0.0; }   // two A entries the data say nothing about: they stay exactly 0
            for (int a = 0; a < Q * D * R; a++) {
                const int h = nlik + a;
                if (!prior.flag_cov[a]) continue;
                if (prior.type_cov[a] == 0) { g[h] = 0.0; continue; }
                const double mean = (double)prior.fix_param_cov[a][0], var = (double)prior.fix_param_cov[a][1];
                const double lp = -1.0 * (th[h] - mean) * (th[h] - mean) / (2.0 * var) - std::log(2 * PI * var) / 2.0;
                f -= lp;
                g[h] -= -1.0 * (th[h] - mean) / var;
            }
            nev++;
            m.feed(true, f, g);
            (void)m.prior_changed();
        }
        vec head = {(double)(100 + variant), (double)nev, m.opt_loss};
        dump_vec(fo, head); dump_vec(fo, m.opt_parameter); dump_vec(fo, prior.get_cov_varEM_all());
        vec types;
        for (int a = 0; a < Q * D * R; a++) types.push_back((double)prior.type_cov[a]);
        dump_vec(fo, types);
    }
    fclose(fo);
    printf("OPTDUMP ok\n");
    return 0;
}

// WorkPool (medgp_workpool.hpp): every index runs exactly once, repeatedly, on any number of threads; an exception thrown on a
// worker thread is rethrown on the calling thread (never std::terminate), the pool stays usable afterwards.
static int test_pool() {
    for (int nt : {1, 2, 5, 8}) {
        medgp::WorkPool pool(nt);
        for (int rep = 0; rep < 50; rep++) {
            const int n = 1 + (rep * 37) % 301;
            std::vector<int> hits(n, 0);
            pool.parallel_for(n, [&](int i) { hits[i] += 1; });
            for (int i = 0; i < n; i++) if (hits[i] != 1) { printf("POOL_FAIL threads %d n %d index %d hit %d times\n", nt, n, i, hits[i]); return 1; }
        }
        bool caught = false;
        try {
            pool.parallel_for(200, [&](int i) { if (i == 137) throw std::runtime_error("boom at 137"); });
        } catch (const std::runtime_error &e) {
            caught = std::string(e.what()) == "boom at 137";
        }
        if (!caught) { printf("POOL_FAIL exception not propagated (threads %d)\n", nt); return 1; }
        std::vector<int> hits(64, 0);
        pool.parallel_for(64, [&](int i) { hits[i] += 1; });          // the pool survives a failed job
        for (int h : hits) if (h != 1) { printf("POOL_FAIL pool unusable after an exception\n"); return 1; }
    }
    printf("POOL_OK usable_cores %d\n", medgp::usable_cores());
    return 0;
}

// The read-ahead loader of medgp_train (medgp_loader.hpp): every ticket of the list is delivered exactly once whatever the number of
// readers, the capacity, the consumer's pace and the shape of the list (also lists shorter than the number of readers: its first
// version lost a reader's exit there and the trainer never ended); blocking takes wait for a full wave; a shared counter file splits
// one list between two loaders without loss or duplicate; a counter file that cannot be opened is reported.
static int test_loader(const char *tmpdir) {
    struct Item { size_t k; };
    for (int nthreads : {1, 2, 4, 8})
        for (size_t count : {(size_t)0, (size_t)1, (size_t)3, (size_t)64, (size_t)257})
            for (size_t cap : {(size_t)1, (size_t)2, (size_t)16, (size_t)400})
                for (int rep = 0; rep < 3; rep++) {
                    medgp::Tickets tk;
                    std::atomic<int> loads{0};
                    medgp::ReadAhead<Item> ra(count, tk, cap, nthreads, [&](size_t k) { loads++; if ((k + rep) % 7 == 0) std::this_thread::yield(); return std::unique_ptr<Item>(new Item{k}); });
                    std::vector<int> hit(count, 0);
                    size_t got = 0;
                    int spins = 0;
                    while (true) {
                        const bool block = (rep != 1);
                        auto v = ra.take(rep == 2 ? 5 : 1000, block);
                        for (auto &it : v) { if (it->k >= count) { printf("LOADER_FAIL ticket %zu beyond %zu\n", it->k, count); return 1; } hit[it->k]++; got++; }
                        if (v.empty()) {
                            if (ra.exhausted()) break;
                            if (!block) { std::this_thread::yield(); if (++spins > 50000000) { printf("LOADER_FAIL never exhausted (threads %d count %zu cap %zu)\n", nthreads, count, cap); return 1; } }
                        }
                        if (rep == 0 && got == count / 2) ra.set_cap(std::max<size_t>(1, cap / 8));
                    }
                    for (size_t k = 0; k < count; k++) if (hit[k] != 1) { printf("LOADER_FAIL ticket %zu delivered %d times (threads %d count %zu cap %zu rep %d)\n", k, hit[k], nthreads, count, cap, rep); return 1; }
                    if ((size_t)loads.load() != count || ra.taken() != (long long)count || ra.failed()) { printf("LOADER_FAIL loads %d taken %lld of %zu\n", loads.load(), ra.taken(), count); return 1; }
                }
    {   // a blocking take returns a FULL wave (min(want, cap)) or everything that is left
        medgp::Tickets tk;
        medgp::ReadAhead<Item> ra(100, tk, 32, 3, [&](size_t k) { return std::unique_ptr<Item>(new Item{k}); });
        size_t got = 0;
        bool first = true;
        while (!ra.exhausted()) {
            auto v = ra.take(48, true);
            if (first && v.size() != 32) { printf("LOADER_FAIL first blocking wave %zu != 32\n", v.size()); return 1; }
            first = false;
            got += v.size();
        }
        if (got != 100) { printf("LOADER_FAIL blocking waves delivered %zu of 100\n", got); return 1; }
    }
    {   // two loaders on ONE counter file
        const std::string q = std::string(tmpdir) + "/loader_queue.cnt";
        unlink(q.c_str());
        medgp::Tickets ta, tb;
        ta.path = q; tb.path = q;
        const size_t count = 500;
        std::vector<std::atomic<int>> hit(count);
        for (auto &h : hit) h.store(0);
        auto consume = [&](medgp::Tickets &tk, int nthreads) {
            medgp::ReadAhead<Item> ra(count, tk, 8, nthreads, [&](size_t k) { return std::unique_ptr<Item>(new Item{k}); });
            while (!ra.exhausted()) for (auto &it : ra.take(4, true)) hit[it->k]++;
            return !ra.failed();
        };
        bool oka = true, okb = true;
        std::thread t1([&] { oka = consume(ta, 3); }), t2([&] { okb = consume(tb, 2); });
        t1.join(); t2.join();
        for (size_t k = 0; k < count; k++) if (hit[k].load() != 1) { printf("LOADER_FAIL shared counter: ticket %zu delivered %d times\n", k, hit[k].load()); return 1; }
        if (!oka || !okb) { printf("LOADER_FAIL shared counter reported a failure\n"); return 1; }
        unlink(q.c_str());
    }
    {   // a counter file that cannot be opened
        medgp::Tickets tk;
        tk.path = std::string(tmpdir) + "/no/such/dir/q.cnt";
        medgp::ReadAhead<Item> ra(10, tk, 4, 2, [&](size_t k) { return std::unique_ptr<Item>(new Item{k}); });
        while (!ra.exhausted()) ra.take(4, true);
        if (!ra.failed() || ra.taken() != 0) { printf("LOADER_FAIL unreadable counter not reported\n"); return 1; }
    }
    printf("LOADER_OK\n");
    return 0;
}

// The host's c_prior (medgp_host.cpp: the mirror of the reference's prior/c_prior.cpp the trainer and the tester build their
// descriptors with) printed in the format of tests/golden/ref_prior.json.gz, which oracle/ref_prior_dump.cpp wrote from the
// REFERENCE's own compiled c_prior: same cases, same keys (tests/test_ref_prior.py compares them entry by entry).
static void pd_bools(const char *name, const std::vector<bool> &v) {
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) printf("%d%s", v[i] ? 1 : 0, i + 1 < v.size() ? "," : "");
    printf("], ");
}
static void pd_ints(const char *name, const std::vector<int> &v, bool last = false) {
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) printf("%d%s", v[i], i + 1 < v.size() ? "," : "");
    printf("]%s", last ? "" : ", ");
}
static void pd_dbls(const char *name, const std::vector<double> &v, bool last = false) {
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) printf("%.17g%s", v[i], i + 1 < v.size() ? "," : "");
    printf("]%s", last ? "" : ", ");
}
static void pd_fix(const char *name, const std::vector<std::vector<float>> &v) {
    std::vector<int> len;
    std::vector<double> p0, p1;
    for (const auto &e : v) { len.push_back((int)e.size()); p0.push_back(e.size() > 0 ? (double)e[0] : -1e30); p1.push_back(e.size() > 1 ? (double)e[1] : -1e30); }
    char nm[64];
    snprintf(nm, sizeof nm, "%s_len", name); pd_ints(nm, len);
    snprintf(nm, sizeof nm, "%s_p0", name); pd_dbls(nm, p0);
    snprintf(nm, sizeof nm, "%s_p1", name); pd_dbls(nm, p1);
}
static void pd_prior(const c_prior &p, const char *key, int ncov, bool last = false) {
    printf("\"%s\": {", key);
    pd_bools("flag_lik", p.flag_lik); pd_bools("flag_cov", p.flag_cov); pd_bools("flag_mean", p.flag_mean);
    pd_bools("exp_lik", p.exp_lik); pd_bools("exp_cov", p.exp_cov); pd_bools("exp_mean", p.exp_mean);
    pd_ints("type_lik", p.type_lik); pd_ints("type_cov", p.type_cov); pd_ints("type_mean", p.type_mean);
    pd_fix("fix_lik", p.fix_param_lik); pd_fix("fix_cov", p.fix_param_cov); pd_fix("fix_mean", p.fix_param_mean);
    pd_dbls("cov_varEM", p.get_cov_varEM_all());
    std::vector<double> fx;
    for (int i = 0; i < 5 && !p.get_cov_varEM_all().empty(); i++) fx.push_back(p.get_cov_varEM_fix_one(i));
    pd_dbls("cov_varEM_fix", fx, true);
    (void)ncov;
    printf("}%s", last ? "" : ", ");
}
static int prior_dump() {
    std::cout.rdbuf(std::cerr.rdbuf());   // (c_prior prints the reference's progress lines on cout)
    const int shapes[3][3] = {{5, 2, 2}, {5, 24, 8}, {5, 64, 8}};
    const float eta = 0.01f, beta_lam = 0.01f;
    printf("{\"shapes\": [");
    for (int s = 0; s < 3; s++) {
        const int Q = shapes[s][0], D = shapes[s][1], R = shapes[s][2], ncov = Q * (D * R + 2 + D), nlik = D;
        const std::vector<int> kp = {Q, D, R};
        const std::vector<float> pp = {eta, beta_lam};
        printf("{\"Q\": %d, \"D\": %d, \"R\": %d, ", Q, D, R);
        c_prior p0(ncov, 0, nlik); p0.setup_param(7, kp, 0, pp); pd_prior(p0, "mode0", ncov);
        c_prior p2(ncov, 0, nlik); p2.setup_param(7, kp, 2, pp); pd_prior(p2, "mode2", ncov);
        c_prior pd(ncov, 0, nlik); pd.setup_param(7, kp, 2, std::vector<float>()); pd_prior(pd, "mode2_default", ncov);
        c_prior pk(3, 0, 1); pk.setup_param(0, kp, 2, pp); pd_prior(pk, "mode2_kernel0", 3);
        std::vector<double> mode((size_t)(nlik + ncov));
        for (size_t i = 0; i < mode.size(); i++) mode[i] = 0.125 * (double)((int)(i % 7) - 3);
        c_prior pt(ncov, 0, nlik); pt.setup_param(7, kp, 2, pp); pt.init_test_prior(7, kp, mode); pd_prior(pt, "mode2_test", ncov);
        c_prior pt0(ncov, 0, nlik); pt0.setup_param(7, kp, 0, pp); pt0.init_test_prior(7, kp, mode); pd_prior(pt0, "mode0_test", ncov);
        std::vector<int> gf, gt;
        for (int i = 0; i < nlik + ncov; i++) { gf.push_back(pt.get_one_prior_flag(i) ? 1 : 0); gt.push_back(pt.get_one_prior_type(i)); }
        pd_ints("test_flag_theta_order", gf);
        pd_ints("test_type_theta_order", gt);
        // the flat theta-order arrays medgp_set_prior receives
        std::vector<uint8_t> f, e; std::vector<int32_t> t; std::vector<float> a, b;
        pt.flatten(f, t, e, a, b);
        std::vector<int> fi(f.begin(), f.end()), ei(e.begin(), e.end()), ti(t.begin(), t.end());
        std::vector<double> ad(a.begin(), a.end()), bd(b.begin(), b.end());
        pd_ints("flat_flag", fi); pd_ints("flat_type", ti); pd_ints("flat_exp", ei); pd_dbls("flat_p0", ad); pd_dbls("flat_p1", bd, true);
        printf("}%s", s < 2 ? ", " : "");
    }
    printf("]}\n");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "priordump")) return prior_dump();
    if (argc >= 3 && !strcmp(argv[1], "loader")) return test_loader(argv[2]);
    if (argc >= 2 && !strcmp(argv[1], "scg")) return test_scg();
    if (argc >= 2 && !strcmp(argv[1], "pool")) return test_pool();
    if (argc >= 3 && !strcmp(argv[1], "optdump")) return opt_dump(argv[2]);
    if (argc >= 4 && !strcmp(argv[1], "hyp")) {
        c_experiment e;
        if (!e.load(argv[2])) { printf("ERROR: %s\n", e.error().c_str()); return 1; }
        std::vector<vec> g;
        e.get_global_hyp(g);
        FILE *f = fopen(argv[3], "wb");
        for (auto &h : g) fwrite(h.data(), 8, h.size(), f);
        fclose(f);
        printf("HYP %d %d lik %d cov %d\n", (int)g.size(), e.get_hyp_num(), e.get_lik_num(), e.get_cov_num());
        return 0;
    }
    if (argc >= 5 && !strcmp(argv[1], "data")) {
        c_experiment e;
        if (!e.load(argv[2])) { printf("ERROR: %s\n", e.error().c_str()); return 1; }
        std::vector<int> m; std::vector<float> t, y;
        if (!e.get_one_patient_data(argv[3], m, t, y)) { printf("ERROR: %s\n", e.error().c_str()); return 1; }
        FILE *f = fopen(argv[4], "wb");
        int n = (int)t.size();
        fwrite(&n, 4, 1, f); fwrite(m.data(), 4, n, f); fwrite(t.data(), 4, n, f); fwrite(y.data(), 4, n, f);
        fclose(f);
        printf("DATA %d\n", n);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "cfg")) {
        c_experiment e;
        if (!e.load(argv[2])) { printf("ERROR: %s\n", e.error().c_str()); return 1; }
        const std::vector<int> kp = e.get_kernel_param(), fi = e.get_feature_index();
        const std::vector<float> ph = e.get_prior_hyp();
        printf("CFG kernel_index %d\nCFG Q %d\nCFG D %d\nCFG R %d\n", e.get_kernel_index(), kp[0], kp[1], kp[2]);
        printf("CFG prior_index %d\nCFG cv_fold_num %d\nCFG random_init_num %d\nCFG top_iteration_num %d\nCFG iteration_num_per_update %d\n",
               e.get_prior_mode(), e.get_cv_fold_num(), e.get_scg_init_num(), e.get_scg_max_iter_num(), e.get_prior_sub_opt_iter());
        printf("CFG online_learn_rate %.17g\nCFG online_momentum %.17g\n", e.get_online_learn_rate(), e.get_online_momentum());
        for (size_t i = 0; i < ph.size(); i++) printf("CFG prior_hyp%zu %.9g\n", i, (double)ph[i]);   // float, as GetFloat() narrows them
        printf("CFG feature_index");
        for (int f : fi) printf(" %d", f);
        printf("\nCFG H %d\nCFG lik %d\nCFG cov %d\n", e.get_hyp_num(), e.get_lik_num(), e.get_cov_num());
        printf("CFG train_dir %s\nCFG test_dir %s\nCFG kernel_dir %s\n", e.get_exp_train_dir().c_str(), e.get_exp_test_dir().c_str(), e.get_exp_kernel_dir().c_str());
        for (size_t i = 0; i < e.lb().size(); i++) printf("BOUND %zu %.17g %.17g\n", i, e.lb()[i], e.ub()[i]);
        return 0;
    }
    if (argc >= 6 && !strcmp(argv[1], "mode")) {
        c_experiment e;
        if (!e.load(argv[2])) { printf("ERROR: %s\n", e.error().c_str()); return 1; }
        std::vector<int> kp;
        vec mp;
        if (!e.get_test_kernel_param(atoi(argv[3]), argv[4], kp) || !e.get_test_mode_param(atoi(argv[3]), argv[4], mp)) { printf("ERROR: %s\n", e.error().c_str()); return 1; }
        FILE *f = fopen(argv[5], "wb");
        fwrite(mp.data(), 8, mp.size(), f);
        fclose(f);
        printf("MODE Q %d D %d R %d cov %d n %d\n", kp[0], kp[1], kp[2], e.get_test_cov_num(kp), (int)mp.size());
        return 0;
    }
    printf("usage: host_logic_test scg | hyp cfg out | data cfg PAN out | cfg cfg | mode cfg fold alg out\n");
    return 2;
}
