// medgp_test.cpp -- MI355X host of MedGP's online imputation test (the reference's main_one_test,
// ref: main_one_test.cpp:45-481), same CLI, inputs and output files:
//     medgp_test --cfg exp_setup.json --pan <PAN> --thread <n> --fold <k> --kernclust-alg <alg>
// Two passes as the reference (:137-141): "mean_wo_update" (mode kernel fixed) and "mean_w_update" (momentum
// updates of the hypers every > 5 minutes on the last 72 h of data, :289, :308-349).
//
// What changes is the schedule, not the arithmetic: HOT LOOP C (:269-444) re-factorises a growing subset of
// the patient for EVERY observation.  All (time stamp, observation) problems of a pass are independent once
// the hyper trajectory is known, so the trajectory is computed first (one gradient evaluation per update
// time, sequential as in the reference) and the problems are then solved in batches with
// medgp_fit_predict_batch (each problem = its own patient slot holding the training subset).
// Outputs (:447-472): test_<mode>_{feature,ci,flag}_<PAN>.txt, test_<mode>_{etime,error,pred}_<PAN>.bin.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>
#include <vector>

#include "medgp_experiment.hpp"
#include "medgp_host.hpp"

using namespace medgp;
using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace {

struct Problem {
    vector<int> train;     // indices into the patient's arrays
    int test_idx;          // the imputed observation
    int tt;                // time-stamp index (selects the hyper vector)
};

bool run_test_one(c_experiment &curr_exp, medgp_ctx *ctx, const string &PAN, int fold, bool flag_update,
                  const string &output_prefix, const string &alg, const vector<int> &test_kernel_param, int max_batch, bool per_problem) {
    cout << "running online imputation: " << (flag_update ? "with online updating" : "without online updating") << endl;
    cout << "testing patinet: " << PAN << " in cross-validation fold " << fold << endl;
    vector<int> meta_array;
    vector<float> time_array, value_array;
    if (!curr_exp.get_one_patient_data(PAN, meta_array, time_array, value_array)) { cout << "ERROR: " << curr_exp.error() << endl; return false; }
    const int n_all = (int)time_array.size();
    cout << "number of data points = " << n_all << endl;
    const int kidx = curr_exp.get_kernel_index();
    const int H = medgp_num_hyp(ctx);
    bool test_flag = true;
    if (n_all == 0) {
        cout << "Warning: no samples for testing" << endl;
        test_flag = false;
    } else {
        vector<float> uniq(time_array);
        std::sort(uniq.begin(), uniq.end());
        uniq.resize(std::distance(uniq.begin(), std::unique(uniq.begin(), uniq.end())));
        cout << "total # of unique time stamps: " << uniq.size() << endl;
        const double learn_rate = curr_exp.get_online_learn_rate(), momentum = curr_exp.get_online_momentum();
        vector<double> mode_parameter;
        if (!curr_exp.get_test_mode_param(fold, alg, mode_parameter)) { cout << "ERROR: " << curr_exp.error() << endl; return false; }
        if ((int)mode_parameter.size() != H) { cout << "ERROR: mode parameter file holds " << mode_parameter.size() << " values, expected " << H << endl; return false; }
        vector<double> best_parameter(mode_parameter), delta_parameter(mode_parameter.size(), 0.0);

        // test-time prior: clamp the A entries that are exactly zero in the mode kernel (ref c_prior.cpp:118-140)
        c_prior prior(curr_exp.get_test_cov_num(test_kernel_param), curr_exp.get_mean_num(), curr_exp.get_lik_num());
        prior.init_test_prior(kidx, test_kernel_param, mode_parameter);
        {
            vector<uint8_t> fl, ex; vector<int32_t> ty; vector<float> p0, p1;
            prior.flatten(fl, ty, ex, p0, p1);
            if (medgp_set_prior(ctx, -1, fl.data(), ty.data(), ex.data(), p0.data(), p1.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
        }

        // ---- pass 1: subsets per time stamp + hyper trajectory (sequential only where the reference is)
        vector<vector<double>> theta_version(1, best_parameter);
        vector<int> version_of_tt(uniq.size(), 0);
        vector<Problem> problems;
        vector<vector<int>> past_of_tt(uniq.size()), curr_of_tt(uniq.size());
        float last_update_time = uniq[0];
        for (int tt = 0; tt < (int)uniq.size(); tt++) {
            vector<int> &past = past_of_tt[tt], &curr = curr_of_tt[tt];
            for (int ii = 0; ii < n_all; ii++) {
                if (time_array[ii] < uniq[tt]) {
                    if (!flag_update || std::fabs(time_array[ii] - uniq[tt]) <= 72.0) past.push_back(ii);   // ref :287-300
                } else if (time_array[ii] == uniq[tt]) curr.push_back(ii);
            }
            if (flag_update && (tt > 3) && (uniq[tt] - last_update_time) > 5.0 / 60.0) {   // ref :308-349
                last_update_time = uniq[tt];
                bool obj_flag = false;
                double best_loss = 0.0;
                vector<double> best_grads(H, 0.0);
                if ((int)past.size() > 2) {   // c_objective_one's guard (ref util/c_objective_one.cpp:51)
                    vector<int> m; vector<float> t, y;
                    for (int ii : past) { m.push_back(meta_array[ii]); t.push_back(time_array[ii]); y.push_back(value_array[ii]); }
                    int32_t slot = 0, st = -1;
                    if (medgp_set_patient(ctx, 0, (int)t.size(), kidx == 7 ? (const int32_t *)m.data() : nullptr, t.data(), y.data()) ||
                        medgp_nlml_grad(ctx, 1, &slot, best_parameter.data(), 1, &best_loss, best_grads.data(), &st)) {
                        cout << "ERROR: " << medgp_last_error(ctx) << endl;
                        return false;
                    }
                    obj_flag = st >= 0;
                }
                if (obj_flag) {
                    for (int h = 0; h < H; h++) {
                        const bool prior_flag = prior.get_one_prior_flag(h);
                        const int prior_type = prior.get_one_prior_type(h);
                        if ((!prior_flag) | (prior_type != 0)) {
                            delta_parameter[h] = momentum * delta_parameter[h] + learn_rate * best_grads[h];
                            best_parameter[h] -= delta_parameter[h];
                        }
                    }
                } else {
                    cout << "Warning: failed to update at t[" << tt << "] = " << uniq[tt] << "; reset to mode parameters" << endl;
                    best_parameter = mode_parameter;
                    std::fill(delta_parameter.begin(), delta_parameter.end(), 0.0);
                }
                theta_version.push_back(best_parameter);
            }
            version_of_tt[tt] = (int)theta_version.size() - 1;
            for (int jj = 0; jj < (int)curr.size(); jj++) {
                Problem p;
                p.tt = tt;
                p.test_idx = curr[jj];
                p.train = past;
                for (int kk = 0; kk < (int)curr.size(); kk++)
                    if (kk != jj) p.train.push_back(curr[kk]);   // same-time observations of the other covariates (ref :358-365)
                problems.push_back(std::move(p));
            }
        }

        // ---- pass 2: all imputation problems
        const auto t_pass2 = std::chrono::steady_clock::now();
        const int np = (int)problems.size();
        vector<float> pmean(np, 0.f), pvar(np, 0.f);
        vector<int32_t> pstat(np, -2);   // -2: no training observations
        bool shared_done = false;
        if (!flag_update && !per_problem) {
            // Without online updating every problem uses the SAME hypers and its training set is `all observations before the
            // time stamp` + `the other observations AT the time stamp` (ref :287-300, :358-365).  With the observations in time
            // order that is a leading block of ONE Gram matrix plus the next rows: a single factorisation of the whole patient
            // (medgp_factor) holds every prefix factor (L[0:p,0:p]), every prefix solve (z[0:p]) and, in the rows of the c
            // observations of a time stamp, their conditional covariance given the past, G = L_cc L_cc^T, and their residuals
            // r = L_cc z_c.  Each imputation is then the Gaussian conditional of one component of N(y_c - r, G) given the
            // others: O(c^3) host work instead of one O(N^3) factorisation per observation.  Same mathematics as the
            // reference (train(false) + predict on each subset), different association; outputs are float either way.
            vector<int> order(n_all);
            for (int i = 0; i < n_all; i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return time_array[a] < time_array[b]; });
            vector<int32_t> sm(n_all); vector<float> st(n_all), sy(n_all);
            for (int i = 0; i < n_all; i++) { sm[i] = meta_array[order[i]]; st[i] = time_array[order[i]]; sy[i] = value_array[order[i]]; }
            vector<double> Lf((size_t)n_all * n_all), zf(n_all);
            int32_t fst = -1;
            if (medgp_set_patient(ctx, 0, n_all, kidx == 7 ? sm.data() : nullptr, st.data(), sy.data()) ||
                medgp_factor(ctx, 0, mode_parameter.data(), Lf.data(), zf.data(), &fst)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
            cout << "INFO: upload + shared factorisation " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pass2).count() << " ms" << endl;
            if (fst == 0) {   // a jittered or failed factorisation is not shared: the reference would jitter each subset on its own
                vector<int> pos_of(n_all);
                for (int i = 0; i < n_all; i++) pos_of[order[i]] = i;
                int k = 0;
                for (int tt = 0; tt < (int)uniq.size(); tt++) {
                    const vector<int> &curr = curr_of_tt[tt];
                    const int c = (int)curr.size(), p = c ? pos_of[curr[0]] : 0;   // stable sort: curr occupies positions p .. p+c-1 in order
                    vector<double> G((size_t)c * c, 0.0), r(c, 0.0);
                    for (int a = 0; a < c; a++) {
                        for (int b = 0; b <= a; b++) {
                            double s = 0.0;
                            for (int q = 0; q <= b; q++) s += Lf[(size_t)(p + a) * n_all + p + q] * Lf[(size_t)(p + b) * n_all + p + q];
                            G[(size_t)a * c + b] = G[(size_t)b * c + a] = s;
                        }
                        double s = 0.0;
                        for (int q = 0; q <= a; q++) s += Lf[(size_t)(p + a) * n_all + p + q] * zf[p + q];
                        r[a] = s;
                    }
                    for (int jj = 0; jj < c; jj++, k++) {
                        if (p == 0 && c == 1) continue;            // no training observations: stays -2
                        const int m1 = c - 1;
                        // S = G[sub,sub] = C C^T;  w = S^-1 r_sub,  u = S^-1 g,  g = G[sub,jj]
                        vector<double> C((size_t)m1 * m1, 0.0), g(m1), rs(m1);
                        vector<int> sub;
                        for (int a = 0; a < c; a++) if (a != jj) sub.push_back(a);
                        bool ok = true;
                        for (int a = 0; a < m1 && ok; a++) {
                            g[a] = G[(size_t)sub[a] * c + jj]; rs[a] = r[sub[a]];
                            for (int b = 0; b <= a; b++) {
                                double s = G[(size_t)sub[a] * c + sub[b]];
                                for (int q = 0; q < b; q++) s -= C[(size_t)a * m1 + q] * C[(size_t)b * m1 + q];
                                if (a == b) { if (!(s > 0.0)) { ok = false; break; } C[(size_t)a * m1 + a] = std::sqrt(s); }
                                else C[(size_t)a * m1 + b] = s / C[(size_t)b * m1 + b];
                            }
                        }
                        // a non-positive pivot of the small conditional factor: the reference would jitter THIS subset on its own
                        // (up to 10 noise additions, ref: inference/c_inference_exact.cpp:99-111) and still predict -- leave the
                        // problem to the per-problem path below instead of reporting a failure the reference would not see
                        if (!ok) { pstat[k] = -3; continue; }
                        // forward solves  C a1 = g,  C a2 = r_sub;  mean = (y - r_jj) + a1 . a2,  var = G_jj - a1 . a1
                        vector<double> a1(m1), a2(m1);
                        for (int a = 0; a < m1; a++) {
                            double s1 = g[a], s2 = rs[a];
                            for (int q = 0; q < a; q++) { s1 -= C[(size_t)a * m1 + q] * a1[q]; s2 -= C[(size_t)a * m1 + q] * a2[q]; }
                            a1[a] = s1 / C[(size_t)a * m1 + a]; a2[a] = s2 / C[(size_t)a * m1 + a];
                        }
                        double mean = (double)value_array[curr[jj]] - r[jj], var = G[(size_t)jj * c + jj];
                        for (int a = 0; a < m1; a++) { mean += a1[a] * a2[a]; var -= a1[a] * a1[a]; }
                        pmean[k] = (float)mean; pvar[k] = (float)var; pstat[k] = 0;
                    }
                }
                shared_done = true;
                cout << "finish testing " << np << "/" << np << " imputations (one shared factorisation)" << endl;
            }
        }
        // problems for the per-problem path: all of them, or -- after the shared pass -- the few it handed back (-3)
        vector<int> todo;
        for (int k = 0; k < np; k++) if (!shared_done || pstat[k] == -3) todo.push_back(k);
        for (int k : todo) if (pstat[k] == -3) pstat[k] = -2;
        const int ntodo = (int)todo.size();
        for (int c0 = 0; c0 < ntodo; c0 += max_batch) {
            vector<int32_t> slots, meta2, which;
            vector<float> t2;
            vector<double> thetas;
            // the training subsets of the whole batch are packed into ONE upload (medgp_set_patients): one H2D transfer and
            // no device wait per problem (one medgp_set_patient per problem was 8 copies + a stream sync each)
            vector<int32_t> pm; vector<float> pt, py;
            vector<int64_t> poff(1, 0);
            for (int kq = c0; kq < std::min(ntodo, c0 + max_batch); kq++) {
                const int k = todo[kq];
                const Problem &p = problems[k];
                if (p.train.empty()) continue;
                for (int ii : p.train) { pm.push_back(meta_array[ii]); pt.push_back(time_array[ii]); py.push_back(value_array[ii]); }
                poff.push_back((int64_t)pt.size());
                slots.push_back((int)slots.size());
                which.push_back(k);
                meta2.push_back(meta_array[p.test_idx]);
                t2.push_back(time_array[p.test_idx]);
                const vector<double> &th = theta_version[version_of_tt[p.tt]];
                thetas.insert(thetas.end(), th.begin(), th.end());
            }
            if (slots.empty()) continue;
            if (medgp_set_patients(ctx, (int)slots.size(), slots.data(), poff.data(), kidx == 7 ? pm.data() : nullptr, pt.data(), py.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
            vector<float> mean(slots.size()), var(slots.size());
            vector<int32_t> st(slots.size());
            if (medgp_fit_predict_batch(ctx, (int)slots.size(), slots.data(), thetas.data(), meta2.data(), t2.data(), mean.data(), var.data(), st.data())) {
                cout << "ERROR: " << medgp_last_error(ctx) << endl;
                return false;
            }
            for (size_t k = 0; k < slots.size(); k++) { pmean[which[k]] = mean[k]; pvar[which[k]] = var[k]; pstat[which[k]] = st[k]; }
            cout << "finish testing " << std::min(ntodo, c0 + max_batch) << "/" << ntodo << " imputations" << (shared_done ? " (handed back by the shared pass)" : "") << endl;
        }

        cout << "INFO: " << np << " imputations in " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_pass2).count()
             << " ms (" << (shared_done ? "one shared factorisation" : "one factorisation per imputed observation, batched") << ")" << endl;
        // ---- outputs in the reference's order (ref :376-444)
        vector<int> out_feature, out_ci;
        vector<double> out_etime, out_error, out_pred;
        for (int k = 0; k < np; k++) {
            const Problem &p = problems[k];
            const float obs = value_array[p.test_idx];
            const int tm = meta_array[p.test_idx];
            if (pstat[k] >= 0) {
                out_pred.push_back((double)pmean[k]);
                const double impute_error = pmean[k] - obs;   // float subtraction, then widened (ref :400)
                out_error.push_back(impute_error);
                out_ci.push_back(std::fabs(impute_error) <= 1.96 * std::sqrt(pvar[k]) ? 1 : 0);
            } else {
                if (pstat[k] == -2) cout << "Warning: no training observations; predict with zero mean" << endl;
                else cout << "Warning: failed to predict with current parameter" << endl;
                out_pred.push_back(0.0);
                const double impute_error = 0.0 - obs;
                out_error.push_back(impute_error);
                const double prior_var = std::exp(mode_parameter[tm]);   // ref :417-418, :431-432
                out_ci.push_back(std::fabs(impute_error) <= 1.96 * prior_var ? 1 : 0);
            }
            out_feature.push_back(curr_exp.get_feature_index()[tm]);
            out_etime.push_back(time_array[p.test_idx] - uniq[p.tt]);
        }
        if (!out_pred.empty()) {
            const string pre = curr_exp.get_exp_test_dir() + "test_" + output_prefix + "_";
            c_experiment::output_int_txt(pre + "feature_" + PAN, out_feature);
            c_experiment::output_double_bin(pre + "etime_" + PAN, out_etime);
            c_experiment::output_int_txt(pre + "ci_" + PAN, out_ci);
            c_experiment::output_double_bin(pre + "error_" + PAN, out_error);
            c_experiment::output_double_bin(pre + "pred_" + PAN, out_pred);
        }
    }
    c_experiment::output_int_txt(curr_exp.get_exp_test_dir() + "test_" + output_prefix + "_flag_" + PAN, {(int)test_flag});
    cout << "finish (" << output_prefix << ") testing individual PAN " << PAN << " w/ " << n_all << " samples; flag = " << test_flag << endl;
    return true;
}

}  // namespace

int main(int argc, const char *argv[]) {
    string exp_cfg, PAN, alg;
    int thread_num = 1, fold = 0, device = 0, max_batch = 0;
    bool per_problem = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--cfg") && i + 1 < argc) exp_cfg = argv[++i];
        else if (!strcmp(argv[i], "--pan") && i + 1 < argc) PAN = argv[++i];
        else if (!strcmp(argv[i], "--thread") && i + 1 < argc) thread_num = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fold") && i + 1 < argc) fold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--kernclust-alg") && i + 1 < argc) alg = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-batch") && i + 1 < argc) max_batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--per-problem")) per_problem = true;   // one factorisation per imputed observation in both passes (A/B)
        else { cout << "Error: unknown argument: " << argv[i] << endl; return 1; }
    }
    if (exp_cfg.empty() || PAN.empty() || alg.empty()) {
        cout << "usage:\n\t --cfg:\t the JSON configuration file\n\t --pan:\t ID of the testing patient\n\t --thread:\t accepted for compatibility\n"
             << "\t --fold:\t cross-validation fold of the patient\n\t --kernclust-alg:\t kernel clustering algorithm prefix of the mode files\n";
        return 1;
    }
    (void)thread_num;
    c_experiment curr_exp;
    if (!curr_exp.load(exp_cfg)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
    const int kidx = curr_exp.get_kernel_index();
    vector<int> test_kernel_param;
    if (!curr_exp.get_test_kernel_param(fold, alg, test_kernel_param)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
    cout << "# of mixture for testing: " << test_kernel_param[0] << endl;

    // size the device context from the patient (all of it can be one training subset)
    vector<int> m; vector<float> t, y;
    if (!curr_exp.get_one_patient_data(PAN, m, t, y)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
    const int n = std::max<int>(1, (int)t.size());
    const long long ldn = (n + 63) / 64 * 64;
    if (max_batch <= 0) max_batch = (int)std::max<long long>(1, std::min<long long>(256, (8LL << 30) / (32 * ldn * ldn)));
    medgp_ctx *ctx = nullptr;
    if (medgp_create(&ctx, device, kidx, test_kernel_param[0], test_kernel_param[1], test_kernel_param[2])) { cout << "ERROR: " << medgp_last_error(nullptr) << endl; return 1; }
    if (medgp_reserve(ctx, max_batch, n, max_batch)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }

    {   // first-use costs of the device (module load, first launches) are not part of either pass
        vector<double> th0(medgp_num_hyp(ctx), 0.0);
        int32_t s0 = 0, st0 = 0;
        double f0 = 0.0;
        if (!t.empty() && !medgp_set_patient(ctx, 0, (int)t.size(), kidx == 7 ? (const int32_t *)m.data() : nullptr, t.data(), y.data()))
            (void)medgp_nlml_grad(ctx, 1, &s0, th0.data(), 0, &f0, nullptr, &st0);
        vector<double> Lw((size_t)t.size() * t.size()), zw(t.size());
        if (!t.empty()) (void)medgp_factor(ctx, 0, th0.data(), Lw.data(), zw.data(), &st0);
        if (!t.empty()) {   // a full-size batch of the per-problem path: grows the staging / scratch buffers once
            const int nbw = max_batch, nw = (int)t.size();
            vector<int32_t> sl(nbw), pm, m2(nbw, m.empty() ? 0 : m[0]), stw(nbw);
            vector<float> pt, py, t2w(nbw, t[0]), mw(nbw), vw(nbw);
            vector<int64_t> off(1, 0);
            vector<double> thw;
            for (int b = 0; b < nbw; b++) {
                sl[b] = b;
                pm.insert(pm.end(), m.begin(), m.end()); pt.insert(pt.end(), t.begin(), t.end()); py.insert(py.end(), y.begin(), y.end());
                off.push_back((int64_t)pt.size());
                thw.insert(thw.end(), th0.begin(), th0.end());
            }
            (void)nw;
            if (!medgp_set_patients(ctx, nbw, sl.data(), off.data(), kidx == 7 ? pm.data() : nullptr, pt.data(), py.data()))
                (void)medgp_fit_predict_batch(ctx, nbw, sl.data(), thw.data(), m2.data(), t2w.data(), mw.data(), vw.data(), stw.data());
        }
    }
    time_t t1, t2;
    time(&t1);
    bool ok = run_test_one(curr_exp, ctx, PAN, fold, false, "mean_wo_update", alg, test_kernel_param, max_batch, per_problem) &&
              run_test_one(curr_exp, ctx, PAN, fold, true, "mean_w_update", alg, test_kernel_param, max_batch, per_problem);
    medgp_destroy(ctx);
    time(&t2);
    cout << "Finish all jobs. Total elapsed time = " << difftime(t2, t1) << " seconds" << endl;
    return ok ? 0 : 1;
}
