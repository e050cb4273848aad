// medgp_test.cpp -- MI355X host of MedGP's online imputation test (the reference's main_one_test,
// ref: main_one_test.cpp:45-481), same CLI, inputs and output files:
//     medgp_test --cfg exp_setup.json --pan <PAN> --thread <n> --fold <k> --kernclust-alg <alg>
// plus the cohort form the GPU build is for (the reference fans test patients out as one scheduler job each,
// ref: scripts/test_della.sh:46, medgpc/util/run_exp_generator.py:213-260):
//     medgp_test --cfg exp_setup.json --pan-list pans.txt --fold <k> --kernclust-alg <alg> [--device d] [--max-batch B]
// Two passes as the reference (:137-141): "mean_wo_update" (mode kernel fixed) and "mean_w_update" (momentum
// updates of the hypers every > 5 minutes on the last 72 h of data, :289, :308-349).
//
// What changes is the schedule, not the arithmetic.  HOT LOOP C (:269-444) re-factorises a growing subset of the patient
// for EVERY observation, one patient per process.  Here, per pass and for ALL patients of the list together:
//   1. the update times of a patient depend on its time stamps alone (:308), so every patient's list of update events is
//      known up front; the hyper TRAJECTORIES advance in lock step -- round r evaluates the r-th update of every patient
//      that has one: ONE packed upload of the 72-h windows (medgp_set_patients) and ONE batched medgp_nlml_grad per round
//      instead of a serial chain of single-entry calls per patient;
//   2. no-update pass: one shared factorisation per patient (every training subset is a leading block of the time-ordered
//      Gram matrix), all patients of a chunk in ONE medgp_factor_batch, the small Gaussian conditionals on the host threads;
//   3. update pass (changing hypers, 72-h windows): all (patient, time stamp, observation) problems of the cohort are packed
//      into batches of --max-batch through medgp_fit_predict_batch.
// The context is route-pinned (medgp_pin_route): a patient's results are bit-identical whatever the batch it was evaluated
// in, so a cohort run writes the same bytes as one run per patient (tests/test_test_host_gpu.py).
// Outputs (:447-472): test_<mode>_{feature,ci,flag}_<PAN>.txt, test_<mode>_{etime,error,pred}_<PAN>.bin.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "medgp_experiment.hpp"
#include "medgp_host.hpp"
#include "medgp_workpool.hpp"

using namespace medgp;
using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace {

struct Problem {
    int tt;                // time-stamp index (selects the training subset and the hyper vector)
    int jj;                // position of the imputed observation among the observations of that time stamp
    int test_idx;          // the imputed observation
};

struct TestPatient {
    string PAN;
    vector<int> meta;
    vector<float> t, y;
    int n_all = 0;
    string load_err;
    // ---- state of the current pass
    vector<float> uniq;
    vector<vector<int>> past_of_tt, curr_of_tt;
    vector<int> version_of_tt;
    vector<int> events;                    // time-stamp indices at which the hypers are updated (ref :308)
    vector<vector<double>> theta_version;  // [0] = mode kernel, [1 + e] = after update event e
    vector<double> best, delta;
    vector<Problem> problems;
    vector<float> pmean, pvar;
    vector<int32_t> pstat;                 // >= 0 ok (jitter count), -1 failed, -2 no training observations, -3 handed back
    bool shared_done = false;
    std::ostringstream log;

    // training subset of problem k: everything before the time stamp (72-h window in the update pass) + the same-time
    // observations of the other covariates (ref :287-300, :358-365)
    void train_of(const Problem &p, vector<int> &out) const {
        out = past_of_tt[p.tt];
        const vector<int> &curr = curr_of_tt[p.tt];
        for (int kk = 0; kk < (int)curr.size(); kk++) if (kk != p.jj) out.push_back(curr[kk]);
    }
};

double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// subsets per time stamp, update events, hyper-version index per time stamp, problem list: host work that depends on the
// patient's time stamps alone (ref :269-306, :352-365)
void prepare_patient(TestPatient &P, bool flag_update) {
    P.uniq.clear(); P.past_of_tt.clear(); P.curr_of_tt.clear(); P.version_of_tt.clear(); P.events.clear();
    P.theta_version.clear(); P.problems.clear(); P.shared_done = false;
    const int n_all = P.n_all;
    if (n_all == 0) return;
    P.uniq = P.t;
    std::sort(P.uniq.begin(), P.uniq.end());
    P.uniq.resize(std::distance(P.uniq.begin(), std::unique(P.uniq.begin(), P.uniq.end())));
    const int nu = (int)P.uniq.size();
    P.past_of_tt.assign(nu, {}); P.curr_of_tt.assign(nu, {}); P.version_of_tt.assign(nu, 0);
    float last_update_time = P.uniq[0];
    for (int tt = 0; tt < nu; tt++) {
        vector<int> &past = P.past_of_tt[tt], &curr = P.curr_of_tt[tt];
        for (int ii = 0; ii < n_all; ii++) {
            if (P.t[ii] < P.uniq[tt]) {
                if (!flag_update || std::fabs(P.t[ii] - P.uniq[tt]) <= 72.0) past.push_back(ii);   // ref :287-300
            } else if (P.t[ii] == P.uniq[tt]) curr.push_back(ii);
        }
        if (flag_update && (tt > 3) && (P.uniq[tt] - last_update_time) > 5.0 / 60.0) {   // ref :308-349
            last_update_time = P.uniq[tt];
            P.events.push_back(tt);
        }
        P.version_of_tt[tt] = (int)P.events.size();
        for (int jj = 0; jj < (int)curr.size(); jj++) P.problems.push_back({tt, jj, curr[jj]});
    }
    const int np = (int)P.problems.size();
    P.pmean.assign(np, 0.f); P.pvar.assign(np, 0.f); P.pstat.assign(np, -2);
}

// The c observations of one time stamp given the shared factor: each imputation is the Gaussian conditional of one component
// of N(y_c - r, G) given the others (G = L_cc L_cc^T the conditional covariance given the past, r = L_cc z_c).
void shared_conditionals(TestPatient &P, const vector<int> &order, const vector<double> &Lf, const vector<double> &zf) {
    const int n_all = P.n_all;
    vector<int> pos_of(n_all);
    for (int i = 0; i < n_all; i++) pos_of[order[i]] = i;
    int k = 0;
    for (int tt = 0; tt < (int)P.uniq.size(); tt++) {
        const vector<int> &curr = P.curr_of_tt[tt];
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
            // problem to the per-problem path instead of reporting a failure the reference would not see
            if (!ok) { P.pstat[k] = -3; continue; }
            // forward solves  C a1 = g,  C a2 = r_sub;  mean = (y - r_jj) + a1 . a2,  var = G_jj - a1 . a1
            vector<double> a1(m1), a2(m1);
            for (int a = 0; a < m1; a++) {
                double s1 = g[a], s2 = rs[a];
                for (int q = 0; q < a; q++) { s1 -= C[(size_t)a * m1 + q] * a1[q]; s2 -= C[(size_t)a * m1 + q] * a2[q]; }
                a1[a] = s1 / C[(size_t)a * m1 + a]; a2[a] = s2 / C[(size_t)a * m1 + a];
            }
            double mean = (double)P.y[curr[jj]] - r[jj], var = G[(size_t)jj * c + jj];
            for (int a = 0; a < m1; a++) { mean += a1[a] * a2[a]; var -= a1[a] * a1[a]; }
            P.pmean[k] = (float)mean; P.pvar[k] = (float)var; P.pstat[k] = 0;
        }
    }
    P.shared_done = true;
}

// one pass (with or without online updating) over ALL patients of the list
bool run_test_pass(c_experiment &curr_exp, medgp_ctx *ctx, vector<std::unique_ptr<TestPatient>> &pts, int fold, bool flag_update,
                   const string &output_prefix, const string &alg, const vector<int> &test_kernel_param, int max_batch,
                   bool per_problem, WorkPool &pool) {
    const bool cohort = pts.size() > 1;
    cout << "running online imputation: " << (flag_update ? "with online updating" : "without online updating") << endl;
    const int kidx = curr_exp.get_kernel_index();
    const int H = medgp_num_hyp(ctx);
    const double learn_rate = curr_exp.get_online_learn_rate(), momentum = curr_exp.get_online_momentum();
    vector<double> mode_parameter;
    bool any_data = false;
    for (auto &pp : pts) any_data = any_data || pp->n_all > 0;
    vector<uint8_t> may_update(H, 1);
    if (any_data) {
        if (!curr_exp.get_test_mode_param(fold, alg, mode_parameter)) { cout << "ERROR: " << curr_exp.error() << endl; return false; }
        if ((int)mode_parameter.size() != H) { cout << "ERROR: mode parameter file holds " << mode_parameter.size() << " values, expected " << H << endl; return false; }
        // test-time prior: clamp the A entries that are exactly zero in the mode kernel (ref c_prior.cpp:118-140); the same
        // mode kernel, hence the same prior, for every patient of the fold
        c_prior prior(curr_exp.get_test_cov_num(test_kernel_param), curr_exp.get_mean_num(), curr_exp.get_lik_num());
        prior.init_test_prior(kidx, test_kernel_param, mode_parameter);
        vector<uint8_t> fl, ex; vector<int32_t> ty; vector<float> p0, p1;
        prior.flatten(fl, ty, ex, p0, p1);
        if (medgp_set_prior(ctx, -1, fl.data(), ty.data(), ex.data(), p0.data(), p1.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
        for (int h = 0; h < H; h++) may_update[h] = ((!prior.get_one_prior_flag(h)) | (prior.get_one_prior_type(h) != 0)) ? 1 : 0;   // ref :331-337
    }

    // ---- per patient: subsets per time stamp, update events, problems (host threads)
    pool.parallel_for((int)pts.size(), [&](int i) {
        TestPatient &P = *pts[i];
        P.log.str("");
        P.log << "testing patinet: " << P.PAN << " in cross-validation fold " << fold << endl;
        P.log << "number of data points = " << P.n_all << endl;
        prepare_patient(P, flag_update);
        if (P.n_all == 0) P.log << "Warning: no samples for testing" << endl;
        else P.log << "total # of unique time stamps: " << P.uniq.size() << endl;
        P.best = mode_parameter;
        P.delta.assign(mode_parameter.size(), 0.0);
        if (P.n_all > 0) P.theta_version.assign(1, mode_parameter);
    });
    for (auto &pp : pts) cout << pp->log.str();

    // ---- hyper trajectories in lock step (update pass): round r = the r-th update event of every patient that has one
    const auto t_traj = std::chrono::steady_clock::now();
    long long n_updates = 0, n_rounds = 0, n_calls = 0;
    if (flag_update) {
        size_t max_events = 0;
        for (auto &pp : pts) max_events = std::max(max_events, pp->events.size());
        vector<int32_t> slots, pm;
        vector<float> pt, py;
        vector<int64_t> poff;
        vector<double> thetas, loss, grads;
        vector<int32_t> st;
        vector<TestPatient *> who;
        for (size_t r = 0; r < max_events; r++) {
            vector<TestPatient *> act;
            for (auto &pp : pts) if (pp->events.size() > r) act.push_back(pp.get());
            n_rounds++;
            for (size_t c0 = 0; c0 < act.size(); c0 += (size_t)max_batch) {
                const size_t c1 = std::min(act.size(), c0 + (size_t)max_batch);
                slots.clear(); pm.clear(); pt.clear(); py.clear(); poff.assign(1, 0); thetas.clear(); who.clear();
                for (size_t a = c0; a < c1; a++) {
                    TestPatient &P = *act[a];
                    const vector<int> &past = P.past_of_tt[P.events[r]];
                    if ((int)past.size() <= 2) continue;   // c_objective_one's guard (ref util/c_objective_one.cpp:51): no evaluation, reset below
                    for (int ii : past) { pm.push_back(P.meta[ii]); pt.push_back(P.t[ii]); py.push_back(P.y[ii]); }
                    poff.push_back((int64_t)pt.size());
                    slots.push_back((int32_t)slots.size());
                    thetas.insert(thetas.end(), P.best.begin(), P.best.end());
                    who.push_back(&P);
                }
                const int nb = (int)slots.size();
                loss.assign(std::max(nb, 1), 0.0); grads.assign((size_t)std::max(nb, 1) * H, 0.0); st.assign(std::max(nb, 1), -1);
                if (nb > 0) {
                    if (medgp_set_patients(ctx, nb, slots.data(), poff.data(), kidx == 7 ? pm.data() : nullptr, pt.data(), py.data()) ||
                        medgp_nlml_grad(ctx, nb, slots.data(), thetas.data(), 1, loss.data(), grads.data(), st.data())) {
                        cout << "ERROR: " << medgp_last_error(ctx) << endl;
                        return false;
                    }
                    n_calls++; n_updates += nb;
                }
                // momentum step (ref :326-349) -- or the reset to the mode kernel when the objective could not be evaluated
                int w = 0;
                for (size_t a = c0; a < c1; a++) {
                    TestPatient &P = *act[a];
                    const int tt = P.events[r];
                    const bool evaluated = w < nb && who[w] == &P;
                    const bool obj_flag = evaluated && st[w] >= 0;
                    if (obj_flag) {
                        const double *g = grads.data() + (size_t)w * H;
                        for (int h = 0; h < H; h++) {
                            if (may_update[h]) {
                                P.delta[h] = momentum * P.delta[h] + learn_rate * g[h];
                                P.best[h] -= P.delta[h];
                            }
                        }
                    } else {
                        cout << "Warning: failed to update at t[" << tt << "] = " << P.uniq[tt] << "; reset to mode parameters" << (cohort ? " (" + P.PAN + ")" : string()) << endl;
                        P.best = mode_parameter;
                        std::fill(P.delta.begin(), P.delta.end(), 0.0);
                    }
                    if (evaluated) w++;
                    P.theta_version.push_back(P.best);
                }
            }
        }
        cout << "INFO: hyper trajectories: " << n_updates << " gradient evaluations in " << n_calls << " batched calls (" << n_rounds
             << " lock-step rounds) " << ms_since(t_traj) << " ms" << endl;
    }

    // ---- all imputation problems
    const auto t_pass2 = std::chrono::steady_clock::now();
    long long np_total = 0;
    for (auto &pp : pts) np_total += (long long)pp->problems.size();
    if (!flag_update && !per_problem) {
        // Without online updating every problem of a patient uses the SAME hypers and its training set is `all observations before
        // the time stamp` + `the other observations AT the time stamp` (ref :287-300, :358-365).  With the observations in time
        // order that is a leading block of ONE Gram matrix plus the next rows: a single factorisation of the whole patient holds
        // every prefix factor (L[0:p,0:p]), every prefix solve (z[0:p]) and, in the rows of the c observations of a time stamp,
        // their conditional covariance given the past and their residuals.  O(c^3) host work per imputation instead of one O(N^3)
        // factorisation per observation.  Same mathematics as the reference (train(false) + predict on each subset), different
        // association; outputs are float either way.  All patients of a chunk share ONE medgp_factor_batch call.
        vector<TestPatient *> todo;
        for (auto &pp : pts) if (pp->n_all > 0) todo.push_back(pp.get());
        size_t a0 = 0;
        while (a0 < todo.size()) {
            size_t a1 = a0;
            size_t doubles = 0;
            while (a1 < todo.size() && (int)(a1 - a0) < max_batch) {
                const size_t need = (size_t)todo[a1]->n_all * todo[a1]->n_all;
                if (a1 > a0 && doubles + need > ((size_t)1 << 27)) break;   // <= 1 GB of exported factors at once
                doubles += need; a1++;
            }
            const int nb = (int)(a1 - a0);
            vector<vector<int>> order(nb);
            vector<int32_t> slots(nb), sm;
            vector<float> stt, sy;
            vector<int64_t> poff(1, 0);
            vector<double> thetas;
            vector<vector<double>> Lf(nb), zf(nb);
            vector<double *> Lp(nb), zp(nb);
            for (int b = 0; b < nb; b++) {
                TestPatient &P = *todo[a0 + b];
                order[b].resize(P.n_all);
                for (int i = 0; i < P.n_all; i++) order[b][i] = i;
                std::stable_sort(order[b].begin(), order[b].end(), [&](int x, int y2) { return P.t[x] < P.t[y2]; });
                for (int i = 0; i < P.n_all; i++) { sm.push_back(P.meta[order[b][i]]); stt.push_back(P.t[order[b][i]]); sy.push_back(P.y[order[b][i]]); }
                poff.push_back((int64_t)stt.size());
                slots[b] = b;
                thetas.insert(thetas.end(), mode_parameter.begin(), mode_parameter.end());
                Lf[b].assign((size_t)P.n_all * P.n_all, 0.0); zf[b].assign(P.n_all, 0.0);
                Lp[b] = Lf[b].data(); zp[b] = zf[b].data();
            }
            vector<int32_t> fst(nb, -1);
            if (medgp_set_patients(ctx, nb, slots.data(), poff.data(), kidx == 7 ? sm.data() : nullptr, stt.data(), sy.data()) ||
                medgp_factor_batch(ctx, nb, slots.data(), thetas.data(), Lp.data(), zp.data(), fst.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
            // a jittered or failed factorisation is not shared: the reference would jitter each subset on its own
            pool.parallel_for(nb, [&](int b) { if (fst[b] == 0) shared_conditionals(*todo[a0 + b], order[b], Lf[b], zf[b]); });
            a0 = a1;
        }
        cout << "INFO: upload + shared factorisation of " << todo.size() << " patient(s) " << ms_since(t_pass2) << " ms" << endl;
        if (!cohort && !todo.empty() && todo[0]->shared_done)
            cout << "finish testing " << todo[0]->problems.size() << "/" << todo[0]->problems.size() << " imputations (one shared factorisation)" << endl;
    }
    // problems for the per-problem path: all of them, or -- after the shared pass -- the few it handed back (-3)
    vector<std::pair<TestPatient *, int>> todo;
    bool any_shared = false;
    for (auto &pp : pts) {
        TestPatient &P = *pp;
        any_shared = any_shared || P.shared_done;
        for (int k = 0; k < (int)P.problems.size(); k++)
            if (!P.shared_done || P.pstat[k] == -3) { P.pstat[k] = -2; todo.push_back({&P, k}); }
    }
    const long long ntodo = (long long)todo.size();
    {
        vector<int32_t> slots, meta2, pm, st;
        vector<std::pair<TestPatient *, int>> which;
        vector<float> t2, pt, py, mean, var;
        vector<double> thetas;
        vector<int64_t> poff;
        vector<int> train;
        for (long long c0 = 0; c0 < ntodo; c0 += max_batch) {
            slots.clear(); meta2.clear(); which.clear(); t2.clear(); thetas.clear(); pm.clear(); pt.clear(); py.clear(); poff.assign(1, 0);
            // the training subsets of the whole batch are packed into ONE upload (medgp_set_patients): one H2D transfer and
            // no device wait per problem
            for (long long kq = c0; kq < std::min<long long>(ntodo, c0 + max_batch); kq++) {
                TestPatient &P = *todo[kq].first;
                const Problem &p = P.problems[todo[kq].second];
                P.train_of(p, train);
                if (train.empty()) continue;
                for (int ii : train) { pm.push_back(P.meta[ii]); pt.push_back(P.t[ii]); py.push_back(P.y[ii]); }
                poff.push_back((int64_t)pt.size());
                slots.push_back((int)slots.size());
                which.push_back(todo[kq]);
                meta2.push_back(P.meta[p.test_idx]);
                t2.push_back(P.t[p.test_idx]);
                const vector<double> &th = P.theta_version[P.version_of_tt[p.tt]];
                thetas.insert(thetas.end(), th.begin(), th.end());
            }
            if (slots.empty()) continue;
            if (medgp_set_patients(ctx, (int)slots.size(), slots.data(), poff.data(), kidx == 7 ? pm.data() : nullptr, pt.data(), py.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
            mean.assign(slots.size(), 0.f); var.assign(slots.size(), 0.f); st.assign(slots.size(), -1);
            if (medgp_fit_predict_batch(ctx, (int)slots.size(), slots.data(), thetas.data(), meta2.data(), t2.data(), mean.data(), var.data(), st.data())) {
                cout << "ERROR: " << medgp_last_error(ctx) << endl;
                return false;
            }
            for (size_t k = 0; k < slots.size(); k++) {
                TestPatient &P = *which[k].first;
                P.pmean[which[k].second] = mean[k]; P.pvar[which[k].second] = var[k]; P.pstat[which[k].second] = st[k];
            }
            cout << "finish testing " << std::min<long long>(ntodo, c0 + max_batch) << "/" << ntodo << " imputations" << (any_shared ? " (handed back by the shared pass)" : "") << endl;
        }
    }
    cout << "INFO: " << np_total << " imputations in " << ms_since(t_pass2) << " ms ("
         << (any_shared ? "one shared factorisation per patient" : "one factorisation per imputed observation, batched") << ")" << endl;

    // ---- outputs in the reference's order (ref :376-444), patient by patient
    for (auto &pp : pts) {
        TestPatient &P = *pp;
        const bool test_flag = P.n_all > 0;
        const int np = (int)P.problems.size();
        vector<int> out_feature, out_ci;
        vector<double> out_etime, out_error, out_pred;
        for (int k = 0; k < np; k++) {
            const Problem &p = P.problems[k];
            const float obs = P.y[p.test_idx];
            const int tm = P.meta[p.test_idx];
            if (P.pstat[k] >= 0) {
                out_pred.push_back((double)P.pmean[k]);
                const double impute_error = P.pmean[k] - obs;   // float subtraction, then widened (ref :400)
                out_error.push_back(impute_error);
                out_ci.push_back(std::fabs(impute_error) <= 1.96 * std::sqrt(P.pvar[k]) ? 1 : 0);
            } else {
                if (P.pstat[k] == -2) cout << "Warning: no training observations; predict with zero mean" << endl;
                else cout << "Warning: failed to predict with current parameter" << endl;
                out_pred.push_back(0.0);
                const double impute_error = 0.0 - obs;
                out_error.push_back(impute_error);
                const double prior_var = std::exp(mode_parameter[tm]);   // ref :417-418, :431-432
                out_ci.push_back(std::fabs(impute_error) <= 1.96 * prior_var ? 1 : 0);
            }
            out_feature.push_back(curr_exp.get_feature_index()[tm]);
            out_etime.push_back(P.t[p.test_idx] - P.uniq[p.tt]);
        }
        if (!out_pred.empty()) {
            const string pre = curr_exp.get_exp_test_dir() + "test_" + output_prefix + "_";
            c_experiment::output_int_txt(pre + "feature_" + P.PAN, out_feature);
            c_experiment::output_double_bin(pre + "etime_" + P.PAN, out_etime);
            c_experiment::output_int_txt(pre + "ci_" + P.PAN, out_ci);
            c_experiment::output_double_bin(pre + "error_" + P.PAN, out_error);
            c_experiment::output_double_bin(pre + "pred_" + P.PAN, out_pred);
        }
        c_experiment::output_int_txt(curr_exp.get_exp_test_dir() + "test_" + output_prefix + "_flag_" + P.PAN, {(int)test_flag});
        cout << "finish (" << output_prefix << ") testing individual PAN " << P.PAN << " w/ " << P.n_all << " samples; flag = " << test_flag << endl;
    }
    return true;
}

}  // namespace

int main(int argc, const char *argv[]) {
    string exp_cfg, pan_arg, pan_list, alg;
    int thread_num = 1, fold = 0, device = 0, max_batch = 0, host_threads = 0;
    bool per_problem = false, pin_route = true;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--cfg") && i + 1 < argc) exp_cfg = argv[++i];
        else if (!strcmp(argv[i], "--pan") && i + 1 < argc) pan_arg = argv[++i];
        else if (!strcmp(argv[i], "--pan-list") && i + 1 < argc) pan_list = argv[++i];
        else if (!strcmp(argv[i], "--thread") && i + 1 < argc) thread_num = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--fold") && i + 1 < argc) fold = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--kernclust-alg") && i + 1 < argc) alg = argv[++i];
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-batch") && i + 1 < argc) max_batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--host-threads") && i + 1 < argc) host_threads = atoi(argv[++i]);   // 0 = usable cores (<= 8)
        else if (!strcmp(argv[i], "--per-problem")) per_problem = true;   // one factorisation per imputed observation in both passes (A/B)
        else if (!strcmp(argv[i], "--auto-route")) pin_route = false;     // fastest schedule per call; last bits may then depend on the batch
        else { cout << "Error: unknown argument: " << argv[i] << endl; return 1; }
    }
    if (exp_cfg.empty() || (pan_arg.empty() && pan_list.empty()) || alg.empty()) {
        cout << "usage:\n\t --cfg:\t the JSON configuration file\n\t --pan:\t ID of the testing patient (comma separated for several)\n"
             << "\t --pan-list:\t file with one patient ID per line (cohort mode; all patients of one cross-validation fold)\n"
             << "\t --thread:\t accepted for compatibility\n"
             << "\t --fold:\t cross-validation fold of the patient(s)\n\t --kernclust-alg:\t kernel clustering algorithm prefix of the mode files\n";
        return 1;
    }
    (void)thread_num;
    try {
        c_experiment curr_exp;
        if (!curr_exp.load(exp_cfg)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
        const int kidx = curr_exp.get_kernel_index();
        vector<int> test_kernel_param;
        if (!curr_exp.get_test_kernel_param(fold, alg, test_kernel_param)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
        cout << "# of mixture for testing: " << test_kernel_param[0] << endl;

        vector<string> pans;
        {
            size_t a = 0;
            while (a < pan_arg.size()) { size_t b = pan_arg.find(',', a); if (b == string::npos) b = pan_arg.size(); if (b > a) pans.push_back(pan_arg.substr(a, b - a)); a = b + 1; }
            if (!pan_list.empty()) {
                std::ifstream f(pan_list.c_str());
                if (!f) { cout << "ERROR: File " << pan_list << " could not be opened." << endl; return 1; }
                string s;
                while (f >> s) pans.push_back(s);
            }
        }
        if (pans.empty()) { cout << "ERROR: no patient given" << endl; return 1; }

        // ---- load the patients on the host threads (the reference loads ONE patient per process, ref: dataio/c_experiment.cpp:254-309)
        if (host_threads <= 0) host_threads = std::min(8, usable_cores());
        WorkPool pool(std::max(1, host_threads));
        const auto t_load0 = std::chrono::steady_clock::now();
        vector<std::unique_ptr<TestPatient>> pts(pans.size());
        pool.parallel_for((int)pans.size(), [&](int i) {
            c_experiment ex = curr_exp;                 // own error string per task
            std::unique_ptr<TestPatient> p(new TestPatient());
            p->PAN = pans[i];
            if (!ex.get_one_patient_data(p->PAN, p->meta, p->t, p->y)) p->load_err = ex.error();
            p->n_all = (int)p->t.size();
            pts[i] = std::move(p);
        });
        // A patient that cannot be read is reported and left out -- the reference runs one process per patient, so the others would
        // still get their test_* files (ref: medgpc/util/run_exp_generator.py:213-260) -- and the run ends with a non-zero exit code.
        int n_max = 1, n_unreadable = 0;
        long long n_problems = 0;
        {
            vector<std::unique_ptr<TestPatient>> keep;
            for (auto &pp : pts) {
                if (!pp->load_err.empty()) { cout << "ERROR: " << pp->load_err << " (patient " << pp->PAN << " skipped)" << endl; n_unreadable++; continue; }
                n_max = std::max(n_max, pp->n_all);
                n_problems += pp->n_all;
                keep.push_back(std::move(pp));
            }
            pts.swap(keep);
        }
        if (pts.empty()) { cout << "ERROR: no readable patient" << endl; return 1; }
        if (pts.size() > 1)
            cout << "INFO: loaded " << pts.size() << " patients x " << curr_exp.get_feature_index().size() << " feature files in "
                 << ms_since(t_load0) << " ms on " << pool.size() << " host threads" << endl;

        // ---- device context sized from the largest patient (all of it can be one training subset)
        const long long ldn = (n_max + 63) / 64 * 64;
        if (max_batch <= 0) max_batch = (int)std::max<long long>(1, std::min<long long>(pts.size() > 1 ? 1024 : 256, (8LL << 30) / (32 * ldn * ldn)));
        medgp_ctx *ctx = nullptr;
        if (medgp_create(&ctx, device, kidx, test_kernel_param[0], test_kernel_param[1], test_kernel_param[2])) { cout << "ERROR: " << medgp_last_error(nullptr) << endl; return 1; }
        if (medgp_reserve(ctx, max_batch, n_max, max_batch)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
        if (medgp_pin_route(ctx, pin_route ? 1 : 0)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }

        time_t t1, t2;
        time(&t1);   // (the total below includes the warm-up; the per-pass times do not)
        {   // first-use costs of the device (module load, first launches, staging growth) are not part of either pass
            const TestPatient *big = nullptr;
            for (auto &pp : pts) if (!big || pp->n_all > big->n_all) big = pp.get();
            const vector<int> &m = big->meta;
            const vector<float> &t = big->t, &y = big->y;
            vector<double> th0(medgp_num_hyp(ctx), 0.0);
            int32_t s0 = 0, st0 = 0;
            double f0 = 0.0;
            if (!t.empty() && !medgp_set_patient(ctx, 0, (int)t.size(), kidx == 7 ? (const int32_t *)m.data() : nullptr, t.data(), y.data()))
                (void)medgp_nlml_grad(ctx, 1, &s0, th0.data(), 0, &f0, nullptr, &st0);
            vector<double> Lw((size_t)t.size() * t.size()), zw(t.size());
            if (!t.empty()) (void)medgp_factor(ctx, 0, th0.data(), Lw.data(), zw.data(), &st0);
            if (!t.empty()) {   // a full-size batch of the per-problem path: grows the staging / scratch buffers once
                const int nbw = (int)std::max<long long>(1, std::min<long long>(max_batch, n_problems));   // (never more problems than a pass has)
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
                if (!medgp_set_patients(ctx, nbw, sl.data(), off.data(), kidx == 7 ? pm.data() : nullptr, pt.data(), py.data()))
                    (void)medgp_fit_predict_batch(ctx, nbw, sl.data(), thw.data(), m2.data(), t2w.data(), mw.data(), vw.data(), stw.data());
            }
        }
        const auto tw0 = std::chrono::steady_clock::now();
        bool ok = run_test_pass(curr_exp, ctx, pts, fold, false, "mean_wo_update", alg, test_kernel_param, max_batch, per_problem, pool);
        const double ms_wo = ms_since(tw0);
        const auto tw1 = std::chrono::steady_clock::now();
        ok = ok && run_test_pass(curr_exp, ctx, pts, fold, true, "mean_w_update", alg, test_kernel_param, max_batch, per_problem, pool);
        cout << "INFO: pass wall time: without updating " << ms_wo << " ms, with updating " << ms_since(tw1) << " ms (" << pts.size() << " patient(s))" << endl;
        medgp_destroy(ctx);
        time(&t2);
        cout << "Finish all jobs. Total elapsed time = " << difftime(t2, t1) << " seconds" << endl;
        if (n_unreadable) cout << "ERROR: " << n_unreadable << " patient(s) could not be read" << endl;
        return (ok && n_unreadable == 0) ? 0 : 1;
    } catch (const std::exception &e) {
        cout << "ERROR: " << e.what() << endl;
        return 1;
    }
}
