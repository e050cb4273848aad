// medgp_train.cpp -- MI355X host of MedGP's per-patient training (the reference's main_one_train,
// ref: main_one_train.cpp:41-324), with the same CLI, config, inputs and output files:
//     medgp_train --cfg exp_setup.json --pan <PAN> --thread <n>
// plus a cohort mode that is the point of the GPU build: several patients are trained in LOCK STEP on one
// device -- the random-init screening (HOT LOOP A, ref :228-253) and every optimiser step (HOT LOOP B,
// ref c_optimizer_scg.cpp:65,120,221) of all patients become single batched medgp_nlml_grad calls:
//     medgp_train --cfg exp_setup.json --pan-list pans.txt [--device d] [--max-batch B]
// Outputs per patient (ref :257-323): train_init_hyp_<PAN>.bin, train_hyp_<PAN>.bin, train_var_hyp_<PAN>.bin
// (prior mode 2), train_num_<PAN>.txt, train_flag_<PAN>.txt.
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "medgp_experiment.hpp"
#include "medgp_host.hpp"
#include "medgp_optimizer.hpp"

using namespace medgp;
using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace {

struct Patient {
    string PAN;
    vector<int> meta;
    vector<float> t, y;
    bool sample_flag = true, success = false, flag_data = false;
    int slot = -1;
    c_prior prior;
    double best_loss = std::numeric_limits<double>::max();
    vector<double> best_init, opt_parameter;
    scg_machine scg;
    varem_machine vem;
    bool use_vem = false, active = false;
};

bool upload_prior(medgp_ctx *ctx, Patient &p) {
    vector<uint8_t> flag, ex;
    vector<int32_t> type;
    vector<float> p0, p1;
    p.prior.flatten(flag, type, ex, p0, p1);
    return medgp_set_prior(ctx, p.slot, flag.data(), type.data(), ex.data(), p0.data(), p1.data()) == 0;
}

}  // namespace

int main(int argc, const char *argv[]) {
    string exp_cfg, pan_arg, pan_list;
    int thread_num = 1, device = 0, max_batch = 1024;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--cfg") && i + 1 < argc) exp_cfg = argv[++i];
        else if (!strcmp(argv[i], "--pan") && i + 1 < argc) pan_arg = argv[++i];
        else if (!strcmp(argv[i], "--pan-list") && i + 1 < argc) pan_list = argv[++i];
        else if (!strcmp(argv[i], "--thread") && i + 1 < argc) thread_num = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-batch") && i + 1 < argc) max_batch = atoi(argv[++i]);
        else { cout << "Error: unknown argument: " << argv[i] << endl; return 1; }
    }
    if (exp_cfg.empty() || (pan_arg.empty() && pan_list.empty())) {
        cout << "usage:\n\t --cfg:\t the JSON configuration file\n\t --pan:\t ID of the training patient (comma separated for several)\n"
             << "\t --pan-list:\t file with one patient ID per line (cohort mode)\n\t --thread:\t accepted for compatibility, unused\n";
        return 1;
    }
    cout << "current configuration file: " << exp_cfg << endl;
    cout << "current threading number for matrix operation: " << thread_num << " (ignored: the GPU path has no host threads)" << endl;

    c_experiment curr_exp;
    if (!curr_exp.load(exp_cfg)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
    const int kidx = curr_exp.get_kernel_index();
    const vector<int> kparam = curr_exp.get_kernel_param();
    const int H = curr_exp.get_hyp_num();

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
    time_t t_start;
    time(&t_start);

    // ---------------- load patients, data-quality check (ref :185-197)
    vector<std::unique_ptr<Patient>> pts;
    int max_n = 1;
    for (const string &PAN : pans) {
        std::unique_ptr<Patient> p(new Patient());
        p->PAN = PAN;
        cout << "running individual training..." << endl << "current patinet PAN = " << PAN << endl;
        if (!curr_exp.get_one_patient_data(PAN, p->meta, p->t, p->y)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
        cout << "current number of data points = " << p->t.size() << endl;
        vector<int> count_array(curr_exp.get_feature_index().size(), 0);
        for (size_t k = 0; k < p->t.size(); k++) count_array[p->meta[k]] += 1;
        for (int c : count_array) if (c < 2) { p->sample_flag = false; break; }
        if (!p->sample_flag) cout << "skip due to insufficient # of samples" << endl;
        max_n = std::max(max_n, (int)p->t.size());
        pts.push_back(std::move(p));
    }

    // ---------------- device context: every usable patient resident in its own slot
    vector<Patient *> live;
    for (auto &p : pts) if (p->sample_flag) live.push_back(p.get());
    medgp_ctx *ctx = nullptr;
    const int nslot = std::max<int>(1, (int)live.size());
    if (!live.empty()) {
        if (medgp_create(&ctx, device, kidx, kparam[0], kparam[1], kparam[2])) { cout << "ERROR: " << medgp_last_error(nullptr) << endl; return 1; }
        {   // evaluations that can exist at once, and what fits a memory budget: every batch entry owns two ldn x ldn fp64
            // matrices (medgp_reserve allocates them up front); the screening and lock-step loops chunk by max_batch
            const long long ldn = (max_n + 63) / 64 * 64;
            const long long want = (long long)live.size() * std::max(curr_exp.get_scg_init_num(), 1);
            const long long fit = std::max<long long>(1, (48LL << 30) / (16 * ldn * ldn + 4096));
            max_batch = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(std::max(1, max_batch), want), fit));
        }
        if (medgp_reserve(ctx, nslot, max_n, max_batch)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
        for (size_t s = 0; s < live.size(); s++) {
            Patient &p = *live[s];
            p.slot = (int)s;
            const int32_t *mp = (kidx == 7) ? (const int32_t *)p.meta.data() : nullptr;
            if (medgp_set_patient(ctx, p.slot, (int)p.t.size(), mp, p.t.data(), p.y.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
            p.prior.initialize_param(curr_exp.get_cov_num(), curr_exp.get_mean_num(), curr_exp.get_lik_num());   // ref :222-226
        }
        cout << "finish initialization of prior" << endl;
    }

    // ---------------- HOT LOOP A: random-init screening, nlml only (ref :228-253), batched over (patient, init)
    vector<vector<double>> global_hyp_array;
    curr_exp.get_global_hyp(global_hyp_array);
    const int ninit = curr_exp.get_scg_init_num();
    if (!live.empty()) {
        vector<vector<double>> loss(live.size(), vector<double>(ninit, 0.0));
        vector<vector<int32_t>> stat(live.size(), vector<int32_t>(ninit, 0));
        vector<int32_t> slots;
        vector<double> thetas, nl;
        vector<int32_t> st;
        vector<std::pair<int, int>> who;
        auto flush = [&]() -> bool {
            if (slots.empty()) return true;
            nl.resize(slots.size()); st.resize(slots.size());
            if (medgp_nlml_grad(ctx, (int)slots.size(), slots.data(), thetas.data(), 0, nl.data(), nullptr, st.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
            for (size_t k = 0; k < slots.size(); k++) { loss[who[k].first][who[k].second] = nl[k]; stat[who[k].first][who[k].second] = st[k]; }
            slots.clear(); thetas.clear(); who.clear();
            return true;
        };
        for (size_t s = 0; s < live.size(); s++)
            for (int init = 0; init < ninit; init++) {
                slots.push_back(live[s]->slot);
                thetas.insert(thetas.end(), global_hyp_array[init].begin(), global_hyp_array[init].end());
                who.push_back({(int)s, init});
                if ((int)slots.size() == max_batch && !flush()) return 1;
            }
        if (!flush()) return 1;
        for (size_t s = 0; s < live.size(); s++) {
            Patient &p = *live[s];
            p.success = false;
            for (int init = 0; init < ninit; init++) {
                const bool ok = stat[s][init] >= 0;
                p.success = ok;
                if (!ok) { cout << "WARNING: failed in computing objective!" << endl; break; }   // ref :243-246
                if (loss[s][init] < p.best_loss) { p.best_loss = loss[s][init]; p.best_init = global_hyp_array[init]; }
            }
            cout << "INFO: finish initialization " << ninit << " for " << p.PAN << "; best loss = " << p.best_loss << endl;
            c_experiment::output_double_bin(curr_exp.get_exp_train_dir() + "train_init_hyp_" + p.PAN, p.best_init);
        }
    }

    // ---------------- HOT LOOP B: optimisation in lock step (ref :260-292)
    vector<Patient *> running;
    for (Patient *p : live) {
        if (!p->success) continue;
        p->prior.setup_param(kidx, kparam, curr_exp.get_prior_mode(), curr_exp.get_prior_hyp());
        if (!upload_prior(ctx, *p)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
        p->use_vem = curr_exp.get_prior_mode() == 2;
        if (p->use_vem) p->vem.start((-1) * curr_exp.get_scg_max_iter_num(), p->best_init, &p->prior, kparam, curr_exp.get_lik_num(), curr_exp.get_prior_sub_opt_iter(), true);
        else p->scg.start((-1) * curr_exp.get_scg_max_iter_num(), p->best_init);
        p->active = p->use_vem ? !p->vem.done() : !p->scg.done();
        running.push_back(p);
    }
    cout << "start doing optimization" << endl;
    long long total_evals = 0, steps = 0;
    double t_dev = 0.0, t_host = 0.0;   // seconds inside medgp_nlml_grad / in the optimiser state machines (incl. prior uploads)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    while (true) {
        vector<Patient *> act;
        for (Patient *p : running) if (p->active) act.push_back(p);
        if (act.empty()) break;
        for (size_t c0 = 0; c0 < act.size(); c0 += max_batch) {
            const int nb = (int)std::min<size_t>(max_batch, act.size() - c0);
            const auto t0 = now();
            vector<int32_t> slots(nb), st(nb);
            vector<double> thetas((size_t)nb * H), nl(nb), gr((size_t)nb * H);
            for (int k = 0; k < nb; k++) {
                Patient *p = act[c0 + k];
                slots[k] = p->slot;
                const vector<double> &rq = p->use_vem ? p->vem.request() : p->scg.request();
                std::copy(rq.begin(), rq.end(), thetas.begin() + (size_t)k * H);
            }
            const auto t1 = now();
            if (medgp_nlml_grad(ctx, nb, slots.data(), thetas.data(), 1, nl.data(), gr.data(), st.data())) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
            const auto t2 = now();
            total_evals += nb;
            // (the state machines are independent and were tried under `#pragma omp parallel for`: on the GPU box -- a 16-core cgroup
            //  quota on a host with many more hardware threads -- libgomp's default team and its spinning workers starved the HIP
            //  runtime: 0.065 -> 10.4 s of host time and 0.15 -> 1.2 s inside medgp_nlml_grad for 94 steps of 256 patients.  The
            //  serial loop is 30 % of the optimisation loop's wall time, 9 % of the whole run: left serial.)
            for (int k = 0; k < nb; k++) {
                Patient *p = act[c0 + k];
                vector<double> g(gr.begin() + (size_t)k * H, gr.begin() + (size_t)(k + 1) * H);
                const bool ok = st[k] >= 0;
                if (p->use_vem) {
                    p->vem.feed(ok, nl[k], g);
                    if (p->vem.prior_changed() && !upload_prior(ctx, *p)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
                    p->active = !p->vem.done();
                } else {
                    p->scg.feed(ok, nl[k], g);
                    p->active = !p->scg.done();
                }
            }
            t_dev += secs(t1, t2);
            t_host += secs(t0, t1) + secs(t2, now());
        }
        steps++;
    }
    cout << "optimization finished: " << total_evals << " nlml+grad evaluations in " << steps << " lock-step batches" << endl;
    cout << "INFO: lock-step optimisation: " << t_dev << " s in medgp_nlml_grad, " << t_host << " s in the host optimiser" << endl;

    // ---------------- outputs (ref :297-323)
    for (auto &pp : pts) {
        Patient &p = *pp;
        if (p.sample_flag && p.success) {
            if (p.use_vem) { p.best_loss = p.vem.opt_loss; p.opt_parameter = p.vem.opt_parameter; }
            else { p.best_loss = p.scg.opt_loss; p.opt_parameter = p.scg.opt_parameter; }
            c_experiment::output_double_bin(curr_exp.get_exp_train_dir() + "train_hyp_" + p.PAN, p.opt_parameter);
            if (curr_exp.get_prior_mode() == 2)
                c_experiment::output_double_bin(curr_exp.get_exp_train_dir() + "train_var_hyp_" + p.PAN, p.prior.get_cov_varEM_all());
        }
        p.flag_data = p.sample_flag && p.success;
        cout << "finish individual id: " << p.PAN << " w/ " << p.t.size() << " samples; flag = " << p.flag_data
             << "; final loss = " << p.best_loss << endl;
        c_experiment::output_int_txt(curr_exp.get_exp_train_dir() + "train_num_" + p.PAN, {(int)p.t.size()});
        c_experiment::output_int_txt(curr_exp.get_exp_train_dir() + "train_flag_" + p.PAN, {(int)p.flag_data});
    }
    if (ctx) medgp_destroy(ctx);
    time_t t_end;
    time(&t_end);
    cout << "Finish all jobs. Total elapsed time = " << difftime(t_end, t_start) << " seconds" << endl;
    return 0;
}
