// medgp_train.cpp -- MI355X host of MedGP's per-patient training (the reference's main_one_train,
// ref: main_one_train.cpp:41-324), with the same CLI, config, inputs and output files:
//     medgp_train --cfg exp_setup.json --pan <PAN> --thread <n>
// plus a cohort mode that is the point of the GPU build: several patients are trained in LOCK STEP on one
// device -- the random-init screening (HOT LOOP A, ref :228-253) and every optimiser step (HOT LOOP B,
// ref c_optimizer_scg.cpp:65,120,221) of all patients become single batched medgp_nlml_grad calls:
//     medgp_train --cfg exp_setup.json --pan-list pans.txt [--device d] [--max-batch B]
// Outputs per patient (ref :257-323): train_init_hyp_<PAN>.bin, train_hyp_<PAN>.bin, train_var_hyp_<PAN>.bin
// (prior mode 2), train_num_<PAN>.txt, train_flag_<PAN>.txt.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <sstream>
#include <thread>
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
#include "medgp_workpool.hpp"

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

// prior descriptors of several patients in ONE transfer (medgp_set_priors): every variational-EM outer iteration changes psi of
// every patient that finished its sub-optimisation in this step (ref: util/c_optimizer_varEM.cpp:98-162)
bool upload_priors(medgp_ctx *ctx, const vector<Patient *> &ps, int H, WorkPool &pool) {
    if (ps.empty()) return true;
    const size_t n = ps.size();
    vector<int32_t> slots(n), type(n * H);
    vector<uint8_t> flag(n * H), ex(n * H);
    vector<float> p0(n * H), p1(n * H);
    pool.parallel_for((int)n, [&](int k) {
        vector<uint8_t> f, e;
        vector<int32_t> t;
        vector<float> a, b;
        ps[k]->prior.flatten(f, t, e, a, b);
        slots[k] = ps[k]->slot;
        std::copy(f.begin(), f.end(), flag.begin() + (size_t)k * H);
        std::copy(t.begin(), t.end(), type.begin() + (size_t)k * H);
        std::copy(e.begin(), e.end(), ex.begin() + (size_t)k * H);
        std::copy(a.begin(), a.end(), p0.begin() + (size_t)k * H);
        std::copy(b.begin(), b.end(), p1.begin() + (size_t)k * H);
    });
    return medgp_set_priors(ctx, (int)n, slots.data(), flag.data(), type.data(), ex.data(), p0.data(), p1.data()) == 0;
}

}  // namespace

static int train_main(int argc, const char *argv[]) {
    string exp_cfg, pan_arg, pan_list;
    int thread_num = 1, device = 0, max_batch = 1024, host_threads = 0, pingpong_min = 1024;
    bool pin_route = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--cfg") && i + 1 < argc) exp_cfg = argv[++i];
        else if (!strcmp(argv[i], "--pan") && i + 1 < argc) pan_arg = argv[++i];
        else if (!strcmp(argv[i], "--pan-list") && i + 1 < argc) pan_list = argv[++i];
        else if (!strcmp(argv[i], "--thread") && i + 1 < argc) thread_num = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-batch") && i + 1 < argc) max_batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--host-threads") && i + 1 < argc) host_threads = atoi(argv[++i]);   // 0 = usable cores (<= 8)
        else if (!strcmp(argv[i], "--pingpong-min") && i + 1 < argc) pingpong_min = atoi(argv[++i]);   // active patients from which the lock-step loop splits them in two alternating halves
        else if (!strcmp(argv[i], "--pin-route")) pin_route = true;   // medgp_pin_route: bit-identical results whatever the batch (slower for few large patients)
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

    // ---------------- load patients, data-quality check (ref :185-197).  The reference loads ONE patient per process
    // (ref: dataio/c_experiment.cpp:254-309, D feature files each); a cohort is loaded by a pool of host threads over the
    // patients, the log lines are printed afterwards in patient order, and the whole cohort goes to the device as ONE packed
    // upload (medgp_set_patients: one transfer, one scatter kernel)
    if (host_threads <= 0) host_threads = std::min(8, usable_cores());
    WorkPool pool(std::max(1, host_threads));
    const auto t_load0 = std::chrono::steady_clock::now();
    vector<std::unique_ptr<Patient>> pts(pans.size());
    vector<string> load_log(pans.size()), load_err(pans.size());
    pool.parallel_for((int)pans.size(), [&](int i) {
        c_experiment ex = curr_exp;                 // own error string per task
        std::unique_ptr<Patient> p(new Patient());
        p->PAN = pans[i];
        std::ostringstream os;
        os << "running individual training..." << endl << "current patinet PAN = " << p->PAN << endl;
        if (!ex.get_one_patient_data(p->PAN, p->meta, p->t, p->y)) { load_err[i] = ex.error(); pts[i] = std::move(p); return; }
        os << "current number of data points = " << p->t.size() << endl;
        vector<int> count_array(ex.get_feature_index().size(), 0);
        for (size_t k = 0; k < p->t.size(); k++) count_array[p->meta[k]] += 1;
        for (int c : count_array) if (c < 2) { p->sample_flag = false; break; }
        if (!p->sample_flag) os << "skip due to insufficient # of samples" << endl;
        load_log[i] = os.str();
        pts[i] = std::move(p);
    });
    int max_n = 1;
    for (size_t i = 0; i < pans.size(); i++) {
        cout << load_log[i];
        if (!load_err[i].empty()) { cout << "ERROR: " << load_err[i] << endl; return 1; }
        max_n = std::max(max_n, (int)pts[i]->t.size());
    }
    const auto t_load1 = std::chrono::steady_clock::now();

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
        if (pin_route && medgp_pin_route(ctx, 1)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
        // packed SoA cohort arrays (stacked meta / t / y with offsets)
        vector<int32_t> slots(live.size());
        vector<int64_t> offs(live.size() + 1, 0);
        for (size_t s = 0; s < live.size(); s++) { live[s]->slot = (int)s; slots[s] = (int)s; offs[s + 1] = offs[s] + (int64_t)live[s]->t.size(); }
        vector<int32_t> meta_all((size_t)offs.back());
        vector<float> t_all((size_t)offs.back()), y_all((size_t)offs.back());
        pool.parallel_for((int)live.size(), [&](int s) {
            const Patient &p = *live[s];
            std::copy(p.meta.begin(), p.meta.end(), meta_all.begin() + offs[s]);
            std::copy(p.t.begin(), p.t.end(), t_all.begin() + offs[s]);
            std::copy(p.y.begin(), p.y.end(), y_all.begin() + offs[s]);
        });
        if (medgp_set_patients(ctx, (int)live.size(), slots.data(), offs.data(), kidx == 7 ? meta_all.data() : nullptr, t_all.data(), y_all.data())) {
            cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1;
        }
        for (Patient *p : live) p->prior.initialize_param(curr_exp.get_cov_num(), curr_exp.get_mean_num(), curr_exp.get_lik_num());   // ref :222-226
        cout << "finish initialization of prior" << endl;
    }
    {
        const auto t_up = std::chrono::steady_clock::now();
        cout << "INFO: loaded " << pans.size() << " patients x " << curr_exp.get_feature_index().size() << " feature files in "
             << std::chrono::duration<double>(t_load1 - t_load0).count() << " s on " << pool.size() << " host threads; packed upload "
             << std::chrono::duration<double>(t_up - t_load1).count() << " s" << endl;
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
    const bool verbose = live.size() == 1;   // the per-iteration lines of one patient; a cohort's state machines run on host threads
    for (Patient *p : live) {
        if (!p->success) continue;
        p->prior.setup_param(kidx, kparam, curr_exp.get_prior_mode(), curr_exp.get_prior_hyp());
        p->use_vem = curr_exp.get_prior_mode() == 2;
        if (p->use_vem) p->vem.start((-1) * curr_exp.get_scg_max_iter_num(), p->best_init, &p->prior, kparam, curr_exp.get_lik_num(), curr_exp.get_prior_sub_opt_iter(), verbose);
        else p->scg.start((-1) * curr_exp.get_scg_max_iter_num(), p->best_init);
        p->active = p->use_vem ? !p->vem.done() : !p->scg.done();
        running.push_back(p);
    }
    if (ctx && !upload_priors(ctx, running, H, pool)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
    cout << "start doing optimization" << endl;
    long long total_evals = 0, steps = 0;
    double t_wait = 0.0, t_host = 0.0;   // seconds blocked in medgp_wait / in the optimiser state machines (incl. request copies, prior uploads)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const auto t_loop0 = now();
    // Groups of patients that advance together: each group's step is one batched evaluation on one of the two asynchronous lanes
    // of the context (medgp_nlml_grad_async); while the device works on one group the host threads run the state machines of the
    // other (ref: the strictly sequential objective calls of util/c_optimizer_scg.cpp:65,120,221).  One group (no overlap, the
    // whole active set in one launch -- the most efficient use of the device) unless the active set is large enough that each
    // half still fills the chip (--pingpong-min) or exceeds max_batch.
    struct Group { vector<Patient *> mem; double *th = nullptr, *nl = nullptr, *gr = nullptr; int32_t *st = nullptr; vector<int32_t> slots; int nb = 0; };
    vector<Group> groups;
    {
        vector<Patient *> act;
        for (Patient *p : running) if (p->active) act.push_back(p);
        int G = (int)((act.size() + max_batch - 1) / std::max(1, max_batch));
        if ((int)act.size() >= 2 * std::max(1, pingpong_min)) G = std::max(G, 2);
        G = std::max(G, 1);
        const size_t per = (act.size() + G - 1) / std::max(G, 1);
        for (int g = 0; g < G; g++) {
            Group gp;
            for (size_t k = g * per; k < std::min(act.size(), (g + 1) * per); k++) gp.mem.push_back(act[k]);
            if (gp.mem.empty()) continue;
            const size_t cap = gp.mem.size();
            gp.th = (double *)medgp_host_alloc(sizeof(double) * cap * H);
            gp.nl = (double *)medgp_host_alloc(sizeof(double) * cap);
            gp.gr = (double *)medgp_host_alloc(sizeof(double) * cap * H);
            gp.st = (int32_t *)medgp_host_alloc(sizeof(int32_t) * cap);
            if (!gp.th || !gp.nl || !gp.gr || !gp.st) { cout << "ERROR: pinned host allocation failed" << endl; return 1; }
            groups.push_back(std::move(gp));
        }
    }
    // rows of g.th hold the pending requests of g.mem (same order); feed() refreshes them in the same parallel pass that runs the
    // state machines, so a lock-step step costs ONE fork/join of the host threads
    auto fill_requests = [&](Group &g) {
        g.nb = (int)g.mem.size();
        pool.parallel_for(g.nb, [&](int k) {
            Patient *p = g.mem[k];
            const vector<double> &rq = p->use_vem ? p->vem.request() : p->scg.request();
            std::copy(rq.begin(), rq.end(), g.th + (size_t)k * H);
        });
    };
    for (Group &g : groups) fill_requests(g);
    const size_t groups_initial = groups.size();
    int merges = 0;
    auto submit = [&](Group &g, int lane) -> bool {   // queue the evaluation of the group's pending requests
        g.nb = (int)g.mem.size();
        if (g.nb == 0) return true;
        g.slots.resize(g.nb);
        for (int k = 0; k < g.nb; k++) g.slots[k] = g.mem[k]->slot;
        if (medgp_nlml_grad_async(ctx, lane, g.nb, g.slots.data(), g.th, 1, g.nl, g.gr, g.st)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
        total_evals += g.nb;
        return true;
    };
    auto feed = [&](Group &g) -> bool {
        vector<uint8_t> dirty(g.nb, 0);
        pool.parallel_for(g.nb, [&](int k) {
            Patient *p = g.mem[k];
            vector<double> gk(g.gr + (size_t)k * H, g.gr + (size_t)(k + 1) * H);
            const bool ok = g.st[k] >= 0;
            if (p->use_vem) {
                p->vem.feed(ok, g.nl[k], gk);
                dirty[k] = p->vem.prior_changed() ? 1 : 0;
                p->active = !p->vem.done();
            } else {
                p->scg.feed(ok, g.nl[k], gk);
                p->active = !p->scg.done();
            }
            if (p->active) {   // next request into the same row (rows are compacted below when somebody finished)
                const vector<double> &rq = p->use_vem ? p->vem.request() : p->scg.request();
                std::copy(rq.begin(), rq.end(), g.th + (size_t)k * H);
            }
        });
        vector<Patient *> ch;
        for (int k = 0; k < g.nb; k++) if (dirty[k] && g.mem[k]->active) ch.push_back(g.mem[k]);
        if (!upload_priors(ctx, ch, H, pool)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
        int w = 0;
        for (int k = 0; k < g.nb; k++) {
            if (!g.mem[k]->active) continue;
            if (w != k) { std::memmove(g.th + (size_t)w * H, g.th + (size_t)k * H, sizeof(double) * H); g.mem[w] = g.mem[k]; }
            w++;
        }
        g.mem.resize(w);
        return true;
    };
    {
        std::deque<std::pair<int, int>> inflight;   // (group, lane), oldest first
        std::deque<int> ready;
        for (int g = 0; g < (int)groups.size(); g++) ready.push_back(g);
        int free_lanes[2] = {1, 1};
        // Re-forming: groups are cut once, from the active set at the start; patients finish at different times (early stop of the
        // variational-EM outer loop, ref: util/c_optimizer_varEM.cpp:89-95; line-search failures, ref: util/c_optimizer_scg.cpp:125-131),
        // and once what is left fits ONE launch that no longer fills the chip twice over, two half-empty launches per step are
        // slower than one.  When that point is reached nothing new is queued until the device is idle, then the groups are merged.
        auto want_merge = [&]() {
            size_t act = 0, nonempty = 0;
            for (const Group &g : groups) { act += g.mem.size(); nonempty += g.mem.empty() ? 0 : 1; }
            return nonempty >= 2 && (int)act <= max_batch && (int)act < 2 * std::max(1, pingpong_min);
        };
        while (!ready.empty() || !inflight.empty()) {
            if (inflight.empty() && groups.size() > 1 && want_merge()) {
                Group mg;
                for (Group &g : groups) mg.mem.insert(mg.mem.end(), g.mem.begin(), g.mem.end());
                const size_t cap = mg.mem.size();
                mg.th = (double *)medgp_host_alloc(sizeof(double) * cap * H);
                mg.nl = (double *)medgp_host_alloc(sizeof(double) * cap);
                mg.gr = (double *)medgp_host_alloc(sizeof(double) * cap * H);
                mg.st = (int32_t *)medgp_host_alloc(sizeof(int32_t) * cap);
                if (!mg.th || !mg.nl || !mg.gr || !mg.st) { cout << "ERROR: pinned host allocation failed" << endl; return 1; }
                size_t w = 0;
                for (Group &g : groups) {   // the pending request rows travel with their patients
                    if (!g.mem.empty()) std::memcpy(mg.th + w * H, g.th, sizeof(double) * g.mem.size() * H);
                    w += g.mem.size();
                    medgp_host_free(g.th); medgp_host_free(g.nl); medgp_host_free(g.gr); medgp_host_free(g.st);
                }
                cout << "INFO: " << groups.size() << " groups merged into one of " << cap << " active patients" << endl;
                merges++;
                groups.clear();
                groups.push_back(std::move(mg));
                ready.clear();
                ready.push_back(0);
            }
            while (!ready.empty() && (free_lanes[0] || free_lanes[1]) && !(groups.size() > 1 && !inflight.empty() && want_merge())) {
                const int g = ready.front(); ready.pop_front();
                const int lane = free_lanes[0] ? 0 : 1;
                const auto t0 = now();
                if (!submit(groups[g], lane)) return 1;
                t_host += secs(t0, now());
                if (groups[g].nb == 0) continue;          // every patient of the group is done
                free_lanes[lane] = 0;
                inflight.push_back({g, lane});
                steps++;
            }
            if (inflight.empty()) break;
            const auto pr = inflight.front(); inflight.pop_front();
            const auto t1 = now();
            if (medgp_wait(ctx, pr.second)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
            const auto t2 = now();
            if (!feed(groups[pr.first])) return 1;
            t_wait += secs(t1, t2);
            t_host += secs(t2, now());
            free_lanes[pr.second] = 1;
            ready.push_back(pr.first);
        }
        for (Group &g : groups) { medgp_host_free(g.th); medgp_host_free(g.nl); medgp_host_free(g.gr); medgp_host_free(g.st); }
    }
    const double t_loop = secs(t_loop0, now());
    cout << "optimization finished: " << total_evals << " nlml+grad evaluations in " << steps << " lock-step batches" << endl;
    cout << "INFO: lock-step optimisation: " << t_loop << " s wall (" << t_wait << " s waiting for the device, " << t_host
         << " s in the host optimiser on " << pool.size() << " threads, " << groups_initial << " group(s)" << (merges ? ", merged into one when the active set had shrunk" : "") << ")" << endl;

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

// exceptions of the host side (bad_alloc, a throwing loader on a worker thread: WorkPool rethrows them on the calling thread)
// end the run with the reference's error convention -- an ERROR line and a non-zero exit code -- not with std::terminate
int main(int argc, const char *argv[]) {
    try {
        return train_main(argc, argv);
    } catch (const std::exception &e) {
        cout << "ERROR: " << e.what() << endl;
        return 1;
    } catch (...) {
        cout << "ERROR: unknown exception" << endl;
        return 1;
    }
}
