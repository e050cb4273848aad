// medgp_train.cpp -- MI355X host of MedGP's per-patient training (the reference's main_one_train,
// ref: main_one_train.cpp:41-324), with the same CLI, config, inputs and output files:
//     medgp_train --cfg exp_setup.json --pan <PAN> --thread <n>
// plus a cohort mode that is the point of the GPU build: patients are trained in LOCK STEP on one
// device -- the random-init screening (HOT LOOP A, ref :228-253) and every optimiser step (HOT LOOP B,
// ref c_optimizer_scg.cpp:65,120,221) of all resident patients become single batched medgp_nlml_grad calls:
//     medgp_train --cfg exp_setup.json --pan-list pans.txt [--device d] [--resident R] [--max-batch B] [--queue file]
// Round 5: CONTINUOUS ADMISSION.  The trainer is long lived: it keeps up to --resident patients on the device and, whenever
// patients finish (early stop of the variational-EM loop, ref: util/c_optimizer_varEM.cpp:89-95; failed line searches, ref:
// util/c_optimizer_scg.cpp:125-131; budgets), pulls the next patients of the list into the freed slots -- load (a background
// thread reads ahead), upload, screening, optimiser start -- so the lock-step batches stay full until the list is exhausted.
// The list is walked longest patient first (cost model N^3); with --queue <file> several trainers (one per GPU, see
// medgp_amd/train_cohort.py) take their patients from ONE shared counter (flock), which balances them the way the reference's
// scheduler does with one job per patient (ref: medgpc/util/run_exp_generator.py:213-260).
// Outputs per patient (ref :257-323), written when the patient finishes: train_init_hyp_<PAN>.bin, train_hyp_<PAN>.bin,
// train_var_hyp_<PAN>.bin (prior mode 2), train_num_<PAN>.txt, train_flag_<PAN>.txt.
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
#include <numeric>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include "medgp_experiment.hpp"
#include "medgp_host.hpp"
#include "medgp_optimizer.hpp"
#include "medgp_workpool.hpp"
#include "medgp_loader.hpp"

using namespace medgp;
using std::cout;
using std::endl;
using std::string;
using std::vector;

namespace {

struct Patient {
    string PAN;
    int index = -1;            // position in the patient list (output order of the closing summary)
    vector<int> meta;
    vector<float> t, y;
    bool sample_flag = true, success = false, flag_data = false;
    string load_log, load_err;
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

// observations of a patient from the headers of its feature files (first token = count, ref: dataio/c_experiment.cpp:296-299);
// 0 when a file is missing (the real load then reports it)
int header_count(const c_experiment &ex, const string &PAN) {
    int n = 0;
    for (int fi : ex.get_feature_index()) {
        std::ifstream f((ex.get_data_dir() + PAN + "/feature" + std::to_string((long long)fi) + ".txt").c_str());
        float c = 0;
        if (f >> c) n += (int)c;
    }
    return n;
}

}  // namespace

static int train_main(int argc, const char *argv[]) {
    string exp_cfg, pan_arg, pan_list, queue_file, order_arg = "size";
    int thread_num = 1, device = 0, max_batch = 1024, host_threads = 0, pingpong_min = 512, merge_below = 256, resident = 1024, admit_min = -1, max_n_arg = 0, share = 1;
    bool pin_route = false, pingpong_given = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--cfg") && i + 1 < argc) exp_cfg = argv[++i];
        else if (!strcmp(argv[i], "--pan") && i + 1 < argc) pan_arg = argv[++i];
        else if (!strcmp(argv[i], "--pan-list") && i + 1 < argc) pan_list = argv[++i];
        else if (!strcmp(argv[i], "--thread") && i + 1 < argc) thread_num = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--max-batch") && i + 1 < argc) max_batch = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--host-threads") && i + 1 < argc) host_threads = atoi(argv[++i]);   // 0 = usable cores (<= 8)
        else if (!strcmp(argv[i], "--pingpong-min") && i + 1 < argc) { pingpong_min = atoi(argv[++i]); pingpong_given = true; }   // resident patients from which the lock-step loop runs them as two alternating halves
        else if (!strcmp(argv[i], "--merge-below") && i + 1 < argc) merge_below = atoi(argv[++i]);   // active patients below which the alternating groups are merged into one (tail of the list)
        else if (!strcmp(argv[i], "--resident") && i + 1 < argc) resident = atoi(argv[++i]);           // patients kept on the device at once (continuous admission)
        else if (!strcmp(argv[i], "--admit-min") && i + 1 < argc) admit_min = atoi(argv[++i]);         // free places of a group from which new patients are admitted (default: an eighth of the group)
        else if (!strcmp(argv[i], "--queue") && i + 1 < argc) queue_file = argv[++i];                  // shared work counter: several trainers walk ONE list
        else if (!strcmp(argv[i], "--order") && i + 1 < argc) order_arg = argv[++i];                   // size (longest first, default) | list
        else if (!strcmp(argv[i], "--max-n") && i + 1 < argc) max_n_arg = atoi(argv[++i]);             // largest observation count of the list (skips the header scan)
        else if (!strcmp(argv[i], "--share") && i + 1 < argc) share = std::max(1, atoi(argv[++i]));   // trainers that walk the --queue list together: bounds this trainer's first read-ahead to its share
        else if (!strcmp(argv[i], "--pin-route")) pin_route = true;   // medgp_pin_route: bit-identical results whatever the batch (slower for few large patients)
        else { cout << "Error: unknown argument: " << argv[i] << endl; return 1; }
    }
    if (exp_cfg.empty() || (pan_arg.empty() && pan_list.empty())) {
        cout << "usage:\n\t --cfg:\t the JSON configuration file\n\t --pan:\t ID of the training patient (comma separated for several)\n"
             << "\t --pan-list:\t file with one patient ID per line, optionally followed by its observation count (cohort mode)\n"
             << "\t --resident:\t patients resident on the device at once\n\t --queue:\t shared work-counter file\n\t --thread:\t accepted for compatibility, unused\n";
        return 1;
    }
    if (order_arg != "size" && order_arg != "list") { cout << "Error: --order must be size or list" << endl; return 1; }
    cout << "current configuration file: " << exp_cfg << endl;
    cout << "current threading number for matrix operation: " << thread_num << " (ignored: the GPU path has no host threads)" << endl;

    c_experiment curr_exp;
    if (!curr_exp.load(exp_cfg)) { cout << "ERROR: " << curr_exp.error() << endl; return 1; }
    const int kidx = curr_exp.get_kernel_index();
    const vector<int> kparam = curr_exp.get_kernel_param();
    const int H = curr_exp.get_hyp_num();

    vector<string> pans;
    vector<int> nhint;      // observation count per patient: second column of the list, or -1
    {
        size_t a = 0;
        while (a < pan_arg.size()) { size_t b = pan_arg.find(',', a); if (b == string::npos) b = pan_arg.size(); if (b > a) { pans.push_back(pan_arg.substr(a, b - a)); nhint.push_back(-1); } a = b + 1; }
        if (!pan_list.empty()) {
            std::ifstream f(pan_list.c_str());
            if (!f) { cout << "ERROR: File " << pan_list << " could not be opened." << endl; return 1; }
            string line;
            while (std::getline(f, line)) {
                std::istringstream ls(line);
                string s;
                long long n = -1;
                if (!(ls >> s)) continue;
                if (!(ls >> n)) n = -1;
                pans.push_back(s);
                nhint.push_back((int)n);
            }
        }
    }
    time_t t_start;
    time(&t_start);
    if (host_threads <= 0) host_threads = std::min(8, usable_cores());
    WorkPool pool(std::max(1, host_threads));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };

    // ---------------- sizes (for the walk order and the device capacity): the list's second column, else the feature files' headers
    const auto t_scan0 = now();
    {
        vector<int> need;
        for (size_t i = 0; i < pans.size(); i++) if (nhint[i] < 0) need.push_back((int)i);
        pool.parallel_for((int)need.size(), [&](int k) { nhint[need[k]] = header_count(curr_exp, pans[need[k]]); });
    }
    int max_n = std::max(1, max_n_arg);
    for (int n : nhint) max_n = std::max(max_n, n);
    vector<int> order(pans.size());
    std::iota(order.begin(), order.end(), 0);
    if (order_arg == "size") std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nhint[a] > nhint[b]; });
    const double t_scan = secs(t_scan0, now());

    // ---------------- device context: `resident` patient slots; per-entry matrices are allocated by the library as the calls need them
    resident = std::max(1, std::min<int>(resident, (int)pans.size()));
    const int ninit = curr_exp.get_scg_init_num();
    max_batch = std::max(1, max_batch);
    medgp_ctx *ctx = nullptr;
    if (medgp_create(&ctx, device, kidx, kparam[0], kparam[1], kparam[2])) { cout << "ERROR: " << medgp_last_error(nullptr) << endl; return 1; }
    if (medgp_reserve(ctx, resident, max_n, std::max(max_batch, resident))) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
    if (pin_route && medgp_pin_route(ctx, 1)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
    vector<int> free_slots;
    for (int s = resident - 1; s >= 0; s--) free_slots.push_back(s);
    vector<uint8_t> slot_used((size_t)resident, 0);   // slots that have held a patient (and may hold its prior on the device)

    Tickets tickets;
    tickets.path = queue_file;
    // (read-ahead: the whole resident set while it is filled for the first time -- on all host threads, nothing else runs yet -- then an
    //  eighth of it, set below once every group has been submitted once)
    // one patient of the list, loaded (ref: dataio/c_experiment.cpp:254-309, D feature files each) and checked (ref :185-197)
    auto load_patient = [&](size_t k) -> std::unique_ptr<Patient> {
        std::unique_ptr<Patient> p(new Patient());
        p->index = order[k];
        p->PAN = pans[(size_t)p->index];
        c_experiment e = curr_exp;                 // own error string per reader
        std::ostringstream os;
        os << "running individual training..." << endl << "current patinet PAN = " << p->PAN << endl;
        if (!e.get_one_patient_data(p->PAN, p->meta, p->t, p->y)) p->load_err = e.error();
        else {
            os << "current number of data points = " << p->t.size() << endl;
            vector<int> count_array(e.get_feature_index().size(), 0);
            for (size_t i = 0; i < p->t.size(); i++) count_array[p->meta[i]] += 1;
            for (int c : count_array) if (c < 2) { p->sample_flag = false; break; }
            if (!p->sample_flag) os << "skip due to insufficient # of samples" << endl;
        }
        p->load_log = os.str();
        return p;
    };
    // first fill: the whole resident set -- but with a shared counter at most this trainer's share of the list, so that the split
    // between the GPUs is decided by load, not by which rank's readers reached the file system first (--share from train_cohort.py)
    const size_t first_cap = queue_file.empty() ? (size_t)resident : std::min<size_t>((size_t)resident, (pans.size() + (size_t)share - 1) / (size_t)share);
    ReadAhead<Patient> loader(order.size(), tickets, std::max<size_t>(first_cap, 1), std::max(1, host_threads), load_patient);
    bool filled = false;

    vector<vector<double>> global_hyp_array;
    curr_exp.get_global_hyp(global_hyp_array);
    vector<double> init_block;              // [ninit][H], contiguous: what medgp_screen uploads
    for (const auto &v : global_hyp_array) init_block.insert(init_block.end(), v.begin(), v.end());
    const bool verbose = pans.size() == 1;   // the per-iteration lines of one patient; a cohort's state machines run on host threads

    // closing summary lines in list order (the reference prints one per process)
    vector<string> final_line(pans.size());
    vector<uint8_t> mine(pans.size(), 0);
    long long n_finished = 0;
    auto finish = [&](Patient &p) {   // outputs of one patient (ref :297-323); its device slot goes back to the pool
        if (p.sample_flag && p.success) {
            if (p.use_vem) { p.best_loss = p.vem.opt_loss; p.opt_parameter = p.vem.opt_parameter; }
            else { p.best_loss = p.scg.opt_loss; p.opt_parameter = p.scg.opt_parameter; }
            c_experiment::output_double_bin(curr_exp.get_exp_train_dir() + "train_hyp_" + p.PAN, p.opt_parameter);
            if (curr_exp.get_prior_mode() == 2)
                c_experiment::output_double_bin(curr_exp.get_exp_train_dir() + "train_var_hyp_" + p.PAN, p.prior.get_cov_varEM_all());
        }
        p.flag_data = p.sample_flag && p.success;
        std::ostringstream os;
        os << "finish individual id: " << p.PAN << " w/ " << p.t.size() << " samples; flag = " << p.flag_data
           << "; final loss = " << p.best_loss << endl;
        final_line[(size_t)p.index] = os.str();
        mine[(size_t)p.index] = 1;
        c_experiment::output_int_txt(curr_exp.get_exp_train_dir() + "train_num_" + p.PAN, {(int)p.t.size()});
        c_experiment::output_int_txt(curr_exp.get_exp_train_dir() + "train_flag_" + p.PAN, {(int)p.flag_data});
        if (p.slot >= 0) { free_slots.push_back(p.slot); p.slot = -1; }
        n_finished++;
    };

    long long total_evals = 0, steps = 0, screen_evals = 0, admissions = 0, n_unreadable = 0;
    double t_wait = 0.0, t_host = 0.0, t_screen = 0.0, t_admit = 0.0;   // seconds blocked in medgp_wait / in the optimiser state machines / in the screening calls / in admissions as a whole
    vector<std::unique_ptr<Patient>> owned;   // every patient currently resident (or being admitted)

    // ---------------- admission: upload, HOT LOOP A (random-init screening, nlml only, ref :228-253, batched over (patient, init)),
    // optimiser start (ref :260-292).  Returns the patients that now take part in the lock-step loop.
    auto admit = [&](vector<std::unique_ptr<Patient>> &&in, vector<Patient *> &started) -> bool {
        const auto ta0 = now();
        vector<Patient *> live;
        for (auto &up : in) {
            Patient *p = up.get();
            cout << p->load_log;
            // A patient that cannot be read, or is larger than the list announced, is skipped like one with too few samples: flag 0
            // through finish(), the run goes on, the exit status is non-zero at the end (the reference's one-process-per-patient model
            // loses exactly that patient too; medgp_test does the same, n_unreadable).  Round 5 ended the whole trainer here, dropping
            // every resident patient's progress and the tickets it had already pulled from a shared queue.
            if (!p->load_err.empty()) { cout << "ERROR: " << p->load_err << " -- patient " << p->PAN << " skipped" << endl; p->sample_flag = false; n_unreadable++; }
            else if ((int)p->t.size() > max_n) {
                cout << "ERROR: patient " << p->PAN << " has " << p->t.size() << " observations, more than the " << max_n << " the list announced -- skipped" << endl;
                p->sample_flag = false; n_unreadable++;
            }
            owned.push_back(std::move(up));
            if (!p->sample_flag) { finish(*p); continue; }
            p->slot = free_slots.back(); free_slots.pop_back();
            live.push_back(p);
        }
        in.clear();
        if (live.empty()) { t_admit += secs(ta0, now()); return true; }
        admissions++;
        {   // packed SoA arrays (stacked meta / t / y with offsets): one transfer, one scatter kernel
            vector<int32_t> slots(live.size());
            vector<int64_t> offs(live.size() + 1, 0);
            for (size_t s = 0; s < live.size(); s++) { slots[s] = live[s]->slot; offs[s + 1] = offs[s] + (int64_t)live[s]->t.size(); }
            vector<int32_t> meta_all((size_t)offs.back());
            vector<float> t_all((size_t)offs.back()), y_all((size_t)offs.back());
            pool.parallel_for((int)live.size(), [&](int s) {
                const Patient &p = *live[s];
                std::copy(p.meta.begin(), p.meta.end(), meta_all.begin() + offs[s]);
                std::copy(p.t.begin(), p.t.end(), t_all.begin() + offs[s]);
                std::copy(p.y.begin(), p.y.end(), y_all.begin() + offs[s]);
            });
            if (medgp_set_patients(ctx, (int)live.size(), slots.data(), offs.data(), kidx == 7 ? meta_all.data() : nullptr, t_all.data(), y_all.data())) {
                cout << "ERROR: " << medgp_last_error(ctx) << endl; return false;
            }
        }
        vector<Patient *> unprior;   // (a RE-USED slot still carries its previous patient's prior: the screening runs without one, ref :222-226)
        for (Patient *p : live) {
            p->prior.initialize_param(curr_exp.get_cov_num(), curr_exp.get_mean_num(), curr_exp.get_lik_num());
            if (slot_used[(size_t)p->slot]) unprior.push_back(p);
            slot_used[(size_t)p->slot] = 1;
        }
        if (!upload_priors(ctx, unprior, H, pool)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
        if (admissions == 1) cout << "finish initialization of prior" << endl;
        {
            const auto ts0 = now();
            // one call: the ninit vectors are the same for every patient (ref: dataio/c_experiment.cpp:418-441), so they go to the device
            // once and every (patient, vector) entry reads its row there (medgp_screen)
            vector<double> loss_flat((size_t)live.size() * ninit);
            vector<int32_t> stat_flat((size_t)live.size() * ninit);
            vector<int32_t> slots(live.size());
            for (size_t s = 0; s < live.size(); s++) slots[s] = live[s]->slot;
            if (ninit > 0 && medgp_screen(ctx, (int)live.size(), slots.data(), ninit, init_block.data(), loss_flat.data(), stat_flat.data())) {
                cout << "ERROR: " << medgp_last_error(ctx) << endl; return false;
            }
            screen_evals += (long long)live.size() * ninit;
            auto loss = [&](size_t s, int init) { return loss_flat[s * ninit + init]; };
            auto stat = [&](size_t s, int init) { return stat_flat[s * ninit + init]; };
            for (size_t s = 0; s < live.size(); s++) {
                Patient &p = *live[s];
                p.success = false;
                for (int init = 0; init < ninit; init++) {
                    const bool ok = stat(s, init) >= 0;
                    p.success = ok;
                    if (!ok) { cout << "WARNING: failed in computing objective!" << endl; break; }   // ref :243-246
                    if (loss(s, init) < p.best_loss) { p.best_loss = loss(s, init); p.best_init = global_hyp_array[init]; }
                }
                cout << "INFO: finish initialization " << ninit << " for " << p.PAN << "; best loss = " << p.best_loss << endl;
                c_experiment::output_double_bin(curr_exp.get_exp_train_dir() + "train_init_hyp_" + p.PAN, p.best_init);
            }
            t_screen += secs(ts0, now());
        }
        vector<Patient *> go;
        for (Patient *p : live) {
            if (!p->success) { finish(*p); continue; }
            p->prior.setup_param(kidx, kparam, curr_exp.get_prior_mode(), curr_exp.get_prior_hyp());
            p->use_vem = curr_exp.get_prior_mode() == 2;
            if (p->use_vem) p->vem.start((-1) * curr_exp.get_scg_max_iter_num(), p->best_init, &p->prior, kparam, curr_exp.get_lik_num(), curr_exp.get_prior_sub_opt_iter(), verbose);
            else p->scg.start((-1) * curr_exp.get_scg_max_iter_num(), p->best_init);
            p->active = p->use_vem ? !p->vem.done() : !p->scg.done();
            go.push_back(p);
        }
        if (!upload_priors(ctx, go, H, pool)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return false; }
        for (Patient *p : go) { if (p->active) started.push_back(p); else finish(*p); }
        t_admit += secs(ta0, now());
        return true;
    };

    // ---------------- HOT LOOP B: optimisation in lock step (ref :260-292)
    // Groups of patients that advance together: each group's step is one batched evaluation on one of the two asynchronous lanes
    // of the context (medgp_nlml_grad_async); while the device works on one group the host threads run the state machines of the
    // other (ref: the strictly sequential objective calls of util/c_optimizer_scg.cpp:65,120,221).  One group (no overlap, all
    // resident patients in one launch) unless there are enough of them that each half still fills the chip (--pingpong-min) or
    // they exceed max_batch.  A group that has lost an eighth of its members (--admit-min) takes new patients in before its next step.
    struct Group { vector<Patient *> mem; double *th = nullptr, *nl = nullptr, *gr = nullptr; int32_t *st = nullptr; vector<int32_t> slots; int nb = 0; size_t cap = 0; };
    // A heavy-tailed list (its largest patient at least four times the median): two alternating groups already from 2 x 256 resident
    // patients.  The steps of such a cohort are bound by the look-ahead chains of its few large patients; with two groups one group's
    // host work and transfers hide behind the other's chains (512 log-normal patients up to N = 5832 at the real budget, three
    // alternating runs: 14.57 s against 14.96 s as one group).  A uniform cohort keeps one group up to 2 x 512 (round 5: groups of 256
    // lose more on the device than the overlap wins there).
    if (!pingpong_given && !nhint.empty()) {
        vector<int> sorted_n(nhint);
        std::sort(sorted_n.begin(), sorted_n.end());
        const int med = sorted_n[sorted_n.size() / 2];
        if (med > 0 && sorted_n.back() >= 4 * med) pingpong_min = std::min(pingpong_min, 256);
    }
    int G = (resident + max_batch - 1) / max_batch;
    if (resident >= 2 * std::max(1, pingpong_min)) G = std::max(G, 2);
    G = std::max(G, 1);
    const size_t gcap = ((size_t)resident + G - 1) / G;            // members a group is filled up to
    const size_t bufcap = std::min<size_t>((size_t)resident, std::max<size_t>((size_t)max_batch, gcap));   // (a merged group may hold everybody that is left)
    // ---------------- the device's per-entry arenas, sized ONCE (round 6): the largest lock-step call is a group of the gcap largest
    // patients of the list (it is walked longest first; a merged tail group holds fewer and smaller ones), the largest screening chunks
    // are theirs too -- medgp_reserve_plan lays both out on the sizes alone and allocates that much.  Obtaining device memory can take seconds
    // on this platform when an earlier process has used it (medgp_hip.h); that wait now happens here, not inside the loop.
    double t_plan = 0.0;
    {
        vector<int32_t> top;
        vector<int> by_size(nhint);
        std::sort(by_size.begin(), by_size.end(), [](int a, int b) { return a > b; });
        for (size_t i = 0; i < by_size.size() && i < gcap; i++) top.push_back(std::max(0, by_size[i]));
        const auto tp0 = now();
        if (!top.empty() && medgp_reserve_plan(ctx, (int)top.size(), top.data(), ninit)) { cout << "ERROR: " << medgp_last_error(ctx) << endl; return 1; }
        t_plan = secs(tp0, now());
    }
    vector<Group> groups((size_t)G);
    for (Group &g : groups) {
        g.cap = gcap;
        g.th = (double *)medgp_host_alloc(sizeof(double) * bufcap * H);
        g.nl = (double *)medgp_host_alloc(sizeof(double) * bufcap);
        g.gr = (double *)medgp_host_alloc(sizeof(double) * bufcap * H);
        g.st = (int32_t *)medgp_host_alloc(sizeof(int32_t) * bufcap);
        if (!g.th || !g.nl || !g.gr || !g.st) { cout << "ERROR: pinned host allocation failed" << endl; return 1; }
    }
    if (admit_min < 0) admit_min = std::max<int>(1, (int)gcap / 8);
    const size_t groups_initial = groups.size();
    int merges = 0;
    bool announced = false;
    // new members: their first requests go into the rows behind the group's current members
    auto refill = [&](Group &g, bool must) -> bool {
        if (loader.exhausted()) return true;
        const size_t room = g.cap > g.mem.size() ? g.cap - g.mem.size() : 0;
        if (room == 0 || (!must && room < (size_t)admit_min && !g.mem.empty())) return true;
        // do not wait for the reader while there is work on the device; with an empty group (start, or everybody finished at once) wait
        vector<std::unique_ptr<Patient>> in = loader.take(std::min(room, free_slots.size()), must || g.mem.empty());
        if (in.empty()) return true;
        vector<Patient *> started;
        if (!admit(std::move(in), started)) return false;
        if (!announced && !started.empty()) { cout << "start doing optimization" << endl; announced = true; }
        for (Patient *p : started) {
            const vector<double> &rq = p->use_vem ? p->vem.request() : p->scg.request();
            std::copy(rq.begin(), rq.end(), g.th + g.mem.size() * H);
            g.mem.push_back(p);
        }
        return true;
    };
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
            if (!g.mem[k]->active) { finish(*g.mem[k]); continue; }
            if (w != k) { std::memmove(g.th + (size_t)w * H, g.th + (size_t)k * H, sizeof(double) * H); g.mem[w] = g.mem[k]; }
            w++;
        }
        g.mem.resize(w);
        // finished patients leave the host as well
        owned.erase(std::remove_if(owned.begin(), owned.end(), [](const std::unique_ptr<Patient> &p) { return p->slot < 0; }), owned.end());
        return true;
    };
    const auto t_loop0 = now();
    {
        std::deque<std::pair<int, int>> inflight;   // (group, lane), oldest first
        std::deque<int> ready;
        for (int g = 0; g < (int)groups.size(); g++) ready.push_back(g);
        int free_lanes[2] = {1, 1};
        // Re-forming at the tail: once the list is exhausted patients only leave, and when what is left no longer fills HALF the chip as
        // one launch (fewer than --merge-below = 256 patients altogether: two groups of < 128), two small launches per step are slower than
        // one.  (Down to there two groups of 255 cost the device what one group of 510 costs -- one workgroup per CU in the 8-wave shape
        // against two in the 4-wave shape -- and keep the host's share of a step hidden behind the other group's evaluation.)  When that point
        // is reached nothing new is queued until the device is idle, then the groups are merged.  (Merging earlier -- round 4 merged at
        // twice that size -- gives up the overlap of one group's state machines with the other group's evaluation for the whole tail:
        // 2048 patients through 1024 slots spent half of the run in one merged group of 979.)
        auto want_merge = [&]() {
            if (!loader.exhausted()) return false;
            size_t act = 0, nonempty = 0;
            for (const Group &g : groups) { act += g.mem.size(); nonempty += g.mem.empty() ? 0 : 1; }
            return nonempty >= 2 && act <= bufcap && (int)act <= max_batch && (int)act < std::max(1, std::min(merge_below, pingpong_min));
        };
        while (!ready.empty() || !inflight.empty()) {
            if (inflight.empty() && groups.size() > 1 && want_merge()) {
                Group &mg = groups[0];
                size_t total = mg.mem.size();
                for (size_t gi = 1; gi < groups.size(); gi++) {   // the pending request rows travel with their patients
                    Group &g = groups[gi];
                    if (!g.mem.empty()) std::memcpy(mg.th + total * H, g.th, sizeof(double) * g.mem.size() * H);
                    mg.mem.insert(mg.mem.end(), g.mem.begin(), g.mem.end());
                    total += g.mem.size();
                    g.mem.clear();
                }
                mg.cap = bufcap;
                cout << "INFO: " << groups.size() << " groups merged into one of " << total << " active patients" << endl;
                merges++;
                for (size_t gi = 1; gi < groups.size(); gi++) { Group &g = groups[gi]; medgp_host_free(g.th); medgp_host_free(g.nl); medgp_host_free(g.gr); medgp_host_free(g.st); }
                groups.resize(1);
                ready.clear();
                ready.push_back(0);
            }
            // queue a step of every group that is not in flight (at most one per lane); a group below its size takes new patients in first
            size_t tries = ready.size();
            while (tries-- > 0 && !ready.empty() && (free_lanes[0] || free_lanes[1]) && !(groups.size() > 1 && !inflight.empty() && want_merge())) {
                const int g = ready.front(); ready.pop_front();
                const int lane = free_lanes[0] ? 0 : 1;
                const auto t0 = now();
                // with nothing in flight the device is idle anyway: wait for the reader then (start of the run; everybody finished at once)
                if (!refill(groups[g], inflight.empty())) return 1;
                if (!submit(groups[g], lane)) return 1;
                t_host += secs(t0, now());
                if (groups[g].nb == 0) {                  // nobody to evaluate
                    if (!loader.exhausted()) ready.push_back(g);   // (the reader has nothing ready yet: asked again after the next wait)
                    continue;                                      // list exhausted: this group is done for good
                }
                free_lanes[lane] = 0;
                inflight.push_back({g, lane});
                steps++;
            }
            if (!filled && steps >= (long long)groups.size()) { loader.set_cap((size_t)std::max(2, resident / 8)); filled = true; }
            if (inflight.empty()) {
                if (loader.exhausted()) {
                    bool any = false;
                    for (const Group &g : groups) any = any || !g.mem.empty();
                    if (!any) break;
                }
                continue;
            }
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
    cout << "INFO: continuous admission: " << n_finished << " patients through " << resident << " resident slots in " << admissions << " admissions ("
         << t_admit << " s, of which screening " << t_screen << " s for " << screen_evals << " nlml-only evaluations); sizes from "
         << (t_scan > 0 ? "the list / file headers in " : "") << t_scan << " s" << endl;
    if (t_loop - t_admit > 0)
        cout << "INFO: gradient evaluations per second of lock-step time outside admissions: " << (double)total_evals / (t_loop - t_admit)
             << "; of the whole loop: " << (double)total_evals / t_loop << endl;

    // ---------------- closing lines, list order (ref :297-323; the files were written when each patient finished)
    for (size_t i = 0; i < pans.size(); i++) if (mine[i]) cout << final_line[i];
    owned.clear();
    const bool counter_failed = loader.failed();
    if (counter_failed) cout << "ERROR: the work counter " << queue_file << " could not be read or updated; patients of the list may be untrained" << endl;
    {   // where the host waited for device memory instead of queueing work (medgp_alloc_stats)
        double as = 0.0; int64_t an = 0, ab = 0;
        if (medgp_alloc_stats(ctx, &as, &an, &ab) == 0)
            cout << "INFO: device memory: " << as << " s in " << an << " management calls (" << t_plan << " s of it announcing the sizes up front), "
                 << (double)ab / 1073741824.0 << " GB held by the per-entry arenas" << endl;
    }
    if (n_unreadable) cout << "ERROR: " << n_unreadable << " patient(s) could not be read or exceeded the announced size; they were skipped (flag 0)" << endl;
    if (ctx) medgp_destroy(ctx);
    time_t t_end;
    time(&t_end);
    cout << "Finish all jobs. Total elapsed time = " << difftime(t_end, t_start) << " seconds" << endl;
    return (counter_failed || n_unreadable) ? 1 : 0;
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
