// medgp_experiment.hpp -- experiment configuration + per-patient file I/O of the MedGP hosts, byte compatible
// with the reference's c_experiment (ref: dataio/c_experiment.h, dataio/c_experiment.cpp):
//   exp_setup.json keys                      ref c_experiment.cpp:45-159 (writer: medgpc/util/config.py:5-35)
//   hyp_bound.txt  "lb\nub\n" per hyper      ref :395-417           (writer: config.py:38-66)
//   <data_dir>/<PAN>/feature<idx>.txt        count, then (t, v) pairs                ref :296-307
//   <data_dir>/feature<idx>_stat.bin         two doubles (mean, std)                 ref :276-284
//   outputs: raw little-endian double .bin / one-int-per-line .txt                  ref :470-491
//   random initial hypers via srand/rand() % 4096                                    ref :418-441, :493-588
// rapidjson (the reference's parser) is replaced by a small reader for the flat object the reference uses.
#pragma once
#include <map>
#include <string>
#include <vector>

namespace medgp {

class c_experiment {
public:
    c_experiment() {}
    // returns false (message in error()) instead of the reference's assert()/exit(1)
    bool load(const std::string &cfg_name);
    const std::string &error() const { return err; }

    std::string get_data_dir() const { return exp_data_dir; }
    std::string get_exp_train_dir() const { return exp_train_dir; }
    std::string get_exp_test_dir() const { return exp_test_dir; }
    std::string get_exp_kernel_dir() const { return exp_kernel_dir; }
    int get_kernel_index() const { return kernel_index; }
    std::vector<int> get_kernel_param() const { return kernel_param; }
    std::vector<int> get_feature_index() const { return feature_index; }
    std::vector<float> get_prior_hyp() const { return prior_hyp; }
    int get_prior_mode() const { return prior_mode; }
    int get_cv_fold_num() const { return cv_fold_num; }
    int get_scg_init_num() const { return scg_init_num; }
    int get_scg_max_iter_num() const { return scg_max_iter_num; }
    int get_prior_sub_opt_iter() const { return prior_sub_opt_iter; }
    double get_online_learn_rate() const { return learn_rate; }
    double get_online_momentum() const { return momentum; }
    int get_hyp_num() const { return get_lik_num() + get_cov_num() + get_mean_num(); }
    int get_cov_num() const;    // ref :311-337
    int get_lik_num() const;    // ref :367-386
    int get_mean_num() const { return 0; }
    const std::vector<double> &lb() const { return hyp_array_lb; }
    const std::vector<double> &ub() const { return hyp_array_ub; }

    // ref :254-309 -- appends feature by feature (=> grouped by output), z-scores with the cohort mean/std
    bool get_one_patient_data(const std::string &PAN, std::vector<int> &meta_vec, std::vector<float> &time_vec,
                              std::vector<float> &value_vec);
    // ref :418-441 -- srand(seed) then scg_init_num draws of the full hyper vector
    void get_global_hyp(std::vector<std::vector<double>> &global_hyp_array);
    // ref :179-219 -- test-time kernel (mode) parameters written by the clustering step
    bool get_test_kernel_param(int fold, const std::string &alg, std::vector<int> &test_kernel_param);
    bool get_test_mode_param(int fold, const std::string &alg, std::vector<double> &mode_param);
    int get_test_cov_num(const std::vector<int> &test_kernel_param) const;

    static bool output_double_bin(const std::string &file_prefix, const std::vector<double> &a);   // ref :470-479
    static bool output_float_bin(const std::string &file_prefix, const std::vector<float> &a);
    static bool output_int_txt(const std::string &file_prefix, const std::vector<int> &a);        // ref :481-491

private:
    double get_one_random(const double &lb, const double &ub, const double &scale, const bool &flag_inv,
                          const bool &flag_log);                                                  // ref :493-517
    void get_hyp_SE(std::vector<double> &h);       // ref :519-530
    void get_hyp_LMC_SM(std::vector<double> &h);   // ref :532-564
    void get_hyp_SM(std::vector<double> &h);       // ref :566-588
    bool get_hyp_bounds();

    std::string err, exp_cfg_file, exp_data_dir, exp_top_dir, exp_train_dir, exp_test_dir, exp_kernel_dir, exp_hyp_bound_file;
    int kernel_index = 7, prior_mode = 0, srand_seed = 0, cv_fold_num = 1, scg_init_num = 0, scg_max_iter_num = 0,
        prior_sub_opt_iter = 0;
    double learn_rate = 0.0, momentum = 0.0;
    std::vector<int> kernel_param, feature_index;
    std::vector<float> prior_hyp;
    std::vector<double> hyp_array_lb, hyp_array_ub;
};

// flat JSON object reader: string / number values only (what config.py:5-35 writes)
struct json_value {
    bool is_string = false, is_number = false, is_int = false;   // is_int: literal without '.', 'e', 'E'
    std::string s;
    double d = 0.0;
};
bool parse_flat_json(const std::string &text, std::map<std::string, json_value> &out, std::string &err);

}  // namespace medgp
