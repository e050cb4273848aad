// medgp_experiment.cpp -- see medgp_experiment.hpp.  "ref:" = /root/reference/medgpc/src/dataio/c_experiment.cpp
#include "medgp_experiment.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace medgp {

static const double REF_PI = 3.14159265;   // ref: util/global_settings.h:6

// ------------------------------------------------------------------------------------------- JSON
bool parse_flat_json(const std::string &t, std::map<std::string, json_value> &out, std::string &err) {
    size_t i = 0;
    auto ws = [&]() { while (i < t.size() && (t[i] == ' ' || t[i] == '\n' || t[i] == '\t' || t[i] == '\r')) i++; };
    auto str = [&](std::string &s) -> bool {
        if (i >= t.size() || t[i] != '"') return false;
        i++;
        s.clear();
        while (i < t.size() && t[i] != '"') {
            if (t[i] == '\\' && i + 1 < t.size()) {
                char c = t[i + 1];
                s.push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c);
                i += 2;
            } else s.push_back(t[i++]);
        }
        if (i >= t.size()) return false;
        i++;
        return true;
    };
    ws();
    if (i >= t.size() || t[i] != '{') { err = "config is not a JSON object"; return false; }
    i++;
    ws();
    if (i < t.size() && t[i] == '}') return true;
    while (true) {
        ws();
        std::string key;
        if (!str(key)) { err = "bad key near offset " + std::to_string(i); return false; }
        ws();
        if (i >= t.size() || t[i] != ':') { err = "':' expected after key " + key; return false; }
        i++;
        ws();
        json_value v;
        if (i < t.size() && t[i] == '"') {
            v.is_string = true;
            if (!str(v.s)) { err = "bad string value for " + key; return false; }
        } else {
            size_t j = i;
            while (j < t.size() && (isdigit((unsigned char)t[j]) || t[j] == '-' || t[j] == '+' || t[j] == '.' || t[j] == 'e' || t[j] == 'E')) j++;
            if (j == i) { err = "unsupported value for key " + key; return false; }
            v.s = t.substr(i, j - i);
            v.is_number = true;
            v.is_int = v.s.find_first_of(".eE") == std::string::npos;
            v.d = atof(v.s.c_str());
            i = j;
        }
        out[key] = v;
        ws();
        if (i < t.size() && t[i] == ',') { i++; continue; }
        if (i < t.size() && t[i] == '}') return true;
        err = "',' or '}' expected after key " + key;
        return false;
    }
}

// ------------------------------------------------------------------------------------------- config
bool c_experiment::load(const std::string &cfg_name) {
    exp_cfg_file = cfg_name;
    std::cout << "read in config. file " << exp_cfg_file << std::endl;
    std::ifstream ifs(cfg_name.c_str());
    if (!ifs) { err = "File " + cfg_name + " could not be opened."; return false; }
    std::stringstream ss;
    ss << ifs.rdbuf();
    std::map<std::string, json_value> d;
    if (!parse_flat_json(ss.str(), d, err)) return false;
    // the reference asserts the JSON type of every key (ref :52-148); report instead of aborting
    auto S = [&](const char *k, std::string &o) { auto it = d.find(k); if (it == d.end() || !it->second.is_string) { err = std::string("config key '") + k + "' must be a string"; return false; } o = it->second.s; return true; };
    auto I = [&](const char *k, int &o) { auto it = d.find(k); if (it == d.end() || !it->second.is_number || !it->second.is_int) { err = std::string("config key '") + k + "' must be an integer"; return false; } o = (int)it->second.d; return true; };
    // rapidjson IsFloat()/IsDouble(): true only for literals parsed as double (an integer literal fails the assert)
    auto F = [&](const char *k, double &o) { auto it = d.find(k); if (it == d.end() || !it->second.is_number || it->second.is_int) { err = std::string("config key '") + k + "' must be a floating-point literal"; return false; } o = it->second.d; return true; };
    if (!S("data_dir", exp_data_dir) || !S("exp_top_dir", exp_top_dir) || !S("exp_train_dir", exp_train_dir) ||
        !S("exp_test_dir", exp_test_dir) || !S("exp_kernel_dir", exp_kernel_dir)) return false;
    exp_data_dir += "/"; exp_top_dir += "/"; exp_train_dir += "/"; exp_test_dir += "/"; exp_kernel_dir += "/";
    int Q, D, R;
    if (!I("kernel_index", kernel_index) || !I("Q", Q) || !I("D", D) || !I("R", R)) return false;
    kernel_param = {Q, D, R};
    for (int i = 0; i < 3; i++) std::cout << "kernel_param[" << i << "] = " << kernel_param[i] << std::endl;
    if (!I("prior_index", prior_mode)) return false;
    if (prior_mode == 2) {
        double eta, bl;
        if (!F("eta", eta) || !F("beta_lam", bl)) return false;
        prior_hyp = {(float)eta, (float)bl};   // GetFloat(): narrowed like the reference (ref :105-109)
    }
    std::string fs;
    if (!S("feature_index", fs)) return false;
    {
        std::istringstream is(fs);
        feature_index.clear();
        for (int k = 0; k < D; k++) { int val = 0; is >> val; feature_index.push_back(val); }
    }
    std::string exp_cfg_dir, hyp_bound_file;
    if (!I("random_seed", srand_seed) || !I("cv_fold_num", cv_fold_num) || !I("random_init_num", scg_init_num) ||
        !I("top_iteration_num", scg_max_iter_num) || !I("iteration_num_per_update", prior_sub_opt_iter) ||
        !F("online_learn_rate", learn_rate) || !F("online_momentum", momentum) || !S("exp_cfg_dir", exp_cfg_dir) ||
        !S("hyp_bound_file", hyp_bound_file)) return false;
    if (kernel_index != 0 && kernel_index != 7 && kernel_index != 8) {
        err = "ERROR: unknown mode for getting covariance parameter for kernel (" + std::to_string(kernel_index) + ")";
        return false;
    }
    exp_hyp_bound_file = exp_cfg_dir + "/" + hyp_bound_file;
    return get_hyp_bounds();
}

int c_experiment::get_cov_num() const {
    const int Q = kernel_param[0], D = kernel_param[1], R = kernel_param[2];
    return kernel_index == 0 ? 2 : (kernel_index == 7 ? Q * (D * R + 2 + D) : 3 * Q);
}
int c_experiment::get_lik_num() const { return kernel_index == 7 ? kernel_param[1] : 1; }
int c_experiment::get_test_cov_num(const std::vector<int> &p) const {
    return kernel_index == 0 ? 2 : (kernel_index == 7 ? p[0] * (p[1] * p[2] + 2 + p[1]) : 3 * p[0]);
}

bool c_experiment::get_hyp_bounds() {
    std::ifstream data(exp_hyp_bound_file.c_str());
    if (!data) { err = "File " + exp_hyp_bound_file + " could not be opened."; return false; }
    hyp_array_lb.clear(); hyp_array_ub.clear();
    for (int i = 0; i < get_hyp_num(); i++) {
        double lbv = 0, ubv = 0;
        data >> lbv; data >> ubv;
        hyp_array_lb.push_back(lbv); hyp_array_ub.push_back(ubv);
    }
    return true;
}

// ------------------------------------------------------------------------------------------- patient data
bool c_experiment::get_one_patient_data(const std::string &PAN, std::vector<int> &meta_vec, std::vector<float> &time_vec,
                                        std::vector<float> &value_vec) {
    meta_vec.clear(); time_vec.clear(); value_vec.clear();
    for (int j = 0; j < (int)feature_index.size(); j++) {
        std::vector<double> stat;
        std::string fn = exp_data_dir + "feature" + std::to_string((long long)feature_index[j]) + "_stat.bin";
        {
            std::ifstream databin(fn, std::ios::binary);
            double f;
            while (databin.read(reinterpret_cast<char *>(&f), sizeof(double))) stat.push_back(f);
        }
        if (stat.size() < 2) { err = "File " + fn + " could not be read (need mean, std)."; return false; }
        fn = exp_data_dir + PAN + "/feature" + std::to_string((long long)feature_index[j]) + ".txt";
        std::ifstream data(fn.c_str());
        if (!data) { err = "File " + fn + " could not be opened."; return false; }
        float vec_len = 0, temp = 0;
        data >> vec_len;
        for (int i = 0; i < (int)vec_len; i++) {
            meta_vec.push_back(j);
            data >> temp;
            time_vec.push_back(temp);
            data >> temp;
            double norm_temp = ((double)temp - stat[0]) / stat[1];
            value_vec.push_back((float)norm_temp);
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------- random initial hypers
double c_experiment::get_one_random(const double &lb, const double &ub, const double &scale, const bool &flag_inv,
                                    const bool &flag_log) {
    const int rand_max = (int)std::floor(std::pow(2.0, 12));   // 4096
    double temp = ((double)(rand() % rand_max)) + 1.0;         // glibc rand(), same sequence as the reference
    temp *= (ub - lb);
    temp = temp / ((double)rand_max);
    double a = scale * (temp + lb);
    if (flag_inv) a = 1.0 / a;
    if (flag_log) a = std::log(a);
    return a;
}
void c_experiment::get_hyp_SE(std::vector<double> &h) {
    for (int i = 0; i < get_hyp_num(); i++)
        h.push_back(get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, i < get_lik_num() + get_cov_num()));
}
void c_experiment::get_hyp_LMC_SM(std::vector<double> &h) {
    const int Q = kernel_param[0], D = kernel_param[1], R = kernel_param[2], nl = get_lik_num();
    for (int i = 0; i < get_hyp_num(); i++) {
        double temp;
        if (i < nl) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, true);
        else if (i < nl + Q * D * R) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 0.9 / std::sqrt((double)Q * (double)R), false, false);
        else if (i < nl + Q * (D * R + 1)) temp = std::log(1.0 / get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, false));
        else if (i < nl + Q * (D * R + 2)) temp = std::log(1.0 / (2 * REF_PI * get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, false)));
        else if (i < nl + Q * (D * R + 2 + D)) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 0.1 / (double)Q, false, true);
        else temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, false);
        h.push_back(temp);
    }
}
void c_experiment::get_hyp_SM(std::vector<double> &h) {
    const int Q = kernel_param[0], nl = get_lik_num();
    for (int i = 0; i < get_hyp_num(); i++) {
        double temp;
        if (i < nl) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, true);
        else if (i < nl + Q) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0 / (double)Q, false, true);
        else if (i < nl + 2 * Q) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, true, true);
        else if (i < nl + 3 * Q) temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 2 * REF_PI, true, true);
        else temp = get_one_random(hyp_array_lb[i], hyp_array_ub[i], 1.0, false, false);
        h.push_back(temp);
    }
}
void c_experiment::get_global_hyp(std::vector<std::vector<double>> &g) {
    std::cout << "generating random hyperparameters..." << std::endl;
    srand(srand_seed);
    for (int i = 0; i < scg_init_num; i++) {
        std::vector<double> h;
        if (kernel_index == 0) get_hyp_SE(h);
        else if (kernel_index == 7) get_hyp_LMC_SM(h);
        else get_hyp_SM(h);
        g.push_back(h);
    }
}

// ------------------------------------------------------------------------------------------- test-time readers
bool c_experiment::get_test_kernel_param(int fold, const std::string &alg, std::vector<int> &p) {
    p = kernel_param;
    if (kernel_index == 7 || kernel_index == 8) {
        std::string fn = exp_kernel_dir + "fold" + std::to_string((long long)fold) + "/" + alg + "_mode_mixture_num.txt";
        std::cout << "read in new mixture number from " << fn << std::endl;
        std::ifstream data(fn.c_str());
        if (!data) { err = "File " + fn + " could not be opened."; return false; }
        int q = 0;
        data >> q;
        p[0] = q;
    }
    return true;
}
bool c_experiment::get_test_mode_param(int fold, const std::string &alg, std::vector<double> &mode_param) {
    std::string fn = exp_kernel_dir + "fold" + std::to_string((long long)fold) + "/" + alg + "_mode_param.bin";
    std::cout << "read in mode parameters from " << fn << std::endl;
    std::ifstream databin(fn, std::ios::binary);
    if (!databin) { err = "File " + fn + " could not be opened."; return false; }
    mode_param.clear();
    double one;
    while (databin.read(reinterpret_cast<char *>(&one), sizeof(double))) mode_param.push_back(one);
    std::cout << "read in " << mode_param.size() << " parameters" << std::endl;
    return true;
}

// ------------------------------------------------------------------------------------------- writers
bool c_experiment::output_double_bin(const std::string &prefix, const std::vector<double> &a) {
    std::ofstream data((prefix + ".bin").c_str(), std::ios::binary);
    if (!data.is_open()) return false;
    for (double v : a) data.write(reinterpret_cast<const char *>(&v), sizeof(double));
    return true;
}
bool c_experiment::output_float_bin(const std::string &prefix, const std::vector<float> &a) {
    std::ofstream data((prefix + ".bin").c_str(), std::ios::binary);
    if (!data.is_open()) return false;
    for (float v : a) data.write(reinterpret_cast<const char *>(&v), sizeof(float));
    return true;
}
bool c_experiment::output_int_txt(const std::string &prefix, const std::vector<int> &a) {
    std::ofstream data((prefix + ".txt").c_str());
    if (!data.is_open()) return false;
    for (int v : a) data << v << "\n";
    return true;
}

}  // namespace medgp
