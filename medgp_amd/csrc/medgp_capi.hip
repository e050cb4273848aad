// medgp_capi.hip -- C ABI of libmedgp_hip.so (see include/medgp_hip.h) and the launch pipeline.
// gfx950 only; there is deliberately no CPU fallback: without a HIP device every compute entry
// point fails with MEDGP_ERR_NODEVICE.
#include "../../include/medgp_hip.h"
#include "medgp_dev.h"
#include "kernels_v0.h"
#include "kernels_cholinv.h"
#include "kernels_cholinv_mc.h"
#include "kernels_assemble.h"
#include "kernels_wgrad.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum KernelId { KID_PREP = 0, KID_ASSEMBLE, KID_POTRF, KID_TRTRI, KID_CHOLINV, KID_CI_PANEL, KID_CI_TRSM, KID_LAUUM, KID_GRADBINS, KID_WGRAD, KID_EPILOGUE, KID_PREDICT, KID_COUNT };
const char *const kKernelNames[KID_COUNT] = {"k_prep", "k_assemble", "k_potrf", "k_trtri", "k_cholinv", "k_ci_panel", "k_ci_trsm", "k_lauum", "k_gradbins", "k_wgrad", "k_epilogue", "k_predict"};

std::string g_create_error;

struct EvPair { int kid; hipEvent_t a, b; };

}  // namespace

struct medgp_ctx {
    int device = -1;
    int kidx = 0, Q = 0, D = 0, R = 0, H = 0, nlik = 0;
    double pi = 3.14159265;   // ref: util/global_settings.h:6
    hipStream_t own_stream = nullptr, stream = nullptr;
    int max_slots = 0, max_n = 0, max_batch = 0, ldn = 0;
    MedgpDev dev{};
    // device allocations
    std::vector<void *> allocs;
    int *d_proff = nullptr, *d_pcoff = nullptr, *d_jit = nullptr;
    int *d_pn = nullptr, *d_pmeta = nullptr, *d_pseg = nullptr, *d_bslot = nullptr, *d_status = nullptr;
    double *d_pt = nullptr, *d_py = nullptr;
    MedgpPrior *d_prior = nullptr;
    uint8_t *d_prior_on = nullptr;
    double *d_theta = nullptr, *d_nlml = nullptr, *d_grad = nullptr;   // staging for the host-pointer API
    int *d_status_out = nullptr;
    // host mirrors
    std::vector<int> h_n;
    std::vector<std::vector<int>> h_perm;   // internal index -> caller index
    std::vector<uint8_t> h_perm_identity;
    std::vector<int> h_bslot;
    int last_nbatch = 0;
    // predict scratch
    double *d_t2 = nullptr, *d_ks = nullptr;
    int *d_meta2 = nullptr;
    float *d_mean = nullptr, *d_var = nullptr;
    int pred_cap = 0;
    // profiling
    bool profiling = false;
    bool use_v0 = false;      // MEDGP_V0=1: baseline kernels (debug / A-B parity)
    int cholinv_nw = 0;       // MEDGP_CHOLINV_NW=44|84 forces the workgroup shape (0 = auto)
    int force_mc = 0;         // MEDGP_MULTI_CU=1 forces / -1 forbids the multi-CU factorisation (0 = auto)
    int num_cu = 256;
    int nsplit = 1;           // MEDGP_STREAMS=2 splits large batches over two streams (measured: 98.9k vs 104.6k evals/s -> off)
    hipStream_t aux[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    std::vector<EvPair> events;
    double prof_ms[KID_COUNT] = {0};
    int64_t prof_n[KID_COUNT] = {0};
    std::string err;
};

namespace {

int fail(medgp_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return fail((c), MEDGP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dalloc(medgp_ctx *c, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) return fail(c, MEDGP_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = (T *)q;
    return MEDGP_OK;
}

void free_all(medgp_ctx *c) {
    for (void *p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
}

int num_cov(int kidx, int Q, int D, int R) {
    switch (kidx) {
    case MEDGP_KERNEL_LMC_SM: return Q * (D * R + 2 + D);
    case MEDGP_KERNEL_SM: return 3 * Q;
    case MEDGP_KERNEL_SE: return 2;
    default: return -1;
    }
}

struct Launcher {
    medgp_ctx *c;
    int kid;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    Launcher(medgp_ctx *c_, int kid_, hipStream_t st_ = nullptr) : c(c_), kid(kid_), st(st_ ? st_ : c_->stream) {
        if (c->profiling) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, st);
        }
    }
    ~Launcher() {
        if (c->profiling) {
            (void)hipEventRecord(b, st);
            if (kid >= 0) c->events.push_back({kid, a, b});
            else { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
        }
    }
};

int drain_events(medgp_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &e : c->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.kid] += ms;
            c->prof_n[e.kid] += 1;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    c->events.clear();
    return MEDGP_OK;
}

int set_batch(medgp_ctx *c, int nbatch, const int32_t *slots, int *max_n_out) {
    if (nbatch < 1 || nbatch > c->max_batch) return fail(c, MEDGP_ERR_CAPACITY, "nbatch %d outside [1, %d]", nbatch, c->max_batch);
    int mx = 0;
    for (int b = 0; b < nbatch; b++) {
        int s = slots[b];
        if (s < 0 || s >= c->max_slots || c->h_n[s] < 0) return fail(c, MEDGP_ERR_ARG, "slots[%d] = %d is not a resident patient", b, s);
        mx = std::max(mx, c->h_n[s]);
    }
    *max_n_out = mx;
    bool same = (nbatch == c->last_nbatch) && std::memcmp(c->h_bslot.data(), slots, sizeof(int) * nbatch) == 0;
    if (!same) {
        std::memcpy(c->h_bslot.data(), slots, sizeof(int) * nbatch);
        HIPCHK(c, hipMemcpyAsync(c->d_bslot, c->h_bslot.data(), sizeof(int) * nbatch, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // h_bslot may be rewritten by the next call
        c->last_nbatch = nbatch;
    }
    return MEDGP_OK;
}

inline int tri(int n) { return n * (n + 1) / 2; }

// one kernel chain for the batch entries described by L (possibly a shifted view of c->dev) on `stream`
int run_pipeline_one(medgp_ctx *c, hipStream_t stream, const MedgpDev &L, int nbatch, int max_n, const double *theta_dev,
                     int flag_grad, bool need_inverse, int min_n, double *nlml_dev, double *grad_dev, int32_t *status_dev) {
    const int nt64 = medgp_roundup(std::max(max_n, 1), 64) / 64;
    { Launcher l(c, KID_PREP, stream); hipLaunchKernelGGL(k_prep, dim3(nbatch), dim3(256), 0, stream, L, theta_dev, min_n); }
    auto launch_assemble = [&]() {
        Launcher l(c, KID_ASSEMBLE, stream);
        const dim3 tg(tri(nt64), nbatch), tb(256);
        switch (c->use_v0 ? 0 : L.Q) {
        case 1: hipLaunchKernelGGL(k_assemble_t<1>, tg, tb, 0, stream, L); break;
        case 2: hipLaunchKernelGGL(k_assemble_t<2>, tg, tb, 0, stream, L); break;
        case 3: hipLaunchKernelGGL(k_assemble_t<3>, tg, tb, 0, stream, L); break;
        case 4: hipLaunchKernelGGL(k_assemble_t<4>, tg, tb, 0, stream, L); break;
        case 5: hipLaunchKernelGGL(k_assemble_t<5>, tg, tb, 0, stream, L); break;
        case 6: hipLaunchKernelGGL(k_assemble_t<6>, tg, tb, 0, stream, L); break;
        case 7: hipLaunchKernelGGL(k_assemble_t<7>, tg, tb, 0, stream, L); break;
        case 8: hipLaunchKernelGGL(k_assemble_t<8>, tg, tb, 0, stream, L); break;
        default: hipLaunchKernelGGL(k_assemble_v0, tg, tb, 0, stream, L); break;
        }
    };
    const bool inv = flag_grad || need_inverse;
    // few large patients: one workgroup per 64-row block and two launches per panel (kernels_cholinv_mc.h)
    // measured on MI355X (D=24, factorisation ms, multi-CU vs one 8-wave workgroup per patient; scratch/quick_shapes.py):
    // N=512: 64 patients 0.60 vs 0.84, 128: 0.80 vs 0.89, 192: 1.18 vs 0.93; N=1024: 64: 2.0 vs 4.8, 128: 3.9 vs 4.9,
    // 200: 6.0 vs 5.1; N=2048: 16: 4.4 vs 32, 64: 11.4 vs 33.5.  The multi-CU time grows linearly with the batch, the
    // single-workgroup time is flat up to one patient per CU: the crossover sits near 0.6 #CU for every N >= 512.
    const bool multi_cu = !c->use_v0 && (c->force_mc > 0 || (c->force_mc == 0 && nt64 >= 2 && nbatch <= (c->num_cu * 3) / 5));
    if (c->use_v0) {
        launch_assemble();
        { Launcher l(c, KID_POTRF, stream); hipLaunchKernelGGL(k_potrf_v0, dim3(nbatch), dim3(256), 0, stream, L); }
        if (inv) { Launcher l(c, KID_TRTRI, stream); hipLaunchKernelGGL(k_trtri_v0, dim3(nbatch), dim3(256), 0, stream, L); }
    } else if (multi_cu) {
        std::vector<int> hst(nbatch), hjit(nbatch, 0);
        for (int attempt = 0;; attempt++) {
            launch_assemble();
            for (int k = 0; k < nt64; k++) {
                { Launcher l(c, KID_CI_PANEL, stream); hipLaunchKernelGGL(k_ci_panel, dim3(nbatch, nt64), dim3(MC_THREADS), 0, stream, L, k, inv ? 1 : 0); }
                if (nt64 > 1) { Launcher l(c, KID_CI_TRSM, stream); hipLaunchKernelGGL(k_ci_trsm, dim3(nt64 - 1, nbatch), dim3(MC_THREADS), 0, stream, L, k, inv ? 1 : 0); }
            }
            hipLaunchKernelGGL(k_ci_finish, dim3(nbatch), dim3(256), 0, stream, L);
            HIPCHK(c, hipMemcpyAsync(hst.data(), L.status, sizeof(int) * nbatch, hipMemcpyDeviceToHost, stream));
            HIPCHK(c, hipStreamSynchronize(stream));
            bool retry = false;
            for (int bb = 0; bb < nbatch; bb++)
                if (hst[bb] == -2) {   // ref c_inference_exact.cpp:99-111: add the noise again, at most 10 times
                    if (hjit[bb] >= 10) hst[bb] = -1;
                    else { hjit[bb]++; hst[bb] = 0; retry = true; }
                } else if (hst[bb] >= 0) hst[bb] = 0;
            if (!retry) {
                bool fix = false;
                for (int bb = 0; bb < nbatch; bb++) fix = fix || (hjit[bb] >= 10 && hst[bb] == -1);
                if (fix) {   // final failures: publish -1
                    std::vector<int> cur(nbatch);
                    HIPCHK(c, hipMemcpy(cur.data(), L.status, sizeof(int) * nbatch, hipMemcpyDeviceToHost));
                    for (int bb = 0; bb < nbatch; bb++) if (cur[bb] == -2) cur[bb] = -1;
                    HIPCHK(c, hipMemcpy(L.status, cur.data(), sizeof(int) * nbatch, hipMemcpyHostToDevice));
                }
                break;
            }
            // failed problems restart with one more noise addition; finished ones are recomputed identically
            std::vector<int> cur(nbatch);
            HIPCHK(c, hipMemcpy(cur.data(), L.status, sizeof(int) * nbatch, hipMemcpyDeviceToHost));
            for (int bb = 0; bb < nbatch; bb++) if (cur[bb] == -2 || cur[bb] >= 0) cur[bb] = (hst[bb] == -1) ? -1 : 0;
            HIPCHK(c, hipMemcpy(L.status, cur.data(), sizeof(int) * nbatch, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(L.jit, hjit.data(), sizeof(int) * nbatch, hipMemcpyHostToDevice));
            // logdet accumulators restart
            std::vector<double> zero4(4 * (size_t)nbatch, 0.0);
            HIPCHK(c, hipMemcpy(L.scal, zero4.data(), sizeof(double) * 4 * nbatch, hipMemcpyHostToDevice));
            (void)attempt;
        }
    } else {
        launch_assemble();
        Launcher l(c, KID_CHOLINV, stream);
        // more patients than CUs: 4-wave workgroups, two per CU (the serial diagonal phase of one overlaps the
        // MFMA phase of the other); else 8 waves for the lowest latency per patient
        const int shape = c->cholinv_nw ? c->cholinv_nw : (nbatch > c->num_cu ? 44 : 84);
        if (shape == 44) hipLaunchKernelGGL((k_cholinv<4, 4>), dim3(nbatch), dim3(256), 0, stream, L, inv ? 1 : 0);
        else hipLaunchKernelGGL((k_cholinv<8, 4>), dim3(nbatch), dim3(512), 0, stream, L, inv ? 1 : 0);
    }
    int from_slab = 0;
#ifdef MEDGP_STAMPS
    if (getenv("MEDGP_DBG_NOWGRAD")) { HIPCHK(c, hipGetLastError()); return MEDGP_OK; }
#endif
    if (flag_grad) {
        const int wg_tiles = tri(nt64);
        const dim3 tg(8 * ((nbatch + 7) / 8) * wg_tiles), tb(WG_THREADS);
        from_slab = 1;
        Launcher *lw = new Launcher(c, KID_WGRAD, stream);
        switch (c->use_v0 ? 0 : L.Q) {
        case 1: hipLaunchKernelGGL(k_wgrad<1>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 2: hipLaunchKernelGGL(k_wgrad<2>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 3: hipLaunchKernelGGL(k_wgrad<3>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 4: hipLaunchKernelGGL(k_wgrad<4>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 5: hipLaunchKernelGGL(k_wgrad<5>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 6: hipLaunchKernelGGL(k_wgrad<6>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 7: hipLaunchKernelGGL(k_wgrad<7>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        case 8: hipLaunchKernelGGL(k_wgrad<8>, tg, tb, 0, stream, L, nbatch, wg_tiles); break;
        default: from_slab = 0; break;   // Q > 8 (or MEDGP_V0): generic kernels below
        }
        if (from_slab) delete lw; else { lw->kid = -1; delete lw; }
        if (!from_slab) {
            { Launcher l(c, KID_LAUUM, stream); hipLaunchKernelGGL(k_lauum_v0, dim3(tri(nt64), nbatch), dim3(256), 0, stream, L); }
            const int nbins = L.Q * tri(L.D);
            { Launcher l(c, KID_GRADBINS, stream); hipLaunchKernelGGL(k_gradbins_v0, dim3((nbins + 255) / 256, nbatch), dim3(256), 0, stream, L); }
        }
    }
    if (nlml_dev) {
        Launcher l(c, KID_EPILOGUE, stream);
        hipLaunchKernelGGL(k_epilogue, dim3(nbatch), dim3(256), 0, stream, L, theta_dev, flag_grad, from_slab, nlml_dev, grad_dev, (int *)status_dev);
    }
    HIPCHK(c, hipGetLastError());
    return MEDGP_OK;
}


// view of the batch-indexed buffers starting at entry b0
MedgpDev shifted_view(const MedgpDev &L, int b0) {
    MedgpDev V = L;
    const size_t ld = L.ldn, Q = L.Q, D = L.D;
    V.bslot = L.bslot + b0;
    V.hyp = L.hyp + (size_t)b0 * L.hyp_stride;
    V.cs = L.cs + (size_t)b0 * Q * ld; V.sn = L.sn + (size_t)b0 * Q * ld;
    V.Kmat = L.Kmat + (size_t)b0 * ld * ld; V.Linv = L.Linv + (size_t)b0 * ld * ld;
    V.z = L.z + (size_t)b0 * ld; V.alpha = L.alpha + (size_t)b0 * ld; V.wdiag = L.wdiag + (size_t)b0 * ld;
    V.scal = L.scal + (size_t)b0 * 4; V.status = L.status + b0; V.jit = L.jit + b0; V.xk = L.xk + (size_t)b0 * 64 * 64;
    V.S = L.S + (size_t)b0 * Q * D * D; V.SM = L.SM + (size_t)b0 * Q * D * D; V.SV = L.SV + (size_t)b0 * Q * D * D;
    V.slab = L.slab + (size_t)b0 * L.slab_stride;
    return V;
}

// The evaluation pipeline; everything is asynchronous on c->stream.  Optional (MEDGP_STREAMS=2): large batches split
// in two halves on two auxiliary streams so that k_cholinv of one half co-runs with the VALU-heavy kernels of the
// other half.  Measured on MI355X at the headline shape: no gain (98.9k vs 104.6k evals/s), so it is off by default.
int run_pipeline(medgp_ctx *c, int nbatch, int max_n, const double *theta_dev, int flag_grad, bool need_inverse, int min_n,
                 double *nlml_dev, double *grad_dev, int32_t *status_dev) {
    const bool split = !c->use_v0 && c->nsplit >= 2 && nbatch >= 2 * c->num_cu && c->aux[0] && c->aux[1];
    if (!split) return run_pipeline_one(c, c->stream, c->dev, nbatch, max_n, theta_dev, flag_grad, need_inverse, min_n, nlml_dev, grad_dev, status_dev);
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    const int h = (nbatch / 2 + 1) & ~1;   // even: keeps the (b & 1) wave mirroring of k_cholinv consistent
    const int b0[2] = {0, h}, nb[2] = {h, nbatch - h};
    for (int i = 0; i < 2; i++) {
        HIPCHK(c, hipStreamWaitEvent(c->aux[i], c->ev_fork, 0));
        MedgpDev V = shifted_view(c->dev, b0[i]);
        int rc = run_pipeline_one(c, c->aux[i], V, nb[i], max_n, theta_dev + (size_t)b0[i] * c->H, flag_grad, need_inverse, min_n,
                                  nlml_dev ? nlml_dev + b0[i] : nullptr, grad_dev ? grad_dev + (size_t)b0[i] * c->H : nullptr,
                                  status_dev ? status_dev + b0[i] : nullptr);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev_join[i], c->aux[i]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[i], 0));
    }
    return MEDGP_OK;
}

}  // namespace

extern "C" {

int medgp_abi_version(void) { return 1; }

int medgp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *medgp_last_error(const medgp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int medgp_create(medgp_ctx **out, int device, int kernel_index, int Q, int D, int R) {
    if (!out) return fail(nullptr, MEDGP_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (kernel_index == MEDGP_KERNEL_SE) { Q = 1; D = 1; R = 0; }
    if (kernel_index == MEDGP_KERNEL_SM) { D = 1; R = 0; }
    int nc = num_cov(kernel_index, Q, D, R);
    if (nc < 0) return fail(nullptr, MEDGP_ERR_ARG, "unsupported kernel_index %d (supported: 0 SE, 7 LMC-SM, 8 SM)", kernel_index);
    if (Q < 1 || D < 1 || R < 0) return fail(nullptr, MEDGP_ERR_ARG, "bad Q/D/R = %d/%d/%d", Q, D, R);
    if (D > MEDGP_MAX_D) return fail(nullptr, MEDGP_ERR_ARG, "D = %d exceeds the supported %d outputs", D, MEDGP_MAX_D);
    int ndev = medgp_device_count();
    if (ndev <= 0) return fail(nullptr, MEDGP_ERR_NODEVICE, "no HIP device visible; libmedgp_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, MEDGP_ERR_ARG, "device %d outside [0, %d)", device, ndev);
    medgp_ctx *c = new medgp_ctx();
    c->device = device;
    c->kidx = kernel_index; c->Q = Q; c->D = D; c->R = R;
    c->nlik = (kernel_index == MEDGP_KERNEL_LMC_SM) ? D : 1;
    c->H = c->nlik + nc;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, MEDGP_ERR_HIP, "cannot initialise device %d", device);
    }
    c->stream = c->own_stream;
    { const char *e = getenv("MEDGP_V0"); c->use_v0 = e && e[0] == '1'; }
    { const char *e = getenv("MEDGP_CHOLINV_NW"); c->cholinv_nw = e ? atoi(e) : 0; }
    { const char *e = getenv("MEDGP_MULTI_CU"); c->force_mc = e ? atoi(e) : 0; }
    { const char *e = getenv("MEDGP_STREAMS"); c->nsplit = e ? atoi(e) : 1; }
    for (int i = 0; i < 2; i++) {
        (void)hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
    }
    (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) c->num_cu = pr.multiProcessorCount; }
    *out = c;
    return MEDGP_OK;
}

void medgp_destroy(medgp_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    free_all(c);
    for (int i = 0; i < 2; i++) {
        if (c->aux[i]) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); }
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int medgp_num_hyp(const medgp_ctx *c) { return c ? c->H : MEDGP_ERR_ARG; }

int medgp_set_pi(medgp_ctx *c, double pi) {
    if (!c || !(pi > 0)) return MEDGP_ERR_ARG;
    c->pi = pi;
    c->dev.pi = pi;
    return MEDGP_OK;
}

int medgp_set_stream(medgp_ctx *c, void *s) {
    if (!c) return MEDGP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return MEDGP_OK;
}

int medgp_synchronize(medgp_ctx *c) {
    if (!c) return MEDGP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MEDGP_OK;
}

int medgp_reserve(medgp_ctx *c, int max_slots, int max_n, int max_batch) {
    if (!c) return MEDGP_ERR_ARG;
    if (max_slots < 1 || max_n < 1 || max_batch < 1) return fail(c, MEDGP_ERR_ARG, "medgp_reserve: non-positive capacity");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_all(c);
    c->max_slots = max_slots; c->max_n = max_n; c->max_batch = max_batch;
    // leading dimension = padded n.  (Padding it off the power of two was measured: no effect -- the HBM channel hash
    // already spreads the 4096-byte row stride.)
    const int ldn = medgp_roundup(max_n, 64);
    c->ldn = ldn;
    const size_t S = max_slots, B = max_batch, Q = c->Q, D = c->D, H = c->H;
    int rc;
    if ((rc = dalloc(c, &c->d_pn, S))) return rc;
    if ((rc = dalloc(c, &c->d_pt, S * ldn))) return rc;
    if ((rc = dalloc(c, &c->d_py, S * ldn))) return rc;
    if ((rc = dalloc(c, &c->d_pmeta, S * ldn))) return rc;
    if ((rc = dalloc(c, &c->d_pseg, S * (D + 1)))) return rc;
    if ((rc = dalloc(c, &c->d_proff, S * (D + 1)))) return rc;
    if ((rc = dalloc(c, &c->d_pcoff, S * (D + 1)))) return rc;
    if ((rc = dalloc(c, &c->d_prior, S * H))) return rc;
    if ((rc = dalloc(c, &c->d_prior_on, S))) return rc;
    if ((rc = dalloc(c, &c->d_bslot, B))) return rc;
    if ((rc = dalloc(c, &c->d_status, B))) return rc;
    if ((rc = dalloc(c, &c->d_jit, B))) return rc;
    if ((rc = dalloc(c, &c->d_theta, B * H))) return rc;
    if ((rc = dalloc(c, &c->d_nlml, B))) return rc;
    if ((rc = dalloc(c, &c->d_grad, B * H))) return rc;
    if ((rc = dalloc(c, &c->d_status_out, B))) return rc;
    MedgpDev &L = c->dev;
    L.kidx = c->kidx; L.Q = c->Q; L.D = c->D; L.R = c->R; L.H = c->H; L.nlik = c->nlik;
    L.ldn = ldn; L.max_slots = max_slots; L.max_batch = max_batch;
    L.hyp_stride = (int)(D + Q * D * D + 2 * Q);
    L.pi = c->pi;
    double *hyp, *cs, *sn, *Kmat, *Linv, *z, *alpha, *scal, *Sb, *SMb, *SVb, *slab, *wdiag;
    L.slab_R = ldn / 16 + (int)D;
    L.slab_C = ldn / 64 + (int)D;
    L.slab_stride = (size_t)3 * Q * L.slab_R * L.slab_C;
    if ((rc = dalloc(c, &slab, B * L.slab_stride))) return rc;
    if ((rc = dalloc(c, &wdiag, B * ldn))) return rc;
    double *xk;
    if ((rc = dalloc(c, &xk, B * 64 * 64))) return rc;
    L.xk = xk; L.jit = c->d_jit;
    if ((rc = dalloc(c, &hyp, B * L.hyp_stride))) return rc;
    if ((rc = dalloc(c, &cs, B * Q * ldn))) return rc;
    if ((rc = dalloc(c, &sn, B * Q * ldn))) return rc;
    if ((rc = dalloc(c, &Kmat, B * ldn * ldn))) return rc;
    if ((rc = dalloc(c, &Linv, B * ldn * ldn))) return rc;
    if ((rc = dalloc(c, &z, B * ldn))) return rc;
    if ((rc = dalloc(c, &alpha, B * ldn))) return rc;
    if ((rc = dalloc(c, &scal, B * 4))) return rc;
    if ((rc = dalloc(c, &Sb, B * Q * D * D))) return rc;
    if ((rc = dalloc(c, &SMb, B * Q * D * D))) return rc;
    if ((rc = dalloc(c, &SVb, B * Q * D * D))) return rc;
    L.pn = c->d_pn; L.pt = c->d_pt; L.py = c->d_py; L.pmeta = c->d_pmeta; L.pseg = c->d_pseg;
    L.proff = c->d_proff; L.pcoff = c->d_pcoff; L.slab = slab; L.wdiag = wdiag;
    L.prior = c->d_prior; L.prior_on = c->d_prior_on; L.bslot = c->d_bslot;
    L.hyp = hyp; L.cs = cs; L.sn = sn; L.Kmat = Kmat; L.Linv = Linv; L.z = z; L.alpha = alpha; L.scal = scal;
    L.status = c->d_status; L.S = Sb; L.SM = SMb; L.SV = SVb;
    HIPCHK(c, hipMemsetAsync(c->d_prior_on, 0, S, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_pn, 0, S * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_n.assign(max_slots, -1);
    c->h_perm.assign(max_slots, {});
    c->h_perm_identity.assign(max_slots, 1);
    c->h_bslot.assign(max_batch, -1);
    c->last_nbatch = 0;
    c->pred_cap = 0;
    return MEDGP_OK;
}

int medgp_set_patient(medgp_ctx *c, int slot, int n, const int32_t *meta, const float *t, const float *y) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (slot < 0 || slot >= c->max_slots) return fail(c, MEDGP_ERR_CAPACITY, "slot %d outside [0, %d)", slot, c->max_slots);
    if (n < 0 || n > c->max_n) return fail(c, MEDGP_ERR_CAPACITY, "n = %d outside [0, %d]", n, c->max_n);
    if (n > 0 && (!t || !y)) return fail(c, MEDGP_ERR_ARG, "t / y is NULL");
    const int D = c->D, ldn = c->ldn;
    const bool use_meta = (c->kidx == MEDGP_KERNEL_LMC_SM);
    if (use_meta && n > 0 && !meta) return fail(c, MEDGP_ERR_ARG, "meta is NULL for the multi-output kernel");
    // stable grouping by output (counting sort)
    std::vector<int> seg(D + 1, 0), perm(n);
    if (use_meta) {
        for (int i = 0; i < n; i++) {
            if (meta[i] < 0 || meta[i] >= D) return fail(c, MEDGP_ERR_ARG, "meta[%d] = %d outside [0, %d)", i, meta[i], D);
            seg[meta[i] + 1]++;
        }
        for (int d = 0; d < D; d++) seg[d + 1] += seg[d];
        std::vector<int> pos(seg.begin(), seg.end() - 1);
        for (int i = 0; i < n; i++) perm[pos[meta[i]]++] = i;
    } else {
        seg[1] = n;
        for (int i = 0; i < n; i++) perm[i] = i;
    }
    bool ident = true;
    for (int i = 0; i < n; i++) ident = ident && (perm[i] == i);
    std::vector<double> ht(ldn, 0.0), hy(ldn, 0.0);
    std::vector<int> hm(ldn, 0);
    for (int i = 0; i < n; i++) {
        ht[i] = (double)t[perm[i]];
        hy[i] = (double)y[perm[i]];   // zero mean: ref mean/c_meanfunc_zero.cpp:32-50
        hm[i] = use_meta ? meta[perm[i]] : 0;
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->d_pt + (size_t)slot * ldn, ht.data(), sizeof(double) * ldn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_py + (size_t)slot * ldn, hy.data(), sizeof(double) * ldn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pmeta + (size_t)slot * ldn, hm.data(), sizeof(int) * ldn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pseg + (size_t)slot * (D + 1), seg.data(), sizeof(int) * (D + 1), hipMemcpyHostToDevice, c->stream));
    // slab slots of k_wgrad: 16-row pieces and 64-column pieces each output's segment intersects
    std::vector<int> roff(D + 1, 0), coff(D + 1, 0);
    for (int d = 0; d < D; d++) {
        const int a = seg[d], e = seg[d + 1];
        roff[d + 1] = roff[d] + (e > a ? (e - 1) / 16 - a / 16 + 1 : 0);
        coff[d + 1] = coff[d] + (e > a ? (e - 1) / 64 - a / 64 + 1 : 0);
    }
    HIPCHK(c, hipMemcpyAsync(c->d_proff + (size_t)slot * (D + 1), roff.data(), sizeof(int) * (D + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pcoff + (size_t)slot * (D + 1), coff.data(), sizeof(int) * (D + 1), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_pn + slot, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_n[slot] = n;
    c->h_perm[slot] = perm;
    c->h_perm_identity[slot] = ident ? 1 : 0;
    return MEDGP_OK;
}

int medgp_set_prior(medgp_ctx *c, int slot, const uint8_t *flag, const int32_t *type, const uint8_t *is_exp,
                    const float *p0, const float *p1) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (slot < -1 || slot >= c->max_slots) return fail(c, MEDGP_ERR_CAPACITY, "slot %d outside [-1, %d)", slot, c->max_slots);
    if (flag && (!type || !is_exp || !p0 || !p1)) return fail(c, MEDGP_ERR_ARG, "prior arrays must all be given");
    const int H = c->H;
    std::vector<MedgpPrior> hp(H);
    for (int h = 0; h < H; h++) {
        MedgpPrior p{};
        if (flag) {
            if (type[h] < -1 || type[h] > 2) return fail(c, MEDGP_ERR_ARG, "prior type[%d] = %d unsupported (KDE prior type 3 is never constructed by the reference's mains)", h, type[h]);
            p.p0 = p0[h]; p.p1 = p1[h]; p.type = (int8_t)type[h]; p.flag = flag[h] ? 1 : 0; p.is_exp = is_exp[h] ? 1 : 0;
        } else { p.type = -1; }
        hp[h] = p;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const uint8_t on = flag ? 1 : 0;
    const int s0 = slot < 0 ? 0 : slot, s1 = slot < 0 ? c->max_slots : slot + 1;
    for (int s = s0; s < s1; s++) {
        HIPCHK(c, hipMemcpyAsync(c->d_prior + (size_t)s * H, hp.data(), sizeof(MedgpPrior) * H, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_prior_on + s, &on, 1, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MEDGP_OK;
}

int medgp_nlml_grad_device(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta_dev, int flag_grad,
                           double *nlml_dev, double *grad_dev, int32_t *status_dev) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta_dev || !nlml_dev) return fail(c, MEDGP_ERR_ARG, "NULL argument");
    if (flag_grad && !grad_dev) return fail(c, MEDGP_ERR_ARG, "grad is NULL with flag_grad set");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    HIPCHK(c, hipSetDevice(c->device));
    int max_n = 0, rc;
    if ((rc = set_batch(c, nbatch, slots, &max_n))) return rc;
    return run_pipeline(c, nbatch, max_n, theta_dev, flag_grad, false, 3, nlml_dev, grad_dev, status_dev);
}

int medgp_nlml_grad(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta, int flag_grad, double *nlml,
                    double *grad, int32_t *status) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta || !nlml) return fail(c, MEDGP_ERR_ARG, "NULL argument");
    if (flag_grad && !grad) return fail(c, MEDGP_ERR_ARG, "grad is NULL with flag_grad set");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (nbatch < 1 || nbatch > c->max_batch) return fail(c, MEDGP_ERR_CAPACITY, "nbatch %d outside [1, %d]", nbatch, c->max_batch);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t H = c->H;
    HIPCHK(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * nbatch * H, hipMemcpyHostToDevice, c->stream));
    int rc = medgp_nlml_grad_device(c, nbatch, slots, c->d_theta, flag_grad, c->d_nlml, c->d_grad, c->d_status_out);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(nlml, c->d_nlml, sizeof(double) * nbatch, hipMemcpyDeviceToHost, c->stream));
    if (flag_grad) HIPCHK(c, hipMemcpyAsync(grad, c->d_grad, sizeof(double) * nbatch * H, hipMemcpyDeviceToHost, c->stream));
    if (status) HIPCHK(c, hipMemcpyAsync(status, c->d_status_out, sizeof(int32_t) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MEDGP_OK;
}

int medgp_get_factor(medgp_ctx *c, int b, float *alpha, float *linv, float *beta) {
    if (!c) return MEDGP_ERR_ARG;
    if (b < 0 || b >= c->last_nbatch) return fail(c, MEDGP_ERR_ARG, "batch entry %d outside the last call's [0, %d)", b, c->last_nbatch);
    HIPCHK(c, hipSetDevice(c->device));
    const int slot = c->h_bslot[b], n = c->h_n[slot], ld = c->ldn;
    const std::vector<int> &perm = c->h_perm[slot];
    int st = 0;
    HIPCHK(c, hipMemcpyAsync(&st, c->d_status + b, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st < 0) return fail(c, MEDGP_ERR_ARG, "batch entry %d failed (status %d); no factor available", b, st);
    if (alpha) {
        std::vector<double> ha(n);
        HIPCHK(c, hipMemcpy(ha.data(), c->dev.alpha + (size_t)b * ld, sizeof(double) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) alpha[perm[i]] = (float)ha[i];
    }
    if (beta) {
        double sc[4];
        HIPCHK(c, hipMemcpy(sc, c->dev.scal + (size_t)b * 4, sizeof(sc), hipMemcpyDeviceToHost));
        *beta = (float)sc[1];
    }
    if (linv) {
        if (!c->h_perm_identity[slot])
            return fail(c, MEDGP_ERR_ARG, "L^-1 is only exported for patients already grouped by output (the reference loader's order)");
        std::vector<double> hx((size_t)n * ld);
        HIPCHK(c, hipMemcpy(hx.data(), c->dev.Linv + (size_t)b * ld * ld, sizeof(double) * n * ld, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)   // device holds U = L^-T: (L^-1)[i][j] = U[j][i]; strict upper zeroed as ref c_inference_exact.cpp:139-143
                linv[(size_t)i * n + j] = (j <= i) ? (float)hx[(size_t)j * ld + i] : 0.0f;
    }
    return MEDGP_OK;
}

// common part of medgp_fit_predict / medgp_fit_predict_batch: nbatch problems x nstar test points each
static int fit_predict_impl(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta, int nstar,
                            const int32_t *meta2, const float *t2, float *mean, float *var, int32_t *status) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta || !t2 || !mean || !var || nstar < 1 || nbatch < 1) return fail(c, MEDGP_ERR_ARG, "bad argument");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (c->kidx == MEDGP_KERNEL_LMC_SM && !meta2) return fail(c, MEDGP_ERR_ARG, "meta2 is NULL for the multi-output kernel");
    HIPCHK(c, hipSetDevice(c->device));
    const int ntot = nbatch * nstar;
    if (ntot > c->pred_cap) {
        int cap = std::max(ntot, 64), rc;
        if ((rc = dalloc(c, &c->d_t2, cap))) return rc;
        if ((rc = dalloc(c, &c->d_meta2, cap))) return rc;
        if ((rc = dalloc(c, &c->d_mean, cap))) return rc;
        if ((rc = dalloc(c, &c->d_var, cap))) return rc;
        if ((rc = dalloc(c, &c->d_ks, (size_t)cap * c->ldn))) return rc;
        c->pred_cap = cap;
    }
    int max_n = 0, rc;
    if ((rc = set_batch(c, nbatch, slots, &max_n))) return rc;
    std::vector<double> ht2(ntot);
    std::vector<int> hm2(ntot, 0);
    for (int j = 0; j < ntot; j++) {
        ht2[j] = (double)t2[j];
        if (meta2 && c->kidx == MEDGP_KERNEL_LMC_SM) {
            if (meta2[j] < 0 || meta2[j] >= c->D) return fail(c, MEDGP_ERR_ARG, "meta2[%d] = %d outside [0, %d)", j, meta2[j], c->D);
            hm2[j] = meta2[j];
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * c->H * nbatch, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_t2, ht2.data(), sizeof(double) * ntot, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_meta2, hm2.data(), sizeof(int) * ntot, hipMemcpyHostToDevice, c->stream));
    if ((rc = run_pipeline(c, nbatch, max_n, c->d_theta, 0, true, 1, nullptr, nullptr, nullptr))) return rc;
    {
        Launcher l(c, KID_PREDICT);
        hipLaunchKernelGGL(k_predict_v0, dim3(nstar, nbatch), dim3(256), 0, c->stream, c->dev, nstar, c->d_meta2, c->d_t2, c->d_ks, c->d_mean, c->d_var);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(mean, c->d_mean, sizeof(float) * ntot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(var, c->d_var, sizeof(float) * ntot, hipMemcpyDeviceToHost, c->stream));
    if (status) HIPCHK(c, hipMemcpyAsync(status, c->d_status, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MEDGP_OK;
}

int medgp_fit_predict(medgp_ctx *c, int slot, const double *theta, int nstar, const int32_t *meta2, const float *t2,
                      float *mean, float *var, int32_t *status) {
    int32_t s1 = slot;
    return fit_predict_impl(c, 1, &s1, theta, nstar, meta2, t2, mean, var, status);
}

int medgp_fit_predict_batch(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta, const int32_t *meta2,
                            const float *t2, float *mean, float *var, int32_t *status) {
    if (c && nbatch > c->max_batch) return fail(c, MEDGP_ERR_CAPACITY, "nbatch %d outside [1, %d]", nbatch, c->max_batch);
    return fit_predict_impl(c, nbatch, slots, theta, 1, meta2, t2, mean, var, status);
}

#ifdef MEDGP_STAMPS
// diagnostic build only: read (and clear) the phase counters of diag_factor_wave
#ifdef MEDGP_STAMPS
extern "C" int medgp_debug_read_diag(unsigned long long *out) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_dbg), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_diag_dbg), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
// diagnostic build only: copy out the stamp words k_cholinv left in the slab of batch entry b
int medgp_debug_read_slab(medgp_ctx *c, int b, void *out, int nbytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dev.slab + (size_t)b * c->dev.slab_stride, nbytes, hipMemcpyDeviceToHost));
    return 0;
}
int medgp_debug_read_xk(medgp_ctx *c, int b, void *out, int nbytes, int clear) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->dev.xk + (size_t)b * 64 * 64, nbytes, hipMemcpyDeviceToHost));
    if (clear) HIPCHK(c, hipMemset(c->dev.xk + (size_t)b * 64 * 64, 0, nbytes));
    return 0;
}
#endif

int medgp_profile_enable(medgp_ctx *c, int enable) {
    if (!c) return MEDGP_ERR_ARG;
    if (!enable && c->profiling) { int rc = drain_events(c); if (rc) return rc; }
    c->profiling = enable != 0;
    return MEDGP_OK;
}
int medgp_profile_num_kernels(void) { return KID_COUNT; }
const char *medgp_profile_kernel_name(int k) { return (k >= 0 && k < KID_COUNT) ? kKernelNames[k] : ""; }
int medgp_profile_read(medgp_ctx *c, int k, double *ms_total, int64_t *launches) {
    if (!c || k < 0 || k >= KID_COUNT) return MEDGP_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drain_events(c);
    if (rc) return rc;
    if (ms_total) *ms_total = c->prof_ms[k];
    if (launches) *launches = c->prof_n[k];
    return MEDGP_OK;
}
int medgp_profile_reset(medgp_ctx *c) {
    if (!c) return MEDGP_ERR_ARG;
    int rc = drain_events(c);
    if (rc) return rc;
    for (int k = 0; k < KID_COUNT; k++) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return MEDGP_OK;
}

}  // extern "C"
