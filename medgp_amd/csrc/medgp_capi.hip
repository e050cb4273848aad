// medgp_capi.hip -- C ABI of libmedgp_hip.so (see include/medgp_hip.h) and the launch pipeline.
// gfx950 only; there is deliberately no CPU fallback: without a HIP device every compute entry
// point fails with MEDGP_ERR_NODEVICE.
#include "../../include/medgp_hip.h"
#include "medgp_dev.h"
#include "kernels_core.h"
#include "kernels_v0.h"
#include "kernels_cholinv.h"
#include "kernels_cholinv_la.h"
#include "kernels_assemble.h"
#include "kernels_wgrad.h"
#include "kernels_io.h"
#include "kernels_cohort.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum KernelId { KID_PREP = 0, KID_ASSEMBLE, KID_CHOLINV, KID_LA_STEP, KID_LA_AUX, KID_LAUUM, KID_GRADBINS, KID_WGRAD, KID_EPILOGUE, KID_PREDICT, KID_COUNT };
const char *const kKernelNames[KID_COUNT] = {"k_prep", "k_assemble", "k_cholinv", "k_la_step", "k_la_aux", "k_lauum", "k_gradbins", "k_wgrad", "k_epilogue", "k_predict"};

thread_local std::string g_create_error;   // last medgp_create error of the calling thread

struct EvPair { int kid; hipEvent_t a, b; };

// ---- the plan of a call (round 5) --------------------------------------------------------------------------------------------
// A call's entries are ordered by size internally and cut into SIZE CLASSES (64-block count in (2^(j-1), 2^j]): every class is a view
// of the batch buffers with its own leading dimension (the class's largest n rounded up to 64), its own launch geometry and its own
// factorisation route, and the classes of one call run beside each other on separate streams.  Why: the reference gives patients
// resources by size (ref: scripts/slurm_della.json:6-62, medgpc/util/run_exp_generator.py:213-260); rounds 1-4 chose ONE route per
// call from the call's largest patient, so one N ~ 6000 patient in a batch of 300 ran on one workgroup and set the time of the call.
enum Route { ROUTE_WG44 = 0, ROUTE_WG84 = 1, ROUTE_LA = 2 };
struct SizeClass {
    int b0 = 0, count = 0;     // internal entries [b0, b0 + count)
    int nbmax = 1;             // 64-blocks of the class's largest entry
    int ld = 64;               // leading dimension of the class view
    int wave = 0;              // memory wave of the call the class runs in (round 6; BatchPlan::nwaves)
    size_t off_mat = 0, off_vec = 0, off_tab = 0, off_slab = 0;   // offsets (doubles) of the class inside Kmat/Linv, z/alpha/wdiag, cs/sn, slab (relative to its wave: waves reuse the arenas)
    long long tsum = 0;        // sum of the cost model over its entries (route rule)
    int route = ROUTE_WG84;    // last route taken (diagnostics: medgp_last_plan)
};
struct BatchPlan {
    bool identity = true;      // internal order == caller order
    std::vector<int> order;    // internal index -> caller index
    std::vector<int> inv;      // caller index -> internal index
    std::vector<int> en;       // n of every entry, internal order
    std::vector<SizeClass> cls;
    size_t need_mat = 0, need_vec = 0, need_tab = 0, need_slab = 0;   // doubles the call needs of each arena (largest wave)
    // Memory waves (round 6): a call whose per-entry matrices exceed the context's budget (512 resident patients of N ~ 6000 would
    // need 296 GB) is run as consecutive WAVES of whole size classes that each fit it; the waves reuse the arenas in stream order.
    int nwaves = 1;
    bool with_u = true;        // laid out for Kmat AND Linv (false: an nlml-only plan, 8 ld^2 bytes per entry instead of 16)
    // Lane base (medgp_screen's two lanes, round 6): the plan's entries use rows [row0, row0 + n) of the batch-indexed buffers and the
    // arenas from these offsets on (doubles), so that two plans can be in flight on two streams at once.  0 for every other call.
    int row0 = 0;
    size_t mat0 = 0, vec0 = 0, tab0 = 0, la_part0 = 0, la_small0 = 0;
};
constexpr int kAuxStreams = 4;

// ---- device memory arenas (round 6) -----------------------------------------------------------------------------------------
// The per-entry buffers of a context live in ARENAS: one hipMalloc block each, sized ONCE where the caller can say what it will need
// (medgp_reserve for capacities below 8 GB of matrices -- every BASELINE configuration; medgp_reserve_plan from the patients' sizes)
// and otherwise replaced by a larger block when a call outgrows it.  What round 6 measured about memory on this platform
// (scratch/alloc_cost.hip, alloc_dirty.hip, vmm_repro.hip, vmm_stress.hip -> profiles/r06_memory_findings.txt; MI355X, ROCm 7.2):
//   * hipMalloc / hipFree cost microseconds on a fresh device (48 GB: 0.3 ms), but memory that has been USED and freed -- by this or
//     an earlier process -- is wiped by the driver, and an allocation that is handed such memory waits for the wipe: 24 GB after
//     another process had freed 250 GB took 6.5 s, a 100 GB block 2-3 s on its second use inside one process.  Rounds 1-5 grew the
//     arenas by hipDeviceSynchronize + hipFree + hipMalloc(1.5 x) and sized screening chunks at 26 + 9 GB: BENCH_r05's 9.5 s of
//     screening on the driver's box against 5.1 s on the builder's was this.  The cure is to obtain LITTLE memory, ONCE: nlml-only
//     plans have no Linv (half the bytes), screening chunks hold 2 GB instead of 48, the trainer announces its sizes before its loop.
//   * Virtual-memory management (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess: grow in place, never free, never
//     move) was built first and REMOVED: on this runtime hipMemSetAccess on the second chunk of a range fails with "invalid argument"
//     for some size / address patterns (24 MB then 2 MB; 2 MB then 24 MB with alignment 0), a reservation is booked against the
//     device's free-memory counter as if it were memory (one 256 GB reservation: "Free memory set to zero", every later call fails),
//     and re-mapping chunks into a larger range left stale translations behind -- a later range read another arena's data
//     (vmm_stress: mismatches in every arena after the first).  Plain blocks it is.
//   * A growth waits for NOTHING: the outgrown block is retired (queued kernels may still read it) and freed at the context's next idle
//     point (round 5: hipDeviceSynchronize + hipFree first -- every other context sharing the GPU stalled); it asks for 1.5 x and
//     retries with exactly what is needed if that fails.
struct Arena {
    char *base = nullptr;
    size_t bytes = 0;    // size of the block
};
constexpr size_t kArenaEager = (size_t)8 << 30;
enum ArenaId { AR_K = 0, AR_U, AR_Z, AR_ALPHA, AR_WDIAG, AR_CS, AR_SN, AR_SLAB, AR_LA_PART, AR_LA_SMALL, AR_COUNT };

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
    int *d_proff = nullptr, *d_pcoff = nullptr, *d_jit = nullptr, *d_bn = nullptr;
    int *d_pn = nullptr, *d_pmeta = nullptr, *d_pseg = nullptr, *d_bslot = nullptr, *d_status = nullptr;
    double *d_pt = nullptr, *d_py = nullptr;
    MedgpPrior *d_prior = nullptr;
    uint8_t *d_prior_on = nullptr;
    double *d_theta = nullptr, *d_nlml = nullptr, *d_grad = nullptr;   // staging for the host-pointer API (= lane 0)
    int *d_status_out = nullptr;
    // second lane of the asynchronous host-pointer API (medgp_nlml_grad_async): lane 0 uses the buffers above
    double *d_theta1 = nullptr, *d_nlml1 = nullptr, *d_grad1 = nullptr;
    int *d_status1 = nullptr;
    hipEvent_t ev_lane[2] = {nullptr, nullptr};
    // copy streams of the asynchronous lanes: the theta upload of one lane and the result download of the other run beside the
    // kernels on c->stream (on the compute stream they were 9 MB of PCIe traffic per 512-patient step, serialised with the kernels)
    hipStream_t s_up = nullptr, s_down = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr};
    bool lane_pending[2] = {false, false};
    // pinned staging ring for small host-to-device uploads (slot tables, prior descriptors): the source of an asynchronous
    // copy must stay untouched until the copy has run, and nothing here waits for the device on the normal path
    char *pin_buf[2] = {nullptr, nullptr};
    size_t pin_cap = 0, pin_off = 0;
    int pin_cur = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_pending[2] = {false, false};
    char *h_small = nullptr;               // pinned landing area of the host-pointer operator's results (small calls)
    char *d_small = nullptr;               // ... and their device image [nlml | grad | status]: ONE copy brings a small call's results back
    MedgpPrior *d_prior_stage = nullptr;   // device staging of medgp_set_prior[s] rows
    int *d_prior_slots = nullptr;
    size_t prior_stage_rows = 0;
    // host mirrors
    std::vector<int> h_n;
    std::vector<std::vector<int>> h_perm;   // internal index -> caller index
    std::vector<uint8_t> h_perm_identity;
    std::vector<int> h_bslot;       // effective slots of the last call, CALLER order
    int last_nbatch = 0;
    BatchPlan plan;                 // of the last call
    // arenas of the per-entry buffers (Kmat, Linv, z, alpha, wdiag, cs, sn, slab, look-ahead scratch): what medgp_reserve's capacities
    // would need at most (full_*).  Up to kArenaEager bytes they are allocated by medgp_reserve; beyond
    // that (a ragged cohort whose largest patient is far above the median: max_batch x max_n^2 would not fit 288 GB) they grow with the
    // calls or are sized once by medgp_reserve_plan (struct Arena).
    size_t full_mat = 0, full_vec = 0, full_tab = 0, full_slab = 0;
    Arena arena[AR_COUNT];
    size_t mem_budget = (size_t)64 << 30;   // bytes of per-entry matrices one wave of a call may use (MEDGP_MEM_BUDGET_GB)
    size_t screen_budget = (size_t)2 << 30; // the same for one chunk of medgp_screen (MEDGP_SCREEN_BUDGET_GB)
    long long screen_work = 32768;  // block pairs (sum nb^2) at which a medgp_screen chunk of look-ahead entries is closed (MEDGP_SCREEN_WORK)
    // blocks that outgrown arenas left behind: kernels queued before the growth may still read them, so they are NOT freed there (no
    // wait at the growth) but at the next point where the context's streams are known to be idle (free_retired)
    std::vector<void *> retired;
    size_t retired_bytes = 0;
    double alloc_s = 0.0;           // wall seconds inside device / pinned memory management calls (medgp_alloc_stats)
    long long alloc_calls = 0;
    int *d_bpos = nullptr;
    int *d_tpos = nullptr;          // theta rows of the entries (medgp_screen), internal order
    bool tpos_on = false;
    double *d_screen_theta = nullptr;   // the block of hyper vectors of medgp_screen
    size_t screen_theta_cap = 0;
    bool last_has_inverse = false;   // the last pipeline run formed alpha and U = L^-T (medgp_get_factor is valid)
    // upload staging (one pinned host buffer + one device buffer, reused; guarded by ev_stage)
    char *h_stage = nullptr, *d_stage = nullptr;
    size_t stage_cap = 0;
    hipEvent_t ev_stage = nullptr;
    bool stage_pending = false;
    // scratch of the look-ahead multi-CU factorisation (kernels_cholinv_la.h), grown on demand
    // (arena[AR_LA_PART], arena[AR_LA_SMALL])
    char *h_bounce = nullptr;        // pinned bounce buffer for the large device-to-host exports (factor matrices)
    size_t bounce_cap = 0;
    int *d_one_slot = nullptr;       // single-entry slot table for the caller-order re-factorisation of medgp_get_factor
    // predict scratch
    double *d_t2 = nullptr, *d_ks = nullptr;
    int *d_meta2 = nullptr;
    float *d_mean = nullptr, *d_var = nullptr;
    int pred_cap = 0;
    // profiling
    bool profiling = false;
    int profile_only = -1;    // >= 0: only launches of this kernel id are bracketed (medgp_profile_enable(ctx, 2 + id))
    int wgrad_deep = -1;      // MEDGP_WGRAD_DEEP=1/2: force the prefetch depth of k_wgrad's operand stream (A-B; same bits); -1: by launch size
    bool use_v0 = false;      // MEDGP_V0=1: the generic (non-templated) pair kernels of the Q > 8 route for any Q (debug / A-B parity)
    int cholinv_nw = 0;       // MEDGP_CHOLINV_NW=44|84 forces the workgroup shape <waves, 16-row units per wave> (0 = auto)
    int la_park = 256;        // MEDGP_LA_PARK=<workgroup id>|0: where the look-ahead schedule parks its sleeping workgroup (0 = off)
    int la_park_maxbatch = 8; // MEDGP_LA_PARK_MAXBATCH: largest batch the parking is used for (measured: 4 x N=2048 -5 %, 16 x N=2048 +2 %)
    int force_mc = 0;         // MEDGP_MULTI_CU=1 forces / -1 forbids the multi-CU factorisation (0 = auto)
    int pin_route = 0;        // medgp_pin_route: every entry is factored by k_cholinv<8,4> whatever the batch (reproducible bits)
    int num_cu = 256;
    int dbg_fail = 0;         // MEDGP_DEBUG_FAIL_ATTEMPTS=k: test hook, see MedgpDev::dbg_fail
    int class_streams = kAuxStreams;   // MEDGP_CLASS_STREAMS=0: the size classes of a call run back to back on the call's stream (A-B)
    int no_classes = 0;       // MEDGP_NO_CLASSES=1: rounds 1-4 behaviour -- one class per call, one route from its largest entry (A-B)
    hipStream_t aux[kAuxStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kAuxStreams] = {nullptr, nullptr, nullptr, nullptr};
    // second lane of medgp_screen (round 6): chunks alternate between c->stream and s_screen1, each with its own rows of the batch
    // buffers and its own half of the arenas -- the assembly of one chunk runs beside the factorisation of the other
    hipStream_t s_screen1 = nullptr;
    hipEvent_t ev_fork1 = nullptr, ev_screen0 = nullptr;
    int screen_lanes = 2;           // MEDGP_SCREEN_LANES=1: one lane (rounds 1-5)
    char *h_screen_tab = nullptr;   // pinned: the slot / position / theta-row tables of ALL chunks of one medgp_screen call
    size_t screen_tab_cap = 0;
    std::vector<EvPair> events;
    std::vector<hipEvent_t> ev_pool;   // recycled timing events: a profiled launch creates none once the pool is warm
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
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    c->alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->alloc_calls++;
    if (e != hipSuccess) return fail(c, MEDGP_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = (T *)q;
    return MEDGP_OK;
}

void arena_release(medgp_ctx *c, Arena &A);
void free_all(medgp_ctx *c) {
    for (void *p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    for (int i = 0; i < AR_COUNT; i++) arena_release(c, c->arena[i]);
    for (void *p : c->retired) (void)hipFree(p);
    c->retired.clear();
    c->retired_bytes = 0;
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
    bool on;
    Launcher(medgp_ctx *c_, int kid_, hipStream_t st_ = nullptr) : c(c_), kid(kid_), st(st_ ? st_ : c_->stream) {
        on = c->profiling && (c->profile_only < 0 || c->profile_only == kid);
        if (on) {
            a = take(); b = take();
            (void)hipEventRecord(a, st);
        }
    }
    hipEvent_t take() {
        hipEvent_t e = nullptr;
        if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    }
    void finish() {   // closes the bracket now (the destructor then does nothing)
        if (on) {
            (void)hipEventRecord(b, st);
            if (kid >= 0) c->events.push_back({kid, a, b});
            else { c->ev_pool.push_back(a); c->ev_pool.push_back(b); }   // recorded but never read: safe to re-record later
            on = false;
        }
    }
    ~Launcher() { finish(); }
};

int drain_events(medgp_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (auto &e : c->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.kid] += ms;
            c->prof_n[e.kid] += 1;
        }
        c->ev_pool.push_back(e.a);
        c->ev_pool.push_back(e.b);
    }
    c->events.clear();
    return MEDGP_OK;
}

// A pinned region of `bytes` that stays untouched until every copy queued from it on c->stream so far has run.  Two buffers:
// when one is full its event is recorded and the other one is taken (waiting only if THAT buffer's copies from a whole
// ring revolution ago are still in flight, which in practice never happens).
int pin_stage(medgp_ctx *c, size_t bytes, void **out) {
    bytes = (bytes + 63) & ~(size_t)63;
    if (bytes > c->pin_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < 2; i++) { if (c->pin_buf[i]) (void)hipHostFree(c->pin_buf[i]); c->pin_buf[i] = nullptr; c->pin_pending[i] = false; }
        const size_t cap = std::max<size_t>(2 * bytes, (size_t)1 << 20);
        for (int i = 0; i < 2; i++) HIPCHK(c, hipHostMalloc((void **)&c->pin_buf[i], cap, hipHostMallocDefault));
        c->pin_cap = cap; c->pin_off = 0; c->pin_cur = 0;
    }
    for (int i = 0; i < 2; i++) if (!c->pin_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
    if (c->pin_off + bytes > c->pin_cap) {
        HIPCHK(c, hipEventRecord(c->pin_ev[c->pin_cur], c->stream));
        c->pin_pending[c->pin_cur] = true;
        c->pin_cur ^= 1;
        if (c->pin_pending[c->pin_cur]) { HIPCHK(c, hipEventSynchronize(c->pin_ev[c->pin_cur])); c->pin_pending[c->pin_cur] = false; }
        c->pin_off = 0;
    }
    *out = c->pin_buf[c->pin_cur] + c->pin_off;
    c->pin_off += bytes;
    return MEDGP_OK;
}

inline int tri(int n) { return n * (n + 1) / 2; }
inline int blocks64(int n) { return (std::max(n, 1) + 63) / 64; }
// size class of an entry of nb 64-blocks: 0 -> {1}, 1 -> {2}, 2 -> {3, 4}, 3 -> {5 .. 8}, ...
inline int size_bucket(int nb) { int j = 0; while ((1 << j) < nb) j++; return j; }
// Cost model of one entry on ONE workgroup (k_cholinv), fitted to profiles/r04_route_table.txt (ms = 4.4e-4 nb^2 (nb + 17):
// N = 256 0.15, 512 0.70, 768 1.83, 1024 3.7; N = 8192: 1.05 s against 1.33 s measured).  Integer, so the route rule is exact.
inline long long wg_cost(int nb) { return (long long)nb * nb * (nb + 17); }

// Lay out the plan of a call from the sizes of its entries alone (en[b] = n of caller entry b): internal order, size classes, memory
// waves, offsets, needs.  with_u: the call forms U = L^-T (gradient / factor outputs / predict); an nlml-only call touches neither
// Linv nor the gradient slab.  Pure host arithmetic: medgp_reserve_plan runs it on announced sizes to find the high-water marks.
void layout_plan(const medgp_ctx *c, const int *en, int nbatch, bool with_u, BatchPlan &P) {
    P.order.resize(nbatch); P.inv.resize(nbatch); P.en.resize(nbatch);
    P.cls.clear();
    P.with_u = with_u;
    int mx = 0;
    for (int b = 0; b < nbatch; b++) { P.order[b] = b; mx = std::max(mx, en[b]); }
    const bool classes = !c->no_classes;
    // by 64-block count, largest first (what the hardware dispatches first runs longest: LPT inside every launch); ties keep the caller's order
    if (classes) std::stable_sort(P.order.begin(), P.order.end(), [&](int a, int b) { return blocks64(en[a]) > blocks64(en[b]); });
    P.identity = true;
    for (int i = 0; i < nbatch; i++) { P.inv[P.order[i]] = i; P.en[i] = en[P.order[i]]; P.identity = P.identity && P.order[i] == i; }
    const size_t Q = c->Q, D = c->D, bpe = with_u ? 16 : 8;   // bytes of per-entry matrices per ld^2
    size_t om = 0, ov = 0, ot = 0, os = 0, wave_bytes = 0;
    int wave = 0;
    P.need_mat = P.need_vec = P.need_tab = P.need_slab = 0;
    for (int i = 0; i < nbatch;) {
        SizeClass k;
        k.b0 = i;
        const int bk = size_bucket(blocks64(P.en[i]));
        k.nbmax = classes ? blocks64(P.en[i]) : blocks64(mx);   // (sorted: the first entry of a class is its largest)
        k.ld = 64 * k.nbmax;
        // a class is cut where its matrices would exceed the budget of one wave (512 entries of N ~ 6000: 296 GB)
        const size_t per = bpe * (size_t)k.ld * k.ld;
        const int cmax = classes ? (int)std::max<size_t>(1, c->mem_budget / per) : nbatch;
        int j = i;
        while (j < nbatch && j - i < cmax && (!classes || size_bucket(blocks64(P.en[j])) == bk)) { k.tsum += wg_cost(blocks64(P.en[j])); j++; }
        k.count = j - i;
        const size_t kbytes = per * k.count;
        if (classes && wave_bytes > 0 && wave_bytes + kbytes > c->mem_budget) { wave++; om = ov = ot = os = 0; wave_bytes = 0; }
        k.wave = wave;
        wave_bytes += kbytes;
        k.off_mat = om; k.off_vec = ov; k.off_tab = ot; k.off_slab = os;
        om += (size_t)k.count * k.ld * k.ld; ov += (size_t)k.count * k.ld; ot += (size_t)k.count * Q * k.ld;
        if (with_u) os += (size_t)k.count * 3 * Q * (k.ld / 16 + D) * (k.ld / 64 + D);
        P.need_mat = std::max(P.need_mat, om); P.need_vec = std::max(P.need_vec, ov); P.need_tab = std::max(P.need_tab, ot); P.need_slab = std::max(P.need_slab, os);
        P.cls.push_back(k);
        i = j;
    }
    P.nwaves = wave + 1;
}

// Select the batch and lay out its plan.  caller_order: entries whose patient was not uploaded grouped by output use the caller-order
// copy of the patient (slot + max_slots), so that the factor is the one the caller's order defines (no gradient on that copy).
int set_batch(medgp_ctx *c, int nbatch, const int32_t *slots, int *max_n_out, bool caller_order, bool with_u) {
    if (nbatch < 1 || nbatch > c->max_batch) return fail(c, MEDGP_ERR_CAPACITY, "nbatch %d outside [1, %d]", nbatch, c->max_batch);
    int mx = 0;
    std::vector<int> eff(nbatch), en(nbatch);
    for (int b = 0; b < nbatch; b++) {
        int s = slots[b];
        if (s < 0 || s >= c->max_slots || c->h_n[s] < 0) return fail(c, MEDGP_ERR_ARG, "slots[%d] = %d is not a resident patient", b, s);
        mx = std::max(mx, c->h_n[s]);
        en[b] = c->h_n[s];
        eff[b] = (caller_order && !c->h_perm_identity[s]) ? s + c->max_slots : s;
    }
    *max_n_out = mx;
    bool same = (nbatch == c->last_nbatch) && c->plan.with_u == with_u && std::memcmp(c->h_bslot.data(), eff.data(), sizeof(int) * nbatch) == 0;
    if (same) return MEDGP_OK;
    std::memcpy(c->h_bslot.data(), eff.data(), sizeof(int) * nbatch);
    BatchPlan &P = c->plan;
    layout_plan(c, en.data(), nbatch, with_u, P);
    // the tables travel through the pinned ring: no wait for the device (the lock-step optimiser changes the active set on
    // most steps; stream order puts the copy behind the kernels of the previous call that still read the old tables)
    const bool need_pos = !(P.identity && P.cls.size() == 1);
    void *pin = nullptr;
    int rcp = pin_stage(c, sizeof(int) * nbatch * (need_pos ? 2 : 1), &pin);
    if (rcp) return rcp;
    int *hp = (int *)pin;
    for (int i = 0; i < nbatch; i++) hp[i] = eff[P.order[i]];
    HIPCHK(c, hipMemcpyAsync(c->d_bslot, hp, sizeof(int) * nbatch, hipMemcpyHostToDevice, c->stream));
    if (need_pos) {
        for (int i = 0; i < nbatch; i++) hp[nbatch + i] = P.order[i];
        HIPCHK(c, hipMemcpyAsync(c->d_bpos, hp + nbatch, sizeof(int) * nbatch, hipMemcpyHostToDevice, c->stream));
    }
    c->last_nbatch = nbatch;
    return MEDGP_OK;
}

// the view of the batch buffers one size class works in (entry 0 of the view = internal entry k.b0)
MedgpDev class_view(const medgp_ctx *c, const BatchPlan &P, const SizeClass &k) {
    const MedgpDev &L = c->dev;
    MedgpDev V = L;
    const size_t Q = L.Q, D = L.D, b0 = (size_t)k.b0 + (size_t)P.row0;
    V.ldn = k.ld;
    V.slab_R = k.ld / 16 + (int)D; V.slab_C = k.ld / 64 + (int)D;
    V.slab_stride = (size_t)3 * Q * V.slab_R * V.slab_C;
    V.bslot = L.bslot + b0;
    V.bpos = (P.identity && P.cls.size() == 1) ? nullptr : c->d_bpos + b0;
    V.tpos = c->tpos_on ? c->d_tpos + b0 : nullptr;
    V.hyp = L.hyp + b0 * L.hyp_stride;
    V.cs = L.cs + P.tab0 + k.off_tab; V.sn = L.sn + P.tab0 + k.off_tab;
    V.Kmat = L.Kmat + P.mat0 + k.off_mat; V.Linv = L.Linv + (P.with_u ? P.mat0 + k.off_mat : 0);
    V.z = L.z + P.vec0 + k.off_vec; V.alpha = L.alpha + P.vec0 + k.off_vec; V.wdiag = L.wdiag + P.vec0 + k.off_vec;
    V.epi_lp = L.epi_lp + b0 * MEDGP_EPI_PARTS; V.epi_ticket = L.epi_ticket + b0;
    V.scal = L.scal + b0 * 4; V.status = L.status + b0; V.jit = L.jit + b0; V.bn = L.bn + b0; V.xk = L.xk + b0 * 64 * 64;
    V.S = L.S + b0 * Q * D * D; V.SM = L.SM + b0 * Q * D * D; V.SV = L.SV + b0 * Q * D * D;
    V.slab = L.slab + k.off_slab;
    return V;
}

// view of the batch-indexed buffers starting at entry b0 of the view L
MedgpDev shifted_view(const MedgpDev &L, int b0) {
    MedgpDev V = L;
    const size_t ld = L.ldn, Q = L.Q, D = L.D;
    V.bslot = L.bslot + b0;
    if (L.bpos) V.bpos = L.bpos + b0;
    if (L.tpos) V.tpos = L.tpos + b0;
    V.hyp = L.hyp + (size_t)b0 * L.hyp_stride;
    V.cs = L.cs + (size_t)b0 * Q * ld; V.sn = L.sn + (size_t)b0 * Q * ld;
    V.Kmat = L.Kmat + (size_t)b0 * ld * ld; V.Linv = L.Linv + (size_t)b0 * ld * ld;
    V.z = L.z + (size_t)b0 * ld; V.alpha = L.alpha + (size_t)b0 * ld; V.wdiag = L.wdiag + (size_t)b0 * ld;
    V.epi_lp = L.epi_lp + (size_t)b0 * MEDGP_EPI_PARTS; V.epi_ticket = L.epi_ticket + b0;
    V.scal = L.scal + (size_t)b0 * 4; V.status = L.status + b0; V.jit = L.jit + b0; V.bn = L.bn + b0; V.xk = L.xk + (size_t)b0 * 64 * 64;
    V.S = L.S + (size_t)b0 * Q * D * D; V.SM = L.SM + (size_t)b0 * Q * D * D; V.SV = L.SV + (size_t)b0 * Q * D * D;
    V.slab = L.slab + (size_t)b0 * L.slab_stride;
    return V;
}

// the one-entry view of CALLER entry b of the last call
MedgpDev entry_view(const medgp_ctx *c, int b) {
    const int i = c->plan.inv[b];
    for (const SizeClass &k : c->plan.cls)
        if (i >= k.b0 && i < k.b0 + k.count) return shifted_view(class_view(c, c->plan, k), i - k.b0);
    return c->dev;   // (not reached: every entry belongs to a class)
}

// device -> host copy of `bytes` through a pinned bounce buffer (a pageable hipMemcpy of this size pays ~10 ms of one-time
// runtime staging set-up on its first use); returns the pinned pointer, valid until the next call
int d2h_pinned(medgp_ctx *c, const void *dev, size_t bytes, const char **out) {
    if (bytes > c->bounce_cap) {
        if (c->h_bounce) (void)hipHostFree(c->h_bounce);
        c->h_bounce = nullptr; c->bounce_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_bounce, bytes + bytes / 4, hipHostMallocDefault));
        c->bounce_cap = bytes + bytes / 4;
    }
    HIPCHK(c, hipMemcpyAsync(c->h_bounce, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = c->h_bounce;
    return MEDGP_OK;
}

// ---- arenas: one block each, sized once where possible, replaced (never waited for) where not (see struct Arena) ----------------
struct AllocTimer {   // wall time of the memory-management calls, for medgp_alloc_stats
    medgp_ctx *c;
    std::chrono::steady_clock::time_point t0;
    explicit AllocTimer(medgp_ctx *c_) : c(c_), t0(std::chrono::steady_clock::now()) {}
    ~AllocTimer() { c->alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); c->alloc_calls++; }
};

void arena_release(medgp_ctx *c, Arena &A) {
    AllocTimer tm(c);
    if (A.base) (void)hipFree(A.base);
    A = Arena{};
}

// every stream of the context is idle afterwards (NOT the device: other contexts / ranks that share it keep running)
int sync_ctx_streams(medgp_ctx *c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < kAuxStreams; i++) if (c->aux[i]) HIPCHK(c, hipStreamSynchronize(c->aux[i]));
    if (c->s_up) HIPCHK(c, hipStreamSynchronize(c->s_up));
    if (c->s_down) HIPCHK(c, hipStreamSynchronize(c->s_down));
    if (c->s_screen1) HIPCHK(c, hipStreamSynchronize(c->s_screen1));
    return MEDGP_OK;
}

// the blocks outgrown arenas left behind; the caller guarantees that every stream of the context is idle
void free_retired(medgp_ctx *c, bool timed = true) {
    if (c->retired.empty()) return;
    const auto t0 = std::chrono::steady_clock::now();
    for (void *p : c->retired) (void)hipFree(p);
    if (timed) { c->alloc_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); c->alloc_calls++; }
    c->retired.clear();
    c->retired_bytes = 0;
}

// make at least `bytes` of arena `id` usable.  exact: the caller knows this is the high-water mark (medgp_reserve, medgp_reserve_plan):
// allocate just that; otherwise 1.5 x what was there (at most `limit` bytes, the most the capacities of medgp_reserve allow), with a
// retry at exactly `bytes`.  *moved is set when the block was replaced: its contents are gone.
// The old block is RETIRED, not freed: kernels queued earlier may still read it, and neither they nor anything else is waited for here
// (round 5 synchronised the whole device and freed first).  Retired blocks are freed at the next idle point of the context (the end of a
// blocking call, medgp_synchronize), at once when they add up to more than half the memory budget, or when the device is out of memory.
int arena_ensure(medgp_ctx *c, int id, size_t bytes, size_t limit, bool exact, bool *moved) {
    Arena &A = c->arena[id];
    if (bytes <= A.bytes) return MEDGP_OK;
    AllocTimer tm(c);
    const size_t old = A.bytes;
    if (A.base) { c->retired.push_back(A.base); c->retired_bytes += A.bytes; }
    A = Arena{};
    if (c->retired_bytes > c->mem_budget / 2) { int rc = sync_ctx_streams(c); if (rc) return rc; free_retired(c, false); }
    size_t want = exact ? bytes : std::max(bytes, old + old / 2);
    if (limit >= bytes) want = std::min(want, limit);
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess && !c->retired.empty()) {   // out of memory with retired blocks around: give them back first
        (void)hipGetLastError();
        int rc = sync_ctx_streams(c); if (rc) return rc;
        free_retired(c, false);
        e = hipMalloc(&q, want);
    }
    if (e != hipSuccess && want > bytes) { (void)hipGetLastError(); want = bytes; e = hipMalloc(&q, want); }   // near capacity the 1.5 x request can fail where `bytes` fits
    if (e != hipSuccess) return fail(c, MEDGP_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    A.base = (char *)q; A.bytes = want;
    if (moved) *moved = true;
    return MEDGP_OK;
}

// the arenas of the per-entry buffers of the batch views: doubles needed of Kmat, Linv (0: an nlml-only plan never touches it),
// z / alpha / wdiag, cs / sn, slab.  exact: see arena_ensure.
int ensure_arena(medgp_ctx *c, size_t need_k, size_t need_u, size_t need_vec, size_t need_tab, size_t need_slab, bool exact = false) {
    MedgpDev &L = c->dev;
    bool moved = false, mv;
    int rc;
    auto one = [&](int id, size_t need, size_t full, double **p) -> int {
        mv = false;
        if ((rc = arena_ensure(c, id, std::max<size_t>(need, 8) * sizeof(double), full * sizeof(double), exact, &mv))) return rc;
        *p = (double *)c->arena[id].base;
        moved = moved || mv;
        return MEDGP_OK;
    };
    if ((rc = one(AR_K, need_k, c->full_mat, &L.Kmat))) return rc;
    if ((rc = one(AR_U, need_u, c->full_mat, &L.Linv))) return rc;
    if ((rc = one(AR_Z, need_vec, c->full_vec, &L.z))) return rc;
    if ((rc = one(AR_ALPHA, need_vec, c->full_vec, &L.alpha))) return rc;
    if ((rc = one(AR_WDIAG, need_vec, c->full_vec, &L.wdiag))) return rc;
    if ((rc = one(AR_CS, need_tab, c->full_tab, &L.cs))) return rc;
    if ((rc = one(AR_SN, need_tab, c->full_tab, &L.sn))) return rc;
    if ((rc = one(AR_SLAB, need_slab, c->full_slab, &L.slab))) return rc;
    if (moved) c->last_has_inverse = false;   // (the factors of earlier calls are gone)
    return MEDGP_OK;
}

// scratch of the look-ahead factorisation for the classes of a wave that take it: two arenas, carved up per class -- the classes run
// on different streams.  with_u = false (nlml only): no U row blocks, so the partial-sum slab holds nbmax + 1 row blocks, not 2 nbmax + 1.
struct LaNeed { int count, nbmax, ld; LaArgs A; };
inline size_t la_part_doubles(const LaNeed &e, bool with_u) { return (size_t)e.count * 2 * ((with_u ? 2 : 1) * e.nbmax + 1) * ((e.nbmax + LA_SLICE - 1) / LA_SLICE) * 4096; }
inline size_t la_small_doubles(const LaNeed &e) { return (size_t)e.count * (64 * (size_t)e.ld + 5 * 2 * 4096 + 2 * (size_t)((e.nbmax + LA_SLICE - 1) / LA_SLICE) * 4096 + 1); }
int ensure_la(medgp_ctx *c, std::vector<LaNeed> &v, bool with_u, size_t part0 = 0, size_t small0 = 0) {
    const size_t nring = 2;
    size_t need_part = part0, need_small = small0;   // (part0 / small0: the lane base of the plan, doubles)
    for (const LaNeed &e : v) { need_part += la_part_doubles(e, with_u); need_small += la_small_doubles(e); }
    int rc;
    // (a block that must grow is replaced; the old one is retired, not freed: kernels queued on any of the context's streams may still read it)
    if ((rc = arena_ensure(c, AR_LA_PART, need_part * sizeof(double), 0, false, nullptr))) return rc;
    if ((rc = arena_ensure(c, AR_LA_SMALL, need_small * sizeof(double), 0, false, nullptr))) return rc;
    double *pp = (double *)c->arena[AR_LA_PART].base + part0, *ps = (double *)c->arena[AR_LA_SMALL].base + small0;
    for (LaNeed &e : v) {
        LaArgs A{};
        const size_t nb = e.count;
        A.nbmax = e.nbmax;
        A.maxslice = (e.nbmax + LA_SLICE - 1) / LA_SLICE;
        A.rows = (with_u ? 2 : 1) * e.nbmax + 1;
        A.ring = 1;
        A.nbatch = e.count;
        A.part = pp;
        A.ybuf = ps;
        A.xk2 = ps + nb * 64 * e.ld;
        A.pnx = A.xk2 + nb * nring * 4096;
        A.pnx2 = A.pnx + nb * nring * 4096;
        A.dterm = A.pnx2 + nb * nring * 4096;
        A.dsum = A.dterm + nb * nring * 4096;
        A.dpart = A.dsum + nb * nring * 4096;
        A.flag = (int *)(A.dpart + nb * 2 * A.maxslice * 4096);
        e.A = A;
        pp += la_part_doubles(e, with_u); ps += la_small_doubles(e);
    }
    return MEDGP_OK;
}

// one kernel chain for the `nbatch` entries of the view L (a size class of the call) on `stream`; nt64 = 64-blocks of its largest entry,
// entry_n = their sizes (host mirror), route = how they are factored, la = the look-ahead scratch (ROUTE_LA only)
int run_pipeline_one(medgp_ctx *c, hipStream_t stream, const MedgpDev &L, int nbatch, int nt64, const double *theta_dev,
                     int flag_grad, bool need_inverse, int min_n, double *nlml_dev, double *grad_dev, int32_t *status_dev,
                     bool store_ukk, const int *entry_n, int route, const LaArgs *la_in) {
    bool ragged = false;   // entries of different 64-block counts in this class
    for (int bb = 1; bb < nbatch; bb++) ragged = ragged || blocks64(entry_n[bb]) != blocks64(entry_n[0]);
    { Launcher l(c, KID_PREP, stream); hipLaunchKernelGGL(k_prep, dim3(nbatch, 1 + (L.Q * L.ldn + PREP_CHUNK - 1) / PREP_CHUNK + ((theta_dev && L.kidx == 7) ? (L.Q * L.D * L.D + PREP_BCHUNK - 1) / PREP_BCHUNK : 0)), dim3(256), 0, stream, L, theta_dev, min_n); }
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
        // 9 .. 16 components: the first eight, then the rest added to the same tiles (kernels_assemble.h)
#define MEDGP_ASM2(QR) case 8 + QR: hipLaunchKernelGGL(k_assemble_t<8>, tg, tb, 0, stream, L); hipLaunchKernelGGL((k_assemble_t<QR, 8>), tg, tb, 0, stream, L); break;
        MEDGP_ASM2(1) MEDGP_ASM2(2) MEDGP_ASM2(3) MEDGP_ASM2(4) MEDGP_ASM2(5) MEDGP_ASM2(6) MEDGP_ASM2(7) MEDGP_ASM2(8)
#undef MEDGP_ASM2
        default: hipLaunchKernelGGL(k_assemble_v0, tg, tb, 0, stream, L); break;   // Q > 16 (or MEDGP_V0): generic kernel
        }
    };
    const bool inv = flag_grad || need_inverse;
    const int want_mode = inv ? 1 : (store_ukk ? 2 : 0);   // bit 0: U rows + alpha; 2: diagonal blocks U_kk only (k_predict)
    // Few large patients: the multi-CU look-ahead schedule (kernels_cholinv_la.h) instead of one workgroup per patient; which classes
    // of a call take it is decided in run_pipeline (route rule).
    const bool multi_cu = route == ROUTE_LA;
    if (multi_cu) {
        // which entries the multi-CU schedule factors: more than one 64-block (host mirror of the patient sizes)
        bool any_small = false;
        for (int bb = 0; bb < nbatch; bb++) any_small = any_small || entry_n[bb] <= 64;
        {
            const LaArgs la = *la_in;
            launch_assemble();
            // entries of a single 64-block: one workgroup each (the same kernel, hence the same bits, as in any other call)
            if (any_small) { Launcher l(c, KID_CHOLINV, stream); hipLaunchKernelGGL((k_cholinv<8, 4, 1>), dim3(nbatch), dim3(512), 0, stream, L, want_mode); }
            const int nbx = ragged ? (nbatch | 1) : nbatch;   // x extent of the look-ahead grids: odd for a ragged class (LaArgs::nbatch)
            { Launcher l(c, KID_LA_AUX, stream); hipLaunchKernelGGL(k_la_prologue, dim3(nbx, 1 + nt64), dim3(LA_THREADS), 0, stream, L, la, want_mode); }
            auto step_counts = [&](int k, int *nF, int *nLrows, int *nsl) {
                const int nMF = std::max(nt64 - (k + 2), 0), nUF = (want_mode & 1) ? k + 1 : 0, nUL = (want_mode & 1) ? k : 0;
                *nF = nMF + nUF + 1;
                *nLrows = (k + 2 < nt64 && k >= 1) ? nMF + nUL + 1 : 0;
                *nsl = std::min(la.maxslice, (k + la_slice_len(k) - 1) / la_slice_len(k));   // history slices that exist at step k
            };
            // look-ahead schedule: one launch per 64-wide step (kernels_cholinv_la.h)
            for (int k = 0; k < nt64; k++) {
                int nF, nLrows, nsl;
                step_counts(k, &nF, &nLrows, &nsl);
                // single entry: park a sleeping workgroup where the dispatcher would put the chain's first neighbour (kernels_cholinv_la.h)
                const int ntask = 1 + LA_NAUX(k) + nF + nLrows * nsl;   // D, the F row blocks, R and / or H, the look-ahead slices
                // (workgroup ids are y * nbatch + x: with nbatch entries the chains are ids 0 .. nbatch-1 and their first neighbours
                //  ids 256 .. 256+nbatch-1, i.e. task y = 256 / nbatch of every entry)
                const int pk = (c->la_park > 0 && nbx == nbatch && nbatch <= c->la_park_maxbatch && c->la_park % nbatch == 0) ? c->la_park / nbatch : -1;
                const int park = (pk > 0 && ntask > pk && k + 1 < nt64) ? pk : -1;
                Launcher l(c, KID_LA_STEP, stream);
                hipLaunchKernelGGL(k_la_step, dim3(nbx, ntask + (park >= 0 ? 1 : 0)), dim3(LA_THREADS), 0, stream, L, la, k, want_mode, nLrows, park);
            }
            { Launcher l(c, KID_LA_AUX, stream); hipLaunchKernelGGL(k_la_finish, dim3(nbx, 4 * nt64), dim3(256), 0, stream, L, la, want_mode); }
            // The reference's retry loop (c_inference_exact.cpp:99-111: add the noise vector again, at most 10 times), device
            // driven: entries whose one attempt above failed (status -2) are re-assembled and factored by k_cholinv's in-kernel
            // loop; for healthy entries this launch is one workgroup that reads a status word.  No host read-back, no wait.
            { Launcher l(c, KID_CHOLINV, stream); hipLaunchKernelGGL((k_cholinv<8, 4, 2>), dim3(nbatch), dim3(512), 0, stream, L, want_mode); }
        }
    } else {
        launch_assemble();
        Launcher l(c, KID_CHOLINV, stream);
        // more patients than CUs: 4-wave workgroups, two per CU (the serial diagonal phase of one overlaps the MFMA phase of the
        // other).  At most one patient per CU: 8 waves (8 block slots per pass) once a step has more than 4 row blocks, else the
        // 4-wave shape, whose 4 slots already cover every block of n <= 256 (measured, 256 patients x N=256, D=2: <4,4> 0.211 ms,
        // <8,4> 0.232 ms -- half of its 8 slots idle).
        // Only shapes with at most two waves per SIMD are instantiated: the out-of-line diagonal factor (diag_factor_wave) is
        // compiled ONCE for the tightest register budget among its callers -- with a <*,2> shape (four waves per SIMD, 128 VGPRs)
        // in the library it is held to 128 VGPRs and carries 182 scratch accesses on the serial path of EVERY shape (248 VGPRs and
        // 18 without; found when the legacy k_ci_panel caller that had masked this left the build: k_cholinv<4,4> 1.37 -> 1.48 ms).
#ifdef MEDGP_PRICE_NLML
        // pricing build (results meaningless): MEDGP_PRICE_SHAPE=43 / 42 runs the nlml-only calls in a shape with 3 / 4 workgroups per CU
        static const int price_shape = getenv("MEDGP_PRICE_SHAPE") ? atoi(getenv("MEDGP_PRICE_SHAPE")) : 0;
        if (want_mode == 0 && route == ROUTE_WG44 && price_shape == 43) hipLaunchKernelGGL((k_cholinv<4, 3, 0>), dim3(nbatch), dim3(256), 0, stream, L, want_mode);
        else if (want_mode == 0 && route == ROUTE_WG44 && price_shape == 42) hipLaunchKernelGGL((k_cholinv<4, 2, 0>), dim3(nbatch), dim3(256), 0, stream, L, want_mode);
        else
#endif
        if (route == ROUTE_WG44) hipLaunchKernelGGL((k_cholinv<4, 4, 0>), dim3(nbatch), dim3(256), 0, stream, L, want_mode);
        else hipLaunchKernelGGL((k_cholinv<8, 4, 0>), dim3(nbatch), dim3(512), 0, stream, L, want_mode);
    }
    int from_slab = 0;
#ifdef MEDGP_STAMPS
    if (getenv("MEDGP_DBG_NOWGRAD")) { HIPCHK(c, hipGetLastError()); return MEDGP_OK; }
#endif
    if (flag_grad) {
        const int wg_tiles = tri(nt64);
        // entries of different sizes in a launch of few entries: odd stride of the entry index, so that every entry's tiles go to all
        // XCDs (kernels_wgrad.h); equally large entries keep the stride nbatch (balanced as it is, and the measured form)
        const int nbp = (ragged && nbatch < 64) ? (nbatch | 1) : nbatch;
        const dim3 tg(std::max(8 * ((nbatch + 7) / 8), nbp) * wg_tiles), tb(WG_THREADS);
        from_slab = 1;
        Launcher lw(c, KID_WGRAD, stream);
        // few large patients (the launch fills the chip less than four times): operand prefetch two chunks ahead + serpentine tile order (kernels_wgrad.h)
        const int pf = c->wgrad_deep >= 0 ? c->wgrad_deep : ((long)nbatch * wg_tiles <= 16L * c->num_cu ? 2 : 1);
#define MEDGP_WGL(QQ, Q0) do { if (pf >= 2) hipLaunchKernelGGL((k_wgrad<QQ, Q0, 2>), tg, tb, 0, stream, L, nbatch, wg_tiles, nbp); \
                               else hipLaunchKernelGGL((k_wgrad<QQ, Q0, 1>), tg, tb, 0, stream, L, nbatch, wg_tiles, nbp); } while (0)
#define MEDGP_WG1(QQ) case QQ: MEDGP_WGL(QQ, 0); break;
        // 9 .. 16 components: two launches, each reducing its own components into its own slab planes (kernels_wgrad.h)
#define MEDGP_WG2(QR) case 8 + QR: MEDGP_WGL(8, 0); MEDGP_WGL(QR, 8); break;
        switch (c->use_v0 ? 0 : L.Q) {
        MEDGP_WG1(1) MEDGP_WG1(2) MEDGP_WG1(3) MEDGP_WG1(4) MEDGP_WG1(5) MEDGP_WG1(6) MEDGP_WG1(7) MEDGP_WG1(8)
        MEDGP_WG2(1) MEDGP_WG2(2) MEDGP_WG2(3) MEDGP_WG2(4) MEDGP_WG2(5) MEDGP_WG2(6) MEDGP_WG2(7) MEDGP_WG2(8)
#undef MEDGP_WG1
#undef MEDGP_WG2
#undef MEDGP_WGL
        default: from_slab = 0; break;   // Q > 16 (or MEDGP_V0): generic kernels below
        }
        if (!from_slab) lw.kid = -1;   // nothing was launched under this label: its events go back to the pool unread
        lw.finish();
        if (!from_slab) {
            { Launcher l(c, KID_LAUUM, stream); hipLaunchKernelGGL(k_lauum_v0, dim3(tri(nt64), nbatch), dim3(256), 0, stream, L); }
            const int nbins = L.Q * tri(L.D);
            { Launcher l(c, KID_GRADBINS, stream); hipLaunchKernelGGL(k_gradbins_v0, dim3((nbins + 255) / 256, nbatch), dim3(256), 0, stream, L); }
        }
    }
    if (flag_grad && from_slab) {
        const int nbins3 = 3 * L.Q * tri(L.D);
        Launcher l(c, KID_EPILOGUE, stream);
        hipLaunchKernelGGL(k_slabsum, dim3(nbatch, (nbins3 + 255) / 256), dim3(256), 0, stream, L);
    }
    if (nlml_dev) {
        Launcher l(c, KID_EPILOGUE, stream);
        // few entries: the hypers of an entry are spread over workgroups (H = 2954 at D = 64: 0.18 -> 0.07 ms for one entry); with a
        // workgroup per CU anyway, one part per entry is faster (each part stages S and A again: 0.09 vs 0.20 ms at 512 entries)
        const int nparts = (2 * nbatch >= c->num_cu) ? 1 : std::min(MEDGP_EPI_PARTS, (L.H + 255) / 256);
        hipLaunchKernelGGL(k_epilogue, dim3(nbatch, nparts), dim3(256), 0, stream, L, theta_dev, flag_grad, from_slab, nlml_dev, grad_dev, (int *)status_dev);
    }
    HIPCHK(c, hipGetLastError());
    return MEDGP_OK;
}


// The evaluation pipeline; everything is asynchronous and ordered on c->stream.  Every size class of the plan gets its own kernel
// chain; with more than one class the chains run on auxiliary streams forked from / joined into c->stream, so that the workgroup-per-
// patient launches of the small classes fill the CUs a look-ahead chain of the large ones leaves idle.
//
// Route rule.  Measured on MI355X, round 4 (scratch/route_sweep.py -> profiles/r04_route_table.txt; factorisation ms per call, nlml +
// gradient, LA = look-ahead schedule, 44 / 84 = k_cholinv<4,4> / <8,4>; the same table at D = 2 and D = 24):
//   N=128: 44 wins at every batch size (0.069 vs LA 0.073 at 8 entries, 0.081 vs 0.112 at 256)
//   N=256: LA <= 96 entries (0.127 / 0.168 vs 44: 0.168 / 0.182 at 8 / 96), 44 from 128 on (0.185 vs LA 0.197)
//   N=384: LA <= 128 (0.390 vs 84: 0.395), 84 from 160 on (0.412 vs LA 0.473)
//   N=512: LA <= 96 (0.560 vs 0.683), tie at 128 (0.709 / 0.705), 84 from 160 on (0.712 vs 0.885)
//   N=768 / 1024: LA <= 128 (1.78 vs 1.86; 3.46 vs 3.69), 84 from 160 on (1.91 vs 2.19; 3.74 vs 4.34)
// The LA time grows linearly with the batch, the single-workgroup time is flat up to one patient per CU.  For a uniform call that gave:
// never for two blocks; up to 7/16 #CU entries (112) for three or four blocks; up to 9/16 #CU (144) from five blocks on.  Round 5 states the
// same rule per size CLASS of a ragged call: with t(nb) the one-workgroup cost model (wg_cost) and S the summed cost of the entries not
// yet given to the look-ahead schedule, a class (largest first) takes the look-ahead schedule when  t(nb_max) * #CU * f >= S  (f = 7/16
// or 9/16 as above) -- i.e. when one of its entries on one workgroup would stick out of the average load per CU of everything that is
// left.  For a uniform call S = n t and the rule is the old one (n <= 112 / 144); in a ragged call the heavy tail is peeled off class by
// class until the rest is balanced.
// routes of the classes of a plan (the rule above); returns the look-ahead scratch entries, la_of[i] = index into them or -1
void choose_routes(const medgp_ctx *c, BatchPlan &P, std::vector<LaNeed> &las, std::vector<int> &la_of) {
    long long S = 0;
    for (const SizeClass &k : P.cls) S += k.tsum;
    las.clear();
    la_of.assign(P.cls.size(), -1);
    for (size_t i = 0; i < P.cls.size(); i++) {
        SizeClass &k = P.cls[i];
        bool la = false;
        if (!c->use_v0 && !c->pin_route && k.nbmax >= 2) {
            if (c->force_mc > 0) la = true;   // (forced, A-B and tests: also for two blocks)
            else if (c->force_mc == 0 && k.nbmax >= 3 && !c->no_classes) la = wg_cost(k.nbmax) * c->num_cu * (k.nbmax <= 4 ? 7 : 9) >= 16 * S;
            else if (c->force_mc == 0 && k.nbmax >= 3) la = k.count <= (c->num_cu * (k.nbmax <= 4 ? 7 : 9)) / 16;   // rounds 1-4: by entry count alone
        }
        if (la) {
            k.route = ROUTE_LA;
            S -= k.tsum;
            la_of[i] = (int)las.size();
            las.push_back({k.count, k.nbmax, k.ld, LaArgs{}});
        } else {
            // more patients than CUs: 4-wave workgroups, two per CU (the serial diagonal phase of one overlaps the MFMA phase of the
            // other).  At most one patient per CU: 8 waves (8 block slots per pass) once a step has more than 4 row blocks, else the
            // 4-wave shape, whose 4 slots already cover every block of n <= 256 (measured, 256 patients x N=256, D=2: <4,4> 0.211 ms,
            // <8,4> 0.232 ms -- half of its 8 slots idle).
            const int shape = c->pin_route ? 84 : (c->cholinv_nw ? c->cholinv_nw : ((k.count > c->num_cu || k.nbmax <= 4) ? 44 : 84));
            k.route = shape == 44 ? ROUTE_WG44 : ROUTE_WG84;
        }
    }
}

// what a plan needs of the look-ahead scratch arenas (doubles): the largest wave
void la_needs(const BatchPlan &P, const std::vector<LaNeed> &las, const std::vector<int> &la_of, size_t *part, size_t *small) {
    std::vector<size_t> wp(P.nwaves, 0), ws(P.nwaves, 0);
    for (size_t i = 0; i < P.cls.size(); i++)
        if (la_of[i] >= 0) { wp[P.cls[i].wave] += la_part_doubles(las[la_of[i]], P.with_u); ws[P.cls[i].wave] += la_small_doubles(las[la_of[i]]); }
    *part = *small = 0;
    for (int w = 0; w < P.nwaves; w++) { *part = std::max(*part, wp[w]); *small = std::max(*small, ws[w]); }
}

// End of the medgp_screen chunk that starts at entry e0 of the walk: entry e = (patient e / ninit of the walk, vector e % ninit); ns =
// the patients' sizes in walk order (largest first).  A chunk holds at most max_batch entries and at most screen_budget bytes of
// Gram matrices (8 ld^2 per entry: an nlml-only evaluation never forms U; ld taken at the upper end of the entry's size bucket, which
// bounds the leading dimension of whatever class it lands in).  Entries of >= 45 blocks (64 MB of matrix each) take the look-ahead
// schedule in any chunk this rule forms, and that schedule gains little beyond ~ 32 k block pairs per launch (measured on the four
// largest patients of the heavy-tailed cohort, N = 3258 .. 5832, 200 vectors each: 1234 / 1073 / 799 / 679 / 665 ms for chunks
// closed at 4 / 8 / 16 / 32 / 64 k block pairs, scratch/screen_work_sweep.sh): such a chunk is closed there or by the byte budget --
// N = 5832: 3 entries = 0.83 GB of matrices + 0.42 GB of scratch per chunk where round 5 took 32 entries = 26 GB + 9 GB (obtaining
// that much memory can cost seconds, see struct Arena).  The budget (2 GB) still gives every size its efficient route: 1024 entries of N <= 512 (one workgroup each, two per
// CU), 256 of N <= 1024 (one per CU), 64 of N <= 2048 (look-ahead schedule, saturated from 16 on).
size_t screen_chunk_end(const medgp_ctx *c, const std::vector<int> &ns, int ninit, size_t e0, size_t total, int max_entries) {
    size_t e = e0, bytes = 0;
    long long work = 0;
    while (e < total && (int)(e - e0) < max_entries) {
        const int nb = blocks64(ns[e / ninit]);
        const size_t ldb = (size_t)64 << size_bucket(nb), per = 8 * ldb * ldb;
        if (e > e0 && bytes + per > c->screen_budget) break;
        if (e > e0 && nb >= 45 && work >= c->screen_work) break;
        bytes += per; work += (long long)nb * nb; e++;
    }
    return e;
}

// How medgp_screen cuts `total` = walk_n.size() * ninit entries into chunks, whether it runs them on two lanes, and what ONE lane needs of
// every arena (doubles; the largest chunk, laid out once per distinct composition).  Shared with medgp_reserve_plan.
struct ScreenChunk { size_t e0, e1; };
struct ScreenCut {
    std::vector<ScreenChunk> chunks;
    bool two = false;
    int lane_rows = 0;
    size_t cap_mat = 0, cap_vec = 0, cap_tab = 0, cap_part = 0, cap_small = 0;
};
void screen_cut(const medgp_ctx *c, const std::vector<int> &walk_n, int ninit, ScreenCut &S) {
    const size_t total = walk_n.size() * (size_t)ninit;
    S.two = c->screen_lanes >= 2 && c->max_batch >= 2 && screen_chunk_end(c, walk_n, ninit, 0, total, c->max_batch) < total;
    S.lane_rows = S.two ? c->max_batch / 2 : c->max_batch;   // a lane's rows of the batch-indexed buffers = its chunks' entry cap
    S.chunks.clear();
    for (size_t e0 = 0; e0 < total;) { const size_t e = screen_chunk_end(c, walk_n, ninit, e0, total, S.lane_rows); S.chunks.push_back({e0, e}); e0 = e; }
    BatchPlan P;
    std::vector<LaNeed> las;
    std::vector<int> la_of, en;
    int lf = -1, ll = -1;
    size_t lc = 0;
    for (const ScreenChunk &ch : S.chunks) {
        const int nf = walk_n[ch.e0 / ninit], nl = walk_n[(ch.e1 - 1) / ninit];
        if (nf == nl && nf == lf && nl == ll && ch.e1 - ch.e0 == lc) continue;   // (runs of identical chunks: laid out once)
        en.resize(ch.e1 - ch.e0);
        for (size_t x = ch.e0; x < ch.e1; x++) en[x - ch.e0] = walk_n[x / ninit];
        layout_plan(c, en.data(), (int)en.size(), false, P);
        choose_routes(c, P, las, la_of);
        size_t lp = 0, ls = 0;
        la_needs(P, las, la_of, &lp, &ls);
        S.cap_mat = std::max(S.cap_mat, P.need_mat); S.cap_vec = std::max(S.cap_vec, P.need_vec); S.cap_tab = std::max(S.cap_tab, P.need_tab);
        S.cap_part = std::max(S.cap_part, lp); S.cap_small = std::max(S.cap_small, ls);
        lf = nf; ll = nl; lc = ch.e1 - ch.e0;
    }
}

// persist: the caller reads per-entry buffers of ALL entries after the call (factor exports, k_predict): such a call must fit one wave.
// Pp / st_main / aux0, naux: the plan to run, the stream its first class runs on and the auxiliary streams its other classes may use --
// the context's own (c->plan, c->stream, all of c->aux) unless medgp_screen runs two plans at once on two lanes.
int run_pipeline(medgp_ctx *c, int nbatch, int max_n, const double *theta_dev, int flag_grad, bool need_inverse, int min_n,
                 double *nlml_dev, double *grad_dev, int32_t *status_dev, bool store_ukk = false, bool persist = false,
                 BatchPlan *Pp = nullptr, hipStream_t st_main = nullptr, int aux0 = 0, int naux = kAuxStreams, hipEvent_t ev_fork_in = nullptr) {
    (void)max_n; (void)nbatch;
    BatchPlan &P = Pp ? *Pp : c->plan;
    if (!st_main) st_main = c->stream;
    hipEvent_t ev_fork = ev_fork_in ? ev_fork_in : c->ev_fork;
    const bool forms_u = flag_grad || need_inverse || store_ukk;
    if (forms_u && !P.with_u) return fail(c, MEDGP_ERR_ARG, "internal: plan laid out without Linv for a call that forms it");
    if (persist && P.nwaves > 1)
        return fail(c, MEDGP_ERR_CAPACITY, "the call's per-entry matrices exceed the memory budget of %zu GB (MEDGP_MEM_BUDGET_GB) and its outputs need all of them at once: split the call",
                    c->mem_budget >> 30);
    { int rc = ensure_arena(c, P.mat0 + P.need_mat, P.with_u ? P.mat0 + P.need_mat : 0, P.vec0 + P.need_vec, P.tab0 + P.need_tab, P.need_slab); if (rc) return rc; }
    c->last_has_inverse = (flag_grad || need_inverse) && P.nwaves == 1;
    std::vector<LaNeed> las;
    std::vector<int> la_of;
    choose_routes(c, P, las, la_of);
    const int nstr = std::max(0, std::min(c->class_streams, naux));
    for (int w = 0; w < P.nwaves; w++) {
        // this wave's classes [i0, i1) and their look-ahead scratch
        size_t i0 = 0, i1;
        while (i0 < P.cls.size() && P.cls[i0].wave != w) i0++;
        i1 = i0;
        while (i1 < P.cls.size() && P.cls[i1].wave == w) i1++;
        std::vector<LaNeed> wl;
        std::vector<int> wl_of(P.cls.size(), -1);
        for (size_t i = i0; i < i1; i++) if (la_of[i] >= 0) { wl_of[i] = (int)wl.size(); wl.push_back(las[la_of[i]]); }
        if (!wl.empty()) { int rc = ensure_la(c, wl, P.with_u, P.la_part0, P.la_small0); if (rc) return rc; }
        const bool fork = i1 - i0 > 1 && nstr > 0 && c->aux[aux0];
        if (fork) HIPCHK(c, hipEventRecord(ev_fork, st_main));
        bool used[kAuxStreams] = {false, false, false, false};
        for (size_t i = i0; i < i1; i++) {
            const SizeClass &k = P.cls[i];
            hipStream_t st = st_main;
            int ai = -1;
            if (fork && i > i0) {   // the wave's first class (its largest entries) stays on the call's stream
                ai = aux0 + (int)((i - i0 - 1) % nstr);
                st = c->aux[ai];
                if (!used[ai]) { HIPCHK(c, hipStreamWaitEvent(st, ev_fork, 0)); used[ai] = true; }
            }
            const MedgpDev V = class_view(c, P, k);
            int rc = run_pipeline_one(c, st, V, k.count, k.nbmax, theta_dev, flag_grad, need_inverse, min_n, nlml_dev, grad_dev, status_dev, store_ukk,
                                      P.en.data() + k.b0, k.route, wl_of[i] >= 0 ? &wl[wl_of[i]].A : nullptr);
            if (rc) return rc;
        }
        if (fork)
            for (int ai = aux0; ai < aux0 + nstr; ai++)
                if (used[ai]) {
                    HIPCHK(c, hipEventRecord(c->ev_join[ai], c->aux[ai]));
                    HIPCHK(c, hipStreamWaitEvent(st_main, c->ev_join[ai], 0));
                }
    }
    return MEDGP_OK;
}

}  // namespace

extern "C" {

int medgp_abi_version(void) { return 3; }

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
    { const char *e = getenv("MEDGP_CHOLINV_NW"); c->cholinv_nw = e ? atoi(e) : 0; }
    { const char *e = getenv("MEDGP_LA_PARK"); if (e) c->la_park = atoi(e); }
    { const char *e = getenv("MEDGP_LA_PARK_MAXBATCH"); if (e) c->la_park_maxbatch = atoi(e); }
    { const char *e = getenv("MEDGP_MULTI_CU"); c->force_mc = e ? atoi(e) : 0; }
    { const char *e = getenv("MEDGP_CLASS_STREAMS"); if (e) c->class_streams = std::max(0, std::min(kAuxStreams, atoi(e))); }
    { const char *e = getenv("MEDGP_NO_CLASSES"); c->no_classes = e ? atoi(e) : 0; }
    { const char *e = getenv("MEDGP_WGRAD_DEEP"); c->wgrad_deep = e ? std::max(1, atoi(e)) : -1; }
    { const char *e = getenv("MEDGP_DEBUG_FAIL_ATTEMPTS"); c->dbg_fail = e ? atoi(e) : 0; }
    { const char *e = getenv("MEDGP_MEM_BUDGET_GB"); if (e && atof(e) > 0) c->mem_budget = (size_t)(atof(e) * 1073741824.0); }
    { const char *e = getenv("MEDGP_SCREEN_LANES"); if (e && atoi(e) >= 1) c->screen_lanes = std::min(2, atoi(e)); }
    { const char *e = getenv("MEDGP_SCREEN_WORK"); if (e && atoll(e) > 0) c->screen_work = atoll(e); }
    { const char *e = getenv("MEDGP_SCREEN_BUDGET_GB"); if (e && atof(e) > 0) c->screen_budget = (size_t)(atof(e) * 1073741824.0); }
    for (int i = 0; i < kAuxStreams; i++) {
        (void)hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
    }
    (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    (void)hipStreamCreateWithFlags(&c->s_screen1, hipStreamNonBlocking);
    for (hipEvent_t *e : {&c->ev_fork1, &c->ev_screen0}) (void)hipEventCreateWithFlags(e, hipEventDisableTiming);
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess) c->num_cu = pr.multiProcessorCount; }
    *out = c;
    return MEDGP_OK;
}

void medgp_destroy(medgp_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto &e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    free_all(c);
    for (int i = 0; i < kAuxStreams; i++) {
        if (c->aux[i]) { (void)hipStreamSynchronize(c->aux[i]); (void)hipStreamDestroy(c->aux[i]); }
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->s_screen1) { (void)hipStreamSynchronize(c->s_screen1); (void)hipStreamDestroy(c->s_screen1); }
    for (hipEvent_t e : {c->ev_fork1, c->ev_screen0}) if (e) (void)hipEventDestroy(e);
    if (c->h_screen_tab) (void)hipHostFree(c->h_screen_tab);
    for (hipStream_t st : {c->s_up, c->s_down}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int i = 0; i < 2; i++) { if (c->ev_up[i]) (void)hipEventDestroy(c->ev_up[i]); if (c->ev_k[i]) (void)hipEventDestroy(c->ev_k[i]); }
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_bounce) (void)hipHostFree(c->h_bounce);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->d_small) (void)hipFree(c->d_small);
    for (int i = 0; i < 2; i++) {
        if (c->pin_buf[i]) (void)hipHostFree(c->pin_buf[i]);
        if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]);
        if (c->ev_lane[i]) (void)hipEventDestroy(c->ev_lane[i]);
    }
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
    if (c->s_down) HIPCHK(c, hipStreamSynchronize(c->s_down));   // (result downloads of the asynchronous lanes)
    free_retired(c);   // (every kernel of the context is queued on, or joined into, its stream: idle now)
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
    // patient rows: [0, S) grouped by output (what the gradient kernels need); [S, 2S) the same patient in the CALLER's
    // observation order, filled only when that order is not already grouped (used for caller-order factors)
    if ((rc = dalloc(c, &c->d_pn, 2 * S))) return rc;
    if ((rc = dalloc(c, &c->d_pt, 2 * S * ldn))) return rc;
    if ((rc = dalloc(c, &c->d_py, 2 * S * ldn))) return rc;
    if ((rc = dalloc(c, &c->d_pmeta, 2 * S * ldn))) return rc;
    if ((rc = dalloc(c, &c->d_pseg, 2 * S * (D + 1)))) return rc;
    if ((rc = dalloc(c, &c->d_proff, 2 * S * (D + 1)))) return rc;
    if ((rc = dalloc(c, &c->d_pcoff, 2 * S * (D + 1)))) return rc;
    if ((rc = dalloc(c, &c->d_one_slot, 1))) return rc;
    if ((rc = dalloc(c, &c->d_prior, S * H))) return rc;
    if ((rc = dalloc(c, &c->d_prior_on, S))) return rc;
    if ((rc = dalloc(c, &c->d_bslot, B))) return rc;
    if ((rc = dalloc(c, &c->d_bpos, B))) return rc;
    if ((rc = dalloc(c, &c->d_tpos, B))) return rc;
    c->d_screen_theta = nullptr; c->screen_theta_cap = 0; c->tpos_on = false;   // (freed by free_all above)
    if ((rc = dalloc(c, &c->d_status, B))) return rc;
    if ((rc = dalloc(c, &c->d_jit, B))) return rc;
    if ((rc = dalloc(c, &c->d_bn, B))) return rc;
    if ((rc = dalloc(c, &c->d_theta, B * H))) return rc;
    if ((rc = dalloc(c, &c->d_nlml, B))) return rc;
    if ((rc = dalloc(c, &c->d_grad, B * H))) return rc;
    if ((rc = dalloc(c, &c->d_status_out, B))) return rc;
    if ((rc = dalloc(c, &c->d_theta1, B * H))) return rc;
    if ((rc = dalloc(c, &c->d_nlml1, B))) return rc;
    if ((rc = dalloc(c, &c->d_grad1, B * H))) return rc;
    if ((rc = dalloc(c, &c->d_status1, B))) return rc;
    c->lane_pending[0] = c->lane_pending[1] = false;
    c->d_prior_stage = nullptr; c->d_prior_slots = nullptr; c->prior_stage_rows = 0;   // freed by free_all above
    MedgpDev &L = c->dev;
    L.kidx = c->kidx; L.Q = c->Q; L.D = c->D; L.R = c->R; L.H = c->H; L.nlik = c->nlik;
    L.ldn = ldn; L.pld = ldn; L.max_slots = max_slots; L.max_batch = max_batch;
    L.bpos = nullptr;
    L.tpos = nullptr;
    L.hyp_stride = (int)(D + Q * D * D + 2 * Q);
    L.pi = c->pi;
    L.dbg_fail = c->dbg_fail;
    double *hyp, *scal, *Sb, *SMb, *SVb;
    L.slab_R = ldn / 16 + (int)D;
    L.slab_C = ldn / 64 + (int)D;
    L.slab_stride = (size_t)3 * Q * L.slab_R * L.slab_C;
    c->full_mat = B * ldn * ldn; c->full_vec = B * ldn; c->full_tab = B * Q * ldn; c->full_slab = B * L.slab_stride;
    L.Kmat = L.Linv = L.z = L.alpha = L.wdiag = L.cs = L.sn = L.slab = nullptr;
    {   // a wave of a call never uses more than the budget, and the budget never more than 70 % of what the device has free now
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr > 0) {
            c->mem_budget = std::min(c->mem_budget, fr / 10 * 7);
            c->screen_budget = std::min(c->screen_budget, c->mem_budget);
        }
    }
    // every configuration whose capacities stay below 8 GB of matrices (all of BASELINE.json's) is allocated in full here; beyond that
    // (a ragged cohort: max_batch x max_n^2 would not fit) the arenas start as token blocks and are sized by medgp_reserve_plan, or grow with the calls
    if (2 * c->full_mat * sizeof(double) <= kArenaEager) rc = ensure_arena(c, c->full_mat, c->full_mat, c->full_vec, c->full_tab, c->full_slab, true);
    else rc = ensure_arena(c, 0, 0, 0, 0, 0, true);   // (a token block each: the views never hold null pointers)
    if (rc) return rc;
    double *xk;
    if ((rc = dalloc(c, &xk, B * 64 * 64))) return rc;
    L.xk = xk; L.jit = c->d_jit; L.bn = c->d_bn;
    if ((rc = dalloc(c, &hyp, B * L.hyp_stride))) return rc;
    if ((rc = dalloc(c, &scal, B * 4))) return rc;
    if ((rc = dalloc(c, &L.epi_lp, B * MEDGP_EPI_PARTS))) return rc;
    if ((rc = dalloc(c, &L.epi_ticket, B))) return rc;
    HIPCHK(c, hipMemsetAsync(L.epi_ticket, 0, B * sizeof(int), c->stream));
    if ((rc = dalloc(c, &Sb, B * Q * D * D))) return rc;
    if ((rc = dalloc(c, &SMb, B * Q * D * D))) return rc;
    if ((rc = dalloc(c, &SVb, B * Q * D * D))) return rc;
    L.pn = c->d_pn; L.pt = c->d_pt; L.py = c->d_py; L.pmeta = c->d_pmeta; L.pseg = c->d_pseg;
    L.proff = c->d_proff; L.pcoff = c->d_pcoff;
    L.prior = c->d_prior; L.prior_on = c->d_prior_on; L.bslot = c->d_bslot;
    L.hyp = hyp; L.scal = scal;
    L.status = c->d_status; L.S = Sb; L.SM = SMb; L.SV = SVb;
    HIPCHK(c, hipMemsetAsync(c->d_prior_on, 0, S, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_pn, 0, 2 * S * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_n.assign(max_slots, -1);
    c->h_perm.assign(max_slots, {});
    c->h_perm_identity.assign(max_slots, 1);
    c->h_bslot.assign(max_batch, -1);
    c->last_nbatch = 0;
    c->plan = BatchPlan{};
    c->last_has_inverse = false;
    c->pred_cap = 0;
    c->d_stage = nullptr;   // freed by free_all above
    if (c->h_stage) { (void)hipHostFree(c->h_stage); c->h_stage = nullptr; }
    c->stage_cap = 0;
    c->stage_pending = false;
    return MEDGP_OK;
}

int medgp_reserve_plan(medgp_ctx *c, int count, const int32_t *n, int ninit) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (count < 1 || !n || ninit < 0) return fail(c, MEDGP_ERR_ARG, "medgp_reserve_plan: bad argument");
    for (int i = 0; i < count; i++)
        if (n[i] < 0 || n[i] > c->max_n) return fail(c, MEDGP_ERR_CAPACITY, "medgp_reserve_plan: n[%d] = %d outside [0, %d]", i, n[i], c->max_n);
    HIPCHK(c, hipSetDevice(c->device));
    // largest first, as medgp_screen walks them and as the lock-step trainer admits them
    std::vector<int> ns(n, n + count);
    std::stable_sort(ns.begin(), ns.end(), [](int a, int b) { return a > b; });
    size_t nk = 0, nu = 0, nvec = 0, ntab = 0, nslab = 0, npart = 0, nsmall = 0;
    BatchPlan P;
    std::vector<LaNeed> las;
    std::vector<int> la_of;
    auto take = [&](const int *en, int nb, bool with_u) {
        layout_plan(c, en, nb, with_u, P);
        choose_routes(c, P, las, la_of);
        size_t lp = 0, ls = 0;
        la_needs(P, las, la_of, &lp, &ls);
        nk = std::max(nk, P.need_mat); if (with_u) nu = std::max(nu, P.need_mat);
        nvec = std::max(nvec, P.need_vec); ntab = std::max(ntab, P.need_tab); nslab = std::max(nslab, P.need_slab);
        npart = std::max(npart, lp); nsmall = std::max(nsmall, ls);
    };
    // (a) one nlml + gradient call over the announced patients (the largest max_batch of them)
    take(ns.data(), std::min(count, c->max_batch), true);
    // (b) the chunks medgp_screen forms of them (two lanes: twice the largest chunk)
    if (ninit > 0) {
        ScreenCut SC;
        screen_cut(c, ns, ninit, SC);
        const size_t f = SC.two ? 2 : 1;
        nk = std::max(nk, f * SC.cap_mat); nvec = std::max(nvec, f * SC.cap_vec); ntab = std::max(ntab, f * SC.cap_tab);
        npart = std::max(npart, f * SC.cap_part); nsmall = std::max(nsmall, f * SC.cap_small);
    }
    int rc;
    if ((rc = ensure_arena(c, nk, nu, nvec, ntab, nslab, true))) return rc;
    if (npart && (rc = arena_ensure(c, AR_LA_PART, npart * sizeof(double), 0, true, nullptr))) return rc;
    if (nsmall && (rc = arena_ensure(c, AR_LA_SMALL, nsmall * sizeof(double), 0, true, nullptr))) return rc;
    // a set-up call: wait for the context's streams once and give the replaced blocks back now, so that the loop that follows starts with
    // nothing left to release
    if (!c->retired.empty()) { if ((rc = sync_ctx_streams(c))) return rc; free_retired(c); }
    return MEDGP_OK;
}

int medgp_alloc_stats(const medgp_ctx *c, double *seconds, int64_t *calls, int64_t *arena_bytes) {
    if (!c) return MEDGP_ERR_ARG;
    if (seconds) *seconds = c->alloc_s;
    if (calls) *calls = c->alloc_calls;
    if (arena_bytes) { int64_t b = 0; for (int i = 0; i < AR_COUNT; i++) b += (int64_t)c->arena[i].bytes; *arena_bytes = b; }
    return MEDGP_OK;
}

// ---- patient upload: any number of patients are packed into one pinned staging buffer, copied with ONE H2D transfer and
// scattered into their padded slot rows by k_scatter_patients; nothing waits for the device (the staging buffer is
// guarded by an event that is only waited for when the next upload starts).
namespace {
struct UpEntry { int slot, n; const int32_t *meta; const float *t, *y; };

inline size_t up_payload_bytes(int n, int D) { return ((size_t)n * 20 + (size_t)(D + 1) * 12 + 7) & ~(size_t)7; }

int upload_patients(medgp_ctx *c, const std::vector<UpEntry> &ents) {
    const int D = c->D, S = c->max_slots, ldn = c->ldn;
    const bool use_meta = (c->kidx == MEDGP_KERNEL_LMC_SM);
    // validate everything before touching any state
    std::vector<uint8_t> seen(ents.size() > 1 ? (size_t)S : 0, 0);
    for (const UpEntry &e : ents) {
        if (e.slot < 0 || e.slot >= S) return fail(c, MEDGP_ERR_CAPACITY, "slot %d outside [0, %d)", e.slot, S);
        if (!seen.empty()) {   // two scatter workgroups would write the same rows, and the host mirrors would keep whichever came last
            if (seen[e.slot]) return fail(c, MEDGP_ERR_ARG, "slot %d appears twice in one packed upload", e.slot);
            seen[e.slot] = 1;
        }
        if (e.n < 0 || e.n > c->max_n) return fail(c, MEDGP_ERR_CAPACITY, "n = %d outside [0, %d]", e.n, c->max_n);
        if (e.n > 0 && (!e.t || !e.y)) return fail(c, MEDGP_ERR_ARG, "t / y is NULL");
        if (use_meta && e.n > 0 && !e.meta) return fail(c, MEDGP_ERR_ARG, "meta is NULL for the multi-output kernel");
        if (use_meta)
            for (int i = 0; i < e.n; i++)
                if (e.meta[i] < 0 || e.meta[i] >= D) return fail(c, MEDGP_ERR_ARG, "meta[%d] = %d outside [0, %d)", i, e.meta[i], D);
    }
    // worst case: every patient also needs its caller-order copy
    size_t bytes = 0;
    for (const UpEntry &e : ents) bytes += 2 * (sizeof(MedgpUpHdr) + up_payload_bytes(e.n, D));
    HIPCHK(c, hipSetDevice(c->device));
    if (c->stage_pending) { HIPCHK(c, hipEventSynchronize(c->ev_stage)); c->stage_pending = false; }
    if (bytes > c->stage_cap) {
        const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        HIPCHK(c, hipHostMalloc((void **)&c->h_stage, cap, hipHostMallocDefault));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // the old device buffer may still be read by a queued scatter
        if (c->d_stage) {
            (void)hipFree(c->d_stage);
            c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), (void *)c->d_stage), c->allocs.end());
            c->d_stage = nullptr;
        }
        int rc = dalloc(c, &c->d_stage, cap);
        if (rc) return rc;
        c->stage_cap = cap;
    }
    if (!c->ev_stage) HIPCHK(c, hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
    // count the device entries (grouped copy of every patient + caller-order copy of the ungrouped ones)
    std::vector<std::vector<int>> perms(ents.size());
    std::vector<uint8_t> ident(ents.size(), 1);
    int nent = 0;
    for (size_t k = 0; k < ents.size(); k++) {
        const UpEntry &e = ents[k];
        std::vector<int> &perm = perms[k];
        perm.resize(e.n);
        if (use_meta) {   // stable grouping by output (counting sort)
            std::vector<int> pos(D + 1, 0);
            for (int i = 0; i < e.n; i++) pos[e.meta[i] + 1]++;
            for (int d = 0; d < D; d++) pos[d + 1] += pos[d];
            for (int i = 0; i < e.n; i++) perm[pos[e.meta[i]]++] = i;
        } else {
            for (int i = 0; i < e.n; i++) perm[i] = i;
        }
        for (int i = 0; i < e.n; i++) if (perm[i] != i) { ident[k] = 0; break; }
        nent += ident[k] ? 1 : 2;
    }
    MedgpUpHdr *hdr = (MedgpUpHdr *)c->h_stage;
    size_t off = (size_t)nent * sizeof(MedgpUpHdr);
    int ie = 0;
    auto pack = [&](const UpEntry &e, int dev_slot, const int *perm) {
        hdr[ie].slot = dev_slot; hdr[ie].n = e.n; hdr[ie].off = (long long)off; ie++;
        double *pt = (double *)(c->h_stage + off), *py = pt + e.n;
        int *pm = (int *)(py + e.n), *seg = pm + e.n, *roff = seg + (D + 1), *coff = roff + (D + 1);
        for (int i = 0; i < e.n; i++) {
            const int src = perm ? perm[i] : i;
            pt[i] = (double)e.t[src];
            py[i] = (double)e.y[src];   // zero mean: ref mean/c_meanfunc_zero.cpp:32-50
            pm[i] = use_meta ? e.meta[src] : 0;
        }
        for (int d = 0; d <= D; d++) seg[d] = roff[d] = coff[d] = 0;
        if (perm || !use_meta) {   // grouped copy: output segments + slab slots of k_wgrad (16-row / 64-column pieces per output)
            if (use_meta) { for (int i = 0; i < e.n; i++) seg[pm[i] + 1]++; for (int d = 0; d < D; d++) seg[d + 1] += seg[d]; }
            else seg[1] = e.n;
            for (int d = 0; d < D; d++) {
                const int a = seg[d], z = seg[d + 1];
                roff[d + 1] = roff[d] + (z > a ? (z - 1) / 16 - a / 16 + 1 : 0);
                coff[d + 1] = coff[d] + (z > a ? (z - 1) / 64 - a / 64 + 1 : 0);
            }
        }
        off += up_payload_bytes(e.n, D);
    };
    for (size_t k = 0; k < ents.size(); k++) {
        pack(ents[k], ents[k].slot, perms[k].data());
        if (!ident[k]) pack(ents[k], ents[k].slot + S, nullptr);   // caller order (factor export / predict in that order)
    }
    HIPCHK(c, hipMemcpyAsync(c->d_stage, c->h_stage, off, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_stage, c->stream));
    c->stage_pending = true;
    hipLaunchKernelGGL(k_scatter_patients, dim3(nent), dim3(256), 0, c->stream, (const char *)c->d_stage, D, ldn, c->d_pn, c->d_pt,
                       c->d_py, c->d_pmeta, c->d_pseg, c->d_proff, c->d_pcoff);
    HIPCHK(c, hipGetLastError());
    for (size_t k = 0; k < ents.size(); k++) {
        const int s = ents[k].slot;
        c->h_n[s] = ents[k].n;
        c->h_perm[s] = std::move(perms[k]);
        c->h_perm_identity[s] = ident[k];
    }
    c->last_nbatch = 0;   // cached batch selection / factors no longer describe the resident patients
    c->last_has_inverse = false;
    return MEDGP_OK;
}
}  // namespace

int medgp_set_patient(medgp_ctx *c, int slot, int n, const int32_t *meta, const float *t, const float *y) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    return upload_patients(c, {UpEntry{slot, n, meta, t, y}});
}

int medgp_set_patients(medgp_ctx *c, int nslots, const int32_t *slots, const int64_t *offsets, const int32_t *meta,
                       const float *t, const float *y) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (nslots < 1 || !slots || !offsets) return fail(c, MEDGP_ERR_ARG, "medgp_set_patients: bad argument");
    std::vector<UpEntry> ents(nslots);
    for (int k = 0; k < nslots; k++) {
        const int64_t a = offsets[k], z = offsets[k + 1];
        if (a < 0 || z < a || z - a > c->max_n) return fail(c, MEDGP_ERR_CAPACITY, "offsets[%d..%d] = %lld..%lld: bad length", k, k + 1, (long long)a, (long long)z);
        ents[k] = UpEntry{slots[k], (int)(z - a), meta ? meta + a : nullptr, t ? t + a : nullptr, y ? y + a : nullptr};
    }
    return upload_patients(c, ents);
}

namespace {
// rows: nrows descriptors of H hypers each (flag == NULL: "no prior"); slots == NULL: ONE row for every slot of the context.
// One pinned staging copy, one H2D transfer, one scatter kernel; nothing waits for the device.
int upload_priors(medgp_ctx *c, int nrows, const int32_t *slots, const uint8_t *flag, const int32_t *type, const uint8_t *is_exp,
                  const float *p0, const float *p1) {
    const int H = c->H;
    if (flag && (!type || !is_exp || !p0 || !p1)) return fail(c, MEDGP_ERR_ARG, "prior arrays must all be given");
    std::vector<uint8_t> seen((slots && nrows > 1) ? (size_t)c->max_slots : 0, 0);
    for (int k = 0; slots && k < nrows; k++) {
        if (slots[k] < 0 || slots[k] >= c->max_slots) return fail(c, MEDGP_ERR_CAPACITY, "slot %d outside [0, %d)", slots[k], c->max_slots);
        if (!seen.empty()) {   // two scatter workgroups would write the same prior rows: which one wins is undefined
            if (seen[slots[k]]) return fail(c, MEDGP_ERR_ARG, "slot %d appears twice in one medgp_set_priors", slots[k]);
            seen[slots[k]] = 1;
        }
    }
    if (flag)
        for (size_t h = 0; h < (size_t)nrows * H; h++)
            if (type[h] < -1 || type[h] > 2) return fail(c, MEDGP_ERR_ARG, "prior type[%zu] = %d unsupported (KDE prior type 3 is never constructed by the reference's mains)", h, type[h]);
    HIPCHK(c, hipSetDevice(c->device));
    if ((size_t)nrows > c->prior_stage_rows) {
        { int rcs = sync_ctx_streams(c); if (rcs) return rcs; }   // (the context's streams, not the device: other contexts keep running)
        for (void *q : {(void *)c->d_prior_stage, (void *)c->d_prior_slots})
            if (q) { (void)hipFree(q); c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), q), c->allocs.end()); }
        c->d_prior_stage = nullptr; c->d_prior_slots = nullptr; c->prior_stage_rows = 0;
        const size_t rows = std::max<size_t>(nrows, 16);
        int rc;
        if ((rc = dalloc(c, &c->d_prior_stage, rows * H))) return rc;
        if ((rc = dalloc(c, &c->d_prior_slots, rows))) return rc;
        c->prior_stage_rows = rows;
    }
    const size_t row_bytes = sizeof(MedgpPrior) * (size_t)H, slot_bytes = ((sizeof(int) * (size_t)nrows) + 15) & ~(size_t)15;
    void *pin = nullptr;
    int rc = pin_stage(c, slot_bytes + row_bytes * nrows, &pin);
    if (rc) return rc;
    int *hs = (int *)pin;
    MedgpPrior *hp = (MedgpPrior *)((char *)pin + slot_bytes);
    for (int k = 0; k < nrows; k++) {
        hs[k] = slots ? slots[k] : -1;
        for (int h = 0; h < H; h++) {
            MedgpPrior p{};
            const size_t e = (size_t)k * H + h;
            if (flag) {
                p.p0 = p0[e]; p.p1 = p1[e]; p.type = (int8_t)type[e]; p.flag = flag[e] ? 1 : 0; p.is_exp = is_exp[e] ? 1 : 0;
                if (type[e] == 2) p.lg2b = logf(2.0f * p1[e]);   // the reference's float log (MedgpPrior::lg2b)
            }
            else p.type = -1;
            hp[e] = p;
        }
    }
    if (slots) HIPCHK(c, hipMemcpyAsync(c->d_prior_slots, hs, sizeof(int) * nrows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_prior_stage, hp, row_bytes * nrows, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scatter_priors, dim3(slots ? nrows : c->max_slots), dim3(256), 0, c->stream, (const MedgpPrior *)c->d_prior_stage,
                       slots ? (const int *)c->d_prior_slots : (const int *)nullptr, H, c->d_prior, c->d_prior_on, (uint8_t)(flag ? 1 : 0));
    HIPCHK(c, hipGetLastError());
    return MEDGP_OK;
}
}  // namespace

int medgp_set_prior(medgp_ctx *c, int slot, const uint8_t *flag, const int32_t *type, const uint8_t *is_exp,
                    const float *p0, const float *p1) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (slot < -1 || slot >= c->max_slots) return fail(c, MEDGP_ERR_CAPACITY, "slot %d outside [-1, %d)", slot, c->max_slots);
    const int32_t s1 = slot;
    return upload_priors(c, 1, slot < 0 ? nullptr : &s1, flag, type, is_exp, p0, p1);
}

int medgp_set_priors(medgp_ctx *c, int nslots, const int32_t *slots, const uint8_t *flag, const int32_t *type,
                     const uint8_t *is_exp, const float *p0, const float *p1) {
    if (!c) return MEDGP_ERR_ARG;
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (nslots < 1 || !slots) return fail(c, MEDGP_ERR_ARG, "medgp_set_priors: bad argument");
    return upload_priors(c, nslots, slots, flag, type, is_exp, p0, p1);
}

int medgp_nlml_grad_device(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta_dev, int flag_grad,
                           double *nlml_dev, double *grad_dev, int32_t *status_dev) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta_dev || !nlml_dev) return fail(c, MEDGP_ERR_ARG, "NULL argument");
    if (flag_grad & ~(MEDGP_FLAG_GRAD | MEDGP_FLAG_KEEP_FACTOR)) return fail(c, MEDGP_ERR_ARG, "unknown bits in flag_grad = %d", flag_grad);
    const int grad = flag_grad & MEDGP_FLAG_GRAD;
    const bool keep = (flag_grad & MEDGP_FLAG_KEEP_FACTOR) != 0;
    if (grad && !grad_dev) return fail(c, MEDGP_ERR_ARG, "grad is NULL with flag_grad set");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    HIPCHK(c, hipSetDevice(c->device));
    int max_n = 0, rc;
    // factor wanted but no gradient: patients that were not uploaded grouped by output are evaluated in the CALLER's order,
    // so that L^-1 is the factor the reference would hand to GP_Regression::predict (ref: core/gp_regression.cpp:181-196)
    if ((rc = set_batch(c, nbatch, slots, &max_n, keep && !grad, grad || keep))) return rc;
    return run_pipeline(c, nbatch, max_n, theta_dev, grad, keep, 3, nlml_dev, grad_dev, status_dev, false, keep);
}

int medgp_nlml_grad(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta, int flag_grad, double *nlml,
                    double *grad, int32_t *status) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta || !nlml) return fail(c, MEDGP_ERR_ARG, "NULL argument");
    if ((flag_grad & MEDGP_FLAG_GRAD) && !grad) return fail(c, MEDGP_ERR_ARG, "grad is NULL with flag_grad set");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (nbatch < 1 || nbatch > c->max_batch) return fail(c, MEDGP_ERR_CAPACITY, "nbatch %d outside [1, %d]", nbatch, c->max_batch);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t H = c->H;
    const bool want_grad = (flag_grad & MEDGP_FLAG_GRAD) != 0;
    // Small calls (a few patients, the D = 2 configurations): the pageable copies of the runtime cost more than the kernels they
    // bracket (256 x N=256, D=2: 0.41 ms of kernels, 0.49 ms per call).  Up to 1 MB the arguments travel through pinned memory:
    // theta through the upload ring, the results into a pinned landing area, one host memcpy each.
    const size_t th_bytes = sizeof(double) * nbatch * H, out_bytes = sizeof(double) * nbatch * (1 + (want_grad ? H : 0)) + sizeof(int32_t) * nbatch;
    const size_t kSmall = (size_t)1 << 20;
    if (th_bytes <= kSmall && out_bytes <= kSmall) {
        if (!c->h_small) HIPCHK(c, hipHostMalloc((void **)&c->h_small, kSmall + 64, hipHostMallocDefault));
        if (!c->d_small) HIPCHK(c, hipMalloc((void **)&c->d_small, kSmall + 64));
        void *pin = nullptr;
        int rcp = pin_stage(c, th_bytes, &pin);
        if (rcp) return rcp;
        std::memcpy(pin, theta, th_bytes);
        HIPCHK(c, hipMemcpyAsync(c->d_theta, pin, th_bytes, hipMemcpyHostToDevice, c->stream));
        // the kernels write nlml, gradients and status into ONE device block laid out like the landing area: one copy instead of three
        // (each copy is a packet of its own on the stream: ~ 4 us apiece behind a 0.4 ms call)
        double *dn = (double *)c->d_small, *dg = dn + nbatch;
        int32_t *ds = (int32_t *)(dg + (want_grad ? (size_t)nbatch * H : 0));
        int rc = medgp_nlml_grad_device(c, nbatch, slots, c->d_theta, flag_grad, dn, want_grad ? dg : nullptr, ds);
        if (rc) return rc;
        double *hn = (double *)c->h_small, *hg = hn + nbatch;
        int32_t *hs = (int32_t *)(hg + (want_grad ? (size_t)nbatch * H : 0));
        HIPCHK(c, hipMemcpyAsync(hn, dn, out_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memcpy(nlml, hn, sizeof(double) * nbatch);
        if (want_grad) std::memcpy(grad, hg, sizeof(double) * nbatch * H);
        if (status) std::memcpy(status, hs, sizeof(int32_t) * nbatch);
        free_retired(c);   // (the stream is idle: blocks that outgrown arenas left behind can go)
        return MEDGP_OK;
    }
    // (this path shares lane 0's result staging: a download of that lane still in flight on the copy stream must have read it first)
    if (c->lane_pending[0] && c->ev_lane[0]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lane[0], 0));
    HIPCHK(c, hipMemcpyAsync(c->d_theta, theta, th_bytes, hipMemcpyHostToDevice, c->stream));
    int rc = medgp_nlml_grad_device(c, nbatch, slots, c->d_theta, flag_grad, c->d_nlml, c->d_grad, c->d_status_out);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(nlml, c->d_nlml, sizeof(double) * nbatch, hipMemcpyDeviceToHost, c->stream));
    if (want_grad) HIPCHK(c, hipMemcpyAsync(grad, c->d_grad, sizeof(double) * nbatch * H, hipMemcpyDeviceToHost, c->stream));
    if (status) HIPCHK(c, hipMemcpyAsync(status, c->d_status_out, sizeof(int32_t) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_retired(c);
    return MEDGP_OK;
}

int medgp_screen(medgp_ctx *c, int nslots, const int32_t *slots, int ninit, const double *theta, double *nlml, int32_t *status) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta || !nlml || nslots < 1 || ninit < 1) return fail(c, MEDGP_ERR_ARG, "medgp_screen: bad argument");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t H = c->H, total = (size_t)nslots * ninit;
    for (int s = 0; s < nslots; s++)
        if (slots[s] < 0 || slots[s] >= c->max_slots || c->h_n[slots[s]] < 0) return fail(c, MEDGP_ERR_ARG, "slots[%d] = %d is not a resident patient", s, slots[s]);
    if ((size_t)ninit * H > c->screen_theta_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->d_screen_theta) { (void)hipFree(c->d_screen_theta); c->allocs.erase(std::remove(c->allocs.begin(), c->allocs.end(), (void *)c->d_screen_theta), c->allocs.end()); c->d_screen_theta = nullptr; }
        int rc = dalloc(c, &c->d_screen_theta, (size_t)ninit * H);
        if (rc) return rc;
        c->screen_theta_cap = (size_t)ninit * H;
    }
    // (the result staging is lane 0's: a download of that lane still in flight on the copy stream must have read it first)
    if (c->lane_pending[0] && c->ev_lane[0]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_lane[0], 0));
    HIPCHK(c, hipMemcpyAsync(c->d_screen_theta, theta, sizeof(double) * ninit * H, hipMemcpyHostToDevice, c->stream));
    // pinned landing area of all results: [nlml: total doubles | status: total ints]
    const size_t land = sizeof(double) * total + sizeof(int32_t) * total;
    if (land > c->bounce_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->h_bounce) (void)hipHostFree(c->h_bounce);
        c->h_bounce = nullptr; c->bounce_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_bounce, land + land / 4, hipHostMallocDefault));
        c->bounce_cap = land + land / 4;
    }
    double *hn = (double *)c->h_bounce;
    int32_t *hs = (int32_t *)(c->h_bounce + sizeof(double) * total);
    // The patients are walked LARGEST FIRST whatever the caller's order (results go back to the caller's rows at the end): a chunk that
    // holds the last vectors of one patient and the first of the next then holds two patients of similar size -- one size class, one
    // launch set -- where the caller's order (the trainer's read-ahead delivers patients as its readers finish) would put a handful of
    // entries of a very different size into a class and a kernel chain of their own: measured on the 512-patient heavy-tailed cohort,
    // 1000 vectors each: 8.7-11.3 s in arrival order against 5.0 s sorted.
    std::vector<int> perm(nslots);
    for (int s = 0; s < nslots; s++) perm[s] = s;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return c->h_n[slots[a]] > c->h_n[slots[b]]; });
    std::vector<int> walk_n(nslots);
    for (int s = 0; s < nslots; s++) walk_n[s] = c->h_n[slots[perm[s]]];
    // ---- chunks of consecutive (patient, vector) entries (screen_chunk_end).  ONE chunk (everything fits one call): the context's own
    // plan on its own stream, exactly the call medgp_nlml_grad(flag_grad = 0) makes of the same entries.  SEVERAL chunks: they
    // alternate between two LANES -- two streams, each with its own rows of the batch-indexed buffers, its own half of the arenas and
    // its own auxiliary streams -- so that the assembly of one chunk (fp64 VALU bound) runs beside the factorisation of the other (MFMA
    // + latency bound) and the tail of one chunk's launch is filled by the other's.  Measured before it was built, with two contexts
    // screening the same patient at once: 2.24 against 2.48 ms per 1000 evaluations at N = 512 (scratch/screen_two_ctx.py).  A chunk of a
    // two-lane call holds at most max_batch / 2 entries (its rows of the buffers).
    ScreenCut SC;
    screen_cut(c, walk_n, ninit, SC);
    const bool two = SC.two;
    const std::vector<ScreenChunk> &chunks = SC.chunks;
    const int lane_rows = SC.lane_rows;
    const size_t cap_mat = SC.cap_mat, cap_vec = SC.cap_vec, cap_tab = SC.cap_tab, cap_part = SC.cap_part, cap_small = SC.cap_small;
    BatchPlan lane_plan[2];
    if (two) {
        int rc0;
        if ((rc0 = ensure_arena(c, 2 * cap_mat, 0, 2 * cap_vec, 2 * cap_tab, 0))) return rc0;
        if (cap_part && (rc0 = arena_ensure(c, AR_LA_PART, 2 * cap_part * sizeof(double), 0, false, nullptr))) return rc0;
        if (cap_small && (rc0 = arena_ensure(c, AR_LA_SMALL, 2 * cap_small * sizeof(double), 0, false, nullptr))) return rc0;
        // the tables of ALL chunks in one pinned block: a lane's copies read their own region, which nobody rewrites during the call
        // (the two-buffer upload ring is guarded by events on c->stream only)
        const size_t tab_bytes = sizeof(int) * 3 * total;
        if (tab_bytes > c->screen_tab_cap) {
            { int rcs = sync_ctx_streams(c); if (rcs) return rcs; }
            if (c->h_screen_tab) (void)hipHostFree(c->h_screen_tab);
            c->h_screen_tab = nullptr; c->screen_tab_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->h_screen_tab, tab_bytes + tab_bytes / 4, hipHostMallocDefault));
            c->screen_tab_cap = tab_bytes + tab_bytes / 4;
        }
        // lane 1 starts behind everything queued on the context's stream so far (the theta block, earlier calls that use the arenas)
        HIPCHK(c, hipEventRecord(c->ev_screen0, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->s_screen1, c->ev_screen0, 0));
    }
    std::vector<int32_t> cs;
    std::vector<int> tp, en;
    int rc = MEDGP_OK;
    c->tpos_on = true;
    for (size_t ci = 0; ci < chunks.size() && rc == MEDGP_OK; ci++) {
        const size_t e0 = chunks[ci].e0, e1 = chunks[ci].e1;
        const int nb = (int)(e1 - e0);
        cs.clear(); tp.clear();
        for (size_t x = e0; x < e1; x++) { cs.push_back(slots[perm[x / ninit]]); tp.push_back((int)(x % ninit)); }
        hipError_t he = hipSuccess;
        if (!two) {
            int max_n = 0;
            if ((rc = set_batch(c, nb, cs.data(), &max_n, false, false))) break;
            {   // theta rows in the plan's internal order
                void *pin = nullptr;
                if ((rc = pin_stage(c, sizeof(int) * nb, &pin))) break;
                int *hp = (int *)pin;
                for (int i = 0; i < nb; i++) hp[i] = tp[c->plan.order[i]];
                he = hipMemcpyAsync(c->d_tpos, hp, sizeof(int) * nb, hipMemcpyHostToDevice, c->stream);
                if (he != hipSuccess) { rc = fail(c, MEDGP_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(he)); break; }
            }
            if ((rc = run_pipeline(c, nb, max_n, c->d_screen_theta, 0, false, 3, c->d_nlml, nullptr, c->d_status_out))) break;
            he = hipMemcpyAsync(hn + e0, c->d_nlml, sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipMemcpyAsync(hs + e0, c->d_status_out, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, c->stream);
        } else {
            const int lane = (int)(ci & 1);
            hipStream_t st = lane ? c->s_screen1 : c->stream;
            BatchPlan &P = lane_plan[lane];
            en.resize(nb);
            for (int i = 0; i < nb; i++) en[i] = c->h_n[cs[i]];
            layout_plan(c, en.data(), nb, false, P);
            P.row0 = lane * lane_rows;
            P.mat0 = lane * cap_mat; P.vec0 = lane * cap_vec; P.tab0 = lane * cap_tab; P.la_part0 = lane * cap_part; P.la_small0 = lane * cap_small;
            int *hb = (int *)c->h_screen_tab + 3 * e0, *hpz = hb + nb, *ht = hpz + nb;
            for (int i = 0; i < nb; i++) { hb[i] = cs[P.order[i]]; hpz[i] = P.order[i]; ht[i] = tp[P.order[i]]; }
            he = hipMemcpyAsync(c->d_bslot + P.row0, hb, sizeof(int) * nb, hipMemcpyHostToDevice, st);
            if (he == hipSuccess) he = hipMemcpyAsync(c->d_bpos + P.row0, hpz, sizeof(int) * nb, hipMemcpyHostToDevice, st);
            if (he == hipSuccess) he = hipMemcpyAsync(c->d_tpos + P.row0, ht, sizeof(int) * nb, hipMemcpyHostToDevice, st);
            if (he != hipSuccess) { rc = fail(c, MEDGP_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(he)); break; }
            if ((rc = run_pipeline(c, nb, 0, c->d_screen_theta, 0, false, 3, c->d_nlml + P.row0, nullptr, c->d_status_out + P.row0, false, false,
                                   &P, st, lane * (kAuxStreams / 2), kAuxStreams / 2, lane ? c->ev_fork1 : c->ev_fork))) break;
            he = hipMemcpyAsync(hn + e0, c->d_nlml + P.row0, sizeof(double) * nb, hipMemcpyDeviceToHost, st);
            if (he == hipSuccess) he = hipMemcpyAsync(hs + e0, c->d_status_out + P.row0, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, st);
        }
        if (he != hipSuccess) { rc = fail(c, MEDGP_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(he)); break; }
    }
    c->tpos_on = false;
    c->last_nbatch = 0;   // (the cached plan carries this call's theta rows: the next call lays its own out)
    if (two) {
        c->plan = lane_plan[0];   // (diagnostics: medgp_last_plan reports lane 0's last chunk)
        c->last_has_inverse = false;
        HIPCHK(c, hipStreamSynchronize(c->s_screen1));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_retired(c);
    if (rc) return rc;
    for (int sp = 0; sp < nslots; sp++) {   // walk order -> the caller's rows
        std::memcpy(nlml + (size_t)perm[sp] * ninit, hn + (size_t)sp * ninit, sizeof(double) * ninit);
        if (status) std::memcpy(status + (size_t)perm[sp] * ninit, hs + (size_t)sp * ninit, sizeof(int32_t) * ninit);
    }
    return MEDGP_OK;
}

void *medgp_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void medgp_host_free(void *p) { if (p) (void)hipHostFree(p); }

int medgp_nlml_grad_async(medgp_ctx *c, int lane, int nbatch, const int32_t *slots, const double *theta, int flag_grad,
                          double *nlml, double *grad, int32_t *status) {
    if (!c) return MEDGP_ERR_ARG;
    if (lane < 0 || lane > 1) return fail(c, MEDGP_ERR_ARG, "lane %d outside {0, 1}", lane);
    if (!slots || !theta || !nlml) return fail(c, MEDGP_ERR_ARG, "NULL argument");
    if ((flag_grad & MEDGP_FLAG_GRAD) && !grad) return fail(c, MEDGP_ERR_ARG, "grad is NULL with flag_grad set");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    if (nbatch < 1 || nbatch > c->max_batch) return fail(c, MEDGP_ERR_CAPACITY, "nbatch %d outside [1, %d]", nbatch, c->max_batch);
    if (c->lane_pending[lane]) return fail(c, MEDGP_ERR_ARG, "lane %d still holds a call: medgp_wait it first", lane);
    HIPCHK(c, hipSetDevice(c->device));
    const size_t H = c->H;
    double *dth = lane ? c->d_theta1 : c->d_theta, *dnl = lane ? c->d_nlml1 : c->d_nlml, *dgr = lane ? c->d_grad1 : c->d_grad;
    int *dst = lane ? c->d_status1 : c->d_status_out;
    if (!c->s_up) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->s_down, hipStreamNonBlocking));
    }
    for (hipEvent_t *e : {&c->ev_up[lane], &c->ev_k[lane], &c->ev_lane[lane]})
        if (!*e) HIPCHK(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    // theta: on the upload stream (the lane's staging is free: the caller has waited for the lane's previous call)
    HIPCHK(c, hipMemcpyAsync(dth, theta, sizeof(double) * nbatch * H, hipMemcpyHostToDevice, c->s_up));
    HIPCHK(c, hipEventRecord(c->ev_up[lane], c->s_up));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_up[lane], 0));
    int rc = medgp_nlml_grad_device(c, nbatch, slots, dth, flag_grad, dnl, dgr, dst);
    if (rc) return rc;
    // results: on the download stream, behind this call's kernels
    HIPCHK(c, hipEventRecord(c->ev_k[lane], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->s_down, c->ev_k[lane], 0));
    HIPCHK(c, hipMemcpyAsync(nlml, dnl, sizeof(double) * nbatch, hipMemcpyDeviceToHost, c->s_down));
    if (flag_grad & MEDGP_FLAG_GRAD) HIPCHK(c, hipMemcpyAsync(grad, dgr, sizeof(double) * nbatch * H, hipMemcpyDeviceToHost, c->s_down));
    if (status) HIPCHK(c, hipMemcpyAsync(status, dst, sizeof(int32_t) * nbatch, hipMemcpyDeviceToHost, c->s_down));
    HIPCHK(c, hipEventRecord(c->ev_lane[lane], c->s_down));
    c->lane_pending[lane] = true;
    return MEDGP_OK;
}

int medgp_wait(medgp_ctx *c, int lane) {
    if (!c) return MEDGP_ERR_ARG;
    if (lane < 0 || lane > 1) return fail(c, MEDGP_ERR_ARG, "lane %d outside {0, 1}", lane);
    if (!c->lane_pending[lane]) return MEDGP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev_lane[lane]));
    c->lane_pending[lane] = false;
    return MEDGP_OK;
}

int medgp_get_factor(medgp_ctx *c, int b, float *alpha, float *linv, float *beta) {
    if (!c) return MEDGP_ERR_ARG;
    if (b < 0 || b >= c->last_nbatch) return fail(c, MEDGP_ERR_ARG, "batch entry %d outside the last call's [0, %d)", b, c->last_nbatch);
    if (!c->last_has_inverse)
        return fail(c, MEDGP_ERR_ARG, "no factor available: the last call formed neither gradients nor the factor "
                                      "(pass MEDGP_FLAG_GRAD or MEDGP_FLAG_KEEP_FACTOR to medgp_nlml_grad)");
    HIPCHK(c, hipSetDevice(c->device));
    const int S = c->max_slots;
    const MedgpDev E = entry_view(c, b);   // the entry's rows of the batch buffers (its size class's view, shifted)
    const int ld = E.ldn;
    int eff = c->h_bslot[b];
    const int slot = eff >= S ? eff - S : eff, n = c->h_n[slot];
    const std::vector<int> &perm = c->h_perm[slot];
    int st = 0;
    HIPCHK(c, hipMemcpyAsync(&st, E.status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (st < 0) return fail(c, MEDGP_ERR_ARG, "batch entry %d failed (status %d); no factor available", b, st);
    if (linv && eff < S && !c->h_perm_identity[slot]) {
        // The entry was factored in the grouped order (the gradient needs it), but L^-1 is order dependent: re-factor this
        // one entry from the caller-order copy of the patient (same hyper block; tables, Gram, factor and inverse redone).
        const int shadow = slot + S;
        HIPCHK(c, hipMemcpyAsync(c->d_one_slot, &shadow, sizeof(int), hipMemcpyHostToDevice, c->stream));
        MedgpDev V = E;
        V.bslot = c->d_one_slot;
        V.bpos = nullptr;
        const bool prof = c->profiling;
        c->profiling = false;   // not part of the evaluation being measured
        // (one entry: the route rule of run_pipeline for a uniform call of one)
        const int nb1 = blocks64(n);
        const bool la1 = !c->use_v0 && !c->pin_route && nb1 >= 2 && (c->force_mc > 0 || (c->force_mc == 0 && nb1 >= 3));
        std::vector<LaNeed> las;
        if (la1) { las.push_back({1, nb1, ld, LaArgs{}}); int rcl = ensure_la(c, las, true); if (rcl) return rcl; }
        const int route1 = la1 ? ROUTE_LA : (c->pin_route ? ROUTE_WG84 : (c->cholinv_nw ? (c->cholinv_nw == 44 ? ROUTE_WG44 : ROUTE_WG84) : (nb1 <= 4 ? ROUTE_WG44 : ROUTE_WG84)));
        int rc = run_pipeline_one(c, c->stream, V, 1, nb1, nullptr, 0, true, 1, nullptr, nullptr, nullptr, false, &n, route1, la1 ? &las[0].A : nullptr);
        c->profiling = prof;
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync((int *)E.bslot, c->d_one_slot, sizeof(int), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(&st, E.status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->h_bslot[b] = eff = shadow;
        if (st < 0) return fail(c, MEDGP_ERR_ARG, "batch entry %d: the factorisation in the caller's order failed (status %d)", b, st);
    }
    const bool caller_order = eff >= S || c->h_perm_identity[slot];
    if (alpha) {
        std::vector<double> ha(n);
        HIPCHK(c, hipMemcpy(ha.data(), E.alpha, sizeof(double) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) alpha[caller_order ? i : perm[i]] = (float)ha[i];
    }
    if (beta) {
        double sc[4];
        HIPCHK(c, hipMemcpy(sc, E.scal, sizeof(sc), hipMemcpyDeviceToHost));
        *beta = (float)sc[1];
    }
    if (linv) {
        const char *hp = nullptr;
        int rc2 = d2h_pinned(c, E.Linv, sizeof(double) * n * ld, &hp);
        if (rc2) return rc2;
        const double *hx = (const double *)hp;
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
        if ((rc = dalloc(c, &c->d_ks, (size_t)cap * c->ldn))) return rc;   // k* -> v = L^-1 k* work rows
        c->pred_cap = cap;
    }
    int max_n = 0, rc;
    // patients are used in the caller's order when that differs from the grouped one?  No: mean / var are permutation
    // invariant, the grouped copy serves.
    if ((rc = set_batch(c, nbatch, slots, &max_n, false, true))) return rc;   // (the diagonal blocks U_kk live in Linv)
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
    // factor + z = L^-1 y only (no inverse): k* rides along as one more right-hand side in k_predict
    if ((rc = run_pipeline(c, nbatch, max_n, c->d_theta, 0, false, 1, nullptr, nullptr, nullptr, true, true))) return rc;
    for (const SizeClass &k : c->plan.cls) {   // (behind the join of the classes' chains: one launch per class view)
        Launcher l(c, KID_PREDICT);
        hipLaunchKernelGGL(k_predict, dim3(nstar, k.count), dim3(256), 0, c->stream, class_view(c, c->plan, k), nstar, c->d_meta2, c->d_t2, c->d_ks, c->d_mean, c->d_var);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(mean, c->d_mean, sizeof(float) * ntot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(var, c->d_var, sizeof(float) * ntot, hipMemcpyDeviceToHost, c->stream));
    std::vector<int> st(nbatch, 0);
    if (status) HIPCHK(c, hipMemcpyAsync(st.data(), c->d_status, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (status) for (int i = 0; i < nbatch; i++) status[c->plan.order[i]] = st[i];   // internal order -> the caller's
    return MEDGP_OK;
}

int medgp_factor_batch(medgp_ctx *c, int nbatch, const int32_t *slots, const double *theta, double *const *Lout, double *const *zout,
                       int32_t *status) {
    if (!c) return MEDGP_ERR_ARG;
    if (!slots || !theta || nbatch < 1) return fail(c, MEDGP_ERR_ARG, "bad argument");
    if (c->max_slots == 0) return fail(c, MEDGP_ERR_CAPACITY, "call medgp_reserve first");
    HIPCHK(c, hipSetDevice(c->device));
    int max_n = 0, rc;
    if ((rc = set_batch(c, nbatch, slots, &max_n, true, false))) return rc;       // the CALLER's observation order; size classes; L and z only: no Linv
    HIPCHK(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * c->H * nbatch, hipMemcpyHostToDevice, c->stream));
    if ((rc = run_pipeline(c, nbatch, max_n, c->d_theta, 0, false, 1, nullptr, nullptr, nullptr, false, true))) return rc;
    std::vector<int> st(nbatch, 0);
    {
        std::vector<int> sti(nbatch, 0);
        HIPCHK(c, hipMemcpyAsync(sti.data(), c->d_status, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < nbatch; i++) st[c->plan.order[i]] = sti[i];   // internal order -> the caller's
    }
    if (status) for (int b = 0; b < nbatch; b++) status[b] = st[b];
    // every entry's rows of the batch buffers (its size class's view) and leading dimension
    std::vector<MedgpDev> ev(nbatch);
    for (int b = 0; b < nbatch; b++) ev[b] = entry_view(c, b);
    // export: the n x ld leading rows of every successful entry, through the pinned bounce buffer in groups of <= 256 MB
    size_t b0 = 0;
    while (Lout && b0 < (size_t)nbatch) {
        size_t b1 = b0, bytes = 0;
        std::vector<size_t> off;
        while (b1 < (size_t)nbatch) {
            const int n = c->h_n[slots[b1]];
            const size_t need = (st[b1] >= 0 && Lout[b1] && n > 0) ? sizeof(double) * n * ev[b1].ldn : 0;
            if (b1 > b0 && bytes + need > ((size_t)256 << 20)) break;
            off.push_back(bytes); bytes += need; b1++;
        }
        if (bytes > c->bounce_cap) {
            if (c->h_bounce) (void)hipHostFree(c->h_bounce);
            c->h_bounce = nullptr; c->bounce_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->h_bounce, bytes + bytes / 4, hipHostMallocDefault));
            c->bounce_cap = bytes + bytes / 4;
        }
        for (size_t b = b0; b < b1; b++) {
            const int n = c->h_n[slots[b]];
            if (st[b] >= 0 && Lout[b] && n > 0)
                HIPCHK(c, hipMemcpyAsync(c->h_bounce + off[b - b0], ev[b].Kmat, sizeof(double) * n * ev[b].ldn, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (size_t b = b0; b < b1; b++) {
            const int n = c->h_n[slots[b]];
            if (!(st[b] >= 0 && Lout[b] && n > 0)) continue;
            const double *hl = (const double *)(c->h_bounce + off[b - b0]);
            const size_t ld = ev[b].ldn;
            double *Lo = Lout[b];
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) Lo[(size_t)i * n + j] = (j <= i) ? hl[(size_t)i * ld + j] : 0.0;
        }
        b0 = b1;
    }
    if (zout) {
        for (int b = 0; b < nbatch; b++) {
            const int n = c->h_n[slots[b]];
            if (st[b] >= 0 && zout[b] && n > 0)
                HIPCHK(c, hipMemcpyAsync(zout[b], ev[b].z, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return MEDGP_OK;
}

int medgp_factor(medgp_ctx *c, int slot, const double *theta, double *Lout, double *zout, int32_t *status) {
    int32_t s1 = slot;
    double *Lp[1] = {Lout}, *zp[1] = {zout};
    return medgp_factor_batch(c, 1, &s1, theta, Lout ? Lp : nullptr, zout ? zp : nullptr, status);
}

int medgp_pin_route(medgp_ctx *c, int pinned) {
    if (!c) return MEDGP_ERR_ARG;
    c->pin_route = pinned ? 1 : 0;
    return MEDGP_OK;
}

int medgp_last_plan(const medgp_ctx *c, int max_classes, int32_t *count, int32_t *blocks, int32_t *route) {
    if (!c) return MEDGP_ERR_ARG;
    const int nc = (int)c->plan.cls.size();
    for (int i = 0; i < nc && i < max_classes; i++) {
        if (count) count[i] = c->plan.cls[i].count;
        if (blocks) blocks[i] = c->plan.cls[i].nbmax;
        if (route) route[i] = c->plan.cls[i].route;
    }
    return nc;
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
#ifdef MEDGP_DSTAMPS
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
int medgp_debug_clear_slab(medgp_ctx *c, int b, int nbytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemset(c->dev.slab + (size_t)b * c->dev.slab_stride, 0, nbytes));
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
    if (enable) {   // warm the event pool outside any timed region (20 steps x 7 launches x 2 events of a default bench run)
        HIPCHK(c, hipSetDevice(c->device));
        while (c->ev_pool.size() < 512) { hipEvent_t e = nullptr; HIPCHK(c, hipEventCreate(&e)); c->ev_pool.push_back(e); }
    }
    c->profiling = enable != 0;
    c->profile_only = (enable >= 2 && enable - 2 < KID_COUNT) ? enable - 2 : -1;
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

// ---- cohort statistics (no context: a one-shot call on `device`) ----------------------------------------------------------
int medgp_kde_mode_at(int device, int nseries, const int64_t *off, const int32_t *cnt, const double *data, const int64_t *toff,
                      const int32_t *tcnt, const double *test, int weighted, double *mode, double *bw, int32_t *status, double *kernel_ms);
int medgp_kde_mode(int device, int nseries, const int64_t *off, const int32_t *cnt, const double *data, int weighted,
                   double *mode, double *bw, int32_t *status, double *kernel_ms) {
    return medgp_kde_mode_at(device, nseries, off, cnt, data, nullptr, nullptr, nullptr, weighted, mode, bw, status, kernel_ms);
}
int medgp_kde_mode_at(int device, int nseries, const int64_t *off, const int32_t *cnt, const double *data, const int64_t *toff,
                      const int32_t *tcnt, const double *test, int weighted, double *mode, double *bw, int32_t *status, double *kernel_ms) {
    medgp_ctx *none = nullptr;
    if ((tcnt != nullptr) != (toff != nullptr) || (tcnt != nullptr) != (test != nullptr)) return fail(none, MEDGP_ERR_ARG, "medgp_kde_mode_at: toff, tcnt and test go together");
    if (nseries < 0 || (nseries > 0 && (!off || !cnt || !data || !mode || !status))) return fail(none, MEDGP_ERR_ARG, "medgp_kde_mode: bad argument");
    if (nseries == 0) return MEDGP_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(none, MEDGP_ERR_NODEVICE, "medgp_kde_mode: no GPU (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(none, MEDGP_ERR_ARG, "medgp_kde_mode: device %d of %d", device, ndev);
    int64_t total = 0;
    for (int s = 0; s < nseries; s++) {
        if (cnt[s] < 0 || off[s] < 0) return fail(none, MEDGP_ERR_ARG, "medgp_kde_mode: series %d has a negative size / offset", s);
        if (off[s] + cnt[s] > total) total = off[s] + cnt[s];
    }
    int64_t ttotal = 0;
    int maxnt = 0;
    for (int s = 0; tcnt && s < nseries; s++) {
        if (tcnt[s] < 0 || toff[s] < 0) return fail(none, MEDGP_ERR_ARG, "medgp_kde_mode_at: series %d has a negative grid size / offset", s);
        if (toff[s] + tcnt[s] > ttotal) ttotal = toff[s] + tcnt[s];
        maxnt = tcnt[s] > maxnt ? tcnt[s] : maxnt;
    }
    HIPCHK(none, hipSetDevice(device));
    long long *d_toff = nullptr; int *d_tcnt = nullptr; double *d_test = nullptr;
    long long *d_off = nullptr; int *d_cnt = nullptr, *d_st = nullptr; double *d_x = nullptr, *d_mode = nullptr, *d_bw = nullptr, *d_stats = nullptr, *d_part = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MEDGP_OK;
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == MEDGP_OK) rc = fail(none, MEDGP_ERR_HIP, "medgp_kde_mode: %s failed: %s", what, hipGetErrorString(e)); return e == hipSuccess; };
    static_assert(sizeof(long long) == sizeof(int64_t), "offsets are passed through as 64-bit");
    int maxn = 0;
    for (int s = 0; s < nseries; s++) maxn = cnt[s] > maxn ? cnt[s] : maxn;
    const int rankchunks = maxn > 0 ? (maxn + KDE_THREADS - 1) / KDE_THREADS : 1;
    const int maxe = maxn > maxnt ? maxn : maxnt;
    const int maxchunks = maxe > 0 ? (maxe + KDE_THREADS - 1) / KDE_THREADS : 1;
    if (chk(hipMalloc(&d_off, sizeof(long long) * nseries), "hipMalloc") && chk(hipMalloc(&d_cnt, sizeof(int) * nseries), "hipMalloc") &&
        chk(hipMalloc(&d_st, sizeof(int) * nseries), "hipMalloc") && chk(hipMalloc(&d_x, sizeof(double) * (total > 0 ? total : 1)), "hipMalloc") &&
        chk(hipMalloc(&d_mode, sizeof(double) * nseries), "hipMalloc") && chk(hipMalloc(&d_bw, sizeof(double) * nseries), "hipMalloc") &&
        chk(hipMalloc(&d_stats, sizeof(double) * KDE_NSTAT * nseries), "hipMalloc") &&
        chk(hipMalloc(&d_part, sizeof(double) * 4 * (size_t)nseries * maxchunks), "hipMalloc") &&
        chk(hipEventCreate(&e0), "hipEventCreate") && chk(hipEventCreate(&e1), "hipEventCreate") &&
        chk(hipMemcpy(d_off, off, sizeof(long long) * nseries, hipMemcpyHostToDevice), "hipMemcpy") &&
        chk(hipMemcpy(d_cnt, cnt, sizeof(int) * nseries, hipMemcpyHostToDevice), "hipMemcpy") &&
        chk(hipMemcpy(d_x, data, sizeof(double) * total, hipMemcpyHostToDevice), "hipMemcpy") &&
        chk(hipMemset(d_stats, 0, sizeof(double) * KDE_NSTAT * nseries), "hipMemset") &&
        (!tcnt || (chk(hipMalloc(&d_toff, sizeof(long long) * nseries), "hipMalloc") && chk(hipMalloc(&d_tcnt, sizeof(int) * nseries), "hipMalloc") &&
                   chk(hipMalloc(&d_test, sizeof(double) * (ttotal > 0 ? ttotal : 1)), "hipMalloc") &&
                   chk(hipMemcpy(d_toff, toff, sizeof(long long) * nseries, hipMemcpyHostToDevice), "hipMemcpy") &&
                   chk(hipMemcpy(d_tcnt, tcnt, sizeof(int) * nseries, hipMemcpyHostToDevice), "hipMemcpy") &&
                   chk(hipMemcpy(d_test, test, sizeof(double) * ttotal, hipMemcpyHostToDevice), "hipMemcpy")))) {
        chk(hipEventRecord(e0, nullptr), "hipEventRecord");
        hipLaunchKernelGGL(k_kde_stats, dim3(nseries), dim3(KDE_THREADS), 0, nullptr, nseries, d_off, d_cnt, d_x, d_stats);
        hipLaunchKernelGGL(k_kde_rank, dim3(nseries, rankchunks), dim3(KDE_THREADS), 0, nullptr, d_off, d_cnt, d_x, d_stats);
        hipLaunchKernelGGL(k_kde_dens, dim3(nseries, maxchunks), dim3(KDE_THREADS), 0, nullptr, d_off, d_cnt, d_x, d_toff, d_tcnt, d_test, d_stats,
                           d_part, maxchunks);
        hipLaunchKernelGGL(k_kde_final, dim3((nseries + 255) / 256), dim3(256), 0, nullptr, nseries, d_off, d_cnt, d_x, d_toff, d_tcnt, d_test,
                           d_stats, d_part, maxchunks, weighted ? 1 : 0, d_mode, d_bw, d_st);
        chk(hipGetLastError(), "kde kernel launch");
        chk(hipEventRecord(e1, nullptr), "hipEventRecord");
        chk(hipMemcpy(mode, d_mode, sizeof(double) * nseries, hipMemcpyDeviceToHost), "hipMemcpy");
        chk(hipMemcpy(status, d_st, sizeof(int) * nseries, hipMemcpyDeviceToHost), "hipMemcpy");
        if (bw) chk(hipMemcpy(bw, d_bw, sizeof(double) * nseries, hipMemcpyDeviceToHost), "hipMemcpy");
        if (kernel_ms && rc == MEDGP_OK) { float ms = 0.f; chk(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime"); *kernel_ms = ms; }
        // a non-finite grid point makes the reference's arg-max / weighted mean NaN (np.argmax returns the NaN's index): report
        // the series as failed instead of silently skipping the point
        for (int s = 0; tcnt && rc == MEDGP_OK && s < nseries; s++)
            for (int i = 0; i < tcnt[s]; i++)
                if (!std::isfinite(test[toff[s] + i])) { status[s] = -1; mode[s] = std::nan(""); break; }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (void *q : {(void *)d_off, (void *)d_cnt, (void *)d_st, (void *)d_x, (void *)d_mode, (void *)d_bw, (void *)d_stats, (void *)d_part, (void *)d_toff, (void *)d_tcnt, (void *)d_test}) (void)hipFree(q);
    return rc;
}

}  // extern "C"
