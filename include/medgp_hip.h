/*
 * medgp_hip.h -- C ABI of the MI355X-native MedGP hot path (libmedgp_hip.so).
 *
 * This is the drop-in boundary for the per-patient negative-log-marginal-likelihood + gradient
 * operator of bee-hive/MedGP.  Every entry point cites the reference interface it replaces
 * ("ref:" = /root/reference/medgpc/src/...).  Plain pointers and sizes only; no C++ or torch types;
 * never throws, never exit()s; 0 = success, negative = error (message via medgp_last_error).
 *
 * Threading: one medgp_ctx is used by one host thread at a time (the reference's callers are
 * single-threaded, ref: inference/c_inference_exact.cpp:55-57); different contexts (e.g. one per GPU)
 * are fully independent and may run concurrently.
 *
 * Hyper-parameter vector theta (doubles, the reference's optimiser variables, ref:
 * core/c_hyperparam.cpp:99-122, kernel/c_kernel_LMC_SM.cpp:51-70):
 *   kernel_index 7 (LMC-SM): [log sigma_d (D) | A_q[d][r] raw, q-major (Q*D*R) | log mu_q (Q) | log v_q (Q) | log kappa_q[d] (Q*D)]
 *   kernel_index 8 (SM)    : [log sigma | log w_q (Q) | log mu_q (Q) | log v_q (Q)]      (D = 1)
 *   kernel_index 0 (SE)    : [log sigma | log l | log sf]                                   (Q = D = 1)
 * Gradients come back in the same order.
 */
#ifndef MEDGP_HIP_H
#define MEDGP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct medgp_ctx medgp_ctx;

/* kernel_index values: ref main_one_train.cpp:85-93 */
#define MEDGP_KERNEL_SE      0
#define MEDGP_KERNEL_LMC_SM  7
#define MEDGP_KERNEL_SM      8

/* error codes */
#define MEDGP_OK            0
#define MEDGP_ERR_ARG      -1   /* bad argument                                   */
#define MEDGP_ERR_HIP      -2   /* a HIP runtime call failed                      */
#define MEDGP_ERR_NODEVICE -3   /* no usable GPU: the product has NO CPU fallback */
#define MEDGP_ERR_CAPACITY -4   /* slot / batch / n beyond medgp_reserve          */

/* prior type codes, ref: prior/c_prior.h:50-53 */
#define MEDGP_PRIOR_NONE    -1
#define MEDGP_PRIOR_CLAMP    0
#define MEDGP_PRIOR_NORMAL   1
#define MEDGP_PRIOR_LAPLACE  2

/* flag_grad bits of medgp_nlml_grad / medgp_nlml_grad_device.  The reference passes a bool (`flag_grad`,
 * ref: inference/c_inference.h:38-52); 0 and 1 keep that meaning.  Bit 1 asks for the factor outputs of the same call
 * (chol_alpha / chol_factor_inv / beta, ref: inference/c_inference_exact.cpp:124-147) to be formed even when no gradient
 * is wanted -- GP_Regression::train(false) followed by GP_Regression::predict, ref: core/gp_regression.cpp:102-214,
 * caller main_one_test.cpp:386-399 -- so that medgp_get_factor is valid afterwards. */
#define MEDGP_FLAG_GRAD        1
#define MEDGP_FLAG_KEEP_FACTOR 2

/* ABI version, bumped on any signature change */
int medgp_abi_version(void);   /* 3: medgp_reserve_plan, medgp_alloc_stats (round 6) */

/* number of visible HIP devices (0 if none; never initialises a context) */
int medgp_device_count(void);

/* Create a context bound to one device and one covariance family.
 * Replaces the construction of c_kernel_* / c_likelihood_* / c_inference_* objects,
 * ref: main_one_train.cpp:103-152 (run_model_LMC_SM / run_model_SE / run_model_SM).
 * R is ignored for SE/SM. */
int  medgp_create(medgp_ctx **out, int device, int kernel_index, int Q, int D, int R);
void medgp_destroy(medgp_ctx *ctx);
const char *medgp_last_error(const medgp_ctx *ctx);   /* ctx may be NULL: last create() error */

/* total number of hypers H = lik + cov (ref: c_kernel_LMC_SM.cpp:64-70; gaussianMO: D) */
int medgp_num_hyp(const medgp_ctx *ctx);

/* pi used inside the kernels; default 3.14159265, the reference's literal
 * (ref: util/global_settings.h:6) */
int medgp_set_pi(medgp_ctx *ctx, double pi);

/* Run all work of this context on the given hipStream_t (e.g. torch's current stream).
 * NULL = the context's own stream (default). */
int medgp_set_stream(medgp_ctx *ctx, void *hip_stream);

/* (Re)allocate device storage: patient slots, largest padded n, largest batch per call.
 * Replaces the per-evaluation new[]/delete[] of N*N buffers, ref: core/gp_regression.cpp:102-117,
 * inference/c_inference_exact.cpp:66-68,168. Existing patients are discarded.
 * The per-entry matrices (two padded n x n fp64 matrices per batch entry) are allocated here when max_batch x max_n^2 of them stay
 * below 8 GB; beyond that (a cohort whose largest patient is far above the rest) they are sized once by medgp_reserve_plan from the
 * patients' sizes, or grown by the calls to what their size classes need (see medgp_last_plan) -- a call that outgrows a buffer replaces
 * it without waiting for anything (the old block is released at the context's next idle point) and drops the factors of earlier calls.  A call whose matrices exceed the memory budget (64 GB, at most 70 % of what the device has free;
 * MEDGP_MEM_BUDGET_GB) is run as consecutive waves of size classes that reuse the arenas; calls whose OUTPUTS need every entry's matrix
 * afterwards (MEDGP_FLAG_KEEP_FACTOR, medgp_factor_batch, medgp_fit_predict_batch) fail with MEDGP_ERR_CAPACITY instead. */
int medgp_reserve(medgp_ctx *ctx, int max_slots, int max_n, int max_batch);

/* Announce the sizes of the patients that will be resident together (n[count] observation counts, any order) and the width of the
 * random-initialisation screening (ninit hyper vectors per patient, 0 = none): the per-entry arenas are mapped ONCE to the high-water
 * mark of (a) one nlml + gradient call over them and (b) the chunks medgp_screen forms of them, so that no later call has to obtain
 * device memory.  Optional -- without it the buffers grow with the calls -- but on this platform obtaining memory that an earlier
 * process has used can take seconds (the driver wipes it first; 0.27 s per GB was measured), and a long-lived trainer wants that wait
 * once, before its loop, not inside it, and for as few bytes as its calls really need.  A cohort host has the sizes before its first call: the reference's
 * job generator reads them to bucket patients by N (ref: medgpc/util/run_exp_generator.py:213-260, scripts/slurm_della.json:6-62),
 * medgp_train takes them from its patient list.  Replaces nothing the reference allocates ahead: it news / deletes its N x N buffers
 * per evaluation (ref: core/gp_regression.cpp:102-117, inference/c_inference_exact.cpp:66-68,168). */
int medgp_reserve_plan(medgp_ctx *ctx, int count, const int32_t *n, int ninit);

/* Memory-management accounting of the context: wall seconds spent obtaining / releasing device memory so far, the number of such
 * calls, and the bytes the per-entry buffers (arenas) hold at present.  Any pointer may be NULL.  (bench.py and medgp_train report it as
 * `alloc_s`: time the host spent waiting for memory instead of queueing work.) */
int medgp_alloc_stats(const medgp_ctx *ctx, double *seconds, int64_t *calls, int64_t *arena_bytes);

/* Upload one patient (meta[i] in [0,D), t = time stamps, y = z-scored values; host pointers, copied).
 * Replaces c_objective_one's constructor, ref: util/c_objective_one.cpp:23-36 and
 * util/c_objective_one.h:40-45.  meta may be NULL for SE/SM.  Observations are stably grouped
 * by output internally (the reference's loader already produces that order,
 * ref: dataio/c_experiment.cpp:272-308), which leaves nlml/gradients unchanged. */
int medgp_set_patient(medgp_ctx *ctx, int slot, int n, const int32_t *meta, const float *t, const float *y);

/* Packed upload of nslots patients in ONE host-to-device transfer and without waiting for the device: patient k goes to
 * slots[k] and owns elements [offsets[k], offsets[k+1]) of meta / t / y (offsets has nslots + 1 entries).
 * Replaces a loop of c_objective_one constructions over a cohort, ref: util/c_objective_one.cpp:23-36, the stacked
 * per-feature loader arrays of dataio/c_experiment.cpp:254-309, and the per-(time stamp, observation) training subsets
 * of the imputation loop, ref: main_one_test.cpp:352-365. */
int medgp_set_patients(medgp_ctx *ctx, int nslots, const int32_t *slots, const int64_t *offsets, const int32_t *meta,
                       const float *t, const float *y);

/* Per-hyper prior descriptor of one slot, H entries each in theta order; flag == NULL removes the prior.
 * Replaces the public vectors of c_prior read by c_inference_prior::compute_nlml,
 * ref: prior/c_prior.h:35-53, inference/c_inference_prior.cpp:60-150.
 * slot = -1 applies the descriptor to every slot. */
int medgp_set_prior(medgp_ctx *ctx, int slot, const uint8_t *flag, const int32_t *type,
                    const uint8_t *is_exp, const float *p0, const float *p1);

/* The same for nslots slots in ONE host-to-device transfer and without waiting for the device: row k of the [nslots][H] arrays
 * describes slot slots[k] (flag == NULL removes the prior of all of them).  This is what a cohort trainer calls once per
 * variational-EM outer iteration for every patient whose psi changed -- the reference rebuilds the public vectors of one
 * c_prior object per patient there, ref: util/c_optimizer_varEM.cpp:98-162, prior/c_prior.cpp:222-279. */
int medgp_set_priors(medgp_ctx *ctx, int nslots, const int32_t *slots, const uint8_t *flag, const int32_t *type,
                     const uint8_t *is_exp, const float *p0, const float *p1);

/* THE OPERATOR.  nbatch independent evaluations: problem b = patient slots[b] with hypers
 * theta[b*H .. (b+1)*H).  Replaces c_objective_one::compute_objective -> GP_Regression::train ->
 * c_inference_prior::compute_nlml -> c_inference_exact::compute_nlml,
 * ref: util/c_objective_one.cpp:40-82, core/gp_regression.cpp:102-126,
 *      inference/c_inference_prior.cpp:25-154, inference/c_inference_exact.cpp:29-244,
 *      kernel/c_kernel_LMC_SM.cpp:152-327.
 * flag_grad: 0 = nlml only, MEDGP_FLAG_GRAD = nlml + gradient, | MEDGP_FLAG_KEEP_FACTOR = also keep alpha / L^-1.
 * status[b]: 0..10 = jitter rounds applied (ref: c_inference_exact.cpp:96-111); -1 = the
 * reference's `return false` (Cholesky failed after 10 jitters, or n <= 2,
 * ref: util/c_objective_one.cpp:51,79-81); nlml/grad of a failed problem are NaN.
 * All pointers are HOST memory; grad may be NULL when flag_grad == 0. */
int medgp_nlml_grad(medgp_ctx *ctx, int nbatch, const int32_t *slots, const double *theta, int flag_grad,
                    double *nlml, double *grad, int32_t *status);

/* Random-initialisation screening: HOT LOOP A of the reference (ref: main_one_train.cpp:228-253 -- every one of the
 * random_init_num hyper vectors evaluated on the patient, nlml only, the smallest wins).  The ninit vectors theta[ninit][H] are the same
 * for every patient (c_experiment::get_global_hyp draws them once from the seed, ref: dataio/c_experiment.cpp:418-441), so they
 * travel to the device ONCE per call and every (patient, vector) entry reads its row there; the nslots * ninit evaluations are
 * queued in calls of at most max_batch entries without a host wait in between.  nlml / status: [nslots][ninit], host memory
 * (status may be NULL).  Each evaluation is bit-identical to medgp_nlml_grad(flag_grad = 0) on the same (patient, vector) in a batch
 * of the same composition. */
int medgp_screen(medgp_ctx *ctx, int nslots, const int32_t *slots, int ninit, const double *theta, double *nlml, int32_t *status);

/* Same operator with theta / nlml / grad / status in DEVICE memory of ctx's device, queued on the context's stream
 * (slots stays a host array: it only selects resident patients; it is copied before the call returns).  Asynchronous for
 * EVERY route: the one-workgroup-per-patient kernel runs the reference's jitter loop (ref: inference/c_inference_exact.cpp:
 * 99-111) in-kernel, and the multi-CU schedule used for few, large entries (at most 0.6 x #CU entries with n > 64) hands the
 * entries whose single attempt failed to that same in-kernel loop on the device (k_cholinv, sel = 2) -- the host reads no
 * status back, and a call that has to grow a buffer waits for nothing either (round 6: the outgrown block is released at the
 * context's next idle point). */
int medgp_nlml_grad_device(medgp_ctx *ctx, int nbatch, const int32_t *slots, const double *theta_dev,
                           int flag_grad, double *nlml_dev, double *grad_dev, int32_t *status_dev);

/* Asynchronous form of medgp_nlml_grad for hosts that overlap their own work (the optimiser state machines of the patients of
 * one half of a cohort, ref: util/c_optimizer_scg.cpp:87-281) with the device working on the other half: the call queues the
 * theta upload, the evaluation and the result downloads of `lane` (0 or 1; each lane has its own device staging) on the context's
 * stream and returns; medgp_wait(ctx, lane) blocks until that lane's results are in nlml / grad / status.  The host arrays must
 * stay valid (and untouched) until then, and overlap needs them in pinned memory: medgp_host_alloc / medgp_host_free
 * (hipHostMalloc; pageable memory works but makes the copies synchronous).  A lane holds one call at a time. */
void *medgp_host_alloc(size_t bytes);
void  medgp_host_free(void *p);
int   medgp_nlml_grad_async(medgp_ctx *ctx, int lane, int nbatch, const int32_t *slots, const double *theta, int flag_grad,
                            double *nlml, double *grad, int32_t *status);
int   medgp_wait(medgp_ctx *ctx, int lane);

/* After medgp_nlml_grad*(…, flag_grad with MEDGP_FLAG_GRAD or MEDGP_FLAG_KEEP_FACTOR): copy out K^-1 (y - m), L^-1 and
 * beta = (y - m)^T K^-1 (y - m) of batch entry b as the reference's float buffers (alpha[n]; linv[n*n] row-major lower,
 * strict upper zero), ALWAYS in the caller's original observation order -- L^-1 is the inverse Cholesky factor of the
 * Gram matrix in that order, exactly what GP_Regression::predict multiplies the caller-order cross Gram with
 * (ref: core/gp_regression.cpp:181-196).  If the entry was evaluated in the internal grouped order (gradient calls on
 * patients not uploaded grouped by output), asking for linv re-factors that one entry in the caller's order.
 * Fails (MEDGP_ERR_ARG) when the last call formed no factor (flag_grad == 0, or medgp_fit_predict*).
 * Replaces the chol_alpha / chol_factor_inv / beta outputs of c_inference::compute_nlml,
 * ref: inference/c_inference.h:38-52, inference/c_inference_exact.cpp:124-147. Any pointer may be NULL. */
int medgp_get_factor(medgp_ctx *ctx, int b, float *alpha, float *linv, float *beta);

/* Factor once with theta, predict nstar points.  Replaces GP_Regression::train(false) + predict,
 * ref: core/gp_regression.cpp:128-214, kernel/c_kernel_LMC_SM.cpp:329-372 (cross Gram), :122-150 (diag);
 * caller: main_one_test.cpp:386-399.  mean/var are float like the reference. */
int medgp_fit_predict(medgp_ctx *ctx, int slot, const double *theta, int nstar, const int32_t *meta2,
                      const float *t2, float *mean, float *var, int32_t *status);

/* Cholesky factor of one patient's Gram matrix and z = L^-1 (y - m), fp64, in the CALLER's observation order:
 * L [n*n] row-major lower (strict upper zero), z [n]; either may be NULL.  This is the state LAPACKE_spotrf + the first
 * half of spotrs leave inside c_inference_exact::compute_nlml (ref: inference/c_inference_exact.cpp:96-125) before it is
 * turned into chol_alpha / chol_factor_inv.  It is what the online imputation loop can SHARE between its problems: with the
 * observations in time order every training subset `past(t)` of main_one_test.cpp:287-300 is a leading block of one
 * factorisation (L[0:p, 0:p] is the factor of the first p observations, z[0:p] their solve), and the same-time observations
 * appended at :358-365 are the next rows -- medgp_test's no-update pass does one medgp_factor per patient instead of one
 * factorisation per imputed observation.  status as medgp_nlml_grad (no n > 2 guard, like GP_Regression::train). */
int medgp_factor(medgp_ctx *ctx, int slot, const double *theta, double *L, double *z, int32_t *status);
/* The same for nbatch patients in ONE call (one pipeline run, one read-back): entry b uses patient slots[b] and hypers
 * theta[b*H ..); L[b] receives n_b*n_b doubles, z[b] n_b doubles (either array, or single entries, may be NULL); status[nbatch].
 * What the cohort form of the no-update imputation pass uses (medgp_test --pan-list: one shared factorisation per patient, all
 * patients of a chunk in one call; ref: main_one_test.cpp:287-300, :358-365, :386-399). */
int medgp_factor_batch(medgp_ctx *ctx, int nbatch, const int32_t *slots, const double *theta, double *const *L, double *const *z,
                       int32_t *status);

/* nbatch independent (train(false) + predict ONE point) problems in one call: problem b uses patient
 * slots[b], hypers theta[b*H..), test point (meta2[b], t2[b]).  This is the inner body of the online
 * imputation loop, ref: main_one_test.cpp:352-409 (N* = 1 there, :369-372), batched over the (time stamp,
 * observation) pairs of a patient, each uploaded as its own slot (a subset of the patient's observations). */
int medgp_fit_predict_batch(medgp_ctx *ctx, int nbatch, const int32_t *slots, const double *theta,
                            const int32_t *meta2, const float *t2, float *mean, float *var, int32_t *status);

/* Cohort statistics, the step after training: for each of nseries independent sample series
 * (series s = data[off[s] .. off[s] + cnt[s])) the Gaussian kernel density estimate with Silverman's bandwidth evaluated AT the
 * samples, and its "mode": weighted != 0: sum x dens / sum dens; weighted == 0: the sample with the largest density (first one).
 * Replaces compute_kde + compute_mode, ref: medgpc/clustering/mode_estimate.py:438-450 (statsmodels KDEUnivariate, kernel "gau",
 * bw "silverman"), as called per nugget / per cluster mu, v / per element of the aggregated B matrices by
 * output_mode_LMC_SM (ref: :277-279, :339-351, :410-413).  One-shot call on `device`, host arrays in and out, no context.
 * status[s]: 0 ok, -1 when the reference's fit would raise (n < 2, non-finite sample, zero bandwidth); mode[s] is NaN then.
 * bw (the bandwidths) and kernel_ms (HIP-event time of the kernel) may be NULL.  Errors: medgp_last_error(NULL). */
int medgp_kde_mode(int device, int nseries, const int64_t *off, const int32_t *cnt, const double *data, int weighted,
                   double *mode, double *bw, int32_t *status, double *kernel_ms);
/* The same with the density of series s evaluated on its own grid test[toff[s] .. toff[s] + tcnt[s]) (tcnt[s] == 0: at the
 * samples, as above); the mode is then taken over the grid points.  This is how output_mode_SE evaluates the length-scale
 * (ref: mode_estimate.py:54-57) and output_mode_SM the period / length-scale densities (ref: :170-184): 100001-point grids,
 * arg-max mode.  toff / tcnt / test are all NULL or all given. */
int medgp_kde_mode_at(int device, int nseries, const int64_t *off, const int32_t *cnt, const double *data, const int64_t *toff,
                      const int32_t *tcnt, const double *test, int weighted, double *mode, double *bw, int32_t *status,
                      double *kernel_ms);

/* Route pinning.  By default the library picks the factorisation schedule of a call from the batch it is given (one workgroup per
 * patient in two shapes, or the multi-CU look-ahead schedule for few large patients): fastest, and every schedule meets the parity
 * bar, but the LAST BITS of a patient's results can then depend on how many batch-mates it had.  pinned != 0: every entry of every
 * call is factored by the one 8-wave single-workgroup kernel, so a patient's results are bit-identical whatever the batch -- what a
 * caller needs whose outputs must not depend on how patients were grouped into calls (medgp_test: a cohort run writes the same
 * bytes as one run per patient).  The reference has no such choice: one patient per process, ref: main_one_test.cpp:45-481. */
int medgp_pin_route(medgp_ctx *ctx, int pinned);

/* How the last medgp_nlml_grad* call was scheduled (diagnostics; round 5).  A call's entries are cut into size classes by their
 * number of 64-observation blocks, (2^(j-1), 2^j], and every class is given its own launch geometry and factorisation route, the way
 * the reference's job generator gives patients resources by size (ref: scripts/slurm_della.json:6-62,
 * medgpc/util/run_exp_generator.py:213-260).  Writes, largest class first and for at most max_classes classes: count[i] entries,
 * blocks[i] = 64-blocks of its largest entry, route[i] = 0 / 1 one workgroup per entry in the 4- / 8-wave shape, 2 the multi-CU
 * look-ahead schedule.  Returns the number of classes of the call (possibly > max_classes), or an error code. */
int medgp_last_plan(const medgp_ctx *ctx, int max_classes, int32_t *count, int32_t *blocks, int32_t *route);

/* block until all work queued on the context's stream is complete */
int medgp_synchronize(medgp_ctx *ctx);

/* ---- measurement hooks (bench.py roofline leg; no effect on results) ------------------------- */
/* enable = 1: bracket every kernel launch with HIP events on the launch stream; enable = 2 + k: only the launches of kernel k
   (medgp_profile_kernel_name) -- two events per step instead of two per launch (bench.py times its headline region this way:
   the events of all seven launches cost 1.6 % of a 512-patient step); enable = 0: off */
int medgp_profile_enable(medgp_ctx *ctx, int enable);
/* number of distinct kernels the library launches, and their names */
int medgp_profile_num_kernels(void);
const char *medgp_profile_kernel_name(int k);
/* synchronises, then returns accumulated milliseconds and launch count of kernel k since the
 * last medgp_profile_reset */
int medgp_profile_read(medgp_ctx *ctx, int k, double *ms_total, int64_t *launches);
int medgp_profile_reset(medgp_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
