// medgp_dev.h -- device-side data layout shared by all kernels of libmedgp_hip.so.
//
// HBM layout (one context = one device, one covariance family):
//   patient slots   : t, y (fp64, widened once from the caller's float), meta (int32), all
//                     [max_slots][ldn]; observations stably grouped by output, seg[D+1] offsets.
//   prior descriptor: [max_slots][H] packed {type, flag, exp, p0, p1}.
//   per batch entry : hyp block (sigma^2, B_q, w_q, c_q), cos/sin tables [Q][ldn],
//                     two ldn x ldn fp64 matrices (K -> L -> W, and L^-1), vectors z / alpha [ldn],
//                     block sums S, SM, SV [Q][D][D], scalars, status.
// ldn = max_n rounded up to 64; matrices are row-major with leading dimension ldn.
#pragma once
#define MEDGP_EPI_PARTS 16   // k_epilogue splits the H hypers of an entry over up to this many workgroups
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MEDGP_TILE 64

#define MEDGP_MAX_D 256   // outputs per context (per-output offset tables are staged in LDS)

struct MedgpPrior {          // one hyper of one slot (16 bytes)
    float p0, p1;
    float lg2b;              // Laplace only: log(2 p1) AS THE REFERENCE EVALUATES IT -- `log(2*param[1])` with a float argument is
                             // std::log(float) (ref: prior/c_prior.cpp:404, found by running the reference's compiled c_prior, round 6):
                             // single precision, 5e-10 relative away from the fp64 value.  Filled on the host by the C library's logf
                             // (medgp_set_prior[s]), so the device adds the very float the reference's libm produced.
    int8_t type;             // -1 none, 0 clamp, 1 normal, 2 laplace  (ref: prior/c_prior.h:50-53)
    uint8_t flag, is_exp, pad;
};

struct MedgpDev {
    const int *tpos;         // [nbatch] row of `theta` the entry reads, or null = its caller row (bpos): medgp_screen evaluates ONE block of
                             // hyper vectors on many patients
    // family
    int kidx, Q, D, R, H, nlik;
    int ldn, max_slots, max_batch;   // ldn: leading dimension of THIS VIEW's batch buffers (a size class of the call: the class's largest n rounded up to 64)
    int pld;                 // row stride of the patient arrays pt / py / pmeta (the context's max_n rounded up to 64)
    int hyp_stride;          // doubles per batch entry in `hyp`
    double pi;
    int dbg_fail;            // test hook (MEDGP_DEBUG_FAIL_ATTEMPTS=k): the first k factorisation attempts of every problem are
                             // treated as failed, which drives the reference's jitter loop deterministically (0 in production)
    // patients
    const int *pn;           // [slot] n
    const double *pt, *py;   // [slot][ldn]
    const int *pmeta;        // [slot][ldn]
    const int *pseg;         // [slot][D+1]
    const int *proff;        // [slot][D+1] prefix of 16-row pieces per output (slab row slots)
    const int *pcoff;        // [slot][D+1] prefix of 64-column pieces per output (slab column slots)
    const MedgpPrior *prior; // [slot][H]
    const uint8_t *prior_on; // [slot]
    // batch
    const int *bslot;        // [nbatch]
    const int *bpos;         // [nbatch] position of the entry in the CALLER's batch (theta, nlml, gradient, status rows), or null = identity:
                             // a call's entries are ordered by size internally (size classes are contiguous, medgp_capi.hip: BatchPlan)
    double *hyp;             // [batch][hyp_stride]: sig2[D] | B[Q*D*D] | w[Q] | c[Q]
    double *cs, *sn;         // [batch][Q][ldn]
    double *Kmat, *Linv;     // [batch][ldn*ldn]
                             // INVARIANT for readers: only the LOWER triangle of Kmat (L) and the UPPER triangle of Linv (U = L^-T) are
                             // defined after a factorisation.  The other halves hold leftovers (diag16 stores whole rows of a 16 x 16
                             // tile, the look-ahead chain leaves stale X_k tiles): every consumer masks with column <= row (row <=
                             // column for U) -- medgp_factor[_batch], medgp_get_factor, k_predict, k_wgrad.  tests/test_parity2_gpu.py
                             // (test_exported_factors_have_exact_zero_triangles) holds the exports to it on both routes.
    double *z, *alpha;       // [batch][ldn]
    double *scal;            // [batch][4]: logdet, quad, -, -
    int *status;             // [batch]
    int *bn;                 // [batch] n of the entry's patient, written by k_prep (one load instead of the bslot -> pn chain)
    double *epi_lp;          // [batch][MEDGP_EPI_PARTS] prior log-density of each part of k_epilogue's hyper range
    int *epi_ticket;         // [batch] arrival counter of those parts (zero between launches: the last part resets it)
    int *jit;                // [batch] jitter rounds applied so far (extra noise additions in the assembly)
    double *xk;              // [batch][64*64] L_kk^-1 of the current panel (multi-CU factorisation)
    double *S, *SM, *SV;     // [batch][Q*D*D]
    double *slab;            // [batch][3][Q][slab_R][slab_C] piecewise block sums written by k_wgrad
    double *wdiag;           // [batch][ldn] diag(W)
    int slab_R, slab_C;
    size_t slab_stride;
};

__host__ __device__ inline int medgp_roundup(int x, int m) { return (x + m - 1) / m * m; }
// offsets inside the hyp block
__host__ __device__ inline int hyp_off_sig2(const MedgpDev &) { return 0; }
__host__ __device__ inline int hyp_off_B(const MedgpDev &d) { return d.D; }
__host__ __device__ inline int hyp_off_w(const MedgpDev &d) { return d.D + d.Q * d.D * d.D; }
__host__ __device__ inline int hyp_off_c(const MedgpDev &d) { return d.D + d.Q * d.D * d.D + d.Q; }
