// kernels_cholinv_la.h -- look-ahead multi-CU factorisation for FEW, LARGE patients (BASELINE configs 3 and 5: one
// patient with N = 2048 / 4096).  Replaces LAPACKE_spotrf + spotrs + strtri of the reference
// (ref: inference/c_inference_exact.cpp:96-143) with the same recurrence as k_cholinv (kernels_cholinv.h): the
// left-looking panel recurrence on T = [K; I; y^T], whose rows turn into L, U = L^-T and z^T = (L^-1 y)^T.
//
// Why a second schedule.  The first multi-CU version ran two launches per 64-wide step and gave the WHOLE history product of
// a 64-row block (64 x 64 x 64k flops) to one workgroup: nb + k workgroups on 256 CUs, and the diagonal block's product
// sat in front of the serial 64x64 factorisation on the critical path (N = 2048: 3.4 ms, 3 % of the fp64 peak).  Here
//   * ONE launch per step k (k_la_step) holds three kinds of workgroups that do not depend on each other inside the launch:
//     D  (1 per patient)   diagonal chain: L[C_k+1, C_k] = P X_k^T, the two newest rank-64 terms of the NEXT diagonal
//                          block, its 64x64 factorisation on four waves (diag_factor_wg) -> X_k+1.  The chain never waits
//                          for a deep product: everything older than two panels was summed ahead of time by L tasks.
//     F  (1 per row block) finish panel k: row_r[C_k] = P_r,k X_k^T, then build the pre-solve block of panel k+1:
//                          P_r,k+1 = init - sum(partials) - row_r[C_k-1] L[C_k+1,C_k-1]^T - row_r[C_k] L[C_k+1,C_k]^T.
//     H  (1 per patient)   diagonal head start of block k+2 for the NEXT launch's chain:  -K[k+2,k+2] + row[C_k] row[C_k]^T +
//                          row[C_k-1] row[C_k-1]^T, with row[C_k] re-derived from a copy of the pre-solve block (round 4: this used
//                          to be the tail of the F task of row block k+2, which made that task as long as the chain).
//     L  (look-ahead)      for panel k+2 and every row block: partial sums over history slices of LA_SLICE panels
//                          (columns <= C_k-1, all final before the launch) into a scratch slab -- split-K over as many
//                          workgroups as the chip has CUs, summed later in a fixed order (bitwise reproducible).
//   * the serial work per step is the diagonal chain only: about 40 + 168 MFMAs per wave + diag_factor_wg.
// Row blocks of a step: M_i (K rows, i > k), U_rho (inverse rows, rho <= k, only with want_mode bit 0), Y (the y^T row, a
// 64-row block of which row 0 is real: z comes out of the same recurrence as everything else).
// The reference's jitter loop (ref: c_inference_exact.cpp:99-108): this schedule makes ONE attempt; a failed pivot marks the
// problem (status -2), the later launches of the schedule skip it, and the launch that follows the schedule --
// k_cholinv<8,4>(sel = 2) -- re-assembles exactly those entries with one more noise addition and carries the retry loop on
// in-kernel.  No status is read back by the host: calls that take this schedule are asynchronous like all others.
#pragma once
#include "kernels_cholinv.h"

#define LA_KC 32          // history columns staged per barrier
#define LA_THREADS 256
#define LA_SLICE 4        // panels (64 columns each) per look-ahead slice: the shortest slice (scratch is dimensioned for it)
// Panels per slice of the partial sums PRODUCED at step k (consumed one step later).  It depends on the step alone -- not on the batch
// or on the batch-mates' sizes, so a patient's arithmetic is the same in any call.  Late steps of a long factorisation use longer
// slices: a single N = 4096 evaluation has 526-600 tasks per step from k = 37 on, a few more than the chip's 512 workgroup slots,
// and paid a second, nearly empty round of 24-us tasks per step.  Measured at N = 4096 (k_la_step, interleaved on one box): 4 panels
// throughout 2.20 ms; 5 from step 36 2.17; 5 from 36 + 6 from 52 2.11; 5 from 32 + 6 from 44 (this rule) 2.08; longer slices or
// earlier switches 2.09-2.12.  (N <= 2048 never reaches step 32: unchanged.)
__host__ __device__ inline int la_slice_len(int k) { return k < 32 ? LA_SLICE : (k < 44 ? LA_SLICE + 1 : LA_SLICE + 2); }
#define LA_S 66           // LDS row stride of the 64x64 operand tiles
#ifndef LA_F_LATE
#define LA_F_LATE 0       // 1: F tasks request X_k / the pre-solve copy after their partial sums instead of up front
#endif
#ifndef LA_D_EARLY
#define LA_D_EARLY 1      // 1: the chain's wave 0 factors the first 16 x 16 tile of the next diagonal block beside the other waves' rank-64 update (step (5))
#endif
#ifndef LA_H_ROLE
#define LA_H_ROLE 1       // 1: the diagonal head start of block k+2 is formed by a workgroup of its own (H) instead of by the F task of that row block
#endif
#ifndef LA_FOLD_MAX
#define LA_FOLD_MAX 8     // up to this many diagonal look-ahead slices the H workgroup adds them itself and no R workgroup is launched
#endif
// Diagonal look-ahead slices that exist for block k+2 at step k, and who adds them: few slices -> the H workgroup (the chain of the
// next launch then reads ONE head-start slab: N = 2048 k_la_step 0.700 -> 0.677 ms); many slices (steps > 33 of N = 4096) -> an R
// workgroup beside H, as rounds 3-4a did (left to H its chain of slab round trips made H the longest task there: 1.86 -> 1.89 ms).
// A rule of the step alone; the additions are the same either way (slices summed from zero in order, added once): same bits.
__host__ __device__ inline int la_nsd(int k) { const int sl = la_slice_len(k - 1); return (k >= 2) ? (k - 1 + sl - 1) / sl : 0; }
__host__ __device__ inline bool la_fold(int k) { return LA_H_ROLE && la_nsd(k) <= LA_FOLD_MAX; }
#define LA_NAUX(k) ((la_fold(k) ? 0 : 1) + LA_H_ROLE)   // single tasks of step k behind the F row blocks: R and / or H
#ifndef LA_UKK_BY_F
#define LA_UKK_BY_F 1     // 1: the diagonal blocks of U are stored by the F task of their row block one launch later, not by the chain
#endif
#ifndef LA_NSUM
#define LA_NSUM 3         // partial-sum slabs an F task requests per memory round trip
#endif

struct LaArgs {
    double *ybuf;         // [batch][64][ldn]   the Y row block (row 0 = y^T -> z^T)
    double *part;         // [batch][2][rows][maxslice][64*64]   look-ahead partial sums, indexed by panel parity
    double *xk2;          // [batch][2][64*64]  X_k = L_kk^-1, indexed by panel parity
    double *pnx;          // [batch][2][64*64]  copy of the pre-solve block P_k+1,k (row-major), indexed by panel parity: every
                          //                    workgroup of step k re-derives L[C_k+1,C_k] from it while D overwrites the
                          //                    in-place block with the solved values
    double *pnx2;         // [batch][2][64*64]  copy of the pre-solve block P_k+2,k (row block k+2, panel k), indexed by panel parity: read by the H
                          //                    workgroup of step k, written one launch earlier by the F workgroup of that row block (which
                          //                    overwrites the in-place block with the solved values during step k)
    double *dterm;        // [batch][2][64*64]  head start of the NEXT launch's diagonal block, indexed by block parity: -K[blk, blk] + every
                          //                    history term up to the panel finished in this launch, written by the F workgroup of that
                          //                    row block (see k_la_step, "diagonal head start"); element layout of a partial-sum slab
    double *dsum;         // [batch][2][64*64]  sum of the diagonal look-ahead slices of a block (written by the R workgroup one launch
                          //                    before the chain reads it), indexed by block parity
    double *dpart;        // [batch][2][maxslice][64*64]  look-ahead partial sums of a DIAGONAL block (block x block^T over history
                          //                    slices), written two launches before the chain needs the block, indexed by block parity
    int *flag;            // [batch] step counter of the diagonal chain: k + 1 once the D workgroup of step k is done (parking, below)
    int nbmax;            // 64-blocks of the largest patient of the batch
    int maxslice;         // slices per row block the scratch is dimensioned for
    int rows;             // row blocks the scratch is dimensioned for: 2 nbmax + 1, or nbmax + 1 for an nlml-only call (no U row blocks)
    int ring;             // index mask of the chain's hand-off slabs xk2 / pnx / dterm / dsum: 1 = by step parity
    int nbatch;           // entries of the class.  The grids' x extent is nbatch, or nbatch | 1 for a class whose entries differ in size:
                          // workgroup ids (y * extent + x) go round-robin over the 8 XCDs, so with an extent of 2, 4, 8, ... every entry's
                          // tasks land on half, a quarter, an eighth of the chip -- balanced when the entries are equally large, but in a
                          // ragged class the largest entry's tasks queue there (round 5: N = 3595 + 3 x 2400 k_la_step 3.73 ms, + 4 x 2400
                          // 2.74 ms).  An odd extent puts every entry on all XCDs; the workgroups of the padding column exit.
};

// row block index inside the scratch: M_i -> i, U_rho -> nbmax + rho, Y -> the last row block (2 nbmax, or nbmax without U rows)
struct LaRow {
    int kind;             // 0 = M, 1 = U, 2 = Y
    int blk;              // i or rho
};

struct LaSmem {
    union {
        double Bs[2][64][LA_KC + 2];     // staged shared operand of the history GEMM
        double Xs[64][LA_S];             // X_k (trsm), later D_k+1 (factor input)
    };
    double Ls[64][LA_S];                 // L[C_k+1, C_k] (B operand of the newest rank-64 term), later X_k+1 (factor output)
    alignas(16) double dv[64 + 128];                 // diag(L_kk) + the panel scratch of diag16
    double X00[16][CI_S];                // inverse of the first 16 x 16 tile while Ls is still an operand (early start of the factor, step (5))
    double logdet;
    int fail;
};
static_assert(sizeof(LaSmem) <= 80 * 1024, "two workgroups per CU");
static_assert(sizeof(double) * 2 * 64 * (LA_KC + 2) <= sizeof(double) * 64 * LA_S + 2048, "Bs and Xs share storage");

__device__ __forceinline__ double *la_row_base(const MedgpDev &L, const LaArgs &A, int b, LaRow r) {
    const size_t ld = L.ldn;
    if (r.kind == 0) return L.Kmat + (size_t)b * ld * ld + (size_t)(64 * r.blk) * ld;
    if (r.kind == 1) return L.Linv + (size_t)b * ld * ld + (size_t)(64 * r.blk) * ld;
    return A.ybuf + (size_t)b * 64 * ld;
}
__device__ __forceinline__ int la_row_index(const LaArgs &A, LaRow r) {
    return r.kind == 0 ? r.blk : (r.kind == 1 ? A.nbmax + r.blk : A.rows - 1);
}
__device__ __forceinline__ double *la_part(const LaArgs &A, int b, int parity, LaRow r, int slice) {
    return A.part + ((((size_t)b * 2 + parity) * A.rows + la_row_index(A, r)) * A.maxslice + slice) * 4096;
}
// first history panel of a row block (U_rho starts at its own diagonal block)
__device__ __forceinline__ int la_first_panel(LaRow r) { return r.kind == 1 ? r.blk : 0; }

// acc[ct] += rows(row0 + 16 w ..) of Hist[:, 64 j0 .. 64 j1) times rows 64 c .. of Lb[:, same columns]^T   (the k_ci_panel loop)
// acc[ct][r] <-> (row 16 w + 4 r + g, column 16 ct + li) of the 64x64 product.
__device__ __forceinline__ void la_gemm(const double *Hist, const double *Bpanel, int ld, int j0, int j1, v4d (&acc)[4],
                                        LaSmem &sm, int tid, int w, int li, int g) {
    const int nch = (64 * (j1 - j0)) / LA_KC;
    if (nch <= 0) return;
    const int kstart = 64 * j0;
    const double *Arow = Hist + (size_t)(16 * w + li) * ld + kstart + 2 * g;
    // staging of the shared operand: a wave instruction covers 4 rows x 256 contiguous bytes (whole 128-byte lines; the former
    // 16 rows x 4 x 16 bytes at a 64-byte stride touched 32 lines per instruction)
    const int srow = tid >> 4, scol = (tid & 15) * 2;
    const double *Bsrc = Bpanel + (size_t)srow * ld + kstart + scol;
    v2d bst[4], an[4];
#pragma unroll
    for (int u = 0; u < 4; u++) bst[u] = *(const v2d *)(Bsrc + (size_t)(16 * u) * ld);
#pragma unroll
    for (int h = 0; h < 4; h++) an[h] = *(const v2d *)(Arow + 8 * h);
#pragma unroll
    for (int u = 0; u < 4; u++) *(v2d *)&sm.Bs[0][srow + 16 * u][scol] = bst[u];
    __syncthreads();
    for (int c = 0; c < nch; c++) {
        const int buf = c & 1;
        v2d ac[4];
#pragma unroll
        for (int h = 0; h < 4; h++) ac[h] = an[h];
        if (c + 1 < nch) {
#pragma unroll
            for (int u = 0; u < 4; u++) bst[u] = *(const v2d *)(Bsrc + (size_t)(16 * u) * ld + (c + 1) * LA_KC);
#pragma unroll
            for (int h = 0; h < 4; h++) an[h] = *(const v2d *)(Arow + (c + 1) * LA_KC + 8 * h);
        }
        // the shared operand of the next 8-column group is read from LDS before the MFMAs of the current one
        v2d b_cur = *(const v2d *)&sm.Bs[buf][li][2 * g];
#pragma unroll
        for (int hct = 0; hct < 16; hct++) {
            const int h = hct >> 2, ct = hct & 3;
            v2d b_nxt = b_cur;
            if (hct < 15) b_nxt = *(const v2d *)&sm.Bs[buf][16 * ((hct + 1) & 3) + li][8 * ((hct + 1) >> 2) + 2 * g];
#pragma unroll
            for (int s = 0; s < 2; s++) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[h][s], b_cur[s], acc[ct], 0, 0, 0);
            b_cur = b_nxt;
        }
        if (c + 1 < nch) {
#pragma unroll
            for (int u = 0; u < 4; u++) *(v2d *)&sm.Bs[buf ^ 1][srow + 16 * u][scol] = bst[u];
        }
        __syncthreads();
    }
}

// out^T tiles of  block * X^T :  o[ct][r] = out(row li of the wave's 16 rows, column 16 ct + 4 r + g); X in LDS (lower).
// val[cp][r] = block(row li, column 16 cp + 4 r + g)
__device__ __forceinline__ void la_trsm(const double (*Xs)[LA_S], const v4d (&val)[4], v4d (&o)[4], int li, int g) {
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        o[ct] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int cp = 0; cp <= ct; cp++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                o[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[16 * ct + li][16 * cp + 4 * r + g], val[cp][r], o[ct], 0, 0, 0);
    }
}
__device__ __forceinline__ void la_load_t(const double *blk, int ld, v4d (&val)[4], int li, int g) {
#pragma unroll
    for (int cp = 0; cp < 4; cp++)
#pragma unroll
        for (int r = 0; r < 4; r++) val[cp][r] = blk[(size_t)li * ld + 16 * cp + 4 * r + g];
}
__device__ __forceinline__ void la_store_t(double *blk, int ld, const v4d (&o)[4], int li, int g) {
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) blk[(size_t)li * ld + 16 * ct + 4 * r + g] = o[ct][r];
}

// X_k = L_kk^-1 travels between launches as its ten lower 16 x 16 tiles only (tile-major, tile (tr, tc <= tr) at index tr (tr + 1) / 2 + tc,
// 256 doubles each: 20 KB instead of 32 KB per D / F / H task and per chain store; the tiles above the diagonal are exact zeros and
// la_trsm never reads them).  v2d number p of the packed image <-> (row, column pair) of the 64 x 64 matrix:
#ifndef LA_XPACK
#define LA_XPACK 1
#endif
#define LA_XV2 (LA_XPACK ? 5 : 8)     // 16-byte loads per thread for one X_k
__device__ __forceinline__ void la_x_rc(int p, int &rr, int &cc) {
#if LA_XPACK
    const int tile = p >> 7, within = p & 127;
    const int tr = (tile >= 1) + (tile >= 3) + (tile >= 6), tc = tile - (tr * (tr + 1)) / 2;
    rr = 16 * tr + (within >> 3); cc = 16 * tc + 2 * (within & 7);
#else
    rr = p >> 5; cc = 2 * (p & 31);
#endif
}
__device__ __forceinline__ void la_x_load(const double *Xg, v2d (&xreg)[8], int tid) {
#pragma unroll
    for (int e = 0; e < LA_XV2; e++) xreg[e] = *(const v2d *)(Xg + 2 * (tid + LA_THREADS * e));
}
__device__ __forceinline__ void la_x_to_lds(double (*Xs)[LA_S], const v2d (&xreg)[8], int tid) {
#pragma unroll
    for (int e = 0; e < LA_XV2; e++) { int rr, cc; la_x_rc(tid + LA_THREADS * e, rr, cc); *(v2d *)&Xs[rr][cc] = xreg[e]; }
}

// ---- prologue: Y row block <- [y^T; 0], diagonal block 0 factored -> X_0 ------------------------------------------------
// grid = (nbatch, 1 + nbmax): y = 0 is the diagonal role, y >= 1 initialise 64 columns of the Y block each (y = 1 also
// seeds the copy of the pre-solve block P_1,0 = K[block 1][C_0])
__global__ void __launch_bounds__(LA_THREADS) k_la_prologue(MedgpDev L, LaArgs A, int want_mode) {
    __shared__ LaSmem sm;
    const int b = blockIdx.x;
    if (b >= A.nbatch) return;   // (padding column of a ragged class)
    // status, size (k_prep's copy) and slot are requested together: `status -> branch -> slot -> size of the slot` was a chain of three
    // scalar-memory round trips in front of the first factorisation
    const int st0 = L.status[b], n0 = L.bn[b], slot = L.bslot[b];
    const int n = __builtin_amdgcn_readfirstlane(n0 & ~(st0 >> 31)), ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    if (nb < 2) return;   // failed entries (n = 0) and single-block entries: the latter are factored by k_cholinv (see its only_small switch)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (blockIdx.y == 0 && tid == 0) A.flag[b] = 0;
    if (blockIdx.y >= 1) {
        const int cb = blockIdx.y - 1;
        if (cb >= nb) return;
        double *Y = A.ybuf + (size_t)b * 64 * ld;
        const double *y = L.py + (size_t)slot * L.pld;
        for (int e = tid; e < 64 * 64; e += LA_THREADS) {
            const int rr = e >> 6, cc = 64 * cb + (e & 63);
            Y[(size_t)rr * ld + cc] = (rr == 0 && cc < n) ? y[cc] : 0.0;
        }
        if (cb == 0 && nb > 1) {
            const double *K10 = L.Kmat + (size_t)b * ld * ld + (size_t)64 * ld;
            double *Pn = A.pnx + ((size_t)b * (A.ring + 1) + 0) * 4096;
            for (int e = tid; e < 64 * 64; e += LA_THREADS) Pn[e] = K10[(size_t)(e >> 6) * ld + (e & 63)];
        }
        if (LA_H_ROLE && cb == 1 && nb > 2) {   // P_2,0 for the H workgroup of step 0
            const double *K20 = L.Kmat + (size_t)b * ld * ld + (size_t)128 * ld;
            double *Pn2 = A.pnx2 + ((size_t)b * (A.ring + 1) + 0) * 4096;
            for (int e = tid; e < 64 * 64; e += LA_THREADS) Pn2[e] = K20[(size_t)(e >> 6) * ld + (e & 63)];
        }
        return;
    }
    double *Lb = L.Kmat + (size_t)b * ld * ld, *Ub = L.Linv + (size_t)b * ld * ld;
    if (tid == 0) { sm.fail = 0; sm.logdet = 0.0; }
    for (int e = tid; e < 64 * 64; e += LA_THREADS) { const int rr = e >> 6, cc = e & 63; sm.Xs[rr][cc] = (cc <= rr) ? Lb[(size_t)rr * ld + cc] : 0.0; }
    __syncthreads();
    diag_factor_wg((ld_t *)&sm.Xs[0][0], (ld_t *)&sm.Ls[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, w, lane);
    __syncthreads();
    if (sm.fail) { if (tid == 0) L.status[b] = -2; return; }
    double *Xg = A.xk2 + ((size_t)b * (A.ring + 1) + 0) * 4096;
    for (int e = tid; e < 64 * 64; e += LA_THREADS) {
        const int rr = e >> 6, cc = e & 63;
        if (cc <= rr) Lb[(size_t)rr * ld + cc] = sm.Xs[rr][cc];
        if (want_mode) Ub[(size_t)rr * ld + cc] = (cc >= rr) ? sm.Ls[cc][rr] : 0.0;
#if !LA_XPACK
        Xg[e] = sm.Ls[rr][cc];
#endif
    }
#if LA_XPACK
    for (int p = tid; p < 1280; p += LA_THREADS) { int rr, cc; la_x_rc(p, rr, cc); *(v2d *)&Xg[2 * p] = *(const v2d *)&sm.Ls[rr][cc]; }
#endif
    if (tid == 0) L.scal[b * 4 + 0] = sm.logdet;
}

// ---- one step ----------------------------------------------------------------------------------------------------------
// grid = (nbatch, ntask): task 0 = D, tasks 1 .. nF = F (row blocks), task nF + 1 = R, then L tasks (slice-major: slice x row block, nLrowsL
// row-block slots per slice as counted by the host for the largest entry).
// Row-block enumeration for F at step k (panel k is finished, panel k+1 prepared):
//   M_i, i = k+2 .. nb-1   |  U_rho, rho = 0 .. k (inverse only)  |  Y
// and for L at step k (partials of panel k+2 over history panels <= k-1; rows that exist in panel k+2 and have such history):
//   M_i, i = k+2 .. nb-1   |  U_rho, rho = 0 .. k-1 (inverse only) |  Y       x  slice index
#ifdef LA_STAMPS
// diagnostic build (scratch/la_stamps.py): longest workgroup of each role per step (s_memtime ticks = shader cycles), kept in
// the idle slab of entry 0
#ifdef LA_STAMPS_ABS   // absolute times instead: slot 7 = earliest start (stored negated for atomicMax), slots 0-2 = latest end per role
//                        (wall_clock64 = s_memrealtime, 100 MHz, one counter for the whole device; s_memtime differs between XCDs)
#define LA_T0() const unsigned long long la_t0 = wall_clock64(), la_tm0 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicMax((unsigned long long *)L.slab + 8 * k + 7, ~la_t0)
#define LA_TEND(role) do { if (threadIdx.x == 0) atomicMax((unsigned long long *)L.slab + 8 * k + (role), (unsigned long long)wall_clock64()); } while (0)
#else
#define LA_T0() const unsigned long long la_t0 = __builtin_amdgcn_s_memtime()
#define LA_TEND(role) do { if (threadIdx.x == 0) atomicMax((unsigned long long *)L.slab + 8 * k + (role), __builtin_amdgcn_s_memtime() - la_t0); } while (0)
#endif
#ifdef LA_STAMPS_ABS
#define LA_TD(slot) do { if (is_D && threadIdx.x == 0 && (slot) == 6) ((unsigned long long *)L.slab)[8 * k + 6] = __builtin_amdgcn_s_memtime() - la_tm0; } while (0)
#else
#define LA_TD(slot) do { if (is_D && threadIdx.x == 0) ((unsigned long long *)L.slab)[8 * k + (slot)] = __builtin_amdgcn_s_memtime() - la_t0; } while (0)
#endif
#define LA_TF(slot) do { if (role == 1 && row.kind == 0 && (row.blk == k + 2 || row.blk == k + 3) && threadIdx.x == 0) ((unsigned long long *)L.slab)[640 + 16 * k + 8 * (row.blk - k - 2) + (slot)] = __builtin_amdgcn_s_memtime() - la_t0; } while (0)
#else
#define LA_TF(slot) do {} while (0)
#define LA_TD(slot) do {} while (0)
#define LA_T0() do {} while (0)
#define LA_TEND(role) do {} while (0)
#endif
// One task of step k: the body of k_la_step.
struct LaTask {
    int role;             // 0 = D (diagonal chain), 1 = F, 2 = L (look-ahead), 3 = R, 4 = H (diagonal head start)
    LaRow row;
    int slice;
    bool diag_ahead;
};
// slab index of a chain hand-off buffer for step / block x
__device__ __forceinline__ int la_ring(const LaArgs &A, int x) { return x & A.ring; }

// task index of a step's list (0 = D, 1 .. nF = F, nF + 1 = R, nF + 2 = H, then the L tasks slice-major) -> role; false: empty slot
__device__ __forceinline__ bool la_decode(const LaArgs &A, int k, int want_inv, int task, int nLrowsL, LaTask &T) {
    // the task lists are laid out for the largest patient of the batch (A.nbmax)
    const int nM_F = A.nbmax - (k + 2) > 0 ? A.nbmax - (k + 2) : 0;
    const int nU_F = want_inv ? k + 1 : 0;
    const int nF = nM_F + nU_F + 1;
    T.slice = 0;
    T.diag_ahead = false;
    if (task == 0) { T.role = 0; T.row.kind = 0; T.row.blk = k + 1; }
    else if (task <= nF) {
        T.role = 1;
        int t = task - 1;
        if (t < nM_F) { T.row.kind = 0; T.row.blk = k + 2 + t; }
        else if (t < nM_F + nU_F) { T.row.kind = 1; T.row.blk = t - nM_F; }
        else { T.row.kind = 2; T.row.blk = 0; }
    } else if (!la_fold(k) && task == nF + 1) {
        T.role = 3; T.row.kind = 0; T.row.blk = k + 2;   // R: sum of the diagonal look-ahead slices of block k+2
    } else if (LA_H_ROLE && task == nF + LA_NAUX(k)) {
        T.role = 4; T.row.kind = 0; T.row.blk = k + 2;   // H: diagonal head start of block k+2 (incl. the slices when la_fold(k))
    } else {
        T.role = 2;
        // slice-major, and only the ceil(k / la_slice_len(k)) slices that exist at this step are launched: the L tasks that have work
        // are then consecutive in dispatch order, which the hardware deals round-robin over the 8 XCDs.  (Row-major over
        // maxslice slots per row put the live tasks -- slices 0, 1 of every row early on -- at ids = 0, 1 mod 16, i.e. on TWO
        // of the eight XCDs, behind ~850 empty workgroups: at N = 4096 the L role ended at 49 us of a step whose diagonal
        // chain needs 34 us.)
        int t = task - 1 - LA_NAUX(k) - nF;
        T.slice = t / nLrowsL;
        t -= T.slice * nLrowsL;
        if (T.slice >= A.maxslice) return false;
        const int nU_L = want_inv ? k : 0;
        // slot 0 of every slice used to be row M_k+2 of panel k+2 -- the NEXT diagonal block, consumed by the chain alone.  The chain
        // now gets its block ready-made (diagonal head start), so the slot instead looks one block further ahead: partial sums of
        // the diagonal block k+3 (rows of block k+3 times themselves) over the same history slice, summed by the F workgroup of that
        // row block in the next launch -- off the chain.
        T.diag_ahead = (t == 0 && nM_F >= 1);
        if (T.diag_ahead) { T.row.kind = 0; T.row.blk = k + 3; }
        else if (t < nM_F) { T.row.kind = 0; T.row.blk = k + 2 + t; }
        else if (t < nM_F + nU_L) { T.row.kind = 1; T.row.blk = t - nM_F; }
        else if (t == nM_F + nU_L) { T.row.kind = 2; T.row.blk = 0; }
        else return false;
    }
    return true;
}

// returns 1 when the D role met a non-positive pivot (the caller marks the entry: status -2), else 0
__device__ __forceinline__ int la_body(const MedgpDev &L, const LaArgs &A, LaSmem &sm, int b, int n_in, int k, int want_mode, const LaTask &T) {
    LA_T0();
    const int ld = L.ldn;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = T.role, slice = T.slice;
    const LaRow row = T.row;
    const bool diag_ahead = T.diag_ahead;
    const bool is_D = (role == 0);
    // ---- (0) D and F: every global operand whose address depends on (entry, step) alone is requested before anything else -- X_k, the
    //      pre-solve copy P_k+1,k, the chain's head-start slabs -- and the entry's status / size words travel WITH them: one memory
    //      round trip at the head of the step's critical path instead of two dependent ones (the slabs exist for every step of
    //      the schedule, so the requests are valid whatever the patient's size turns out to be)
    const double *Xg = A.xk2 + ((size_t)b * (A.ring + 1) + la_ring(A, k)) * 4096;
    v2d xreg[8];
    v4d pval[4];
    const bool head_start = is_D && k >= 1;
    v4d acc[4];           // pre-solve block of panel k+1 (non-transposed tiles); the chain's head start lands here directly
    double dsv[16];
    const bool is_H = (role == 4);
    if ((role <= 1 && (is_D || !LA_F_LATE)) || is_H) {
        la_x_load(Xg, xreg, tid);
        la_load_t((is_H ? A.pnx2 : A.pnx) + ((size_t)b * (A.ring + 1) + la_ring(A, k)) * 4096 + (size_t)(16 * w) * 64, 64, pval, li, g);
    }
    int n = n_in;
    if (n_in < 0) {
        // the two words are requested together and merged arithmetically: with `if (st0 < 0) return` in between, the compiler
        // issues the size word only behind the branch -- a second scalar-memory round trip at the head of every task.  A failed
        // entry (status < 0) gets n = 0, which the size check below turns into the same early return.
        const int st0 = L.status[b], n0 = L.bn[b];
        n = __builtin_amdgcn_readfirstlane(n0 & ~(st0 >> 31));
    }
    const int npad = medgp_roundup(n, 64), nb = npad / 64;
    if (k >= nb || nb < 2) return 0;
    const int want_inv = want_mode & 1;
    double *Lb = L.Kmat + (size_t)b * ld * ld, *Ub = L.Linv + (size_t)b * ld * ld;
    const int c0 = 64 * k, c1 = 64 * (k + 1);
    const bool has_next = (k + 1 < nb);
    (void)want_inv;
    if (row.kind == 0 && row.blk >= nb && !(role == 0)) return 0;   // beyond this patient's blocks
    double *Rb = la_row_base(L, A, b, row);
    LA_TF(0);

    // ============================== H: diagonal head start of block k+2 for the next launch's chain ======================
    // -K[k+2, k+2] + row[C_k] row[C_k]^T + row[C_k-1] row[C_k-1]^T  with  row[C_k] = P_k+2,k X_k^T  re-derived from the copy the F
    // workgroup of this row block left one launch ago (the same 40 MFMAs per wave that workgroup runs on the in-place block: same
    // operands, same order, same bits), then exactly the product / panel sequence that used to be the tail of that F task.
    if (is_H) {
        if (k + 2 >= nb) return 0;
        const int c2 = 64 * (k + 2);
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[ct][r] = -Rb[(size_t)(16 * w + 4 * r + g) * ld + c2 + 16 * ct + li];
        la_x_to_lds(sm.Xs, xreg, tid);
        __syncthreads();
        v4d oh[4];
        la_trsm(sm.Xs, pval, oh, li, g);
        __syncthreads();   // every wave is done reading X_k
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) sm.Xs[16 * w + li][16 * ct + 4 * r + g] = oh[ct][r];
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double a = oh[ct][r];
#pragma unroll
                for (int cb = 0; cb < 4; cb++)
                    acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sm.Xs[16 * cb + li][16 * ct + 4 * r + g], acc[cb], 0, 0, 0);
            }
        __syncthreads();   // the rows in Xs have been read: la_gemm stages through the same LDS
        if (k >= 1) la_gemm(Rb, Lb + (size_t)c2 * ld, ld, k - 1, k, acc, sm, tid, w, li, g);
        if (la_fold(k)) {   // ... + the diagonal look-ahead slices over panels 0 .. k-2 (written by the L tasks of the previous launch): summed from zero in
            // slice order and added ONCE, exactly what the R workgroup and the chain's `acc += dsum` did between them -- same bits
            const int nsd = la_nsd(k);
            const double *Dp = A.dpart + (((size_t)b * 2 + ((k + 2) & 1)) * A.maxslice) * 4096 + (size_t)w * 1024 + lane;
            double sacc[16];
#pragma unroll
            for (int e = 0; e < 16; e++) sacc[e] = 0.0;
            int sd = 0;
            for (; sd + 1 < nsd; sd += 2) {
                double pv[2][16];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int e = 0; e < 16; e++) pv[u][e] = Dp[(size_t)(sd + u) * 4096 + e * 64];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int e = 0; e < 16; e++) sacc[e] += pv[u][e];
            }
            for (; sd < nsd; sd++) {
#pragma unroll
                for (int e = 0; e < 16; e++) sacc[e] += Dp[(size_t)sd * 4096 + e * 64];
            }
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e >> 2][e & 3] += sacc[e];
        }
        double *Dt = A.dterm + ((size_t)b * (A.ring + 1) + la_ring(A, k + 2)) * 4096 + (size_t)w * 1024 + lane;
#pragma unroll
        for (int e = 0; e < 16; e++) Dt[e * 64] = acc[e >> 2][e & 3];
        return 0;
    }

    // ============================== R: diagonal look-ahead slices of block k+2 -> one slab ===============================
    // (slices over panels 0 .. k-2, written by the diag-ahead L tasks of the previous launch; read by the chain of the next launch
    //  together with A.dterm.  All loads of a group of four slices are independent; fixed summation order.)
    if (role == 3) {
        if (k + 2 >= nb) return 0;
        const int sl_prev = la_slice_len(k - 1);   // the slices were written one step earlier
        const int nsd = (k >= 2) ? (k - 1 + sl_prev - 1) / sl_prev : 0;
        const double *Dp = A.dpart + (((size_t)b * 2 + ((k + 2) & 1)) * A.maxslice) * 4096 + (size_t)w * 1024 + lane;
        double sacc[16];
#pragma unroll
        for (int e = 0; e < 16; e++) sacc[e] = 0.0;
        int sd = 0;
        for (; sd + 3 < nsd; sd += 4) {
            double pv[4][16];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 16; e++) pv[u][e] = Dp[(size_t)(sd + u) * 4096 + e * 64];
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 16; e++) sacc[e] += pv[u][e];
        }
        for (; sd < nsd; sd++) {
#pragma unroll
            for (int e = 0; e < 16; e++) sacc[e] += Dp[(size_t)sd * 4096 + e * 64];
        }
        double *Ds = A.dsum + ((size_t)b * (A.ring + 1) + la_ring(A, k + 2)) * 4096 + (size_t)w * 1024 + lane;
#pragma unroll
        for (int e = 0; e < 16; e++) Ds[e * 64] = sacc[e];
        return 0;
    }

    // ============================== L: look-ahead partial sum of panel k+2 =============================================
    if (role == 2) {
        if (k + 2 >= nb) return 0;
        const int sl_cur = la_slice_len(k);
        const int j0 = la_first_panel(row) + sl_cur * slice;
        int j1 = j0 + sl_cur;
        if (j1 > k) j1 = k;                       // history panels <= k-1
        if (j0 >= j1) return 0;
        v4d acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[ct] = (v4d){0.0, 0.0, 0.0, 0.0};
        // shared operand: the rows of block k+2 (panel k+2), or -- diagonal look-ahead -- the task's own rows (block k+3)
        la_gemm(Rb, Lb + (size_t)(64 * (diag_ahead ? k + 3 : k + 2)) * ld, ld, j0, j1, acc, sm, tid, w, li, g);
        double *P = (diag_ahead ? A.dpart + (((size_t)b * 2 + ((k + 3) & 1)) * A.maxslice + slice) * 4096
                                : la_part(A, b, (k + 2) & 1, row, slice)) + (size_t)w * 1024 + lane;
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) P[(ct * 4 + r) * 64] = acc[ct][r];
        LA_TEND(2);
        return 0;
    }

    // ============================== D and F ============================================================================
    if (is_D && !has_next) return 0;                // the last panel has no next diagonal block
#ifdef LA_STAMPS_ABS
    if (is_D && threadIdx.x == 0) ((unsigned long long *)L.slab)[8 * k + 5] = la_t0;   // the chain's own start (wall clock)
#endif
    // (s_setprio 3 for the diagonal chain, which shares its CU with a bulk workgroup once the chip is full: measured, no effect --
    //  the chain's cycle count does not change with the load, the shader clock does: 71 k cycles take 29.7 us early and 37.5 us
    //  from step 16 of N = 4096 on, when more than 256 L workgroups run fp64 MFMA)
    double *oblk = Rb + (size_t)(16 * w) * ld + c0;   // the role's own block of panel k: read at its use (F roles are not on
                                                      // the critical path; holding it from here cost the kernel its second
                                                      // workgroup per CU)
    // ---- (1) pre-solve block of panel k+1: acc = -init + partials + term of panel k-1     (non-transposed tiles)
    if (!head_start) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[ct] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    const int jf = la_first_panel(row);
    // Diagonal head start (D role, k >= 1): the F workgroup of row block k+1 left  -K[k+1, k+1] + sum_{j <= k-1} row[C_j] row[C_j]^T
    // in A.dterm during the previous launch (panel k-1 from its registers, panel k-2 as a 64 x 64 x 64 product, everything older
    // from the diagonal look-ahead slices A.dpart), so the chain neither loads its initial block, nor walks the chain of partial-sum
    // round trips (0.9 us per two slices, up to 15 slices at N = 4096), nor runs the product of panel k-1.
    if (has_next) {
        if (head_start) {
            const double *Dt = A.dterm + ((size_t)b * (A.ring + 1) + la_ring(A, k + 1)) * 4096 + (size_t)w * 1024 + lane;
            const double *Ds = A.dsum + ((size_t)b * (A.ring + 1) + la_ring(A, k + 1)) * 4096 + (size_t)w * 1024 + lane;
            // (with LA_D_EARLY only the tiles left of / on the diagonal -- column tile <= the wave's row tile -- are ever used: the
            //  others are not requested: 40 of the 64 KB of the two slabs)
#pragma unroll
            for (int e = 0; e < 16; e++) if (!LA_D_EARLY || (e >> 2) <= w) acc[e >> 2][e & 3] = Dt[e * 64];
            if (!la_fold(k - 1)) {   // (the slab exists only when step k-1 ran an R workgroup)
#pragma unroll
            for (int e = 0; e < 16; e++) if (!LA_D_EARLY || (e >> 2) <= w) dsv[e] = Ds[e * 64];     // (written by the R workgroup of the previous launch, k >= 1: zeros when there were no slices)
            }
            // (the two slabs are added in front of step (5): their round trip runs under the solve)
        } else if (row.kind != 1) {   // K rows and the y row carry their own initial values in place; U rows start from zero
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[ct][r] = -Rb[(size_t)(16 * w + 4 * r + g) * ld + c1 + 16 * ct + li];
        }
        // partial sums over history panels jf .. k-2, written by the L tasks of the previous launch (the chain's block arrives
        // with them already added)
        const int hist_end = head_start ? jf : k - 1;               // panels jf .. hist_end-1
        // (two slices per iteration, both requested before either is added: the loop is a chain of memory round trips --
        //  0.9 us per slice on the diagonal chain, 15 slices at the end of N = 4096; the order of the additions is unchanged)
        const int sl_prev = la_slice_len(k - 1);   // the partial sums were written one step earlier
        const int nsum = (hist_end - jf + sl_prev - 1) / sl_prev;
        const double *P0 = la_part(A, b, (k + 1) & 1, row, 0) + (size_t)w * 1024 + lane;
        int s = 0;
        // (LA_NSUM slabs per memory round trip, all 64 loads requested before the first addition -- the loop is a chain of round trips
        //  of 2-4 k cycles each, 12-15 slabs at the end of N = 4096, and the F tasks are the longest of a late step; the order of
        //  the additions is unchanged)
        for (; s + LA_NSUM - 1 < nsum; s += LA_NSUM) {
            double pv[LA_NSUM][16];
#pragma unroll
            for (int u = 0; u < LA_NSUM; u++)
#pragma unroll
                for (int e = 0; e < 16; e++) pv[u][e] = P0[(size_t)(s + u) * 4096 + e * 64];
#pragma unroll
            for (int u = 0; u < LA_NSUM; u++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[e >> 2][e & 3] += pv[u][e];
        }
        for (; s < nsum; s++) {
#pragma unroll
            for (int e = 0; e < 16; e++) acc[e >> 2][e & 3] += P0[(size_t)s * 4096 + e * 64];
        }
        // F: X_k and the pre-solve copy are requested here, in front of the product of panel k-1 that hides their round trip (held from
        // the top of the task they would take the registers of two of the four slabs in flight above)
        if (!is_D && LA_F_LATE) {
            la_x_load(Xg, xreg, tid);
            la_load_t(A.pnx + ((size_t)b * (A.ring + 1) + la_ring(A, k)) * 4096 + (size_t)(16 * w) * 64, 64, pval, li, g);
        }
        LA_TF(1);
        // panel k-1 (final since the previous launch); the D role got this term with its head start
        if (!head_start && k >= 1 && jf <= k - 1) la_gemm(Rb, Lb + (size_t)c1 * ld, ld, k - 1, k, acc, sm, tid, w, li, g);
    }
    if (!has_next && !is_D && LA_F_LATE) {   // last step: only the solve of panel k is left
        la_x_load(Xg, xreg, tid);
    }
    LA_TD(3);
    LA_TF(2);
    // ---- (2) X_k -> LDS (Bs is dead: la_gemm ends with a barrier)
    la_x_to_lds(sm.Xs, xreg, tid);
    if (is_D && tid == 0) { sm.fail = 0; sm.logdet = 0.0; }   // (before the first barrier: wave 0 starts factoring ahead of the others, step (5))
    __syncthreads();
    // ---- (3) L[C_k+1, C_k] = P_k+1,k X_k^T for the wave's 16 rows of block k+1 -> LDS (and, D only, to memory)
    v4d o[4];
    if (has_next) {
        la_trsm(sm.Xs, pval, o, li, g);
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) sm.Ls[16 * w + li][16 * ct + 4 * r + g] = o[ct][r];
        // (D: the final L[C_k+1, C_k] goes to memory from this LDS tile, row-contiguous, in front of the factorisation -- from the
        //  registers it took 16 store instructions of 16 rows x 32 bytes each: 3.1 k cycles of issue on the chain)
    }
    // ---- (4) own rows of panel k: row_r[C_k] = P_r,k X_k^T   (D: that is the block above, already in o;  U_k: the
    //      diagonal block U_kk itself, final since the previous launch)
    if (!is_D) {
        if (row.kind == 1 && row.blk == k) {
#if LA_UKK_BY_F
            // U_kk = X_k^T is not stored by the chain any more (32 KB less on its store tail): this task, which has X_k in LDS anyway,
            // derives its rows from it -- the tiles left of the diagonal are zero, the diagonal tile of X_k has exact zeros above its
            // diagonal -- and stores them; the first readers of this block (history of row block U_k) run one launch later
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) o[ct][r] = (ct >= w) ? sm.Xs[16 * ct + 4 * r + g][16 * w + li] : 0.0;
            la_store_t(oblk, ld, o, li, g);
#else
            la_load_t(oblk, ld, o, li, g);
#endif
        } else {
            v4d oval[4];
            la_load_t(oblk, ld, oval, li, g);
            la_trsm(sm.Xs, oval, o, li, g);
            la_store_t(oblk, ld, o, li, g);
        }
    }
    if (!has_next) { LA_TEND(1); return 0; }
    __syncthreads();   // Ls complete
    LA_TD(4);
    LA_TF(3);
    // ---- (5) newest rank-64 term: acc += row_r[C_k] L[C_k+1,C_k]^T, straight from the trsm registers (A operand)
    if (head_start && !la_fold(k - 1)) {
#pragma unroll
        for (int e = 0; e < 16; e++) if (!LA_D_EARLY || (e >> 2) <= w) acc[e >> 2][e & 3] += dsv[e];
    }
#if !LA_D_EARLY
    const bool d_split = false;
#else
    const bool d_split = is_D;
#endif
    if (d_split) {
        // D role: the next diagonal block is symmetric and only its lower tiles (column tile <= the wave's row tile) are ever read,
        // and the factorisation of its FIRST 16 x 16 tile needs nothing but the 16 rows wave 0 owns.  Wave 0 therefore computes that
        // one tile (16 MFMAs) and starts diag16 on it at once, while waves 1-3 run their rows of the rank-64 update (2, 3, 4 tiles):
        // the serial chain of the block's 64 pivots begins ~6 k cycles earlier than behind the full update and a barrier (round 4,
        // profiles/r04_la_phase_stamps.txt).  Every tile receives the same MFMAs in the same order as before: same bits.
        // X_k in Xs is dead (every wave passed the barrier behind its trsm), so the tiles of D_k+1 go there directly; the inverse
        // of the first tile goes to sm.X00 because the other waves still read Ls as the B operand of their updates.
#pragma unroll
        for (int cb = 0; cb < 4; cb++) {
            if (cb <= w) {
#pragma unroll
                for (int ct = 0; ct < 4; ct++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(o[ct][r], sm.Ls[16 * cb + li][16 * ct + 4 * r + g], acc[cb], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; r++) sm.Xs[16 * w + 4 * r + g][16 * cb + li] = -acc[cb][r];
            }
        }
        if (w == 0) {
            __builtin_amdgcn_wave_barrier();
            if (!diag16<LA_FASTLOADS>((ld_t *)&sm.Xs[0][0], (ld_t *)&sm.X00[0][0], (ld_t *)sm.dv, (ld_t *)sm.dv + 64, lane)) { if (lane == 0) sm.fail = 1; }
        } else {
            // final L[C_k+1, C_k] to memory from the LDS tile (complete since the barrier above; 16-byte accesses, whole lines): by the
            // three waves that would otherwise wait for wave 0 at the next barrier
            for (int e = tid - 64; e < 64 * 32; e += LA_THREADS - 64) {
                const int rr = e >> 5, cc = 2 * (e & 31);
                *(v2d *)&Lb[(size_t)(c1 + rr) * ld + c0 + cc] = *(const v2d *)&sm.Ls[rr][cc];
            }
        }
    } else {
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const double a = o[ct][r];
#pragma unroll
            for (int cb = 0; cb < 4; cb++)
                acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sm.Ls[16 * cb + li][16 * ct + 4 * r + g], acc[cb], 0, 0, 0);
        }
    }
    if (!is_D) {
        // pre-solve block of panel k+1 (value = -acc), stored in place
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) Rb[(size_t)(16 * w + 4 * r + g) * ld + c1 + 16 * ct + li] = -acc[ct][r];
        LA_TF(4);
#if LA_H_ROLE
        if (row.kind == 0 && row.blk == k + 3) {   // next step's P_k+3,k+1 for the H workgroup of that step (this task overwrites the in-place block then)
            double *Pn2 = A.pnx2 + ((size_t)b * (A.ring + 1) + la_ring(A, k + 1)) * 4096;
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) Pn2[(16 * w + 4 * r + g) * 64 + 16 * ct + li] = -acc[ct][r];
        }
#endif
        if (row.kind == 0 && row.blk == k + 2) {   // next step's P_k+2,k+1: everybody reads this copy
            double *Pn = A.pnx + ((size_t)b * (A.ring + 1) + la_ring(A, k + 1)) * 4096;
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) Pn[(16 * w + 4 * r + g) * 64 + 16 * ct + li] = -acc[ct][r];
#if !LA_H_ROLE
            // diagonal head start for the NEXT launch's chain (row block k+2 is its D role):  -K[k+2, k+2] + row[C_k] row[C_k]^T,
            // this workgroup's 64 rows of panel k (o, final) times themselves.  X_k in Xs is dead (every wave passed the barrier
            // after its trsm), so the rows go there as the shared operand; acc is free again.
            const int c2 = 64 * (k + 2);
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sm.Xs[16 * w + li][16 * ct + 4 * r + g] = o[ct][r];
                    acc[ct][r] = -Rb[(size_t)(16 * w + 4 * r + g) * ld + c2 + 16 * ct + li];
                }
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < 4; ct++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double a = o[ct][r];
#pragma unroll
                    for (int cb = 0; cb < 4; cb++)
                        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sm.Xs[16 * cb + li][16 * ct + 4 * r + g], acc[cb], 0, 0, 0);
                }
            // ... + panel k-1 (final since the previous launch); the diagonal look-ahead slices over panels 0 .. k-2 are summed by
            // the R workgroup of this launch into A.dsum (left to this workgroup, its chain of up to 15 slab round trips made it
            // the longest task of the launch at N = 4096)
            __syncthreads();   // the rows in Xs have been read: la_gemm stages through the same LDS
            LA_TF(5);
            if (k >= 1) la_gemm(Rb, Lb + (size_t)c2 * ld, ld, k - 1, k, acc, sm, tid, w, li, g);
            double *Dt = A.dterm + ((size_t)b * (A.ring + 1) + la_ring(A, k + 2)) * 4096 + (size_t)w * 1024 + lane;
#pragma unroll
            for (int e = 0; e < 16; e++) Dt[e * 64] = acc[e >> 2][e & 3];
            LA_TF(6);
#endif
        }
        LA_TEND(1);
        return 0;
    }
    LA_TD(5);
    // ---- (6) D: factor the next diagonal block
    __syncthreads();   // every wave is done reading Xs / Ls
#if !LA_D_EARLY
    for (int e = tid; e < 64 * 32; e += LA_THREADS) {   // final L[C_k+1, C_k]: 16-byte accesses, whole lines
        const int rr = e >> 5, cc = 2 * (e & 31);
        *(v2d *)&Lb[(size_t)(c1 + rr) * ld + c0 + cc] = *(const v2d *)&sm.Ls[rr][cc];
    }
#endif
#if LA_D_EARLY
    // (the tiles of D_k+1 are in Xs already, the first one factored, L[C_k+1, C_k] stored: step (5); the barrier above covers all of it)
#ifdef LA_FSTAMPS
    diag_factor_wg((ld_t *)&sm.Xs[0][0], (ld_t *)&sm.Ls[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, w, lane, (const ld_t *)&sm.X00[0][0], (unsigned long long *)L.slab + 2048 + 16 * k);
#else
    diag_factor_wg((ld_t *)&sm.Xs[0][0], (ld_t *)&sm.Ls[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, w, lane, (const ld_t *)&sm.X00[0][0]);
#endif
#else
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) sm.Xs[16 * w + 4 * r + g][16 * ct + li] = -acc[ct][r];
    __syncthreads();
#ifdef LA_FSTAMPS
    diag_factor_wg((ld_t *)&sm.Xs[0][0], (ld_t *)&sm.Ls[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, w, lane, nullptr, (unsigned long long *)L.slab + 2048 + 16 * k);
#else
    diag_factor_wg((ld_t *)&sm.Xs[0][0], (ld_t *)&sm.Ls[0][0], (ld_t *)sm.dv, (li_t *)&sm.fail, (ld_t *)&sm.logdet, w, lane);
#endif
#endif
    __syncthreads();
    LA_TD(6);
    if (sm.fail) return 1;
    double *Xn = A.xk2 + ((size_t)b * (A.ring + 1) + la_ring(A, k + 1)) * 4096;
    // (two columns per lane and store: 16-byte accesses, 8 passes instead of 16 on the tail of the chain; unrolled, so that the LDS reads
    //  of all passes are in flight together instead of one read-latency per pass)
#pragma unroll
    for (int e = tid; e < 64 * 32; e += LA_THREADS) {
        const int rr = e >> 5, cc = 2 * (e & 31);
        const v2d xv = *(const v2d *)&sm.Ls[rr][cc];
        if (cc + 1 <= rr) *(v2d *)&Lb[(size_t)(c1 + rr) * ld + c1 + cc] = *(const v2d *)&sm.Xs[rr][cc];
        else if (cc == rr) Lb[(size_t)(c1 + rr) * ld + c1 + cc] = sm.Xs[rr][cc];
        // (with U rows in the schedule -- want_mode bit 0 -- the F task of row block U_k+1 stores U_k+1,k+1 in the next launch: LA_UKK_BY_F)
        if (LA_UKK_BY_F ? (want_mode == 2) : (want_mode != 0)) *(v2d *)&Ub[(size_t)(c1 + rr) * ld + c1 + cc] = (v2d){(cc >= rr) ? sm.Ls[cc][rr] : 0.0, (cc + 1 >= rr) ? sm.Ls[cc + 1][rr] : 0.0};
#if !LA_XPACK
        *(v2d *)&Xn[2 * e] = xv;
#else
        (void)xv;
#endif
    }
#if LA_XPACK
#pragma unroll
    for (int p = tid; p < 1280; p += LA_THREADS) { int rr, cc; la_x_rc(p, rr, cc); *(v2d *)&Xn[2 * p] = *(const v2d *)&sm.Ls[rr][cc]; }
#endif
    if (tid == 0) L.scal[b * 4 + 0] += sm.logdet;   // the chain's steps are ordered: fixed summation order
    LA_TEND(0);
    return 0;
}

// park >= 0 (single-entry calls): the workgroup with blockIdx.y == park does no work -- it keeps its slot (its LDS) until the
// diagonal chain of this step is done.  Workgroup 0 (the chain) is dispatched first and lands on the first CU of XCD 0; once every
// CU holds one workgroup, the next one dispatched to that XCD (id 256) is placed on the same CU, and from then on a bulk (L / F)
// workgroup shares the chain's CU and its fp64 pipe: the chain slows from 63 k to 82-97 k cycles.  Parked there, a sleeping
// workgroup costs one of 512 slots and nothing else; if the placement guess is wrong it is merely a wasted slot.  It waits on a
// flag with a bounded number of polls (no way to hang).
__global__ void __launch_bounds__(LA_THREADS, 2) k_la_step(MedgpDev L, LaArgs A, int k, int want_mode, int nLrowsL, int park) {
    __shared__ LaSmem sm;
    const int b = blockIdx.x;
    if (b >= A.nbatch) return;   // (padding column of a ragged class)
    if (park >= 0 && (int)blockIdx.y == park) {
        const int st0 = L.status[b], n0 = L.bn[b];
        const int nb = medgp_roundup(__builtin_amdgcn_readfirstlane(n0), 64) / 64;
        if (st0 < 0 || threadIdx.x >= 64 || k + 1 >= nb || nb < 2) return;   // one wave holds the slot; the last step has no chain
        for (int it = 0; it < 2000; it++) {              // <= ~3 ms, far beyond any step
            if (__hip_atomic_load(&A.flag[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > k) break;
            __builtin_amdgcn_s_sleep(64);
        }
        return;
    }
    const int task = (park >= 0 && (int)blockIdx.y > park) ? (int)blockIdx.y - 1 : (int)blockIdx.y;
    LaTask T;
    if (!la_decode(A, k, want_mode & 1, task, nLrowsL, T)) return;
    const int failed = la_body(L, A, sm, b, -1, k, want_mode, T);   // (-1: the body reads the entry's status and size itself)
    if (T.role == 0 && k + 1 < A.nbmax && threadIdx.x == 0) {
        if (failed) L.status[b] = -2;
        __hip_atomic_store(&A.flag[b], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // releases the parked workgroup
    }
}

// ---- epilogue of the factorisation: z, quad = z^T z, alpha = U z, status = jitter count -----------------------------------
// grid = (nbatch, 4 nbmax): block y owns rows 16 y .. 16 y + 15 of alpha (16 lanes per row, four independent loads in flight
// per lane); block 0 also publishes quad and the status
__global__ void __launch_bounds__(256) k_la_finish(MedgpDev L, LaArgs A, int want_mode) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (b >= A.nbatch) return;   // (padding column of a ragged class)
    const int st0 = L.status[b], n0 = L.bn[b];
    if (st0 < 0) return;
    const int n = n0, ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    const int rq = blockIdx.y;
    if (16 * rq >= npad || nb < 2) return;
    const double *zrow = A.ybuf + (size_t)b * 64 * ld;   // row 0 of the Y block
    const int i = 16 * rq + 4 * w + (lane >> 4), li = lane & 15;
    if (want_mode & 1) {
        // alpha_i = sum_{c >= 64 (i / 64)} U[i][c] z[c]   (U has exact zeros left of the diagonal inside its diagonal block)
        const double *ur = L.Linv + (size_t)b * ld * ld + (size_t)i * ld;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int c = (i & ~63) + li;
        for (; c + 48 < npad; c += 64) {
            s0 += ur[c] * zrow[c];
            s1 += ur[c + 16] * zrow[c + 16];
            s2 += ur[c + 32] * zrow[c + 32];
            s3 += ur[c + 48] * zrow[c + 48];
        }
        for (; c < npad; c += 16) s0 += ur[c] * zrow[c];
        double s = (s0 + s1) + (s2 + s3);
        for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (li == 0) L.alpha[(size_t)b * ld + i] = s;
    }
    if (tid < 16) L.z[(size_t)b * ld + 16 * rq + tid] = zrow[16 * rq + tid];
    if (rq == 0) {
        double s = 0.0;
        for (int k2 = tid; k2 < npad; k2 += 256) s += zrow[k2] * zrow[k2];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) red[w] = s;
        __syncthreads();
        if (tid == 0) {
            L.scal[b * 4 + 1] = red[0] + red[1] + red[2] + red[3];
            // attempt 0 succeeded: no jitter.  (Test hook dbg_fail >= 1: the attempt counts as failed, the retry launch takes over.)
            L.status[b] = (L.dbg_fail > 0) ? -2 : 0;
        }
    }
}
