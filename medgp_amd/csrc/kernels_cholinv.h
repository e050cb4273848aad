// kernels_cholinv.h -- fused Cholesky + triangular inverse + forward solve for one patient per
// workgroup, on fp64 MFMA (v_mfma_f64_16x16x4_f64), gfx950.
//
// Replaces LAPACKE_spotrf + spotrs + strtri of the reference (ref: inference/c_inference_exact.cpp:96-143).
//
// Formulation.  Append the identity below K:  T = [K; I]  (2N x N).  Running the left-looking blocked
// Cholesky panel recurrence on ALL rows of T,
//      row_i[C_k] <- ( init_i[C_k] - sum_{j<k} row_i[C_j] L[C_k, C_j]^T ) L_kk^-T ,
// turns the K rows into L and the identity rows into U = L^-T (the same recurrence that turns an
// appended y^T into z^T = (L^-1 y)^T).  Every 64-wide step k therefore updates exactly N rows:
// (nb - k) blocks of K rows with history length 64k and k blocks of U rows with history 64(k - rho):
// uniform work per step, split over the waves in 16-row units (see cholinv_attempt), all GEMMs on MFMA.
//
// Data (row-major, leading dimension ldn):  Kmat: K lower in, L lower out.  Linv: U = L^-T upper out,
// i.e. (L^-1)[i][j] = U[j][i]; the strictly-lower part of each diagonal 64-block of U is zeroed.
// MFMA orientation: tiles are computed TRANSPOSED (acc[c][i], c = column inside the block, i = row), so
// that an accumulator tile is directly the B operand of the triangular solve  out^T = L_kk^-1 acc^T
// (C/D layout row = (lane>>4) + 4 reg  <->  B operand k = 4 step + (lane>>4)).
#pragma once
#include "medgp_dev.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
// Explicit global-address-space views.  k_cholinv hands its by-value argument struct to noinline helpers by reference,
// after which hipcc no longer proves that the pointers inside are global and emits FLAT loads / stores (which also
// count on lgkmcnt, i.e. every LDS wait then waits for the HBM operands too).
typedef __attribute__((address_space(1))) double gd_t;
typedef __attribute__((address_space(1))) v2d gv2d_t;
// LDS views for the helpers that receive tile pointers (an out-of-line callee otherwise falls back to FLAT accesses)
typedef __attribute__((address_space(3))) double ld_t;
typedef __attribute__((address_space(3))) v2d lv2d_t;
typedef __attribute__((address_space(3))) int li_t;
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v2d ci_bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    union { v4i i; v2d d; } cv;
    cv.i = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return cv.d;
}

#ifndef CI_KC
#define CI_KC 16          // k-columns staged per barrier; the history operand is loaded 16 columns at a time
#endif

// Diagnostic build only (-DMEDGP_STAMPS, never shipped): per-phase s_memtime sums per wave, written to a debug
// buffer that no other code reads (guide section 7, "In-kernel stamps").
#ifdef MEDGP_STAMPS
#define STAMP(idx)                                                                                 \
    do {                                                                                           \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                \
        if (lane == 0) sm.stamps[tid >> 6][idx] += t_ - st_last;   /* LDS, not SGPRs: 16 more live SGPRs broke the build */ \
        st_last = t_;                                                                              \
    } while (0)
#else
// (An earlier form of the loop needed these positions as scheduling anchors to keep hipcc from spilling inside the MFMA
//  block; with fixed-body phases and scalar-pinned tables it no longer does -- scratch/hotloop_spills.sh checks the ISA.)
#define STAMP(idx) do {} while (0)
#endif
#define CI_ST 18          // row stride (doubles) of a wave's transposition slab: 144 B keeps 16-byte alignment

template <int NW, int UPW>
struct CholInvSmem {
    union {                            // never live at the same time (barriers in cholinv_attempt separate the uses)
        double Bs[2][64][CI_KC + 2];   // history GEMM: staged chunk of L[C_k rows][kc .. kc+16)
        double Dk[64][66];             // diagonal phase: in D, out L_kk (lower)
        double St[NW][16 * UPW][CI_ST];   // panel store: per-wave (16 UPW) x 16 slab that turns row-contiguous
                                       // 16-byte global accesses into the transposed accumulator layout
#ifdef MEDGP_PRICE_NLML
        // PRICING BUILD ONLY (results meaningless, never shipped; DESIGN_LOG R6): X_k aliased onto D_k so that the structure fits 40 KB
        // -- what an nlml-only shape with packed triangular tiles would need -- to measure what 3 / 4 co-resident workgroups are worth
        // before building them
        double Xk[64][66];
        double zpart[NW][64];
    };
#else
    };
    union {
        double Xk[64][66];         // L_kk^-1 (lower, exact zeros above the diagonal)
        double zpart[NW][64];      // end of the history GEMM of pass 0 (the previous step's Xk is dead, the factor has not
                                   // written the new one yet): the waves' partial z products; wave 0 sums them BEFORE it factors
    };
#endif
    alignas(16) double rhs[64 + 128];          // diag(L_kk) during the factor (+ 128 doubles of diag16 scratch), then the z right-hand side
    double zk[64];                 // z of the current panel (for the in-step alpha accumulation)
#ifdef MEDGP_PRICE_NLML
    double zs[512];                // (pricing build: n <= 512 only)
#else
    double zs[1024];               // z history kept in LDS for n <= 1024: the in-loop z product then issues no
                                   // global loads, whose wait (vmcnt) would also drain the operand loads
#endif
    double red[16];
    double logdet;
    int fail;
#ifdef MEDGP_STAMPS
    unsigned long long stamps[NW][8];
#endif
};
static_assert(sizeof(CholInvSmem<4, 4>) <= 80 * 1024, "two 4-wave workgroups per CU need <= 80 KB each");
#ifdef MEDGP_PRICE_NLML
static_assert(sizeof(CholInvSmem<4, 2>) <= 40 * 1024 && sizeof(CholInvSmem<4, 3>) <= 53 * 1024, "pricing shapes: 4 / 3 workgroups per CU");
#endif

#define CI_S 66   // LDS row stride (doubles) of Dk / Xk
// v_mfma_f64_16x16x4_f64 takes its last immediate (BLGP) as NEG[a, b, c]: 1 = the product enters with the opposite sign, for free
// (-(a b) + c is the same correctly rounded FMA as (-a) b + c: results are bit-identical to an explicit v_xor of the operand).
// Every "-x" that used to feed an MFMA (48 v_xor per pass in the panel solve, 64 in the panel init, the trailing updates of the
// diagonal factor) is gone: next to the co-resident workgroup's MFMA stream each VALU instruction waits for the pipe.
#define MFMA_NEGA 1
#define CI_MFMA_LOOP(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, MFMA_NEGA)
#define CI_MFMA_TRSM(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ inline double readlane_d(double v, int srclane) {   // srclane must be wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// 16x16 tile product on one wave, operands in LDS (row stride CI_S): returns c + op(A) op(B),
// op(A)[i][k] = TA ? A[k][i] : A[i][k], op(B)[k][j] = TB ? B[j][k] : B[k][j].
template <bool TA, bool TB>
__device__ inline v4d tile_mm(const ld_t *Ap, const ld_t *Bp, v4d c, int li, int g) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
        double a = TA ? Ap[(4 * s + g) * CI_S + li] : Ap[li * CI_S + 4 * s + g];
        double b = TB ? Bp[li * CI_S + 4 * s + g] : Bp[(4 * s + g) * CI_S + li];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    return c;
}
__device__ inline v4d tile_ld(const ld_t *Cp, int li, int g) {
    v4d c;
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = Cp[(4 * r + g) * CI_S + li];
    return c;
}
__device__ inline void tile_st(ld_t *Cp, v4d c, int li, int g) {
#pragma unroll
    for (int r = 0; r < 4; r++) Cp[(4 * r + g) * CI_S + li] = c[r];
}

// Cholesky + inverse of one 16x16 diagonal tile, register resident, cross-lane traffic by v_readlane.
// Lanes 0..15 hold the rows of the tile, lanes 16..31 the rows of an appended identity: the right-looking column
// recurrence  row[j] *= 1/l_jj;  row[c] -= row[j] l_cj (c > j)  turns the tile rows into L and the identity rows into
// L^-T -- the inverse costs no extra instructions (the same trick k_cholinv plays with whole panels).  Lanes 32..63
// mirror lanes 0..31.  T: tile in Dk (in: D lower, out: L lower); X: tile in Xk (out: the full 16x16 inverse, exact
// zeros above the diagonal); dv[0..16) receives diag(L).  Returns false on a bad pivot (LAPACK potf2 rule: pivot <= 0
// or NaN); there is no early exit -- the 16 steps are ONE basic block, so the column updates of step j overlap the
// reciprocal-square-root chain of step j + 1 (a bad pivot just propagates NaNs that nobody uses).
// Instruction count is what bounds it (one wave alone issues a 64-bit VALU instruction every ~8 cycles; 8.0 k cycles per
// tile with all 15 + 14 + ... trailing updates fed by v_readlane pairs).  The columns are therefore taken four at a time:
// inside a 4-column panel the (critical-path) updates use v_readlane, the rows of the finished panel are then published to
// a 512-byte LDS scratch and the updates of the columns right of the panel read their four multipliers as broadcast
// ds_read_b128 pairs -- 60 LDS reads instead of 240 v_readlane per tile.  Every a[c] still receives the same FMAs in the
// same (ascending j) order: results are bit-identical to the all-readlane form.  scr: 128 doubles of LDS owned by this wave.
// FL (template): form of the initial loads.  false: `ident ? (c == i) : T[i][c]`, which hipcc turns into one exec-mask region per
// column; true: identity lanes read their one-hot row from a 31-entry strip Z = [0 x 15, 1, 0 x 15] in the wave's scratch (row i =
// Z[15 - i .. 30 - i]) and the tile lanes their row of T through ONE per-lane base address: 16 plain ds_read, no select, no branch.
// Measured (round 4, profiles/r04_diag16_variants.txt): true is 11 % faster in isolation and 2 % on the look-ahead chain, but makes
// k_cholinv 7-11 % slower in situ -- so the one-wave factor of k_cholinv instantiates false, the four-wave factor true.
template <bool FL = false>
__device__ __forceinline__ bool diag16(ld_t *T, ld_t *X, ld_t *dv, ld_t *scr, int lane) {
    // LDL^T order of operations: the column recurrence only needs the RECIPROCAL of each pivot (v_rcp_f64 + two Newton steps,
    // 5 instructions on the serial chain) -- column j stays unscaled, the multipliers are t = a[j] / d_j; the reciprocal
    // square roots of all 16 pivots are taken once at the end, lane-parallel, and every column is scaled by its own.
    // (Scaling each column as it is finished put v_rsq_f64 + two coupled Goldschmidt steps + a residual correction, 12
    //  instructions, on the chain of every column.)
    const int i = lane & 15, i5 = lane & 31;
    const bool ident = i5 >= 16;
    double a[16];
    if constexpr (FL) {
        // (Z sits in the half of the scratch that the panel loop first writes at panel 1; LDS operations of a wave complete in order)
        ld_t *Z = scr + 64;
        if (lane < 31) Z[lane] = (lane == 15) ? 1.0 : 0.0;
        __builtin_amdgcn_wave_barrier();
        const ld_t *src = ident ? (const ld_t *)(Z + 15 - i) : (const ld_t *)(T + i * CI_S);
#pragma unroll
        for (int c = 0; c < 16; c++) a[c] = src[c];
    } else {
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const double t = T[i * CI_S + c];
            a[c] = ident ? ((c == i) ? 1.0 : 0.0) : t;
        }
    }
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        double tm[4];   // multipliers of the panel's columns: lane c holds a[c][j] / d_j
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 4 * p + k;
            const double piv = readlane_d(a[j], j);
            ok = ok && (piv > 0.0);
            double r = __builtin_amdgcn_rcp(piv);
            r = fma(fma(-piv, r, 1.0), r, r);
            r = fma(fma(-piv, r, 1.0), r, r);
            tm[k] = a[j] * r;
#pragma unroll
            for (int k2 = k + 1; k2 < 4; k2++) a[4 * p + k2] -= a[j] * readlane_d(tm[k], 4 * p + k2);
        }
        if (p < 3) {
            ld_t *sp = scr + 64 * (p & 1);
            if (lane < 16) {
                *(lv2d_t *)&sp[4 * lane] = (v2d){tm[0], tm[1]};
                *(lv2d_t *)&sp[4 * lane + 2] = (v2d){tm[2], tm[3]};
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 4 * p + 4; c < 16; c++) {
                const v2d l01 = *(const lv2d_t *)&sp[4 * c], l23 = *(const lv2d_t *)&sp[4 * c + 2];
                a[c] -= a[4 * p] * l01[0];
                a[c] -= a[4 * p + 1] * l01[1];
                a[c] -= a[4 * p + 2] * l23[0];
                a[c] -= a[4 * p + 3] * l23[1];
            }
        }
    }
    // pivots: lane j (< 16) holds d_j in a[j]
    double dj = 1.0;
#pragma unroll
    for (int c = 0; c < 16; c++) dj = (i5 == c) ? a[c] : dj;
    // sqrt and reciprocal sqrt of all pivots at once: v_rsq_f64 seed + two coupled Goldschmidt steps + one residual correction
    const double y0 = __builtin_amdgcn_rsq(dj);
    double gg = dj * y0, hh = 0.5 * y0;
    double rr = fma(-gg, hh, 0.5);
    gg = fma(gg, rr, gg); hh = fma(hh, rr, hh);
    rr = fma(-gg, hh, 0.5);
    gg = fma(gg, rr, gg); hh = fma(hh, rr, hh);
    const double dd = fma(-gg, gg, dj);
    const double sq = fma(dd, hh, gg), rinv = hh + hh;   // sqrt(d_j), 1 / sqrt(d_j) in lane j
#pragma unroll
    for (int c = 0; c < 16; c++) a[c] *= readlane_d(rinv, c);
    if (lane < 16) {
        dv[lane] = sq;
        // whole rows, 16 bytes at a time (round 4): the part right of the diagonal (c > i) receives values nobody reads -- every
        // consumer of a diagonal tile of L takes its lower triangle only (exports: `cc <= rr`; the tile products read off-diagonal
        // tiles) -- where a store per column under `c <= i` cost one exec-mask region each.  Same bits; look-ahead chain -2 %,
        // k_cholinv -0 .. 1.5 % (profiles/r04_diag16_variants.txt).
#pragma unroll
        for (int m = 0; m < 8; m++) *(lv2d_t *)&T[i * CI_S + 2 * m] = (v2d){a[2 * m], a[2 * m + 1]};
    } else if (lane < 32) {   // identity rows: a[c] = (L^-T)[i][c] = (L^-1)[c][i], exactly zero for c < i
#pragma unroll
        for (int c = 0; c < 16; c++) X[c * CI_S + i] = a[c];
    }
    return ok;
}

#if defined(MEDGP_STAMPS) && !defined(MEDGP_NO_DSTAMPS)
#define MEDGP_DSTAMPS
#endif
#ifdef MEDGP_DSTAMPS
__device__ unsigned long long g_diag_dbg[8];
#define DSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) atomicAdd(&g_diag_dbg[k], t_ - dlast); dlast = t_; } while (0)
#else
#define DSTAMP(k) do {} while (0)
#endif
// Cholesky of the 64x64 block D (lower, in place) and its inverse into X (lower), on ONE wave: 16x16 tiles, diagonal
// tiles in registers (diag16), everything else as MFMA tile products.  D, X: row stride CI_S; dv: 64 doubles of scratch
// (diag of L); *fail is set on a bad pivot; *logdet += sum log diag (one lane-parallel log + a fixed butterfly).
// Register-level operand reuse (v_mfma_f64_16x16x4_f64: A lane (i = l & 15, k = l >> 4), B lane (k = l >> 4, j = l & 15),
// C/D lane (col = l & 15, row = (l >> 4) + 4 reg)):  a C-layout tile c, used register by register as the operand of
// k-step r, is the matrix c^T as an A operand and the matrix c itself as a B operand.  The panel tiles are therefore
// computed TRANSPOSED (lt = X(t,t) D(s,t)^T = L(s,t)^T): lt is at once the A operand L(s,t) and the B operand L(u,t)^T of
// the trailing update, and the partial sums P of the inverse feed X(s,s) P straight from their accumulators -- no LDS
// round trips between dependent products, and all LDS reads of a stage are issued before its MFMAs.
__device__ __forceinline__ void diag_factor_wave_body(ld_t *D, ld_t *X, ld_t *dv, li_t *fail, ld_t *logdet, int lane) {
    const int li = lane & 15, g = lane >> 4;
#define TD(s, t) (D + (16 * (s)) * CI_S + 16 * (t))
#define TX(s, t) (X + (16 * (s)) * CI_S + 16 * (t))
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
#ifdef MEDGP_DSTAMPS
    unsigned long long dlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast)::"memory");
#endif
#pragma unroll 1
    for (int t = 0; t < 4; t++) {
        if (!diag16(TD(t, t), TX(t, t), dv + 16 * t, dv + 64, lane)) { if (lane == 0) *fail = 1; return; }
        __builtin_amdgcn_wave_barrier();
        DSTAMP(0);
        if (t == 3) break;
        // ---- panel below, transposed:  lt[si] = X(t,t) D(s,t)^T = L(s,t)^T,  s = t + 1 + si
        double xa[4];
#pragma unroll
        for (int k = 0; k < 4; k++) xa[k] = TX(t, t)[li * CI_S + 4 * k + g];
        double db[3][4];
#pragma unroll
        for (int si = 0; si < 3; si++)
            if (t + 1 + si < 4) {
#pragma unroll
                for (int k = 0; k < 4; k++) db[si][k] = TD(t + 1 + si, t)[li * CI_S + 4 * k + g];
            }
        // trailing tiles D(s,u), t < u <= s, fetched while the panel products run
        v4d dc[3][3];
#pragma unroll
        for (int si = 0; si < 3; si++)
#pragma unroll
            for (int ui = 0; ui <= si; ui++)
                if (t + 1 + si < 4) dc[si][ui] = tile_ld(TD(t + 1 + si, t + 1 + ui), li, g);
        v4d lt[3];
#pragma unroll
        for (int si = 0; si < 3; si++) lt[si] = zero4;
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int si = 0; si < 3; si++)
                if (t + 1 + si < 4) lt[si] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[k], db[si][k], lt[si], 0, 0, 0);
#pragma unroll
        for (int si = 0; si < 3; si++)
            if (t + 1 + si < 4) {
#pragma unroll
                for (int r = 0; r < 4; r++) TD(t + 1 + si, t)[li * CI_S + 4 * r + g] = lt[si][r];   // L(s,t), untransposed
            }
        DSTAMP(1);
        // ---- trailing update  D(s,u) -= L(s,t) L(u,t)^T  straight from the lt registers
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int si = 0; si < 3; si++)
#pragma unroll
                for (int ui = 0; ui <= si; ui++)
                    if (t + 1 + si < 4) dc[si][ui] = __builtin_amdgcn_mfma_f64_16x16x4f64(lt[si][r], lt[ui][r], dc[si][ui], 0, 0, MFMA_NEGA);
#pragma unroll
        for (int si = 0; si < 3; si++)
#pragma unroll
            for (int ui = 0; ui <= si; ui++)
                if (t + 1 + si < 4) tile_st(TD(t + 1 + si, t + 1 + ui), dc[si][ui], li, g);
        __builtin_amdgcn_wave_barrier();
        DSTAMP(2);
    }
    // ---- inverse, block row s:  X(s,t) = -X(s,s) P,  P = sum_{u=t}^{s-1} L(s,u) X(u,t)  (chains t < s are independent)
#pragma unroll
    for (int s2 = 1; s2 < 4; s2++) {
        v4d pp[3];
#pragma unroll
        for (int t = 0; t < s2; t++) pp[t] = zero4;
#pragma unroll
        for (int u = 0; u < s2; u++) {
            double la[4];
#pragma unroll
            for (int k = 0; k < 4; k++) la[k] = TD(s2, u)[li * CI_S + 4 * k + g];
#pragma unroll
            for (int t = 0; t <= u; t++) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    pp[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[k], TX(u, t)[(4 * k + g) * CI_S + li], pp[t], 0, 0, 0);
            }
        }
        double xs[4];
#pragma unroll
        for (int r = 0; r < 4; r++) xs[r] = TX(s2, s2)[li * CI_S + 4 * r + g];
        v4d xo[3];
#pragma unroll
        for (int t = 0; t < s2; t++) xo[t] = zero4;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int t = 0; t < s2; t++) xo[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[r], pp[t][r], xo[t], 0, 0, MFMA_NEGA);
#pragma unroll
        for (int t = 0; t < s2; t++) tile_st(TX(s2, t), xo[t], li, g);
        __builtin_amdgcn_wave_barrier();
    }
    DSTAMP(3);
    // zero the strictly-upper tiles of X (U_kk is exported from X with zeros)
#pragma unroll
    for (int t = 1; t < 4; t++)
#pragma unroll
        for (int s2 = 0; s2 < t; s2++) tile_st(TX(s2, t), zero4, li, g);
    __builtin_amdgcn_wave_barrier();
    double lg = log(dv[lane]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_xor(lg, off);
    if (lane == 0) *logdet += lg;
    DSTAMP(4);
#ifdef MEDGP_DSTAMPS
    if (lane == 0) atomicAdd(&g_diag_dbg[7], 1ull);
#endif
#undef TD
#undef TX
}
// Out of line: bounds the register pressure around the call.  (A 16-wave shape -- 8 waves x 2 units, 128 VGPRs, four waves
// per SIMD -- was measured at 2.78 ms against 1.66 ms for 4 waves x 4 units at 512 x N=512: the callee does not inherit a
// smaller register budget, and inlined into 128 VGPRs the kernel spills in every phase.)
__device__ __attribute__((noinline)) void diag_factor_wave(ld_t *D, ld_t *X, ld_t *dv, li_t *fail, ld_t *logdet, int lane) {
    diag_factor_wave_body(D, X, dv, fail, logdet, lane);
}

// Cholesky of the 64x64 block D and its inverse X on FOUR waves (all waves of the workgroup call it; waves >= 4 only take part
// in the barriers).  Same tile algebra as diag_factor_wave_body; the serial chain is the four register-resident 16x16
// factorisations (diag16) on wave 0, everything else runs beside it:
//   phase A(t): panel tiles L(s,t) = D(s,t) X(t,t)^T, one per wave                                     | barrier
//   phase B(t): wave 0: trailing tile (t+1,t+1), then diag16(t+1) right away (its own data, no barrier);
//               waves 1-3: the other trailing tiles D(s,u) -= L(s,t) L(u,t)^T, then the finished rows of the inverse
//               X(t, 0..t-1) = -X(t,t) sum_u L(t,u) X(u,.)  (needs only diagonal inverses <= t)       | barrier
//   tail      : X(3, 0..2) on three waves, zero tiles, log det.
// Critical path: 4 diag16 + 3 (panel + one trailing tile + 2 barriers) + one inverse row.
__device__ __forceinline__ void inv_row_tiles(ld_t *D, ld_t *X, int s2, int t, int li, int g) {
    // X(s2,t) = -X(s2,s2) P,  P = sum_{u=t}^{s2-1} L(s2,u) X(u,t)
#define TD(s, t) (D + (16 * (s)) * CI_S + 16 * (t))
#define TX(s, t) (X + (16 * (s)) * CI_S + 16 * (t))
    v4d pp = {0.0, 0.0, 0.0, 0.0};
    for (int u = t; u < s2; u++) pp = tile_mm<false, false>(TD(s2, u), TX(u, t), pp, li, g);
    v4d xo = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; r++) xo = __builtin_amdgcn_mfma_f64_16x16x4f64(TX(s2, s2)[li * CI_S + 4 * r + g], pp[r], xo, 0, 0, MFMA_NEGA);
    tile_st(TX(s2, t), xo, li, g);
}
__device__ __forceinline__ void trail_tile(ld_t *D, int s2, int u, int t, int li, int g) {
    v4d c = tile_ld(TD(s2, u), li, g);
#pragma unroll
    for (int k = 0; k < 4; k++)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(TD(s2, t)[li * CI_S + 4 * k + g], TD(u, t)[li * CI_S + 4 * k + g], c, 0, 0, MFMA_NEGA);
    tile_st(TD(s2, u), c, li, g);
}
#ifndef LA_FASTLOADS
#define LA_FASTLOADS true   // load form of diag16 in the four-wave factor (see diag16)
#endif
// (inlined into its callers: as an out-of-line function it claimed 248 VGPRs + 32 AGPRs for callee-saved traffic, and a kernel is
//  allocated the maximum over its call graph -- k_la_step lost its second workgroup per CU to a callee it runs in one role)
#define DFW_SETFAIL() do { if (lane == 0) *fail = 1; } while (0)
// x00 != nullptr (a compile-time property of the call site): wave 0 of the CALLER has already run diag16 on tile (0,0) -- L(0,0) is in
// place in D, dv[0..16) is set, a bad pivot is flagged -- and left the tile's inverse in the 16 x 16 scratch tile x00 (row stride
// CI_S) because X's own tile was still being read by the other waves; it is copied into place here.  (The look-ahead chain starts
// the first tile's factorisation as soon as its own 16 rows of the rank-64 update are done, kernels_cholinv_la.h step (5).)
#ifdef LA_FSTAMPS   // diagnostic build: cumulative s_memtime stamps of wave 0 inside the four-wave factor (scratch/la_stamps.py)
#define FST(i) do { if (fst && wave == 0 && lane == 0) fst[i] = __builtin_amdgcn_s_memtime() - fst_t0; } while (0)
__device__ __forceinline__ void diag_factor_wg(ld_t *D, ld_t *X, ld_t *dv, li_t *fail, ld_t *logdet, int wave, int lane, const ld_t *x00 = nullptr, unsigned long long *fst = nullptr) {
    const unsigned long long fst_t0 = __builtin_amdgcn_s_memtime();
#else
#define FST(i) do {} while (0)
__device__ __forceinline__ void diag_factor_wg(ld_t *D, ld_t *X, ld_t *dv, li_t *fail, ld_t *logdet, int wave, int lane, const ld_t *x00 = nullptr) {
#endif
    const int li = lane & 15, g = lane >> 4;
    const v4d zero4 = {0.0, 0.0, 0.0, 0.0};
    if (wave == 0) {
        if (x00) tile_st(TX(0, 0), tile_ld(x00, li, g), li, g);
        else if (!diag16<LA_FASTLOADS>(TD(0, 0), TX(0, 0), dv, dv + 64, lane)) { DFW_SETFAIL(); }
    }
    FST(0);
    __syncthreads();
    if (*fail) return;
#pragma unroll 1
    for (int t = 0; t < 3; t++) {
        // ---- phase A: panel below the diagonal tile
        if (wave < 3 - t) {
            const int s2 = t + 1 + wave;
            v4d lt = tile_mm<false, true>(TD(s2, t), TX(t, t), zero4, li, g);   // D(s,t) X(t,t)^T
            tile_st(TD(s2, t), lt, li, g);
        }
        __syncthreads();
        FST(1 + 3 * t);
        // ---- phase B
        if (wave == 0) {
            trail_tile(D, t + 1, t + 1, t, li, g);
            __builtin_amdgcn_wave_barrier();
            FST(2 + 3 * t);
            if (!diag16<LA_FASTLOADS>(TD(t + 1, t + 1), TX(t + 1, t + 1), dv + 16 * (t + 1), dv + 64, lane)) { DFW_SETFAIL(); }
            FST(3 + 3 * t);
        } else if (wave < 4) {
            // remaining trailing tiles (s,u), t+1 <= u <= s <= 3, (s,u) != (t+1,t+1): dealt round-robin to waves 1..3
            int e = 0;
            for (int s2 = t + 1; s2 < 4; s2++)
                for (int u = t + 1; u <= s2; u++) {
                    if (s2 == t + 1 && u == t + 1) continue;
                    if (e % 3 == wave - 1) trail_tile(D, s2, u, t, li, g);
                    e++;
                }
            // finished inverse row t (t >= 1): tiles X(t, 0..t-1)
            if (t >= 1 && wave - 1 < t) inv_row_tiles(D, X, t, wave - 1, li, g);
        }
        __syncthreads();
        if (*fail) return;
    }
    FST(10);
    // ---- tail: inverse rows 3 (and row 2 was done in phase B(2))
    if (wave < 3) inv_row_tiles(D, X, 3, wave, li, g);
    else
    if (wave == 3) {
#pragma unroll
        for (int t = 1; t < 4; t++)
            for (int s2 = 0; s2 < t; s2++) tile_st(TX(s2, t), zero4, li, g);   // strictly-upper tiles of X are zero
        double lg = log(dv[lane]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lg += __shfl_xor(lg, off);
        if (lane == 0) *logdet += lg;
    }
    __syncthreads();
    FST(11);
#undef TD
#undef TX
}

// ---- one factorisation attempt; returns false if a pivot failed ---------------------------------------------------
// Split of a step's row blocks over the waves: a pass covers NW block slots (M blocks first, then the U blocks by
// ascending row block) and wave w owns the 16-row unit (w & 3) of FOUR slots, so all waves carry the same history
// profile (the U block of row block rho only has history from column 64 rho on): the per-chunk barrier never makes a
// short-history wave wait for a long-history one (whole-block ownership: 75 % balance at N = 512), and chunks before the
// pass's first non-zero history column are skipped for the whole workgroup.  The active units of a wave at chunk c
// are a prefix of its slots, so the chunk body is instantiated per prefix length.
// One phase of the chunk loop: chunks [lo, hi) during which exactly the first NA units of this wave are active.
// One phase of the chunk loop: chunks [lo, hi) during which exactly the first NA units of this wave are active.
// Software pipeline (CI_KC = 16 columns per chunk = two 8-column halves):
//   * the history operand of the NEXT chunk is requested half by half as soon as the MFMAs that read the registers of that
//     half have been issued -- hc[u][0] right after half 0 (the 32 MFMAs of half 1 cover the request), hc[u][1] after half 1
//     (the staging store, the barrier and the next chunk's half 0 cover it).  No extra registers: the loads go back into
//     the registers their half has just released.  (Requested at the top of the chunk that consumes them, as before, the
//     first MFMA of every chunk waited for a full memory round trip.)
//   * the shared operand is read from LDS one (half, column tile) group ahead of the MFMAs that use it, instead of directly in
//     front of them (eight exposed LDS latencies per chunk).
// Which units the next chunk needs is a wave-uniform count (CI_NACT): inside a phase it is NA, at the phase's last chunk the
// units that become active are requested as well.
#define CI_NACT(x) (((x) >= pb[0]) + ((x) >= pb[1]) + ((x) >= pb[2]) + ((x) >= pb[3]))
// buffer_load with the matrix descriptor in SGPRs, a 32-bit lane offset and a scalar (unit row + chunk) offset: no 64-bit
// per-lane address arithmetic and no address registers held across the loop (T8 of the guide)
#define CI_HLOAD(u, h, cc) hc[u][h] = ci_bload(rsu[u], voffh + 64 * (h), urow[u] + (cc) * (CI_KC * 8))
// The loop body is straight-line code (the waitcnt pass counts outstanding loads exactly only without branches around
// them): the requests for chunk c + 1 are unconditional, in the last chunk they re-read that chunk (index clamped) and
// the result is dropped.  Units that become active at a phase boundary are requested by CI_PHASE_ENTER between the loops.
#define CI_PHASE_ENTER(NA0, NA1, cc)                                                                               \
    if ((cc) < nch) {                                                                                              \
        _Pragma("unroll") for (int u = NA0; u < UPW; u++)                                                          \
            if (u < (NA1)) { CI_HLOAD(u, 0, (cc)); CI_HLOAD(u, 1, (cc)); }                                         \
    }
// Staging of the shared operand: chunk c + 1 is requested at the top of iteration c and stored into LDS at its end.
#define CI_STAGE_LOAD(c, cn) { _Pragma("unroll") for (int e = 0; e < SPT; e++) bnext[e] = Bsrc[(cn) * CI_KC + e]; }
#define CI_CHUNK_PHASE(NA, lo, hi)                                                                                 \
    for (int c = (lo); c < (hi); c++) {                                                                            \
        const int buf = c & 1;                                                                                     \
        const int cn = (c + 1 < nch) ? c + 1 : c;                 /* next chunk, clamped */                        \
        CI_STAGE_LOAD(c, cn)                                                                                      \
        if (NA > 0) {                                                                                              \
            v2d a_cur = *(const v2d *)&sm.Bs[buf][li][2 * g];                                                      \
            _Pragma("unroll") for (int hct = 0; hct < 8; hct++) {                                                  \
                const int h = hct >> 2, ct = hct & 3;                                                              \
                v2d a_nxt = a_cur;                                                                                 \
                if (hct < 7) a_nxt = *(const v2d *)&sm.Bs[buf][16 * ((hct + 1) & 3) + li][8 * ((hct + 1) >> 2) + 2 * g]; \
                _Pragma("unroll") for (int s = 0; s < 2; s++)                                                      \
                    _Pragma("unroll") for (int u = 0; u < NA; u++)                                                 \
                        acc[ct][u] = CI_MFMA_LOOP(a_cur[s], hc[u][h][s], acc[ct][u]); \
                a_cur = a_nxt;                                                                                     \
                if (ct == 3) {   /* half h is consumed: request it for the next chunk */                           \
                    _Pragma("unroll") for (int u = 0; u < NA; u++) { CI_HLOAD(u, h, cn); }                         \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        /* z history product (pass 0 only): lane = panel column, every wave takes CI_KC / NW of the chunk's columns (left to */ \
        /* one wave, its 16 + 16 operands in flight made hipcc spill two accumulator tiles inside this loop)            */ \
        if (pass == 0) {                                                                                           \
            if (npad <= 1024) {   /* z history in LDS: no vector-memory wait in this path */                       \
                _Pragma("unroll") for (int kq = 0; kq < CI_KC / NW; kq++)                                          \
                    zsum += sm.Bs[buf][lane][(CI_KC / NW) * wave + kq] * sm.zs[c * CI_KC + (CI_KC / NW) * wave + kq]; \
            } else {                                                                                               \
                _Pragma("unroll") for (int kq = 0; kq < CI_KC / NW; kq++)                                          \
                    zsum += sm.Bs[buf][lane][(CI_KC / NW) * wave + kq] * zz[c * CI_KC + (CI_KC / NW) * wave + kq]; \
            }                                                                                                      \
        }                                                                                                          \
        STAMP(4); /* MFMA block (+ z product) */                                                                   \
        if (c + 1 < nch) {                                                                                         \
            _Pragma("unroll") for (int e = 0; e < SPT; e++) sm.Bs[buf ^ 1][srow][scol + e] = bnext[e];             \
        }                                                                                                          \
        STAMP(6); /* staging store (waits for this iteration's B loads) */                                         \
        __syncthreads();                                                                                           \
        STAMP(7); /* chunk barrier wait */                                                                         \
    }

template <int NW, int UPW>
__device__ bool cholinv_attempt(const MedgpDev &L, int b, int slot, int n, int want_mode, CholInvSmem<NW, UPW> &sm) {
    // want_mode bit 0: U rows + alpha (full inverse); bit 1: only the diagonal blocks U_kk = L_kk^-T are stored (what the
    // predictive solve k_predict needs)
    const int want_inv = want_mode & 1;
    const bool store_ukk = want_mode != 0;
    constexpr int NT = NW * 64;
    constexpr int G = NW / 4;            // wave groups
    constexpr int BPP = G * UPW;         // 64-row block slots per pass (UPW 16-row units per wave)
    static_assert(NW % 4 == 0, "waves come in groups of four 16-row units");
    const int ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    gd_t *Lb = (gd_t *)(L.Kmat + (size_t)b * ld * ld);
    gd_t *Ub = (gd_t *)(L.Linv + (size_t)b * ld * ld);
    gd_t *zz = (gd_t *)(L.z + (size_t)b * ld);
    gd_t *alpha = (gd_t *)(L.alpha + (size_t)b * ld);
    const gd_t *y = (const gd_t *)(L.py + (size_t)slot * L.pld);
    const int tid = threadIdx.x, lane = tid & 63;
    const int hwave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> scalar control flow + scalar bases
    const int wave = (b & 1) ? (NW - 1 - hwave) : hwave;   // mirrored on odd entries: the diagonal-factor waves of two
                                                           // co-resident workgroups sit on different SIMDs
    const int li = lane & 15, g = lane >> 4;
    const int wu = wave & 3, wg = wave >> 2;
    const int loff = li * ld + 2 * g;                      // lane part of a history-operand address
    const int voffh = loff * 8;                            // the same in bytes (buffer_load voffset)
    // descriptors of this patient's two matrices, built from wave-uniform values only
    const int mat_bytes = __builtin_amdgcn_readfirstlane((int)((size_t)ld * ld * sizeof(double)));
    // row-contiguous access of a 64 x 16 column slab: one instruction = 8 rows x 128 B (whole cache lines); the 8-byte
    // accesses of the transposed accumulator layout touch 64 lines per instruction and made panel init + panel store
    // tag-lookup bound in the L1 (a fifth of the kernel).  The slab is transposed through LDS instead.
    // (its lane constants are re-derived from an opaque copy of the lane id at each use: hoisted out of the pass loop
    //  they would stay live across the MFMA loop and push accumulator tiles into scratch)
#define CI_SLAB_LANE()                                   \
    int lane_o = lane;                                   \
    asm volatile("" : "+v"(lane_o));                     \
    const int srow8 = lane_o >> 3, spc = lane_o & 7;     \
    double (*S)[CI_ST] = sm.St[hwave]

    if (tid == 0) { sm.fail = 0; sm.logdet = 0.0; }
    __syncthreads();
#ifdef MEDGP_STAMPS
    unsigned long long st_last;
    if (lane < 8) sm.stamps[tid >> 6][lane] = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

    for (int k = 0; k < nb; k++) {
        const int c0 = 64 * k;                       // first column of the panel
        const int nM = nb - k;                       // K-row blocks (first one is the diagonal block)
        const int ntot = nM + (want_inv ? k : 0);    // + U-row blocks
        const int npass = (ntot + BPP - 1) / BPP;
        const int nch = c0 / CI_KC;                  // history chunks
        for (int pass = 0; pass < npass; pass++) {
            // my four slots (wave-uniform scalars)
            bool act[UPW], isM[UPW];
            int rowb[UPW], cf[UPW];
            const gd_t *ub[UPW];
            int urow[UPW];   // byte offset of the unit's first row inside its matrix
            __amdgpu_buffer_rsrc_t rsu[UPW];   // descriptor of the unit's matrix (L or U of this patient), provably wave-uniform
#pragma unroll
            for (int u = 0; u < UPW; u++) {
                const int bidx = pass * BPP + u * G + wg;
                act[u] = bidx < ntot;
                isM[u] = bidx < nM;
                const int rblk = isM[u] ? (k + bidx) : (bidx - nM);
                rowb[u] = 64 * rblk + 16 * wu;
                // first chunk with a non-zero history operand: a U block starts at its own diagonal block U_rho,rho, which is upper
                // triangular -- the rows of unit wu are zero left of column 16 wu (the chunks before it are skipped, 5 % of the
                // kernel's MFMAs)
                cf[u] = act[u] ? (isM[u] ? 0 : (64 * rblk + 16 * wu) / CI_KC) : (1 << 30);
                ub[u] = (isM[u] ? Lb : Ub) + (size_t)rowb[u] * ld;
                urow[u] = __builtin_amdgcn_readfirstlane(rowb[u] * ld * 8);
                {
                    const unsigned long long pb64 = (unsigned long long)(isM[u] ? L.Kmat : L.Linv) + (unsigned long long)b * (unsigned long long)mat_bytes;
                    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)pb64), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(pb64 >> 32));
                    rsu[u] = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi32 << 32) | lo32), 0, mat_bytes, 0x00020000);
                }
            }
            // first chunk anybody in the workgroup needs: slot 0 of group 0 has the longest history
            const int bidx0 = pass * BPP;
            const int cstart = (bidx0 < nM) ? 0 : (64 * (bidx0 - nM)) / CI_KC;
            // acc starts at init (init = K block for K rows, 0 for U rows); the history GEMM subtracts P (MFMA negation modifier),
            // so acc is the panel value  init - P  itself
            // Panel init through the same per-wave slab as the panel stores: row-contiguous 16-byte loads (8 whole lines per
            // instruction), transposition in LDS.  The loads must be UNCONDITIONAL inside the block (units without a K
            // block read slot 0's and discard it): with a branch per unit hipcc spilled four accumulator tiles inside the
            // MFMA loop (2.2 ms); branch-free the loop is spill free and the kernel gains 3 % over 8-byte loads of the
            // transposed layout (1.69 -> 1.64 ms at 512 x N=512).
            v4d acc[4][UPW];
            {
                CI_SLAB_LANE();
                bool ldu[UPW], anyld = false;
#pragma unroll
                for (int u = 0; u < UPW; u++) { ldu[u] = act[u] && isM[u]; anyld = anyld || ldu[u]; }
                if (anyld) {   // branch-free inside: units without a K block read the block of slot 0 and discard it
                    // all 64 columns are requested at once (the registers the accumulators will occupy are free here): one
                    // memory round trip per pass instead of one per 32-column half
                    v2d kin[4][2 * UPW];
                    double fsel[UPW];
#pragma unroll
                    for (int u = 0; u < UPW; u++) fsel[u] = ldu[u] ? 1.0 : 0.0;
#pragma unroll
                    for (int u = 0; u < UPW; u++) {
                        const gd_t *src = (ldu[u] ? ub[u] : ub[0]) + (size_t)srow8 * ld + c0 + 2 * spc;
#pragma unroll
                        for (int ct = 0; ct < 4; ct++)
#pragma unroll
                            for (int j = 0; j < 2; j++) kin[ct][2 * u + j] = *(const gv2d_t *)(src + (size_t)(8 * j) * ld + 16 * ct);
                    }
#pragma unroll
                    for (int ct = 0; ct < 4; ct++) {
#pragma unroll
                        for (int u = 0; u < UPW; u++)
#pragma unroll
                            for (int j = 0; j < 2; j++) *(v2d *)&S[16 * u + 8 * j + srow8][2 * spc] = kin[ct][2 * u + j];
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int u = 0; u < UPW; u++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const double kv = S[16 * u + (lane_o & 15)][4 * r + (lane_o >> 4)];
                                acc[ct][u][r] = kv * fsel[u];   // one multiply by 1 / 0 (two v_cndmask per double as a select)
                            }
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {
#pragma unroll
                    for (int ct = 0; ct < 4; ct++)
#pragma unroll
                        for (int u = 0; u < UPW; u++) acc[ct][u] = (v4d){0.0, 0.0, 0.0, 0.0};
                }
            }
            STAMP(0);   // panel init
            double zsum = 0.0;
            // ---- GEMM over the history, B chunk (L[C_k rows]) staged through LDS, double buffered
            if (nch > cstart) {
                // staging: 64 x CI_KC doubles per chunk, NT threads -> 64 CI_KC / NT elements each
                constexpr int SPT = 64 * CI_KC / NT;
                const int srow = (tid * SPT) / CI_KC, scol = (tid * SPT) % CI_KC;
                const gd_t *Bsrc = Lb + (size_t)(c0 + srow) * ld + scol;
                double bnext[SPT];
#pragma unroll
                for (int e = 0; e < SPT; e++) bnext[e] = Bsrc[cstart * CI_KC + e];
                __syncthreads();   // every wave is done with its init slab (St aliases Bs)
#pragma unroll
                for (int e = 0; e < SPT; e++) sm.Bs[cstart & 1][srow][scol + e] = bnext[e];
                __syncthreads();
                // The history operand of a chunk is loaded and consumed in the same iteration: NO software prefetch.
                // Measured: hipcc turns every use of a prefetched register into `s_waitcnt vmcnt(0)`, so a register
                // prefetch ring hid nothing and its 32-64 extra VGPRs pushed accumulator tiles into scratch inside the
                // loop; without it the loop is spill free and the other workgroup on the CU covers the latency.
                // The active set only grows with c (cf ascending), so the loop runs as up to five phases with a fixed
                // body each (one loop with a switch on the prefix length made hipcc shuffle and spill the accumulators).
                auto clampc = [&](int v) { return v < cstart ? cstart : (v > nch ? nch : v); };
                int pb[5];
#pragma unroll
                for (int u = 0; u < 5; u++) pb[u] = (u < UPW) ? clampc(cf[u < UPW ? u : 0]) : nch;
                static_assert(UPW <= 4 && CI_KC == 16, "the pipelined chunk loop is written for 16-column chunks and <= 4 units");
                // history operand registers: CI_PHASE_ENTER requests a unit's first chunk, the loop every later one half a chunk ahead
                v2d hc[UPW][2];
#pragma unroll
                for (int u = 0; u < UPW; u++) { hc[u][0] = (v2d){0.0, 0.0}; hc[u][1] = (v2d){0.0, 0.0}; }
                CI_CHUNK_PHASE(0, cstart, pb[0])
                CI_PHASE_ENTER(0, CI_NACT(pb[0]), pb[0])
                CI_CHUNK_PHASE(1, pb[0], pb[1])
                if constexpr (UPW >= 2) { CI_PHASE_ENTER(1, CI_NACT(pb[1]), pb[1]) CI_CHUNK_PHASE(2, pb[1], pb[2]) }
                if constexpr (UPW >= 3) { CI_PHASE_ENTER(2, CI_NACT(pb[2]), pb[2]) CI_CHUNK_PHASE(3, pb[2], pb[3]) }
                if constexpr (UPW >= 4) { CI_PHASE_ENTER(3, CI_NACT(pb[3]), pb[3]) CI_CHUNK_PHASE(4, pb[3], pb[4]) }
            } else {
                __syncthreads();   // init slabs done before Dk (same LDS) is written below
            }
            STAMP(1);
            if (pass == 0) {
                // diagonal block = slot 0 of group 0: every wave of the group holds 16 of its rows
                if (wg == 0) {
#pragma unroll
                    for (int ct = 0; ct < 4; ct++)
#pragma unroll
                        for (int r = 0; r < 4; r++) sm.Dk[16 * wu + li][16 * ct + 4 * r + g] = acc[ct][0][r];
                }
                sm.zpart[wave][lane] = zsum;
                __syncthreads();
                STAMP(0);   // wait for the slowest GEMM wave (folded into 0)
                // one wave factors the block while the co-resident workgroup's waves own the SIMDs (measured at 512 x N=512, two
                // workgroups per CU: one-wave factor 1.52 ms, the four-wave diag_factor_wg 1.62 ms -- its barriers and the
                // register traffic around the out-of-line call in all four waves cost more than its shorter critical path
                // gains; one workgroup per CU: no difference).  The look-ahead multi-CU schedule, whose diagonal chain IS the
                // critical path, uses diag_factor_wg.
                double zacc = 0.0;   // fixed order over the waves' partial products (zpart shares its LDS with Xk: read first)
                if (wave == 0) {
#pragma unroll
                    for (int w2 = 0; w2 < NW; w2++) zacc += sm.zpart[w2][lane];
                }
                // (k_cholinv<8,2> with the factor inlined and 128 VGPRs -- two 8-wave workgroups per CU, four waves per SIMD --
                //  was tried: alone it matches <8,4> (0.81 ms at 256 patients), but at 128 VGPRs hipcc spills inside the chunk
                //  loops and 376 scratch operations into the factor: 2.30 ms at 512 patients against 1.38 ms for <4,4>)
                if (wave == 0) diag_factor_wave((ld_t *)&sm.Dk[0][0], (ld_t *)&sm.Xk[0][0], (ld_t *)sm.rhs, (li_t *)&sm.fail, (ld_t *)&sm.logdet, lane);
                __syncthreads();   // factor done: Dk = L_kk, Xk = L_kk^-1 (or fail)
                STAMP(3);   // diagonal factor
#ifndef MEDGP_PRICE_NLML
                if (sm.fail) return false;
#endif
                if (wave == 0) {
                    // z_k = L_kk^-1 (y_k - zacc).  Xk has exact zeros above the diagonal, so fixed-length loops add the same terms
                    // as triangular ones, and the 64 terms are split over four independent accumulators (columns c, c+16, c+32,
                    // c+48): the dependent FMA chain is 16 long instead of 64 -- next to a co-resident MFMA stream every link of
                    // that chain waits for the pipe (11 k cycles per panel in situ for the two solves with one accumulator each)
                    sm.rhs[lane] = ((c0 + lane < n) ? y[c0 + lane] : 0.0) - zacc;
                    __builtin_amdgcn_wave_barrier();
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
                    for (int cc = 0; cc < 16; cc++) {
                        s0 += sm.Xk[lane][cc] * sm.rhs[cc];
                        s1 += sm.Xk[lane][cc + 16] * sm.rhs[cc + 16];
                        s2 += sm.Xk[lane][cc + 32] * sm.rhs[cc + 32];
                        s3 += sm.Xk[lane][cc + 48] * sm.rhs[cc + 48];
                    }
                    const double s = (s0 + s1) + (s2 + s3);
                    zz[c0 + lane] = s;
                    sm.zk[lane] = s;
                    if (c0 + lane < 1024) sm.zs[c0 + lane] = s;
                    if (want_inv) {
                        // alpha = U z accumulated panel by panel; the diagonal block U_kk = L_kk^-T opens rows C_k
                        __builtin_amdgcn_wave_barrier();
                        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
                        for (int cc = 0; cc < 16; cc++) {
                            a0 += sm.Xk[cc][lane] * sm.zk[cc];
                            a1 += sm.Xk[cc + 16][lane] * sm.zk[cc + 16];
                            a2 += sm.Xk[cc + 32][lane] * sm.zk[cc + 32];
                            a3 += sm.Xk[cc + 48][lane] * sm.zk[cc + 48];
                        }
                        alpha[c0 + lane] = (a0 + a1) + (a2 + a3);
                    }
                } else {
                    // meanwhile the other waves store L_kk (lower) and U_kk = L_kk^-T (upper, zeros below)
                    for (int e = (wave - 1) * 64 + lane; e < 64 * 64; e += NT - 64) {   // waves 1 .. NW-1
                        int rr = e >> 6, cc = e & 63;
                        if (cc <= rr) Lb[(size_t)(c0 + rr) * ld + c0 + cc] = sm.Dk[rr][cc];
                        if (store_ukk) Ub[(size_t)(c0 + rr) * ld + c0 + cc] = (cc >= rr) ? sm.Xk[cc][rr] : 0.0;
                    }
                }
                STAMP(2);   // z / alpha solves of wave 0 beside the L_kk / U_kk stores of the others
                // Dk fully read before the store slabs (same LDS) are written: LDS-only barrier (no wait for the global
                // stores above, which nobody reads before the next step)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            STAMP(3);   // L_kk / U_kk stores (folded into diag)
            // ---- triangular solve as GEMM: out^T[ct] = sum_{ct' <= ct} Xk[ct, ct'] acc^T[ct'] on all four units (an
            //      inactive or diagonal unit computes on zeros / unused values: 40 MFMAs, never stored); each 16-column
            //      slab goes through the wave's LDS slab and leaves as whole 128-byte lines
            {
                CI_SLAB_LANE();
                bool st[UPW];
                double pal[UPW];
#pragma unroll
                for (int u = 0; u < UPW; u++) { st[u] = act[u] && !(pass == 0 && wg == 0 && u == 0); pal[u] = 0.0; }
#pragma unroll
                for (int ct = 3; ct >= 0; ct--) {
                    v4d o[UPW];
#pragma unroll
                    for (int u = 0; u < UPW; u++) o[u] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int cp = 0; cp <= ct; cp++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            double a = sm.Xk[16 * ct + li][16 * cp + 4 * r + g];
#pragma unroll
                            for (int u = 0; u < UPW; u++) o[u] = CI_MFMA_TRSM(a, acc[cp][u][r], o[u]);
                        }
#pragma unroll
                    for (int u = 0; u < UPW; u++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) S[16 * u + (lane_o & 15)][4 * r + (lane_o >> 4)] = o[u][r];
                        if (st[u] && !isM[u]) {   // U rows: alpha_i += sum_c U[i][c] z_k[c]
#pragma unroll
                            for (int r = 0; r < 4; r++) pal[u] += o[u][r] * sm.zk[16 * ct + 4 * r + g];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 2 * UPW; i++) {
                        const int u = i >> 1;
                        if (st[u]) {
                            gd_t *Out = (isM[u] ? Lb : Ub) + (size_t)(rowb[u] + 8 * (i & 1) + srow8) * ld + c0 + 16 * ct + 2 * spc;
                            *(gv2d_t *)Out = *(const v2d *)&S[8 * i + srow8][2 * spc];
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int u = 0; u < UPW; u++) {
                    if (st[u] && !isM[u]) {
                        double v = pal[u];
                        v += __shfl_xor(v, 16);
                        v += __shfl_xor(v, 32);
                        if (g == 0) alpha[rowb[u] + li] += v;   // this wave owns these rows in this step
                    }
                }
            }
            STAMP(5);   // panel solve + stores
            // Visibility of this pass's stores: columns C_k are first read in step k + 1, at chunk 4k - 1 (B staging) or
            // later, i.e. after >= 3 chunk barriers (each a full __syncthreads with vmcnt(0)) when k >= 1 -- no need to
            // wait for the store acknowledgements here.  Step 1 reads C_0 right away: full barrier after step 0 only.
            // LDS needs no barrier either: slabs are per wave, Xk / zk are next written after the next step's GEMM.
            if (k == 0) __syncthreads();
            STAMP(5);   // end-of-step barrier (folded into trsm)
        }
    }
#ifdef MEDGP_STAMPS
    if (lane == 0 && b < 64) {
        unsigned long long *dbg = (unsigned long long *)(L.slab + (size_t)b * L.slab_stride);
        for (int e = 0; e < 8; e++) dbg[(tid >> 6) * 8 + e] = sm.stamps[tid >> 6][e];
    }
#endif
    return true;
}

// grid = nbatch, block = NW * 64.  NW = 8: one workgroup per CU (lowest latency per patient);
// NW = 4: two workgroups per CU, the serial diagonal-block phase of one overlaps the MFMA phase of the other.
// sel = 0: every live entry.
// sel = 1: only entries of a single 64-block; larger ones are left to the multi-CU schedule (kernels_cholinv_la.h) -- which
//          schedule factors an entry is a function of its own n and of how many LARGE entries the call holds, never of its
//          batch-mates' sizes.
// sel = 2: only entries the multi-CU schedule has marked as failed (status -2: a pivot failed in its one attempt, or the test
//          hook dbg_fail).  This is the device-driven continuation of the reference's retry loop (ref: c_inference_exact.cpp:
//          99-111) for that schedule: the entry is re-assembled with one more noise addition and factored HERE, in-kernel loop
//          and all, so a call that takes the multi-CU schedule needs no status read-back on the host.  Healthy entries cost
//          one workgroup that loads a status word and leaves.
// (sel is a TEMPLATE parameter: as a run-time argument the extra entry paths moved hipcc's register allocation of the whole
//  kernel -- one scratch reload inside the 64-MFMA chunk loop and three times the spill traffic in the panel solve of <4,4>)
template <int NW, int UPW, int SEL>
__global__ void __launch_bounds__(NW * 64, UPW == 3 ? 3 : 8 / UPW) k_cholinv(MedgpDev L, int want_inv) {
    constexpr int NT = NW * 64;
    constexpr int sel = SEL;
    __shared__ CholInvSmem<NW, UPW> sm;
    const int b = blockIdx.x, tid = threadIdx.x;
    if constexpr (SEL == 2) { if (L.status[b] != -2) return; }
    else { if (L.status[b] < 0) return; }
    if constexpr (SEL == 1) { if (L.pn[L.bslot[b]] > 64) return; }
    // loaded values the whole workgroup agrees on: pin them to scalar registers, otherwise every quantity derived from
    // n (block counts, slot tables, row bases) lives in VGPRs across the MFMA loop and its branches run on exec masks
    const int slot = __builtin_amdgcn_readfirstlane(L.bslot[b]);
    const int n = __builtin_amdgcn_readfirstlane(L.pn[slot]);
    const int ld = L.ldn, npad = medgp_roundup(n, 64);
    int count = 0;
    if constexpr (SEL == 2) {   // attempt 0 was the multi-CU schedule's: continue with the first retry
        count = 1;
        reassemble_wg(L, b, slot, n, npad, count);
    }
    (void)sel;
    while (true) {
#ifdef MEDGP_PRICE_NLML
        (void)cholinv_attempt<NW, UPW>(L, b, slot, n, want_inv, sm);
        break;
#else
        if (cholinv_attempt<NW, UPW>(L, b, slot, n, want_inv, sm) && count >= L.dbg_fail) break;
#endif
        __syncthreads();
        if (count >= 10) {   // ref: c_inference_exact.cpp:99,109-111
            if (tid == 0) L.status[b] = -1;
            return;
        }
        count++;
        reassemble_wg(L, b, slot, n, npad, count);
    }
    __syncthreads();
    const gd_t *zz = (const gd_t *)(L.z + (size_t)b * ld);
    // quad = z^T z (fixed-order tree)
    {
        double s = 0.0;
        for (int i = tid; i < npad; i += NT) s += zz[i] * zz[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if ((tid & 63) == 0) sm.red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) {
            double q = 0.0;
            for (int w = 0; w < NW; w++) q += sm.red[w];
            L.status[b] = count;
            L.scal[b * 4 + 0] = sm.logdet;
            L.scal[b * 4 + 1] = q;
        }
    }
    // alpha = L^-T z was accumulated panel by panel inside cholinv_attempt
}
