// kernels_assemble.h -- Gram assembly, tuned: one workgroup (4 waves) per lower 64x64 tile.
//   K_ij = sum_q B_q[m_i,m_j] cos(w_q (t_i - t_j)) exp(-c_q (t_i - t_j)^2)  + [i == j] sigma^2_{m_i}
//   ref: kernel/c_kernel_LMC_SM.cpp:152-196 (+ SE :72-89, SM :75-110), inference/c_inference_exact.cpp:88-92
// lane = column j (its t, meta, cos/sin table entries stay in registers), wave = 16-row group (row constants are
// wave-uniform scalar loads); one fp64 exp per (pair, component), no trigonometry in the pair loop; every
// row is stored as one coalesced 512-byte segment.  Identity padding up to the next multiple of 64.
#pragma once
#include "medgp_dev.h"

// exp(-x) for x >= 0, ~1e-15 relative: Cody-Waite reduction by ln2/1, degree-11 Taylor/Horner, v_ldexp_f64.
// (results that underflow come out as 0, exactly what the envelope needs)
__device__ __forceinline__ double exp_neg(double x) {
    const double LOG2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    double y = -x;
    double nf = rint(y * LOG2E);
    double r = fma(-nf, LN2_HI, y);
    r = fma(-nf, LN2_LO, r);
    double p = 2.50521083854417187751e-08;            // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);        // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);        // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);        // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);        // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);        // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);        // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);        // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);        // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    // v_cvt_i32_f64 saturates, and v_ldexp_f64 flushes exponents below the denormal range to zero, so a hugely negative
    // nf needs no clamp (arguments here are c_q dt^2 with dt in hours: |nf| stays far inside the int range anyway)
    return ldexp(p, __double2int_rn(nf));
}

// 2^z for z <= 0.  The pair loops call it with z = (-c_q log2 e) dt^2: the base change is folded into the per-component constant, the
// reduction  f = z - rint(z)  is one exact subtraction (no Cody-Waite pair), and the polynomial is the degree-10 minimax
// approximation of 2^f on [-1/2, 1/2] with p(0) = 1 exactly (Remez exchange on the relative error, scratch/exp2_minimax.py): max
// relative error 3.0e-16 with the coefficients rounded to double -- the degree-11 Taylor form it replaces had 8.6e-15 for one more
// fma.  14 instructions against 18 for exp_neg(c_q dt^2).
__device__ __forceinline__ double exp2_nonpos(double z) {
    const double nf = rint(z);
    const double f = z - nf;                          // |f| <= 1/2, exact
    double p = 7.111758687557852e-09;
    p = fma(p, f, 1.0210506458816145e-07);
    p = fma(p, f, 1.321523563477604e-06);
    p = fma(p, f, 1.525264774769592e-05);
    p = fma(p, f, 0.00015403530800818256);
    p = fma(p, f, 0.001333355824648072);
    p = fma(p, f, 0.009618129107380715);
    p = fma(p, f, 0.05550410866434658);
    p = fma(p, f, 0.24022650695910472);
    p = fma(p, f, 0.6931471805599516);
    p = fma(p, f, 1.0);
    return ldexp(p, __double2int_rn(nf));
}
#define MEDGP_LOG2E 1.4426950408889634074
// pin a value every lane agrees on to a scalar register pair
__device__ __forceinline__ double uniform_d(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// value of `v` in lane `srclane` (wave-uniform index) as a scalar: two v_readlane_b32
__device__ __forceinline__ double lane_bcast(double v, int srclane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// QT = mixture components this launch evaluates, Q0 = index of the first one.  Q <= 8: one launch <Q, 0>.  8 < Q <= 16 (Q is a free
// configuration key of the reference, ref: kernel/c_kernel_LMC_SM.cpp:51-70): <8, 0> followed by <Q - 8, 8>, which ADDS its
// components to the tile the first launch wrote (components in ascending order, as the reference sums them, :182-191).
template <int QT, int Q0 = 0>
__global__ void __launch_bounds__(256) k_assemble_t(MedgpDev L) {
    const int b = blockIdx.y;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    int I, J;
    tile_decode(blockIdx.x, I, J);
    if (I >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = L.D;
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double *B = hyp + hyp_off_B(L) + (size_t)Q0 * D * D;
    const double *t = L.pt + (size_t)slot * L.pld;
    const int *meta = L.pmeta + (size_t)slot * L.pld;
    const double *csb = L.cs + ((size_t)b * L.Q + Q0) * ld, *snb = L.sn + ((size_t)b * L.Q + Q0) * ld;
    double *K = L.Kmat + (size_t)b * ld * ld;
    double cq2n[QT];   // -c_q log2(e): exp(-c_q dt^2) = 2^(cq2n dt^2)
#pragma unroll
    for (int q = 0; q < QT; q++) cq2n[q] = uniform_d(-hyp[hyp_off_c(L) + Q0 + q] * MEDGP_LOG2E);
    const int j = 64 * J + lane;
    const bool jv = j < n;
    const double tj = t[j];
    const int mj = meta[j];
    double csj[QT], snj[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) { csj[q] = csb[q * ld + j]; snj[q] = snb[q * ld + j]; }
    // B_q[m_i][m_j]: observations are grouped by output, so m_i changes every ~N/D rows; the Q gathered values are kept
    // in registers and re-gathered only when the (wave-uniform) row output changes -- 80 -> ~10 gathers per lane and
    // tile (the per-element gathers kept the texture-address unit as busy as the VALU).
    double bq[QT];
    int mcur = -1;
    // row constants of the wave's 16 rows: loaded once, lane r holds row r (lanes >= 16 mirror) -- instead of 2 + 2Q
    // dependent scalar loads (and their lgkmcnt(0) stall) in every row iteration
    const int irow = 64 * I + 16 * w + (lane & 15);
    const double r_t = t[irow];
    const int r_m = meta[irow];
    double r_cs[QT], r_sn[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) { r_cs[q] = csb[q * ld + irow]; r_sn[q] = snb[q * ld + irow]; }
    // ... and handed to the row loop through LDS: a same-address read is a broadcast on the LDS pipe, where v_readlane
    // pairs cost 2 + 4Q VALU per row in a VALU-bound loop
    __shared__ __attribute__((aligned(16))) double rowc[4][16][2 + 2 * QT];
    if (lane < 16) {
        rowc[w][lane][0] = r_t;
#pragma unroll
        for (int q = 0; q < QT; q++) { rowc[w][lane][2 + 2 * q] = r_cs[q]; rowc[w][lane][3 + 2 * q] = r_sn[q]; }
    }
    __builtin_amdgcn_wave_barrier();
    for (int rr = 0; rr < 16; rr++) {
        const int i = 64 * I + 16 * w + rr;          // wave-uniform
        double v;
        if (i < n) {
            const int mi = __builtin_amdgcn_readlane(r_m, rr);
            if (mi != mcur) {
                mcur = mi;
                const double *Brow = B + mi * D + mj;
#pragma unroll
                for (int q = 0; q < QT; q++) bq[q] = Brow[q * D * D];
            }
            const double dt = rowc[w][rr][0] - tj, dd = dt * dt;
            double acc = 0.0;
            if constexpr (Q0 > 0) acc = K[(size_t)i * ld + j];   // the components below Q0 (and the noise) are in the tile already
#pragma unroll
            for (int q = 0; q < QT; q++) {
                const v2d csn = *(const v2d *)&rowc[w][rr][2 + 2 * q];
                const double cd = csn[0] * csj[q] + csn[1] * snj[q];
                acc += bq[q] * (cd * exp2_nonpos(cq2n[q] * dd));
            }
            if constexpr (Q0 == 0) {
                if (i == j) { const double lik = hyp[mj]; acc += lik; for (int r = 0; r < L.jit[b]; r++) acc += lik; }   // ref c_inference_exact.cpp:88-92, :101-104
            }
            v = jv ? acc : 0.0;
        } else v = (i == j) ? 1.0 : 0.0;
        K[(size_t)i * ld + j] = v;
    }
}
