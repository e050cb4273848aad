// kernels_v0.h -- the generic (non-templated, no MFMA in the pair loops) route for Q > 16 mixture components: k_assemble_v0,
// k_lauum_v0, k_gradbins_v0 (the tuned kernels cover Q <= 8 in one launch and 9 <= Q <= 16 in two; Q is a free configuration key
// of the reference, ref: kernel/c_kernel_LMC_SM.cpp:51-70).  Checked against the same oracle as the tuned kernels
// (tests/test_parity_gpu.py, Q = 17); MEDGP_V0=1 selects them for any Q (A/B parity).  The dense stages always run on k_cholinv / the look-ahead schedule.  (The
// first-generation dense kernels k_potrf_v0 / k_trtri_v0 and the first multi-CU schedule k_ci_* left the tree in round 4: git
// history keeps them.)
#pragma once
#include "kernels_core.h"

// ------------------------------------------------------------------------------------------
// stage 1: Gram assembly, lower 64x64 tiles, noise on the diagonal, identity padding up to npad64
//   ref: c_kernel_LMC_SM.cpp:152-196, c_inference_exact.cpp:88-92
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_assemble_v0(MedgpDev L) {
    const int b = blockIdx.y;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], npad = medgp_roundup(n, MEDGP_TILE), nt = npad / MEDGP_TILE;
    int I, J;
    tile_decode(blockIdx.x, I, J);
    if (I >= nt) return;
    const int ld = L.ldn;
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double *cs = L.cs + (size_t)b * L.Q * ld, *sn = L.sn + (size_t)b * L.Q * ld;
    const double *t = L.pt + (size_t)slot * L.pld;
    const int *meta = L.pmeta + (size_t)slot * L.pld;
    double *K = L.Kmat + (size_t)b * ld * ld;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    for (int a = 0; a < 4; a++)
        for (int bb = 0; bb < 4; bb++) {
            int i = I * 64 + ty + 16 * a, j = J * 64 + tx + 16 * bb;
            double v;
            if (i < n && j < n) {
                v = cov_elem(L, hyp, cs, sn, t, meta, i, j);
                if (i == j) { const double lik = hyp[meta[i]]; v += lik; for (int r = 0; r < L.jit[b]; r++) v += lik; }
            } else v = (i == j) ? 1.0 : 0.0;
            K[(size_t)i * ld + j] = v;
        }
}

// whole-matrix re-assembly by one workgroup with `count` extra noise additions (jitter path)
//   ref: c_inference_exact.cpp:99-108

// ------------------------------------------------------------------------------------------
// stage 4: W = L^-T L^-1 - alpha alpha^T = U U^T - alpha alpha^T, lower 64x64 tiles, written over the (dead) L buffer
//   ref: c_inference_exact.cpp:168-172
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lauum_v0(MedgpDev L) {
    __shared__ double Ai[16][64], Aj[16][64];
    const int b = blockIdx.y;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], np = medgp_roundup(n, 64), nt = np / 64, ld = L.ldn;
    int I, J;
    tile_decode(blockIdx.x, I, J);
    if (I >= nt) return;
    const double *U = L.Linv + (size_t)b * ld * ld;   // U[i][k] = (L^-1)[k][i]
    const double *alpha = L.alpha + (size_t)b * ld;
    double *W = L.Kmat + (size_t)b * ld * ld;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int i0 = I * 64, j0 = J * 64;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int bb = 0; bb < 4; bb++) acc[a][bb] = 0.0;
    for (int kc = i0; kc < np; kc += 16) {
        {
            int rr = tid >> 2, k4 = (tid & 3) * 4;
            int ii = i0 + rr, jj = j0 + rr;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int k = kc + k4 + u;
                Ai[k4 + u][rr] = (k < np && k >= ii) ? U[(size_t)ii * ld + k] : 0.0;
                Aj[k4 + u][rr] = (k < np && k >= jj) ? U[(size_t)jj * ld + k] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++)
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int bb = 0; bb < 4; bb++) acc[a][bb] += Ai[kk][ty + 16 * a] * Aj[kk][tx + 16 * bb];
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            int i = i0 + ty + 16 * a, j = j0 + tx + 16 * bb;
            W[(size_t)i * ld + j] = acc[a][bb] - alpha[i] * alpha[j];
        }
}

// ------------------------------------------------------------------------------------------
// stage 5: block reductions over the lower triangle of W (observations are grouped by output):
//   S_q[d,e]  = sum_{i in d, j in e} W_ij k_q(ij)
//   SM_q[d,e] = sum W_ij * ( -(w d) sin(w d) E )          (d k_q / d log mu_q)
//   SV_q[d,e] = sum W_ij * ( -2 c d^2 k_q )               (d k_q / d log v_q)
// for e <= d; diagonal blocks count off-diagonal pairs twice.  One thread per (q, d, e): fixed
// summation order => bitwise reproducible.
//   ref: c_kernel_LMC_SM.cpp:198-327 regrouped; k/km/kv :374-391
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gradbins_v0(MedgpDev L) {
    const int b = blockIdx.y;
    if (L.status[b] < 0) return;
    const int Q = L.Q, D = L.D, nb = D * (D + 1) / 2, ld = L.ldn;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Q * nb) return;
    const int q = idx / nb;
    int d, e;
    tile_decode(idx - q * nb, d, e);
    const int slot = L.bslot[b];
    const int *seg = L.pseg + (size_t)slot * (D + 1);
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double wq = hyp[hyp_off_w(L) + q], cq = hyp[hyp_off_c(L) + q];
    const double *cs = L.cs + ((size_t)b * Q + q) * ld, *sn = L.sn + ((size_t)b * Q + q) * ld;
    const double *t = L.pt + (size_t)slot * L.pld;
    const double *W = L.Kmat + (size_t)b * ld * ld;
    const int r0 = seg[d], r1 = seg[d + 1], c0 = seg[e], c1 = seg[e + 1];
    double aS = 0.0, aM = 0.0, aV = 0.0;
    for (int i = r0; i < r1; i++) {
        const double ti = t[i], ci = cs[i], si = sn[i];
        const int jend = (d == e) ? i + 1 : c1;
        for (int j = c0; j < jend; j++) {
            double wv = W[(size_t)i * ld + j];
            if (d == e && j < i) wv *= 2.0;
            double dt = ti - t[j], dd = dt * dt;
            double E = exp(-cq * dd);
            double cd = ci * cs[j] + si * sn[j];
            double sd = si * cs[j] - ci * sn[j];
            double k = cd * E;
            double wd = wq * dt;
            aS += wv * k;
            aM += wv * (-(wd * sd) * E);
            aV += wv * (-2.0 * cq * dd * k);
        }
    }
    const size_t o = (size_t)b * Q * D * D + (size_t)q * D * D + d * D + e;
    L.S[o] = aS;
    L.SM[o] = aM;
    L.SV[o] = aV;
}
