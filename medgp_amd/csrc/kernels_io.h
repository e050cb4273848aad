// kernels_io.h -- patient upload (packed staging buffer -> padded slot rows) and the predictive solve.
#pragma once
#include "medgp_dev.h"

// ------------------------------------------------------------------------------------------
// Packed patient upload.  The host packs any number of patients into ONE staging buffer (one H2D copy, no
// per-patient synchronisation); this kernel scatters entry e into the padded rows of its slot.
//   ref: util/c_objective_one.cpp:23-36 (one object per patient), dataio/c_experiment.cpp:254-309 (loader order)
// Staging layout: MedgpUpHdr[nent], then per entry at byte offset hdr.off (8-byte aligned):
//   double t[n] | double y[n] | int meta[n] | int seg[D+1] | int roff[D+1] | int coff[D+1]
// ------------------------------------------------------------------------------------------
struct MedgpUpHdr {
    int slot, n;
    long long off;
};

__global__ void __launch_bounds__(256) k_scatter_patients(const char *__restrict__ stage, int D, int ldn, int *pn, double *pt,
                                                          double *py, int *pmeta, int *pseg, int *proff, int *pcoff) {
    const MedgpUpHdr h = ((const MedgpUpHdr *)stage)[blockIdx.x];
    const int n = h.n, slot = h.slot, tid = threadIdx.x;
    const double *st = (const double *)(stage + h.off), *sy = st + n;
    const int *sm = (const int *)(sy + n), *sseg = sm + n, *sro = sseg + (D + 1), *sco = sro + (D + 1);
    for (int i = tid; i < ldn; i += blockDim.x) {
        pt[(size_t)slot * ldn + i] = (i < n) ? st[i] : 0.0;
        py[(size_t)slot * ldn + i] = (i < n) ? sy[i] : 0.0;
        pmeta[(size_t)slot * ldn + i] = (i < n) ? sm[i] : 0;
    }
    for (int i = tid; i <= D; i += blockDim.x) {
        pseg[(size_t)slot * (D + 1) + i] = sseg[i];
        proff[(size_t)slot * (D + 1) + i] = sro[i];
        pcoff[(size_t)slot * (D + 1) + i] = sco[i];
    }
    if (tid == 0) pn[slot] = n;
}

// Prior descriptors (ref: prior/c_prior.h:35-53): row blockIdx.x of the staged rows goes to slot slots[blockIdx.x]; with
// slots == NULL the ONE staged row is replicated into slot blockIdx.x (medgp_set_prior(slot = -1)).
__global__ void __launch_bounds__(256) k_scatter_priors(const MedgpPrior *__restrict__ rows, const int *__restrict__ slots, int H,
                                                        MedgpPrior *prior, uint8_t *prior_on, uint8_t on) {
    const int k = blockIdx.x, slot = slots ? slots[k] : k;
    const MedgpPrior *src = rows + (slots ? (size_t)k * H : 0);
    for (int h = threadIdx.x; h < H; h += blockDim.x) prior[(size_t)slot * H + h] = src[h];
    if (threadIdx.x == 0) prior_on[slot] = on;
}

// ------------------------------------------------------------------------------------------
// predict: mean* = k*^T K^-1 y = v^T z,  var* = k** - v^T v + sigma^2,  v = L^-1 k*,  z = L^-1 y
//   ref: core/gp_regression.cpp:128-214 (sgemv with chol_alpha, strmm with chol_factor_inv, sdsdot),
//        kernel/c_kernel_LMC_SM.cpp:329-372 (cross Gram), :122-150 (self diagonal)
// k* is one more right-hand side of the forward solve the factorisation already does for y: no inverse, no alpha.
// Blocked forward substitution over the 64-wide panels: v_k = L_kk^-1 (k*_k - L[C_k, 0:64k] v[0:64k]); L_kk^-1 is the
// diagonal block the factorisation leaves in Linv (stored as U_kk = L_kk^-T).  One workgroup per test point;
// grid = (nstar, entries of the class): test point js of the problem in caller row p lives at index p * nstar + js of meta2 / t2 / mean / var.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_predict(MedgpDev L, int nstar, const int *__restrict__ meta2, const double *__restrict__ t2,
                                                 double *__restrict__ vs_buf, float *__restrict__ mean, float *__restrict__ var) {
    __shared__ double red[256];
    __shared__ double rk[64];
    // (b = entry of this size class's view; the test points, means and variances are indexed by the CALLER's row of the entry)
    const int b = blockIdx.y, js = (L.bpos ? L.bpos[b] : b) * nstar + blockIdx.x, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, w = tid >> 6;
    const int slot = L.bslot[b], n = L.pn[slot], ld = L.ldn, Q = L.Q, D = L.D, npad = medgp_roundup(n, 64);
    if (L.status[b] < 0) {
        if (tid == 0) { mean[js] = __builtin_nanf(""); var[js] = __builtin_nanf(""); }
        return;
    }
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double *B = hyp + hyp_off_B(L), *wq = hyp + hyp_off_w(L), *c = hyp + hyp_off_c(L);
    const double *t = L.pt + (size_t)slot * L.pld;
    const int *meta = L.pmeta + (size_t)slot * L.pld;
    const double *zz = L.z + (size_t)b * ld;
    const double *Lm = L.Kmat + (size_t)b * ld * ld, *U = L.Linv + (size_t)b * ld * ld;
    double *v = vs_buf + (size_t)js * L.pld;   // (work rows at the context's stride: the classes of a call have different leading dimensions)
    const int ms = meta2 ? meta2[js] : 0;
    const double ts = t2[js];
    for (int i = tid; i < npad; i += nt) {
        double acc = 0.0;
        if (i < n) {
            const double d = t[i] - ts, dd = d * d;
            for (int q = 0; q < Q; q++) acc += B[q * D * D + meta[i] * D + ms] * (cos(wq[q] * d) * exp(-c[q] * dd));
        }
        v[i] = acc;   // k*, overwritten block by block with v
    }
    __syncthreads();
    for (int c0 = 0; c0 < npad; c0 += 64) {
        // r = k*_k - L[C_k, 0:c0] v[0:c0]: wave w takes rows w, w+4, ...; a row is read as coalesced 512-byte segments
        for (int r = w; r < 64; r += 4) {
            const double *lr = Lm + (size_t)(c0 + r) * ld;
            double s = 0.0;
            for (int j = lane; j < c0; j += 64) s += lr[j] * v[j];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) rk[r] = v[c0 + r] - s;
        }
        __syncthreads();
        if (tid < 64) {   // v_k = L_kk^-1 r;  (L_kk^-1)[i][j] = U[c0 + j][c0 + i], j <= i
            double s = 0.0;
            for (int j = 0; j <= tid; j++) s += U[(size_t)(c0 + j) * ld + c0 + tid] * rk[j];
            v[c0 + tid] = s;
        }
        __syncthreads();
    }
    double m = 0.0, qv = 0.0;
    for (int i = tid; i < n; i += nt) { const double vi = v[i]; m += vi * zz[i]; qv += vi * vi; }
    red[tid] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    const double mval = red[0];
    __syncthreads();
    red[tid] = qv;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) red[tid] += red[tid + off]; __syncthreads(); }
    if (tid == 0) {
        double kss = 0.0;
        for (int q = 0; q < Q; q++) kss += B[q * D * D + ms * D + ms];
        mean[js] = (float)mval;
        var[js] = (float)(kss - red[0] + hyp[ms]);
    }
}
