// kernels_v0.h -- the first correct HIP statement of the stages (no MFMA, simple tiling).  Product code: the generic route for
// Q > 8 mixture components (k_assemble_v0, k_lauum_v0, k_gradbins_v0 -- the tuned kernels are instantiated for Q <= 8), checked
// against the same oracle as the tuned kernels (tests/test_parity_gpu.py, Q = 9).  The first-generation dense kernels
// (k_potrf_v0, k_trtri_v0: Cholesky + forward solve, triangular inverse) are A/B code and only compiled with
// -DMEDGP_LEGACY_AB (make LEGACY=1 builds such a library; MEDGP_V0=1 then selects them at run time).
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
    const double *t = L.pt + (size_t)slot * ld;
    const int *meta = L.pmeta + (size_t)slot * ld;
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
#ifdef MEDGP_LEGACY_AB
// ------------------------------------------------------------------------------------------
// stage 2: Cholesky (lower, in place) + forward solve z = L^-1 y + log-det + quad, with the
// reference's jitter-retry loop.  Right-looking, 16-wide panels, one workgroup per problem.
//   ref: c_inference_exact.cpp:96-125, 146   (LAPACKE_spotrf / spotrs live in MKL)
// ------------------------------------------------------------------------------------------
struct PotrfSmem {
    double Akk[16][17];
    double Pi[64][17];
    double Pj[64][17];
    double zb[16];
    double red[256];
    double logdet;
    int fail;
};

__device__ bool potrf_wg_v0(double *A, int ld, int np, double *zz, PotrfSmem &sm) {
    const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
    if (tid == 0) { sm.fail = 0; sm.logdet = 0.0; }
    __syncthreads();
    for (int k0 = 0; k0 < np; k0 += 16) {
        sm.Akk[r][c] = A[(size_t)(k0 + r) * ld + k0 + c];
        __syncthreads();
        for (int j = 0; j < 16; j++) {
            if (tid == 0) {
                double piv = sm.Akk[j][j];
                if (!(piv > 0.0)) sm.fail = 1;   // LAPACK potf2: ajj <= 0 or NaN
                else { piv = sqrt(piv); sm.Akk[j][j] = piv; sm.logdet += log(piv); }
            }
            __syncthreads();
            if (sm.fail) return false;
            if (c == j && r > j) sm.Akk[r][j] /= sm.Akk[j][j];
            __syncthreads();
            if (c > j && r >= c) sm.Akk[r][c] -= sm.Akk[r][j] * sm.Akk[c][j];
            __syncthreads();
        }
        if (r >= c) A[(size_t)(k0 + r) * ld + k0 + c] = sm.Akk[r][c];
        // z block: L_kk z_k = y_k (already updated by previous panels)
        if (tid == 0) {
            for (int rr = 0; rr < 16; rr++) {
                double s = zz[k0 + rr];
                for (int cc = 0; cc < rr; cc++) s -= sm.Akk[rr][cc] * sm.zb[cc];
                s /= sm.Akk[rr][rr];
                sm.zb[rr] = s;
                zz[k0 + rr] = s;
            }
        }
        __syncthreads();
        // panel: rows below solve x L_kk^T = a; update z
        for (int i = k0 + 16 + tid; i < np; i += blockDim.x) {
            double x[16];
            double *ai = A + (size_t)i * ld + k0;
            double zacc = 0.0;
#pragma unroll
            for (int cc = 0; cc < 16; cc++) x[cc] = ai[cc];
#pragma unroll
            for (int cc = 0; cc < 16; cc++) {
                double s = x[cc];
#pragma unroll
                for (int c2 = 0; c2 < cc; c2++) s -= x[c2] * sm.Akk[cc][c2];
                s /= sm.Akk[cc][cc];
                x[cc] = s;
                zacc += s * sm.zb[cc];
            }
#pragma unroll
            for (int cc = 0; cc < 16; cc++) ai[cc] = x[cc];
            zz[i] -= zacc;
        }
        __syncthreads();
        // trailing update, 64x64 tiles of rows/cols >= k0+16
        const int m0 = k0 + 16, mt = (np - m0 + 63) / 64;
        for (int ti = 0; ti < mt; ti++)
            for (int tj = 0; tj <= ti; tj++) {
                const int I0 = m0 + ti * 64, J0 = m0 + tj * 64;
                {
                    int rr = tid >> 2, c4 = (tid & 3) * 4;
                    int gi = I0 + rr, gj = J0 + rr;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        sm.Pi[rr][c4 + u] = (gi < np) ? A[(size_t)gi * ld + k0 + c4 + u] : 0.0;
                        sm.Pj[rr][c4 + u] = (gj < np) ? A[(size_t)gj * ld + k0 + c4 + u] : 0.0;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) {
                        int li = r + 16 * a, lj = c + 16 * bb;
                        int i = I0 + li, j = J0 + lj;
                        if (i < np && j <= i) {
                            double s = 0.0;
#pragma unroll
                            for (int cc = 0; cc < 16; cc++) s += sm.Pi[li][cc] * sm.Pj[lj][cc];
                            A[(size_t)i * ld + j] -= s;
                        }
                    }
                __syncthreads();
            }
    }
    return true;
}

__global__ void __launch_bounds__(256) k_potrf_v0(MedgpDev L) {
    __shared__ PotrfSmem sm;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], np = medgp_roundup(n, 16), ld = L.ldn;
    double *A = L.Kmat + (size_t)b * ld * ld;
    double *zz = L.z + (size_t)b * ld;
    const double *y = L.py + (size_t)slot * ld;
    int count = 0;
    while (true) {
        for (int i = tid; i < ld; i += blockDim.x) zz[i] = (i < n) ? y[i] : 0.0;
        __syncthreads();
        if (potrf_wg_v0(A, ld, np, zz, sm)) break;
        __syncthreads();
        if (count >= 10) {   // ref: c_inference_exact.cpp:99,109-111
            if (tid == 0) L.status[b] = -1;
            return;
        }
        count++;
        reassemble_wg(L, b, slot, n, np, count);
    }
    // quad = z^T z, deterministic tree
    double s = 0.0;
    for (int i = tid; i < np; i += blockDim.x) s += zz[i] * zz[i];
    sm.red[tid] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) sm.red[tid] += sm.red[tid + off];
        __syncthreads();
    }
    if (tid == 0) {
        L.status[b] = count;
        L.scal[b * 4 + 0] = sm.logdet;
        L.scal[b * 4 + 1] = sm.red[0];
    }
}

// ------------------------------------------------------------------------------------------
// stage 3 (v0 fallback): L^-1 by forward substitution, one thread per column; alpha = L^-T z.
//   ref: c_inference_exact.cpp:124-143 (spotrs, strtri)
// Stored like k_cholinv does: Linv holds U = L^-T (upper, row-major), (L^-1)[i][j] = U[j][i].
// Rows/cols in [np16, npad64) are identity padding.
// ------------------------------------------------------------------------------------------
#define XU(i, j) X[(size_t)(j) * ld + (i)]
__global__ void __launch_bounds__(256) k_trtri_v0(MedgpDev L) {
    const int b = blockIdx.x;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], np = medgp_roundup(n, 16), npad = medgp_roundup(n, 64), ld = L.ldn;
    const double *Lm = L.Kmat + (size_t)b * ld * ld;
    double *X = L.Linv + (size_t)b * ld * ld;
    const double *zz = L.z + (size_t)b * ld;
    double *alpha = L.alpha + (size_t)b * ld;
    const int lane = threadIdx.x & 63;
    for (int j = threadIdx.x; j < npad; j += blockDim.x) {
        const int jw = j - lane;   // first column of this wave (uniform)
        for (int i = (j & ~63); i < j; i++) XU(i, j) = 0.0;   // zero the strictly-lower part of U's diagonal block
        if (jw >= np) {            // whole wave in the identity padding
            for (int i = j; i < npad; i++) XU(i, j) = (i == j) ? 1.0 : 0.0;
            alpha[j] = 0.0;
            continue;
        }
        const bool real = j < np;
        if (real) XU(j, j) = 1.0 / Lm[(size_t)j * ld + j];
        for (int i = jw + 1; i < np; i++) {
            double s = 0.0;
            const double *li = Lm + (size_t)i * ld;
            for (int k = jw; k < i; k++) {
                double xv = (real && k >= j) ? XU(k, j) : 0.0;
                s += li[k] * xv;
            }
            if (real && i > j) XU(i, j) = -s / li[i];
        }
        if (real) {
            for (int i = np; i < npad; i++) XU(i, j) = 0.0;
            double a = 0.0;
            for (int i = j; i < np; i++) a += XU(i, j) * zz[i];
            alpha[j] = a;
        } else {
            for (int i = j; i < npad; i++) XU(i, j) = (i == j) ? 1.0 : 0.0;
            alpha[j] = 0.0;
        }
    }
}
#undef XU

#endif  // MEDGP_LEGACY_AB

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
    const double *t = L.pt + (size_t)slot * ld;
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
