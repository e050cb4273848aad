// kernels_cholinv_mc.h -- multi-CU variant of the fused Cholesky + triangular inverse for FEW, LARGE patients
// (BASELINE configs 3 and 5: one patient with N = 2048 / 4096).  Same formulation and arithmetic as
// k_cholinv (kernels_cholinv.h): the left-looking panel recurrence on T = [K; I]; but every 64-wide step is two
// launches whose workgroups each own ONE 64-row block, so a step uses N/64 CUs instead of one:
//   k_ci_panel(k): block <- init - history GEMM (MFMA, shared operand staged through LDS); the diagonal block's
//                  workgroup also factors it (same single-wave tile code), publishes L_kk, U_kk, L_kk^-1, z_k.
//   k_ci_trsm(k) : every other block <- block * L_kk^-T (MFMA), alpha += U[rows, C_k] z_k.
// The kernel boundary is the grid-wide synchronisation (about 2 us each, 2 N/64 of them).
// The reference's jitter loop (ref: inference/c_inference_exact.cpp:99-108) is driven from the host here: a failed
// pivot marks the problem (status -2), the host bumps its jitter count and re-runs assembly + factorisation.
// A/B code: compiled only with -DMEDGP_LEGACY_AB (make LEGACY=1); MEDGP_MC_OLD=1 then selects it at run time.
#pragma once
#include "kernels_cholinv.h"
#ifdef MEDGP_LEGACY_AB

#define MC_KC 32
#define MC_THREADS 256

struct McSmem {
    union {
        double Bs[2][64][MC_KC + 2];
        double Dk[64][66];
    };
    double Xk[64][66];
    double zacc[64];
    alignas(16) double rhs[64 + 128];
    double zk[64], rdiag[16];
    double logdet;
    int fail;
};
static_assert(sizeof(double) * 2 * 64 * (MC_KC + 2) >= sizeof(double) * 64 * 66, "Dk must fit in the staging buffers");

// the diagonal block is factored by the same single-wave tile code as k_cholinv
__device__ inline void mc_diag_factor(McSmem &sm, int lane) {
    if (lane == 0) sm.logdet = 0.0;
    __builtin_amdgcn_wave_barrier();
    diag_factor_wave((ld_t *)&sm.Dk[0][0], (ld_t *)&sm.Xk[0][0], (ld_t *)sm.rhs, (li_t *)&sm.fail, (ld_t *)&sm.logdet, lane);
}

// grid = (nbatch, max blocks), block = 256 (4 waves x 16 rows)
__global__ void __launch_bounds__(MC_THREADS) k_ci_panel(MedgpDev L, int k, int want_mode) {
    __shared__ McSmem sm;
    const int want_inv = want_mode & 1;   // bit 1 alone: only the diagonal blocks U_kk are stored (k_predict)
    // grid = (nbatch, blocks): block-slot major, so the diagonal-block workgroups of ALL patients (slot 0: GEMM + the
    // serial 64x64 factorisation, the long pole of the launch) are dispatched first and the history-only workgroups fill
    // in behind them by decreasing history length
    const int b = blockIdx.x;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    if (k >= nb) return;
    const int nM = nb - k, ntot = nM + (want_inv ? k : 0);
    const int bidx = blockIdx.y;
    if (bidx >= ntot) return;
    const bool isM = bidx < nM, is_diag = (bidx == 0);
    const int rblk = isM ? (k + bidx) : (bidx - nM);
    const int c0 = 64 * k;
    double *Lb = L.Kmat + (size_t)b * ld * ld, *Ub = L.Linv + (size_t)b * ld * ld;
    double *zz = L.z + (size_t)b * ld;
    const double *Hist = isM ? Lb : Ub;
    double *Out = isM ? Lb : Ub;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = 64 * rblk + 16 * w;
    const int kstart = isM ? 0 : 64 * rblk;

    // acc[ct]: rows row0 + 4r + g, columns c0 + 16 ct + li ; starts at -init, the GEMM adds the history product
    v4d acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        acc[ct] = (v4d){0.0, 0.0, 0.0, 0.0};
        if (isM) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc[ct][r] = -Lb[(size_t)(row0 + 4 * r + g) * ld + c0 + 16 * ct + li];
        }
    }
    double zsum = 0.0;
    const int nch = (c0 - kstart) / MC_KC;
    if (nch > 0) {
        const double *Arow = Hist + (size_t)(row0 + li) * ld + kstart + 2 * g;
        const int srow = tid >> 2, scg = (tid & 3) * 8;
        const double *Bsrc = Lb + (size_t)(c0 + srow) * ld + kstart + scg;
        v2d bst[4], an[4];
#pragma unroll
        for (int u = 0; u < 4; u++) bst[u] = *(const v2d *)(Bsrc + 2 * u);
#pragma unroll
        for (int h = 0; h < 4; h++) an[h] = *(const v2d *)(Arow + 8 * h);
#pragma unroll
        for (int u = 0; u < 4; u++) *(v2d *)&sm.Bs[0][srow][scg + 2 * u] = bst[u];
        __syncthreads();
        for (int c = 0; c < nch; c++) {
            const int buf = c & 1;
            v2d ac[4];
#pragma unroll
            for (int h = 0; h < 4; h++) ac[h] = an[h];
            if (c + 1 < nch) {
#pragma unroll
                for (int u = 0; u < 4; u++) bst[u] = *(const v2d *)(Bsrc + (c + 1) * MC_KC + 2 * u);
#pragma unroll
                for (int h = 0; h < 4; h++) an[h] = *(const v2d *)(Arow + (c + 1) * MC_KC + 8 * h);
            }
#pragma unroll
            for (int h = 0; h < 4; h++)
#pragma unroll
                for (int ct = 0; ct < 4; ct++) {
                    const v2d bf = *(const v2d *)&sm.Bs[buf][16 * ct + li][8 * h + 2 * g];
#pragma unroll
                    for (int s = 0; s < 2; s++) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[h][s], bf[s], acc[ct], 0, 0, 0);
                }
            if (is_diag && w == 3) {   // z history: L[C_k, chunk] z[chunk], lane = panel column
#pragma unroll 8
                for (int kk = 0; kk < MC_KC; kk++) zsum += sm.Bs[buf][lane][kk] * zz[kstart + c * MC_KC + kk];
            }
            if (c + 1 < nch) {
#pragma unroll
                for (int u = 0; u < 4; u++) *(v2d *)&sm.Bs[buf ^ 1][srow][scg + 2 * u] = bst[u];
            }
            __syncthreads();
        }
    }
    if (!is_diag) {
        // pre-solve panel values; k_ci_trsm(k) finishes them
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) Out[(size_t)(row0 + 4 * r + g) * ld + c0 + 16 * ct + li] = -acc[ct][r];
        return;
    }
    // ---- diagonal block: factor, invert, z_k, publish
    if (tid == 0) { sm.fail = 0; sm.logdet = 0.0; }
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) sm.Dk[16 * w + 4 * r + g][16 * ct + li] = -acc[ct][r];
    if (w == 3) sm.zacc[lane] = zsum;
    __syncthreads();
    const double *y = L.py + (size_t)slot * ld;
    double *alpha = L.alpha + (size_t)b * ld;
    if (w == 0) {
        mc_diag_factor(sm, lane);
        if (!sm.fail) {
            sm.rhs[lane] = ((c0 + lane < n) ? y[c0 + lane] : 0.0) - sm.zacc[lane];
            __builtin_amdgcn_wave_barrier();
            double s = 0.0;
            for (int cc = 0; cc <= lane; cc++) s += sm.Xk[lane][cc] * sm.rhs[cc];
            zz[c0 + lane] = s;
            sm.zk[lane] = s;
            if (want_inv) {
                __builtin_amdgcn_wave_barrier();
                double a0 = 0.0;
                for (int cc = lane; cc < 64; cc++) a0 += sm.Xk[cc][lane] * sm.zk[cc];
                alpha[c0 + lane] = a0;
            }
            if (lane == 0) L.scal[b * 4 + 0] += sm.logdet;   // steps are ordered launches: fixed summation order
        }
    }
    __syncthreads();
    if (sm.fail) {
        if (tid == 0) L.status[b] = -2;   // host-driven jitter retry
        return;
    }
    double *Xg = L.xk + (size_t)b * 64 * 64;
    for (int e = tid; e < 64 * 64; e += MC_THREADS) {
        int rr = e >> 6, cc = e & 63;
        if (cc <= rr) Lb[(size_t)(c0 + rr) * ld + c0 + cc] = sm.Dk[rr][cc];
        if (want_mode) Ub[(size_t)(c0 + rr) * ld + c0 + cc] = (cc >= rr) ? sm.Xk[cc][rr] : 0.0;
        Xg[e] = sm.Xk[rr][cc];
    }
}

// grid = (max blocks - 1, nbatch): every block of step k except the diagonal one
__global__ void __launch_bounds__(MC_THREADS) k_ci_trsm(MedgpDev L, int k, int want_mode) {
    __shared__ double Xs[64][66];
    const int want_inv = want_mode & 1;
    const int b = blockIdx.y;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    if (k >= nb) return;
    const int nM = nb - k, ntot = nM + (want_inv ? k : 0);
    const int bidx = blockIdx.x + 1;
    if (bidx >= ntot) return;
    const bool isM = bidx < nM;
    const int rblk = isM ? (k + bidx) : (bidx - nM);
    const int c0 = 64 * k;
    double *Out = (isM ? L.Kmat : L.Linv) + (size_t)b * ld * ld;
    const double *Xg = L.xk + (size_t)b * 64 * 64;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = tid >> 6;
    const int row0 = 64 * rblk + 16 * w;
    for (int e = tid; e < 64 * 64; e += MC_THREADS) Xs[e >> 6][e & 63] = Xg[e];
    // val^T tiles: val[cp][r] = value at (row row0 + li, column c0 + 16 cp + 4 r + g)  -> B operand of the solve
    v4d val[4];
#pragma unroll
    for (int cp = 0; cp < 4; cp++)
#pragma unroll
        for (int r = 0; r < 4; r++) val[cp][r] = Out[(size_t)(row0 + li) * ld + c0 + 16 * cp + 4 * r + g];
    __syncthreads();
    const double *zz = L.z + (size_t)b * ld;
    double pal = 0.0;
#pragma unroll
    for (int ct = 0; ct < 4; ct++) {
        v4d o = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int cp = 0; cp <= ct; cp++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                o = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[16 * ct + li][16 * cp + 4 * r + g], val[cp][r], o, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            Out[(size_t)(row0 + li) * ld + c0 + 16 * ct + 4 * r + g] = o[r];
            if (!isM) pal += o[r] * zz[c0 + 16 * ct + 4 * r + g];
        }
    }
    if (!isM) {
        pal += __shfl_xor(pal, 16);
        pal += __shfl_xor(pal, 32);
        if (g == 0) L.alpha[(size_t)b * ld + row0 + li] += pal;
    }
}

// grid = nbatch: quad = z^T z in a fixed order
__global__ void __launch_bounds__(256) k_ci_finish(MedgpDev L) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (L.status[b] < 0) return;
    const int slot = L.bslot[b], n = L.pn[slot], ld = L.ldn, npad = medgp_roundup(n, 64);
    const double *zz = L.z + (size_t)b * ld;
    double s = 0.0;
    for (int i = tid; i < npad; i += 256) s += zz[i] * zz[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        L.scal[b * 4 + 1] = red[0] + red[1] + red[2] + red[3];
        L.status[b] = L.jit[b];
    }
}
#endif  // MEDGP_LEGACY_AB
