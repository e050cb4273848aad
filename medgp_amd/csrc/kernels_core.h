// kernels_core.h -- hot-path kernels around the dense algebra: hyper transform + cos/sin tables (k_prep), whole-matrix
// re-assembly of the in-kernel jitter path (reassemble_wg), block-sum reduction (k_slabsum), gradient / prior / nlml
// epilogue (k_epilogue); small helpers shared by every kernel file (cov_elem, tile_decode).
#pragma once
#include "medgp_dev.h"

// ------------------------------------------------------------------------------------------
// stage 0: theta -> sigma^2, B_q, w_q = 2 pi mu_q, c_q = 2 (pi v_q)^2, cos/sin(w_q t_i) tables
//   ref: c_kernel_LMC_SM.cpp:51-115 (exp transform, B_q), c_likelihood.cpp:38-43,
//        c_kernel_SM.cpp:41-46, c_kernel_SE.cpp:47-52
// The tables implement cos(w (t_i - t_j)) = cs_i cs_j + sn_i sn_j, so the N^2 pair loop needs no
// trigonometric evaluation (c_kernel_LMC_SM.cpp:374-378 evaluates cos per pair).
// ------------------------------------------------------------------------------------------
// grid = (nbatch, 1 + table chunks + B chunks): block y = 0 transforms the hypers (and resets the per-entry state), the next blocks fill
// PREP_CHUNK entries of the cos / sin tables each, the last ones PREP_BCHUNK entries of the B_q each (they derive w_q from theta themselves, so no block waits for another;
// one workgroup per entry made a single N = 4096, Q = 5 evaluation spend 0.12 ms here).
#define PREP_CHUNK 2048
#define PREP_BCHUNK 256   // LMC-SM: entries of the coregionalisation matrices B_q per workgroup of the blocks behind the table chunks
__device__ __forceinline__ double prep_w(const MedgpDev &L, const double *th, int q) {
    if (L.kidx == 7) return 2.0 * L.pi * exp(th[L.D + L.Q * L.D * L.R + q]);
    if (L.kidx == 8) return 2.0 * L.pi * exp(th[1 + L.Q + q]);
    return 0.0;
}
__global__ void __launch_bounds__(256) k_prep(MedgpDev L, const double *__restrict__ theta, int min_n) {
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int slot = L.bslot[b];
    const int n = L.pn[slot];
    const double *th = theta + (size_t)(L.tpos ? L.tpos[b] : (L.bpos ? L.bpos[b] : b)) * L.H;
    double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    double *sig2 = hyp, *B = hyp + hyp_off_B(L), *w = hyp + hyp_off_w(L), *c = hyp + hyp_off_c(L);
    const int Q = L.Q, D = L.D, R = L.R;
    const double pi = L.pi;
    const int ntab = (Q * L.ldn + PREP_CHUNK - 1) / PREP_CHUNK;
    if ((int)blockIdx.y > ntab) {
        // ---- B_q = A_q A_q^T + diag(kappa_q), one entry per thread (left to the block of y = 0, its 256 threads walked Q D^2 / 256
        //      entries of R dependent load pairs each: 80 of the 89 us this kernel took for one D = 64 evaluation)
        if (theta == nullptr || L.kidx != 7) return;
        const int idx = ((int)blockIdx.y - ntab - 1) * PREP_BCHUNK + tid;
        if (idx >= Q * D * D) return;
        const double *A = th + D, *lk = th + D + Q * (D * R + 2);
        const int q = idx / (D * D), rem = idx - q * D * D, i = rem / D, j = rem - i * D;
        const double *Ai = A + ((size_t)q * D + i) * R, *Aj = A + ((size_t)q * D + j) * R;
        double s = 0.0;
        for (int r = 0; r < R; r++) s += Ai[r] * Aj[r];
        if (i == j) s += exp(lk[q * D + i]);
        B[idx] = s;
        return;
    }
    if (blockIdx.y >= 1) {
        // ---- cos / sin tables.  theta == nullptr (tables only, medgp_get_factor's caller-order re-factorisation): the
        //      hyper block of this entry is kept, w_q is read from it
        const double *t = L.pt + (size_t)slot * L.pld;
        double *cs = L.cs + (size_t)b * Q * L.ldn, *sn = L.sn + (size_t)b * Q * L.ldn;
        const int lo = (blockIdx.y - 1) * PREP_CHUNK, hi = min(lo + PREP_CHUNK, Q * L.ldn);
        for (int idx = lo + tid; idx < hi; idx += nt) {
            const int q = idx / L.ldn, i = idx - q * L.ldn;
            double s = 0.0, co = 0.0;
            if (i < n) {
                const double wq = theta ? prep_w(L, th, q) : w[q];
                sincos(wq * t[i], &s, &co);
            }
            cs[idx] = co;
            sn[idx] = s;
        }
        return;
    }
    if (tid == 0) {
        // objective path: n > 2 (ref util/c_objective_one.cpp:51); train(false)+predict path: any n >= 1
        // (GP_Regression::train has no such guard, ref core/gp_regression.cpp:102-126)
        L.status[b] = (n >= min_n) ? 0 : -1;
        L.scal[b * 4 + 0] = 0.0;
        L.scal[b * 4 + 1] = 0.0;
        L.jit[b] = 0;
        L.bn[b] = n;
    }
    if (theta == nullptr) return;
    if (L.kidx == 7) {
        for (int d = tid; d < D; d += nt) { double s = exp(th[d]); sig2[d] = s * s; }
        for (int q = tid; q < Q; q += nt) {
            double v = exp(th[D + Q * D * R + Q + q]);
            double pv = pi * v;
            w[q] = prep_w(L, th, q);
            c[q] = 2.0 * (pv * pv);
        }
    } else if (L.kidx == 8) {  // SM: theta = [log sigma | log w | log mu | log v]
        if (tid == 0) { double s = exp(th[0]); sig2[0] = s * s; }
        for (int q = tid; q < Q; q += nt) {
            B[q] = exp(th[1 + q]);
            double v = exp(th[1 + 2 * Q + q]);
            double pv = pi * v;
            w[q] = prep_w(L, th, q);
            c[q] = 2.0 * (pv * pv);
        }
    } else {  // SE: theta = [log sigma | log l | log sf];  k = sf^2 exp(-d^2 / (2 l^2))
        if (tid == 0) {
            double s = exp(th[0]), l = exp(th[1]), sf = exp(th[2]);
            sig2[0] = s * s;
            B[0] = sf * sf;
            w[0] = 0.0;
            c[0] = 0.5 / (l * l);
        }
    }
}

// covariance of observations i, j of one problem (no noise)
__device__ inline double cov_elem(const MedgpDev &L, const double *hyp, const double *cs, const double *sn,
                                  const double *t, const int *meta, int i, int j) {
    const int Q = L.Q, D = L.D;
    const double *B = hyp + hyp_off_B(L), *c = hyp + hyp_off_c(L);
    double d = t[i] - t[j], dd = d * d, acc = 0.0;
    int bo = meta[i] * D + meta[j];
    for (int q = 0; q < Q; q++) {
        double E = exp(-c[q] * dd);
        double cd = cs[q * L.ldn + i] * cs[q * L.ldn + j] + sn[q * L.ldn + i] * sn[q * L.ldn + j];
        acc += B[q * D * D + bo] * (cd * E);
    }
    return acc;
}

__device__ inline void tile_decode(int x, int &I, int &J) {
    int i = (int)((sqrt(8.0 * (double)x + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= x) i++;
    while (i * (i + 1) / 2 > x) i--;
    I = i;
    J = x - i * (i + 1) / 2;
}

__device__ __attribute__((noinline)) void reassemble_wg(const MedgpDev &L, int b, int slot, int n, int np, int count) {
    const int ld = L.ldn;
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double *cs = L.cs + (size_t)b * L.Q * ld, *sn = L.sn + (size_t)b * L.Q * ld;
    const double *t = L.pt + (size_t)slot * L.pld;
    const int *meta = L.pmeta + (size_t)slot * L.pld;
    double *K = L.Kmat + (size_t)b * ld * ld;
    for (int idx = threadIdx.x; idx < np * np; idx += blockDim.x) {
        int i = idx / np, j = idx - i * np;
        if (j > i) continue;
        double v;
        if (i < n && j < n) {
            v = cov_elem(L, hyp, cs, sn, t, meta, i, j);
            if (i == j) {
                double lik = hyp[meta[i]];
                v += lik;
                for (int r = 0; r < count; r++) v += lik;
            }
        } else v = (i == j) ? 1.0 : 0.0;
        K[(size_t)i * ld + j] = v;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// stage 6: gradient in theta order + priors + nlml
//   ref: c_inference_exact.cpp:146-152 (nlml), :177-219 (order), c_inference_prior.cpp:60-150,
//        prior/c_prior.cpp:383-421
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// stage 5b: block sums S, SM, SV from the pieces k_wgrad left in the slab, added in a FIXED order (row pieces outer, column
// pieces inner): bitwise reproducible.  grid = (nbatch, ceil(3 Q D(D+1)/2 / 256)), one bin per thread (inside k_epilogue a
// single workgroup per entry walked all bins: 0.44 ms for one D = 64, N = 4096 evaluation).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_slabsum(MedgpDev L) {
    __shared__ int s_roff[MEDGP_MAX_D + 1], s_coff[MEDGP_MAX_D + 1], s_seg[MEDGP_MAX_D + 1];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (L.status[b] < 0) return;
    const int Q = L.Q, D = L.D;
    const int slot = L.bslot[b];
    for (int i = tid; i <= D; i += nt) {
        s_roff[i] = L.proff[(size_t)slot * (D + 1) + i];
        s_coff[i] = L.pcoff[(size_t)slot * (D + 1) + i];
        s_seg[i] = L.pseg[(size_t)slot * (D + 1) + i];
    }
    __syncthreads();
    const int *roff = s_roff, *coff = s_coff, *seg = s_seg;
    const double *slab = L.slab + (size_t)b * L.slab_stride;
    const int nbins = D * (D + 1) / 2;
    const int idx = blockIdx.y * nt + tid;
    if (idx >= 3 * Q * nbins) return;
    const int pq = idx / nbins;          // plane * Q + q
    int d, e;
    tile_decode(idx - pq * nbins, d, e);
    const double *sl = slab + (size_t)pq * L.slab_R * L.slab_C;
    double s = 0.0;
    for (int rs = roff[d]; rs < roff[d + 1]; rs++) {
        const int It = (seg[d] / 16 + (rs - roff[d])) / 4;
        for (int cs = coff[e]; cs < coff[e + 1]; cs++) {
            const int Jt = seg[e] / 64 + (cs - coff[e]);
            if (Jt <= It) s += sl[(size_t)rs * L.slab_C + cs];
        }
    }
    const int pl = pq / Q, q = pq - pl * Q;
    double *dst = (pl == 0 ? L.S : (pl == 1 ? L.SM : L.SV)) + (size_t)b * Q * D * D;
    dst[(size_t)q * D * D + d * D + e] = s;
}

__device__ inline void prior_apply(const MedgpPrior &p, double hv, double pi, bool want_grad, double &lp_sum, double &g) {
    if (!p.flag) return;
    if (p.type == 0) { if (want_grad) g = 0.0; return; }
    double lp, dlp;
    if (p.type == 1) {
        lp = -1.0 * (hv - p.p0) * (hv - p.p0) / (2.0 * p.p1);
        lp = lp - log(2 * pi * p.p1) / 2.0;
        dlp = -1.0 * (hv - p.p0) / p.p1;
    } else if (p.type == 2) {
        lp = (-1.0 * fabs(hv - p.p0) / p.p1) - (double)p.lg2b;   // (float log, as the reference: MedgpPrior::lg2b)
        if (hv == p.p0) dlp = 0.0;
        else dlp = -1.0 * ((hv > p.p0) ? 1.0 : -1.0) / p.p1;
    } else return;
    lp_sum += lp;
    if (want_grad) g -= p.is_exp ? hv * dlp : dlp;
}

__device__ inline double sym_get(const double *S, int D, int d, int e) { return d >= e ? S[d * D + e] : S[e * D + d]; }

__global__ void __launch_bounds__(256) k_epilogue(MedgpDev L, const double *__restrict__ theta, int flag_grad, int from_slab,
                                                  double *__restrict__ nlml_out, double *__restrict__ grad_out,
                                                  int *__restrict__ status_out) {
    __shared__ double red2[MEDGP_EPI_PARTS][256];   // per-chunk partial sums of the prior log-density
    // LDS copies of S_q (all q) and of A for the Q D R gradients dA_q = S_q A_q (each a D-term dot product whose
    // operands otherwise come from global memory one dependent pair at a time); used when they fit
    constexpr int EPI_S_MAX = 4096, EPI_A_MAX = 1280;
    __shared__ double s_S[EPI_S_MAX], s_A[EPI_A_MAX];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int H = L.H, Q = L.Q, D = L.D, R = L.R, ld = L.ldn;
    // grid = (entries, parts): part p owns the hypers [h0, h1) (one workgroup per entry walked H = 2954 hypers of a D = 64
    // evaluation with 12 dependent 64-term dot products per thread: 0.18 ms of a 3.3 ms evaluation)
    // The hyper range is cut into nchunk <= MEDGP_EPI_PARTS chunks that depend on H only; a part owns whole chunks, and the prior
    // log-density is summed per chunk, then over the chunks in order -- the same bits however many parts the host launches.
    const int part = blockIdx.y, nparts = gridDim.y;
    const int nchunk = min(MEDGP_EPI_PARTS, (H + 255) / 256), hchunk = (H + nchunk - 1) / nchunk;
    const int cpp = (nchunk + nparts - 1) / nparts, ch0 = part * cpp, ch1 = min(nchunk, ch0 + cpp);
    const int h0 = ch0 * hchunk, h1 = min(H, ch1 * hchunk);
    const int hA0 = L.kidx == 7 ? D : H, hA1 = L.kidx == 7 ? D + Q * D * R : H, hMV1 = L.kidx == 7 ? D + Q * (D * R + 2) : H;
    const bool own_A = h0 < hA1 && h1 > hA0, own_muv = h0 < hMV1 && h1 > hA1;
    const bool lds_sa = flag_grad && L.kidx == 7 && own_A && Q * D * D <= EPI_S_MAX && Q * D * R <= EPI_A_MAX;
    const int st = L.status[b];
    const int pos = L.bpos ? L.bpos[b] : b;   // the caller's row of this entry
    double *g = grad_out ? grad_out + (size_t)pos * H : nullptr;
    if (tid == 0 && part == 0 && status_out) status_out[pos] = st;
    if (st < 0) {
        if (tid == 0 && part == 0) nlml_out[pos] = __builtin_nan("");
        if (flag_grad && g) for (int h = h0 + tid; h < h1; h += nt) g[h] = __builtin_nan("");
        return;
    }
    const int slot = L.bslot[b], n = L.pn[slot];
    const double *th = theta + (size_t)(L.tpos ? L.tpos[b] : pos) * H;
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double *B = hyp + hyp_off_B(L);
    const double *S = L.S + (size_t)b * Q * D * D, *SM = L.SM + (size_t)b * Q * D * D, *SV = L.SV + (size_t)b * Q * D * D;
    const int *seg = L.pseg + (size_t)slot * (D + 1);
    // diag(W): k_wgrad exports it; the v0 path keeps the full W in the Kmat buffer
    const double *Wd = from_slab ? L.wdiag + (size_t)b * ld : L.Kmat + (size_t)b * ld * ld;
    const size_t wds = from_slab ? 1 : (size_t)ld + 1;
    if (lds_sa) {
        for (int i = tid; i < Q * D * D; i += nt) s_S[i] = S[i];   // lower triangles are the ones sym_get reads
        for (int i = tid; i < Q * D * R; i += nt) s_A[i] = th[D + i];
        __syncthreads();
    }
    // the 2Q frequency / length-scale gradients are D(D+1)/2-term contractions  sum B_q o SM_q,  sum B_q o SV_q : one wave
    // each (lanes stride over the bins, fixed butterfly) instead of one thread each -- left to single threads they were
    // the critical path of this kernel (300 dependent load pairs at D = 24 while 246 threads idled)
    __shared__ double smuv[64];
    const bool par_muv = flag_grad && L.kidx == 7 && own_muv && 2 * Q <= 64;
    if (par_muv) {
        const int nbins = D * (D + 1) / 2, lane = tid & 63, nwave = nt >> 6;
        for (int w2 = tid >> 6; w2 < 2 * Q; w2 += nwave) {
            const int q = (w2 < Q) ? w2 : w2 - Q;
            const double *X = ((w2 < Q) ? SM : SV) + (size_t)q * D * D, *Bq = B + (size_t)q * D * D;
            double sacc = 0.0;
            // (the bin index is decoded once and then advanced: tile_decode -- a double square root and two correction loops --
            //  per term was 26 of the 62 us this kernel took at D = 64; same terms in the same order)
            int d, e;
            tile_decode(lane, d, e);
            for (int idx = lane; idx < nbins; idx += 64) {
                sacc += ((d == e) ? 1.0 : 2.0) * Bq[d * D + e] * X[d * D + e];
                e += 64;
                while (e > d) { e -= d + 1; d++; }
            }
            for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
            if (lane == 0) smuv[w2] = sacc;
        }
        __syncthreads();
    }
    // the D noise gradients are sums of diag(W) over the observations of one output: one wave each (lanes stride over the segment,
    // fixed butterfly) -- a single thread per output walked N / D dependent loads (85 at N = 2048, D = 24: 40 of the 46 us this
    // kernel took for one such evaluation)
    __shared__ double s_sig[MEDGP_MAX_D];
    const bool par_sig = flag_grad && L.kidx == 7 && h0 < D;
    if (par_sig) {
        const int lane = tid & 63, nwave = nt >> 6, dhi = min(D, h1);
        for (int d = h0 + (tid >> 6); d < dhi; d += nwave) {
            double sacc = 0.0;
            for (int i = seg[d] + lane; i < seg[d + 1]; i += 64) sacc += Wd[(size_t)i * wds];
            for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
            if (lane == 0) s_sig[d] = sacc;
        }
        __syncthreads();
    }
    // a caller-order copy of a patient (slot + max_slots, nlml-only evaluations) shares the prior of its patient
    const int pslot = slot >= L.max_slots ? slot - L.max_slots : slot;
    const MedgpPrior *pr = L.prior_on[pslot] ? L.prior + (size_t)pslot * H : nullptr;
    double lp_part = 0.0;   // thread 0: sum of this part's chunk sums, in chunk order
  for (int ch = ch0; ch < ch1; ch++) {
    double lp_local = 0.0;
    const int hc1 = min(H, (ch + 1) * hchunk);
    for (int h = ch * hchunk + tid; h < hc1; h += nt) {
        double gv = 0.0, hv;   // hv = transformed hyper value (what the prior is evaluated at)
        if (L.kidx == 7) {
            if (h < D) {
                hv = exp(th[h]);
                if (flag_grad) gv = (hv * hv) * s_sig[h];   // ref: c_inference_exact.cpp:194-202
            } else {
                int hc = h - D;
                if (hc < Q * D * R) {
                    hv = th[h];
                    if (flag_grad) {
                        int q = hc / (D * R), rem = hc - q * D * R, d = rem / R, r = rem - d * R;
                        double s = 0.0;
                        if (lds_sa) {
                            const double *A = s_A + q * D * R, *Sq = s_S + q * D * D;
                            for (int e = 0; e < D; e++) s += sym_get(Sq, D, d, e) * A[e * R + r];
                        } else {
                            // (S_q does not fit the LDS copy, D > 28 at Q = 5: one load per term through a selected INDEX and eight
                            //  terms in flight -- with a branch per term the 2 D loads of a dot product came back one at a time:
                            //  45 of the 62 us of this kernel at D = 64)
                            const double *A = th + D + (size_t)q * D * R;
                            const double *Sq = S + (size_t)q * D * D;
                            int e = 0;
                            for (; e + 8 <= D; e += 8) {
                                double sv[8], av[8];
#pragma unroll
                                for (int u = 0; u < 8; u++) {
                                    const int ee = e + u, ix = (d >= ee) ? d * D + ee : ee * D + d;
                                    sv[u] = Sq[ix];
                                    av[u] = A[ee * R + r];
                                }
#pragma unroll
                                for (int u = 0; u < 8; u++) s += sv[u] * av[u];
                            }
                            for (; e < D; e++) s += sym_get(Sq, D, d, e) * A[e * R + r];
                        }
                        gv = s;
                    }
                } else if (hc < Q * (D * R + 2)) {
                    hv = exp(th[h]);
                    if (flag_grad) {
                        bool is_mu = hc < Q * (D * R + 1);
                        int q = is_mu ? hc - Q * D * R : hc - Q * (D * R + 1);
                        if (par_muv) gv = 0.5 * smuv[is_mu ? q : Q + q];
                        else {
                            const double *X = (is_mu ? SM : SV) + (size_t)q * D * D, *Bq = B + (size_t)q * D * D;
                            double s = 0.0;
                            for (int d = 0; d < D; d++)
                                for (int e = 0; e <= d; e++) s += ((d == e) ? 1.0 : 2.0) * Bq[d * D + e] * X[d * D + e];
                            gv = 0.5 * s;
                        }
                    }
                } else {
                    hv = exp(th[h]);
                    if (flag_grad) {
                        int kk = hc - Q * (D * R + 2), q = kk / D, d = kk - q * D;
                        gv = 0.5 * hv * S[(size_t)q * D * D + d * D + d];
                    }
                }
            }
        } else if (L.kidx == 8) {   // SM: [log sigma | log w | log mu | log v]; blocks are 1x1
            hv = exp(th[h]);
            if (flag_grad) {
                if (h == 0) {
                    double s = 0.0;
                    for (int i = 0; i < n; i++) s += hv * hv * Wd[(size_t)i * wds];
                    gv = s;
                } else {
                    int hc = h - 1, mode = hc / Q, q = hc - mode * Q;
                    const double *X = mode == 0 ? S : (mode == 1 ? SM : SV);
                    gv = 0.5 * B[q] * X[q];
                }
            }
        } else {                     // SE: [log sigma | log l | log sf]
            hv = exp(th[h]);
            if (flag_grad) {
                if (h == 0) {
                    double s = 0.0;
                    for (int i = 0; i < n; i++) s += hv * hv * Wd[(size_t)i * wds];
                    gv = s;
                } else if (h == 1) gv = -0.5 * B[0] * SV[0];   // d/dlog l = -d/dlog v  (ref: c_kernel_SE.cpp:106-118)
                else gv = B[0] * S[0];                         // ref: c_kernel_SE.cpp:120-131
            }
        }
        if (pr) prior_apply(pr[h], hv, L.pi, flag_grad != 0, lp_local, gv);
        if (flag_grad && g) g[h] = gv;
    }
    red2[ch - ch0][tid] = lp_local;   // (own slot: no barrier between the chunks)
  }
    // deterministic reduction of every chunk's prior log-density: ONE tree for all chunks of the part (a tree per chunk put
    // 10 barriers between two chunks, 40 of them at the headline shape's five chunks; same tree per chunk -> same bits)
    {
        const int nc = ch1 - ch0;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off)
                for (int c2 = 0; c2 < nc; c2++) red2[c2][tid] += red2[c2][tid + off];
            __syncthreads();
        }
        if (tid == 0)
            for (int c2 = 0; c2 < nc; c2++) {
                if (nparts > 1) __hip_atomic_store(&L.epi_lp[(size_t)b * MEDGP_EPI_PARTS + ch0 + c2], red2[c2][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lp_part += red2[c2][0];
            }
    }
    if (tid == 0) {
        double lp = lp_part;
        if (nparts > 1) {
            // the chunk sums are added in chunk order by whichever part arrives last (agent-scope atomics: the parts of an entry
            // may run on different XCDs): the result depends neither on the arrival order nor on the number of parts
            __threadfence();
            const int ticket = __hip_atomic_fetch_add(&L.epi_ticket[b], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket != nparts - 1) return;
            __threadfence();
            lp = 0.0;
            for (int c2 = 0; c2 < nchunk; c2++)
                lp += __hip_atomic_load(&L.epi_lp[(size_t)b * MEDGP_EPI_PARTS + c2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&L.epi_ticket[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double logdet = L.scal[b * 4 + 0], quad = L.scal[b * 4 + 1];
        double nlml = quad / 2.0 + logdet + n * log(2. * L.pi) / 2.0;   // ref: c_inference_exact.cpp:149-152
        nlml_out[pos] = nlml - lp;
    }
}
