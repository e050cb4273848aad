// kernels_wgrad.h -- fused  W = K^-1 - alpha alpha^T  (never written to HBM) + gradient block reductions.
//
// Replaces the reference's  Q = L^-T L^-1 - alpha alpha^T  (two sgemm, ref: inference/c_inference_exact.cpp:168-172)
// and the per-hyper loop of c_kernel_LMC_SM::compute_self_gradients (ref: kernel/c_kernel_LMC_SM.cpp:198-327),
// regrouped into block sums:
//     S_q [d,e] = sum_{i in d, j in e} W_ij k_q(t_i - t_j)
//     SM_q[d,e] = sum W_ij * ( -(w_q dt) sin(w_q dt) E_q )         (d k_q / d log mu_q,  ref :379-384)
//     SV_q[d,e] = sum W_ij * ( -2 c_q dt^2 k_q )                   (d k_q / d log v_q,   ref :385-391)
// One workgroup (4 waves) per lower 64x64 tile (I, J) of one patient:
//   phase 1  W tile = U[I rows] U[J rows]^T on fp64 MFMA (U = L^-T from k_cholinv; k runs over columns >= 64 I);
//            the J-row fragments are staged through LDS (shared by the 4 waves), the I-row fragments stream
//            from HBM with 16-byte loads.
//   phase 2  tile -> LDS (aliases the staging buffers).
//   phase 3  lane = column j, wave = 16-row group: per element one fp64 exp per mixture component (cos/sin of the
//            time difference come from the per-observation tables by the angle-difference identity).
//            Observations are grouped by output, so the output of the row changes rarely: running sums are
//            flushed at each change and reduced over the column segments in a FIXED order -> bitwise
//            reproducible; every (16-row piece of d) x (column tile piece of e) owns one slab entry, written
//            exactly once (no atomics).  k_epilogue adds the pieces in fixed order.
#pragma once
#include "medgp_dev.h"
#include "kernels_assemble.h"

#define WG_KC 32
#define WG_THREADS 256
#ifndef WG_MINWAVES
#define WG_MINWAVES 3
#endif

// one chunk of the MFMA block of phase 1: rows 16w.. (A fragment ac, streamed) x the four column tiles of the staged J rows
#define WG_CHUNK_MFMA(ac, buf) \
    _Pragma("unroll") for (int h = 0; h < 4; h++) \
        _Pragma("unroll") for (int ct = 0; ct < 4; ct++) { \
            if (ct > ctmax) continue;   /* diagonal tile: columns right of this wave's rows are never used */ \
            const v2d bf = *(const v2d *)&Bs[buf][16 * ct + li][8 * h + 2 * g]; \
            _Pragma("unroll") for (int s = 0; s < 2; s++) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(ac[h][s], bf[s], acc[ct], 0, 0, 0); \
        }

// QT = mixture components this launch reduces, Q0 = index of the first one (Q <= 8: one launch <Q, 0>; 8 < Q <= 16: <8, 0> and
// <Q - 8, 8>, each forming the W tile again -- N^3/3 more MFMA work for a route no BASELINE config takes).
// PF = chunks the phase-1 loop prefetches ahead.  PF = 2 for launches of few large patients (the chip is filled less than four times): a
// workgroup then shares its SIMDs with at most a few others and little hides the latency of its operand stream; measured k_wgrad 0.123 ->
// 0.108 ms (1 x N = 2048), 0.072 -> 0.064 ms (4 x N = 1024), 0.449 -> 0.437 ms (1 x N = 4096), neutral on full launches (which keep PF = 1,
// the round-3 code path); four chunks ahead and a branch-free MFMA block for the off-diagonal tiles were both slower (DESIGN_LOG A.7).  Such a
// launch of at most 1024 workgroups (all resident at once, so placement is static: workgroups x, x + 256, x + 512 share a CU -- traced) also
// deals its tiles in serpentine order, so that the CU holding the longest k range gets the shortest one next.  Scheduling only: same MFMA
// sequence per tile, same bits.
// nbp (launches of fewer than 64 entries): stride of the entry index in the workgroup id, nbatch or nbatch | 1.  Workgroup ids go round-robin
// over the 8 XCDs, so with the entry index fastest entry b's tiles land on 8 / gcd(stride, 8) ... of them: harmless when all entries are
// equally large (every XCD gets the same work: 4 x N = 2048 0.062 ms per entry, 5 x 0.062), but in a RAGGED class of 2, 4 or 8 entries
// the largest entry's tiles -- most of the work -- queue on half, a quarter, an eighth of the chip (round 5, scratch/wgrad_ragged.py:
// N = 3595 + 3 x 2400: 1.18 ms, + 4 x 2400: 0.73 ms).  An odd stride puts every entry on all XCDs; the ids of the padding entry exit.
template <int QT, int Q0 = 0, int PF = 1>
__global__ void __launch_bounds__(WG_THREADS, WG_MINWAVES) k_wgrad(MedgpDev L, int nbatch, int ntiles, int nbp) {
    // staging buffers (phase 1) and the W tile (phase 2/3) share storage
    __shared__ __attribute__((aligned(16))) double smem[2 * 64 * (WG_KC + 2)];
    typedef double (*BsT)[64][WG_KC + 2];
    BsT Bs = (BsT)smem;                       // Bs[2][64][34]
    double (*Ws)[66] = (double (*)[66])smem;   // Ws[64][66]  (4224 <= 4352 doubles)
    __shared__ __attribute__((aligned(16))) double rowc[4][16][2 + 2 * QT];   // row constants of each wave's 16 rows

    // 1-D grid of 8 * ceil(nbatch / 8) * ntiles workgroups, dealt round-robin over the 8 XCDs by the hardware.
    // id -> (patient, tile) keeps a patient on ONE XCD (its U rows are re-read by every tile: they must share an L2) and
    // makes the tiles of a patient consecutive in dispatch order, so that an XCD works on about three patients at a time
    // (3 x 1.2 MB of U fits its 4 MB L2).  Patient-major over the whole batch instead (all patients' tile 0, then tile
    // 1, ...) cycles 64 patients x 1.2 MB through each L2 between two tiles of the same patient: every tile re-read U
    // from HBM (PMC: 3.6 GB fetched per launch for 0.6 GB of U).
    // With fewer than 64 patients that would leave XCDs idle (one patient = one XCD): spread instead, patient index
    // fastest (tiles of a patient then land on XCD b % 8 only when nbatch is a multiple of 8, which no longer matters).
    int b, tix;
    if (nbatch >= 64) {
        const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
        b = (rest / ntiles) * 8 + xcd;
        tix = rest % ntiles;
    } else {
        int x = blockIdx.x;
        const int total = nbp * ntiles;
        if (x >= total) return;
        if (PF > 1 && total <= 1024 && ((x >> 8) & 1)) {   // odd group of 256: reversed
            const int v0 = x & ~255, m = min(256, total - v0);
            x = v0 + (m - 1 - (x & 255));
        }
        b = x % nbp;
        tix = x / nbp;
    }
    if (b >= nbatch) return;
    if (L.status[b] < 0) return;
    const int slot = __builtin_amdgcn_readfirstlane(L.bslot[b]);
    const int n = __builtin_amdgcn_readfirstlane(L.pn[slot]);
    const int ld = L.ldn, npad = medgp_roundup(n, 64), nb = npad / 64;
    int I, J;
    tile_decode(tix, I, J);
    if (I >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double *U = L.Linv + (size_t)b * ld * ld;

#ifdef MEDGP_STAMPS
    unsigned long long wst[4] = {0, 0, 0, 0}, wlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wlast)::"memory");
#define WSTAMP(k) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); wst[k] += t_ - wlast; wlast = t_; } while (0)
#else
#define WSTAMP(k) do {} while (0)
#endif
    // ---------------- phase 1: acc[ct] (rows 16w.. of block I, cols 16ct.. of block J)
    v4d acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ct++) acc[ct] = (v4d){0.0, 0.0, 0.0, 0.0};
    {
        const int k0 = 64 * I, nch = (npad - k0) / WG_KC;
        const int ctmax = (I == J) ? w : 3;   // wave-uniform
        const double *Arow = U + (size_t)(64 * I + 16 * w + li) * ld + k0 + 2 * g;
        const int srow = tid >> 2, scg = (tid & 3) * 8;
        const double *Bsrc = U + (size_t)(64 * J + srow) * ld + k0 + scg;
        if constexpr (PF == 1) {
            v2d bst[4], an[4];
#pragma unroll
            for (int u = 0; u < 4; u++) bst[u] = *(const v2d *)(Bsrc + 2 * u);
#pragma unroll
            for (int h = 0; h < 4; h++) an[h] = *(const v2d *)(Arow + 8 * h);
#pragma unroll
            for (int u = 0; u < 4; u++) *(v2d *)&Bs[0][srow][scg + 2 * u] = bst[u];
            __syncthreads();
            for (int c = 0; c < nch; c++) {
                const int buf = c & 1;
                v2d ac[4];
#pragma unroll
                for (int h = 0; h < 4; h++) ac[h] = an[h];
                if (c + 1 < nch) {
#pragma unroll
                    for (int u = 0; u < 4; u++) bst[u] = *(const v2d *)(Bsrc + (c + 1) * WG_KC + 2 * u);
#pragma unroll
                    for (int h = 0; h < 4; h++) an[h] = *(const v2d *)(Arow + (c + 1) * WG_KC + 8 * h);
                }
                WG_CHUNK_MFMA(ac, buf);
                if (c + 1 < nch) {
#pragma unroll
                    for (int u = 0; u < 4; u++) *(v2d *)&Bs[buf ^ 1][srow][scg + 2 * u] = bst[u];
                }
                __syncthreads();
            }
        } else {
            // PF register sets, set s carries the chunks c = s (mod PF): a set is refilled (chunk c + PF) as soon as its A fragment has been
            // copied out; its staged J rows go to LDS one iteration before they are used (the LDS stays double buffered)
            v2d bst[PF][4], an[PF][4];
#pragma unroll
            for (int st = 0; st < PF; st++)
                if (st < nch) {
#pragma unroll
                    for (int u = 0; u < 4; u++) bst[st][u] = *(const v2d *)(Bsrc + st * WG_KC + 2 * u);
#pragma unroll
                    for (int h = 0; h < 4; h++) an[st][h] = *(const v2d *)(Arow + st * WG_KC + 8 * h);
                }
#pragma unroll
            for (int u = 0; u < 4; u++) *(v2d *)&Bs[0][srow][scg + 2 * u] = bst[0][u];
            __syncthreads();
            for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
                for (int st = 0; st < PF; st++) {
                    const int c = c0 + st;
                    if (c >= nch) break;
                    v2d ac[4];
#pragma unroll
                    for (int h = 0; h < 4; h++) ac[h] = an[st][h];
                    if (c + PF < nch) {
#pragma unroll
                        for (int u = 0; u < 4; u++) bst[st][u] = *(const v2d *)(Bsrc + (c + PF) * WG_KC + 2 * u);
#pragma unroll
                        for (int h = 0; h < 4; h++) an[st][h] = *(const v2d *)(Arow + (c + PF) * WG_KC + 8 * h);
                    }
                    WG_CHUNK_MFMA(ac, (st & 1));
                    if (c + 1 < nch) {
#pragma unroll
                        for (int u = 0; u < 4; u++) *(v2d *)&Bs[(st + 1) & 1][srow][scg + 2 * u] = bst[(st + 1) % PF][u];
                    }
                    __syncthreads();
                }
            }
        }
    }
    WSTAMP(0);
    // ---------------- phase 2: W tile -> LDS (all waves are past the last staging read: barrier above)
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) Ws[16 * w + 4 * r + g][16 * ct + li] = acc[ct][r];
    __syncthreads();

    WSTAMP(1);
    // ---------------- phase 3
    const double *hyp = L.hyp + (size_t)b * L.hyp_stride;
    const double *t = L.pt + (size_t)slot * L.pld;
    const int *meta = L.pmeta + (size_t)slot * L.pld;
    const double *alpha = L.alpha + (size_t)b * ld;
    const int *seg = L.pseg + (size_t)slot * (L.D + 1);
    const int *roff = L.proff + (size_t)slot * (L.D + 1), *coff = L.pcoff + (size_t)slot * (L.D + 1);
    const double *csb = L.cs + ((size_t)b * L.Q + Q0) * ld, *snb = L.sn + ((size_t)b * L.Q + Q0) * ld;
    double *slab = L.slab + (size_t)b * L.slab_stride;
    const int Qall = L.Q;   // the slab planes are [S | SM | SV] x ALL components
    const int Rmax = L.slab_R, Cmax = L.slab_C;

    double wq[QT], cq[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) { wq[q] = hyp[hyp_off_w(L) + Q0 + q]; cq[q] = hyp[hyp_off_c(L) + Q0 + q]; }
    double cq2n[QT];   // -c_q log2(e): exp(-c_q dt^2) = 2^(cq2n dt^2), as in k_assemble_t
#pragma unroll
    for (int q = 0; q < QT; q++) cq2n[q] = uniform_d(-cq[q] * MEDGP_LOG2E);
    // column constants of this lane
    const int j = 64 * J + lane;
    const bool jv = j < n;
    const double tj = t[j], aj = alpha[j];
    const int mj = jv ? meta[j] : -1;
    double csj[QT], snj[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) { csj[q] = csb[q * ld + j]; snj[q] = snb[q * ld + j]; }
    // column segments inside the tile: leader lanes and their segment ends
    const int mprev = __shfl_up(mj, 1);
    const bool leader = (lane == 0) || (mj != mprev);
    const unsigned long long lmask = __ballot(leader);
    int segend;
    {
        unsigned long long above = (lane == 63) ? 0ull : (lmask >> (lane + 1));
        segend = above ? (lane + 1 + __builtin_ctzll(above)) : 64;
    }
    // first lane of this lane's column segment, and whether this lane is its last one (it writes the segment sum)
    const unsigned long long upto = lmask & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int segstart = 63 - __builtin_clzll(upto);
    const bool seglast = (lane == segend - 1);
    const int cslot = (mj >= 0) ? coff[mj] + (J - seg[mj] / 64) : 0;
    const int rg = 4 * I + w;   // global 16-row group of this wave

    double sS[QT], sM[QT], sV[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) { sS[q] = 0.0; sM[q] = 0.0; sV[q] = 0.0; }
    int mcur = -2;
    // row constants of the wave's 16 rows: loaded once (lane r holds row r, lanes >= 16 mirror), instead of 3 + 2Q
    // dependent scalar loads per row iteration
    const int irow = 64 * I + 16 * w + (lane & 15);
    const double r_t = t[irow], r_a = alpha[irow];
    const int r_m = (irow < n) ? meta[irow] : -1;
    double r_cs[QT], r_sn[QT];
#pragma unroll
    for (int q = 0; q < QT; q++) { r_cs[q] = csb[q * ld + irow]; r_sn[q] = snb[q * ld + irow]; }
    // row constants go through LDS: every lane reads the same address (a broadcast on the LDS pipe, no VALU) where two
    // v_readlane per double cost 2 + 4Q VALU per row (measured: k_wgrad 1.12 -> 1.07 ms, and 17 VGPRs fewer)
    if (lane < 16) {
        rowc[w][lane][0] = r_t; rowc[w][lane][1] = r_a;
#pragma unroll
        for (int q = 0; q < QT; q++) { rowc[w][lane][2 + 2 * q] = r_cs[q]; rowc[w][lane][3 + 2 * q] = r_sn[q]; }
    }
    __builtin_amdgcn_wave_barrier();
    for (int rr = 0; rr <= 16; rr++) {
        const int i = 64 * I + 16 * w + rr;
        int mi = -1;
        if (rr < 16) mi = __builtin_amdgcn_readlane(r_m, rr);          // wave-uniform
        if (mi != mcur) {
            // flush the running sums of row output mcur (skip padding / initial state)
            if (mcur >= 0) {
                const int rslot = roff[mcur] + (rg - seg[mcur] / 16);
                // segmented inclusive scan over the lanes (6 shuffle steps, fixed order): the last lane of every column
                // segment ends up with the segment sum.  (A serial per-segment loop here cost as many VALU instructions
                // as the whole pair loop.)  Step-major: one exec-mask region per step covers the adds of all 3 QT values.
                double fv[3 * QT];
#pragma unroll
                for (int q = 0; q < QT; q++) { fv[q] = sS[q]; fv[QT + q] = -wq[q] * sM[q]; fv[2 * QT + q] = -2.0 * cq[q] * sV[q]; }
#pragma unroll
                for (int dlt = 1; dlt < 64; dlt <<= 1) {
                    double up[3 * QT];
#pragma unroll
                    for (int k = 0; k < 3 * QT; k++) up[k] = __shfl_up(fv[k], dlt);
                    if (lane - dlt >= segstart) {
#pragma unroll
                        for (int k = 0; k < 3 * QT; k++) fv[k] += up[k];
                    }
                }
                if (seglast && mj >= 0) {
#pragma unroll
                    for (int k = 0; k < 3 * QT; k++) slab[((size_t)((k / QT) * Qall + Q0 + (k % QT)) * Rmax + rslot) * Cmax + cslot] = fv[k];
                }
            }
#pragma unroll
            for (int q = 0; q < QT; q++) { sS[q] = 0.0; sM[q] = 0.0; sV[q] = 0.0; }
            mcur = mi;
        }
        if (rr == 16 || mi < 0) continue;
        const v2d ta = *(const v2d *)&rowc[w][rr][0];
        const double ti = ta[0], ai = ta[1];
        double wv = Ws[16 * w + rr][lane] - ai * aj;
        if (I == J && j == i) L.wdiag[(size_t)b * ld + i] = wv;   // noise gradient needs diag(W)
        const bool valid = jv && (j <= i);
        wv = valid ? ((mi == mj && j < i) ? 2.0 * wv : wv) : 0.0;
        const double dt = ti - tj, dd = dt * dt;
        // running sums hold  sum w E cd,  sum w E sd dt,  sum w E cd dt^2 ; the constant factors
        // (-w_q, -2 c_q) are applied once per flush -- 3 VALU less per (pair, component)
#pragma unroll
        for (int q = 0; q < QT; q++) {
            const v2d csn = *(const v2d *)&rowc[w][rr][2 + 2 * q];
            const double ci = csn[0], si = csn[1];
            const double we = wv * exp2_nonpos(cq2n[q] * dd);
            const double cd = ci * csj[q] + si * snj[q];
            const double sd = si * csj[q] - ci * snj[q];
            const double p = we * cd;
            sS[q] += p;
            sM[q] += (we * sd) * dt;
            sV[q] += p * dd;
        }
    }
    WSTAMP(2);
#ifdef MEDGP_STAMPS
    if (lane == 0 && b < 64) {
        unsigned long long *dbg = (unsigned long long *)(L.xk + (size_t)b * 64 * 64);
        for (int e = 0; e < 3; e++) atomicAdd(&dbg[e], wst[e]);
        atomicAdd(&dbg[3], 1ull);
    }
#endif
}
