// kernels_cohort.h -- cohort statistics: the KDE "mode" of many independent sample series.
//
// Replaces compute_kde + compute_mode (ref: medgpc/clustering/mode_estimate.py:438-450) as output_mode_LMC_SM calls them
// (ref: :277-279 nuggets, :339-340 / :350-351 mu and v of a cluster, :410-413 every element of the aggregated B matrices):
//     kde = statsmodels KDEUnivariate(data).fit(kernel="gau", bw="silverman");  dens = kde.evaluate(data)
//     mode = nansum(data * dens) / nansum(dens)            (weighted)      |     data[argmax(dens)]      (not weighted)
// statsmodels is a third-party dependency that is absent from /root/reference and from this image; its published
// algorithm (statsmodels/nonparametric/bandwidths.py: bw_silverman, _select_sigma; kernels.py: Gaussian, CustomKernel.density):
//     A  = min(std(x, ddof=1), IQR / 1.349)  if IQR > 0 else std      IQR = scoreatpercentile(x, 75) - scoreatpercentile(x, 25)
//     h  = 0.9 A n^(-1/5)                                             (scipy scoreatpercentile: linear interpolation at q (n-1))
//     dens(x) = 1/h * mean_j phi((x_j - x) / h),  phi(u) = 0.3989422804014327 exp(-u^2 / 2)
// Four launches, so that a handful of long series still fills 256 CUs (one workgroup per series left 326 series of 4096
// samples at 23 % of the VALU roofline: 1.3 waves of workgroups, one wave per SIMD):
//   k_kde_stats  (series)            mean, std (ddof 1), finiteness
//   k_kde_rank   (series x chunks)   rank of each of the chunk's 256 samples among all n (the two percentiles are order
//                                    statistics -- no sort); the four samples with the wanted ranks are written out
//   k_kde_dens   (series x chunks)   bandwidth from the statistics; density at the chunk's 256 samples (thread = sample,
//                                    all n samples staged through LDS in tiles, one fp64 exp per pair -- VALU bound);
//                                    per-chunk partial sums of x dens and dens (and the chunk's arg max).  The evaluation
//                                    points can also be a grid of the series' own (output_mode_SE / _SM evaluate the
//                                    length-scale / period densities on 100001-point grids, ref: :54-57, :170-184)
//   k_kde_final  (series)            the chunks' partials added in chunk order
// All sums run in a fixed order: results are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_assemble.h"   // exp2_nonpos

#define KDE_THREADS 256
#define KDE_TILE 2048

// fixed-order workgroup sum (every thread gets the result)
__device__ inline double kde_wg_sum(double v, double *red, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < KDE_THREADS / 64; w++) s += red[w];
    return s;
}

// per-series scratch: st[0] mean, [1] std, [2] bad count, [3..6] order statistics lo25 hi25 lo75 hi75, [7] bandwidth
#define KDE_NSTAT 8

__global__ void __launch_bounds__(KDE_THREADS) k_kde_stats(int nseries, const long long *off, const int *cnt, const double *data, double *stats) {
    __shared__ double red[KDE_THREADS / 64];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int n = cnt[s];
    const double *x = data + off[s];
    double p = 0.0, bad = 0.0;
    for (int i = tid; i < n; i += KDE_THREADS) { const double v = x[i]; p += v; if (!(fabs(v) <= 1.79769313486231570815e308)) bad += 1.0; }
    const double mean = (n > 0) ? kde_wg_sum(p, red, tid) / n : 0.0;
    const double nbad = kde_wg_sum(bad, red, tid);
    p = 0.0;
    for (int i = tid; i < n; i += KDE_THREADS) { const double d = x[i] - mean; p += d * d; }
    const double ss = kde_wg_sum(p, red, tid);
    if (tid == 0) {
        double *st = stats + (size_t)s * KDE_NSTAT;
        st[0] = mean; st[1] = (n > 1) ? sqrt(ss / (n - 1)) : 0.0; st[2] = nbad + (n < 2 ? 1.0 : 0.0);
    }
}

// q (n-1) = k + f: percentile = x_(k) + f (x_(k+1) - x_(k))
__device__ inline void kde_quartile_ranks(int n, int &k25, int &k25h, int &k75, int &k75h, double &f25, double &f75) {
    const double q25 = 0.25 * (n - 1), q75 = 0.75 * (n - 1);
    k25 = (int)floor(q25); k75 = (int)floor(q75);
    k25h = min(k25 + 1, n - 1); k75h = min(k75 + 1, n - 1);
    f25 = q25 - k25; f75 = q75 - k75;
}

__global__ void __launch_bounds__(KDE_THREADS) k_kde_rank(const long long *off, const int *cnt, const double *data, double *stats) {
    __shared__ double tile[KDE_TILE];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int n = cnt[s], i = blockIdx.y * KDE_THREADS + tid;
    if (blockIdx.y * KDE_THREADS >= n) return;
    double *st = stats + (size_t)s * KDE_NSTAT;
    if (st[2] > 0.0) return;
    const double *x = data + off[s];
    const double xi = (i < n) ? x[i] : 0.0;
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += KDE_TILE) {
        const int m = min(KDE_TILE, n - j0);
        __syncthreads();
        for (int j = tid; j < m; j += KDE_THREADS) tile[j] = x[j0 + j];
        __syncthreads();
        if (i < n) {
#pragma unroll 8
            for (int j = 0; j < m; j++) {
                const double xj = tile[j];
                rank += (xj < xi) || (xj == xi && (j0 + j) < i);   // ties ordered by index: the ranks are a permutation
            }
        }
    }
    if (i < n) {
        int k25, k25h, k75, k75h; double f25, f75;
        kde_quartile_ranks(n, k25, k25h, k75, k75h, f25, f75);
        if (rank == k25) st[3] = xi;
        if (rank == k25h) st[4] = xi;
        if (rank == k75) st[5] = xi;
        if (rank == k75h) st[6] = xi;
    }
}

// part[(s * maxchunks + chunk) * 4 + {0: sum x dens, 1: sum dens, 2: best dens, 3: best index}]
// Evaluation points: the samples themselves, or (tcnt != nullptr and tcnt[s] > 0) the series' own grid test[toff[s] ..)
__global__ void __launch_bounds__(KDE_THREADS) k_kde_dens(const long long *off, const int *cnt, const double *data, const long long *toff,
                                                          const int *tcnt, const double *test, double *stats, double *part, int maxchunks) {
    __shared__ double tile[KDE_TILE];
    __shared__ double red[KDE_THREADS / 64];
    __shared__ double bestd[KDE_THREADS];
    __shared__ int besti[KDE_THREADS];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int n = cnt[s], i = blockIdx.y * KDE_THREADS + tid;
    const bool grid = tcnt && tcnt[s] > 0;
    const int nt = grid ? tcnt[s] : n;
    if (blockIdx.y * KDE_THREADS >= nt) return;
    double *st = stats + (size_t)s * KDE_NSTAT;
    if (st[2] > 0.0) return;
    int k25, k25h, k75, k75h; double f25, f75;
    kde_quartile_ranks(n, k25, k25h, k75, k75h, f25, f75);
    const double p25 = st[3] + f25 * (st[4] - st[3]), p75 = st[5] + f75 * (st[6] - st[5]);
    const double iqr = (p75 - p25) / 1.349, sd = st[1];
    const double A = (iqr > 0.0) ? fmin(sd, iqr) : sd;
    const double h = 0.9 * A * pow((double)n, -0.2);
    if (blockIdx.y == 0 && tid == 0) st[7] = h;
    if (!(h > 0.0)) return;
    const double *x = data + off[s];
    const double c2 = -0.5 * MEDGP_LOG2E / (h * h);            // exp(-u^2/2) = 2^(c2 (xj - xi)^2)
    const double norm = 0.3989422804014327 / (h * (double)n);
    const double *xt = grid ? test + toff[s] : x;
    const double xi = (i < nt) ? xt[i] : 0.0;
    double acc = 0.0;
    for (int j0 = 0; j0 < n; j0 += KDE_TILE) {
        const int m = min(KDE_TILE, n - j0);
        __syncthreads();
        for (int j = tid; j < m; j += KDE_THREADS) tile[j] = x[j0 + j];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < m; j++) {
            const double d = tile[j] - xi;
            acc += exp2_nonpos(c2 * (d * d));
        }
    }
    const double dens = (i < nt) ? acc * norm : 0.0;
    const double a = kde_wg_sum(xi * dens, red, tid), b = kde_wg_sum(dens, red, tid);
    bestd[tid] = (i < nt) ? dens : -1.0; besti[tid] = i;
    __syncthreads();
    if (tid == 0) {
        double md = -1.0; int mi = 0;
        for (int t = 0; t < KDE_THREADS; t++) if (bestd[t] > md) { md = bestd[t]; mi = besti[t]; }   // first maximum
        double *o = part + ((size_t)s * maxchunks + blockIdx.y) * 4;
        o[0] = a; o[1] = b; o[2] = md; o[3] = (double)mi;
    }
}

// status: 0 ok; -1 not finite / n < 2 / zero bandwidth (the reference's KDEUnivariate.fit raises there and
// output_mode_kernel exits, ref: mode_estimate.py:23-26)
__global__ void k_kde_final(int nseries, const long long *off, const int *cnt, const double *data, const long long *toff, const int *tcnt,
                            const double *test, const double *stats, const double *part, int maxchunks, int weighted, double *mode,
                            double *bw, int *status) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseries) return;
    const double *st = stats + (size_t)s * KDE_NSTAT;
    const int n = cnt[s];
    const bool bad = st[2] > 0.0;
    const double h = bad ? nan("") : st[7];
    if (bw) bw[s] = h;
    if (bad || !(h > 0.0)) { mode[s] = nan(""); status[s] = -1; return; }
    const bool grid = tcnt && tcnt[s] > 0;
    const int nt = grid ? tcnt[s] : n;
    const double *xt = grid ? test + toff[s] : data + off[s];
    const int nch = (nt + KDE_THREADS - 1) / KDE_THREADS;
    double a = 0.0, b = 0.0, md = -1.0; int mi = 0;
    for (int c = 0; c < nch; c++) {
        const double *o = part + ((size_t)s * maxchunks + c) * 4;
        a += o[0]; b += o[1];
        if (o[2] > md) { md = o[2]; mi = (int)o[3]; }
    }
    mode[s] = weighted ? a / b : xt[mi];
    status[s] = 0;
}
