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
// One series = one workgroup of 256 threads.  The O(n^2) pair loop does double duty: it counts the rank of every sample
// (the two percentiles are order statistics -- no sort) in a first sweep and sums the Gaussian terms in a second; the
// samples of a series are staged through LDS in tiles.  All sums run in a fixed order (thread-strided partials, then a tree):
// results are bitwise reproducible.  VALU-bound: one fp64 exp per pair.
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

// status: 0 ok; -1 not finite / n < 2 / zero bandwidth (the reference's KDEUnivariate.fit raises there and
// output_mode_kernel exits, ref: mode_estimate.py:23-26)
__global__ void __launch_bounds__(KDE_THREADS) k_kde_mode(int nseries, const long long *off, const int *cnt, const double *data,
                                                          int weighted, double *mode, double *bw, int *status) {
    __shared__ double tile[KDE_TILE];
    __shared__ double red[KDE_THREADS / 64];
    __shared__ double ostat[4];    // order statistics lo25, hi25, lo75, hi75
    __shared__ double bestd[KDE_THREADS];
    __shared__ int besti[KDE_THREADS];
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= nseries) return;
    const int n = cnt[s];
    const double *x = data + off[s];
    if (n < 2) { if (tid == 0) { mode[s] = nan(""); if (bw) bw[s] = nan(""); status[s] = -1; } return; }

    // ---- mean, std (ddof = 1), finiteness
    double p = 0.0, bad = 0.0;
    for (int i = tid; i < n; i += KDE_THREADS) { const double v = x[i]; p += v; if (!(fabs(v) <= 1.79769313486231570815e308)) bad += 1.0; }
    const double mean = kde_wg_sum(p, red, tid) / n;
    const double nbad = kde_wg_sum(bad, red, tid);
    p = 0.0;
    for (int i = tid; i < n; i += KDE_THREADS) { const double d = x[i] - mean; p += d * d; }
    const double sd = sqrt(kde_wg_sum(p, red, tid) / (n - 1));
    if (nbad > 0.0) { if (tid == 0) { mode[s] = nan(""); if (bw) bw[s] = nan(""); status[s] = -1; } return; }

    // ---- percentiles 25 / 75 by rank counting.  q (n-1) = k + f: value = x_(k) + f (x_(k+1) - x_(k))
    const double q25 = 0.25 * (n - 1), q75 = 0.75 * (n - 1);
    const int k25 = (int)floor(q25), k75 = (int)floor(q75);
    const int k25h = min(k25 + 1, n - 1), k75h = min(k75 + 1, n - 1);
    for (int i0 = 0; i0 < n; i0 += KDE_THREADS) {
        const int i = i0 + tid;
        const double xi = (i < n) ? x[i] : 0.0;
        int rank = 0;
        for (int j0 = 0; j0 < n; j0 += KDE_TILE) {
            const int m = min(KDE_TILE, n - j0);
            __syncthreads();
            for (int j = tid; j < m; j += KDE_THREADS) tile[j] = x[j0 + j];
            __syncthreads();
            if (i < n) {
                for (int j = 0; j < m; j++) {
                    const double xj = tile[j];
                    rank += (xj < xi) || (xj == xi && (j0 + j) < i);   // ties ordered by index: ranks are a permutation
                }
            }
        }
        if (i < n) {
            if (rank == k25) ostat[0] = xi;
            if (rank == k25h) ostat[1] = xi;
            if (rank == k75) ostat[2] = xi;
            if (rank == k75h) ostat[3] = xi;
        }
    }
    __syncthreads();
    const double p25 = ostat[0] + (q25 - k25) * (ostat[1] - ostat[0]);
    const double p75 = ostat[2] + (q75 - k75) * (ostat[3] - ostat[2]);
    const double iqr = (p75 - p25) / 1.349;
    const double A = (iqr > 0.0) ? fmin(sd, iqr) : sd;
    const double h = 0.9 * A * pow((double)n, -0.2);
    if (bw && tid == 0) bw[s] = h;
    if (!(h > 0.0)) { if (tid == 0) { mode[s] = nan(""); status[s] = -1; } return; }

    // ---- density at every sample and the weighted mean / arg max
    const double c2 = -0.5 * MEDGP_LOG2E / (h * h);            // exp(-u^2/2) = 2^(c2 (xj - xi)^2)
    const double norm = 0.3989422804014327 / (h * (double)n);
    double sxd = 0.0, sdn = 0.0, bd = -1.0;
    int bi = 0x7fffffff;
    for (int i0 = 0; i0 < n; i0 += KDE_THREADS) {
        const int i = i0 + tid;
        const double xi = (i < n) ? x[i] : 0.0;
        double acc = 0.0;
        for (int j0 = 0; j0 < n; j0 += KDE_TILE) {
            const int m = min(KDE_TILE, n - j0);
            __syncthreads();
            for (int j = tid; j < m; j += KDE_THREADS) tile[j] = x[j0 + j];
            __syncthreads();
            if (i < n) {
#pragma unroll 4
                for (int j = 0; j < m; j++) {
                    const double d = tile[j] - xi;
                    acc += exp2_nonpos(c2 * (d * d));
                }
            }
        }
        if (i < n) {
            const double dens = acc * norm;
            sxd += xi * dens;
            sdn += dens;
            if (dens > bd) { bd = dens; bi = i; }   // i ascends per thread: the first maximum is kept
        }
    }
    if (weighted) {
        const double a = kde_wg_sum(sxd, red, tid), b = kde_wg_sum(sdn, red, tid);
        if (tid == 0) { mode[s] = a / b; status[s] = 0; }
    } else {
        bestd[tid] = bd; besti[tid] = bi;
        __syncthreads();
        if (tid == 0) {   // np.argmax: first index of the maximum
            double md = -1.0; int mi = 0x7fffffff;
            for (int t = 0; t < KDE_THREADS; t++)
                if (bestd[t] > md || (bestd[t] == md && besti[t] < mi)) { md = bestd[t]; mi = besti[t]; }
            mode[s] = x[mi]; status[s] = 0;
        }
    }
}
