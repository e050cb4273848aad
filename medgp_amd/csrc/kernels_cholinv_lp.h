// kernels_cholinv_lp.h -- the look-ahead factorisation of kernels_cholinv_la.h as ONE persistent launch (few, large patients:
// BASELINE configs 3 and 5).  Same tasks, same arithmetic in the same order (results are bit-identical to the one-launch-per-step
// schedule, which stays as the reference implementation and as the route for larger batches); what changes is who waits for whom:
//   * one CHAIN workgroup per entry runs the D role of every step back to back.  It waits only for the two slabs its step needs
//     (the pre-solve copy + diagonal head start from ONE F task of the previous step, the diagonal look-ahead sum from its R
//     task), never for the bulk of a step, and it pays no launch boundary: 28.5 us -> the chain's own work per step.
//   * every other workgroup pulls tasks (F, R, L of step 0, 1, 2, ...) from ONE ticket counter.  A task polls the completion
//     counters of the step(s) it depends on, so steps overlap: the look-ahead tasks of step k run beside the F tasks of step k
//     and beside the tail of step k-1 instead of behind a kernel boundary.
// Deadlock freedom does not rest on co-residency: a workgroup holds a ticket only while it runs, every dependency of a task
// points to tasks with SMALLER tickets or to the chain workgroups (ids 0 .. nbatch-1, dispatched first), so the oldest unfinished
// task can always proceed.  Every poll is bounded all the same: a timeout marks the entry (status -2) and the launch that follows
// the schedule (k_cholinv<8,4,2>) factors it on one workgroup -- slow, never wrong, never hung.
// Visibility (per-XCD L2s are not coherent, a CU's L1 is never refreshed): MI355X guide, Guideline 16 R1 -- producers store
// write-through (sc1, la_st<true>), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane publishes with an
// agent-scope atomic; consumers poll relaxed, ONE agent-scope acquire, s_waitcnt vmcnt(0), workgroup barrier, then plain loads.
#pragma once
#include "kernels_cholinv_la.h"

#define LP_FAIL 0x40000000u       // chain word of an entry whose factorisation failed (non-positive pivot or timeout)
#define LP_SPIN_MAX (1 << 20)     // polls per wait (>= 1 us each): seconds, far beyond any legitimate wait
#define LP_HDR 16                 // words in front of the per-entry blocks: [0] ticket head, [1] timeout count

#ifdef LP_STAMPS   // diagnostic build (scratch/lp_stamps.py): wall-clock stamps (100 MHz) of the chain and of the key F task per step, in the idle slab of entry 0
#define LP_ST(cond, k, slot) do { if ((cond) && threadIdx.x == 0) ((unsigned long long *)L.slab)[16 * (k) + (slot)] = wall_clock64(); } while (0)
#define LP_T(var) const unsigned long long var = wall_clock64()
#define LP_ADD(cond, k, slot, val) do { if ((cond) && threadIdx.x == 0) atomicAdd((unsigned long long *)L.slab + 16 * (k) + (slot), (unsigned long long)(val)); } while (0)
#define LP_MAX(cond, k, slot, val) do { if ((cond) && threadIdx.x == 0) atomicMax((unsigned long long *)L.slab + 16 * (k) + (slot), (unsigned long long)(val)); } while (0)
#else
#define LP_ST(cond, k, slot) do {} while (0)
#define LP_T(var) do {} while (0)
#define LP_ADD(cond, k, slot, val) do {} while (0)
#define LP_MAX(cond, k, slot, val) do {} while (0)
#endif

struct LpArgs {
    unsigned *sync;       // zeroed before every launch: header, then per entry [chain, pad x3, fdone[nbmax], ldone[nbmax], key[nbmax], rkey[nbmax]]
    int nbatch;
    int total;            // tickets of the launch
    int park;             // first workgroup id that is left empty (the chains' CU neighbours), nbatch of them; < 0: none
};
__device__ __forceinline__ unsigned *lp_entry(const LpArgs &P, const LaArgs &A, int b) { return P.sync + LP_HDR + (size_t)b * (4 + 4 * A.nbmax); }
__device__ __forceinline__ unsigned lp_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// tasks of step k per entry in ticket order: F (nF, row block k+2 first), R, L (slice-major)
__device__ __forceinline__ void lp_counts(const LaArgs &A, int k, int want_inv, int &nF, int &nLrows, int &nsl) {
    const int nMF = A.nbmax - (k + 2) > 0 ? A.nbmax - (k + 2) : 0, nUF = want_inv ? k + 1 : 0, nUL = want_inv ? k : 0;
    nF = nMF + nUF + 1;
    nLrows = (k + 2 < A.nbmax && k >= 1) ? nMF + nUL + 1 : 0;
    const int s = (k + LA_SLICE - 1) / LA_SLICE;
    nsl = s < A.maxslice ? s : A.maxslice;
}

// wave 0: lanes 0..2 poll up to three words until each has reached its value; 0 = reached, 2 = timeout
__device__ __forceinline__ int lp_wait(const unsigned *p0, unsigned v0, const unsigned *p1, unsigned v1, const unsigned *p2, unsigned v2,
                                       int lane) {
    const unsigned *p = lane == 0 ? p0 : (lane == 1 ? p1 : p2);
    const unsigned v = lane == 0 ? v0 : (lane == 1 ? v1 : v2);
    const bool act = lane < 3 && p != nullptr;
    for (int it = 0; it < LP_SPIN_MAX; it++) {
        const unsigned x = act ? lp_ld(p) : v;
        if (__all(x >= v)) return 0;
        __builtin_amdgcn_s_sleep(8);
    }
    return 2;
}
// consumer side of the hand-off, after wave 0's poll: one acquire, drained, then the barrier every reading wave joins
__device__ __forceinline__ void lp_acquire_wg(int w) {
    if (w == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
// producer side: every storing wave drains its write-through stores, barrier; the caller's lane 0 then publishes
__device__ __forceinline__ void lp_drain_wg() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__global__ void __launch_bounds__(LA_THREADS, 2) k_lp_run(MedgpDev L, LaArgs A, LpArgs P, int want_mode) {
    __shared__ LaSmem sm;
    __shared__ unsigned s_ticket;
    __shared__ int s_rc;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    const int want_inv = want_mode & 1;
    // ------------------------------------------------ chain workgroups -------------------------------------------------
    if (wg < P.nbatch) {
        const int b = wg;
        unsigned *E = lp_entry(P, A, b);
        unsigned *key = E + 4 + 2 * A.nbmax, *rkey = E + 4 + 3 * A.nbmax;
        const int st0 = L.status[b], n0 = L.bn[b];
        const int n = __builtin_amdgcn_readfirstlane(n0);
        const int nb = medgp_roundup(n, 64) / 64;
        if (st0 < 0 || nb < 2) { if (tid == 0) __hip_atomic_store(E, LP_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        LaTask T;
        T.role = 0; T.slice = 0; T.diag_ahead = false; T.row.kind = 0;
        for (int k = 0; k + 1 < nb; k++) {
            T.row.blk = k + 1;
            LP_ST(b == 0, k, 0);
            if (k >= 1) {
                if (w == 0) { const int rc = lp_wait(key + (k - 1), 1u, rkey + (k - 1), 1u, nullptr, 0u, lane); if (lane == 0) s_rc = rc; }
                lp_acquire_wg(w);
            } else {
                if (tid == 0) s_rc = 0;
                __syncthreads();
            }
            int bad = s_rc;
            LP_ST(b == 0, k, 1);
            if (!bad && (lp_ld(E) & LP_FAIL)) bad = 1;     // a bulk task of this entry timed out
            if (!bad) bad = la_body<true>(L, A, sm, b, n, k, want_mode, T);
            LP_ST(b == 0, k, 2);
            lp_drain_wg();
            LP_ST(b == 0, k, 3);
            if (bad) {
                if (tid == 0) {
                    if (bad == 2) atomicAdd(P.sync + 1, 1u);
                    __hip_atomic_store(&L.status[b], -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(E, LP_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                return;
            }
            if (tid == 0) atomicMax(E, (unsigned)(k + 1));   // X_k+1, L / U blocks published (max: a bulk task's LP_FAIL stays)
        }
        return;
    }
    if (P.park >= 0 && wg >= P.park && wg < P.park + P.nbatch) return;   // the chains' CU neighbours stay empty
    // ------------------------------------------------ bulk workgroups --------------------------------------------------
    int k = 0, base = 0, nF, nLrows, nsl;
    lp_counts(A, 0, want_inv, nF, nLrows, nsl);
    unsigned next = 0;
    if (tid == 0) next = atomicAdd(P.sync, 1u);
    for (;;) {
        if (tid == 0) s_ticket = next;
        __syncthreads();
        const int tk = (int)s_ticket;
        if (tk >= P.total) break;
        if (tid == 0) next = atomicAdd(P.sync, 1u);       // the next ticket travels while this task runs
        while (tk >= base + (nF + 1 + nLrows * nsl) * P.nbatch) {
            base += (nF + 1 + nLrows * nsl) * P.nbatch;
            k++;
            lp_counts(A, k, want_inv, nF, nLrows, nsl);
        }
        const int idx = tk - base, t = idx / P.nbatch, b = idx - t * P.nbatch;   // task-major: every entry's F tasks precede the L tasks
        unsigned *E = lp_entry(P, A, b);
        unsigned *fdone = E + 4, *ldone = E + 4 + A.nbmax, *key = E + 4 + 2 * A.nbmax, *rkey = E + 4 + 3 * A.nbmax;
        const int n = __builtin_amdgcn_readfirstlane(L.bn[b]);
        const int nb = medgp_roundup(n, 64) / 64;
        LaTask T;
        bool live = la_decode(A, k, want_inv, t + 1, nLrows, T) && k < nb && nb >= 2;
        __builtin_assume(T.role != 0);   // the D role lives in the chain workgroups only
        if (live && T.role != 1 && k + 2 >= nb) live = false;
        if (live && T.row.kind == 0 && T.row.blk >= nb) live = false;
        int rc = 0;
        if (live) {
            // F of step k: X_k (chain >= k), its own pre-solve block / the pre-solve copy / panel k-1 (F and R of step k-1), the partial
            // sums of panel k+1 (L of step k-1).  R of step k: the diagonal look-ahead slices (L of step k-1).  L of step k: panels
            // <= k-1 final and the slabs it overwrites consumed (F and R of step k-1).
            int pF, pL, pS;
            if (k >= 1) { int a, c; lp_counts(A, k - 1, want_inv, pF, a, c); pL = a * c; pS = pF + 1; } else { pF = pL = pS = 0; }
            LP_ST(b == 0 && t == 0, k, 4);
            LP_T(ts0);
            if (w == 0) {
                int r;
                if (T.role == 1) r = lp_wait(E, (unsigned)k, k >= 1 ? fdone + (k - 1) : nullptr, (unsigned)pS, k >= 1 ? ldone + (k - 1) : nullptr, (unsigned)pL, lane);
                else if (T.role == 3) r = lp_wait(k >= 1 ? ldone + (k - 1) : nullptr, (unsigned)pL, nullptr, 0u, nullptr, 0u, lane);
                else r = lp_wait(k >= 1 ? fdone + (k - 1) : nullptr, (unsigned)pS, nullptr, 0u, nullptr, 0u, lane);
                if (r == 0 && (lp_ld(E) & LP_FAIL)) r = 1;    // the entry is lost: nothing to compute, the counters still advance
                if (lane == 0) s_rc = r;
                LP_ST(b == 0 && t == 0, k, 5);
            }
            lp_acquire_wg(w);
            LP_ST(b == 0 && t == 0, k, 6);
            LP_T(ts1);
            rc = s_rc;
            if (rc == 0) la_body<true>(L, A, sm, b, n, k, want_mode, T);
            LP_ST(b == 0 && t == 0, k, 7);
            LP_T(ts2);
            lp_drain_wg();
            LP_T(ts3);
            LP_ADD(b == 0 && T.role == 2, k, 9, 1);
            LP_ADD(b == 0 && T.role == 2, k, 10, ts1 - ts0);
            LP_ADD(b == 0 && T.role == 2, k, 11, ts2 - ts1);
            LP_ADD(b == 0 && T.role == 2, k, 12, ts3 - ts2);
            LP_MAX(b == 0 && T.role == 2, k, 13, ts3);
            LP_MAX(b == 0 && T.role == 1, k, 14, ts3);
            LP_MAX(b == 0 && T.role == 1, k, 15, ts3 - ts1);
        }
        lp_drain_wg();
        LP_ST(live && b == 0 && t == 0, k, 8);
        if (tid == 0) {
            if (rc == 2) {   // timeout: give the entry up (the retry launch factors it), let everybody else run through
                atomicAdd(P.sync + 1, 1u);
                __hip_atomic_store(&L.status[b], -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(E, LP_FAIL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t < nF || t == nF) atomicAdd(fdone + k, 1u); else atomicAdd(ldone + k, 1u);
            if (t == 0) __hip_atomic_store(key + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // F of row block k+2: pnx, dterm of the next chain step
            if (t == nF) __hip_atomic_store(rkey + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // R: dsum
        }
    }
}
