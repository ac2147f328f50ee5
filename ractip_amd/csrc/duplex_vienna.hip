// duplex_vienna.hip -- pf_duplex forward/backward/posterior with Vienna loop energies in either semantics of vienna_model.h
// (PARITY UNPINNED): ViennaRNA-1.8 LoopEnergy + dangles, /root/reference/src/pf_duplex.c:304-345 (fw), 347-394 (bk, push form ->
// pulled here), 264-277 (pr_duplex), or ViennaRNA-2.x E_IntLoop / E_ExtLoop, the same recurrences at :128-166, 168-206, 88-103.
// The loader folds the difference into the tables (dxE, loop kinds 3 / 4), the kernels do not branch on it.
// Loop budget MAXLOOP = 30 as in that file (the CONTRAfold DuplexEngine uses 28).
// Same organisation as duplex.hip: two anti-diagonals per launch, inside and outside in the same launch, one
// wavefront per pairable cell, log space.  Generic interior loops and long bulges read tables that already carry
// the source cell's own energy terms (mismatchI resp. TerminalAU); the seven small loop shapes with joint tables
// (stack, 1-bulges, int11, int21, int22) are evaluated explicitly.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lse.h"
#include "vienna_model.h"

namespace rh {

enum DxvTable { V_IN = 0, V_INMM, V_INTAU, V_OUT, V_OUTMM, V_OUTTAU, V_COUNT };

namespace {
__device__ __forceinline__ bool diag_cell_v(int w, int s0, int L1, int L2, int* i, int* j)
{
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int s = s0 + k;
        const int lo = s - L2 > 1 ? s - L2 : 1, hi = s - 1 < L1 ? s - 1 : L1;
        const int cnt = hi - lo + 1;
        if (cnt > 0) {
            if (w < cnt) { *i = lo + w; *j = L2 + 1 - (s - *i); return true; }
            w -= cnt;
        }
    }
    return false;
}
}  // namespace

__global__ __launch_bounds__(256) void dxv_sweep_diag(DxBatch B, const ViennaDx* __restrict__ V, int t)
{
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool outside = blockIdx.z != 0;
    const int s0 = outside ? (L1 + L2) - 2 * t - 1 : 2 + 2 * t;
    if (s0 + 1 < 2 || s0 > L1 + L2) return;
    int i, j;
    if (!diag_cell_v(w, s0, L1, L2, &i, &j)) return;
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const size_t ts = B.tab_stride;
    const int ldd = B.ldd;
    const size_t ij = (size_t)i * ldd + j;
    const int x = s1[i], y = s2[j], xm = s1[i - 1], xp = s1[i + 1], ym = s2[j - 1], yp = s2[j + 1];
    const int type = V->ptype[x * 5 + y];
    const int base = outside ? V_OUT : V_IN;
    if (!type) {
        if (lane == 0) { tab[base * ts + ij] = kNeg; tab[(base + 1) * ts + ij] = kNeg; tab[(base + 2) * ts + ij] = kNeg; }
        return;
    }
    const int rt = V->rtype[type];
    const double tau_here = type > 2 ? V->tau : 0.0;
    const double mm_up = V->mmI[type * 25 + xp * 5 + ym];   // mismatchI[type][S1[i+1]][S2[j-1]]: this pair as the upstream end
    const double mm_dn = V->mmI[rt * 25 + yp * 5 + xm];     // mismatchI[rtype[type]][S2[j+1]][S1[i-1]]: as the downstream end
    Lse acc = lse_empty();
    if (!outside) {
        if (lane == 0)   // pf_duplex.c:321-326
            lse_add(acc, V->duplex_init + V->dxE[type * 36 + (i > 1 ? xm : 5) * 6 + (j < L2 ? yp : 5)]);
        double xs[kMcShapeIters];
#pragma unroll
        for (int u = 0; u < kMcShapeIters; u++) {
            const Shape sh = V->shape[64 * u + lane];
            const int kind = V->kind[64 * u + lane];
            const int k = i - 1 - sh.l1, l = j + 1 + sh.l2;
            xs[u] = kEmptyMax;
            if (k >= 1 && l <= L2) {
                const size_t kl = (size_t)k * ldd + l;
                if (kind == 1) xs[u] = tab[V_INMM * ts + kl] + sh.score + mm_dn;
                else if (kind == 2) xs[u] = tab[V_INTAU * ts + kl] + sh.score + tau_here;
                else {
                    const int t2 = V->ptype[s1[k] * 5 + s2[l]];
                    if (t2 && kind >= 3) {   // 2.x E_IntLoop: 1xn loops score mismatch1nI, 2x3 loops mismatch23I (pf_duplex.c:153)
                        const double* __restrict__ mm = kind == 3 ? V->mm1nI : V->mm23I;
                        xs[u] = tab[V_IN * ts + kl] + sh.score + mm[t2 * 25 + s1[k + 1] * 5 + s2[l - 1]] + mm[rt * 25 + yp * 5 + xm];
                    } else if (t2) xs[u] = tab[V_IN * ts + kl] + vienna_small_loop(V, sh.l1, sh.l2, t2, rt, s1[k + 1], s2[l - 1], xm, yp);
                }
            }
        }
        lse_add_group<kMcShapeIters>(acc, xs);
        const double v = lse_wave_finish(acc);
        if (lane == 0) { tab[V_IN * ts + ij] = v; tab[V_INMM * ts + ij] = v + mm_up; tab[V_INTAU * ts + ij] = v + tau_here; }
    } else {
        if (lane == 0)   // pf_duplex.c:361-365 (close term)
            lse_add(acc, V->dxE[rt * 36 + (j > 1 ? ym : 5) * 6 + (i < L1 ? xp : 5)]);
        double xs[kMcShapeIters];
#pragma unroll
        for (int u = 0; u < kMcShapeIters; u++) {
            const Shape sh = V->shape[64 * u + lane];
            const int kind = V->kind[64 * u + lane];
            const int ii = i + 1 + sh.l1, jj = j - 1 - sh.l2;
            xs[u] = kEmptyMax;
            if (ii <= L1 && jj >= 1) {
                const size_t kl = (size_t)ii * ldd + jj;
                if (kind == 1) xs[u] = tab[V_OUTMM * ts + kl] + sh.score + mm_up;
                else if (kind == 2) xs[u] = tab[V_OUTTAU * ts + kl] + sh.score + tau_here;
                else {
                    const int t2 = V->ptype[s1[ii] * 5 + s2[jj]];
                    if (t2 && kind >= 3) {
                        const double* __restrict__ mm = kind == 3 ? V->mm1nI : V->mm23I;
                        xs[u] = tab[V_OUT * ts + kl] + sh.score + mm[type * 25 + xp * 5 + ym] + mm[V->rtype[t2] * 25 + s2[jj + 1] * 5 + s1[ii - 1]];
                    } else if (t2) xs[u] = tab[V_OUT * ts + kl] + vienna_small_loop(V, sh.l1, sh.l2, type, V->rtype[t2], xp, ym, s1[ii - 1], s2[jj + 1]);
                }
            }
        }
        lse_add_group<kMcShapeIters>(acc, xs);
        const double v = lse_wave_finish(acc);
        if (lane == 0) { tab[V_OUT * ts + ij] = v; tab[V_OUTMM * ts + ij] = v + mm_dn; tab[V_OUTTAU * ts + ij] = v + tau_here; }
    }
}

// Esum = (+)_{i,j} fw[i][j] - E_close(i,j)*10/kT  (pf_duplex.c:337-341); one workgroup per pair
__global__ __launch_bounds__(1024) void dxv_logz(DxBatch B, const ViennaDx* __restrict__ V)
{
    __shared__ double sm[16], ss[16];
    const int pr = blockIdx.x;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const double* __restrict__ in = B.tab + (size_t)pr * B.pair_stride + V_IN * B.tab_stride;
    Lse acc = lse_empty();
    for (int c = threadIdx.x; c < L1 * L2; c += blockDim.x) {
        const int i = c / L2 + 1, j = c % L2 + 1;
        const int type = V->ptype[s1[i] * 5 + s2[j]];
        if (!type) continue;
        const int rt = V->rtype[type];
        lse_add(acc, in[(size_t)i * B.ldd + j] + V->dxE[rt * 36 + (j > 1 ? s2[j - 1] : 5) * 6 + (i < L1 ? s1[i + 1] : 5)]);
    }
    const double M1 = wave_max(acc.m);
    const double S1 = wave_sum(acc.s * exp(acc.m - M1));
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = M1; ss[threadIdx.x >> 6] = S1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double mm = kEmptyMax, s = 0.0;
        for (int k = 0; k < 16; k++) mm = fmax(mm, sm[k]);
        for (int k = 0; k < 16; k++) s += ss[k] * exp(sm[k] - mm);
        B.logz[pr] = lse_norm(mm + log(s));
    }
}

// pr_duplex[i][j] = exp(fw + bk - Esum)   (pf_duplex.c:264-277)
__global__ __launch_bounds__(256) void dxv_posterior(DxBatch B)
{
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L1 * L2) return;
    const int i = c / L2 + 1, j = c % L2 + 1;
    const double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const size_t ij = (size_t)i * B.ldd + j;
    const double e = tab[V_IN * B.tab_stride + ij] + tab[V_OUT * B.tab_stride + ij] - B.logz[pr];
    B.hp[(size_t)pr * B.tab_stride + ij] = e > kNeg / 2 ? exp(e) : 0.0;
}

}  // namespace rh
