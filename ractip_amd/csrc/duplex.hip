// duplex.hip -- duplex (hybridization) partition function sweeps for gfx950.
//
// What the reference computes (/root/reference/src/contrafold/DuplexEngine.ipp):
//   ComputeInside 1015-1077, ComputeOutside 1080-1143 (push form), ComputePosterior 1146-1169;
//   same loop nest as /root/reference/src/pf_duplex.c:128-206 (fw/bk), budget l1+l2 <= 28.
//
// How it is organised here:
//   * inside[i][j] depends only on (p < i, q > j): with a = i, b = L2+1-j the cells
//     of anti-diagonal a+b = s depend on diagonals <= s-2, so TWO diagonals are
//     final per launch; outside (pull form) runs the mirrored order in the SAME
//     launch (blockIdx.z), because it does not depend on inside at all;
//   * one 64-lane wavefront per pairable cell, the <=435 loop shapes spread over lanes;
//   * logZ is a separate grid reduction, the posterior exp(in+out-Z) an elementwise pass.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lse.h"
#include "score_model.h"

namespace rh {

namespace {

constexpr uint32_t kPairMask = (1u << (0 * 5 + 3)) | (1u << (3 * 5 + 0)) | (1u << (1 * 5 + 2)) |
                               (1u << (2 * 5 + 1)) | (1u << (2 * 5 + 3)) | (1u << (3 * 5 + 2));
__device__ __forceinline__ bool complementary(int a, int b) { return (kPairMask >> (a * 5 + b)) & 1u; }

__device__ __forceinline__ double tm4(const ScoreModel* M, int a, int b, int c, int d)
{
    return M->terminal_mismatch[((a * 5 + b) * 5 + c) * 5 + d];
}
__device__ __forceinline__ double hs4(const ScoreModel* M, int a, int b, int c, int d)
{
    return M->helix_stacking[((a * 5 + b) * 5 + c) * 5 + d];
}
// LoopScore (DuplexEngine.ipp:974-1012): nucleotide terms of the three special shapes;
// x = first unpaired letter of strand 1 inside the loop, y = last unpaired letter of strand 2
__device__ __forceinline__ double loop_nucs(const ScoreModel* M, int l1, int l2, int x, int y)
{
    double v = 0.0;
    if (l1 == 0 && l2 == 1) v = M->bulge_0x1[y];
    if (l1 == 1 && l2 == 0) v = M->bulge_1x0[x];
    if (l1 == 1 && l2 == 1) v = M->internal_1x1[x * 5 + y];
    return v;
}
// duplex starts at (i,j): DuplexEngine.ipp:1029-1035 (sentinels zero the edge dangles)
__device__ __forceinline__ double open_score(const ScoreModel* M, int i, int j, int L2, int a, int b, int a_m1, int b_p1)
{
    return M->external_unpaired * (double)(i - 1 + L2 - j) + M->dangle_right[b * 25 + a * 5 + a_m1] +
           M->dangle_left[b * 25 + a * 5 + b_p1] + M->base_pair[b * 5 + a] + M->helix_closing[b * 5 + a];
}
// duplex stops at (i,j): DuplexEngine.ipp:1066-1073
__device__ __forceinline__ double close_score(const ScoreModel* M, int i, int j, int L1, int a, int b, int a_p1, int b_m1)
{
    return M->external_unpaired * (double)(L1 - i + j - 1) + M->dangle_left[a * 25 + b * 5 + a_p1] +
           M->dangle_right[a * 25 + b * 5 + b_m1] + M->helix_closing[a * 5 + b];
}

// map a wave index onto the cells of anti-diagonals s0 and s0+1 (a = i, b = L2+1-j, a+b = s)
__device__ __forceinline__ bool diag_cell(int w, int s0, int L1, int L2, int* i, int* j)
{
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int s = s0 + k;
        const int lo = s - L2 > 1 ? s - L2 : 1, hi = s - 1 < L1 ? s - 1 : L1;
        const int cnt = hi - lo + 1;
        if (cnt > 0) {
            if (w < cnt) {
                *i = lo + w;
                *j = L2 + 1 - (s - *i);
                return true;
            }
            w -= cnt;
        }
    }
    return false;
}

}  // namespace

// ---------------------------------------------------------------------------------
// launch t: inside diagonals {2+2t, 3+2t} (blockIdx.z = 0) and outside diagonals
// {Smax-2t-1, Smax-2t} (blockIdx.z = 1), Smax = L1+L2.
__global__ __launch_bounds__(256) void dx_sweep_diag(DxBatch B, const ScoreModel* __restrict__ M, int t)
{
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
    const bool outside = blockIdx.z != 0;
    const int s0 = outside ? (L1 + L2) - 2 * t - 1 : 2 + 2 * t;
    if (s0 + 1 < 2 || s0 > L1 + L2) return;
    int i, j;
    if (!diag_cell(w, s0, L1, L2, &i, &j)) return;

    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const size_t ts = B.tab_stride;
    const int ldd = B.ldd;
    const size_t ij = (size_t)i * ldd + j;

    const int a = s1[i], b = s2[j], a_m1 = s1[i - 1], a_p1 = s1[i + 1], b_m1 = s2[j - 1], b_p1 = s2[j + 1];
    if (!complementary(a, b)) {
        if (lane == 0) {
            if (!outside) { tab[D_IN * ts + ij] = kNeg; tab[D_INX * ts + ij] = kNeg; }
            else          { tab[D_OUT * ts + ij] = kNeg; tab[D_OUTX * ts + ij] = kNeg; }
        }
        return;
    }
    const double bp_ab = M->base_pair[a * 5 + b];
    Lse acc = lse_empty();

    if (!outside) {
        // inside[i][j] = open(i,j) (+) (+)_{p<i,q>j} inside[p][q] + step(p,q -> i,j)   (DuplexEngine.ipp:1029-1064)
        const double down = tm4(M, b, a, b_p1, a_m1) + bp_ab;  // terms of the downstream pair (i,j)
        const int l1max = i - 2, l2max = L2 - j - 1;           // p = i-1-l1 >= 1, q = j+1+l2 <= L2
        if (l1max >= 0 && l2max >= 0) {
            const double* __restrict__ inx = tab + D_INX * ts + (size_t)(i - 1) * ldd + (j + 1);
            Shape sh[kDxShapeIters];
#pragma unroll
            for (int u = 0; u < kDxShapeIters; u++) sh[u] = M->dx_shape[64 * u + lane];
            double x[kDxShapeIters];
#pragma unroll
            for (int u = 0; u < kDxShapeIters; u++) {
                const bool ok = sh[u].l1 <= l1max && sh[u].l2 <= l2max;
                x[u] = ok ? inx[sh[u].l2 - (ptrdiff_t)sh[u].l1 * ldd] : kEmptyMax;
            }
#pragma unroll
            for (int u = 0; u < kDxShapeIters; u++) x[u] += down;
            if (sh[0].l1 <= 1 && sh[0].l2 <= 1 && sh[0].l1 <= l1max && sh[0].l2 <= l2max)
                x[0] += loop_nucs(M, sh[0].l1, sh[0].l2, s1[i - sh[0].l1], s2[j + sh[0].l2]);  // s1[p+1], s2[q-1]
            if (lane == 0)  // stacked on (i-1,j+1)
                x[0] = tab[D_IN * ts + (size_t)(i - 1) * ldd + (j + 1)] + bp_ab + hs4(M, a_m1, b_p1, a, b);
            lse_add_group<kDxShapeIters>(acc, x);
        }
        if (lane == 0) lse_add(acc, open_score(M, i, j, L2, a, b, a_m1, b_p1));
        const double v = lse_wave_finish(acc);
        if (lane == 0) {
            tab[D_IN * ts + ij] = v;
            tab[D_INX * ts + ij] = v + tm4(M, a, b, a_p1, b_m1);  // as the upstream pair of a later loop
        }
    } else {
        // outside[p][q] = close(p,q) (+) (+)_{i>p,j<q} outside[i][j] + step(p,q -> i,j)  (DuplexEngine.ipp:1094-1129, pulled)
        // here (i,j) is the TARGET (p,q) of the reference's loop nest
        const double up = tm4(M, a, b, a_p1, b_m1);  // terms of the upstream pair (this cell)
        const int l1max = L1 - i - 1, l2max = j - 2;  // ii = i+1+l1 <= L1, jj = j-1-l2 >= 1
        if (l1max >= 0 && l2max >= 0) {
            const double* __restrict__ outx = tab + D_OUTX * ts + (size_t)(i + 1) * ldd + (j - 1);
            Shape sh[kDxShapeIters];
#pragma unroll
            for (int u = 0; u < kDxShapeIters; u++) sh[u] = M->dx_shape[64 * u + lane];
            double x[kDxShapeIters];
#pragma unroll
            for (int u = 0; u < kDxShapeIters; u++) {
                const bool ok = sh[u].l1 <= l1max && sh[u].l2 <= l2max;
                x[u] = ok ? outx[(ptrdiff_t)sh[u].l1 * ldd - sh[u].l2] : kEmptyMax;
            }
#pragma unroll
            for (int u = 0; u < kDxShapeIters; u++) x[u] += up;
            x[0] += loop_nucs(M, sh[0].l1, sh[0].l2, a_p1, b_m1);
            if (lane == 0) {  // (i+1,j-1) stacked on this pair
                const int aa = s1[i + 1], bb = s2[j - 1];
                x[0] = tab[D_OUT * ts + (size_t)(i + 1) * ldd + (j - 1)] + M->base_pair[aa * 5 + bb] + hs4(M, a, b, aa, bb);
            }
            lse_add_group<kDxShapeIters>(acc, x);
        }
        if (lane == 0) lse_add(acc, close_score(M, i, j, L1, a, b, a_p1, b_m1));
        const double v = lse_wave_finish(acc);
        if (lane == 0) {
            tab[D_OUT * ts + ij] = v;
            tab[D_OUTX * ts + ij] = v + tm4(M, b, a, b_p1, a_m1) + bp_ab;  // as the downstream pair of an earlier loop
        }
    }
}

// ---------------------------------------------------------------------------------
// logZ = (+)_{i,j} inside[i][j] + close(i,j)   (DuplexEngine.ipp:1066-1073); one block per pair
__global__ __launch_bounds__(1024) void dx_logz(DxBatch B, const ScoreModel* __restrict__ M)
{
    __shared__ double sm[16], ss[16];
    const int pr = blockIdx.x;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const double* __restrict__ in = B.tab + (size_t)pr * B.pair_stride + D_IN * B.tab_stride;
    Lse acc = lse_empty();
    const int total = L1 * L2;
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int i = c / L2 + 1, j = c % L2 + 1;
        const int a = s1[i], b = s2[j];
        if (!complementary(a, b)) continue;
        lse_add(acc, in[(size_t)i * B.ldd + j] + close_score(M, i, j, L1, a, b, s1[i + 1], s2[j - 1]));
    }
    const double M1 = wave_max(acc.m);
    const double S1 = wave_sum(acc.s * exp(acc.m - M1));
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wv] = M1; ss[wv] = S1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double mm = kEmptyMax;
        for (int k = 0; k < 16; k++) mm = fmax(mm, sm[k]);
        double s = 0.0;
        for (int k = 0; k < 16; k++) s += ss[k] * exp(sm[k] - mm);
        B.logz[pr] = lse_norm(mm + log(s));
    }
}

// hp[i][j] = exp(inside + outside - logZ)   (DuplexEngine.ipp:1146-1169, ractip.cpp:242-244)
__global__ __launch_bounds__(256) void dx_posterior(DxBatch B)
{
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L1 * L2) return;
    const int i = c / L2 + 1, j = c % L2 + 1;
    const double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const size_t ij = (size_t)i * B.ldd + j;
    const double e = tab[D_IN * B.tab_stride + ij] + tab[D_OUT * B.tab_stride + ij] - B.logz[pr];
    B.hp[(size_t)pr * B.tab_stride + ij] = e > kNeg / 2 ? exp(e) : 0.0;
}

}  // namespace rh
