// mccaskill_vlin.hip -- Vienna-BL McCaskill inside / outside / posterior in SCALED LINEAR space (the fast path of the
// default-CLI model; mccaskill_vienna.hip is its log-space form and the fallback).  PARITY UNPINNED, see vienna_model.h.
//
// Same organisation as mccaskill_lin.hip: diagonal-major tables ([d*ld + i] = cell (i, i+d)), one THREAD per cell, 64
// consecutive cells of a diagonal per wavefront, the term loops of a group split over W wavefronts, block products
// (mccaskill_far.hip, unchanged -- it only sees FM1, FM, FM2o) for the far k-terms of the O(n^3) sums.  What differs is
// the model:
//   * loops with one enclosed pair: generic interior loops factor as mismatchI(outer) x w(l1,l2) x mismatchI(inner) and run
//     as LDS-staged filters over the decorated table FCX exactly like the CONTRAfold shapes; bulges of length >= 2 factor
//     as TerminalAU(outer) x w(l) x TerminalAU(inner): two more taps per total length over a second decorated table
//     (FCB), read straight from HBM; the seven small shapes with joint energy tables (stack, 1-bulges, 1x1, 1x2, 2x1,
//     2x2) are gathered per cell from the raw table in the epilogue;
//   * hairpins: length table x mismatchH, tetraloop bonus, TerminalAU for triloops;
//   * the unambiguous multiloop grammar of mccaskill_vienna.hip (FMS = one branch with trailing unpaired letters).
#include <hip/hip_runtime.h>

#include "batch.h"
#include "vienna_model.h"

namespace rh {

namespace {

__device__ __forceinline__ size_t tri_off_vl(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }

__device__ __forceinline__ double wsum_vl(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// pin bits 0..6: sequence on blockIdx.x (1) or blockIdx.y (0); bit 7: windowed grid of the two-molecule sweeps, first cell group in
// bits 8.. (see window_slot_vl)
__device__ __forceinline__ void block_map_vl(int pin, int* sq, int* slot)
{
    *sq = (pin & 127) ? blockIdx.x : blockIdx.y;
    *slot = (pin & 127) ? blockIdx.y : blockIdx.x;
}
// Two-molecule sweeps compute only the groups that hold a cell with letters on both strands: for one diagonal that is a short run
// of consecutive groups around the cut, the same for every pair of a batch of equal lengths.  The host launches that run only (the
// union over the batch; a workgroup that merely finds out that it has nothing to do still costs two dependent scalar loads): grid
// index 0..2 = the F5 / XP / XS groups of the sequence, 3.. = cell group slot0 + index - 3.  False: nothing to do.
// acc += sum_{k = lo + tid, step nt, k <= hi} A(k) * B(k) for the serial sums of the F5 / XP / XS groups: four terms per batch of loads, every
// load unconditional at a clamped index and both factors selected to 0 afterwards (the same FMAs in the same order as the plain loop,
// so the same bits).  In the one-wavefront launches of the look-ahead pairs these groups have up to n/64 terms per thread; one
// exposed memory round trip per term made them the longest workgroups of those launches.
template <class FA, class FB>
__device__ __forceinline__ double dot4_vl(int lo, int hi, int tid, int nt, double acc, FA&& A, FB&& B)
{
    for (int k = lo + tid; k <= hi; k += 4 * nt) {
        double x[4], y[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = k + u * nt;
            ok[u] = kk <= hi;
            const int kc = ok[u] ? kk : hi;
            x[u] = A(kc); y[u] = B(kc);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = fma(ok[u] ? x[u] : 0.0, ok[u] ? y[u] : 0.0, acc);
    }
    return acc;
}
template <bool CUT>
__device__ __forceinline__ bool window_slot_vl(int pin, int ngroup, int* slot)
{
    if (!CUT || !(pin & 128)) return *slot <= ngroup + (CUT ? 2 : 0);
    const int b = *slot;
    if (b < 3) { *slot = ngroup + b; return true; }
    *slot = (pin >> 8) + b - 3;
    return *slot < ngroup;
}

template <int T>
__device__ __forceinline__ double vfilt_fwd(const double* __restrict__ wt, const double* seg)
{
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int l1 = 0; l1 <= T; l1 += 2) {
        s0 = fma(wt[l1], seg[l1], s0);
        if (l1 + 1 <= T) s1 = fma(wt[l1 + 1], seg[l1 + 1], s1);
    }
    return s0 + s1;
}
template <int T>
__device__ __forceinline__ double vfilt_rev(const double* __restrict__ wt, const double* seg)
{
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int l1 = 0; l1 <= T; l1 += 2) {
        s0 = fma(wt[l1], seg[T - l1], s0);
        if (l1 + 1 <= T) s1 = fma(wt[l1 + 1], seg[T - l1 - 1], s1);
    }
    return s0 + s1;
}
// the same filters with a per-lane limit on each gap (two-molecule form: a loop side may not cross the missing gap).
// Only the few 64-cell groups next to the gap take this path: a rolled loop, taps outside [lo, hi] are skipped per lane.
__device__ __forceinline__ double vfilt_fwd_m(int t, const double* __restrict__ wt, const double* seg, int l1max, int l2max)
{
    const int lo = t - l2max > 0 ? t - l2max : 0, hi = t < l1max ? t : l1max;
    double s0 = 0.0;
#pragma unroll 1
    for (int l1 = 0; l1 <= t; l1++) s0 += (l1 >= lo && l1 <= hi) ? wt[l1] * seg[l1] : 0.0;
    return s0;
}
__device__ __forceinline__ double vfilt_rev_m(int t, const double* __restrict__ wt, const double* seg, int l1max, int l2max)
{
    const int lo = t - l2max > 0 ? t - l2max : 0, hi = t < l1max ? t : l1max;
    double s0 = 0.0;
#pragma unroll 1
    for (int l1 = 0; l1 <= t; l1++) s0 += (l1 >= lo && l1 <= hi) ? wt[l1] * seg[t - l1] : 0.0;
    return s0;
}
// look-ahead pair (see vlin_inside_diag MODE 1): staged row r carries filter t = r-1 of diagonal d (taps l < r, weights wA)
// and filter t = r of diagonal d+1 (taps l <= r, weights wB); generic filters exist for t >= 4 only
template <int R>
__device__ __forceinline__ void vfilt_pair(const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
    for (int l = 0; l <= R; l++) {
        const double x = seg[l];
        if (R - 1 >= 4 && l < R) { if (l & 1) a1 = fma(wA[l], x, a1); else a0 = fma(wA[l], x, a0); }
        if (R <= kMaxSingle) { if (l & 1) b1 = fma(wB[l], x, b1); else b0 = fma(wB[l], x, b0); }
    }
    sa = a0 + a1;
    sb = b0 + b1;
}
__device__ __forceinline__ void vfilt_pair_any(int r, const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{
    switch (r) {
#define X(T) case T: vfilt_pair<T>(wA, wB, seg, sa, sb); return;
        X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
        X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#undef X
    }
    sa = 0.0; sb = 0.0;
}
// the same with per-lane gap limits of both cells (rolled; only the groups next to the missing gap)
__device__ __forceinline__ void vfilt_pair_m(int r, const double* __restrict__ wA, const double* __restrict__ wB, const double* seg,
                                             int l1maxA, int l2maxA, int l1maxB, int l2maxB, double& sa, double& sb)
{
    const int tA = r - 1;
    const int loA = tA - l2maxA > 0 ? tA - l2maxA : 0, hiA = tA < l1maxA ? tA : l1maxA;
    const int loB = r - l2maxB > 0 ? r - l2maxB : 0, hiB = r < l1maxB ? r : l1maxB;
    const bool onA = tA >= 4, onB = r <= kMaxSingle;
    double a = 0.0, b = 0.0;
#pragma unroll 1
    for (int l = 0; l <= r; l++) {
        const double x = seg[l];
        a += (onA && l >= loA && l <= hiA) ? wA[l] * x : 0.0;
        b += (onB && l >= loB && l <= hiB) ? wB[l] * x : 0.0;
    }
    sa = a; sb = b;
}
// reversed forms for the outside sweep: x_k = seg[r-k]
template <int R>
__device__ __forceinline__ void vfilt_pair_rev(const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
    for (int l = 0; l <= R; l++) {
        const double x = seg[R - l];
        if (R - 1 >= 4 && l < R) { if (l & 1) a1 = fma(wA[l], x, a1); else a0 = fma(wA[l], x, a0); }
        if (R <= kMaxSingle) { if (l & 1) b1 = fma(wB[l], x, b1); else b0 = fma(wB[l], x, b0); }
    }
    sa = a0 + a1;
    sb = b0 + b1;
}
__device__ __forceinline__ void vfilt_pair_rev_any(int r, const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{
    switch (r) {
#define X(T) case T: vfilt_pair_rev<T>(wA, wB, seg, sa, sb); return;
        X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
        X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#undef X
    }
    sa = 0.0; sb = 0.0;
}
__device__ __forceinline__ void vfilt_pair_rev_m(int r, const double* __restrict__ wA, const double* __restrict__ wB, const double* seg,
                                                 int l1maxA, int l2maxA, int l1maxB, int l2maxB, double& sa, double& sb)
{
    const int tA = r - 1;
    const int loA = tA - l2maxA > 0 ? tA - l2maxA : 0, hiA = tA < l1maxA ? tA : l1maxA;
    const int loB = r - l2maxB > 0 ? r - l2maxB : 0, hiB = r < l1maxB ? r : l1maxB;
    const bool onA = tA >= 4, onB = r <= kMaxSingle;
    double a = 0.0, b = 0.0;
#pragma unroll 1
    for (int l = 0; l <= r; l++) {
        const double x = seg[r - l];
        a += (onA && l >= loA && l <= hiA) ? wA[l] * x : 0.0;
        b += (onB && l >= loB && l <= hiB) ? wB[l] * x : 0.0;
    }
    sa = a; sb = b;
}
// generic loops need l1, l2 >= 1 and t >= 4 (1x1, 1x2, 2x1 are tabulated; 2x2 has weight 0 in shape_w)
#define RH_VT_CASES(X) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30)
__device__ __forceinline__ double vfilt_fwd_any(int t, const double* __restrict__ wt, const double* seg)
{
    switch (t) {
#define X(T) case T: return vfilt_fwd<T>(wt, seg);
        RH_VT_CASES(X)
#undef X
    }
    return 0.0;
}
__device__ __forceinline__ double vfilt_rev_any(int t, const double* __restrict__ wt, const double* seg)
{
    switch (t) {
#define X(T) case T: return vfilt_rev<T>(wt, seg);
        RH_VT_CASES(X)
#undef X
    }
    return 0.0;
}
// letters g and g+1 are neighbours on one strand (cut = 0: one molecule, never equals a gap index >= 1)
__device__ __forceinline__ bool gap_ok_vl(int cut, int g) { return g != cut; }

// the seven tabulated loop shapes: outer pair type t1, inner pair type t2 (0: the letters do not pair), letters si1/sj1
// next to the outer pair inside the loop, sp1/sq1 next to the inner pair
// the address small_w reads (l1, l2 wave-uniform: the branches hold no load)
__device__ __forceinline__ const double* small_w_ptr(const VLinModel* L, int l1, int l2, int t1, int t2, int si1, int sj1, int sp1, int sq1)
{
    const int r2 = vienna_rtype(t2);
    const int tt = t1 * 8 + r2;
    if (l1 == 0 && l2 == 0) return &L->E_stack[tt];
    if (l1 + l2 == 1) return &L->E_bulge1[tt];
    if (l1 == 1 && l2 == 1) return &L->E_int11[tt * 25 + si1 * 5 + sj1];
    if (l1 == 1 && l2 == 2) return &L->E_int21[tt * 125 + (si1 * 5 + sq1) * 5 + sj1];
    if (l1 == 2 && l2 == 1) return &L->E_int21[(r2 * 8 + t1) * 125 + (sq1 * 5 + si1) * 5 + sp1];
    return &L->E_int22[tt * 625 + ((si1 * 5 + sp1) * 5 + sq1) * 5 + sj1];
}
__device__ __forceinline__ double small_w(const VLinModel* L, int l1, int l2, int t1, int t2, int si1, int sj1, int sp1, int sq1)
{
    const int r2 = vienna_rtype(t2);
    const int tt = t1 * 8 + r2;
    if (l1 == 0 && l2 == 0) return L->E_stack[tt];
    if (l1 + l2 == 1) return L->E_bulge1[tt];
    if (l1 == 1 && l2 == 1) return L->E_int11[tt * 25 + si1 * 5 + sj1];
    if (l1 == 1 && l2 == 2) return L->E_int21[tt * 125 + (si1 * 5 + sq1) * 5 + sj1];
    if (l1 == 2 && l2 == 1) return L->E_int21[(r2 * 8 + t1) * 125 + (sq1 * 5 + si1) * 5 + sp1];
    return L->E_int22[tt * 625 + ((si1 * 5 + sp1) * 5 + sq1) * 5 + sj1];
}

}  // namespace

#ifdef RH_VSTAMPS
// tuning build only (tools/build_variant.py vstamps -DRH_VSTAMPS=1): per-phase cycle totals of vlin_inside_diag<.., CUT = (RH_VSTAMPS == 2), MODE 1>,
// wavefront 0 of every cell workgroup (tools/vstamps.py)
__device__ unsigned long long g_vstamps[16];
extern "C" int rh_debug_vstamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vstamps), sizeof(g_vstamps)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_vstamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define RH_VSTAMP(k) do { if (MODE == 1 && CUT == (RH_VSTAMPS == 2) && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_vstamps[k], t_ - t_prev_); t_prev_ = t_; } } while (0)
#define RH_VSTAMP_BEGIN() unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(); if (MODE == 1 && CUT == (RH_VSTAMPS == 2) && threadIdx.x == 0) atomicAdd(&g_vstamps[15], 1ull)
#define RH_VPIN(x) asm volatile("" :: "v"(x))   // the phase's result exists before its stamp is taken
#else
#define RH_VPIN(x) do { } while (0)
#define RH_VSTAMP(k) do { } while (0)
#define RH_VSTAMP_BEGIN() do { } while (0)
#endif
#ifndef RH_VLA_WPE
#define RH_VLA_WPE 4   // wavefronts/SIMD the look-ahead (MODE 1) kernel is compiled for
#endif
#define GAPOK(g) (!CUT || gap_ok_vl(cut, (g)))
// table slots: 3, 4, 7, 10, 11, 12 are the ones mccaskill_far.hip addresses (LinTableFar)
enum VLinTable { VL_FC = 0, VL_FCX, VL_FCA, VL_FM1, VL_FM, VL_FCO, VL_FCOX, VL_FM2O, VL_FMSO, VL_FM1O,
                 VL_FM2F, VL_FMOF, VL_FM1OF, VL_FMS, VL_FCB, VL_FCOB, VL_COUNT };
static_assert((int)VL_COUNT <= kViennaMcTables, "linear tables reuse the log-space table buffer");

__global__ void vlin_init(McBatch B, int* __restrict__ bad)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq >= B.ns) return;
    B.f5i[(size_t)sq * B.ld] = 1.0;
    B.f5o[(size_t)sq * B.ld + B.n[sq]] = 1.0;
    bad[sq] = 0;
    const int cut = B.cut ? B.cut[sq] : 0;
    if (cut > 0) {   // two-molecule form: exterior halves of the loop around the missing gap (see mccaskill_vienna.hip)
        const size_t o = (size_t)sq * B.ld;
        B.xp[o + cut] = 1.0;
        B.xs[o + cut + 1] = 1.0;
        B.xpo[o + B.n[sq]] = 0.0;
        B.xso[o + 1] = 0.0;
    }
}

// two-molecule batch B (pair p = s1+s2, cut after s1) next to the single-molecule batch S of the same upload (sequences 2p, 2p+1),
// both in scaled linear space with the same lam: a cell whose letters all lie on one strand has the value it has in that
// molecule folded alone -- the same recurrences over the same letters, and at a strand end the neighbour letter is "none" in
// both (FCX differs at the strand ends, where it looks across the gap, but only loops whose side crosses the gap read it there,
// and those are excluded).  One launch copies four inside tables of both triangles: FCA, FM1, FMS and FM, the multiloop and exterior
// pieces a cell with letters on both strands is built from.  FC, FCX and FCB of a one-strand cell are never used by such a cell: the
// inner pair of its interior loops, bulges and stacks has letters on both strands too (each side of the loop stays on its strand:
// the per-lane limits l1max / l2max, and the strand mask of the staged rows), and every operand that might be one of them is
// discarded by a select, never by a multiplication with 0.
__global__ __launch_bounds__(256) void vlin_co_seed(McBatch B, McBatch S)
{
    const int p = blockIdx.y, d = blockIdx.x;
    const int n = B.n[p], n1 = B.cut[p];
    constexpr int NT = 4;
    const int tabs[NT] = {VL_FCA, VL_FM1, VL_FMS, VL_FM};
    double* __restrict__ dst = B.tab + (size_t)p * B.seq_stride + (size_t)d * B.ld;
    for (int strand = 0; strand < 2; strand++) {
        const int cells = (strand ? n - n1 : n1) - 1 - d, off = strand ? n1 : 0;
        if (cells < 1) continue;
        const double* __restrict__ src = S.tab + (size_t)(2 * p + strand) * S.seq_stride + (size_t)d * S.ld;
        for (int t = 0; t < NT; t++)
            for (int i = 1 + threadIdx.x; i <= cells; i += 256) dst[tabs[t] * B.tab_stride + off + i] = src[tabs[t] * S.tab_stride + i];
    }
}

// ---------------------------------------------------------------------------------
// inside, diagonal d.  hp_d = lam^d * hairpin length weight of a loop of d unpaired letters.
// MODE 0: one full launch per diagonal.  MODE 1 / 2 = look-ahead pair (as lin_inside_diag MODE 1 / 2 of mccaskill_lin.hip):
// the MODE 1 launch of an even diagonal d also accumulates, from the operands it holds, the sums of diagonal d+1 that do not
// touch row d (FM2 without m = 1 and m = d; the filter of length t+1 over the staged row of the filter of length t; the bulge
// taps) and leaves them in B.rowp; the MODE 2 launch of d+1 (one wavefront per group) adds the two fresh FM2 terms and runs
// the epilogue.  F5 / XP / XS groups run in both launches unchanged.
template <int W, int BS, bool CUT, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 64 : 64 * W) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 6 : RH_VLA_WPE, 8))) void vlin_inside_diag(McBatch B, const VLinModel* __restrict__ L, int d, double hp_d, int pin)
{
    constexpr int WR = MODE == 2 ? 1 : W;
    __shared__ double part[MODE == 1 ? 6 : 3][WR][64];
    __shared__ double gbuf[MODE == 2 ? 1 : W][MODE == 2 ? 1 : 2 * ((kMaxSingle / 2 + W) / W)][MODE == 2 ? 1 : 96];
    int sq, slot;
    block_map_vl(pin, &sq, &slot);
    double* __restrict__ rowp = B.rowp + (size_t)sq * 3 * B.ld;   // look-ahead sums of the next diagonal
    if (sq >= B.ns) return;
    // length and cut in ONE scalar round trip (behind the early return on the length, the cut was a second dependent one in front of
    // every workgroup); CUT = false: one molecule per sequence, every gap test folds away
    int n = B.n[sq], cut = CUT ? B.cut[sq] : 0;
    if (CUT) asm volatile("" : "+s"(n), "+s"(cut));
    if (d > n - 1) return;
    const int ncell = n - 1 - d > 0 ? n - 1 - d : 0;
    const int ngroup = (ncell + 63) >> 6;
    if (!window_slot_vl<CUT>(pin, ngroup, &slot)) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    double* __restrict__ f5i = B.f5i + sq * ld;

    if (slot > ngroup) {
        // XP~[b], b = cut+d+1 (exterior partition function of s2's prefix cut+1..b) or XS~[a], a = cut-d (s1's suffix a..cut)
        const bool is_xp = slot == ngroup + 1;
        const double* __restrict__ fca = tab + VL_FCA * ts;
        double* __restrict__ xv = (is_xp ? B.xp : B.xs) + (size_t)sq * ld;
        const int b = cut + d + 1, a = cut - d;
        if (is_xp ? b > n : a < 1) return;
        double acc = 0.0;
        if (is_xp) acc = dot4_vl(cut, b - 2, threadIdx.x, 64 * WR, acc, [&](int k) { return xv[k]; }, [&](int k) { return fca[(b - k - 2) * ld + (k + 1)]; });
        else acc = dot4_vl(a + 4, cut, threadIdx.x, 64 * WR, acc, [&](int l) { return fca[(l - 1 - a) * ld + a]; }, [&](int l) { return xv[l + 1]; });
        acc = wsum_vl(acc);
        if (lane == 0) part[0][w][0] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < WR; k++) t += part[0][k][0];
            if (is_xp) xv[b] = xv[b - 1] * L->lam + t * L->lam2;
            else xv[a] = xv[a + 1] * L->lam + t * L->lam2;
        }
        return;
    }
    if (slot == ngroup) {
        // F5i~[jj] = F5i~[jj-1]*lam + sum_{k<=jj-2} F5i~[k]*FCA~[k+1,jj-1]*lam^2
        const int jj = d + 1;
        const double* __restrict__ fca = tab + VL_FCA * ts;
        double acc = 0.0;
        acc = dot4_vl(0, jj - 2, threadIdx.x, 64 * WR, acc, [&](int k) { return f5i[k]; }, [&](int k) { return fca[(jj - 2 - k) * ld + (k + 1)]; });
        acc = wsum_vl(acc);
        if (lane == 0) part[0][w][0] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < WR; k++) t += part[0][k][0];
            f5i[jj] = f5i[jj - 1] * L->lam + t * L->lam2;
        }
        return;
    }

    if (CUT && B.seeded && cut > 0) {   // cells whose letters all lie on one strand hold the values of that molecule folded alone (vlin_co_seed)
        const int i0 = 1 + slot * 64, dd = MODE == 1 ? d + 1 : d;   // MODE 1 also feeds diagonal d+1; its MODE 2 launch tests the same dd
        if (i0 + 63 + dd + 1 <= cut || i0 > cut) return;
    }
    RH_VSTAMP_BEGIN();
    const int i = 1 + slot * 64 + lane, j = i + d;
    const bool valid = i <= ncell;
    int s_im1 = 0, s_i = 0, s_ip1 = 0, s_j = 0, s_jp1 = 0, s_jp2 = 0;
    if (valid) { s_im1 = s[i - 1]; s_i = s[i]; s_ip1 = s[i + 1]; s_j = s[j]; s_jp1 = s[j + 1]; s_jp2 = s[j + 2]; }
    const int type = (B.allow && valid && !B.allow[((size_t)sq * ld + i) * ld + (j + 1)]) ? 0 : vienna_ptype(s_i, s_jp1);   // 0: excluded by a structure constraint
    const bool pairable = valid && type != 0;
    // two-molecule form: the missing gap inside the pair limits both sides of an enclosed loop to their own strand
    const bool nick_in = CUT && valid && i <= cut && cut <= j;
    const int l1max = nick_in ? cut - i - 1 : 99, l2max = nick_in ? j - cut - 1 : 99;
    // MODE 1: cell (i, j+1) of diagonal d+1
    const int d1 = d + 1;
    const bool valid1 = MODE == 1 && i <= n - 1 - d1;
    const bool nick_in1 = CUT && valid1 && i <= cut && cut <= j + 1;
    const int l1max1 = nick_in1 ? cut - i - 1 : 99, l2max1 = nick_in1 ? j - cut : 99;
    // Seeded two-molecule sweep: every cell that is computed here has letters on both strands, and the interior loops of such a cell
    // close over inner pairs with letters on both strands only (each side of the loop stays on its strand) -- a property of the INNER
    // cell.  The staged FCX rows are therefore masked once, on their way into LDS (MODE 1), and the filters run unmasked and unrolled;
    // the per-lane limits of vfilt_pair_m (a rolled loop, one LDS and one scalar round trip per tap) remain for the unseeded sweep.
    const bool data_mask = CUT && MODE == 1 && B.seeded && cut > 0;
    const bool masked = CUT && !data_mask && __any(l1max < kMaxSingle || l2max < kMaxSingle || l1max1 < kMaxSingle || l2max1 < kMaxSingle);   // the same for all W wavefronts of the group

    // epilogue operands (wave 0 only), issued ahead of the term loops
    const size_t at = d * ld + i;
    const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1 + s_j;       // (i,j+1) seen from inside
    // seen from outside; a neighbour on the other molecule gives no dangle (letter code 0)
    const int idd = 25 * (5 * s_jp1 + (GAPOK(j + 1) ? s_jp2 : 0)) + 5 * s_i + (GAPOK(i - 1) ? s_im1 : 0);
    const int idx_raw = 25 * (5 * s_jp1 + s_jp2) + 5 * s_i + s_im1;
    RH_VPIN(type); RH_VSTAMP(0);   // letters and pair type
    // ---- MODE 1: the epilogue's operands are gathered HERE, by all W wavefronts (three or four loads each), and travel while the term
    // loops run; they meet in LDS at the barrier that collects the partial sums.  Loaded by wavefront 0 alone behind that barrier (two
    // dependent round trips: letters, then the tables) they were a third of a workgroup's life, spent with seven wavefronts gone and the
    // group's LDS still held.  Wavefront 0: FCA, FM1, FMS of the neighbour cells; wavefront k+1: value and weight of tabulated shape k plus
    // one of TXO TMC TMH TXI TSA tau TNC; wavefronts 1, 2 (two-molecule form): XS, XP.  One address per slot, selected by role -- no branch
    // around a load.  Every value reaches the epilogue unchanged, so the results are bit-identical to the other launch organisations.
    constexpr bool DIST = MODE == 1 && W == 8;
    double opA = 0.0, opB = 0.0, opC = 0.0, opD = 0.0;
    int b_ip2 = 0, b_ip3 = 0, b_jm1 = 0, b_jm2 = 0;
    if constexpr (DIST) {
        if (valid) {
            b_ip2 = s[i + 2 <= n + 1 ? i + 2 : n + 1]; b_ip3 = s[i + 3 <= n + 1 ? i + 3 : n + 1];     // letters i+2, i+3, j-1, j-2
            b_jm1 = s[j - 1 >= 0 ? j - 1 : 0]; b_jm2 = s[j - 2 >= 0 ? j - 2 : 0];
        }
        const int iv = valid ? i : 1;
        const int k = w - 1;   // tabulated shape of this wavefront (w >= 1)
        const int l1 = (k == 2 || k == 3 || k == 4) ? 1 : (k >= 5 ? 2 : 0);
        const int l2 = (k == 1 || k == 3 || k == 5) ? 1 : ((k == 4 || k == 6) ? 2 : 0);
        const int t = l1 + l2;
        const bool sokk = pairable & (d - 2 - t >= 0) & (l1 <= l1max) & (l2 <= l2max);
        const int sp = l1 == 0 ? s_ip1 : (l1 == 1 ? b_ip2 : b_ip3), spm = l1 == 0 ? s_i : (l1 == 1 ? s_ip1 : b_ip2);
        const int sq_ = l2 == 0 ? s_j : (l2 == 1 ? b_jm1 : b_jm2), sqp = l2 == 0 ? s_jp1 : (l2 == 1 ? s_j : b_jm1);
        const int t2 = vienna_ptype(sp, sq_);
        const unsigned dm1 = (unsigned)((d >= 1 ? d - 1 : 0) * ld + iv), dm2 = (unsigned)((d >= 2 ? d - 2 : 0) * ld + iv);
        const double* __restrict__ Lt = (const double*)L;
        constexpr size_t oTXO = offsetof(VLinModel, TXO) / 8, oTMC = offsetof(VLinModel, TMC) / 8, oTMH = offsetof(VLinModel, TMH) / 8,
                         oTXI = offsetof(VLinModel, TXI) / 8, oTSA = offsetof(VLinModel, TSA) / 8, oTNC = offsetof(VLinModel, TNC) / 8,
                         oTAU = offsetof(VLinModel, E_tau) / 8;
        // slot A: a table cell
        const size_t offA = w == 0 ? VL_FCA * ts + dm2 + 1 : VL_FC * ts + (unsigned)((sokk ? d - 2 - t : 0) * ld + iv + 1 + l1);
        // slot B: FM1 of the neighbour cell / the shape's weight
        const double* __restrict__ pB = w == 0 ? tab + (VL_FM1 * ts + dm1 + 1) : small_w_ptr(L, l1, l2, type, t2, s_ip1, s_j, spm, sqp);
        // slot C: FMS of the neighbour cell / one letter-indexed table
        const int tnc = 25 * (5 * s_i + (GAPOK(i) ? s_ip1 : 0)) + 5 * s_jp1 + (GAPOK(j) ? s_j : 0);
        const size_t offC = w == 1 ? oTXO + idx : w == 2 ? oTMC + idx : w == 3 ? oTMH + idx : w == 4 ? oTXI + idx_raw : w == 5 ? oTSA + idd
                          : w == 6 ? oTAU + type : oTNC + (CUT ? tnc : 0);
        const double* __restrict__ pC = w == 0 ? tab + (VL_FMS * ts + dm1) : Lt + offC;
        opA = tab[offA]; opB = *pB; opC = *pC;
        if constexpr (CUT) {
            const bool nk = nick_in & pairable & (d >= kMinHairpin);
            const size_t o = (size_t)sq * ld;
            const double* __restrict__ pD = w == 2 ? B.xp + (o + (nk ? j : 1)) : B.xs + (o + ((nk & (w == 1)) ? i + 1 : 1));
            opD = *pD;
        }
    }
    // ---- FM2[i,d] = sum_{m=1}^{d-1} FM1[m][i] * FM[d-m][i+m]: near terms here, far blocks from FM2F
    double acc2 = 0.0, acc2n = 0.0;
    {
        const double* __restrict__ fm1 = tab + VL_FM1 * ts + i;
        const double* __restrict__ fm = tab + VL_FM * ts + i;
        int kA = 1 << 30, kB = 0;
        if (BS > 0) {
            const int I = i / BS, J = j / BS;
            if (J - I >= 4) { kA = (I + 2) * BS; kB = (J - 1) * BS; }
        }
        int kA1 = 1 << 30, kB1 = 0;   // near set of (i, j+1)
        if (MODE == 1 && BS > 0) {
            const int I = i / BS, J1 = (j + 1) / BS;
            if (J1 - I >= 4) { kA1 = (I + 2) * BS; kB1 = (J1 - 1) * BS; }
        }
        if constexpr (MODE == 2) {
            // the look-ahead sum of the previous launch + the two terms that touch row d-1: six loads, requested together and
            // selected afterwards (a lane past the diagonal reads its neighbours' cells; fma results it does not want are dropped whole)
            {
                const int dm = d >= 1 ? d - 1 : 0;
                const double r0 = rowp[i], a1 = fm1[ld], b1 = fm[dm * ld + 1], a2 = fm1[dm * ld], b2 = fm[ld + dm];
                const double f2 = tab[VL_FM2F * ts + d * ld + i];
                const int k1 = i + 1, k2 = i + d - 1;
                const bool c1 = valid & (d >= 2) & ((k1 < kA) | (k1 >= kB)), c2 = valid & (d >= 3) & ((k2 < kA) | (k2 >= kB));
                acc2 = valid ? r0 : 0.0;
                const double t1 = fma(a1, b1, acc2);
                acc2 = c1 ? t1 : acc2;                                                                              // m = 1
                const double t2 = fma(a2, b2, acc2);
                acc2 = c2 ? t2 : acc2;                                                                              // m = d-1
                acc2 = (valid & (BS > 0) & (kB > 0)) ? acc2 + f2 : acc2;
            }
        } else {
        constexpr int UF = MODE == 1 ? 6 : 8;
        const bool split = BS > 0 && d - 1 > 4 * BS;
        const int lo0 = 1, hi0 = split ? 2 * BS : d - 1;
        const int lo1 = split ? d - 2 * BS : 1, hi1 = split ? d - 1 : 0;
#pragma unroll
        for (int part_i = 0; part_i < 2; part_i++) {
            const int lo = part_i ? lo1 : lo0, hi = part_i ? hi1 : hi0;
            for (int m = lo + w; m <= hi; m += UF * W) {
                double a[UF], b[UF], bn[UF];
#pragma unroll
                for (int u = 0; u < UF; u++) {
                    const int mm = m + u * W, k = i + mm;
                    const bool ok = valid && mm <= hi && (k < kA || k >= kB);
                    const bool ok1 = MODE == 1 && valid1 && mm <= hi && mm >= 2 && (k < kA1 || k >= kB1);
                    a[u] = (ok || ok1) ? fm1[mm * ld] : 0.0;
                    b[u] = ok ? fm[(d - mm) * ld + mm] : 0.0;
                    bn[u] = ok1 ? fm[(d1 - mm) * ld + mm] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UF; u++) { acc2 = fma(b[u] != 0.0 ? a[u] : 0.0, b[u], acc2); if (MODE == 1) acc2n = fma(a[u], bn[u], acc2n); }
            }
        }
        if (BS > 0 && valid && kB > 0) acc2 += (w == 0) ? tab[VL_FM2F * ts + d * ld + i] : 0.0;
        }
    }

    RH_VPIN(acc2); RH_VPIN(acc2n); RH_VSTAMP(1);   // FM2 near terms
    // ---- generic interior loops (LDS-staged filters over FCX) and long bulges (two taps per length over FCB)
    double accc = 0.0, accb = 0.0, acccn = 0.0, accbn = 0.0;
    if constexpr (MODE == 2) {
        { const double rc = rowp[ld + i], rb = rowp[2 * ld + i]; accc = valid ? rc : 0.0; accb = valid ? rb : 0.0; }
        if (!pairable) { accc = 0.0; accb = 0.0; }
    } else if constexpr (MODE == 1) {
        // staged row r = 2..31 is table row d-1-r: filter / bulge length t = r-1 of diagonal d and t = r of diagonal d+1
        if (d >= 3) {
            constexpr int LAST = kMaxSingle + 1, HALF = (kMaxSingle + 1) / 2;
            const int rmax = d - 1 < LAST ? d - 1 : LAST;
            const int i0 = 1 + slot * 64;
            const double* __restrict__ fcx = tab + VL_FCX * ts;
            const double* __restrict__ fcb = tab + VL_FCB * ts;
            constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);
            int rseg[NSEG];
            double bA[NSEG], bB[NSEG];
            // every load of the staging first, unconditional (a segment that is off reads row d-3, which exists, and is dropped): a load
            // inside `if (on)` is waited for inside it, one memory round trip per segment and bulge tap
            double g0[NSEG], g1[NSEG], x0[NSEG], xa[NSEG], xb[NSEG];
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int r = (q & 1) ? LAST - g : g;
                const bool on = g <= HALF && r <= rmax && r >= 2;   // wave-uniform
                const int rr = on ? r : 2;
                const double* __restrict__ row = fcx + (d - 1 - rr) * ld + i0 + 1;
                const double* __restrict__ brow = fcb + (d - 1 - rr) * ld + i + 1;
                g0[q] = row[lane]; g1[q] = row[64 + (lane & 31)];
                x0[q] = brow[0]; xa[q] = brow[rr - 1]; xb[q] = brow[rr];
            }
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                asm volatile("" : "+v"(g0[q])); asm volatile("" : "+v"(g1[q])); asm volatile("" : "+v"(x0[q])); asm volatile("" : "+v"(xa[q])); asm volatile("" : "+v"(xb[q]));
            }
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int r = (q & 1) ? LAST - g : g;
                const bool on = g <= HALF && r <= rmax && r >= 2;   // wave-uniform
                rseg[q] = on ? r : -1;
                const int col0 = i0 + 1;
                if (on && r >= 4) {
                    const int sp = d - 1 - r;   // span of the staged row: inner cell (p, sp) has letters on both strands iff p <= cut <= p + sp
                    const int p0 = col0 + lane, p1 = col0 + 64 + lane;
                    gbuf[w][q][lane] = (p0 < ld && (!data_mask || (p0 <= cut && cut <= p0 + sp))) ? g0[q] : 0.0;
                    if (lane < 32) gbuf[w][q][64 + lane] = (p1 < ld && (!data_mask || (p1 <= cut && cut <= p1 + sp))) ? g1[q] : 0.0;
                }
                // bulges of length t on the 3' side (l1 = 0: column i+1) and on the 5' side (l1 = t: column i+1+t)
                const int tA = r - 1;
                const bool a0 = on & valid & (tA >= 2) & (l1max >= 0) & (tA <= l2max), a1 = on & valid & (tA >= 2) & (tA <= l1max) & (l2max >= 0);
                const bool b0 = on & valid1 & (r <= kMaxSingle) & (l1max1 >= 0) & (r <= l2max1), b1 = on & valid1 & (r <= kMaxSingle) & (r <= l1max1) & (l2max1 >= 0);
                bA[q] = (a0 ? x0[q] : 0.0) + (a1 ? xa[q] : 0.0);
                bB[q] = (b0 ? x0[q] : 0.0) + (b1 ? xb[q] : 0.0);
            }
            RH_VSTAMP(2);   // staging loads arrived, rows in LDS
#pragma unroll
            for (int q = 0; q < NSEG; q++)
                if (rseg[q] >= 0) {
                    const int r = rseg[q];
                    if (r >= 4) {
                        const double* __restrict__ wA = L->shape_w + (r - 1) * r / 2;
                        const double* __restrict__ wB = L->shape_w + (r <= kMaxSingle ? r * (r + 1) / 2 : 0);
                        double sa, sb;
                        if (masked) vfilt_pair_m(r, wA, wB, &gbuf[w][q][lane], l1max, l2max, l1max1, l2max1, sa, sb);
                        else vfilt_pair_any(r, wA, wB, &gbuf[w][q][lane], sa, sb);
                        accc += sa;
                        acccn += sb;
                    }
                    if (r - 1 >= 2) accb = fma(L->WB[r - 1], bA[q], accb);
                    if (r <= kMaxSingle) accbn = fma(L->WB[r], bB[q], accbn);
                }
            if (!pairable) { accc = 0.0; accb = 0.0; }
        } else if (d >= 1) {
            // d = 1, 2: diagonal d has no generic loop yet (t <= 0); diagonal d+1 has lengths t <= d-1 <= 1: none either
        }
    } else
    if (d >= 2) {
        const int tmax = d - 2 < kMaxSingle ? d - 2 : kMaxSingle;
        const int i0 = 1 + slot * 64;
        const double* __restrict__ fcx = tab + VL_FCX * ts;
        const double* __restrict__ fcb = tab + VL_FCB * ts;
        constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);
        int tseg[NSEG];
        double b0[NSEG], b1[NSEG];
#pragma unroll
        for (int q = 0; q < NSEG; q++) {
            const int g = w + (q >> 1) * W;
            const int t = (q & 1) ? kMaxSingle - g : g;
            const bool on = g <= kMaxSingle / 2 && !((q & 1) && t == g) && t <= tmax && t >= 2;   // wave-uniform
            tseg[q] = on ? t : -1;
            b0[q] = b1[q] = 0.0;
            if (on) {
                const int col0 = i0 + 1;
                if (t >= 4) {
                    const double* __restrict__ row = fcx + (d - 2 - t) * ld + col0;
                    gbuf[w][q][lane] = col0 + lane < ld ? row[lane] : 0.0;
                    if (lane < 32) gbuf[w][q][64 + lane] = col0 + 64 + lane < ld ? row[64 + lane] : 0.0;
                }
                if (valid) {   // bulge of length t on the 3' side (l1 = 0) and on the 5' side (l1 = t)
                    if (l1max >= 0 && t <= l2max) b0[q] = fcb[(d - 2 - t) * ld + i + 1];
                    if (t <= l1max && l2max >= 0) b1[q] = fcb[(d - 2 - t) * ld + i + 1 + t];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NSEG; q++)
            if (tseg[q] >= 0) {
                if (tseg[q] >= 4) {
                    const double* __restrict__ wt = L->shape_w + tseg[q] * (tseg[q] + 1) / 2;
                    accc += masked ? vfilt_fwd_m(tseg[q], wt, &gbuf[w][q][lane], l1max, l2max) : vfilt_fwd_any(tseg[q], wt, &gbuf[w][q][lane]);
                }
                accb = fma(L->WB[tseg[q]], b0[q] + b1[q], accb);
            }
        if (!pairable) { accc = 0.0; accb = 0.0; }
    }

    RH_VPIN(accc); RH_VPIN(acccn); RH_VSTAMP(3);   // filters
    if constexpr (DIST) {   // this wavefront's staging rows are dead: its epilogue operands go there
        gbuf[w][0][lane] = opA; gbuf[w][1][lane] = opB; gbuf[w][2][lane] = opC;
        if constexpr (CUT) gbuf[w][3][lane] = opD;
    }
    if constexpr (MODE != 2) {
        part[0][w][lane] = acc2;
        part[1][w][lane] = accc;
        part[2][w][lane] = accb;
        if constexpr (MODE == 1) { part[3][w][lane] = acc2n; part[4][w][lane] = acccn; part[5][w][lane] = accbn; }
        __syncthreads();
        if constexpr (MODE == 1) {
            if (w == 1 && i < ld) {   // the look-ahead sums of diagonal d+1
                double p2 = 0.0, pc = 0.0, pb = 0.0;
#pragma unroll
                for (int k = 0; k < W; k++) { p2 += part[3][k][lane]; pc += part[4][k][lane]; pb += part[5][k][lane]; }
                rowp[i] = p2; rowp[ld + i] = pc; rowp[2 * ld + i] = pb;
            }
        }
    }
    RH_VSTAMP(4);   // partial sums exchanged (barrier)
    if (w != 0 || !valid) return;
    double e_txo = 0, e_tmc = 0, e_tmh = 0, e_txi = 0, e_tsa = 0, e_tau = 1, e_tet = 1, nick = 0.0;
    double o_fca = 0, o_fm1 = 0, o_fms = 0, sm7 = 0.0;
    {   // epilogue operands: loaded here, after the term loops, so that they do not occupy registers during them -- and ALL AT ONCE:
        // every index below follows from letters that arrived long ago (pair types are arithmetic, vienna_ptype), every load is
        // unconditional from a clamped address and selected afterwards.  Behind per-lane `if`s these were ~20 dependent round trips.
        if constexpr (!DIST) {
            b_ip2 = s[i + 2 <= n + 1 ? i + 2 : n + 1]; b_ip3 = s[i + 3 <= n + 1 ? i + 3 : n + 1];     // letters i+2, i+3, j-1, j-2
            b_jm1 = s[j - 1 >= 0 ? j - 1 : 0]; b_jm2 = s[j - 2 >= 0 ? j - 2 : 0];
        }
        double l_txo, l_tmc, l_tmh, l_txi, l_tsa, l_tau, l_fca, l_fm1, l_fms;
        double l_nick = 0.0;
        if constexpr (DIST) {   // gathered at the head of the kernel by the wavefront named in the row index
            l_fca = gbuf[0][0][lane]; l_fm1 = gbuf[0][1][lane]; l_fms = gbuf[0][2][lane];
            l_txo = gbuf[1][2][lane]; l_tmc = gbuf[2][2][lane]; l_tmh = gbuf[3][2][lane]; l_txi = gbuf[4][2][lane]; l_tsa = gbuf[5][2][lane];
            l_tau = gbuf[6][2][lane];
            if constexpr (CUT) {
                const bool nk = nick_in & pairable & (d >= kMinHairpin);
                const double xs = gbuf[1][3][lane], xp = gbuf[2][3][lane], tn = gbuf[7][2][lane];
                l_nick = nk ? xs * xp * tn : 0.0;
            }
        } else {
            l_txo = L->TXO[idx]; l_tmc = L->TMC[idx]; l_tmh = L->TMH[idx]; l_txi = L->TXI[idx_raw]; l_tsa = L->TSA[idd]; l_tau = L->E_tau[type];
            const unsigned dm1 = (unsigned)((d >= 1 ? d - 1 : 0) * ld + i), dm2 = (unsigned)((d >= 2 ? d - 2 : 0) * ld + i);
            l_fca = tab[VL_FCA * ts + dm2 + 1]; l_fm1 = tab[VL_FM1 * ts + dm1 + 1]; l_fms = tab[VL_FMS * ts + dm1];
            if constexpr (CUT) {
                const bool nk = nick_in & pairable & (d >= kMinHairpin);
                const size_t o = (size_t)sq * ld;
                const double xs = B.xs[o + (nk ? i + 1 : 1)], xp = B.xp[o + (nk ? j : 1)];
                const double tn = L->TNC[25 * (5 * s_i + (GAPOK(i) ? s_ip1 : 0)) + 5 * s_jp1 + (GAPOK(j) ? s_j : 0)];
                l_nick = nk ? xs * xp * tn : 0.0;
            }
        }
        // tetraloop bonus (d = 4: the letters i .. i+5 are s_i, s_ip1, i+2, i+3, s_j, s_jp1)
        const bool tet_ok = (d == 4) & pairable & (s_i != 0) & (s_ip1 != 0) & (b_ip2 != 0) & (b_ip3 != 0) & (s_j != 0) & (s_jp1 != 0);
        const int tet_code = tet_ok ? ((((( (s_i - 1) * 4 + (s_ip1 - 1)) * 4 + (b_ip2 - 1)) * 4 + (b_ip3 - 1)) * 4 + (s_j - 1)) * 4 + (s_jp1 - 1)) : 0;
        double l_tet = 1.0;
        if (!DIST || d == 4) l_tet = L->E_tetra[tet_code];   // (wave-uniform)
        // the seven tabulated shapes: inner pair letters (p, q) = (i+1+l1, j-l2), raw table cell (p, q-1)
        const double* __restrict__ fc = tab + VL_FC * ts;
        double sv[7], sw[7];
        bool sok[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int l1 = (k == 2 || k == 3 || k == 4) ? 1 : (k >= 5 ? 2 : 0);
            const int l2 = (k == 1 || k == 3 || k == 5) ? 1 : ((k == 4 || k == 6) ? 2 : 0);
            const int t = l1 + l2;
            sok[k] = pairable & (d - 2 - t >= 0) & (l1 <= l1max) & (l2 <= l2max);
            // letters of the inner pair and next to it, out of the eight letters held: p = i+1+l1, q = j-l2
            const int sp = l1 == 0 ? s_ip1 : (l1 == 1 ? b_ip2 : b_ip3), spm = l1 == 0 ? s_i : (l1 == 1 ? s_ip1 : b_ip2);
            const int sq_ = l2 == 0 ? s_j : (l2 == 1 ? b_jm1 : b_jm2), sqp = l2 == 0 ? s_jp1 : (l2 == 1 ? s_j : b_jm1);
            const int t2 = vienna_ptype(sp, sq_);
            if constexpr (DIST) { sv[k] = gbuf[k + 1][0][lane]; sw[k] = gbuf[k + 1][1][lane]; }
            else {
                sv[k] = fc[(unsigned)((sok[k] ? d - 2 - t : 0) * ld + i + 1 + l1)];
                sw[k] = small_w(L, l1, l2, type, t2, s_ip1, s_j, spm, sqp);
            }
        }
        e_txo = l_txo; e_tmc = l_tmc; e_tmh = l_tmh; e_txi = l_txi; e_tsa = l_tsa; e_tau = l_tau;
        nick = l_nick;
        e_tet = tet_ok ? l_tet : 1.0;
        if (d >= 2) {   // a multiloop element may not touch the missing gap
            o_fca = (GAPOK(i) & GAPOK(j)) ? l_fca : 0.0;
            o_fm1 = (GAPOK(i) & GAPOK(i + 1)) ? l_fm1 : 0.0;
            o_fms = (GAPOK(j - 1) & GAPOK(j)) ? l_fms : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 7; k++) sm7 = fma(sok[k] ? sv[k] : 0.0, sw[k], sm7);
    }

    RH_VPIN(sm7); RH_VSTAMP(5);   // epilogue operands
    double fm2 = acc2, g = accc, gb = accb;
    if constexpr (MODE != 2) {
        fm2 = 0.0; g = 0.0; gb = 0.0;
#pragma unroll
        for (int k = 0; k < W; k++) { fm2 += part[0][k][lane]; g += part[1][k][lane]; gb += part[2][k][lane]; }
    }

    double fc = 0.0;
    if (pairable) {
        double hp = 0.0;
        if (d >= 3) hp = nick_in ? nick : hp_d * (d == 3 ? e_tau : e_tmh * e_tet);
        fc = e_txo * g + e_tau * gb + sm7 + hp + fm2 * e_tmc;
    }
    double fm1v = 0.0, fmsv = 0.0, fmv = 0.0;
    if (d >= 2) {
        fm1v = o_fca * L->w_mp2 + o_fm1 * L->w_mu;
        fmsv = fm1v + o_fms * L->w_mu;
        fmv = fm2 + fmsv;
    }
    // seeded two-molecule sweep: the one-strand cells of a mixed group keep the copied values (their far sums are not computed)
    if (CUT && B.seeded && cut > 0 && valid && (j + 1 <= cut || i > cut)) return;
    tab[VL_FC * ts + at] = fc;
    tab[VL_FCX * ts + at] = fc * e_txi;
    tab[VL_FCB * ts + at] = fc * e_tau;
    tab[VL_FCA * ts + at] = fc * e_tsa;
    tab[VL_FM1 * ts + at] = fm1v;
    tab[VL_FMS * ts + at] = fmsv;
    tab[VL_FM * ts + at] = fmv;
    RH_VSTAMP(6);   // epilogue arithmetic and stores
}

// ---------------------------------------------------------------------------------
// outside (pull form) + posterior, diagonal d; last group: F5o~[d+1]
// MODE 0: one full launch per diagonal.  MODE 1 / 2 = look-ahead pair (see vlin_inside_diag): the MODE 1 launch of an odd
// diagonal d also accumulates the sums of diagonal d-1 that do not touch row d (FMo / FM1o without their e = 1 terms, the
// filter of length t+1 over the staged row of the filter of length t, the bulge taps) into B.rowp; the MODE 2 launch of d-1
// (one wavefront per group) adds the two e = 1 terms and runs the epilogue.  Pairs are (odd, even) whatever the batch.
template <int W, int BS, bool CUT, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 64 : 64 * W) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 6 : RH_VLA_WPE, 8))) void vlin_outside_diag(McBatch B, const VLinModel* __restrict__ L, int d, int pin, int* __restrict__ bad)
{
    constexpr int WR = MODE == 2 ? 1 : W;
    __shared__ double part[MODE == 1 ? 8 : 4][WR][64];
    __shared__ double gbuf[MODE == 2 ? 1 : W][MODE == 2 ? 1 : 2 * ((kMaxSingle / 2 + W) / W)][MODE == 2 ? 1 : 96];
    int sq, slot;
    block_map_vl(pin, &sq, &slot);
    if (sq >= B.ns) return;
    double* __restrict__ rowp = B.rowp + (size_t)sq * 4 * B.ld;   // look-ahead sums of the next diagonal
    int n = B.n[sq], cut = CUT ? B.cut[sq] : 0;   // one scalar round trip (see vlin_inside_diag); CUT = false: every gap test folds away
    if (CUT) asm volatile("" : "+s"(n), "+s"(cut));
    const int ncell = n - 1 - d > 0 ? n - 1 - d : 0;
    const int d1 = d - 1;
    const int ncell1 = MODE == 1 && d1 >= 0 ? n - 1 - d1 : 0;     // diagonal d-1 has one more cell
    if (ncell < 1 && ncell1 < 1) return;
    const int ngroup = ((ncell > ncell1 ? ncell : ncell1) + 63) >> 6;
    if (!window_slot_vl<CUT>(pin, ngroup, &slot)) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ f5i = B.f5i + sq * ld;
    double* __restrict__ f5o = B.f5o + sq * ld;

    if (slot > ngroup) {
        // XPo~[b], b = cut+d+1 <= n-1, or XSo~[a], a = cut-d >= 2: outside counterparts of the exterior halves (pull form;
        // the pairs around the missing gap that feed them have spans >= d+1 and are final)
        const bool is_xp = slot == ngroup + 1;
        const int b = cut + d + 1, a = cut - d;
        if (is_xp ? b > n - 1 : a < 2) return;
        const double* __restrict__ fco = tab + VL_FCO * ts;
        const double* __restrict__ fca = tab + VL_FCA * ts;
        double acc = 0.0, acc2 = 0.0;
        if (is_xp) {
            const double* __restrict__ xs = B.xs + (size_t)sq * ld;
            const double* __restrict__ xpo = B.xpo + (size_t)sq * ld;
            // pairs (i, b+1) around the missing gap, i on s1: the letters of four of them, then their cells and weights, then the FMAs
            // (a pair that does not exist contributes fma(0, 0, acc) = acc: the same bits as skipping it)
            const int s_b1 = s[b + 1], s_b0 = GAPOK(b) ? s[b] : 0;
            for (int i = 1 + threadIdx.x; i <= cut; i += 4 * 64 * WR) {
                int si[4], sn[4], ic[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int ii = i + u * 64 * WR; ok[u] = ii <= cut; ic[u] = ok[u] ? ii : cut; si[u] = s[ic[u]]; sn[u] = s[ic[u] + 1]; }
                double f[4], t[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ok[u] = ok[u] & (b + 1 - ic[u] >= 4) & (vienna_ptype(si[u], s_b1) != 0);
                    const int ix = 25 * (5 * si[u] + (GAPOK(ic[u]) ? sn[u] : 0)) + 5 * s_b1 + s_b0;
                    f[u] = fco[(b - ic[u]) * ld + ic[u]]; t[u] = L->TNC[ix]; x[u] = xs[ic[u] + 1];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) acc = fma(ok[u] ? f[u] * t[u] : 0.0, ok[u] ? x[u] : 0.0, acc);
            }
            acc2 = dot4_vl(b + 2, n, threadIdx.x, 64 * WR, acc2, [&](int bb) { return xpo[bb]; }, [&](int bb) { return fca[(bb - b - 2) * ld + b + 1]; });
        } else {
            const double* __restrict__ xp = B.xp + (size_t)sq * ld;
            const double* __restrict__ xso = B.xso + (size_t)sq * ld;
            // pairs (a-1, j+1) around the missing gap, j+1 on s2 (see above)
            const int s_a1 = s[a - 1], s_a0 = GAPOK(a - 1) ? s[a] : 0;
            for (int j = cut + threadIdx.x; j <= n - 1; j += 4 * 64 * WR) {
                int sj1[4], sj0[4], jc[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int jj = j + u * 64 * WR; ok[u] = jj <= n - 1; jc[u] = ok[u] ? jj : n - 1; sj1[u] = s[jc[u] + 1]; sj0[u] = s[jc[u]]; }
                double f[4], t[4], x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ok[u] = ok[u] & (jc[u] + 1 - (a - 1) >= 4) & (vienna_ptype(s_a1, sj1[u]) != 0);
                    const int ix = 25 * (5 * s_a1 + s_a0) + 5 * sj1[u] + (GAPOK(jc[u]) ? sj0[u] : 0);
                    f[u] = fco[(jc[u] - a + 1) * ld + a - 1]; t[u] = L->TNC[ix]; x[u] = xp[jc[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) acc = fma(ok[u] ? f[u] * t[u] : 0.0, ok[u] ? x[u] : 0.0, acc);
            }
            acc2 = dot4_vl(1, a - 2, threadIdx.x, 64 * WR, acc2, [&](int aa) { return xso[aa]; }, [&](int aa) { return fca[(a - 2 - aa) * ld + aa]; });
        }
        acc = wsum_vl(acc); acc2 = wsum_vl(acc2);
        if (lane == 0) { part[0][w][0] = acc; part[1][w][0] = acc2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0, t2 = 0.0;
#pragma unroll
            for (int q = 0; q < WR; q++) { t += part[0][q][0]; t2 += part[1][q][0]; }
            if (is_xp) { double* xpo = B.xpo + (size_t)sq * ld; xpo[b] = t + xpo[b + 1] * L->lam + t2 * L->lam2; }
            else { double* xso = B.xso + (size_t)sq * ld; xso[a] = t + xso[a - 1] * L->lam + t2 * L->lam2; }
        }
        return;
    }
    if (slot == ngroup) {
        const int k = d + 1;
        if (k > n - 1) return;   // (MODE 1 on a diagonal beyond this sequence's first)
        const double* __restrict__ fca = tab + VL_FCA * ts + (k + 1);
        double acc = 0.0;
        acc = dot4_vl(k + 2, n, threadIdx.x, 64 * WR, acc, [&](int jj) { return f5o[jj]; }, [&](int jj) { return fca[(jj - 2 - k) * ld]; });
        acc = wsum_vl(acc);
        if (lane == 0) part[0][w][0] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < WR; q++) t += part[0][q][0];
            f5o[k] = f5o[k + 1] * L->lam + t * L->lam2;
        }
        return;
    }

    const int i0 = 1 + slot * 64;
    // two-molecule form: only the joint pair matrix of letters on DIFFERENT strands is wanted (src/ractip.cpp:451-454), and the
    // outside value of a span that covers the missing gap depends on spans that cover it too.  A group whose cells all lie on
    // one strand (letters i..j+1 <= cut, or i > cut; for MODE 1 on both diagonals) is therefore not computed at all: nothing reads it.
    if (CUT && cut > 0 && (i0 + 63 + d + 1 <= cut || i0 > cut)) return;
    const int i = i0 + lane, j = i + d;
    const bool valid = i <= ncell;
    int s_im1 = 0, s_i = 0, s_ip1 = 0, s_j = 0, s_jp1 = 0, s_jp2 = 0;
    if (valid) { s_im1 = s[i - 1]; s_i = s[i]; s_ip1 = s[i + 1]; s_j = s[j]; s_jp1 = s[j + 1]; s_jp2 = s[j + 2]; }
    const int type = (B.allow && valid && !B.allow[((size_t)sq * ld + i) * ld + (j + 1)]) ? 0 : vienna_ptype(s_i, s_jp1);   // 0: excluded by a structure constraint
    const bool pairable = valid && type != 0;
    const bool guard_m = d >= 2;
    // two-molecule form: the sides of an ENCLOSING loop (letters io..i and j+1..jo+1) may not cross the missing gap
    const int l1max = (CUT && valid && cut <= i - 1) ? i - 2 - cut : 99;
    const int l2max = (CUT && valid && cut >= j + 1) ? cut - j - 2 : 99;
    // MODE 1: cell (i, j-1) of diagonal d-1
    const bool valid1 = MODE == 1 && i <= ncell1;
    const int j1 = j - 1;
    const int l1max1 = (CUT && valid1 && cut <= i - 1) ? i - 2 - cut : 99;
    const int l2max1 = (CUT && valid1 && cut >= j1 + 1) ? cut - j1 - 2 : 99;
    const bool guard_m1 = MODE == 1 && d1 >= 2;
    // the limits above bind exactly the cells whose letters lie on one strand (i > cut, or j+1 <= cut), and with a cut those are not
    // wanted (nothing reads their outside values; `wanted` in the posterior): the filters run unmasked and unrolled whenever there is
    // a cut.  vfilt_pair_rev_m (a rolled loop, one LDS and one scalar round trip per tap) set the duration of these launches.
    const bool masked = CUT && cut == 0 && __any(l1max < kMaxSingle || l2max < kMaxSingle || l1max1 < kMaxSingle || l2max1 < kMaxSingle);

    const size_t at = d * ld + i;
    const bool up_ok = i - 1 >= 1 && j + 1 <= n - 1;
    const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1 + s_j;
    const int idd = 25 * (5 * s_jp1 + (GAPOK(j + 1) ? s_jp2 : 0)) + 5 * s_i + (GAPOK(i - 1) ? s_im1 : 0);
    const int idx_raw = 25 * (5 * s_jp1 + s_jp2) + 5 * s_i + s_im1;
    double accm = 0.0, acc1 = 0.0, accc = 0.0, accb = 0.0;
    double accmn = 0.0, acc1n = 0.0, acccn = 0.0, accbn = 0.0;   // MODE 1: diagonal d-1 without its e = 1 terms
    // ---- MODE 1: the epilogue's operands are gathered here by all W wavefronts, four loads each, and meet in LDS at the barrier that
    // collects the partial sums (see vlin_inside_diag).  Wavefront 0: FMSo, FM1o (two cells), FC; wavefront k+1: value and weight of
    // tabulated shape k plus one of TXO TMC TXI TSA tau F5o F5i; the fourth slot: Z, and the two factors of the exterior-half stem.
    constexpr bool DIST = MODE == 1 && W == 8;
    double opA = 0.0, opB = 0.0, opC = 0.0, opD = 0.0;
    int b_im2 = 0, b_im3 = 0, b_jp3 = 0, b_jp4 = 0;
    if constexpr (DIST) {
        if (valid) {
            b_im2 = s[i - 2 >= 0 ? i - 2 : 0]; b_im3 = s[i - 3 >= 0 ? i - 3 : 0];
            b_jp3 = s[j + 3 <= n + 1 ? j + 3 : n + 1]; b_jp4 = s[j + 4 <= n + 1 ? j + 4 : n + 1];
        }
        const int iv = valid ? i : 1, jv = iv + d;
        const int k = w - 1;   // tabulated shape of this wavefront (w >= 1)
        const int l1 = (k == 2 || k == 3 || k == 4) ? 1 : (k >= 5 ? 2 : 0);
        const int l2 = (k == 1 || k == 3 || k == 5) ? 1 : ((k == 4 || k == 6) ? 2 : 0);
        const int io = iv - 1 - l1, jo = jv + 1 + l2;
        const bool sokk = pairable & (io >= 1) & (jo <= n - 1) & (l1 <= l1max) & (l2 <= l2max);
        const int s_io = l1 == 0 ? s_im1 : (l1 == 1 ? b_im2 : b_im3), s_io1 = l1 == 0 ? s_i : (l1 == 1 ? s_im1 : b_im2);
        const int s_jo = l2 == 0 ? s_jp1 : (l2 == 1 ? s_jp2 : b_jp3), s_jo1 = l2 == 0 ? s_jp2 : (l2 == 1 ? b_jp3 : b_jp4);
        const int to = vienna_ptype(s_io, s_jo1);
        const bool ok_so = valid & guard_m & (j + 1 <= n - 1) & GAPOK(j) & GAPOK(j + 1), ok_1o = valid & guard_m & (i - 1 >= 1) & GAPOK(i - 1) & GAPOK(i);
        const bool ok_up = valid & up_ok & GAPOK(i - 1) & GAPOK(j + 1);
        const double* __restrict__ Lt = (const double*)L;
        constexpr size_t oTXO = offsetof(VLinModel, TXO) / 8, oTMC = offsetof(VLinModel, TMC) / 8, oTXI = offsetof(VLinModel, TXI) / 8,
                         oTSA = offsetof(VLinModel, TSA) / 8, oTAU = offsetof(VLinModel, E_tau) / 8;
        const size_t offA = w == 0 ? VL_FMSO * ts + (unsigned)((ok_so ? d + 1 : d) * ld + iv)
                                   : VL_FCO * ts + (unsigned)(sokk ? (jo - io) * ld + io : d * ld + iv);
        const double* __restrict__ pB = w == 0 ? tab + (VL_FM1O * ts + (unsigned)((ok_1o ? d + 1 : d) * ld + (ok_1o ? iv - 1 : iv)))
                                               : small_w_ptr(L, l1, l2, to, type, s_io1, s_jo, s_im1, s_jp2);
        const size_t offC = w == 1 ? oTXO + idx : w == 2 ? oTMC + idx : w == 3 ? oTXI + idx_raw : w == 4 ? oTSA + idd : oTAU + type;
        const double* __restrict__ pC = w == 0 ? tab + (VL_FM1O * ts + (unsigned)((ok_up ? d + 2 : d) * ld + (ok_up ? iv - 1 : iv)))
                                      : w == 6 ? f5o + (jv + 1) : w == 7 ? f5i + (iv - 1) : Lt + offC;
        const double* __restrict__ pD = w == 0 ? tab + (VL_FC * ts + (unsigned)(d * ld + iv)) : f5i + n;
        if constexpr (CUT) {
            const size_t o = (size_t)sq * ld;
            const bool right = valid & (i > cut), left = valid & (j + 1 <= cut);
            if (w == 2) pD = (right ? B.xpo : B.xso) + (o + (right ? j + 1 : iv));
            if (w == 3) pD = (right ? B.xp : B.xs) + (o + (right ? i - 1 : (left ? j + 2 : 1)));
        }
        opA = tab[offA]; opB = *pB; opC = *pC; opD = *pD;
    }
    if constexpr (MODE == 2) {
        // the look-ahead sums of the previous launch + the e = 1 terms (row d+1) + the block products: ten loads, requested together
        // (rows d+1 and 1 exist for every d this launch sees; columns of a lane past the diagonal lie in the table) and selected afterwards
        {
            const double r0 = rowp[i], r1 = rowp[B.ld + i], r2 = rowp[2 * B.ld + i], r3 = rowp[3 * B.ld + i];
            const unsigned up = (unsigned)((d + 1) * ld + i), jc = (unsigned)(j <= n ? j : n);
            const double xa = tab[VL_FM2O * ts + up - 1], ya = tab[VL_FM1 * ts + (unsigned)(ld + i - 1)];
            const double xb = tab[VL_FM2O * ts + up], yb = tab[VL_FM * ts + (unsigned)ld + jc];
            const double fa = tab[VL_FMOF * ts + (unsigned)(d * ld + i)], fb = tab[VL_FM1OF * ts + (unsigned)(d * ld + i)];
            int mineA = i - 1, mineB = n - 1 - j;
            if (BS > 0) {
                const int limA = i - (i / (BS > 0 ? BS : 1) - 1) * BS, limB = (j / (BS > 0 ? BS : 1) + 2) * BS - 1 - j;
                mineA = mineA < limA ? mineA : limA; mineB = mineB < limB ? mineB : limB;
            }
            const bool gm = valid & guard_m;
            accm = gm ? r0 : 0.0; acc1 = gm ? r1 : 0.0; accc = valid ? r2 : 0.0; accb = valid ? r3 : 0.0;
            const double ta = fma(xa, ya, accm), tb = fma(xb, yb, acc1);
            accm = (gm & (1 <= mineA)) ? ta : accm;
            acc1 = (gm & (1 <= mineB)) ? tb : acc1;
            if (BS > 0) { accm = gm ? accm + fa : accm; acc1 = gm ? acc1 + fb : acc1; }
        }
        if (!pairable) { accc = 0.0; accb = 0.0; }
    } else {
    if (guard_m || guard_m1) {
        constexpr int UO = MODE == 1 ? 4 : 6;
        {   // FMo[i,d] += FM2o[d+e][i-e] * FM1[e][i-e]; blocks <= I-2 come from FMOF
            const int ncl = MODE == 1 ? ncell1 : ncell;
            const int i_last = ncl < i0 + 63 ? ncl : i0 + 63;
            const int emax = BS > 0 ? (i_last - 1 < 2 * BS ? i_last - 1 : 2 * BS) : i_last - 1;
            const double* __restrict__ x = tab + VL_FM2O * ts + i;
            const double* __restrict__ y = tab + VL_FM1 * ts + i;
            int mine = valid && guard_m ? i - 1 : 0, mine1 = valid1 && guard_m1 ? i - 1 : 0;
            if (BS > 0) { const int lim = i - (i / (BS > 0 ? BS : 1) - 1) * BS; mine = mine < lim ? mine : lim; mine1 = mine1 < lim ? mine1 : lim; }
            for (int e = 1 + w; e <= emax; e += UO * W) {
                double xv[UO], xn[UO], yv[UO];
#pragma unroll
                for (int u = 0; u < UO; u++) {
                    const int ee = e + u * W;
                    const bool ok = ee <= mine, ok1 = MODE == 1 && ee <= mine1 && ee >= 2;
                    xv[u] = ok ? x[(d + ee) * ld - ee] : 0.0;
                    xn[u] = ok1 ? x[(d1 + ee) * ld - ee] : 0.0;
                    yv[u] = (ok || ok1) ? y[ee * ld - ee] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UO; u++) { accm = fma(xv[u], yv[u], accm); if (MODE == 1) accmn = fma(xn[u], yv[u], accmn); }
            }
            if (BS > 0 && valid && guard_m && w == 0) accm += tab[VL_FMOF * ts + d * ld + i];
        }
        {   // FM1o[i,d] += FM2o[d+e][i] * FM[e][i+d]; blocks >= J+2 come from FM1OF
            const int emax_all = n - 1 - (i0 + (MODE == 1 ? d1 : d));
            const int emax = BS > 0 ? (emax_all < 2 * BS ? emax_all : 2 * BS) : emax_all;
            const double* __restrict__ x = tab + VL_FM2O * ts + i;
            const double* __restrict__ y = tab + VL_FM * ts + j;
            int mine = valid && guard_m ? n - 1 - j : 0, mine1 = valid1 && guard_m1 ? n - 1 - j1 : 0;
            if (BS > 0) {
                const int lim = (j / (BS > 0 ? BS : 1) + 2) * BS - 1 - j, lim1 = (j1 / (BS > 0 ? BS : 1) + 2) * BS - 1 - j1;
                mine = mine < lim ? mine : lim; mine1 = mine1 < lim1 ? mine1 : lim1;
            }
            for (int e = 1 + w; e <= emax; e += UO * W) {
                double xv[UO], yv[UO], xn[UO], yn[UO];
#pragma unroll
                for (int u = 0; u < UO; u++) {
                    const int ee = e + u * W;
                    const bool ok = ee <= mine, ok1 = MODE == 1 && ee <= mine1 && ee >= 2;
                    xv[u] = ok ? x[(d + ee) * ld] : 0.0;
                    yv[u] = ok ? y[ee * ld] : 0.0;
                    xn[u] = ok1 ? x[(d1 + ee) * ld] : 0.0;
                    yn[u] = ok1 ? y[ee * ld - 1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UO; u++) { acc1 = fma(xv[u], yv[u], acc1); if (MODE == 1) acc1n = fma(xn[u], yn[u], acc1n); }
            }
            if (BS > 0 && valid && guard_m && w == 0) acc1 += tab[VL_FM1OF * ts + d * ld + i];
        }
    }
    if constexpr (MODE == 1) {
        // staged row r = 2..31 is table row d+1+r: filter / bulge length t = r-1 of diagonal d, t = r of diagonal d-1
        const int room1 = n - 3 - d;
        if (room1 >= 2) {
            constexpr int LAST = kMaxSingle + 1, HALF = (kMaxSingle + 1) / 2;
            const int rmax = room1 < LAST ? room1 : LAST;
            const double* __restrict__ fcox = tab + VL_FCOX * ts;
            const double* __restrict__ fcob = tab + VL_FCOB * ts;
            constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);
            int rseg[NSEG];
            double bA[NSEG], bB[NSEG];
            // all loads of the staging first, unconditional (a segment that is off reads row d+3, which exists, and is dropped; a
            // column outside the interior lies in a neighbouring row of the table and is replaced by 0): see vlin_inside_diag
            double g0[NSEG], g1[NSEG], x0[NSEG], xa[NSEG], xb[NSEG];
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int r = (q & 1) ? LAST - g : g;
                const bool on = g <= HALF && r <= rmax && r >= 2;
                const int rr = on ? r : 2;
                const double* __restrict__ row = fcox + (d + 1 + rr) * ld + (i0 - 1 - rr);
                const double* __restrict__ rowb = fcob + (d + 1 + rr) * ld + (i - 1);
                g0[q] = row[lane]; g1[q] = row[64 + (lane & 31)];
                x0[q] = rowb[0]; xa[q] = rowb[-(rr - 1)]; xb[q] = rowb[-rr];
            }
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                asm volatile("" : "+v"(g0[q])); asm volatile("" : "+v"(g1[q])); asm volatile("" : "+v"(x0[q])); asm volatile("" : "+v"(xa[q])); asm volatile("" : "+v"(xb[q]));
            }
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int r = (q & 1) ? LAST - g : g;
                const bool on = g <= HALF && r <= rmax && r >= 2;
                rseg[q] = on ? r : -1;
                const int cmax = n - 1 - (d + 1 + r);         // last interior column of the source row
                if (on && r >= 4) {
                    const int c = i0 - 1 - r + lane, c2 = c + 64;
                    gbuf[w][q][lane] = ((c >= 1) & (c <= cmax)) ? g0[q] : 0.0;
                    if (lane < 32) gbuf[w][q][64 + lane] = ((c2 >= 1) & (c2 <= cmax)) ? g1[q] : 0.0;
                }
                // bulges: outer pair at column i-1 (3' side) or i-1-t (5' side) of the source row
                const int tA = r - 1, c0 = i - 1, cA = i - 1 - tA, cB = i - 1 - r;
                const bool in0 = (c0 >= 1) & (c0 <= cmax);
                const bool a0 = on & valid & (tA >= 2) & in0 & (l1max >= 0) & (tA <= l2max);
                const bool a1 = on & valid & (tA >= 2) & (cA >= 1) & (cA <= cmax) & (tA <= l1max) & (l2max >= 0);
                const bool b0 = on & valid1 & (r <= kMaxSingle) & in0 & (l1max1 >= 0) & (r <= l2max1);
                const bool b1 = on & valid1 & (r <= kMaxSingle) & (cB >= 1) & (cB <= cmax) & (r <= l1max1) & (l2max1 >= 0);
                bA[q] = (a0 ? x0[q] : 0.0) + (a1 ? xa[q] : 0.0);
                bB[q] = (b0 ? x0[q] : 0.0) + (b1 ? xb[q] : 0.0);
            }
#pragma unroll
            for (int q = 0; q < NSEG; q++)
                if (rseg[q] >= 0) {
                    const int r = rseg[q];
                    if (r >= 4) {
                        const double* __restrict__ wA = L->shape_w + (r - 1) * r / 2;
                        const double* __restrict__ wB = L->shape_w + (r <= kMaxSingle ? r * (r + 1) / 2 : 0);
                        double sa, sb;
                        if (masked) vfilt_pair_rev_m(r, wA, wB, &gbuf[w][q][lane], l1max, l2max, l1max1, l2max1, sa, sb);
                        else vfilt_pair_rev_any(r, wA, wB, &gbuf[w][q][lane], sa, sb);
                        accc += sa;
                        acccn += sb;
                    }
                    if (r - 1 >= 2) accb = fma(L->WB[r - 1], bA[q], accb);
                    if (r <= kMaxSingle) accbn = fma(L->WB[r], bB[q], accbn);
                }
            if (!pairable) { accc = 0.0; accb = 0.0; }
        }
    } else
    {   // enclosing generic loops (staged, zero-filled outside the interior) and long bulges over FCoB
        const int room = n - 4 - d;
        if (room >= 0) {
            const int tmax = room < kMaxSingle ? room : kMaxSingle;
            const double* __restrict__ fcox = tab + VL_FCOX * ts;
            const double* __restrict__ fcob = tab + VL_FCOB * ts;
            constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);
            int tseg[NSEG];
            double b0[NSEG], b1[NSEG];
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int t = (q & 1) ? kMaxSingle - g : g;
                const bool on = g <= kMaxSingle / 2 && !((q & 1) && t == g) && t <= tmax && t >= 2;
                tseg[q] = on ? t : -1;
                b0[q] = b1[q] = 0.0;
                if (on) {
                    const int cmax = n - 1 - (d + 2 + t);         // last interior column of the source row
                    if (t >= 4) {
                        const int col0 = i0 - 1 - t;
                        const double* __restrict__ row = fcox + (d + 2 + t) * ld;
                        const int c = col0 + lane;
                        gbuf[w][q][lane] = (c >= 1 && c <= cmax) ? row[c] : 0.0;
                        const int c2 = col0 + 64 + lane;
                        if (lane < 32) gbuf[w][q][64 + lane] = (c2 >= 1 && c2 <= cmax) ? row[c2] : 0.0;
                    }
                    if (valid) {   // outer pair (i-1, j+2+t): bulge on the 3' side; (i-1-t, j+2): on the 5' side
                        const double* __restrict__ row = fcob + (d + 2 + t) * ld;
                        const int c0 = i - 1, c1 = i - 1 - t;
                        b0[q] = (c0 >= 1 && c0 <= cmax && l1max >= 0 && t <= l2max) ? row[c0] : 0.0;
                        b1[q] = (c1 >= 1 && c1 <= cmax && t <= l1max && l2max >= 0) ? row[c1] : 0.0;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NSEG; q++)
                if (tseg[q] >= 0) {
                    if (tseg[q] >= 4) {
                        const double* __restrict__ wt = L->shape_w + tseg[q] * (tseg[q] + 1) / 2;
                        accc += masked ? vfilt_rev_m(tseg[q], wt, &gbuf[w][q][lane], l1max, l2max) : vfilt_rev_any(tseg[q], wt, &gbuf[w][q][lane]);
                    }
                    accb = fma(L->WB[tseg[q]], b0[q] + b1[q], accb);
                }
            if (!pairable) { accc = 0.0; accb = 0.0; }
        }
    }
    }
    if constexpr (DIST) { gbuf[w][0][lane] = opA; gbuf[w][1][lane] = opB; gbuf[w][2][lane] = opC; gbuf[w][3][lane] = opD; }   // (its staging rows are dead)
    if constexpr (MODE != 2) {
        part[0][w][lane] = accm;
        part[1][w][lane] = acc1;
        part[2][w][lane] = accc;
        part[3][w][lane] = accb;
        if constexpr (MODE == 1) { part[4][w][lane] = accmn; part[5][w][lane] = acc1n; part[6][w][lane] = acccn; part[7][w][lane] = accbn; }
        __syncthreads();
        if constexpr (MODE == 1) {
            if (w == 1 && i < ld) {   // the look-ahead sums of diagonal d-1
                double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
                for (int k = 0; k < W; k++) { p0 += part[4][k][lane]; p1 += part[5][k][lane]; p2 += part[6][k][lane]; p3 += part[7][k][lane]; }
                rowp[i] = p0; rowp[B.ld + i] = p1; rowp[2 * B.ld + i] = p2; rowp[3 * B.ld + i] = p3;
            }
        }
    }
    if (w != 0 || !valid) return;
    double e_txo = 0, e_tmc = 0, e_txi = 0, e_tsa = 0, e_tau = 1;
    double o_fmso = 0, o_fm1o = 0, o_f5o = 0, o_f5i = 0, o_fm1o_up = 0, o_fc = 0, o_z = 1, sm7 = 0.0, o_x = 0.0;
    {   // epilogue operands: loaded after the term loops so that they do not occupy registers during them, all at once and without a
        // branch in front of any load (see vlin_inside_diag): the indices follow from letters, the letters were requested at the start
        const bool ok_so = guard_m & (j + 1 <= n - 1) & GAPOK(j) & GAPOK(j + 1), ok_1o = guard_m & (i - 1 >= 1) & GAPOK(i - 1) & GAPOK(i);
        const bool ok_up = up_ok & GAPOK(i - 1) & GAPOK(j + 1);
        double l_txo, l_tmc, l_txi, l_tsa, l_tau, l_fmso, l_fm1o, l_up, l_f5o, l_f5i, l_z, l_fc;
        double l_x = 0.0;
        if constexpr (DIST) {   // gathered at the head of the kernel by the wavefront named in the row index
            l_fmso = gbuf[0][0][lane]; l_fm1o = gbuf[0][1][lane]; l_up = gbuf[0][2][lane]; l_fc = gbuf[0][3][lane];
            l_txo = gbuf[1][2][lane]; l_tmc = gbuf[2][2][lane]; l_txi = gbuf[3][2][lane]; l_tsa = gbuf[4][2][lane]; l_tau = gbuf[5][2][lane];
            l_f5o = gbuf[6][2][lane]; l_f5i = gbuf[7][2][lane]; l_z = gbuf[1][3][lane];
            if constexpr (CUT) {
                const bool right = i > cut, left = j + 1 <= cut;
                const double a1 = gbuf[2][3][lane], a2 = gbuf[3][3][lane];
                l_x = (right | left) ? a1 * a2 : 0.0;
            }
        } else {
            b_im2 = s[i - 2 >= 0 ? i - 2 : 0]; b_im3 = s[i - 3 >= 0 ? i - 3 : 0];
            b_jp3 = s[j + 3 <= n + 1 ? j + 3 : n + 1]; b_jp4 = s[j + 4 <= n + 1 ? j + 4 : n + 1];
            l_txo = L->TXO[idx]; l_tmc = L->TMC[idx]; l_txi = L->TXI[idx_raw]; l_tsa = L->TSA[idd]; l_tau = L->E_tau[type];
            l_fmso = tab[VL_FMSO * ts + (unsigned)((ok_so ? d + 1 : d) * ld + i)];
            l_fm1o = tab[VL_FM1O * ts + (unsigned)((ok_1o ? d + 1 : d) * ld + (ok_1o ? i - 1 : i))];
            l_up = tab[VL_FM1O * ts + (unsigned)((ok_up ? d + 2 : d) * ld + (ok_up ? i - 1 : i))];
            l_f5o = f5o[j + 1]; l_f5i = f5i[i - 1]; l_z = f5i[n]; l_fc = tab[VL_FC * ts + at];
            if constexpr (CUT) {   // stem of one of the exterior halves of the loop around the missing gap
                const size_t o = (size_t)sq * ld;
                const bool right = i > cut, left = j + 1 <= cut;
                const double a1 = (right ? B.xpo : B.xso)[o + (right ? j + 1 : i)], a2 = (right ? B.xp : B.xs)[o + (right ? i - 1 : (left ? j + 2 : 1))];
                l_x = (right | left) ? a1 * a2 : 0.0;
            }
        }
        // the seven tabulated shapes: outer pair letters (io, jo+1) = (i-1-l1, j+2+l2), raw outside cell (io, jo)
        const double* __restrict__ fco = tab + VL_FCO * ts;
        double sv[7], sw[7];
        bool sok[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int l1 = (k == 2 || k == 3 || k == 4) ? 1 : (k >= 5 ? 2 : 0);
            const int l2 = (k == 1 || k == 3 || k == 5) ? 1 : ((k == 4 || k == 6) ? 2 : 0);
            const int io = i - 1 - l1, jo = j + 1 + l2;
            sok[k] = pairable & (io >= 1) & (jo <= n - 1) & (l1 <= l1max) & (l2 <= l2max);
            // letters io, io+1 and jo, jo+1 out of the eight held
            const int s_io = l1 == 0 ? s_im1 : (l1 == 1 ? b_im2 : b_im3), s_io1 = l1 == 0 ? s_i : (l1 == 1 ? s_im1 : b_im2);
            const int s_jo = l2 == 0 ? s_jp1 : (l2 == 1 ? s_jp2 : b_jp3), s_jo1 = l2 == 0 ? s_jp2 : (l2 == 1 ? b_jp3 : b_jp4);
            const int to = vienna_ptype(s_io, s_jo1);
            if constexpr (DIST) { sv[k] = gbuf[k + 1][0][lane]; sw[k] = gbuf[k + 1][1][lane]; (void)to; (void)s_io1; (void)s_jo; (void)fco; }
            else {
                sv[k] = fco[(unsigned)(sok[k] ? (jo - io) * ld + io : d * ld + i)];
                sw[k] = small_w(L, l1, l2, to, type, s_io1, s_jo, s_im1, s_jp2);
            }
        }
        e_txo = l_txo; e_tmc = l_tmc; e_txi = l_txi; e_tsa = l_tsa; e_tau = l_tau;
        o_fmso = ok_so ? l_fmso : 0.0; o_fm1o = ok_1o ? l_fm1o : 0.0; o_fm1o_up = ok_up ? l_up : 0.0;
        o_f5o = l_f5o; o_f5i = l_f5i; o_z = l_z; o_fc = l_fc; o_x = l_x;
#pragma unroll
        for (int k = 0; k < 7; k++) sm7 = fma(sok[k] ? sv[k] : 0.0, sw[k], sm7);
    }

    double sm = accm, s1 = acc1, g = accc, gb = accb;
    if constexpr (MODE != 2) {
        sm = 0.0; s1 = 0.0; g = 0.0; gb = 0.0;
#pragma unroll
        for (int k = 0; k < W; k++) { sm += part[0][k][lane]; s1 += part[1][k][lane]; g += part[2][k][lane]; gb += part[3][k][lane]; }
    }

    double fmo = 0.0, fmso = 0.0, fm1o = 0.0;
    if (guard_m) {
        fmo = sm;
        fmso = fmo + o_fmso * L->w_mu;
        fm1o = s1 + fmso + o_fm1o * L->w_mu;
    }
    double fco = 0.0;
    if (pairable) {
        const double ext = (o_f5o * o_f5i + o_x) * L->lam2;
        const double multi = o_fm1o_up * L->w_mp2;
        fco = e_tsa * (ext + multi) + e_txi * g + e_tau * gb + sm7;
    }
    const double fm2o = fmo + fco * e_tmc;
    tab[VL_FCO * ts + at] = fco;
    tab[VL_FCOX * ts + at] = fco * e_txo;
    tab[VL_FCOB * ts + at] = fco * e_tau;
    tab[VL_FMSO * ts + at] = fmso;
    tab[VL_FM1O * ts + at] = fm1o;
    tab[VL_FM2O * ts + at] = fm2o;
    double p = d >= kMinHairpin ? fco * o_fc / o_z : 0.0;
    // (two-molecule form: one-strand cells of a mixed group were computed from spans that were skipped; their values are never read)
    const bool wanted = !(CUT && cut > 0) || (i <= cut && j + 1 > cut);
    if (!(p == p) || p > 1e300) { if (wanted) atomicOr(&bad[sq], 1); p = 0.0; }
    p = p > 1.0 ? 1.0 : (p < 0.0 ? 0.0 : p);
    B.bp[(size_t)sq * B.tri_stride + tri_off_vl(n, i) + (j + 1)] = wanted ? p : 0.0;
}

// logZ = log F5i~[n] + s*n; flags a sequence whose scaled values left the double range
__global__ void vlin_finish(McBatch B, const VLinModel* __restrict__ L, double* __restrict__ logz, int* __restrict__ bad)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const double z = B.f5i[(size_t)sq * B.ld + n];
    const double zo = n >= 2 ? B.f5o[(size_t)sq * B.ld + 1] : 1.0;
    if (!(z > 1e-200 && z < 1e200) || !(zo > 1e-200 && zo < 1e200)) atomicOr(&bad[sq], 1);
    logz[sq] = log(z) + L->s * (double)n;
}

// =================================================================================
// accessibility from the linear tables (same decomposition as mccaskill_vienna.hip; every product of an outside and an
// inside quantity divided by Z~ is already free of the scaling, each unpaired letter of the run contributes one lam)
constexpr int VL_S_HP = VL_FM2F;   // scratch: hairpin probabilities, square [p][q] (the far sums of the inside sweep are dead)

// Hp[p][q] = P(letters p, q close a hairpin loop); hplen[d] = lam^d * length weight (host table, as the inside sweep's hp_d)
__global__ __launch_bounds__(256) void vlin_acc_prep(McBatch B, const VLinModel* __restrict__ L, const double* __restrict__ hplen)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int tiles = (ld + 31) / 32;
    const int tr = blockIdx.x / tiles, tc = blockIdx.x % tiles;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    const double Z = B.f5i[(size_t)sq * ld + n];
    double* __restrict__ hp = tab + VL_S_HP * ts;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = tr * 32 + ty + 8 * k, q = tc * 32 + tx;
        if (p >= ld || q >= ld) continue;
        double v = 0.0;
        const int u = q - p - 1;
        if (p >= 1 && q <= n && u >= kMinHairpin) {
            const int type = vienna_ptype(s[p], s[q]);
            if (type) {
                double e = hplen[u];
                if (u == 3) e *= L->E_tau[type];
                else {
                    e *= L->TMH[25 * (5 * s[p] + s[p + 1]) + 5 * s[q] + s[q - 1]];
                    if (u == 4) {
                        int code = 0; bool ok = true;
#pragma unroll
                        for (int c = 0; c < 6; c++) { const int x = s[p + c]; ok = ok && x != 0; code = code * 4 + (x - 1); }
                        if (ok) e *= L->E_tetra[code];
                    }
                }
                v = tab[VL_FCO * ts + (size_t)u * ld + p] * e / Z;
            }
        }
        hp[(size_t)p * ld + q] = v;
    }
}

// gap probabilities of interior loops, one THREAD per (letter, own gap length g >= 1), lanes over the letter:
//   z <  30: GL[p][g] = sum over loops with outer 5' letter p and g unpaired letters p+1..p+g
//   z >= 30: GR[q][g] = the same for the 3' gap q-g..q-1 of the outer 3' letter q
// For a fixed thread the inner pair's 5' letter k = p+1+g (left) resp. 3' letter l = q-1-g (right) is fixed and the
// loops are (inner span r, other gap o): inner cell (k, k+r) resp. (l-1-r, l-1), outer span D = r+2+g+o.  The outer
// values needed for r+1 are those of r shifted by one o: they live in a 31-entry register window, one new load per r
// (instead of 31), the generic weights w(g,o) in scalar registers.  Bulges and the tabulated small loops are a handful
// of (g,o) combinations and are added with direct loads.
__global__ __launch_bounds__(256) void vlin_acc_gaps(McBatch B, const VLinModel* __restrict__ L, double* __restrict__ gaps, int ng, int nchunk, double* __restrict__ part)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    // blockIdx.z = chunk * 2 ng + slice: gap lengths 1..ng per side; nchunk > 1: the inner spans are dealt to nchunk workgroups in
    // contiguous ranges and the partial sums go to `part` (vlin_acc_gsum adds them in chunk order)
    const int chunk = blockIdx.z / (2 * ng), slice = blockIdx.z % (2 * ng);
    const int g = slice % ng + 1;
    const bool right = slice >= ng;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x + 1;   // p (left) or q (right)
    const bool live = pos <= n;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    const double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const double Z = B.f5i[(size_t)sq * ld + n];
    const double* __restrict__ FCO = tab + VL_FCO * ts;
    const double* __restrict__ FCOX = tab + VL_FCOX * ts;
    const double* __restrict__ FCOB = tab + VL_FCOB * ts;
    const double* __restrict__ FC = tab + VL_FC * ts;
    const double* __restrict__ FCX = tab + VL_FCX * ts;
    const double* __restrict__ FCB = tab + VL_FCB * ts;
    constexpr int NW = kMaxSingle + 1;
    // generic weights of (own gap g, other gap o); 0 for bulges, tabulated shapes and o beyond the loop budget
    double wt[NW];
#pragma unroll
    for (int o = 0; o < NW; o++) {
        const int t = g + o;
        wt[o] = t <= kMaxSingle ? L->shape_w[t * (t + 1) / 2 + (right ? o : g)] : 0.0;
    }
    // outer cell of (r, o): span D = r+2+g+o, column p (left) resp. q-1-D (right); 0 outside the interior
    auto outer_at = [&](const double* __restrict__ T, int D) -> double {
        const int pc = right ? pos - 1 - D : pos;
        return (live && pc >= 1 && pc + D <= n - 1) ? T[(size_t)D * ld + pc] : 0.0;
    };
#ifndef RH_ACC_UR
#define RH_ACC_UR 2   // measured: 2 -> 28.2 ms (outside + accessibility), 4 -> 28.7, 8 -> 31.6: the kernel runs at the copy bandwidth (33 GB per 128 sequences)
#endif
    // inner spans per batch of loads.  The window holds the outer values of the WHOLE batch (NW + UR - 1 of them), so that it slides by UR
    // once per batch (30 register moves per UR spans; sliding by one per span cost as many moves as the span has FMAs)
    constexpr int UR = RH_ACC_UR;
    double win[NW + UR - 1];
    const int rcs = (((B.nmax + nchunk - 1) / nchunk) + UR - 1) / UR * UR, r_begin = chunk * rcs;   // this workgroup's inner spans: r_begin .. r_begin+rcs-1
#pragma unroll
    for (int o = 0; o < NW + UR - 1; o++) win[o] = outer_at(FCOX, r_begin + 2 + g + o);
    const int kl = right ? pos - 1 - g : pos + 1 + g;   // inner 3' letter l (right) resp. inner 5' letter k (left)
    const int rmax = right ? kl - 2 : n - 1 - kl;       // inner spans 0..rmax are interior
    double acc = 0.0;
    // letters of the tabulated shapes (g <= 2).  One end of the inner pair and of every outer pair is fixed for a thread; the other
    // ends are the four letters c0 + dir*(r + j), j = 0..3 (left: l, l+1 and q = l+1+o, q-1; right: k, k-1 and p = k-1-o, p+1), a
    // window that slides by one per inner span: one new letter per span instead of sixteen loads
    const auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    const int c0 = right ? kl - 1 : kl + 1, dir = right ? -1 : 1;
    const auto letter = [&](int j) -> int { return s[clampi(c0 + dir * j, n + 1)]; };
    const int f_in0 = s[clampi(kl, n + 1)], f_in1 = s[clampi(right ? kl + 1 : kl - 1, n + 1)];       // left: k, k-1; right: l, l+1
    const int f_out0 = s[clampi(pos, n + 1)], f_out1 = s[clampi(right ? pos - 1 : pos + 1, n + 1)];   // left: p, p+1; right: q, q-1
    int V[UR + 3];
#pragma unroll
    for (int j = 0; j < UR + 3; j++) V[j] = g <= 2 ? letter(r_begin + j) : 0;
    // largest interior inner span of any thread of this block (threads are consecutive letters)
    const int pos_lo = blockIdx.x * blockDim.x + 1, pos_hi = pos_lo + (int)blockDim.x - 1 < n ? pos_lo + (int)blockDim.x - 1 : n;
    const int rlim = right ? pos_hi - 1 - g - 2 : n - 1 - (pos_lo + 1 + g);
    const int r_end = rlim < r_begin + rcs - 1 ? rlim : r_begin + rcs - 1;
    for (int r0 = r_begin; r0 <= r_end; r0 += UR) {
        double xs[UR], wn[UR], bo[UR], bi[UR];
        int nl[UR];
        bool ok[UR];
        size_t ics[UR];
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int r = r0 + u;
            ok[u] = live && r <= rmax && kl >= 1;
            const int kc = right ? kl - 1 - r : kl;          // inner cell column
            ics[u] = (size_t)r * ld + (ok[u] ? kc : 1);
            xs[u] = ok[u] ? FCX[ics[u]] : 0.0;
            wn[u] = outer_at(FCOX, r0 + UR + 2 + g + (NW - 1) + u);   // enters the window of the next batch at index NW-1+u
            bo[u] = (ok[u] && g >= 2) ? outer_at(FCOB, r + 2 + g) : 0.0;   // bulge (own gap g >= 2, other gap 0)
            bi[u] = (ok[u] && g >= 2) ? FCB[ics[u]] : 0.0;
            nl[u] = g <= 2 ? letter(r0 + UR + 3 + u) : 0;     // enters the letter window of the next batch at index 3+u
        }
#pragma unroll
        for (int u = 0; u < UR; u++) {
            const int r = r0 + u;
            double sum = 0.0;
#pragma unroll
            for (int o = 0; o < NW; o++) sum = fma(wt[o], win[u + o], sum);
            acc = fma(xs[u], sum, acc);
            acc = fma(bo[u] * L->WB[g], bi[u], acc);
            // the tabulated shapes that involve this gap length (g <= 2: four of the sixty slices).  Two batches of loads -- letters and
            // table cells, then the loop weights -- with no data-dependent branch in front of any of them: behind `if (fc != 0)` ...
            // `if (fo == 0) continue` every cell paid four or five dependent round trips, and these slices set the kernel's duration
            if (g <= 2) {
                const int k = right ? kl - 1 - r : kl;                         // inner pair (k, l = k+r+1)
                const double fc = FC[ics[u]];
                const int sk = right ? V[u] : f_in0, skm = right ? V[u + 1] : f_in1, sl = right ? f_in0 : V[u], slp = right ? f_in1 : V[u + 1];
                const int ti = vienna_ptype(sk, sl);
                double fo[3], sw[3];
                bool use[3];
#pragma unroll
                for (int o = 0; o <= 2; o++) {
                    const int l1 = right ? o : g, l2 = right ? g : o;
                    const bool tabulated = (l1 <= 2 && l2 <= 2) && !(l1 + l2 == 2 && (l1 == 0 || l2 == 0));   // 0x2 / 2x0 are bulges
                    const int D = r + 2 + g + o;
                    const int p = k - 1 - l1, q = p + D + 1;
                    use[o] = ok[u] & tabulated & (p >= 1) & (q <= n);
                    fo[o] = FCO[(size_t)(use[o] ? D : 0) * ld + (use[o] ? p : 1)];
                    const int sp = right ? V[u + 1 + o] : f_out0, spp = right ? V[u + o] : f_out1;
                    const int sq_ = right ? f_out0 : V[u + 1 + o], sqm = right ? f_out1 : V[u + o];
                    sw[o] = small_w(L, l1, l2, vienna_ptype(sp, sq_), ti, spp, sqm, skm, slp);
                }
#pragma unroll
                for (int o = 0; o <= 2; o++) acc = fma(use[o] ? fo[o] * sw[o] : 0.0, use[o] ? fc : 0.0, acc);
            }
        }
        // slide the window by the batch
#pragma unroll
        for (int o = 0; o + 1 < NW; o++) win[o] = win[o + UR];
#pragma unroll
        for (int u = 0; u < UR; u++) win[NW - 1 + u] = wn[u];
#pragma unroll
        for (int j = 0; j < 3; j++) V[j] = V[j + UR];
#pragma unroll
        for (int u = 0; u < UR; u++) V[3 + u] = nl[u];
    }
    // layout [side][g][pos]: consecutive lanes = consecutive letters (vlin_acc_gsuf / vlin_acc_final read it the same way)
    if (live) {
        if (nchunk == 1) gaps[((size_t)(2 * sq + (right ? 1 : 0)) * 32 + g) * ld + pos] = acc / Z;
        else part[(((size_t)chunk * 2 * B.ns + 2 * sq + (right ? 1 : 0)) * ng + (g - 1)) * ld + pos] = acc / Z;
    }
}

// gaps[side][g][pos] = sum over the chunks of vlin_acc_gaps, in chunk order (g = 1..ng)
__global__ __launch_bounds__(256) void vlin_acc_gsum(McBatch B, double* __restrict__ gaps, const double* __restrict__ part, int ng, int nchunk)
{
    const int sq = blockIdx.y, side = blockIdx.z / ng, g = blockIdx.z % ng + 1;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (pos > B.n[sq]) return;
    double acc = 0.0;
    for (int c = 0; c < nchunk; c++) acc += part[(((size_t)c * 2 * B.ns + 2 * sq + side) * ng + (g - 1)) * B.ld + pos];
    gaps[((size_t)(2 * sq + side) * 32 + g) * B.ld + pos] = acc;
}

// The same sums for the gap lengths g = 3..30 (no tabulated shape involves them) with the LANES over the pairs of gap lengths: one
// letter per wavefront.  The pairs (own gap g, other gap o) with a generic weight are the triangle g + o <= 30; cut into runs of eight
// consecutive o for one g it has exactly 64 runs (kGapJob), one per lane, so a lane does 8 FMAs per inner span and 4 of 5 are useful
// (lanes over g alone: half).  For one letter the outer values of all its (g, o) are ONE sequence in the outer span D = r+2+g+o --
// FCoX[D][p] (left) / FCoX[D][q-1-D] (right), and FCoB likewise for the bulges -- kept as a ring of 56 spans in LDS (mirrored, so a
// lane reads base+offset without wrapping): a block of eight inner spans r0.. reads the spans r0+5 .. r0+47, and sixteen lanes load
// the next eight spans of both rings one block ahead.  A lane keeps its 8 weights w(g, o) AND its window of 8 outer values in
// registers: the window slides by one per step, so a step reads one new value from the ring, and the steps are unrolled in blocks of
// 8 -- a seventh of the ring -- so that the window rotates through fixed registers (other gap 8c+oo at step u of a block is
// win[(u + oo) mod 8]) and every LDS address of a block is one per-block register plus an immediate.  Inner cells FCX / FCB[r][column]:
// one load pair per step, RH_ACCW_PF steps ahead, at 32-bit offsets from the sequence's table block that advance by the row pitch
// and stop at the lane's last interior cell (the host falls back to vlin_acc_gaps when a table block exceeds 4 GB).  A lane past its
// last span is masked out of the arithmetic.
// vlin_acc_gaps (one thread per letter AND gap length, 31-entry register window, UR = 2 spans per batch of loads) re-reads every table
// once per gap length; a first version of this kernel that read its whole window from LDS every step ran at the LDS bandwidth.
__device__ const unsigned char kGapJob[64] = {   // g + 32 c: lane's own gap length and run of other gaps o = 8c .. 8c+7
    3, 35, 67, 99, 4, 36, 68, 100, 5, 37, 69, 101, 6, 38, 70, 102, 7, 39, 71, 8, 40, 72, 9, 41, 73, 10, 42, 74, 11, 43, 75, 12, 44, 76, 13, 45, 77, 14, 46, 78, 15, 47, 16, 48, 17, 49, 18, 50, 19, 51, 20, 52, 21, 53, 22, 54, 23, 24, 25, 26, 27, 28, 29, 30 };
#ifndef RH_ACCW_PF
#define RH_ACCW_PF 4
#endif
__global__ __launch_bounds__(256) void vlin_acc_gaps_wide(McBatch B, const VLinModel* __restrict__ L, double* __restrict__ gaps)
{
    constexpr int NO = 8;                                    // other gaps per lane
    constexpr int RN = 7 * NO;                               // ring entries
    __shared__ double rings[4][2][2 * RN];
    __shared__ double red[4][64 + 4];
    const int sq = blockIdx.y;
    const bool right = blockIdx.z != 0;
    const int n = B.n[sq], ld = B.ld;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pos = blockIdx.x * 4 + w + 1;                  // the wavefront's letter
    if (pos > n) return;                                     // wave-uniform
    const int job = kGapJob[lane], g = job & 31, c = job >> 5;
    const double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const double Z = B.f5i[(size_t)sq * ld + n];
    double wr[NO];
#pragma unroll
    for (int oo = 0; oo < NO; oo++) {
        const int o = NO * c + oo, t = g + o;
        const bool ok = t <= kMaxSingle;              // (unconditional loads, selected afterwards: behind a branch every weight is a round trip of its own)
        const double v = L->shape_w[ok ? t * (t + 1) / 2 + (right ? o : g) : 0];
        wr[oo] = ok ? v : 0.0;                        // 0 for o = 0 (bulge) as in vlin_acc_gaps
    }
    const double wbv = L->WB[g];
    const double wb = c == 0 ? wbv : 0.0;
    double* const ring = &rings[w][0][0];
    double* const ringb = &rings[w][1][0];
    // outer cell of span D for this letter = column p (left) resp. q-1-D (right); 0 outside the interior
    const auto outer_ix = [&](int D, bool* ok) -> unsigned {
        const int pc = right ? pos - 1 - D : pos;
        *ok = pc >= 1 && pc + D <= n - 1;
        return (unsigned)(*ok ? D * ld + pc : 1);
    };
    if (lane < RN) {   // spans 0..55 of both rings, one per lane
        bool ok;
        const unsigned ix = outer_ix(lane, &ok);
        const double a = tab[VL_FCOX * ts + ix], b = tab[VL_FCOB * ts + ix];
        ring[lane] = ring[lane + RN] = ok ? a : 0.0;
        ringb[lane] = ringb[lane + RN] = ok ? b : 0.0;
    }
    // lanes 0..15 feed the rings: lane f loads span r0+56+(f & 7) of FCoX (f < 8) / FCoB during block r0 and stores it at the start of block r0+8
    const bool feeder = lane < 2 * NO;
    const double* __restrict__ ftab = tab + (lane < NO ? VL_FCOX : VL_FCOB) * ts;
    double* const fdst = (lane < NO ? ring : ringb) + (lane & (NO - 1));
    double fv = 0.0;     // (raw: selected where it is stored, one block after its load -- selected at once, the load is waited for at once)
    bool fok = false;
    const int kl = right ? pos - 1 - g : pos + 1 + g;        // inner 3' letter l (right) resp. inner 5' letter k (left)
    const int rmax = right ? kl - 2 : n - 1 - kl;            // inner spans 0..rmax are interior
    const int rl = right ? pos - 6 : n - pos - 5;            // the largest of them in this wavefront (g = 3)
    // inner cells: down one table column (left) or anti-diagonal (right): cell = first + r * stride for r <= last, the cell of `last` afterwards
    const int stride8 = (right ? ld - 1 : ld) * 8;
    const int last = kl < 1 ? -1 : rmax;
    const bool any = last >= 0;
    const int first_ix = !any ? 1 : (right ? kl - 1 : kl);
    const char* const tab8 = (const char*)tab;
    unsigned off = (unsigned)(((size_t)VL_FCX * ts + first_ix) * 8);            // byte offset of the FCX cell of step r
    const unsigned offmax = off + (unsigned)(any ? last : 0) * (unsigned)stride8;
    const unsigned d_b = (unsigned)(((size_t)VL_FCB - (size_t)VL_FCX) * ts * 8);   // (mod 2^32) the same cell of FCB
    constexpr int PF = RH_ACCW_PF;   // inner cells are loaded PF steps ahead (slot r % PF; NO % PF == 0 keeps the slots fixed registers)
    static_assert(NO % PF == 0, "prefetch slots rotate within a block");
    double xa[PF], xb[PF];
#pragma unroll
    for (int u = 0; u < PF; u++) {
        xa[u] = *(const double*)(tab8 + off); xb[u] = *(const double*)(tab8 + (off + d_b));
        if (u + 1 < PF) { const unsigned o2 = off + stride8; off = o2 < offmax ? o2 : offmax; }
    }
    typedef const volatile __attribute__((address_space(3))) double* lds_rp;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the rings are written and read by this wavefront only
    double win[NO];
    {
        const lds_rp rp = (lds_rp)(ring + 2 + g + NO * c);   // (<= 39: no wrap)
#pragma unroll
        for (int oo = 0; oo < NO; oo++) win[oo] = rp[oo];
    }
    const int c_a = g + NO * c, c_b = g;                     // <= 30
    double acc = 0.0, accb = 0.0;
    int r0m = 0;                                             // r0 mod RN
    for (int r0 = 0; r0 <= rl; r0 += NO) {
        if (feeder) {
            if (r0 > 0) {   // spans r0+48 .. r0+55 take the place of r0-8 .. r0-1
                double* const d = fdst + (r0m == 0 ? RN - NO : r0m - NO);
                const double v = fok ? fv : 0.0;
                d[0] = v; d[RN] = v;
            }
            const unsigned ix = outer_ix(r0 + RN + (lane & (NO - 1)), &fok);
            fv = ftab[ix];
        }
        int ba = r0m + c_a, bb = r0m + c_b;
        ba = ba >= RN ? ba - RN : ba; bb = bb >= RN ? bb - RN : bb;
        const lds_rp rpa = (lds_rp)(ring + ba), rpb = (lds_rp)(ringb + bb);
#pragma unroll
        for (int u = 0; u < NO; u++) {
            const int r = r0 + u;
            const double nv = rpa[u + 2 + NO];                // span r+2+g+8c+8: other gap 8c+7 of the next step
            const double bv = rpb[u + 2];                     // bulge: own gap g, other gap 0, outer span r+2+g
            if (r <= last) {
                double s = 0.0;
#pragma unroll
                for (int oo = 0; oo < NO; oo++) s = fma(wr[oo], win[(u + oo) % NO], s);
                acc = fma(xa[u % PF], s, acc);
                accb = fma(xb[u % PF], bv, accb);
            }
            win[u % NO] = nv;
            // loads of step r+PF, into the slots this step has just read (behind the FMAs: in front of them the compiler copies the operands)
            { const unsigned o2 = off + stride8; off = o2 < offmax ? o2 : offmax; }
            xa[u % PF] = *(const double*)(tab8 + off); xb[u % PF] = *(const double*)(tab8 + (off + d_b));
            asm volatile("" : "+v"(acc));   // (a step's FMAs stay in front of the next step's volatile ring reads)
        }
        r0m = r0m + NO == RN ? 0 : r0m + NO;
    }
    acc = fma(wb, accb, acc);
    // the runs of one gap length are consecutive lanes: the first adds them up
    red[w][lane] = acc;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int nc = (31 - g + NO - 1) / NO;
#pragma unroll
    for (int q = 1; q < 4; q++) { const double v = ((lds_rp)&red[w][lane])[q]; if (q < nc) acc += v; }
    // layout [side][g][pos] as vlin_acc_gaps
    if (c == 0) gaps[((size_t)(2 * sq + (right ? 1 : 0)) * 32 + g) * ld + pos] = acc / Z;
}

// gap probabilities -> suffix sums over the gap length: S[g][pos] = sum_{l >= g} G[l][pos]; one thread per (letter, side)
__global__ __launch_bounds__(256) void vlin_acc_gsuf(McBatch B, double* __restrict__ gaps)
{
    const int sq = blockIdx.y, side = blockIdx.z;
    const int ld = B.ld;
    const int pos = blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (pos > B.n[sq]) return;
    double* __restrict__ G = gaps + (size_t)(2 * sq + side) * 32 * ld + pos;
    double run = 0.0;
    for (int g = kMaxSingle; g >= 1; g--) { run += G[(size_t)g * ld]; G[(size_t)g * ld] = run; }
}

// up[(a-1)*max_w + w] = H part: sum_{p<a, q>a+w} Hp[p][q] from the column prefix sums C (square scratch); one wavefront per letter
__global__ __launch_bounds__(256) void vlin_acc_hsum(McBatch B, int max_w)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int a = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 1;
    const int lane = threadIdx.x & 63;
    if (a > n) return;
    const double* __restrict__ C = B.tab + (size_t)sq * B.seq_stride + VL_S_HP * B.tab_stride + (size_t)a * ld;
    double* __restrict__ up = B.up + ((size_t)sq * ld + (a - 1)) * max_w;
    // tail beyond the widest region, then one more column per narrower width
    double tail = 0.0;
    for (int q = a + max_w + lane; q <= n; q += 64) tail += C[q];
    tail = wsum_vl(tail);
    if (lane == 0) {
        double run = tail;
        for (int w = max_w - 1; w >= 0; w--) {
            const int b = a + w;
            up[w] = b <= n ? run : 0.0;
            if (b <= n) run += C[b];     // width w-1 also counts q = b
        }
    }
}

// adds the E, I and M parts; one THREAD per (letter a, width w), lanes over a
__global__ __launch_bounds__(256) void vlin_acc_final(McBatch B, const VLinModel* __restrict__ L, const double* __restrict__ gaps, int max_w)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int a = blockIdx.x * blockDim.x + threadIdx.x + 1;
    const int w = blockIdx.z;
    if (a > n) return;
    const int b = a + w, len = w + 1;
    double* __restrict__ up = B.up + ((size_t)sq * ld + (a - 1)) * max_w + w;
    if (b > n) { *up = 0.0; return; }
    const double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const double* __restrict__ f5o = B.f5o + (size_t)sq * ld;
    const double Z = f5i[n];
    const double* __restrict__ gl = gaps + (size_t)(2 * sq) * 32 * ld;       // suffix sums over the gap length, [g][pos]
    const double* __restrict__ gr = gaps + (size_t)(2 * sq + 1) * 32 * ld;
    double lam_len = 1.0, mu_len = 1.0;
    for (int k = 0; k < len; k++) { lam_len *= L->lam; mu_len *= L->w_mu; }
    double acc = f5i[a - 1] * f5o[b] / Z * lam_len;                                          // E
    // I: the region lies in the 5' gap of the loop whose outer 5' letter is p < a when that gap reaches b: length >= b-p
    for (int p = a - 1; p >= 1 && p >= b - kMaxSingle; p--) acc += gl[(size_t)(b - p) * ld + p];
    for (int q = b + 1; q <= n && q <= a + kMaxSingle; q++) acc += gr[(size_t)(q - a) * ld + q];   // ... 3' gaps
    double m = 0.0;
    if (a >= 2 && b <= n - 3) {     // M, run before a branch: FM1o[a-1, j] * FM1[b, j], j = a+e
        const double* __restrict__ x = tab + VL_FM1O * ts + (a - 1);
        const double* __restrict__ y = tab + VL_FM1 * ts + b;
        for (int e = w + 2; a + e <= n - 1; e++) m = fma(x[(size_t)(e + 1) * ld], y[(size_t)(e - w) * ld], m);
    }
    if (a >= 4 && b <= n - 1) {     // M, run after the last branch: FMSo[i, b] * FMS[i, a-1], i = a-1-e
        const double* __restrict__ x = tab + VL_FMSO * ts;
        const double* __restrict__ y = tab + VL_FMS * ts;
        for (int e = 2; a - 1 - e >= 1; e++) {
            const int i = a - 1 - e;
            m = fma(x[(size_t)(e + len) * ld + i], y[(size_t)e * ld + i], m);
        }
    }
    acc += m * mu_len / Z;
    acc += *up;                                                                               // H (vlin_acc_hsum)
    *up = acc > 1.0 ? 1.0 : acc;
}

// The same with one THREAD per letter and all widths w < 15 in it (vlin_acc_final has one thread per letter AND width: two loads per FMA,
// fifteen times over).  The two multiloop streams are sums over a run of cells whose operands for the fifteen widths overlap:
//   run before a branch: sum_r FM1o[r+w+1][a-1] * FM1[r][a+w]: rows r+1 .. r+15 of ONE column in a ring of 16 staged rows (read at lane),
//     and ONE row of FM1 staged per step (read at lane + w);
//   run after the last branch: sum_e FMSo[e+w+1][a-1-e] * FMS[e][a-1-e]: the second factor does not depend on w (a register), the first
//     comes from a ring of 16 staged row segments (row R: columns a0-R .. a0-R+77 of the wavefront's 64 letters), again at lane + w.
// Everything is private to a wavefront (LDS operations of one wavefront execute in order: no barriers).  The staged values are loaded
// kPF steps before the step that stores them (registers), so that a step's loads have kPF steps of FMAs to arrive.  Cells outside the
// triangle (stale bytes) and terms outside a width's range are zeroed where they are loaded, so the FMAs carry no masks; every width
// adds its terms in the order of vlin_acc_final (plus exact zeros): the same bits.
constexpr int kAccW = 15, kAccPF = 4;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void vlin_acc_final_t(McBatch B, const VLinModel* __restrict__ L, const double* __restrict__ gaps, int max_w)
{
    __shared__ double ring_s[4][16][80];
    __shared__ double yrow_s[4][2][80];
    typedef const volatile __attribute__((address_space(3))) double* lds_vp;
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a0 = blockIdx.x * 256 + wv * 64 + 1, a = a0 + lane;
    if (a0 > n) return;                                        // wave-uniform
    const bool live = a <= n;
    const double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const double* __restrict__ f5o = B.f5o + (size_t)sq * ld;
    const double Z = f5i[n];
    const auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    double (*ring)[80] = ring_s[wv];
    double m[kAccW];
#pragma unroll
    for (int w = 0; w < kAccW; w++) m[w] = 0.0;
    {   // ---- M, run before a branch: r = e - w = 2 .. n-1-a-w
        const double* __restrict__ FM1O = tab + VL_FM1O * ts;
        const double* __restrict__ FM1 = tab + VL_FM1 * ts;
        const int ac = clampi(a - 1, ld - 1);
        const auto xcell = [&](int R) -> double {              // FM1o[R][a-1] where the term exists (a >= 2, R <= n-a), else 0
            const double v = FM1O[(size_t)clampi(R, ld - 1) * ld + ac];
            return (live & (a >= 2) & (R <= n - a)) ? v : 0.0;
        };
        const auto ycell = [&](int r, int c) -> double {       // FM1[r][c] inside the triangle with b = c <= n-3, else 0
            const double v = FM1[(size_t)clampi(r, ld - 1) * ld + clampi(c, ld - 1)];
            return ((c <= n - 3) & (r + c <= n - 1)) ? v : 0.0;
        };
        const int rend = n - 1 - a0;
        if (rend >= 2) {
#pragma unroll
            for (int R = 3; R <= 17; R++) ring[R & 15][lane] = xcell(R);      // rows r+1 .. r+15 of the first step r = 2
        }
        double px[kAccPF], py[kAccPF], py2[kAccPF];                           // loaded for the steps r .. r+kPF-1
#pragma unroll
        for (int k = 0; k < kAccPF; k++) { px[k] = xcell(2 + k + 16); py[k] = ycell(2 + k, a0 + lane); py2[k] = ycell(2 + k, a0 + 64 + (lane < 14 ? lane : 13)); }
        for (int r0 = 2; r0 <= rend; r0 += 16) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int r = r0 + u;                                          // r & 15 = (u + 2) & 15
                double* const yb = yrow_s[wv][u & 1];
                yb[lane] = py[u % kAccPF];
                if (lane < 14) yb[64 + lane] = py2[u % kAccPF];
                const double xnew = px[u % kAccPF];
                // the loads of step r + kPF
                px[u % kAccPF] = xcell(r + kAccPF + 16); py[u % kAccPF] = ycell(r + kAccPF, a0 + lane); py2[u % kAccPF] = ycell(r + kAccPF, a0 + 64 + (lane < 14 ? lane : 13));
                const lds_vp yr = (lds_vp)(yb + lane);
#pragma unroll
                for (int w = 0; w < kAccW; w++) m[w] = fma(((lds_vp)&ring[(u + 3 + w) & 15][lane])[0], yr[w], m[w]);
                ring[(u + 2) & 15][lane] = xnew;                               // row r+16 takes the slot of row r
                asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), "+v"(m[11]), "+v"(m[12]), "+v"(m[13]), "+v"(m[14]) :: "memory");   // (steps stay in order: hoisted, a block's loads spill)
            }
        }
    }
    {   // ---- M, run after the last branch: e = 2 .. a-2
        const double* __restrict__ FMSO = tab + VL_FMSO * ts;
        const double* __restrict__ FMS = tab + VL_FMS * ts;
        const auto xcell = [&](int R, int k) -> double {       // FMSo[R][a0-R+k] inside the triangle, else 0
            const int C = a0 - R + k;
            const double v = FMSO[(size_t)clampi(R, ld - 1) * ld + clampi(C, ld - 1)];
            return ((C >= 1) & (R + C <= n - 1)) ? v : 0.0;
        };
        const auto ycell = [&](int e) -> double {              // FMS[e][a-1-e] where the term exists (a >= 4, a-1-e >= 1), else 0
            const int i = a - 1 - e;
            const double v = FMS[(size_t)clampi(e, ld - 1) * ld + clampi(i, ld - 1)];
            return (live & (a >= 4) & (i >= 1)) ? v : 0.0;
        };
        const int amax = a0 + 63 < n ? a0 + 63 : n, eend = amax - 2;
        const int l2 = lane < 14 ? lane : 13;
        if (eend >= 2) {
#pragma unroll
            for (int R = 3; R <= 17; R++) { ring[R & 15][lane] = xcell(R, lane); if (lane < 14) ring[R & 15][64 + lane] = xcell(R, 64 + l2); }
        }
        double qx[kAccPF], qx2[kAccPF], qy[kAccPF];
#pragma unroll
        for (int k = 0; k < kAccPF; k++) { qx[k] = xcell(2 + k + 16, lane); qx2[k] = xcell(2 + k + 16, 64 + l2); qy[k] = ycell(2 + k); }
        for (int e0 = 2; e0 <= eend; e0 += 16) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int e = e0 + u;                                          // e & 15 = (u + 2) & 15
                const double y = qy[u % kAccPF], xnew = qx[u % kAccPF], xnew2 = qx2[u % kAccPF];
                qx[u % kAccPF] = xcell(e + kAccPF + 16, lane); qx2[u % kAccPF] = xcell(e + kAccPF + 16, 64 + l2); qy[u % kAccPF] = ycell(e + kAccPF);
#pragma unroll
                for (int w = 0; w < kAccW; w++) m[w] = fma(((lds_vp)&ring[(u + 3 + w) & 15][lane + w])[0], y, m[w]);
                ring[(u + 2) & 15][lane] = xnew;                               // row e+16 takes the slot of row e
                if (lane < 14) ring[(u + 2) & 15][64 + lane] = xnew2;
                asm volatile("" : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]), "+v"(m[8]), "+v"(m[9]), "+v"(m[10]), "+v"(m[11]), "+v"(m[12]), "+v"(m[13]), "+v"(m[14]) :: "memory");
            }
        }
    }
    if (!live) return;
    // ---- E, I, the multiloop sums and H (vlin_acc_hsum) per width
    const double* __restrict__ gl = gaps + (size_t)(2 * sq) * 32 * ld;       // suffix sums over the gap length, [g][pos]
    const double* __restrict__ gr = gaps + (size_t)(2 * sq + 1) * 32 * ld;
    double lam_len = 1.0, mu_len = 1.0;
#pragma unroll
    for (int w = 0; w < kAccW; w++) {
        if (w >= max_w) break;
        lam_len *= L->lam; mu_len *= L->w_mu;
        const int b = a + w;
        double* __restrict__ up = B.up + ((size_t)sq * ld + (a - 1)) * max_w + w;
        if (b > n) { *up = 0.0; continue; }
        double acc = f5i[a - 1] * f5o[b] / Z * lam_len;                                          // E
        for (int p = a - 1; p >= 1 && p >= b - kMaxSingle; p--) acc += gl[(size_t)(b - p) * ld + p];
        for (int q = b + 1; q <= n && q <= a + kMaxSingle; q++) acc += gr[(size_t)(q - a) * ld + q];
        acc += m[w] * mu_len / Z;
        acc += *up;                                                                               // H (vlin_acc_hsum)
        *up = acc > 1.0 ? 1.0 : acc;
    }
}

#define RH_VINST(BS, CUT)                                                                                   \
    template __global__ void vlin_inside_diag<8, BS, CUT, 0>(McBatch, const VLinModel*, int, double, int);  \
    template __global__ void vlin_outside_diag<8, BS, CUT, 0>(McBatch, const VLinModel*, int, int, int*);
RH_VINST(16, false) RH_VINST(0, false) RH_VINST(16, true) RH_VINST(0, true)
#undef RH_VINST
template __global__ void vlin_inside_diag<8, 16, false, 1>(McBatch, const VLinModel*, int, double, int);   // look-ahead pairs
template __global__ void vlin_inside_diag<8, 16, false, 2>(McBatch, const VLinModel*, int, double, int);
template __global__ void vlin_inside_diag<8, 16, true, 1>(McBatch, const VLinModel*, int, double, int);
template __global__ void vlin_inside_diag<8, 16, true, 2>(McBatch, const VLinModel*, int, double, int);
template __global__ void vlin_outside_diag<8, 16, false, 1>(McBatch, const VLinModel*, int, int, int*);
template __global__ void vlin_outside_diag<8, 16, false, 2>(McBatch, const VLinModel*, int, int, int*);
template __global__ void vlin_outside_diag<8, 16, true, 1>(McBatch, const VLinModel*, int, int, int*);
template __global__ void vlin_outside_diag<8, 16, true, 2>(McBatch, const VLinModel*, int, int, int*);

}  // namespace rh
