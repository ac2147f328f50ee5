// mccaskill_lin.hip -- McCaskill inside / outside / posterior in SCALED LINEAR space.
//
// Same recurrences as mccaskill.hip (reference: /root/reference/src/contrafold/
// InferenceEngine.ipp:3356-3722 inside, 3731-4080 outside, 4498-4828 posterior), but
//   * values are Q * lambda^span (lin_model.h), so every inner term is one FMA;
//   * tables are stored DIAGONAL-MAJOR: cell (i, j=i+d) lives at [d*ld + i].  With one
//     THREAD per cell and 64 consecutive cells of a diagonal per wavefront, every
//     operand of every recurrence is a contiguous 512-byte row segment:
//        FM2[i,d]   = sum_m FM1[m][i]     * FM[d-m][i+m]
//        FMo[i,d]   = sum_e FM2o[d+e][i-e] * FM1[e][i-e]
//        FM1o[i,d]  = sum_e FM2o[d+e][i]   * FM[e][i+d]
//        FC gather  : FCX[d-2-t][i+1+l1],  FCo gather: FCoX[d+2+t][i-1-l1]
//     so no transposed mirrors, no cross-lane reductions and no 64-wide execution of
//     per-cell scalar work;
//   * the term loops of a 64-cell group are split over the W wavefronts of its
//     workgroup (partial sums meet in LDS) to keep thousands of waves in flight;
//   * loop shapes are wave-uniform, so their (l1,l2,weight) come from scalar loads.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lin_model.h"

#ifndef RH_WPE_IN
#define RH_WPE_IN
#endif
#ifndef RH_WPE_OUT
#define RH_WPE_OUT
#endif

namespace rh {

namespace {

constexpr uint32_t kPairMaskL = (1u << (0 * 5 + 3)) | (1u << (3 * 5 + 0)) | (1u << (1 * 5 + 2)) |
                                (1u << (2 * 5 + 1)) | (1u << (2 * 5 + 3)) | (1u << (3 * 5 + 2));
__device__ __forceinline__ bool pairs(int a, int b) { return (kPairMaskL >> (a * 5 + b)) & 1u; }
__device__ __forceinline__ size_t tri_off(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void block_map(int pin, int* sq, int* slot)
{
    *sq = pin ? blockIdx.x : blockIdx.y;
    *slot = pin ? blockIdx.y : blockIdx.x;
}

// (T+1)-tap filter over an LDS-resident row segment, fully unrolled: the weights are consecutive and
// wave-uniform (wide scalar loads), every tap is one ds_read + one FMA and there is no loop control on the
// scalar unit (a rolled loop costs ~5 SALU instructions per tap and the CU has ONE scalar ALU: measured
// 2.4 SALU per VALU instruction before unrolling).
template <int T>
__device__ __forceinline__ double filt_fwd(const double* __restrict__ wt, const double* seg)
{   // sum_l1 wt[l1] * seg[l1]
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int l1 = 0; l1 <= T; l1 += 2) {
        s0 = fma(wt[l1], seg[l1], s0);
        if (l1 + 1 <= T) s1 = fma(wt[l1 + 1], seg[l1 + 1], s1);
    }
    return s0 + s1;
}
template <int T>
__device__ __forceinline__ double filt_rev(const double* __restrict__ wt, const double* seg)
{   // sum_l1 wt[l1] * seg[T-l1]
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int l1 = 0; l1 <= T; l1 += 2) {
        s0 = fma(wt[l1], seg[T - l1], s0);
        if (l1 + 1 <= T) s1 = fma(wt[l1 + 1], seg[T - l1 - 1], s1);
    }
    return s0 + s1;
}
#define RH_T_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30)
__device__ __forceinline__ double filt_fwd_any(int t, const double* __restrict__ wt, const double* seg)
{
    switch (t) {
#define X(T) case T: return filt_fwd<T>(wt, seg);
        RH_T_CASES(X)
#undef X
    }
    return 0.0;
}
__device__ __forceinline__ double filt_rev_any(int t, const double* __restrict__ wt, const double* seg)
{
    switch (t) {
#define X(T) case T: return filt_rev<T>(wt, seg);
        RH_T_CASES(X)
#undef X
    }
    return 0.0;
}

}  // namespace

enum LinTable { L_FC = 0, L_FCX, L_FCA, L_FM1, L_FM, L_FCO, L_FCOX, L_FM2O, L_FMO, L_FM1O,
                L_FM2F, L_FMOF, L_FM1OF,  // far-block partial sums (mccaskill_far.hip)
                L_COUNT };
static_assert((int)L_COUNT <= (int)T_COUNT, "linear tables reuse the log-space table buffer");

// F5i~[0] = 1, F5o~[n] = 1
__global__ void lin_init(McBatch B, const LinModel* __restrict__ L, int* __restrict__ bad)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq >= B.ns) return;
    B.f5i[(size_t)sq * B.ld] = 1.0;
    B.f5o[(size_t)sq * B.ld + B.n[sq]] = 1.0;
    // F5o~[n-1] = F5o~[n] * ext_unpaired (no pair fits): lin_outside_pair needs it before its first launch
    if (B.n[sq] >= 2) B.f5o[(size_t)sq * B.ld + B.n[sq] - 1] = L->w_eu;
    bad[sq] = 0;
}

// ---------------------------------------------------------------------------------
// inside, diagonal d.  Workgroup = 64 consecutive cells x W wavefronts; the group after
// the last cell group computes F5i~[d+1] with its first wavefront.
// BS > 0: the k-terms that lie in complete blocks (I+2 .. J-2) come from FM2F (mccaskill_far.hip); only the
// <= 4*BS near terms are streamed here.  BS = 0: the whole sum is streamed.
// MODE 0: one diagonal per launch.  MODE 1 / 2 = look-ahead pair: almost every operand of diagonal d+1 is already final
// when diagonal d is computed (only the two FM2 terms that touch row d, and the epilogue, are not), and it is the SAME
// data diagonal d reads -- FM1 row m serves both, the FM rows are the same cache lines one column on, and the filter of
// length t+1 for d+1 runs over the staged row that the filter of length t for d runs over.  So the MODE 1 launch of an
// even diagonal d also accumulates the partial sums of diagonal d+1 from the operands it has in registers / LDS and
// leaves them in B.rowp; the MODE 2 launch of d+1 (one wavefront per group) adds the two fresh terms and runs the epilogue.
template <int R>
__device__ __forceinline__ void filt_pair(const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{   // sa = sum_{l<R} wA[l]*seg[l] (filter t = R-1 of diagonal d), sb = sum_{l<=R} wB[l]*seg[l] (filter t = R of diagonal d+1; none for R = 31)
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
    for (int l = 0; l <= R; l += 2) {
        const double x0 = seg[l];
        if (l < R) a0 = fma(wA[l], x0, a0);
        if (R <= kMaxSingle) b0 = fma(wB[l], x0, b0);
        if (l + 1 <= R) {
            const double x1 = seg[l + 1];
            if (l + 1 < R) a1 = fma(wA[l + 1], x1, a1);
            if (R <= kMaxSingle) b1 = fma(wB[l + 1], x1, b1);
        }
    }
    sa = a0 + a1;
    sb = b0 + b1;
}
__device__ __forceinline__ void filt_pair_any(int r, const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{
    switch (r) {
#define X(T) case T: filt_pair<T>(wA, wB, seg, sa, sb); return;
        RH_T_CASES(X) X(31)
#undef X
    }
    sa = 0.0; sb = 0.0;
}

// every staged row of wavefront WV of a W-wavefront group in the look-ahead kernels: rows r = g and 31-g, g = WV, WV+W, ...
// (<= 15), are compile-time constants here -- straight-line taps, immediate weight offsets, no per-row dispatch
template <int W, int WV, bool REV>
__device__ __forceinline__ void filt_pass_pair(const double* __restrict__ shape_w, const double* seg0, int rmax, double& accA, double& accB)
{
    if constexpr (WV < W) {
        constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W), LAST = kMaxSingle + 1, HALF = (kMaxSingle + 1) / 2;
#pragma unroll
        for (int q = 0; q < NSEG; q++) {
            const int g = WV + (q >> 1) * W;
            const int r = (q & 1) ? LAST - g : g;       // compile-time after unrolling
            if (g > HALF) continue;
            if (r <= rmax) {
                const double* __restrict__ wA = shape_w + (r > 0 ? (r - 1) * r / 2 : 0);
                const double* __restrict__ wB = shape_w + (r <= kMaxSingle ? r * (r + 1) / 2 : 0);
                const double* seg = seg0 + q * 96;
                double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
                for (int k = 0; k <= r; k++) {
                    const double x = REV ? seg[r - k] : seg[k];
                    if (k < r) { if (k & 1) a1 = fma(wA[k], x, a1); else a0 = fma(wA[k], x, a0); }
                    if (r <= kMaxSingle) { if (k & 1) b1 = fma(wB[k], x, b1); else b0 = fma(wB[k], x, b0); }
                }
                accA += a0 + a1;
                accB += b0 + b1;
            }
        }
    }
}

template <int W, int BS, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 64 : 64 * W) RH_WPE_IN void lin_inside_diag(McBatch B, const LinModel* __restrict__ L, int d, double lam_d, int pin)
{
    constexpr bool LA = MODE == 1 || MODE == 3;                 // accumulates the look-ahead sums of diagonal d+1
    constexpr int GS = MODE == 3 ? 63 : 64;                     // columns per group (MODE 3: one column of overlap, see below)
    constexpr int WR = MODE == 2 ? 1 : W;                       // wavefronts per group
    constexpr int NQ = LA ? 4 : 2;                              // per-wavefront partial sums
    constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);       // rows g = 0..15 and LAST-g, dealt round-robin to the wavefronts
    __shared__ double part[MODE == 2 ? 1 : NQ][WR][64];
    __shared__ double gbuf[MODE == 2 ? 1 : W][MODE == 2 ? 1 : NSEG][MODE == 2 ? 1 : 96];
    int sq, slot;
    block_map(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    if (d > (MODE == 3 ? n : n - 1)) return;
    const int ncell = n - 1 - d > 0 ? n - 1 - d : 0;
    const int ngroup = (ncell + GS - 1) / GS;
    if (slot > ngroup) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // w is wave-uniform: scalar loads/addressing
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    double* __restrict__ f5i = B.f5i + sq * ld;
    double* __restrict__ rowp = B.rowp + (size_t)sq * 2 * ld;   // look-ahead partial sums of the next diagonal: FM2, filters

    if (slot == ngroup) {
        // F5i[jj] = F5i[jj-1]*ext_unpaired + sum_{k<=jj-2} F5i[k]*FCA[k+1,jj-1]*ext_paired   (ipp:3692-3717)
        // all wavefronts: the column of FCA is an anti-diagonal of the diagonal-major table (one line per term)
        // MODE 3: F5i[d] and F5i[d+1] (rows <= d-1 of FCA: final before this launch); otherwise F5i[d+1]
        const double* __restrict__ fca = tab + L_FCA * ts;
#pragma unroll 1
        for (int jj = MODE == 3 ? d : d + 1; jj <= d + 1; jj++) {
            if (jj < 1 || jj > n) continue;
            double acc = 0.0;
            for (int k = threadIdx.x; k <= jj - 2; k += 64 * WR) acc = fma(f5i[k], fca[(jj - 2 - k) * ld + (k + 1)], acc);
            acc = wsum(acc);
            if (lane == 0) part[0][w][0] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < WR; k++) t += part[0][k][0];
                f5i[jj] = f5i[jj - 1] * L->w_eu + t * L->w_ep2;
            }
            __syncthreads();   // F5i[jj] is an operand of F5i[jj+1]
        }
        return;
    }

    // MODE 3: a group's 64 lanes hold 64 cells of diagonal d but only 63 of d+1 (lane l of d+1 needs lanes l and l+1 of d),
    // so groups advance by 63 columns and the last column of a group's diagonal d is computed again, bit for bit, as the
    // first of the next group's
    const int i = 1 + slot * GS + lane, j = i + d;
    const bool valid = i <= ncell;
    int s_im1 = 4, s_i = 4, s_ip1 = 4, s_j = 4, s_jp1 = 4, s_jp2 = 4;
    if (valid) { s_im1 = s[i - 1]; s_i = s[i]; s_ip1 = s[i + 1]; s_j = s[j]; s_jp1 = s[j + 1]; s_jp2 = s[j + 2]; }
    const bool pairable = valid && pairs(s_i, s_jp1);

    // epilogue operands (wave 0 only): issued now so that their latency hides behind the term loops
    const size_t at = d * ld + i;
    const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1 + s_j;       // (i,j)   as enclosing pair
    const int idd = 25 * (5 * s_jp1 + s_jp2) + 5 * s_i + s_im1;     // (j+1,i-1) as enclosed pair
    double e_tjb = 0, e_tja = 0, e_tst = 0, e_bp = 0, e_tjbd = 0, e_tjad = 0, e_n01 = 0, e_n10 = 0, e_n11 = 0;
    double o_x01 = 0, o_x10 = 0, o_x11 = 0, o_fc = 0, o_fca = 0, o_fm1 = 0, o_fm = 0;
    const int d1 = d + 1;
    const bool valid1 = LA && i <= n - 1 - d1 && (MODE != 3 || lane < 63);   // cell (i, j+1) of diagonal d+1
    if (MODE != 3 && w == 0 && valid) {   // (MODE 3 loads them after the term loops: two operand sets, see its epilogue)
        e_tjb = L->TJB[idx]; e_tja = L->TJA[idx]; e_tst = L->TST[idx]; e_bp = L->E_bp[s_i * 5 + s_jp1];
        e_tjbd = L->TJB[idd]; e_tjad = L->TJA[idd];
        e_n01 = L->E_b01[s_j]; e_n10 = L->E_b10[s_ip1]; e_n11 = L->E_11[s_ip1 * 5 + s_j];
        if (d >= 3) {
            const double* __restrict__ fcx = tab + L_FCX * ts;
            o_x01 = fcx[(d - 3) * ld + i + 1];
            o_x10 = fcx[(d - 3) * ld + i + 2];
            if (d >= 4) o_x11 = fcx[(d - 4) * ld + i + 2];
        }
        if (d >= 2) {
            o_fc = tab[L_FC * ts + (d - 2) * ld + i + 1];
            o_fca = tab[L_FCA * ts + (d - 2) * ld + i + 1];
            o_fm1 = tab[L_FM1 * ts + (d - 1) * ld + i + 1];
            o_fm = tab[L_FM * ts + (d - 1) * ld + i];
        }
    }

    // near set of a cell (i, j) in k = i+m: k < kA or k >= kB (everything when the tile has no far blocks)
    int kA = 1 << 30, kB = 0;
    if (BS > 0) {
        const int I = i / BS, J = j / BS;
        if (J - I >= 4) { kA = (I + 2) * BS; kB = (J - 1) * BS; }
    }
    int kB1 = 0;                                  // the same for cell (i, j+1) of diagonal d+1 (kA1 = kA where far blocks exist)
    if (LA && BS > 0) {
        const int I = i / BS, J1 = (j + 1) / BS;
        if (J1 - I >= 4) kB1 = (J1 - 1) * BS;
    }
    const int kA1 = kB1 > 0 ? (i / (BS > 0 ? BS : 1) + 2) * BS : 1 << 30;
    double acc2 = 0.0, accc = 0.0;
    if constexpr (MODE == 2) {
        // the look-ahead sums of the previous launch + the two terms that touch row d-1:  m = 1: FM1[1][i]*FM[d-1][i+1],
        // m = d-1: FM1[d-1][i]*FM[1][i+d-1]
        if (valid) {
            acc2 = rowp[i];
            accc = pairable ? rowp[ld + i] : 0.0;
            if (d >= 2) {
                const double* __restrict__ fm1c = tab + L_FM1 * ts + i;
                const double* __restrict__ fmc = tab + L_FM * ts + i;
                const int k1 = i + 1, k2 = i + d - 1;
                if (k1 < kA || k1 >= kB) acc2 = fma(fm1c[ld], fmc[(d - 1) * ld + 1], acc2);
                if (d >= 3 && (k2 < kA || k2 >= kB)) acc2 = fma(fm1c[(d - 1) * ld], fmc[ld + d - 1], acc2);
            }
            if (BS > 0 && kB > 0) acc2 += tab[L_FM2F * ts + d * ld + i];
        }
    } else {
    // ---- FM2[i,d] = sum_{m=1}^{d-1} FM1[m][i] * FM[d-m][i+m]          (ipp:3384-3411)
    double acc2n = 0.0, acccn = 0.0;              // look-ahead: the same sums of diagonal d+1 without the terms that touch row d
    {
        const int ic = valid ? i : (ncell > 0 ? ncell : 1);   // invalid lanes read the last valid cell's operands
        const double* __restrict__ fm1c = tab + L_FM1 * ts + ic;
        const double* __restrict__ fmc = tab + L_FM * ts + ic;
        constexpr int UF = 8;  // 2*UF (3*UF) row segments (512 B each) in flight per wavefront
        // uniform m-ranges that cover every lane's near set: [1, d-1], or its two ends when far blocks exist
        const bool split = BS > 0 && d - 1 > 4 * BS;
        const int lo0 = 1, hi0 = split ? 2 * BS : d - 1;
        const int lo1 = split ? d - 2 * BS : 1, hi1 = split ? d - 1 : 0;
#pragma unroll
        for (int part_i = 0; part_i < 2; part_i++) {
            const int lo = part_i ? lo1 : lo0, hi = part_i ? hi1 : hi0;
            for (int m = lo + w; m <= hi; m += UF * W) {
                // branch-free: every load is issued (a lane or term that is out of range reads a clamped, valid address)
                // so that all loads are in flight together; the product is masked afterwards
                double a[UF], b[UF], bn[UF];
#pragma unroll
                for (int u = 0; u < UF; u++) {
                    const int mm = m + u * W, mc = mm <= hi ? mm : hi;
                    a[u] = fm1c[mc * ld];
                    b[u] = fmc[(d - mc) * ld + mc];
                    if (LA) bn[u] = fmc[(d1 - mc) * ld + mc];   // FM[d+1-m][i+m]: row d-(m-1), final for m >= 2
                }
#pragma unroll
                for (int u = 0; u < UF; u++) {
                    const int mm = m + u * W, k = i + mm;
                    const bool ok = valid && mm <= hi && (k < kA || k >= kB);
                    acc2 = fma(ok ? a[u] : 0.0, ok ? b[u] : 0.0, acc2);   // both operands: a masked term reads a clamped address, and 0 x (a stale Inf) is NaN
                    if (LA) {
                        const bool ok1 = valid1 && mm <= hi && mm >= 2 && (k < kA1 || k >= kB1);
                        acc2n = fma(ok1 ? a[u] : 0.0, ok1 ? bn[u] : 0.0, acc2n);
                    }
                }
            }
        }
        if (BS > 0 && valid && kB > 0) acc2 += (w == 0) ? tab[L_FM2F * ts + d * ld + i] : 0.0;
    }

    // ---- generic single-branch shapes of FC[i,d]: sum_t sum_l1 w(l1,t-l1) * FCX[d-2-t][i+1+l1]   (ipp:3597-3619)
    // For one t the 64 cells of the group read overlapping windows of ONE row: the segment is staged in LDS
    // once and each lane runs a (t+1)-tap filter over it; shape weights are wave-uniform (scalar loads).
    // t and 30-t go to the same wavefront so that every wavefront filters ~62 taps.
    // MODE 1: staged row r = 0..31 is table row d-1-r; it carries filter t = r-1 of diagonal d and filter t = r of d+1.
    if (LA ? d >= 1 : d >= 2) {
        const int tmax = d - 2 < kMaxSingle ? d - 2 : kMaxSingle;
        const int i0 = 1 + slot * GS;
        const double* __restrict__ fcx = tab + L_FCX * ts;
        // pass 1: stage every segment this wavefront filters.  Straight-line code: all row loads are issued back to
        // back (clamped addresses, no branches -- a branch per load makes the compiler wait for each load in turn), then
        // written to LDS
        double r0[NSEG], r1[NSEG];
        const int col0 = i0 + 1;                                  // lane k of a segment = column col0+k of the row
        const int c0 = col0 + lane < ld ? col0 + lane : ld - 1, c1 = col0 + 64 + (lane & 31) < ld ? col0 + 64 + (lane & 31) : ld - 1;
        constexpr int HALF = LA ? (kMaxSingle + 1) / 2 : kMaxSingle / 2;   // rows g and LAST-g share a wavefront
        constexpr int LAST = LA ? kMaxSingle + 1 : kMaxSingle;
        const int rmax = LA ? (d - 1 < LAST ? d - 1 : LAST) : tmax;        // last staged row index that exists
#pragma unroll
        for (int q = 0; q < NSEG; q++) {
            const int g = w + (q >> 1) * W;
            const int t = (q & 1) ? LAST - g : g;
            const bool on = g <= HALF && !((q & 1) && t == g) && t <= rmax;   // wave-uniform
            const int srow = LA ? d - 1 - t : d - 2 - t;
            const double* __restrict__ row = fcx + (on ? srow : 0) * ld;
            r0[q] = row[c0];
            r1[q] = row[c1];
        }
#pragma unroll
        for (int q = 0; q < NSEG; q++) {
            gbuf[w][q][lane] = col0 + lane < ld ? r0[q] : 0.0;
            if (lane < 32) gbuf[w][q][64 + lane] = col0 + 64 + lane < ld ? r1[q] : 0.0;
        }
        // pass 2: the filters
        // rolled: ONE copy of the filter switch (unrolled, the NSEG copies made the kernel larger than the
        // instruction cache two CUs share)
        if constexpr (LA) {
            switch (w) {
#define X(V) case V: filt_pass_pair<W, V, false>(L->shape_w, &gbuf[w][0][lane], rmax, accc, acccn); break;
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#undef X
            }
        } else {
#pragma unroll 1
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int t = (q & 1) ? LAST - g : g;
                const bool on = g <= HALF && !((q & 1) && t == g) && t <= rmax;
                if (on) accc += filt_fwd_any(t, L->shape_w + t * (t + 1) / 2, &gbuf[w][q][lane]);
            }
        }
        if (!pairable) accc = 0.0;
    }
    if constexpr (LA) { part[2][w][lane] = acc2n; part[3][w][lane] = acccn; }
    }

    if constexpr (MODE != 2) {
        part[0][w][lane] = acc2;
        part[1][w][lane] = accc;
        __syncthreads();
        if constexpr (MODE == 1) {
            if (w == 1 && i < ld) {   // the look-ahead sums of diagonal d+1 (any value where d+1 has no cell: never read)
                double p2 = 0.0, pg = 0.0;
#pragma unroll
                for (int k = 0; k < W; k++) { p2 += part[2][k][lane]; pg += part[3][k][lane]; }
                rowp[i] = p2;
                rowp[ld + i] = pg;
            }
        }
    }
    if constexpr (MODE == 3) {
        // wavefront 0 finishes diagonal d and hands FM / FM1 of its row to wavefront 1 through LDS; wavefront 1 meanwhile loads
        // the epilogue operands of diagonal d+1 (cell (i, j+1)), whose two FM2 terms that touch row d need lane l and l+1 of
        // it.  (Two wavefronts: each holds one operand set -- in one wavefront the two sets cost ~50 VGPRs and a wavefront of
        // occupancy; operands are loaded after the term loops for the same reason.)
        if (w > 1) return;
        double (*hand)[64] = part[0];   // FM / FM1 of row d: over wavefront 0's own look-behind sums, which only it reads (keeps the group at 32 KB: 5 per CU)
        double n_tjb = 0, n_tja = 0, n_tst = 0, n_bp = 0, n_tjbd = 0, n_tjad = 0, n_n01 = 0, n_n10 = 0, n_n11 = 0;
        double n_x01 = 0, n_x10 = 0, n_x11 = 0, n_fc = 0, n_fca = 0, n_a1 = 0, n_bd = 0, n_far = 0;
        double fm2n = 0.0, gn = 0.0;
        bool pairable1 = false;
        if (w == 0) {
            if (valid) {
                e_tjb = L->TJB[idx]; e_tja = L->TJA[idx]; e_tst = L->TST[idx]; e_bp = L->E_bp[s_i * 5 + s_jp1];
                e_tjbd = L->TJB[idd]; e_tjad = L->TJA[idd];
                e_n01 = L->E_b01[s_j]; e_n10 = L->E_b10[s_ip1]; e_n11 = L->E_11[s_ip1 * 5 + s_j];
                if (d >= 3) {
                    const double* __restrict__ fcx = tab + L_FCX * ts;
                    o_x01 = fcx[(d - 3) * ld + i + 1];
                    o_x10 = fcx[(d - 3) * ld + i + 2];
                    if (d >= 4) o_x11 = fcx[(d - 4) * ld + i + 2];
                }
                if (d >= 2) {
                    o_fc = tab[L_FC * ts + (d - 2) * ld + i + 1];
                    o_fca = tab[L_FCA * ts + (d - 2) * ld + i + 1];
                    o_fm1 = tab[L_FM1 * ts + (d - 1) * ld + i + 1];
                    o_fm = tab[L_FM * ts + (d - 1) * ld + i];
                }
            }
            double fm2 = 0.0, g = 0.0;
#pragma unroll
            for (int k = 0; k < W; k++) { fm2 += part[0][k][lane]; g += part[1][k][lane]; }
            double fc = 0.0;
            if (pairable) {
                const double sp = L->w01 * e_n01 * o_x01 + L->w10 * e_n10 * o_x10 + L->w11 * e_n11 * o_x11;
                const double hp = d >= 3 ? lam_d * L->E_hairpin[d < 30 ? d : 30] : 0.0;   // ScoreHairpin (ipp:2123-2152)
                const double st = o_fc * L->lam2 * e_tst;                                  // stacking pair (ipp:3595)
                fc = e_tjb * (g + sp + hp) + st + fm2 * e_tja * L->e_mpmb;                 // ipp:3573-3622
            }
            double fm1v = 0.0, fmv = 0.0;
            if (valid && d >= 2) {                                                         // ipp:3641-3688
                fm1v = o_fca * L->w_mp2 + o_fm1 * L->w_mu;
                fmv = fm2 + o_fm * L->w_mu + fm1v;
            }
            if (valid) {
                tab[L_FC * ts + at] = fc;
                tab[L_FCX * ts + at] = fc * e_bp * e_tjbd;
                tab[L_FCA * ts + at] = fc * e_bp * e_tjad;
                tab[L_FM1 * ts + at] = fm1v;
                tab[L_FM * ts + at] = fmv;
            }
            hand[0][lane] = fmv; hand[1][lane] = fm1v;
        } else {
            if (valid1) {
                const int s_jp3 = s[j + 3];
                pairable1 = pairs(s_i, s_jp2);
                const int ix = 25 * (5 * s_i + s_ip1) + 5 * s_jp2 + s_jp1, id = 25 * (5 * s_jp2 + s_jp3) + 5 * s_i + s_im1;
                n_tjb = L->TJB[ix]; n_tja = L->TJA[ix]; n_tst = L->TST[ix]; n_bp = L->E_bp[s_i * 5 + s_jp2];
                n_tjbd = L->TJB[id]; n_tjad = L->TJA[id];
                n_n01 = L->E_b01[s_jp1]; n_n10 = L->E_b10[s_ip1]; n_n11 = L->E_11[s_ip1 * 5 + s_jp1];
                if (d1 >= 3) {
                    const double* __restrict__ fcx = tab + L_FCX * ts;
                    n_x01 = fcx[(d1 - 3) * ld + i + 1];
                    n_x10 = fcx[(d1 - 3) * ld + i + 2];
                    if (d1 >= 4) n_x11 = fcx[(d1 - 4) * ld + i + 2];
                }
                if (d1 >= 2) {
                    n_fc = tab[L_FC * ts + (d1 - 2) * ld + i + 1];
                    n_fca = tab[L_FCA * ts + (d1 - 2) * ld + i + 1];
                    n_a1 = tab[L_FM1 * ts + ld + i];            // FM1[1][i]   (term m = 1 of FM2[i,d+1]; its partner is FM[d][i+1])
                    n_bd = tab[L_FM * ts + ld + i + d];         // FM[1][i+d]  (term m = d; its partner is FM1[d][i])
                }
                if (BS > 0 && kB1 > 0) n_far = tab[L_FM2F * ts + d1 * ld + i];
            }
#pragma unroll
            for (int k = 0; k < W; k++) { fm2n += part[2][k][lane]; gn += part[3][k][lane]; }
        }
        __syncthreads();   // wavefronts 0 and 1 (the others have left)
        if (w == 0 || !valid1) return;
        const double fmv = hand[0][lane], fm1v = hand[1][lane];                    // FM[d][i],   FM1[d][i]
        const double fm_right = hand[0][lane + 1], fm1_right = hand[1][lane + 1];  // FM[d][i+1], FM1[d][i+1]   (lane <= 62)
        if (d1 >= 2) {
            const int k1 = i + 1, k2 = i + d;
            if (k1 < kA1 || k1 >= kB1) fm2n = fma(n_a1, fm_right, fm2n);                 // m = 1
            if (d1 >= 3 && (k2 < kA1 || k2 >= kB1)) fm2n = fma(fm1v, n_bd, fm2n);          // m = d
        }
        fm2n += n_far;
        double fcn = 0.0;
        if (pairable1) {
            const double sp = L->w01 * n_n01 * n_x01 + L->w10 * n_n10 * n_x10 + L->w11 * n_n11 * n_x11;
            const double hp = d1 >= 3 ? lam_d * L->lam * L->E_hairpin[d1 < 30 ? d1 : 30] : 0.0;
            const double st = n_fc * L->lam2 * n_tst;
            fcn = n_tjb * (gn + sp + hp) + st + fm2n * n_tja * L->e_mpmb;
        }
        double fm1n = 0.0, fmn = 0.0;
        if (d1 >= 2) {
            fm1n = n_fca * L->w_mp2 + fm1_right * L->w_mu;
            fmn = fm2n + fmv * L->w_mu + fm1n;
        }
        const size_t at1 = d1 * ld + i;
        tab[L_FC * ts + at1] = fcn;
        tab[L_FCX * ts + at1] = fcn * n_bp * n_tjbd;
        tab[L_FCA * ts + at1] = fcn * n_bp * n_tjad;
        tab[L_FM1 * ts + at1] = fm1n;
        tab[L_FM * ts + at1] = fmn;
    } else {
    if (w != 0 || !valid) return;
    double fm2 = acc2, g = accc;
    if constexpr (MODE != 2) {
        fm2 = 0.0; g = 0.0;
#pragma unroll
        for (int k = 0; k < W; k++) { fm2 += part[0][k][lane]; g += part[1][k][lane]; }
    }

    double fc = 0.0;
    if (pairable) {
        const double sp = L->w01 * e_n01 * o_x01 + L->w10 * e_n10 * o_x10 + L->w11 * e_n11 * o_x11;
        const double hp = d >= 3 ? lam_d * L->E_hairpin[d < 30 ? d : 30] : 0.0;   // ScoreHairpin (ipp:2123-2152)
        const double st = o_fc * L->lam2 * e_tst;                                  // stacking pair (ipp:3595)
        fc = e_tjb * (g + sp + hp) + st + fm2 * e_tja * L->e_mpmb;                 // ipp:3573-3622
    }
    double fm1v = 0.0, fmv = 0.0;
    if (d >= 2) {                                                                  // ipp:3641-3688
        fm1v = o_fca * L->w_mp2 + o_fm1 * L->w_mu;
        fmv = fm2 + o_fm * L->w_mu + fm1v;
    }
    tab[L_FC * ts + at] = fc;
    tab[L_FCX * ts + at] = fc * e_bp * e_tjbd;
    tab[L_FCA * ts + at] = fc * e_bp * e_tjad;
    tab[L_FM1 * ts + at] = fm1v;
    tab[L_FM * ts + at] = fmv;
    }
}

// ---------------------------------------------------------------------------------
// outside (pull form) + posterior, diagonal d; last group: F5o~[d+1].
template <int W, int BS>
__global__ __launch_bounds__(64 * W) RH_WPE_OUT void lin_outside_diag(McBatch B, const LinModel* __restrict__ L, int d, int pin, int* __restrict__ bad)
{
    __shared__ double part[3][W][64];
    __shared__ double gbuf[W][2 * ((kMaxSingle / 2 + W) / W)][96];
    int sq, slot;
    block_map(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int ncell = n - 1 - d;
    if (ncell < 1) return;
    const int ngroup = (ncell + 63) >> 6;
    if (slot > ngroup) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // w is wave-uniform: scalar loads/addressing
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ f5i = B.f5i + sq * ld;
    double* __restrict__ f5o = B.f5o + sq * ld;

    if (slot == ngroup) {
        // F5o[k] = F5o[k+1]*ext_unpaired + sum_{jj>=k+2} F5o[jj]*FCA[k+1,jj-1]*ext_paired   (ipp:3751-3780, pulled)
        const int k = d + 1;
        const double* __restrict__ fca = tab + L_FCA * ts + (k + 1);
        double acc = 0.0;
        for (int jj = k + 2 + threadIdx.x; jj <= n; jj += 64 * W) acc = fma(f5o[jj], fca[(jj - 2 - k) * ld], acc);
        acc = wsum(acc);
        if (lane == 0) part[0][w][0] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < W; q++) t += part[0][q][0];
            f5o[k] = f5o[k + 1] * L->w_eu + t * L->w_ep2;
        }
        return;
    }

    const int i0 = 1 + slot * 64;
    const int i = i0 + lane, j = i + d;
    const bool valid = i <= ncell;
    int s_im1 = 4, s_i = 4, s_ip1 = 4, s_j = 4, s_jp1 = 4, s_jp2 = 4;
    if (valid) { s_im1 = s[i - 1]; s_i = s[i]; s_ip1 = s[i + 1]; s_j = s[j]; s_jp1 = s[j + 1]; s_jp2 = s[j + 2]; }
    const bool pairable = valid && pairs(s_i, s_jp1);
    const bool guard_m = d >= 2;

    // epilogue operands (wave 0 only), issued ahead of the term loops
    const size_t at = d * ld + i;
    const bool up_ok = i - 1 >= 1 && j + 1 <= n - 1;  // the cell (i-1, j+1) is interior
    const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1 + s_j;
    const int idd = 25 * (5 * s_jp1 + s_jp2) + 5 * s_i + s_im1;
    double e_tjb = 0, e_tja = 0, e_tst = 0, e_bp = 0, e_tjbd = 0, e_tjad = 0, e_n01 = 0, e_n10 = 0, e_n11 = 0;
    double o_fmo = 0, o_fm1o = 0, o_f5o = 0, o_f5i = 0, o_fm1o_up = 0, o_fco_up = 0, o_x01 = 0, o_x10 = 0, o_x11 = 0, o_fc = 0, o_z = 1;
    if (w == 0 && valid) {
        e_tjb = L->TJB[idx]; e_tja = L->TJA[idx]; e_bp = L->E_bp[s_i * 5 + s_jp1];
        e_tjbd = L->TJB[idd]; e_tjad = L->TJA[idd];
        e_tst = L->TST[25 * (5 * s_im1 + s_i) + 5 * s_jp2 + s_jp1];
        e_n01 = L->E_b01[s_jp2]; e_n10 = L->E_b10[s_im1]; e_n11 = L->E_11[s_im1 * 5 + s_jp2];
        if (guard_m) {
            if (j + 1 <= n - 1) o_fmo = tab[L_FMO * ts + (d + 1) * ld + i];            // ipp:3806
            if (i - 1 >= 1) o_fm1o = tab[L_FM1O * ts + (d + 1) * ld + i - 1];            // ipp:3833
        }
        o_f5o = f5o[j + 1]; o_f5i = f5i[i - 1]; o_z = f5i[n];
        o_fc = tab[L_FC * ts + at];
        const double* __restrict__ fcox = tab + L_FCOX * ts;
        if (up_ok) {
            o_fm1o_up = tab[L_FM1O * ts + (d + 2) * ld + i - 1];                         // ipp:3828
            o_fco_up = tab[L_FCO * ts + (d + 2) * ld + i - 1];
        }
        if (i - 1 >= 1 && j + 2 <= n - 1) o_x01 = fcox[(d + 3) * ld + i - 1];
        if (i - 2 >= 1 && j + 1 <= n - 1) o_x10 = fcox[(d + 3) * ld + i - 2];
        if (i - 2 >= 1 && j + 2 <= n - 1) o_x11 = fcox[(d + 4) * ld + i - 2];
    }

    double accm = 0.0, acc1 = 0.0, accc = 0.0;
    if (guard_m) {
        constexpr int UO = 6;
        // FMo[i,d] += FM2o[d+e][i-e] * FM1[e][i-e], e = 1..i-1            (ipp:4046-4064, pulled)
        // with blocks: only i' = i-e in blocks I-1, I are streamed; blocks <= I-2 come from FMOF
        {
            const int i_last = ncell < i0 + 63 ? ncell : i0 + 63;
            const int emax = BS > 0 ? (i_last - 1 < 2 * BS ? i_last - 1 : 2 * BS) : i_last - 1;
            const double* __restrict__ x = tab + L_FM2O * ts + i;
            const double* __restrict__ y = tab + L_FM1 * ts + i;
            int mine = valid ? i - 1 : 0;
            if (BS > 0 && valid) { const int lim = i - (i / BS - 1) * BS; mine = mine < lim ? mine : lim; }
            for (int e = 1 + w; e <= emax; e += UO * W) {
                double xv[UO], yv[UO];
#pragma unroll
                for (int u = 0; u < UO; u++) {
                    const int ee = e + u * W;
                    const bool ok = ee <= mine;
                    xv[u] = ok ? x[(d + ee) * ld - ee] : 0.0;
                    yv[u] = ok ? y[ee * ld - ee] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UO; u++) accm = fma(xv[u], yv[u], accm);
            }
            if (BS > 0 && valid && w == 0) accm += tab[L_FMOF * ts + d * ld + i];
        }
        // FM1o[i,d] += FM2o[d+e][i] * FM[e][i+d], e = 1..n-1-j; blocks >= J+2 come from FM1OF
        {
            const int emax_all = n - 1 - (i0 + d);
            const int emax = BS > 0 ? (emax_all < 2 * BS ? emax_all : 2 * BS) : emax_all;
            const double* __restrict__ x = tab + L_FM2O * ts + i;
            const double* __restrict__ y = tab + L_FM * ts + j;
            int mine = valid ? n - 1 - j : 0;
            if (BS > 0 && valid) { const int lim = (j / BS + 2) * BS - 1 - j; mine = mine < lim ? mine : lim; }
            for (int e = 1 + w; e <= emax; e += UO * W) {
                double xv[UO], yv[UO];
#pragma unroll
                for (int u = 0; u < UO; u++) {
                    const int ee = e + u * W;
                    const bool ok = ee <= mine;
                    xv[u] = ok ? x[(d + ee) * ld] : 0.0;
                    yv[u] = ok ? y[ee * ld] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UO; u++) acc1 = fma(xv[u], yv[u], acc1);
            }
            if (BS > 0 && valid && w == 0) acc1 += tab[L_FM1OF * ts + d * ld + i];
        }
    }
    {   // enclosing single-branch loops: sum_t sum_l1 w(l1,t-l1) * FCoX[d+2+t][i-1-l1]       (ipp:4004-4024, pulled)
        // same LDS-staged filter as the inside sweep; a tap is valid when its enclosing pair (i-1-l1, j+1+l2)
        // is interior, i.e. when its column lies in [1, n-1-row] of the source row: the segment is zero-filled
        // outside that range while it is staged, so the taps themselves need no mask
        const int room = n - 4 - d;  // source span d+2+t <= n-2
        if (room >= 0) {
            const int tmax = room < kMaxSingle ? room : kMaxSingle;
            const double* __restrict__ fcox = tab + L_FCOX * ts;
            constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int t = (q & 1) ? kMaxSingle - g : g;
                const bool on = g <= kMaxSingle / 2 && !((q & 1) && t == g) && t <= tmax;   // wave-uniform
                if (on) {
                    const int col0 = i0 - 1 - t;                  // segment column of lane k: col0+k; window of cell i: [i-1-t, i-1]
                    const double* __restrict__ row = fcox + (d + 2 + t) * ld;
                    const int cmax = n - 1 - (d + 2 + t);         // last interior column of that row
                    const int c = col0 + lane;
                    gbuf[w][q][lane] = (c >= 1 && c <= cmax) ? row[c] : 0.0;
                    const int c2 = col0 + 64 + lane;
                    if (lane < 32) gbuf[w][q][64 + lane] = (c2 >= 1 && c2 <= cmax) ? row[c2] : 0.0;
                }
            }
#pragma unroll 1
            for (int q = 0; q < NSEG; q++) {   // rolled: one copy of the filter switch (see the inside kernel)
                const int g = w + (q >> 1) * W;
                const int t = (q & 1) ? kMaxSingle - g : g;
                const bool on = g <= kMaxSingle / 2 && !((q & 1) && t == g) && t <= tmax;
                if (on) accc += filt_rev_any(t, L->shape_w + t * (t + 1) / 2, &gbuf[w][q][lane]);
            }
            if (!pairable) accc = 0.0;
        }
    }
    part[0][w][lane] = accm;
    part[1][w][lane] = acc1;
    part[2][w][lane] = accc;
    __syncthreads();
    if (w != 0 || !valid) return;
    double sm = 0.0, s1 = 0.0, g = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) { sm += part[0][k][lane]; s1 += part[1][k][lane]; g += part[2][k][lane]; }

    double fmo = 0.0, fm1o = 0.0;
    if (guard_m) {
        fmo = sm + o_fmo * L->w_mu;                   // ipp:3806
        fm1o = s1 + fmo + o_fm1o * L->w_mu;           // ipp:3809, 3833
    }
    double fco = 0.0;
    if (pairable) {
        const double ext = o_f5o * o_f5i * L->w_ep2;  // exterior loop, ipp:3768-3776
        const double multi = o_fm1o_up * L->w_mp2;    // branch of a multiloop, ipp:3828
        const double sp = L->w01 * e_n01 * o_x01 + L->w10 * e_n10 * o_x10 + L->w11 * e_n11 * o_x11;
        const double st = o_fco_up * L->lam2 * e_tst; // stacked on (i-1,j+1)
        fco = e_bp * (e_tjad * (ext + multi) + e_tjbd * (g + sp)) + st;
    }
    const double fm2o = fmo + fco * e_tja * L->e_mpmb;                                                // ipp:3803, 4027
    tab[L_FCO * ts + at] = fco;
    tab[L_FCOX * ts + at] = fco * e_tjb;
    tab[L_FMO * ts + at] = fmo;
    tab[L_FM1O * ts + at] = fm1o;
    tab[L_FM2O * ts + at] = fm2o;
    // posterior of pair (i, j+1) = FCo * FCi / Z, clipped to [0,1]                 (ipp:4689-4827)
    double p = fco * o_fc / o_z;
    if (!(p == p) || p > 1e300) { atomicOr(&bad[sq], 1); p = 0.0; }
    p = p > 1.0 ? 1.0 : (p < 0.0 ? 0.0 : p);
    B.bp[(size_t)sq * B.tri_stride + tri_off(n, i) + (j + 1)] = p;
}

// ---------------------------------------------------------------------------------
// outside, diagonals d and d-1 in one launch (the counterpart of lin_inside_diag MODE 3).  Cell (i, j-1) of diagonal d-1
// reads row d only through the two e = 1 terms of its multibranch sums and two epilogue operands, at columns i and i-1:
// everything else is final before the launch and is the data diagonal d reads (FM1 row e serves both near sums, the FM2o /
// FM rows are the same cache lines one column on, the filter of length t+1 runs over the staged row of the filter of
// length t).  All wavefronts accumulate both diagonals' sums; wavefront 0 finishes d, then d-1 with lane l-1's row-d values
// by shuffle.  Lane 0 has no left neighbour: groups advance by 63 columns and lane 0 finishes only diagonal d (except in
// group 0, whose column 0 does not exist); the overlapping column of diagonal d is computed twice, bit for bit.
// Last group: F5o~[k], k = khi (= d) .. d-1: diagonal d-1 needs F5o[>= d+1], which the PREVIOUS launch has to leave behind
// (lin_init before the first one).
template <int R>
__device__ __forceinline__ void filt_pair_rev(const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{   // x_k = seg[R-k];  sa = sum_{k<R} wA[k]*x_k (filter t = R-1 of diagonal d), sb = sum_{k<=R} wB[k]*x_k (filter t = R of d-1; none for R = 31)
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
#pragma unroll
    for (int k = 0; k <= R; k += 2) {
        const double x0 = seg[R - k];
        if (k < R) a0 = fma(wA[k], x0, a0);
        if (R <= kMaxSingle) b0 = fma(wB[k], x0, b0);
        if (k + 1 <= R) {
            const double x1 = seg[R - k - 1];
            if (k + 1 < R) a1 = fma(wA[k + 1], x1, a1);
            if (R <= kMaxSingle) b1 = fma(wB[k + 1], x1, b1);
        }
    }
    sa = a0 + a1;
    sb = b0 + b1;
}
__device__ __forceinline__ void filt_pair_rev_any(int r, const double* __restrict__ wA, const double* __restrict__ wB, const double* seg, double& sa, double& sb)
{
    switch (r) {
#define X(T) case T: filt_pair_rev<T>(wA, wB, seg, sa, sb); return;
        RH_T_CASES(X) X(31)
#undef X
    }
    sa = 0.0; sb = 0.0;
}

template <int W, int BS>
__global__ __launch_bounds__(64 * W) RH_WPE_OUT void lin_outside_pair(McBatch B, const LinModel* __restrict__ L, int d, int khi, int pin, int* __restrict__ bad)
{
    constexpr int NSEG = 2 * ((kMaxSingle / 2 + W) / W);
    __shared__ double part[6][W][64];
    __shared__ double gbuf[W][NSEG][96];
    int sq, slot;
    block_map(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int d1 = d - 1;
    const int ncell = n - 1 - d > 0 ? n - 1 - d : 0;            // cells of diagonal d; diagonal d-1 has one more (if it exists)
    const int ncell1 = d1 >= 0 && n - 1 - d1 > 0 ? n - 1 - d1 : 0;
    if (n - 1 - d1 < 0) return;                                // d-1 > n-1: neither cells nor an F5o entry
    const int ncol = ncell > ncell1 ? ncell : ncell1;            // columns 1..ncol hold a cell of either diagonal
    const int ngroup = ncol > 0 ? (ncol > 1 ? (ncol - 1 + 62) / 63 : 1) : 0;   // group g: columns 1+63g .. 64+63g
    if (slot > ngroup) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ f5i = B.f5i + sq * ld;
    double* __restrict__ f5o = B.f5o + sq * ld;

    if (slot == ngroup) {
        // F5o[k] = F5o[k+1]*ext_unpaired + sum_{jj>=k+2} F5o[jj]*FCA[k+1,jj-1]*ext_paired   (ipp:3751-3780, pulled)
#pragma unroll 1
        for (int k = khi; k >= d1 && k >= 1; k--) {
            if (k > n - 1) continue;
            const double* __restrict__ fca = tab + L_FCA * ts + (k + 1);
            double acc = 0.0;
            for (int jj = k + 2 + threadIdx.x; jj <= n; jj += 64 * W) acc = fma(f5o[jj], fca[(jj - 2 - k) * ld], acc);
            acc = wsum(acc);
            if (lane == 0) part[0][w][0] = acc;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < W; q++) t += part[0][q][0];
                f5o[k] = f5o[k + 1] * L->w_eu + t * L->w_ep2;
            }
            __syncthreads();   // F5o[k] is an operand of F5o[k-1]
        }
        return;
    }

    const int i0 = 1 + slot * 63;
    const int i = i0 + lane, j = i + d, j1 = j - 1;
    const bool valid = i <= ncell;                                       // cell (i, j)   of diagonal d
    const bool valid1 = i <= ncell1 && (lane > 0 || slot == 0);          // cell (i, j-1) of diagonal d-1
    const bool any = valid || valid1;
    int s_im1 = 4, s_i = 4, s_ip1 = 4, s_jm1 = 4, s_j = 4, s_jp1 = 4, s_jp2 = 4;
    if (any) {
        s_im1 = s[i - 1]; s_i = s[i]; s_ip1 = s[i + 1]; s_j = s[j]; s_jp1 = s[j + 1];
        if (valid) s_jp2 = s[j + 2];
        if (j - 1 >= 0) s_jm1 = s[j - 1];
    }
    const bool pairable = valid && pairs(s_i, s_jp1);
    const bool pairable1 = valid1 && pairs(s_i, s_j);
    const bool guard_m = d >= 2, guard_m1 = d1 >= 2;

    double accm = 0.0, acc1 = 0.0, accc = 0.0;      // diagonal d
    double accmn = 0.0, acc1n = 0.0, acccn = 0.0;   // diagonal d-1 without its e = 1 terms
    // per-lane term limits (block structure as in lin_outside_diag)
    int mineA = valid ? i - 1 : 0, mineA1 = valid1 ? i - 1 : 0;
    if (BS > 0) { const int lim = i - (i / (BS > 0 ? BS : 1) - 1) * BS; mineA = mineA < lim ? mineA : lim; mineA1 = mineA1 < lim ? mineA1 : lim; }
    int mineB = valid ? n - 1 - j : 0, mineB1 = valid1 ? n - 1 - j1 : 0;
    if (BS > 0) {
        const int lim = (j / (BS > 0 ? BS : 1) + 2) * BS - 1 - j, lim1 = (j1 / (BS > 0 ? BS : 1) + 2) * BS - 1 - j1;
        mineB = mineB < lim ? mineB : lim; mineB1 = mineB1 < lim1 ? mineB1 : lim1;
    }
    if (!guard_m) { mineA = 0; mineB = 0; }
    if (!guard_m1) { mineA1 = 0; mineB1 = 0; }
    if (guard_m1) {   // guard_m implies guard_m1
        constexpr int UO = 4;
        // FMo[i,d] += FM2o[d+e][i-e] * FM1[e][i-e], e = 1..i-1            (ipp:4046-4064, pulled)
        // with blocks: only i' = i-e in blocks I-1, I are streamed; blocks <= I-2 come from FMOF
        {
            const int i_last = ncell1 < i0 + 63 ? ncell1 : i0 + 63;
            const int emax = BS > 0 ? (i_last - 1 < 2 * BS ? i_last - 1 : 2 * BS) : i_last - 1;
            const double* __restrict__ x = tab + L_FM2O * ts + i;
            const double* __restrict__ y = tab + L_FM1 * ts + i;
            for (int e = 1 + w; e <= emax; e += UO * W) {
                double xv[UO], xn[UO], yv[UO];
#pragma unroll
                for (int u = 0; u < UO; u++) {
                    const int ee = e + u * W;
                    const bool ok = ee <= mineA, ok1 = ee <= mineA1 && ee >= 2;
                    xv[u] = ok ? x[(d + ee) * ld - ee] : 0.0;
                    xn[u] = ok1 ? x[(d1 + ee) * ld - ee] : 0.0;
                    yv[u] = (ok || ok1) ? y[ee * ld - ee] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UO; u++) { accm = fma(xv[u], yv[u], accm); accmn = fma(xn[u], yv[u], accmn); }
            }
            if (BS > 0 && w == 0) {
                if (valid && guard_m) accm += tab[L_FMOF * ts + d * ld + i];
                if (valid1) accmn += tab[L_FMOF * ts + d1 * ld + i];
            }
        }
        // FM1o[i,d] += FM2o[d+e][i] * FM[e][i+d], e = 1..n-1-j; blocks >= J+2 come from FM1OF
        {
            const int emax_all = n - 1 - (i0 + d1);
            const int emax = BS > 0 ? (emax_all < 2 * BS ? emax_all : 2 * BS) : emax_all;
            const double* __restrict__ x = tab + L_FM2O * ts + i;
            const double* __restrict__ y = tab + L_FM * ts + j;
            for (int e = 1 + w; e <= emax; e += UO * W) {
                double xv[UO], xn[UO], yv[UO], yn[UO];
#pragma unroll
                for (int u = 0; u < UO; u++) {
                    const int ee = e + u * W;
                    const bool ok = ee <= mineB, ok1 = ee <= mineB1 && ee >= 2;
                    xv[u] = ok ? x[(d + ee) * ld] : 0.0;
                    yv[u] = ok ? y[ee * ld] : 0.0;
                    xn[u] = ok1 ? x[(d1 + ee) * ld] : 0.0;
                    yn[u] = ok1 ? y[ee * ld - 1] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < UO; u++) { acc1 = fma(xv[u], yv[u], acc1); acc1n = fma(xn[u], yn[u], acc1n); }
            }
            if (BS > 0 && w == 0) {
                if (valid && guard_m) acc1 += tab[L_FM1OF * ts + d * ld + i];
                if (valid1) acc1n += tab[L_FM1OF * ts + d1 * ld + i];
            }
        }
    }
    {   // enclosing single-branch loops: sum_t sum_l1 w(l1,t-l1) * FCoX[d+2+t][i-1-l1]       (ipp:4004-4024, pulled)
        // staged row r = 0..31 is table row d+1+r: filter t = r-1 of diagonal d over columns [i-r, i-1], filter t = r of
        // diagonal d-1 over [i-1-r, i-1]; a tap is valid when its column lies in [1, n-1-row] of the source row: the
        // segment is zero-filled outside that range while it is staged, so the taps themselves need no mask
        const int room1 = n - 3 - d;  // source row d+1+r <= n-2
        if (room1 >= 0) {
            constexpr int LAST = kMaxSingle + 1, HALF = (kMaxSingle + 1) / 2;
            const int rmax = room1 < LAST ? room1 : LAST;
            const double* __restrict__ fcox = tab + L_FCOX * ts;
#pragma unroll
            for (int q = 0; q < NSEG; q++) {
                const int g = w + (q >> 1) * W;
                const int r = (q & 1) ? LAST - g : g;
                const bool on = g <= HALF && r <= rmax;   // wave-uniform (LAST is odd: g never meets LAST-g)
                if (on) {
                    const int col0 = i0 - 1 - r;                  // segment column of lane k: col0+k; window of cell i: [i-1-r, i-1]
                    const double* __restrict__ row = fcox + (d + 1 + r) * ld;
                    const int cmax = n - 1 - (d + 1 + r);         // last interior column of that row
                    const int c = col0 + lane;
                    gbuf[w][q][lane] = (c >= 1 && c <= cmax) ? row[c] : 0.0;
                    const int c2 = col0 + 64 + lane;
                    if (lane < 32) gbuf[w][q][64 + lane] = (c2 >= 1 && c2 <= cmax) ? row[c2] : 0.0;
                }
            }
            switch (w) {
#define X(V) case V: filt_pass_pair<W, V, true>(L->shape_w, &gbuf[w][0][lane], rmax, accc, acccn); break;
                X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
            }
        }
    }
    part[0][w][lane] = accm;  part[1][w][lane] = acc1;  part[2][w][lane] = accc;
    part[3][w][lane] = accmn; part[4][w][lane] = acc1n; part[5][w][lane] = acccn;
    __syncthreads();
    if (w > 1) return;
    // wavefront 0 finishes diagonal d and hands its row-d values to wavefront 1 through LDS; wavefront 1 meanwhile loads the
    // epilogue operands of diagonal d-1.  (Two wavefronts: each holds one operand set -- in one wavefront the two sets cost
    // ~90 VGPRs and half the occupancy; operands are loaded after the term loops for the same reason.)
    __shared__ double hand[3][64];
    const size_t at = d * ld + i;
    const double o_f5i = any ? f5i[i - 1] : 0.0, o_z = any ? f5i[n] : 1.0;
    double n_tjb = 0, n_tja = 0, n_tst = 0, n_bp = 0, n_tjbd = 0, n_tjad = 0, n_n01 = 0, n_n10 = 0, n_n11 = 0;
    double n_f5o = 0, n_fm1o_up = 0, n_fco_up = 0, n_x01 = 0, n_x10 = 0, n_x11 = 0, n_fc = 0, n_y1 = 0, n_y2 = 0;
    double smn = 0.0, s1n = 0.0, gn = 0.0;
    if (w == 0) {
        double e_tjb = 0, e_tja = 0, e_tst = 0, e_bp = 0, e_tjbd = 0, e_tjad = 0, e_n01 = 0, e_n10 = 0, e_n11 = 0;
        double o_fmo = 0, o_fm1o = 0, o_f5o = 0, o_fm1o_up = 0, o_fco_up = 0, o_x01 = 0, o_x10 = 0, o_x11 = 0, o_fc = 0;
        if (valid) {
            const bool up_ok = i - 1 >= 1 && j + 1 <= n - 1;  // the cell (i-1, j+1) is interior
            const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1 + s_j;
            const int idd = 25 * (5 * s_jp1 + s_jp2) + 5 * s_i + s_im1;
            e_tjb = L->TJB[idx]; e_tja = L->TJA[idx]; e_bp = L->E_bp[s_i * 5 + s_jp1];
            e_tjbd = L->TJB[idd]; e_tjad = L->TJA[idd];
            e_tst = L->TST[25 * (5 * s_im1 + s_i) + 5 * s_jp2 + s_jp1];
            e_n01 = L->E_b01[s_jp2]; e_n10 = L->E_b10[s_im1]; e_n11 = L->E_11[s_im1 * 5 + s_jp2];
            if (guard_m) {
                if (j + 1 <= n - 1) o_fmo = tab[L_FMO * ts + (d + 1) * ld + i];            // ipp:3806
                if (i - 1 >= 1) o_fm1o = tab[L_FM1O * ts + (d + 1) * ld + i - 1];            // ipp:3833
            }
            o_f5o = f5o[j + 1];
            o_fc = tab[L_FC * ts + at];
            const double* __restrict__ fcox = tab + L_FCOX * ts;
            if (up_ok) {
                o_fm1o_up = tab[L_FM1O * ts + (d + 2) * ld + i - 1];                         // ipp:3828
                o_fco_up = tab[L_FCO * ts + (d + 2) * ld + i - 1];
            }
            if (i - 1 >= 1 && j + 2 <= n - 1) o_x01 = fcox[(d + 3) * ld + i - 1];
            if (i - 2 >= 1 && j + 1 <= n - 1) o_x10 = fcox[(d + 3) * ld + i - 2];
            if (i - 2 >= 1 && j + 2 <= n - 1) o_x11 = fcox[(d + 4) * ld + i - 2];
        }
        double sm = 0.0, s1 = 0.0, g = 0.0;
#pragma unroll
        for (int k = 0; k < W; k++) { sm += part[0][k][lane]; s1 += part[1][k][lane]; g += part[2][k][lane]; }
        double fmo = 0.0, fm1o = 0.0, fm2o = 0.0;
        if (valid) {
            if (guard_m) {
                fmo = sm + o_fmo * L->w_mu;                   // ipp:3806
                fm1o = s1 + fmo + o_fm1o * L->w_mu;           // ipp:3809, 3833
            }
            double fco = 0.0;
            if (pairable) {
                const double ext = o_f5o * o_f5i * L->w_ep2;  // exterior loop, ipp:3768-3776
                const double multi = o_fm1o_up * L->w_mp2;    // branch of a multiloop, ipp:3828
                const double sp = L->w01 * e_n01 * o_x01 + L->w10 * e_n10 * o_x10 + L->w11 * e_n11 * o_x11;
                const double st = o_fco_up * L->lam2 * e_tst; // stacked on (i-1,j+1)
                fco = e_bp * (e_tjad * (ext + multi) + e_tjbd * (g + sp)) + st;
            }
            fm2o = fmo + fco * e_tja * L->e_mpmb;                                                // ipp:3803, 4027
            tab[L_FCO * ts + at] = fco;
            tab[L_FCOX * ts + at] = fco * e_tjb;
            tab[L_FMO * ts + at] = fmo;
            tab[L_FM1O * ts + at] = fm1o;
            tab[L_FM2O * ts + at] = fm2o;
            // posterior of pair (i, j+1) = FCo * FCi / Z, clipped to [0,1]                 (ipp:4689-4827)
            double p = fco * o_fc / o_z;
            if (!(p == p) || p > 1e300) { atomicOr(&bad[sq], 1); p = 0.0; }
            p = p > 1.0 ? 1.0 : (p < 0.0 ? 0.0 : p);
            B.bp[(size_t)sq * B.tri_stride + tri_off(n, i) + (j + 1)] = p;
        }
        hand[0][lane] = fm2o; hand[1][lane] = fm1o; hand[2][lane] = fmo;
    } else {
        if (valid1) {   // operands of (i, j1 = j-1) on diagonal d1 = d-1; its row-d operands arrive through `hand`
            const bool up_ok = i - 1 >= 1 && j1 + 1 <= n - 1;
            const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_j + s_jm1;
            const int idd = 25 * (5 * s_j + s_jp1) + 5 * s_i + s_im1;
            n_tjb = L->TJB[idx]; n_tja = L->TJA[idx]; n_bp = L->E_bp[s_i * 5 + s_j];
            n_tjbd = L->TJB[idd]; n_tjad = L->TJA[idd];
            n_tst = L->TST[25 * (5 * s_im1 + s_i) + 5 * s_jp1 + s_j];
            n_n01 = L->E_b01[s_jp1]; n_n10 = L->E_b10[s_im1]; n_n11 = L->E_11[s_im1 * 5 + s_jp1];
            n_f5o = f5o[j1 + 1];
            n_fc = tab[L_FC * ts + d1 * ld + i];
            const double* __restrict__ fcox = tab + L_FCOX * ts;
            if (up_ok) {
                n_fm1o_up = tab[L_FM1O * ts + (d1 + 2) * ld + i - 1];
                n_fco_up = tab[L_FCO * ts + (d1 + 2) * ld + i - 1];
            }
            if (i - 1 >= 1 && j1 + 2 <= n - 1) n_x01 = fcox[(d1 + 3) * ld + i - 1];
            if (i - 2 >= 1 && j1 + 1 <= n - 1) n_x10 = fcox[(d1 + 3) * ld + i - 2];
            if (i - 2 >= 1 && j1 + 2 <= n - 1) n_x11 = fcox[(d1 + 4) * ld + i - 2];
            if (guard_m1) {
                if (i - 1 >= 1) n_y1 = tab[L_FM1 * ts + ld + i - 1];     // FM1[1][i-1]: partner of FM2o[d][i-1] (e = 1 of the FMo sum)
                n_y2 = tab[L_FM * ts + ld + j1];                          // FM [1][j-1]: partner of FM2o[d][i]   (e = 1 of the FM1o sum)
            }
        }
#pragma unroll
        for (int k = 0; k < W; k++) { smn += part[3][k][lane]; s1n += part[4][k][lane]; gn += part[5][k][lane]; }
    }
    __syncthreads();   // wavefronts 0 and 1 (the others have left)
    if (w == 0 || !valid1) return;
    {   // ---- diagonal d-1: row-d operands of column i (lane l) and i-1 (lane l-1)
        const double fm2o = hand[0][lane], fmo = hand[2][lane];
        const double fm2o_left = lane > 0 ? hand[0][lane - 1] : 0.0;   // FM2o[d][i-1]
        const double fm1o_left = lane > 0 ? hand[1][lane - 1] : 0.0;   // FM1o[d][i-1]
        double fmon = 0.0, fm1on = 0.0;
        if (guard_m1) {
            if (1 <= mineA1) smn = fma(fm2o_left, n_y1, smn);                      // e = 1: FM2o[d][i-1] * FM1[1][i-1]
            if (1 <= mineB1) s1n = fma(fm2o, n_y2, s1n);                           // e = 1: FM2o[d][i]   * FM [1][j-1]
            const double up_fmo = j1 + 1 <= n - 1 ? fmo : 0.0;                     // FMO [d][i]
            const double up_fm1o = i - 1 >= 1 ? fm1o_left : 0.0;                   // FM1O[d][i-1]
            fmon = smn + up_fmo * L->w_mu;
            fm1on = s1n + fmon + up_fm1o * L->w_mu;
        }
        double fcon = 0.0;
        if (pairable1) {
            const double ext = n_f5o * o_f5i * L->w_ep2;
            const double multi = n_fm1o_up * L->w_mp2;
            const double sp = L->w01 * n_n01 * n_x01 + L->w10 * n_n10 * n_x10 + L->w11 * n_n11 * n_x11;
            const double st = n_fco_up * L->lam2 * n_tst;
            fcon = n_bp * (n_tjad * (ext + multi) + n_tjbd * (gn + sp)) + st;
        }
        const double fm2on = fmon + fcon * n_tja * L->e_mpmb;
        const size_t at1 = d1 * ld + i;
        tab[L_FCO * ts + at1] = fcon;
        tab[L_FCOX * ts + at1] = fcon * n_tjb;
        tab[L_FMO * ts + at1] = fmon;
        tab[L_FM1O * ts + at1] = fm1on;
        tab[L_FM2O * ts + at1] = fm2on;
        double p = fcon * n_fc / o_z;
        if (!(p == p) || p > 1e300) { atomicOr(&bad[sq], 1); p = 0.0; }
        p = p > 1.0 ? 1.0 : (p < 0.0 ? 0.0 : p);
        B.bp[(size_t)sq * B.tri_stride + tri_off(n, i) + (j1 + 1)] = p;
    }
}

// logZ = log F5i~[n] + s*n; flags a sequence whose scaled values left the double range
__global__ void lin_finish(McBatch B, const LinModel* __restrict__ L, double* __restrict__ logz, int* __restrict__ bad)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const double z = B.f5i[(size_t)sq * B.ld + n];
    const double zo = n >= 2 ? B.f5o[(size_t)sq * B.ld + 1] : 1.0;
    if (!(z > 1e-200 && z < 1e200) || !(zo > 1e-200 && zo < 1e200)) atomicOr(&bad[sq], 1);  // F5o~[1] ~ Z~ too
    logz[sq] = log(z) + L->s * (double)n;
}

// the host side instantiates the group width / block size it wants
#define RH_INST(W, BS)                                                                                \
    template __global__ void lin_inside_diag<W, BS, 0>(McBatch, const LinModel*, int, double, int);  \
    template __global__ void lin_outside_diag<W, BS>(McBatch, const LinModel*, int, int, int*);
RH_INST(8, 0) RH_INST(16, 0)
RH_INST(8, 16) RH_INST(16, 16) RH_INST(4, 16)
RH_INST(8, 32) RH_INST(16, 32)
#undef RH_INST
template __global__ void lin_inside_diag<4, 16, 1>(McBatch, const LinModel*, int, double, int);   // look-ahead pair
template __global__ void lin_inside_diag<4, 16, 2>(McBatch, const LinModel*, int, double, int);
template __global__ void lin_inside_diag<4, 16, 3>(McBatch, const LinModel*, int, double, int);   // both diagonals in one launch
template __global__ void lin_outside_pair<8, 16>(McBatch, const LinModel*, int, int, int, int*);
template __global__ void lin_outside_pair<4, 16>(McBatch, const LinModel*, int, int, int, int*);

}  // namespace rh
