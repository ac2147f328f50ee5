// duplex_lin.hip -- duplex (hybridization) partition function in SCALED LINEAR space.
//
// Same recurrences as duplex.hip (reference: /root/reference/src/contrafold/DuplexEngine.ipp:
// 1015-1077 inside, 1080-1143 outside, 1146-1169 posterior; loop nest of src/pf_duplex.c:128-206),
// reorganised for the hardware:
//   * coordinates a = i, b = L2+1-j; cells of anti-diagonal sd = a+b are stored contiguously,
//     [sd*lda + kPad + a]: one THREAD per cell, 64 consecutive cells per wavefront;
//   * values are IN*lam^(a+b) and OUT*lam^((L1+1-a)+(L2+1-b)), lam = exp(-s): a loop that skips
//     t = l1+l2 unpaired letters costs lam^(t+2), and since DuplexEngine has NO length-dependent
//     loop score (its cache_score_single is never read), all t+1 shapes of one t share one weight:
//     their sum is a sliding window over a row segment, staged once in LDS;
//   * rows are padded with kPad zero columns on both sides so windows never need bounds checks;
//   * the 6 shapes with t <= 2 (stacking pair, the nucleotide-dependent 0x1/1x0/1x1 loops) are
//     added per cell in the epilogue.
// A pair whose scaled partition function leaves the double range is flagged and recomputed by the
// log-space kernels (duplex.hip).
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lin_model.h"

#ifndef RH_WPE_DX
#define RH_WPE_DX
#endif

namespace rh {

namespace {
constexpr uint32_t kPairMaskD = (1u << (0 * 5 + 3)) | (1u << (3 * 5 + 0)) | (1u << (1 * 5 + 2)) |
                                (1u << (2 * 5 + 1)) | (1u << (2 * 5 + 3)) | (1u << (3 * 5 + 2));
__device__ __forceinline__ bool pairs(int a, int b) { return (kPairMaskD >> (a * 5 + b)) & 1u; }
// (T+1)-wide window sum over an LDS-resident segment, fully unrolled (no scalar loop control per tap)
// volatile: keeps every tap a plain ds_read_b64 (2 LDS cycles per wavefront, 256 B/clk/CU); merged into
// ds_read2_b64 by the compiler, two taps cost 8 cycles (128 B/clk/CU) on CDNA4
typedef const volatile __attribute__((address_space(3))) double* lds_vptr;
template <int T>
__device__ __forceinline__ double win_sum(const double* seg)
{
    lds_vptr vs = (lds_vptr)seg;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k <= T; k += 2) {
        s0 += vs[k];
        if (k + 1 <= T) s1 += vs[k + 1];
    }
    return s0 + s1;
}
__device__ __forceinline__ double win_sum_any(int t, const double* seg)
{
    switch (t) {
#define X(T) case T: return win_sum<T>(seg);
        X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22)
        X(23) X(24) X(25) X(26) X(27) X(28)
#undef X
    }
    return 0.0;
}
// all windows of wavefront WV of a W-wavefront group (see dxl_sweep): window q has length t = 2+WV+q*W on source row
// sdA -/+ (2+t); diagonal A takes it with weight lam^(t+2) (3 <= t), diagonal B takes it plus one more column with lam^(t+3)
template <int W, int WV>
__device__ __forceinline__ void win_pass(const double* seg0, const double* __restrict__ lam_pow, bool outside, int sdA, int smax, double& accA, double& accB)
{
    if constexpr (WV < W) {
        constexpr int NSEG = (27 + W - 1) / W;
#pragma unroll
        for (int q = 0; q < NSEG; q++) {
            constexpr int t0 = 2 + WV;
            const int t = t0 + q * W;        // compile-time after unrolling
            if (t > 28) continue;
            const int row = outside ? sdA + 2 + t : sdA - 2 - t;
            if (row >= 2 && row <= smax) {
                const double* seg = seg0 + q * 96;
                double s0 = 0.0, s1 = 0.0;
                lds_vptr vs = (lds_vptr)seg;
#pragma unroll
                for (int k = 0; k <= t; k += 2) {
                    s0 += vs[k];
                    if (k + 1 <= t) s1 += vs[k + 1];
                }
                const double wa = s0 + s1;
                const double wb = wa + (outside ? vs[t + 1] : vs[-1]);
                if (t >= 3) accA = fma(lam_pow[t + 2], wa, accA);
                if (t <= 27) accB = fma(lam_pow[t + 3], wb, accB);
            }
        }
    }
}
}  // namespace

// ---------------------------------------------------------------------------------
// launch `step`: inside diagonals A = 2+2*step and B = A+1, outside diagonals A = Smax-2*step and B = A-1
// (Smax = L1+L2) -- blockIdx.z selects inside/outside, blockIdx.x the 64-column group.  The two diagonals of a
// direction do not depend on each other and are computed by the SAME workgroup, one cell of each per lane: the window of
// length t on source row r belongs to diagonal A, and diagonal B reads the same row with length t+1 over the same columns
// plus one -- so every staged row and every LDS tap but one serves both cells.
template <int W>
__global__ __launch_bounds__(64 * W) RH_WPE_DX void dxl_sweep(DxLinBatch B, const DxLinModel* __restrict__ L, int step, int groups)
{
    __shared__ double buf[W][(27 + W - 1) / W][96];
    __shared__ double part[2][W][64];
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const bool outside = blockIdx.z != 0;
    const int grp = blockIdx.x;
    const int smax = L1 + L2;
    const int sdA = outside ? smax - 2 * step : 2 + 2 * step;
    const int sdB = outside ? sdA - 1 : sdA + 1;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a0 = grp * 64;
    if (a0 > B.n1max + 1) return;
    const int a = a0 + lane;
    const int lda = B.lda;
    const size_t ts = B.tab_stride;
    double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const bool okA = sdA >= 2 && sdA <= smax, okB = sdB >= 2 && sdB <= smax;   // the diagonal exists
    if (!okA && !okB) return;
    // does this group hold cells [max(1, sd-L2), min(L1, sd-1)] of either diagonal?
    const bool hasA = okA && !(a0 + 63 < (sdA - L2 > 1 ? sdA - L2 : 1) || a0 > (sdA - 1 < L1 ? sdA - 1 : L1));
    const bool hasB = okB && !(a0 + 63 < (sdB - L2 > 1 ? sdB - L2 : 1) || a0 > (sdB - 1 < L1 ? sdB - 1 : L1));
    if (!hasA && !hasB) {   // only clear the columns of the rows
        if (w < 2 && a <= B.n1max + 1) {
            const int sd = w == 0 ? sdA : sdB;
            if (w == 0 ? okA : okB) {
                const size_t at = (size_t)sd * lda + kDxPad + a;
                tab[(outside ? DL_OUT : DL_IN) * ts + at] = 0.0;
                tab[(outside ? DL_OUTX : DL_INX) * ts + at] = 0.0;
            }
        }
        return;
    }

    // epilogue: wavefront 0 finishes diagonal A, wavefront 1 diagonal B; operands issued before the window work
    const int sd = w == 1 ? sdB : sdA;
    const bool mine = w == 0 ? okA : (w == 1 ? okB : false);
    const int b = sd - a;
    const int i = a, j = L2 + 1 - b;
    const bool incell = mine && a >= 1 && a <= L1 && b >= 1 && b <= L2;
    const size_t at = (size_t)sd * lda + kDxPad + a;
    int x = 4, xm = 4, xp = 4, y = 4, ym = 4, yp = 4;  // s1[i], s1[i-1], s1[i+1], s2[j], s2[j-1], s2[j+1]
    if (incell) { x = s1[i]; xm = s1[i - 1]; xp = s1[i + 1]; y = s2[j]; ym = s2[j - 1]; yp = s2[j + 1]; }
    const bool pairable = incell && pairs(x, y);
    const double* __restrict__ rawt = tab + (outside ? DL_OUT : DL_IN) * ts + kDxPad;
    const double* __restrict__ dect = tab + (outside ? DL_OUTX : DL_INX) * ts + kDxPad;
    double o_st = 0, o_01 = 0, o_10 = 0, o_02 = 0, o_11 = 0, o_20 = 0;
    double e_up = 0, e_dn = 0, e_ends = 0, e_st = 0, e_b01 = 0, e_b10 = 0, e_11 = 0;
    if (pairable) {
        const int dir = outside ? 1 : -1;     // sources lie at rows sd + dir*(2+t), columns a + dir*(1+l1)
        const int r2 = sd + 2 * dir, r3 = sd + 3 * dir, r4 = sd + 4 * dir;
        if (r2 >= 2 && r2 <= smax) o_st = rawt[(size_t)r2 * lda + a + dir];
        if (r3 >= 2 && r3 <= smax) { o_01 = dect[(size_t)r3 * lda + a + dir]; o_10 = dect[(size_t)r3 * lda + a + 2 * dir]; }
        if (r4 >= 2 && r4 <= smax) {
            o_02 = dect[(size_t)r4 * lda + a + dir]; o_11 = dect[(size_t)r4 * lda + a + 2 * dir]; o_20 = dect[(size_t)r4 * lda + a + 3 * dir];
        }
        e_up = L->E_tm[((x * 5 + y) * 5 + xp) * 5 + ym];                          // terminal_mismatch[s1[i]][s2[j]][s1[i+1]][s2[j-1]]
        e_dn = L->E_tm[((y * 5 + x) * 5 + yp) * 5 + xm] * L->E_bp[x * 5 + y];     // terminal_mismatch[s2[j]][s1[i]][s2[j+1]][s1[i-1]] * base_pair
        if (!outside) {
            e_ends = L->E_dr[y * 25 + x * 5 + xm] * L->E_dl[y * 25 + x * 5 + yp] * L->E_bp[y * 5 + x] * L->E_hc[y * 5 + x];
            e_st = L->E_bp[x * 5 + y] * L->E_hs[((xm * 5 + yp) * 5 + x) * 5 + y];
            e_b01 = L->E_b01[yp]; e_b10 = L->E_b10[xm]; e_11 = L->E_11[xm * 5 + yp];
        } else {
            e_ends = L->E_dl[x * 25 + y * 5 + xp] * L->E_dr[x * 25 + y * 5 + ym] * L->E_hc[x * 5 + y];
            e_st = L->E_bp[xp * 5 + ym] * L->E_hs[((x * 5 + y) * 5 + xp) * 5 + ym];
            e_b01 = L->E_b01[ym]; e_b10 = L->E_b10[xp]; e_11 = L->E_11[xp * 5 + ym];
        }
    }

    // ---- windows: source row r = sdA -/+ (2+t), t = 2..28, serves diagonal A with length t (3 <= t <= 28) over columns
    // [a-1-t, a-1] (inside) / [a+1, a+1+t] (outside), and diagonal B with length t+1 (3 <= t+1 <= 28): one more column
    // on the far side.  weight lam^(length+2)
    const double* __restrict__ src = tab + (outside ? DL_OUTX : DL_INX) * ts + kDxPad;
    double accA = 0.0, accB = 0.0;
    constexpr int NSEG = (27 + W - 1) / W;   // rows t = 2..28 dealt round-robin to the W wavefronts
    // pass 1: stage every segment this wavefront sums (all row loads in flight at once)
#pragma unroll
    for (int q = 0; q < NSEG; q++) {
        const int t = 2 + w + q * W;
        const int row = outside ? sdA + 2 + t : sdA - 2 - t;
        const bool on = t <= 28 && row >= 2 && row <= smax;   // wave-uniform
        if (on) {
            // segment start: inside a0-2-t (B's window = [a-2-t, a-1]); outside a0+1 (B's window = [a+1, a+2+t])
            const int c0 = outside ? a0 + 1 : a0 - 2 - t;
            const double* __restrict__ r = src + (size_t)row * lda + c0;
            buf[w][q][lane] = r[lane];
            if (lane < 32) buf[w][q][64 + lane] = r[64 + lane];
        }
    }
    // pass 2: the window sums.  One specialisation per wavefront index: its window lengths t = 2+w, 2+w+W, ... are then
    // compile-time constants (straight-line taps, immediate offsets, no per-window dispatch on the scalar unit), and the code
    // stays one copy of every window length (unrolled over q inside ONE body it was NSEG copies of the whole switch, larger than
    // the instruction cache two CUs share)
    {
        const double* seg0 = &buf[w][0][lane + (outside ? 0 : 1)];       // A's window of segment q = seg0[q*96 .. q*96+t]
        switch (w) {
#define X(V) case V: win_pass<W, V>(seg0, L->lam_pow, outside, sdA, smax, accA, accB); break;
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        }
    }
    part[0][w][lane] = accA;
    part[1][w][lane] = accB;
    __syncthreads();
    if (w > 1 || !mine) return;
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) g += part[w][k][lane];

    double v = 0.0, vx = 0.0;
    if (pairable) {
        // inside : inside[i][j]  = open  + stack + down * (0x1/1x0/t=2 shapes + windows)   (DuplexEngine.ipp:1029-1064)
        // outside: outside[p][q] = close + stack + up   * (...)                             (DuplexEngine.ipp:1094-1129, pulled)
        const double l2 = L->lam_pow[2], l3 = L->lam_pow[3], l4 = L->lam_pow[4];
        const double ends = (outside ? B.pw_out[w == 0 ? 1 : 0] : B.pw_in[w]) * e_ends;
        const double sp = l3 * (e_b01 * o_01 + e_b10 * o_10) + l4 * (o_02 + e_11 * o_11 + o_20);
        const double own = outside ? e_up : e_dn;      // this cell's factor of every loop term
        v = ends + o_st * l2 * e_st + own * (sp + g);
        vx = v * (outside ? e_dn : e_up);              // decorated as the other end of a later loop
    }
    if (a <= B.n1max + 1) {  // every column of the row is rewritten: stale values of other shapes never survive
        tab[(outside ? DL_OUT : DL_IN) * ts + at] = v;
        tab[(outside ? DL_OUTX : DL_INX) * ts + at] = vx;
    }
}

// ---------------------------------------------------------------------------------
// Four anti-diagonals per launch.  Launch `step` covers, per direction, the diagonals X_k = A + k*dir, k = 0..3 (inside:
// A = 2+4*step, dir = +1; outside: A = Smax-4*step, dir = -1).  Diagonal X_k reads only rows X_k - dir*2 and beyond, so
//   * every window row of all four (rows A - dir*(2+t), t = 0..28) is final before the launch: the window of length t on that
//     row belongs to X_0, and X_1 / X_2 / X_3 read the same row with length t+1 / t+2 / t+3 over the same columns plus one /
//     two / three -- the wavefront that sums it adds three more taps and serves all four cells of the lane;
//   * X_0 and X_1 read nothing of this launch; X_2 reads X_0 (stacked pair, column a-dir), X_3 reads X_1 (stacked) and X_0
//     (the 0x1 / 1x0 loops, columns a-dir, a-2*dir).  Wavefronts 0 / 1 finish X_0 / X_1 and leave their values in LDS,
//     wavefronts 2 / 3 have meanwhile loaded the other operands of X_2 / X_3 and finish them after one barrier.
// A lane needs its neighbours a-dir and a-2*dir of the same group: groups advance by 62 columns, X_2 / X_3 are written by the
// 62 lanes that have both neighbours (all 64 in the group at the open end), X_0 / X_1 by all 64 (two columns twice, bit for bit).
struct DxCellOps { double o_st, o_01, o_10, o_02, o_11, o_20, e_up, e_dn, e_ends, e_st, e_b01, e_b10, e_11; };

// coefficient part of the operands of cell (sd, a) and the loads of the rows it may take from memory (mask bit k: row sd+dir*(2+k))
// the loads of a cell that depend on nothing but its position: letters and the (up to six) table cells of the rows before it
struct DxCellRaw { int x, xm, xp, y, ym, yp; double v_st, v_01, v_10, v_02, v_11, v_20; bool incell, ok2, ok3, ok4; };
__device__ __forceinline__ DxCellRaw dx_cell_loads(int lda, const double* __restrict__ rawt, const double* __restrict__ dect,
                                                   const uint8_t* __restrict__ s1, const uint8_t* __restrict__ s2, bool outside, int sd, int a,
                                                   int L1, int L2, int smax, int rows_mask)
{
    // Branch-free: every load is issued whatever the cell is (addresses clamped into the tables), so that the letters and the six
    // table cells travel together and the weight tables follow in ONE more round trip; a load behind `if (!pairs) return` is only
    // issued once the letters have arrived.  The rows have kDxPad >= 3 columns on both sides.
    DxCellRaw r;
    const int b = sd - a;
    r.incell = sd >= 2 && sd <= smax && a >= 1 && a <= L1 && b >= 1 && b <= L2;
    const int i = r.incell ? a : 1, j = r.incell ? L2 + 1 - b : 1;
    r.x = s1[i]; r.xm = s1[i - 1]; r.xp = s1[i + 1]; r.y = s2[j]; r.ym = s2[j - 1]; r.yp = s2[j + 1];
    const int dir = outside ? 1 : -1;     // sources lie at rows sd + dir*(2+t), columns a + dir*(1+l1)
    const int r2 = sd + 2 * dir, r3 = sd + 3 * dir, r4 = sd + 4 * dir;
    r.ok2 = (rows_mask & 1) && r2 >= 2 && r2 <= smax; r.ok3 = (rows_mask & 2) && r3 >= 2 && r3 <= smax; r.ok4 = (rows_mask & 4) && r4 >= 2 && r4 <= smax;
    const int ac = a < 0 ? 0 : (a > L1 + 1 ? L1 + 1 : a);   // a cell has 1 <= a <= L1; anything else only needs an address inside the row
    const int q2 = (r.ok2 ? r2 : 2) * lda + ac, q3 = (r.ok3 ? r3 : 2) * lda + ac, q4 = (r.ok4 ? r4 : 2) * lda + ac;   // >= 2*lda: 32-bit unsigned offsets
    r.v_st = rawt[(unsigned)(q2 + dir)];
    r.v_01 = dect[(unsigned)(q3 + dir)]; r.v_10 = dect[(unsigned)(q3 + 2 * dir)];
    r.v_02 = dect[(unsigned)(q4 + dir)]; r.v_11 = dect[(unsigned)(q4 + 2 * dir)]; r.v_20 = dect[(unsigned)(q4 + 3 * dir)];
    return r;
}
// ... and what depends on the letters: is it a pair, and the weights of its loops
__device__ __forceinline__ DxCellOps dx_cell_weights(const DxLinModel* __restrict__ L, const DxCellRaw& r, bool outside, bool* pairable)
{
    DxCellOps o;
    const int x = r.x, xm = r.xm, xp = r.xp, y = r.y, ym = r.ym, yp = r.yp;
    const bool pr = r.incell && pairs(x, y);
    *pairable = pr;
    o.o_st = pr && r.ok2 ? r.v_st : 0.0;
    o.o_01 = pr && r.ok3 ? r.v_01 : 0.0; o.o_10 = pr && r.ok3 ? r.v_10 : 0.0;
    o.o_02 = pr && r.ok4 ? r.v_02 : 0.0; o.o_11 = pr && r.ok4 ? r.v_11 : 0.0; o.o_20 = pr && r.ok4 ? r.v_20 : 0.0;
    o.e_up = L->E_tm[((x * 5 + y) * 5 + xp) * 5 + ym];                          // terminal_mismatch[s1[i]][s2[j]][s1[i+1]][s2[j-1]]
    o.e_dn = L->E_tm[((y * 5 + x) * 5 + yp) * 5 + xm] * L->E_bp[x * 5 + y];     // terminal_mismatch[s2[j]][s1[i]][s2[j+1]][s1[i-1]] * base_pair
    if (!outside) {
        o.e_ends = L->E_dr[y * 25 + x * 5 + xm] * L->E_dl[y * 25 + x * 5 + yp] * L->E_bp[y * 5 + x] * L->E_hc[y * 5 + x];
        o.e_st = L->E_bp[x * 5 + y] * L->E_hs[((xm * 5 + yp) * 5 + x) * 5 + y];
        o.e_b01 = L->E_b01[yp]; o.e_b10 = L->E_b10[xm]; o.e_11 = L->E_11[xm * 5 + yp];
    } else {
        o.e_ends = L->E_dl[x * 25 + y * 5 + xp] * L->E_dr[x * 25 + y * 5 + ym] * L->E_hc[x * 5 + y];
        o.e_st = L->E_bp[xp * 5 + ym] * L->E_hs[((x * 5 + y) * 5 + xp) * 5 + ym];
        o.e_b01 = L->E_b01[ym]; o.e_b10 = L->E_b10[xp]; o.e_11 = L->E_11[xp * 5 + ym];
    }
    return o;
}
__device__ __forceinline__ DxCellOps dx_cell_ops(int lda, const DxLinModel* __restrict__ L, const double* __restrict__ rawt,
                                                 const double* __restrict__ dect, const uint8_t* __restrict__ s1, const uint8_t* __restrict__ s2,
                                                 bool outside, int sd, int a, int L1, int L2, int smax, int rows_mask, bool* pairable)
{
    const DxCellRaw r = dx_cell_loads(lda, rawt, dect, s1, s2, outside, sd, a, L1, L2, smax, rows_mask);
    return dx_cell_weights(L, r, outside, pairable);
}
// inside : inside[i][j]  = open  + stack + down * (0x1/1x0/t=2 shapes + windows)   (DuplexEngine.ipp:1029-1064)
// outside: outside[p][q] = close + stack + up   * (...)                             (DuplexEngine.ipp:1094-1129, pulled)
__device__ __forceinline__ void dx_cell_value(const DxLinModel* __restrict__ L, const DxCellOps& o, bool outside, double pw, double g, double* v, double* vx)
{
    const double l2 = L->lam_pow[2], l3 = L->lam_pow[3], l4 = L->lam_pow[4];
    const double ends = pw * o.e_ends;
    const double sp = l3 * (o.e_b01 * o.o_01 + o.e_b10 * o.o_10) + l4 * (o.o_02 + o.e_11 * o.o_11 + o.o_20);
    const double own = outside ? o.e_up : o.e_dn;      // this cell's factor of every loop term
    *v = ends + o.o_st * l2 * o.e_st + own * (sp + g);
    *vx = *v * (outside ? o.e_dn : o.e_up);            // decorated as the other end of a later loop
}

// windows of wavefront WV: rows t = WV, WV+4, ... <= 28 (row index relative to X_0: table row A - dir*(2+t)); lengths are
// compile-time constants.  seg0[q*96 + k]: inside column a-4-t+k (X_0's window = seg[3..3+t], X_k adds seg[3-k]); outside
// column a+1+k (X_0's window = seg[0..t], X_k adds seg[t+k])
template <int WV>
__device__ __forceinline__ void win_pass4(const double* seg0, const double* __restrict__ lam_pow, bool outside, int sdA, int smax, double acc[4])
{
    if constexpr (WV < 4) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int t = WV + 4 * q;        // compile-time after unrolling
            if (t > 28) continue;
            const int row = outside ? sdA + 2 + t : sdA - 2 - t;
            if (row >= 2 && row <= smax) {
                lds_vptr vs = (lds_vptr)(seg0 + q * 96);
                double s0 = 0.0, s1 = 0.0;
                const int base = outside ? 0 : 3;
#pragma unroll
                for (int k = 0; k <= t; k += 2) {
                    s0 += vs[base + k];
                    if (k + 1 <= t) s1 += vs[base + k + 1];
                }
                const double w0 = s0 + s1;
                const double w1 = w0 + (outside ? vs[t + 1] : vs[2]);
                const double w2 = w1 + (outside ? vs[t + 2] : vs[1]);
                const double w3 = w2 + (outside ? vs[t + 3] : vs[0]);
                if (t >= 3) acc[0] = fma(lam_pow[t + 2], w0, acc[0]);                 // lengths 3..28 only
                if (t + 1 >= 3 && t + 1 <= 28) acc[1] = fma(lam_pow[t + 3], w1, acc[1]);
                if (t + 2 >= 3 && t + 2 <= 28) acc[2] = fma(lam_pow[t + 4], w2, acc[2]);
                if (t + 3 <= 28) acc[3] = fma(lam_pow[t + 5], w3, acc[3]);
            }
        }
    }
}

__global__ __launch_bounds__(256) RH_WPE_DX void dxl_sweep4(DxLinBatch B, const DxLinModel* __restrict__ L, int step, int groups)
{
    __shared__ double buf[4][8][96];
    __shared__ double part[4][4][64];
    __shared__ double hand[3][64];      // raw X_0, decorated X_0, raw X_1 of this group's columns
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const bool outside = blockIdx.z != 0;
    const int grp = blockIdx.x;
    const int smax = L1 + L2;
    const int dir = outside ? 1 : -1, fwd = -dir;               // X_k = A + k*fwd
    const int sdA = outside ? smax - 4 * step : 2 + 4 * step;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a0 = grp * 62;
    if (a0 > B.n1max + 1) return;
    const int a = a0 + lane;
    const int lda = B.lda;
    const size_t ts = B.tab_stride;
    double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    if (outside ? sdA < 2 : sdA > smax) return;                 // none of the four diagonals exists
    // lanes that write X_2 / X_3: those with both neighbours a+dir, a+2*dir in this group, and everything at the open end
    const bool owns23 = outside ? (lane <= 61 || a0 + 62 > B.n1max + 1) : (lane >= 2 || grp == 0);
    bool has = false;   // does this group hold cells [max(1, sd-L2), min(L1, sd-1)] of any of the four diagonals?
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sd = sdA + k * fwd;
        has = has || (sd >= 2 && sd <= smax && !(a0 + 63 < (sd - L2 > 1 ? sd - L2 : 1) || a0 > (sd - 1 < L1 ? sd - 1 : L1)));
    }
    const int sdw = sdA + w * fwd;      // the diagonal wavefront w finishes
    const bool mine = sdw >= 2 && sdw <= smax;
    const size_t at = (size_t)sdw * lda + kDxPad + a;
    if (!has) {   // only clear the columns of the rows
        if (mine && a <= B.n1max + 1 && (w < 2 || owns23)) {
            tab[(outside ? DL_OUT : DL_IN) * ts + at] = 0.0;
            tab[(outside ? DL_OUTX : DL_INX) * ts + at] = 0.0;
        }
        return;
    }

    // ---- windows (all four wavefronts): stage the rows t = w, w+4, ... then sum
    const double* __restrict__ src = tab + (outside ? DL_OUTX : DL_INX) * ts + kDxPad;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int t = w + 4 * q;
        const int row = outside ? sdA + 2 + t : sdA - 2 - t;
        if (t <= 28 && row >= 2 && row <= smax) {   // wave-uniform
            const int c0 = outside ? a0 + 1 : a0 - 4 - t;
            const double* __restrict__ r = src + (size_t)row * lda + c0;
            buf[w][q][lane] = r[lane];
            if (lane < 32) buf[w][q][64 + lane] = r[64 + lane];
        }
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    switch (w) {
#define X(V) case V: win_pass4<V>(&buf[w][0][lane], L->lam_pow, outside, sdA, smax, acc); break;
        X(0) X(1) X(2) X(3)
#undef X
    }
#pragma unroll
    for (int k = 0; k < 4; k++) part[k][w][lane] = acc[k];

    // ---- epilogues: wavefront w finishes X_w.  Operands from rows of earlier launches are loaded now (before the barrier);
    // X_2 takes its stacked-pair operand, X_3 its stacked-pair and 0x1 / 1x0 operands from `hand`
    const double* __restrict__ rawt = tab + (outside ? DL_OUT : DL_IN) * ts + kDxPad;
    const double* __restrict__ dect = tab + (outside ? DL_OUTX : DL_INX) * ts + kDxPad;
    bool pairable;
    DxCellOps o = dx_cell_ops(lda, L, rawt, dect, s1, s2, outside, sdw, a, L1, L2, smax, w < 2 ? 7 : (w == 2 ? 6 : 4), &pairable);
    __syncthreads();
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) g += part[w][k][lane];
    double v = 0.0, vx = 0.0;
    if (w < 2) {
        if (mine && pairable) dx_cell_value(L, o, outside, B.pw4[w], g, &v, &vx);
        if (w == 0) { hand[0][lane] = v; hand[1][lane] = vx; } else hand[2][lane] = v;
    }
    __syncthreads();
    if (w >= 2) {
        if (!mine || !owns23) return;
        if (pairable) {
            const int l1 = lane + dir, l2 = lane + 2 * dir;                  // neighbours a+dir, a+2*dir; beyond the open end: no cell
            const bool in1 = l1 >= 0 && l1 < 64, in2 = l2 >= 0 && l2 < 64;
            if (w == 2) o.o_st = in1 ? hand[0][l1] : 0.0;                    // X_2 stacks on X_0
            else {                                                           // X_3 stacks on X_1, 0x1 / 1x0 loops close on X_0
                o.o_st = in1 ? hand[2][l1] : 0.0;
                o.o_01 = in1 ? hand[1][l1] : 0.0;
                o.o_10 = in2 ? hand[1][l2] : 0.0;
            }
            dx_cell_value(L, o, outside, B.pw4[w], g, &v, &vx);
        }
    } else if (!mine) return;
    if (a <= B.n1max + 1) {  // every column of the row is rewritten: stale values of other shapes never survive
        tab[(outside ? DL_OUT : DL_IN) * ts + at] = v;
        tab[(outside ? DL_OUTX : DL_INX) * ts + at] = vx;
    }
}

// ---------------------------------------------------------------------------------
// Eight anti-diagonals per launch (the strip form of the McCaskill sweeps, mccaskill_strip.hip, for the duplex).  Launch `step`
// covers X_k = A + k*fwd, k = 0..7 (inside: A = 2+8*step, fwd = +1; outside: A = Smax-8*step, fwd = -1).  X_k reads rows
// X_k - fwd*(2+t), t = 0..28, at columns a + dir*(1..t+1):
//   * the 30 rows A - fwd*r, r = 1..30, are final before the launch and are staged ONCE in LDS (100 columns); on row r the window
//     of X_k has length r+k-1, all starting at the same column, so one running sum over the row serves all eight diagonals;
//     the 8 wavefronts deal the rows round-robin and meet through partial sums in LDS (over the dead staging area);
//   * X_k also reads the strip's own rows k' <= k-2 (kept in LDS, raw and decorated): four time slots, two diagonals each
//     (wavefronts 2s, 2s+1), an LDS-only barrier in between;
//   * X_k reaches k-1-k' columns into row k': the valid lanes shrink by up to 6 from the `dir` side, groups advance by 58 columns
//     and the overlap is recomputed.  Every column 0..n1max+1 of the 8 rows is written (0 where there is no cell).
__device__ __forceinline__ void lds_barrier_dx() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int WV>
__device__ __forceinline__ void win_pass8(const double* seg0, const double* __restrict__ lam_pow, bool outside, int sdA, int smax, double acc[8])
{
    // rows r = 1 + WV + 8q of wavefront WV; seg0[q*104 + c]: inside column a-36+c (window element l = 1.. at index 36-l), outside
    // column a+1+c (element l at index l-1)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 1 + WV + 8 * q;            // compile-time after unrolling
        if (r > 30) continue;
        const int row = outside ? sdA + r : sdA - r;
        if (row >= 2 && row <= smax) {
            lds_vptr vs = (lds_vptr)(seg0 + q * 104);
            const int len_max = r + 6 < 29 ? r + 6 : 29;      // longest window any of the 8 diagonals takes from this row
            double run = 0.0;
            // eight reads at a time: left alone the scheduler hoists the reads of the whole row above the (dependent) running sum,
            // and the registers they occupy cost the kernel a workgroup per CU
#pragma unroll
            for (int l0 = 1; l0 <= len_max; l0 += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) if (l0 + u <= len_max) x[u] = outside ? vs[l0 + u - 1] : vs[36 - l0 - u];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int l = l0 + u;
                    if (l > len_max) continue;
                    run += x[u];
                    const int k = l - r + 1, t = l - 1;        // the window of length l belongs to X_k, t = r+k-2 = l-1
                    if (k >= 0 && k < 8 && t >= 3 && t <= 28) acc[k] = fma(lam_pow[t + 2], run, acc[k]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

#ifdef RH_DSTAMPS
// tuning build only (tools/build_variant.py dstamps -DRH_DSTAMPS): per-phase cycle totals of dxl_strip8 (wavefront 0 of the workgroups that hold cells)
__device__ unsigned long long g_dstamps[16];
extern "C" int rh_debug_dstamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dstamps), sizeof(g_dstamps)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dstamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// (intervals in registers, added to the totals once at the last stamp (8): an atomic per stamp sits in front of the loads it is timing)
#define RH_DSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dst_acc_[k] += t_ - t_prev_; t_prev_ = t_; \
        if ((k) == 8 && threadIdx.x == 0) { for (int k_ = 0; k_ < 9; k_++) atomicAdd(&g_dstamps[k_], dst_acc_[k_]); atomicAdd(&g_dstamps[15], 1ull); } } while (0)
#define RH_DSTAMP_BEGIN() unsigned long long dst_acc_[9] = {}; unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
#else
#define RH_DSTAMP(k) do { } while (0)
#define RH_DSTAMP_BEGIN() do { } while (0)
#endif

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void dxl_strip8(DxLinBatch B, const DxLinModel* __restrict__ L, int step)
{
    constexpr int KD = 8, GS = 58, PAD = 8, CS = 80;
    __shared__ double seg[8][4][104];        // staged rows; afterwards: partial sums [8 wavefronts][8 diagonals][64]
    __shared__ double srow[2][KD][CS];        // the strip's own rows: raw, decorated; lane l at index l+PAD
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const bool outside = blockIdx.z != 0;
    const int smax = L1 + L2;
    const int dir = outside ? 1 : -1, fwd = -dir;
    const int sdA = outside ? smax - KD * step : 2 + KD * step;
    if (outside ? sdA < 2 : sdA > smax) return;                 // none of the eight diagonals exists
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // own lanes: inside (sources at smaller columns) lanes 6..63, outside lanes 0..57
    const int a0 = outside ? (int)blockIdx.x * GS : (int)blockIdx.x * GS - 6;
    const int a = a0 + lane;
    const bool own = outside ? (lane < GS || a0 + GS > B.n1max + 1) : (lane >= 6 || blockIdx.x == 0);
    const int lda = B.lda;
    const size_t ts = B.tab_stride;
    double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const double* __restrict__ rawt = tab + (outside ? DL_OUT : DL_IN) * ts + kDxPad;
    const double* __restrict__ dect = tab + (outside ? DL_OUTX : DL_INX) * ts + kDxPad;

    {   // does this group hold a cell [max(1, sd-L2), min(L1, sd-1)] of any of the eight diagonals?  If not: clear its columns and leave
        bool has = false;
#pragma unroll
        for (int k = 0; k < KD; k++) {
            const int sd = sdA + k * fwd;
            has = has || (sd >= 2 && sd <= smax && !(a0 + 63 < (sd - L2 > 1 ? sd - L2 : 1) || a0 > (sd - 1 < L1 ? sd - 1 : L1)));
        }
        if (!has) {
            if (own && a >= 0 && a <= B.n1max + 1) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int r = w + 8 * q, tbl = r >> 3, sd = sdA + (r & 7) * fwd;
                    if (sd >= 2 && sd <= smax) tab[(tbl ? (outside ? DL_OUTX : DL_INX) : (outside ? DL_OUT : DL_IN)) * ts + (size_t)sd * lda + kDxPad + a] = 0.0;
                }
            }
            return;
        }
    }
    RH_DSTAMP_BEGIN();
    // ---- the 30 window rows: loads first, all of them in flight at once (unconditional, from clamped addresses: a load behind a
    // branch is waited for behind that branch, which made these eight one round trip each), then the cell operands, whose chain of
    // dependent loads (letters -> pair type -> table values) travels behind them; zeros where the padded row has no column
    const int c0 = outside ? a0 + 1 : a0 - 36;
    const int cl = c0 + lane, ch = c0 + 64 + (lane < 40 ? lane : 39);      // 104 columns: 64 + 40
    const int cmax = B.n1max + 1 + kDxPad;
    const bool okl = cl >= -kDxPad && cl <= cmax, okh = ch >= -kDxPad && ch <= cmax;
    const int cls = cl < -kDxPad ? -kDxPad : (cl > cmax ? cmax : cl), chs = ch < -kDxPad ? -kDxPad : (ch > cmax ? cmax : ch);
    double vl[4], vh[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int r = 1 + w + 8 * q;
        const int row = outside ? sdA + r : sdA - r;
        const int rowc = row < 2 ? 2 : (row > smax ? smax : row);
        vl[q] = dect[(unsigned)(rowc * lda + cls)];     // (32-bit offsets from a uniform base: one address register per load;
        vh[q] = dect[(unsigned)(rowc * lda + chs)];     //  rowc >= 2 and cls >= -kDxPad: never negative)
    }
    // ---- operands of X_w that come from rows of earlier launches (rows 2, 3, 4 before it while those lie outside the strip)
    const int sdw = sdA + w * fwd;
    const bool mine = sdw >= 2 && sdw <= smax;
    bool pairable = false;
    const DxCellRaw craw = dx_cell_loads(lda, rawt, dect, s1, s2, outside, sdw, a, L1, L2, smax, (w < 2 ? 1 : 0) | (w < 3 ? 2 : 0) | (w < 4 ? 4 : 0));
    __builtin_amdgcn_sched_barrier(0);
    RH_DSTAMP(0);   // loads issued
#pragma unroll
    for (int q = 0; q < 4; q++) { asm volatile("" : "+v"(vl[q])); asm volatile("" : "+v"(vh[q])); }   // (keeps the loads where they were issued)
    RH_DSTAMP(1);   // window rows arrived
#pragma unroll
    for (int q = 0; q < 4; q++) {   // the window rows were requested first and arrive first: into LDS while the cell's own loads travel
        const int r = 1 + w + 8 * q;
        const int row = outside ? sdA + r : sdA - r;
        const bool rv = r <= 30 && row >= 2 && row <= smax;               // wave-uniform; rows that do not exist are never read
        seg[w][q][lane] = rv && okl ? vl[q] : 0.0;
        if (lane < 40) seg[w][q][64 + lane] = rv && okh ? vh[q] : 0.0;
    }
    __builtin_amdgcn_sched_barrier(0);
    RH_DSTAMP(2);   // staged
    DxCellOps o = dx_cell_weights(L, craw, outside, &pairable);
    RH_DSTAMP(3);   // cell operands arrived, weights computed
    for (int k = threadIdx.x; k < 2 * KD * CS; k += 512) (&srow[0][0][0])[k] = 0.0;
    double acc[KD];
#pragma unroll
    for (int k = 0; k < KD; k++) acc[k] = 0.0;
    switch (w) {   // (each wavefront reads only the rows it staged itself: no barrier needed before the pass)
#define X(V) case V: win_pass8<V>(&seg[w][0][lane], L->lam_pow, outside, sdA, smax, acc); break;
        X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
    }
    RH_DSTAMP(4);   // window pass
    __syncthreads();
    RH_DSTAMP(5);   // barrier
    // partial sums meet over the dead staging area, [8 wavefronts][4 diagonals][64] at a time: diagonals 0..3 (read by their owners,
    // wavefronts 0..3), then diagonals 4..7
    double* const part = &seg[0][0][0];
    // ---- chain: slot s finishes X_2s and X_2s+1 (wavefronts 2s, 2s+1)
    double g = 0.0;
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
        for (int k = 0; k < 4; k++) part[(w * 4 + k) * 64 + lane] = acc[half * 4 + k];
        lds_barrier_dx();
        if ((w >> 2) == half) {
#pragma unroll
            for (int q = 0; q < 8; q++) g += part[(q * 4 + (w & 3)) * 64 + lane];
        }
        lds_barrier_dx();
    }
    RH_DSTAMP(6);   // partial sums exchanged (two halves, four barriers)
    auto finish = [&](int k) {
        // own-row sources: row k' <= k-2 at columns a + dir*(1..t+1), t = k-2-k'
        double v = 0.0, vx = 0.0;
        if (mine && pairable) {
            const int li = PAD + lane;
            if (k >= 2) o.o_st = srow[0][k - 2][li + dir];
            if (k >= 3) { o.o_01 = srow[1][k - 3][li + dir]; o.o_10 = srow[1][k - 3][li + 2 * dir]; }
            if (k >= 4) { o.o_02 = srow[1][k - 4][li + dir]; o.o_11 = srow[1][k - 4][li + 2 * dir]; o.o_20 = srow[1][k - 4][li + 3 * dir]; }
            for (int kp = 0; kp + 5 <= k; kp++) {
                const int t = k - 2 - kp;                          // 3 <= t <= 5
                double run = 0.0;
                for (int l = 1; l <= t + 1; l++) run += srow[1][kp][li + dir * l];
                g = fma(L->lam_pow[t + 2], run, g);
            }
            dx_cell_value(L, o, outside, B.pw8[k], g, &v, &vx);
        }
        srow[0][k][PAD + lane] = v;
        srow[1][k][PAD + lane] = vx;
    };
#pragma unroll
    for (int sl = 0; sl < 4; sl++) {
        if ((w >> 1) == sl) finish(w);
        lds_barrier_dx();
    }
    RH_DSTAMP(7);   // chain (four slots)
    // ---- the 2 x 8 rows go to HBM: row (table, k) by wavefront; every column of the row is rewritten
    if (own && a >= 0 && a <= B.n1max + 1) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int r = w + 8 * q, tbl = r >> 3, k = r & 7, sd = sdA + k * fwd;
            if (sd >= 2 && sd <= smax)
                tab[(tbl ? (outside ? DL_OUTX : DL_INX) : (outside ? DL_OUT : DL_IN)) * ts + (size_t)sd * lda + kDxPad + a] = srow[tbl][k][PAD + lane];
        }
    }
    RH_DSTAMP(8);   // stores issued
}

// Z~ = sum IN~[a,b] * close~(a,b)                                                (DuplexEngine.ipp:1066-1073)
// two stages, both in a fixed summation order (results do not depend on scheduling): kLzRows anti-diagonals per
// workgroup, read along the rows of the table (coalesced), then one thread per pair adds the chunks in order.
// zpart: [NP][nchunk] partial sums, cpart: pairable-cell counts
constexpr int kLzRows = 16;
__global__ __launch_bounds__(256) void dxl_logz_part(DxLinBatch B, const DxLinModel* __restrict__ L, double* __restrict__ zpart, int* __restrict__ cpart, int nchunk)
{
    __shared__ double sm[4];
    __shared__ int sc[4];
    const int pr = blockIdx.y, chunk = blockIdx.x;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const double* __restrict__ in = B.tab + (size_t)pr * B.pair_stride + DL_IN * B.tab_stride + kDxPad;
    double acc = 0.0;
    int npair = 0;
    for (int sd = 2 + chunk * kLzRows; sd < 2 + (chunk + 1) * kLzRows && sd <= L1 + L2; sd++) {
        // close~ = (lam*e^eu)^(L1+L2-sd) * lam^2 * dangles * helix_closing
        const double rowf = pow(L->lam_eu, (double)(L1 + L2 - sd)) * L->lam_pow[2];
        const int alo = sd - L2 > 1 ? sd - L2 : 1, ahi = sd - 1 < L1 ? sd - 1 : L1;
        for (int a = alo + threadIdx.x; a <= ahi; a += 256) {
            const int i = a, j = L2 + 1 - (sd - a);
            const int x = s1[i], y = s2[j];
            if (!pairs(x, y)) continue;
            npair++;
            const double cl = rowf * L->E_dl[x * 25 + y * 5 + s1[i + 1]] * L->E_dr[x * 25 + y * 5 + s2[j - 1]] * L->E_hc[x * 5 + y];
            acc = fma(in[(size_t)sd * B.lda + a], cl, acc);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); npair += __shfl_xor(npair, o, 64); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = acc; sc[threadIdx.x >> 6] = npair; }
    __syncthreads();
    if (threadIdx.x == 0) {
        zpart[(size_t)pr * nchunk + chunk] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        cpart[(size_t)pr * nchunk + chunk] = sc[0] + sc[1] + sc[2] + sc[3];
    }
}
__global__ void dxl_logz_final(DxLinBatch B, const DxLinModel* __restrict__ L, const double* __restrict__ zpart, const int* __restrict__ cpart, int nchunk,
                               double* __restrict__ zbar, double* __restrict__ logz, int* __restrict__ bad)
{
    const int pr = blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= B.np) return;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    double z = 0.0;
    int any = 0;
    for (int k = 0; k < nchunk; k++) { z += zpart[(size_t)pr * nchunk + k]; any += cpart[(size_t)pr * nchunk + k]; }
    zbar[pr] = z;
    // no complementary pair at all: the reference leaves logZ at its -2e20 sentinel and hp at zero
    if (!any) { logz[pr] = RH_NEG_INF; bad[pr] = 0; return; }
    bad[pr] = (z > 1e-200 && z < 1e200) ? 0 : 1;
    logz[pr] = log(z) + L->s * (double)(L1 + L2 + 2);
}

// hp[i][j] = IN~ * OUT~ / Z~                                                      (DuplexEngine.ipp:1146-1169)
// 32x32 tiles of (anti-diagonal sd, column a): read along the table rows, transposed through LDS, written along the rows
// of hp (for a fixed i = a the 32 anti-diagonals of a tile are 32 consecutive j)
__global__ __launch_bounds__(256) void dxl_posterior(DxLinBatch B, const double* __restrict__ zbar, int* __restrict__ bad)
{
    __shared__ double tile[32][33];
    const int pr = blockIdx.z;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const int a0 = 1 + blockIdx.x * 32, s0 = 2 + blockIdx.y * 32;
    if (a0 > L1 || s0 > L1 + L2) return;
    const double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride + kDxPad;
    const double z = zbar[pr];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    bool flag = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sd = s0 + ty + 8 * k, a = a0 + tx, b = sd - a;
        double p = 0.0;
        if (a <= L1 && b >= 1 && b <= L2 && z > 0.0) {
            const size_t at = (size_t)sd * B.lda + a;
            p = tab[DL_IN * B.tab_stride + at] * tab[DL_OUT * B.tab_stride + at] / z;
            if (!(p == p) || p > 1e300) { flag = true; p = 0.0; }
        }
        tile[ty + 8 * k][tx] = p;
    }
    if (flag) atomicOr(&bad[pr], 1);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int a = a0 + ty + 8 * k, sd = s0 + tx, b = sd - a;
        if (a <= L1 && b >= 1 && b <= L2) B.hp[(size_t)pr * B.hp_stride + (size_t)a * B.ldd + (L2 + 1 - b)] = tile[tx][ty + 8 * k];
    }
}

template __global__ void dxl_sweep<2>(DxLinBatch, const DxLinModel*, int, int);
template __global__ void dxl_sweep<4>(DxLinBatch, const DxLinModel*, int, int);
template __global__ void dxl_sweep<8>(DxLinBatch, const DxLinModel*, int, int);

}  // namespace rh
