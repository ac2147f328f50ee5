// duplex_vlin.hip -- pf_duplex (BL* tables, ViennaRNA-1.8 loop energies; PARITY UNPINNED, see vienna_model.h) in SCALED LINEAR
// space.  Recurrences: /root/reference/src/pf_duplex.c:304-345 (fw), 347-394 (bk, pulled), 264-277 (pr_duplex), loop budget
// MAXLOOP = 30 -- the arithmetic of duplex_vienna.hip, organised as duplex_lin.hip's dxl_sweep4:
//   * coordinates a = i, b = L2+1-j; anti-diagonal-major tables [sd*lda + kDxPad + a], sd = a+b, zero pad columns;
//     IN~ = IN*lam^(a+b), OUT~ = OUT*lam^((L1+1-a)+(L2+1-b)); six tables: raw, x mismatchI (operand of generic interior loops),
//     x TerminalAU (operand of long bulges), for each direction;
//   * four anti-diagonals X_k = A + k*fwd per launch and direction.  Source row r = A + dir*(2+t) holds, for X_k, the loops of
//     total length tw = t+k: generic interior loops = a (tw+1)-tap filter over the LDS-staged row with the length- and
//     asymmetry-dependent weights shape_w[tw][l1] (wave-uniform), long bulges = two taps over the TerminalAU copy; every staged
//     element is read once and multiplied into all four cells of the lane (X_3's length-2 bulges sit one row nearer: epilogue);
//   * the seven tabulated shapes (stack, 1-bulges, 1x1, 1x2, 2x1, 2x2) are gathered per cell in the epilogue; X_2 / X_3 take the
//     ones that lie on X_0 / X_1 (stack; stack + 1-bulges) from LDS, where wavefronts 0 / 1 leave their rows.
// A pair whose scaled partition function leaves the double range is flagged and recomputed by the log-space kernels.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "vienna_model.h"

namespace rh {

enum DxvLinTable { VD_IN = 0, VD_INX, VD_OUT, VD_OUTX, VD_INT, VD_OUTT, VD_COUNT };   // IN / OUT at DL_IN / DL_OUT: dxl_posterior reads them

namespace {
typedef const volatile __attribute__((address_space(3))) double* lds_vp;

__device__ __forceinline__ double small_wd(const VLinModel* L, int l1, int l2, int t1, int t2, int si1, int sj1, int sp1, int sq1)
{   // as small_w of mccaskill_vlin.hip: t1 = pair closing the loop seen from outside, t2 = the other pair (its rtype is taken here)
    const int r2 = vienna_rtype(t2);
    const int tt = t1 * 8 + r2;
    if (l1 == 0 && l2 == 0) return L->E_stack[tt];
    if (l1 + l2 == 1) return L->E_bulge1[tt];
    if (l1 == 1 && l2 == 1) return L->E_int11[tt * 25 + si1 * 5 + sj1];
    if (l1 == 1 && l2 == 2) return L->E_int21[tt * 125 + (si1 * 5 + sq1) * 5 + sj1];
    if (l1 == 2 && l2 == 1) return L->E_int21[(r2 * 8 + t1) * 125 + (sq1 * 5 + si1) * 5 + sp1];
    return L->E_int22[tt * 625 + ((si1 * 5 + sp1) * 5 + sq1) * 5 + sj1];
}

// all staged rows of wavefront WV: t = WV, WV+4, ... <= 30 (row A + dir*(2+t)); lengths and weight offsets are compile-time.
// seg0[q*96 + k]: inside column a-4-t+skip+k with skip = max(0, t-27) (X_k reads tap l1 at a-1-l1), outside column a+1+k
// (tap l1 at a+1+l1).  accg[k]: generic loops of X_k (x shape_w), accb[k]: long bulges (x WB), taken from the TerminalAU rows
template <int WV>
__device__ __forceinline__ void vwin_pass4(const double* seg0, const VLinModel* __restrict__ L, const double* __restrict__ taurows, int lda,
                                           bool outside, int sdA, int smax, int a, double accg[4], double accb[4])
{
    if constexpr (WV < 4) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int t = WV + 4 * q;        // compile-time after unrolling
            if (t > kMaxSingle) continue;
            const int row = outside ? sdA + 2 + t : sdA - 2 - t;
            if (row >= 2 && row <= smax) {
                lds_vp vs = (lds_vp)(seg0 + q * 96);
                const int skip = t > 27 ? t - 27 : 0;
                const int dir = outside ? 1 : -1;
                const double* __restrict__ trow = taurows + (size_t)row * lda + a;
                const double near = trow[dir];                                   // bulge with l1 = 0
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int tw = t + k;
                    if (tw < 2 || tw > kMaxSingle) continue;
                    accb[k] = fma(L->WB[tw], near + trow[dir * (1 + tw)], accb[k]);   // ... and with l1 = tw
                    if (tw < 4) continue;                                        // generic loops: l1, l2 >= 1, t >= 4
                    const double* __restrict__ wt = L->shape_w + tw * (tw + 1) / 2;
                    double g0 = 0.0, g1 = 0.0;
#pragma unroll
                    for (int l1 = 1; l1 < tw; l1++) {                            // l1 = 0 and l1 = tw are the bulges
                        const double x = outside ? vs[l1] : vs[3 - skip + t - l1];
                        if (l1 & 1) g1 = fma(wt[l1], x, g1); else g0 = fma(wt[l1], x, g0);
                    }
                    accg[k] += g0 + g1;
                }
            }
        }
    }
}
}  // namespace

__global__ __launch_bounds__(256) void dxvl_sweep4(DxLinBatch B, const VLinModel* __restrict__ L, const VDxLin* __restrict__ D, int step)
{
    __shared__ double buf[4][8][96];
    __shared__ double part[8][4][64];
    __shared__ double hand[2][64];      // raw X_0, raw X_1 of this group's columns
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const bool outside = blockIdx.z != 0;
    const int grp = blockIdx.x;
    const int smax = L1 + L2;
    const int dir = outside ? 1 : -1, fwd = -dir;               // X_k = A + k*fwd; sources at rows sd + dir*(2+t), columns a + dir*(1+l1)
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
    const bool owns23 = outside ? (lane <= 61 || a0 + 62 > B.n1max + 1) : (lane >= 2 || grp == 0);
    bool has = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int sd = sdA + k * fwd;
        has = has || (sd >= 2 && sd <= smax && !(a0 + 63 < (sd - L2 > 1 ? sd - L2 : 1) || a0 > (sd - 1 < L1 ? sd - 1 : L1)));
    }
    const int sdw = sdA + w * fwd;      // the diagonal wavefront w finishes
    const bool mine = sdw >= 2 && sdw <= smax;
    const size_t at = (size_t)sdw * lda + kDxPad + a;
    const int T_RAW = outside ? VD_OUT : VD_IN, T_MM = outside ? VD_OUTX : VD_INX, T_TAU = outside ? VD_OUTT : VD_INT;
    if (!has) {   // only clear the columns of the rows
        if (mine && a <= B.n1max + 1 && (w < 2 || owns23)) { tab[T_RAW * ts + at] = 0.0; tab[T_MM * ts + at] = 0.0; tab[T_TAU * ts + at] = 0.0; }
        return;
    }

    // ---- loops of length >= 2 (all four wavefronts): stage the mismatch-decorated rows t = w, w+4, ..., then filter
    const double* __restrict__ src = tab + T_MM * ts + kDxPad;
    {   // all sixteen loads first, unconditional (a row that does not exist reads row 2 and is never used): behind `if (row exists)` each
        // was waited for inside its branch, one memory round trip per row
        double v0[8], v1[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int t = w + 4 * q;
            const int row = outside ? sdA + 2 + t : sdA - 2 - t;
            const bool on = t <= kMaxSingle && row >= 2 && row <= smax;   // wave-uniform
            const int skip = t > 27 ? t - 27 : 0;
            const int c0 = outside ? a0 + 1 : a0 - 4 - t + skip;           // >= -kDxPad - 3: inside the padded row or the end of the one before
            const double* __restrict__ r = src + (size_t)(on ? row : 2) * lda + c0;
            v0[q] = r[lane]; v1[q] = r[64 + (lane & 31)];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) { asm volatile("" : "+v"(v0[q])); asm volatile("" : "+v"(v1[q])); }
#pragma unroll
        for (int q = 0; q < 8; q++) {
            buf[w][q][lane] = v0[q];
            if (lane < 32) buf[w][q][64 + lane] = v1[q];
        }
    }
    double accg[4] = {0.0, 0.0, 0.0, 0.0}, accb[4] = {0.0, 0.0, 0.0, 0.0};
    switch (w) {
#define X(V) case V: vwin_pass4<V>(&buf[w][0][lane], L, tab + T_TAU * ts + kDxPad, lda, outside, sdA, smax, a, accg, accb); break;
        X(0) X(1) X(2) X(3)
#undef X
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { part[k][w][lane] = accg[k]; part[4 + k][w][lane] = accb[k]; }

    // ---- epilogue of X_w: letters, pair type, the tabulated shapes whose rows are final (before the barrier)
    const int b = sdw - a, i = a, j = L2 + 1 - b;
    const bool incell = mine && a >= 1 && a <= L1 && b >= 1 && b <= L2;
    // letters i-3 .. i+3 of s1 and j-3 .. j+3 of s2 (clamped into the padded sequences): everything the seven tabulated shapes index
    // with, requested at once; pair types are arithmetic, every table value is loaded unconditionally and selected afterwards
    int c1[7], c2[7];
    {
        const int ic = incell ? i : 1, jc = incell ? j : 1;
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int p1 = ic + u - 3, p2 = jc + u - 3;
            c1[u] = s1[p1 < 0 ? 0 : (p1 > L1 + 1 ? L1 + 1 : p1)];
            c2[u] = s2[p2 < 0 ? 0 : (p2 > L2 + 1 ? L2 + 1 : p2)];
        }
    }
    const int x = c1[3], xm = c1[2], xp = c1[4], y = c2[3], ym = c2[2], yp = c2[4];
    const int type = incell ? vienna_ptype(x, y) : 0;
    const bool pairable = type != 0;
    const double* __restrict__ rawt = tab + T_RAW * ts + kDxPad;
    double sm7 = 0.0, e_tau = 1.0, mm_up = 0.0, mm_dn = 0.0, ends = 0.0;
    double c_st = 0.0, c_b01 = 0.0, c_b10 = 0.0;   // weights of the shapes of X_2 / X_3 whose source row belongs to this launch
    {
        const int rt = vienna_rtype(type);
        const double l_tau = L->E_tau[type], l_up = D->E_mmI[type * 25 + xp * 5 + ym], l_dn = D->E_mmI[rt * 25 + yp * 5 + xm];
        const double l_d5i = D->E_d5[type * 5 + xm], l_d3i = D->E_d3[type * 5 + yp], l_d3o = D->E_d3[rt * 5 + xp], l_d5o = D->E_d5[rt * 5 + ym];
        double sv[7], sw[7];
        bool sok[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int l1 = (k == 2 || k == 3 || k == 4) ? 1 : (k >= 5 ? 2 : 0);
            const int l2 = (k == 1 || k == 3 || k == 5) ? 1 : ((k == 4 || k == 6) ? 2 : 0);
            // k: 0 stack, 1 (0,1), 2 (1,0), 3 (1,1), 4 (1,2), 5 (2,1), 6 (2,2)
            const int srow = sdw + dir * (2 + l1 + l2), scol = a + dir * (1 + l1);
            // source cell letters: inside (i-1-l1, j+1+l2), outside (i+1+l1, j-1-l2)
            const int si = i + dir * (1 + l1), sj = j - dir * (1 + l2);
            // held letters: s1[i+o] = c1[3+o], s2[j+o] = c2[3+o]
            const int o1 = dir * (1 + l1), o2 = -dir * (1 + l2);
            const int ts_ = vienna_ptype(c1[3 + o1], c2[3 + o2]);
            const int n1 = c1[3 + o1 - dir], n2 = c2[3 + o2 + dir];   // the letters next to the source pair inside the loop
            sok[k] = pairable & (srow >= 2) & (srow <= smax) & (si >= 1) & (si <= L1) & (sj >= 1) & (sj <= L2) & (ts_ != 0);
            sw[k] = outside ? small_wd(L, l1, l2, type, ts_, xp, ym, n1, n2) : small_wd(L, l1, l2, ts_, type, n1, n2, xm, yp);
            const bool fresh = (w == 2 && k == 0) || (w == 3 && k <= 2);   // row of this launch: value arrives through `hand`
            const int rc = (srow >= 2 && srow <= smax) ? srow : 2, cc = scol < -kDxPad ? -kDxPad : (scol > B.n1max + 1 + kDxPad ? B.n1max + 1 + kDxPad : scol);
            sv[k] = fresh ? 0.0 : rawt[(size_t)rc * lda + cc];
        }
        if (pairable) {
            e_tau = l_tau; mm_up = l_up; mm_dn = l_dn;
            if (!outside) ends = D->E_init * (i > 1 ? l_d5i : 1.0) * (j < L2 ? l_d3i : 1.0) * e_tau;   // pf_duplex.c:321-326
            else ends = (i < L1 ? l_d3o : 1.0) * (j > 1 ? l_d5o : 1.0) * e_tau;                         // pf_duplex.c:361-365
        }
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const bool fresh = (w == 2 && k == 0) || (w == 3 && k <= 2);
            if (!fresh) sm7 = fma(sok[k] ? sv[k] : 0.0, sok[k] ? sw[k] : 0.0, sm7);
            else if (k == 0) c_st = sok[k] ? sw[k] : 0.0;
            else if (k == 1) c_b01 = sok[k] ? sw[k] : 0.0;
            else c_b10 = sok[k] ? sw[k] : 0.0;
        }
    }
    // X_3's bulges of length 2 lie on row A + dir (window index t = -1: no other diagonal uses it, so it is not staged)
    double b2 = 0.0;
    if (w == 3 && pairable) {
        const int r = sdw + 4 * dir;
        if (r >= 2 && r <= smax) {
            const double* __restrict__ trow = tab + T_TAU * ts + kDxPad + (size_t)r * lda + a;
            b2 = L->WB[2] * (trow[dir] + trow[3 * dir]);
        }
    }
    __syncthreads();
    double g = 0.0, gb = b2;
#pragma unroll
    for (int k = 0; k < 4; k++) { g += part[w][k][lane]; gb += part[4 + w][k][lane]; }
    const double pw = B.pw4[w];         // lam^(sd) inside / lam^(L1+L2+2-sd) outside: the scale of the open / close term
    double v = 0.0;
    if (w < 2) {
        if (pairable) v = pw * ends + (outside ? mm_up : mm_dn) * g + e_tau * gb + sm7;
        hand[w][lane] = v;
    }
    __syncthreads();
    if (w >= 2) {
        if (!mine || !owns23) return;
        if (pairable) {
            const int l1 = lane + dir, l2 = lane + 2 * dir;
            const bool in1 = l1 >= 0 && l1 < 64, in2 = l2 >= 0 && l2 < 64;
            sm7 = fma(in1 ? hand[w - 2][l1] : 0.0, c_st, sm7);                  // X_2 stacks on X_0, X_3 on X_1
            if (w == 3) {
                sm7 = fma(in1 ? hand[0][l1] : 0.0, c_b01, sm7);                // 1-bulges of X_3 close on X_0
                sm7 = fma(in2 ? hand[0][l2] : 0.0, c_b10, sm7);
            }
            v = pw * ends + (outside ? mm_up : mm_dn) * g + e_tau * gb + sm7;
        }
    } else if (!mine) return;
    if (a <= B.n1max + 1) {  // every column of the row is rewritten: stale values of other shapes never survive
        tab[T_RAW * ts + at] = v;
        tab[T_MM * ts + at] = v * (outside ? mm_dn : mm_up);   // decorated as the other end of a later generic loop
        tab[T_TAU * ts + at] = v * e_tau;
    }
}

// Z~ = sum IN~[a,b] * close(a,b) * lam^(L1+L2+2-a-b)   (pf_duplex.c:337-341); same two deterministic stages as dxl_logz_part
__global__ __launch_bounds__(256) void dxvl_logz_part(DxLinBatch B, const VLinModel* __restrict__ L, const VDxLin* __restrict__ D,
                                                     double* __restrict__ zpart, int* __restrict__ cpart, int nchunk)
{
    __shared__ double sm[4];
    __shared__ int sc[4];
    const int pr = blockIdx.y, chunk = blockIdx.x;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const double* __restrict__ in = B.tab + (size_t)pr * B.pair_stride + VD_IN * B.tab_stride + kDxPad;
    double acc = 0.0;
    int npair = 0;
    for (int sd = 2 + chunk * 16; sd < 2 + (chunk + 1) * 16 && sd <= L1 + L2; sd++) {
        const double rowf = pow(D->lam, (double)(L1 + L2 + 2 - sd));
        const int alo = sd - L2 > 1 ? sd - L2 : 1, ahi = sd - 1 < L1 ? sd - 1 : L1;
        for (int a = alo + threadIdx.x; a <= ahi; a += 256) {
            const int i = a, j = L2 + 1 - (sd - a);
            const int type = vienna_ptype(s1[i], s2[j]);
            if (!type) continue;
            npair++;
            const int rt = vienna_rtype(type);
            const double cl = rowf * (i < L1 ? D->E_d3[rt * 5 + s1[i + 1]] : 1.0) * (j > 1 ? D->E_d5[rt * 5 + s2[j - 1]] : 1.0) * L->E_tau[type];
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
__global__ void dxvl_logz_final(DxLinBatch B, double s, const double* __restrict__ zpart, const int* __restrict__ cpart, int nchunk,
                                double* __restrict__ zbar, double* __restrict__ logz, int* __restrict__ bad)
{
    const int pr = blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= B.np) return;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    double z = 0.0;
    int any = 0;
    for (int k = 0; k < nchunk; k++) { z += zpart[(size_t)pr * nchunk + k]; any += cpart[(size_t)pr * nchunk + k]; }
    zbar[pr] = z;
    if (!any) { logz[pr] = RH_NEG_INF; bad[pr] = 0; return; }   // no complementary pair at all: as the log-space kernels report it
    bad[pr] = (z > 1e-200 && z < 1e200) ? 0 : 1;
    logz[pr] = log(z) + s * (double)(L1 + L2 + 2);
}

}  // namespace rh
