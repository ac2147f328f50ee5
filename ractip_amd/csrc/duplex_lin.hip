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

namespace rh {

namespace {
constexpr uint32_t kPairMaskD = (1u << (0 * 5 + 3)) | (1u << (3 * 5 + 0)) | (1u << (1 * 5 + 2)) |
                                (1u << (2 * 5 + 1)) | (1u << (2 * 5 + 3)) | (1u << (3 * 5 + 2));
__device__ __forceinline__ bool pairs(int a, int b) { return (kPairMaskD >> (a * 5 + b)) & 1u; }
}  // namespace

// ---------------------------------------------------------------------------------
// launch `step`: inside diagonals {2+2*step, 3+2*step} and outside diagonals
// {Smax-2*step-1, Smax-2*step} (Smax = L1+L2) -- blockIdx.z selects inside/outside,
// blockIdx.x = (which of the two diagonals) * groups + (64-cell group).
template <int W>
__global__ __launch_bounds__(64 * W) void dxl_sweep(DxLinBatch B, const DxLinModel* __restrict__ L, int step, int groups)
{
    __shared__ double buf[W][96];
    __shared__ double part[W][64];
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const bool outside = blockIdx.z != 0;
    const int which = blockIdx.x / groups, grp = blockIdx.x % groups;
    const int smax = L1 + L2;
    const int sd = outside ? smax - 2 * step - 1 + which : 2 + 2 * step + which;
    if (sd < 2 || sd > smax) return;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int a0 = grp * 64;
    if (a0 > B.n1max + 1) return;
    const int a = a0 + lane;
    const int b = sd - a;
    const int i = a, j = L2 + 1 - b;
    const bool incell = a >= 1 && a <= L1 && b >= 1 && b <= L2;
    const int lda = B.lda;
    const size_t ts = B.tab_stride;
    double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride;
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;

    int x = 4, xm = 4, xp = 4, y = 4, ym = 4, yp = 4;  // s1[i], s1[i-1], s1[i+1], s2[j], s2[j-1], s2[j+1]
    if (incell) { x = s1[i]; xm = s1[i - 1]; xp = s1[i + 1]; y = s2[j]; ym = s2[j - 1]; yp = s2[j + 1]; }
    const bool pairable = incell && pairs(x, y);

    // ---- windows t = 3..28: sum_{l1=0..t} SRC[sd -/+ (2+t)][a -/+ (1+l1)] * lam^(t+2)
    const double* __restrict__ src = tab + (outside ? DL_OUTX : DL_INX) * ts + kDxPad;
    double acc = 0.0;
    for (int t = 3 + w; t <= 28; t += W) {
        const int row = outside ? sd + 2 + t : sd - 2 - t;
        if (row < 2 || row > smax) continue;  // wave-uniform
        // segment start: inside a0-1-t (window = [a-1-t, a-1]); outside a0+1 (window = [a+1, a+1+t])
        const int c0 = outside ? a0 + 1 : a0 - 1 - t;
        const double* __restrict__ r = src + (size_t)row * lda + c0;
        buf[w][lane] = r[lane];
        if (lane < 32) buf[w][64 + lane] = r[64 + lane];
        double sum = 0.0;
        for (int k = 0; k <= t; k++) sum += buf[w][lane + k];
        acc = fma(L->lam_pow[t + 2], sum, acc);
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w != 0) return;
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < W; k++) g += part[k][lane];

    const size_t at = (size_t)sd * lda + kDxPad + a;
    double v = 0.0, vx = 0.0;
    if (pairable) {
        const int tm_up = ((x * 5 + y) * 5 + xp) * 5 + ym;    // terminal_mismatch[s1[i]][s2[j]][s1[i+1]][s2[j-1]]
        const int tm_dn = ((y * 5 + x) * 5 + yp) * 5 + xm;    // terminal_mismatch[s2[j]][s1[i]][s2[j+1]][s1[i-1]]
        const double e_up = L->E_tm[tm_up], e_dn = L->E_tm[tm_dn] * L->E_bp[x * 5 + y];
        const double l2 = L->lam_pow[2], l3 = L->lam_pow[3], l4 = L->lam_pow[4];
        if (!outside) {
            // inside[i][j] = open + stack + down * (specials + windows)          (DuplexEngine.ipp:1029-1064)
            const double* __restrict__ inx = tab + DL_INX * ts + kDxPad;
            const double* __restrict__ in = tab + DL_IN * ts + kDxPad;
            const double open = B.pw_in[which] * L->E_dr[y * 25 + x * 5 + xm] * L->E_dl[y * 25 + x * 5 + yp] *
                                L->E_bp[y * 5 + x] * L->E_hc[y * 5 + x];
            const double st = sd >= 4 ? in[(size_t)(sd - 2) * lda + a - 1] * l2 * L->E_bp[x * 5 + y] *
                                            L->E_hs[((xm * 5 + yp) * 5 + x) * 5 + y] : 0.0;
            double sp = 0.0;
            if (sd >= 5) sp += l3 * (L->E_b01[yp] * inx[(size_t)(sd - 3) * lda + a - 1] + L->E_b10[xm] * inx[(size_t)(sd - 3) * lda + a - 2]);
            if (sd >= 6) sp += l4 * (inx[(size_t)(sd - 4) * lda + a - 1] + L->E_11[xm * 5 + yp] * inx[(size_t)(sd - 4) * lda + a - 2] +
                                     inx[(size_t)(sd - 4) * lda + a - 3]);
            v = open + st + e_dn * (sp + g);
            vx = v * e_up;      // as the upstream pair of a later loop
        } else {
            // outside[p][q] = close + stack + up * (specials + windows)           (DuplexEngine.ipp:1094-1129, pulled)
            const double* __restrict__ outx = tab + DL_OUTX * ts + kDxPad;
            const double* __restrict__ out = tab + DL_OUT * ts + kDxPad;
            const double close = B.pw_out[which] * L->E_dl[x * 25 + y * 5 + xp] * L->E_dr[x * 25 + y * 5 + ym] * L->E_hc[x * 5 + y];
            // (i+1, j-1) stacked on this pair: base_pair[s1[i+1]][s2[j-1]] * helix_stacking[s1[i]][s2[j]][s1[i+1]][s2[j-1]]
            const double st = sd + 2 <= smax ? out[(size_t)(sd + 2) * lda + a + 1] * l2 * L->E_bp[xp * 5 + ym] *
                                                   L->E_hs[((x * 5 + y) * 5 + xp) * 5 + ym] : 0.0;
            double sp = 0.0;
            if (sd + 3 <= smax) sp += l3 * (L->E_b01[ym] * outx[(size_t)(sd + 3) * lda + a + 1] + L->E_b10[xp] * outx[(size_t)(sd + 3) * lda + a + 2]);
            if (sd + 4 <= smax) sp += l4 * (outx[(size_t)(sd + 4) * lda + a + 1] + L->E_11[xp * 5 + ym] * outx[(size_t)(sd + 4) * lda + a + 2] +
                                            outx[(size_t)(sd + 4) * lda + a + 3]);
            v = close + st + e_up * (sp + g);
            vx = v * e_dn;      // as the downstream pair of an earlier loop
        }
    }
    if (a <= B.n1max + 1) {  // every column of the row is rewritten: stale values of other shapes never survive
        tab[(outside ? DL_OUT : DL_IN) * ts + at] = v;
        tab[(outside ? DL_OUTX : DL_INX) * ts + at] = vx;
    }
}

// Z~ = sum IN~[a,b] * close~(a,b); one workgroup per pair                        (DuplexEngine.ipp:1066-1073)
__global__ __launch_bounds__(1024) void dxl_logz(DxLinBatch B, const DxLinModel* __restrict__ L, double* __restrict__ zbar,
                                                 double* __restrict__ logz, int* __restrict__ bad)
{
    __shared__ double sm[16];
    __shared__ int sc[16];
    const int pr = blockIdx.x;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const uint8_t* __restrict__ s1 = B.seq + (size_t)(2 * pr) * B.lds;
    const uint8_t* __restrict__ s2 = B.seq + (size_t)(2 * pr + 1) * B.lds;
    const double* __restrict__ in = B.tab + (size_t)pr * B.pair_stride + DL_IN * B.tab_stride + kDxPad;
    double acc = 0.0;
    int npair = 0;
    const int total = L1 * L2;
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
        const int i = c / L2 + 1, j = c % L2 + 1;
        const int a = i, b = L2 + 1 - j;
        const int x = s1[i], y = s2[j];
        if (!pairs(x, y)) continue;
        npair++;
        const double v = in[(size_t)(a + b) * B.lda + a];
        // close~ = (lam*e^eu)^(L1+L2-sd) * lam^2 * dangles * helix_closing
        const double cl = pow(L->lam_eu, (double)(L1 + L2 - a - b)) * L->lam_pow[2] * L->E_dl[x * 25 + y * 5 + s1[i + 1]] *
                          L->E_dr[x * 25 + y * 5 + s2[j - 1]] * L->E_hc[x * 5 + y];
        acc = fma(v, cl, acc);
    }
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o, 64); npair += __shfl_xor(npair, o, 64); }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = acc; sc[threadIdx.x >> 6] = npair; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double z = 0.0;
        int any = 0;
        for (int k = 0; k < 16; k++) { z += sm[k]; any += sc[k]; }
        zbar[pr] = z;
        // no complementary pair at all: the reference leaves logZ at its -2e20 sentinel and hp at zero
        if (!any) { logz[pr] = RH_NEG_INF; bad[pr] = 0; return; }
        bad[pr] = (z > 1e-280 && z < 1e280) ? 0 : 1;
        logz[pr] = log(z) + L->s * (double)(L1 + L2 + 2);
    }
}

// hp[i][j] = IN~ * OUT~ / Z~                                                      (DuplexEngine.ipp:1146-1169)
__global__ __launch_bounds__(256) void dxl_posterior(DxLinBatch B, const double* __restrict__ zbar, int* __restrict__ bad)
{
    const int pr = blockIdx.y;
    const int L1 = B.n[2 * pr], L2 = B.n[2 * pr + 1];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L1 * L2) return;
    const int i = c / L2 + 1, j = c % L2 + 1;
    const int a = i, b = L2 + 1 - j;
    const double* __restrict__ tab = B.tab + (size_t)pr * B.pair_stride + kDxPad;
    const size_t at = (size_t)(a + b) * B.lda + a;
    const double z = zbar[pr];
    double p = z > 0.0 ? tab[DL_IN * B.tab_stride + at] * tab[DL_OUT * B.tab_stride + at] / z : 0.0;
    if (!(p == p) || p > 1e300) { atomicOr(&bad[pr], 1); p = 0.0; }
    B.hp[(size_t)pr * B.hp_stride + (size_t)i * B.ldd + j] = p;
}

template __global__ void dxl_sweep<2>(DxLinBatch, const DxLinModel*, int, int);
template __global__ void dxl_sweep<4>(DxLinBatch, const DxLinModel*, int, int);
template __global__ void dxl_sweep<8>(DxLinBatch, const DxLinModel*, int, int);

}  // namespace rh
