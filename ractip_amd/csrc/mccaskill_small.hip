// mccaskill_small.hip -- McCaskill inside / F5 / outside / posterior of a SHORT sequence (kSmallMin <= n <= kSmallMax = 109) in ONE
// launch, one workgroup per sequence, every table the sums run over resident in the 160 KB of one CU (CONTRAfold model, scaled linear
// space: lin_model.h).  An ORGANISATION OF THE SAME ARITHMETIC as the sweep kernels, opt-in (RH_SMALL=1): at the lengths it covers it
// is not faster than the sweeps (measured below), so the default stays with them.
//
// Why it exists: BASELINE config 5 (1000 dinucleotide shuffles of OxyS / fhlA, 109 and 53 letters; reference loop
// src/ractip.cpp:1638-1657) is thousands of short folds; the sweep organisation launches ~50 kernels per batch whose 64-column groups
// are mostly empty lanes at these lengths.  Here:
//   inside  : FM, FM1, FCX resident (3 x n(n-1)/2 doubles); FC / FCA of the last three diagonals in a ring; FC and FCA also go to HBM
//             (FCA for the exterior sums, FC for the posterior)
//   F5      : FCA is read back into the space of FCX; one wavefront runs the F5i chain, another the F5o chain (neither depends on an
//             outside value: F5o[k] = F5o[k+1] w_eu + w_ep2 sum_jj F5o[jj] FCA[k+1, jj-1])
//   outside : FM stays; FM1 is needed as FM1[e][i-e] -- the cells that END at column i, the same set on every diagonal -- and moves
//             into registers (wavefront (group, q) of a column group holds e = 1+q, 1+q+WPG, ...); FM2o takes FM1's space; FCoX lives in a
//             ring of the 33 newest rows (zero-padded on both sides, so the enclosing-loop filter needs no mask); FMo / FM1o / FCo in
//             rings of three rows.
//   scores  : the 625-entry letter-indexed tables are compressed to the 150 entries a PAIR can index and live in LDS (a gather from
//             HBM kept the CU's address unit busy for 64 cycles per load and its latency exposed on every diagonal).
// Recurrences and arithmetic are those of lin_inside_diag<.,.,0> / lin_outside_diag (mccaskill_lin.hip; reference
// InferenceEngine.ipp:3356-3722, 3731-4080, 4498-4828), one diagonal per step, two LDS-only barriers per step: the eight wavefronts
// split the terms of a diagonal (term sets of the FM2 / FMo / FM1o sums, rows of the single-branch filter), the partial sums meet in
// LDS and the first wavefront of each 64-column group finishes the cells.
// Measured (MI355X, 1000 pairs of 109 + 53 letters): 6.1 ms against 4.9 ms for the sweeps (inside + outside); one 109-letter sequence
// occupies a CU for ~1.1 ms.  What was learned on the way (tools/sstamps.py, SQ counters): (1) with two wavefronts per SIMD a
// wavefront issues one instruction of any kind per four cycles -- masks, register rotation and address arithmetic cost as much as the
// FMAs (first version: 714 VALU + 725 SALU per 102 LDS reads per wavefront and step); (2) a global store is acknowledged ~7000 cycles
// after issue and __syncthreads() waits for it (vmcnt(0)); (3) pure loads carried across a step are recomputed by the optimizer where
// they are used unless pinned; (4) what remains is the LDS rate of the DENSE filter, one 512-byte read per FMA: ~1100 reads x 4 cycles
// per diagonal.  The strips win because their filter is factored (five times fewer operations) and their FM2 far terms run on MFMA;
// porting the factored filter here (fixed-pitch rows instead of the packed triangle) is the step that would make this the default.
// Which sequences come here is decided per sequence by its length alone (rh_api.hip), so a result does not depend on the batch.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lin_model.h"

namespace rh {

namespace {

constexpr uint32_t kPairMaskS = (1u << (0 * 5 + 3)) | (1u << (3 * 5 + 0)) | (1u << (1 * 5 + 2)) |
                                (1u << (2 * 5 + 1)) | (1u << (2 * 5 + 3)) | (1u << (3 * 5 + 2));
__device__ __forceinline__ bool pairs_sm(int a, int b) { return (kPairMaskS >> (a * 5 + b)) & 1u; }
__device__ __forceinline__ size_t tri_off_sm(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }
__device__ __forceinline__ double wsum_sm(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// (volatile: keeps the reads as ds_read_b64 -- merged into ds_read2_b64 they run at half the LDS rate on CDNA4, and the filters are bound by it)
typedef const volatile double __attribute__((address_space(3)))* lds_cp;
// Barrier between the steps of a sweep: what the wavefronts exchange is in LDS.  __syncthreads() also waits for every global store in
// flight (vmcnt(0)), and a store of a diagonal's FC / FCA / posterior row is acknowledged ~7000 cycles after it was issued -- that wait,
// twice per diagonal, was two thirds of this kernel.  The stores are ordered once, by the fence between the sweeps.
__device__ __forceinline__ void lds_barrier_sm() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The single-branch filters of one wavefront: rows t = q, q+WPG, ... <= tmax, row t = sum_{l=0..t} w(t, l) * seg_t[DIR * l], values in LDS,
// weights wave-uniform from the zero-padded table wpad[t][32] (scalar loads, sixteen dwords per block).  With two wavefronts per SIMD a
// wavefront issues one instruction of ANY kind per four cycles, so what counts is the number of instructions per tap: blocks of eight
// taps -- eight reads off one base address, one scalar load, eight FMAs with the weight as scalar operand, no masks (the padding is in
// the table; a tap past t reads a neighbouring, finite table entry: the tables are zero-filled before the sweeps) -- and the stream of
// blocks over all rows is software-pipelined: the next block is requested before the FMAs of the current one (past the last block the
// first one is requested again: no branch around a load).  Rolled: the unrolled switch over t of the sweep kernels is ~50 KB of code.
template <int DIR, class SegOf>
__device__ __forceinline__ double filt_stream(const double* __restrict__ wpad, int q, int WPG, int tmax, SegOf segof)
{
    if (q > tmax) return 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int t = q, l = 0;
    bool more = true;
    double xa[8], wa[8], xb[8], wb[8];
    auto load = [&](double (&x)[8], double (&w)[8]) {   // the block (t, l)
        const lds_cp seg = segof(t) + DIR * l;
        const double* __restrict__ wt = wpad + t * 32 + l;
#pragma unroll
        for (int k = 0; k < 8; k++) { x[k] = seg[DIR * k]; w[k] = wt[k]; }
    };
    auto advance = [&]() {   // the next block of the stream, or (past the end) the first one again
        l += 8;
        if (l > t) { t += WPG; l = 0; }
        more = t <= tmax;
        if (!more) { t = q; l = 0; }
    };
    auto fmas = [&](const double (&x)[8], const double (&w)[8]) {
        s0 = fma(w[0], x[0], s0); s1 = fma(w[1], x[1], s1); s2 = fma(w[2], x[2], s2); s3 = fma(w[3], x[3], s3);
        s0 = fma(w[4], x[4], s0); s1 = fma(w[5], x[5], s1); s2 = fma(w[6], x[6], s2); s3 = fma(w[7], x[7], s3);
    };
    load(xa, wa);
    for (;;) {   // two blocks per pass, in two register sets: no copies
        advance();
        const bool more_b = more;
        load(xb, wb);
        fmas(xa, wa);
        if (!more_b) break;
        advance();
        const bool more_a = more;
        load(xa, wa);
        fmas(xb, wb);
        if (!more_a) break;
    }
    return (s0 + s1) + (s2 + s3);
}

// LDS plan (doubles)
struct SmallPlan {
    static constexpr int NMAX = kSmallMax;
    static constexpr int T = (NMAX - 1) * NMAX / 2;   // cells of a triangle: row d = 0 .. n-2 holds the cells i = 1 .. n-1-d
    static constexpr int OFF_A = 0;                   // FM
    static constexpr int OFF_B = T;                   // FM1 (inside), FM2o (outside)
    static constexpr int OFF_C = 2 * T;               // FCX (inside), FCA (F5 chains), ring of FCoX + the three-row rings (outside)
    static constexpr int OFF_PART = 3 * T;            // inside: [8][2][64] partial sums + FC3 / FCA3 rings; outside: [8][3][64]
    static constexpr int RWS = 112;                   // width of a three-row ring: columns 0 .. 111
    static constexpr int PART_SZ = 8 * 2 * 64 + 2 * 3 * RWS;
    static constexpr int OFF_F5I = OFF_PART + PART_SZ;   // [128]
    static constexpr int OFF_F5O = OFF_F5I + 128;        // [128]
    static constexpr int OFF_SEQ = OFF_F5O + 128;        // n+2 letters (bytes)
    static constexpr int OFF_HPW = OFF_SEQ + 16;         // hairpin weight of span d: lam^d * E_hairpin[min(d, 30)]  [112]
    // the letter-indexed score tables, compressed to what a PAIR can index: [6 pair types][5][5] entries of TJB, TJA, TST, then E_bp[25],
    // E_11[25], E_b01[8], E_b10[8], then the pair type of two letters (25 bytes).  A gather from the 625-entry tables in HBM kept the
    // address unit busy for 64 cycles per load and its latency exposed at every diagonal
    static constexpr int OFF_CT = OFF_HPW + 112;
    static constexpr int CT_TJB = 0, CT_TJA = 150, CT_TST = 300, CT_BP = 450, CT_11 = 475, CT_B01 = 500, CT_B10 = 508, CT_PT = 516;
    static constexpr int SZ = OFF_CT + 520;
    // outside: ring of the 33 newest FCoX rows, 32 zero columns in front of column 0
    static constexpr int PADL = 32, RW = PADL + 112, NR = 33;
    static constexpr int OFF_RING = OFF_C;
    static constexpr int OFF_R3 = OFF_C + NR * RW;    // FMo3, FM1o3, FCo3: [3][3][RWS]
    static_assert(PART_SZ >= 8 * 3 * 64, "outside partial sums");
    static_assert(NR * RW + 9 * RWS <= T, "outside rings do not fit the FCX region");
    static_assert(SZ * 8 <= 160 * 1024, "LDS");
    static_assert(NMAX <= RWS - 1 && NMAX + 2 <= 128, "ring widths");
};

}  // namespace

#ifdef RH_SMALL_STAMPS
// tuning build only (tools/build_variant.py sstamps -DRH_SMALL_STAMPS): cycle totals per phase, wavefront 0 of every workgroup
__device__ unsigned long long g_sstamps[16];
extern "C" int rh_debug_sstamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sstamps), sizeof(g_sstamps)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sstamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// (totals are kept in registers and added to the global counters once, at the end: an atomic per stamp from 256 workgroups at a time
//  queues up in the memory pipeline and shows up as time in whatever phase issues the next store)
#define RH_SSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc_[k] += t_ - t_prev_; t_prev_ = t_; } while (0)
#define RH_SSTAMP_BEGIN() unsigned long long st_acc_[12] = {}; unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
#define RH_SSTAMP_END() do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 12; k_++) atomicAdd(&g_sstamps[k_], st_acc_[k_]); atomicAdd(&g_sstamps[15], 1ull); } } while (0)
#else
#define RH_SSTAMP(k) do { } while (0)
#define RH_SSTAMP_BEGIN() do { } while (0)
#define RH_SSTAMP_END() do { } while (0)
#endif

// list[k]: sequence index of workgroup k
__global__ __launch_bounds__(512) void lin_small_fold(McBatch B, const LinModel* __restrict__ L, const double* __restrict__ wpad, const int* __restrict__ list, int* __restrict__ bad)
{
    using P = SmallPlan;
    __shared__ double lds[P::SZ];
    const int sq = list[blockIdx.x];
    const int n = B.n[sq];
    if (n < kSmallMin || n > kSmallMax) return;   // (host routing guarantees the range; nothing is computed otherwise)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ sg = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    enum { S_FC = 0, S_FCA = 2 };   // table slots shared with mccaskill_lin.hip (L_FC, L_FCA)
    double* const FM = lds + P::OFF_A;
    double* const FM1 = lds + P::OFF_B;
    double* const FCX = lds + P::OFF_C;
    double* const part = lds + P::OFF_PART;
    double* const FC3 = part + 8 * 2 * 64;
    double* const FCA3 = FC3 + 3 * P::RWS;
    double* const f5i = lds + P::OFF_F5I;
    double* const f5o = lds + P::OFF_F5O;
    uint8_t* const s = (uint8_t*)(lds + P::OFF_SEQ);
    // column groups: two of 64 columns when a diagonal can hold more than 64 cells, else one; fixed for the whole kernel (the
    // outside phase keeps per-column operands in registers)
    const bool two = n - 1 > 64;
    const int WPG = two ? 4 : 8;                       // wavefronts (term sets) per column group
    const int grp = two ? w >> 2 : 0, q = two ? w & 3 : w;
    const int i = 1 + grp * 64 + lane;                 // this lane's column
    auto rowoff = [&](int d) { return d * (n - 1) - d * (d - 1) / 2 - 1; };   // cell (i, d) lives at rowoff(d) + i

    RH_SSTAMP_BEGIN();
    for (int k = threadIdx.x; k < n + 2; k += 512) s[k] = sg[k];
    for (int k = threadIdx.x; k < 6 * P::RWS; k += 512) FC3[k] = 0.0;   // (FC3 and FCA3 are adjacent)
    double* const HPW = lds + P::OFF_HPW;
    for (int k = threadIdx.x; k < 3 * P::T; k += 512) lds[k] = 0.0;   // (masked taps and terms read entries that are not written yet)
    for (int k = threadIdx.x; k < 112; k += 512) HPW[k] = k >= 3 ? exp(-L->s * (double)k) * L->E_hairpin[k < 30 ? k : 30] : 0.0;   // ScoreHairpin (ipp:2123-2152)
    double* const CT = lds + P::OFF_CT;
    uint8_t* const PT = (uint8_t*)(CT + P::CT_PT);
    for (int k = threadIdx.x; k < 150; k += 512) {
        const int ty = k / 25, b = (k % 25) / 5, dd = k % 5;
        const int a = ty == 0 ? 0 : ty == 1 ? 3 : ty == 2 ? 1 : ty == 3 ? 2 : ty == 4 ? 2 : 3;   // AU UA CG GC GU UG
        const int c = ty == 0 ? 3 : ty == 1 ? 0 : ty == 2 ? 2 : ty == 3 ? 1 : ty == 4 ? 3 : 2;
        const int src = 25 * (5 * a + b) + 5 * c + dd;
        CT[P::CT_TJB + k] = L->TJB[src]; CT[P::CT_TJA + k] = L->TJA[src]; CT[P::CT_TST + k] = L->TST[src];
    }
    for (int k = threadIdx.x; k < 25; k += 512) {
        CT[P::CT_BP + k] = L->E_bp[k]; CT[P::CT_11 + k] = L->E_11[k];
        const int a = k / 5, c = k % 5;
        PT[k] = (a == 0 && c == 3) ? 0 : (a == 3 && c == 0) ? 1 : (a == 1 && c == 2) ? 2 : (a == 2 && c == 1) ? 3 : (a == 2 && c == 3) ? 4 : (a == 3 && c == 2) ? 5 : 0;
    }
    for (int k = threadIdx.x; k < 8; k += 512) { CT[P::CT_B01 + k] = L->E_b01[k]; CT[P::CT_B10 + k] = L->E_b10[k]; }
    // entry of a compressed table for the pair (a, c) with the neighbours (b, dd) = what the full table holds at 25*(5a+b) + 5c+dd
    auto cidx = [&](int a, int b, int c, int dd) { return (int)PT[a * 5 + c] * 25 + b * 5 + dd; };
    const double w01 = L->w01, w10 = L->w10, w11 = L->w11, lam2 = L->lam2, w_mu = L->w_mu, w_mp2 = L->w_mp2, w_ep2 = L->w_ep2, e_mpmb = L->e_mpmb;
    __syncthreads();

    // ================================================================== inside, diagonals 0 .. n-2
    for (int d = 0; d <= n - 2; d++) {
        const int ncell = n - 1 - d;
        const bool valid = i <= ncell;
        const int ic = valid ? i : ncell;              // (lanes past the diagonal read the last cell's operands and are dropped)
        const bool fin = q == 0 && (grp == 0 || ncell > 64);   // this wavefront finishes its group's cells
        double acc2 = 0.0, accc = 0.0;
        if (grp == 0 || ncell > 64) {
            // FM2[i,d] = sum_{m=1}^{d-1} FM1[m][i] * FM[d-m][i+m]          (ipp:3384-3411)
            double a0 = 0.0, a1 = 0.0;
            {
                // FM1[rowoff(m) + ic] * FM[rowoff(d-m) + ic + m], m = 1+q, 1+q+WPG, ...: the two offsets advance by first differences that
                // advance by a constant (rowoff is quadratic) -- two scalar adds per term instead of two multiplications
                int m = 1 + q;
                int o1 = rowoff(m), o2 = rowoff(d - m) + m;
                int d1 = rowoff(m + WPG) - o1, d2 = rowoff(d - m - WPG) + WPG - rowoff(d - m);
                const int dd = -WPG * WPG;   // second difference of rowoff over steps of WPG (both directions)
                for (; m + WPG <= d - 1; m += 2 * WPG) {
                    const double x0 = FM1[o1 + ic], y0 = FM[o2 + ic];
                    o1 += d1; o2 += d2; d1 += dd; d2 += dd;
                    const double x1 = FM1[o1 + ic], y1 = FM[o2 + ic];
                    o1 += d1; o2 += d2; d1 += dd; d2 += dd;
                    a0 = fma(x0, y0, a0); a1 = fma(x1, y1, a1);
                }
                if (m <= d - 1) a0 = fma(FM1[o1 + ic], FM[o2 + ic], a0);
            }
            acc2 = a0 + a1;
            // generic single-branch shapes: sum_t sum_l1 w(l1,t-l1) * FCX[d-2-t][i+1+l1]   (ipp:3597-3619)
            const int tmax = d - 2 < kMaxSingle ? d - 2 : kMaxSingle;
            accc = filt_stream<1>(wpad, q, WPG, tmax, [&](int t) { return (lds_cp)(FCX + rowoff(d - 2 - t) + ic + 1); });
        }
        RH_SSTAMP(0);   // inside: term loops
        part[(w * 2 + 0) * 64 + lane] = acc2;
        part[(w * 2 + 1) * 64 + lane] = accc;
        lds_barrier_sm();
        RH_SSTAMP(1);   // inside: barrier
        if (fin) {
            const int j = ic + d;
            const int s_im1 = s[ic - 1], s_i = s[ic], s_ip1 = s[ic + 1], s_j = s[j], s_jp1 = s[j + 1], s_jp2 = s[j + 2];
            const bool pairable = valid && pairs_sm(s_i, s_jp1);
            const int idx = cidx(s_i, s_ip1, s_jp1, s_j);       // (i,j)   as enclosing pair
            const int idd = cidx(s_jp1, s_jp2, s_i, s_im1);     // (j+1,i-1) as enclosed pair
            const double e_tjb = CT[P::CT_TJB + idx], e_tja = CT[P::CT_TJA + idx], e_tst = CT[P::CT_TST + idx], e_bp = CT[P::CT_BP + s_i * 5 + s_jp1];
            const double e_tjbd = CT[P::CT_TJB + idd], e_tjad = CT[P::CT_TJA + idd];
            const double e_n01 = CT[P::CT_B01 + s_j], e_n10 = CT[P::CT_B10 + s_ip1], e_n11 = CT[P::CT_11 + s_ip1 * 5 + s_j];
            double fm2 = 0.0, g = 0.0;
            for (int k = 0; k < WPG; k++) { fm2 += part[((grp * 4 + k) * 2 + 0) * 64 + lane]; g += part[((grp * 4 + k) * 2 + 1) * 64 + lane]; }
            double o_x01 = 0, o_x10 = 0, o_x11 = 0, o_fc = 0, o_fca = 0, o_fm1 = 0, o_fm = 0;
            if (d >= 3) {
                o_x01 = FCX[rowoff(d - 3) + ic + 1];
                o_x10 = FCX[rowoff(d - 3) + ic + 2];
                if (d >= 4) o_x11 = FCX[rowoff(d - 4) + ic + 2];
            }
            if (d >= 2) {
                o_fc = FC3[((d - 2) % 3) * P::RWS + ic + 1];
                o_fca = FCA3[((d - 2) % 3) * P::RWS + ic + 1];
                o_fm1 = FM1[rowoff(d - 1) + ic + 1];
                o_fm = FM[rowoff(d - 1) + ic];
            }
            asm volatile("" :: "v"(o_fm), "v"(o_x01), "v"(fm2), "v"(g));
            RH_SSTAMP(9);   // epilogue: operands read
            double fc = 0.0;
            if (pairable) {
                const double sp = w01 * e_n01 * o_x01 + w10 * e_n10 * o_x10 + w11 * e_n11 * o_x11;
                const double hp = HPW[d];
                const double st = o_fc * lam2 * e_tst;                                     // stacking pair (ipp:3595)
                fc = e_tjb * (g + sp + hp) + st + fm2 * e_tja * e_mpmb;                    // ipp:3573-3622
            }
            double fm1v = 0.0, fmv = 0.0;
            if (d >= 2) {                                                                  // ipp:3641-3688
                fm1v = o_fca * w_mp2 + o_fm1 * w_mu;
                fmv = fm2 + o_fm * w_mu + fm1v;
            }
            asm volatile("" :: "v"(fc), "v"(fmv));
            RH_SSTAMP(10);   // epilogue: arithmetic
            if (valid) {
                const double fca = fc * e_bp * e_tjad;
                FM[rowoff(d) + i] = fmv;
                FM1[rowoff(d) + i] = fm1v;
                FCX[rowoff(d) + i] = fc * e_bp * e_tjbd;
                FC3[(d % 3) * P::RWS + i] = fc;
                FCA3[(d % 3) * P::RWS + i] = fca;
                RH_SSTAMP(11);   // epilogue: LDS stores
                tab[S_FC * ts + (size_t)d * ld + i] = fc;
                tab[S_FCA * ts + (size_t)d * ld + i] = fca;
            }
        }
        RH_SSTAMP(2);   // inside: epilogue
        lds_barrier_sm();
        RH_SSTAMP(3);   // inside: second barrier
    }

    // ================================================================== F5i, F5o: FCA back from HBM into the space of FCX
    __threadfence();
    __syncthreads();
    {
        double* const FCA = FCX;
        for (int d = 0; d <= n - 2; d++) {
            const int ncell = n - 1 - d;
            for (int c = 1 + (int)threadIdx.x; c <= ncell; c += 512) FCA[rowoff(d) + c] = tab[S_FCA * ts + (size_t)d * ld + c];
        }
        __syncthreads();
        if (w == 0) {
            // F5i[jj] = F5i[jj-1]*w_eu + w_ep2 * sum_{k<=jj-2} F5i[k]*FCA[jj-2-k][k+1]   (ipp:3692-3717); lane l holds F5i[l], F5i[l+64]
            double v0 = lane == 0 ? 1.0 : 0.0, v1 = 0.0, prev = 1.0;
            for (int jj = 1; jj <= n; jj++) {
                double acc = 0.0;
                { const int k = lane; if (k <= jj - 2) acc = v0 * FCA[rowoff(jj - 2 - k) + k + 1]; }
                { const int k = lane + 64; if (k <= jj - 2) acc = fma(v1, FCA[rowoff(jj - 2 - k) + k + 1], acc); }
                acc = wsum_sm(acc);
                const double val = prev * L->w_eu + acc * L->w_ep2;
                if (jj == lane) v0 = val;
                if (jj == lane + 64) v1 = val;
                prev = val;
            }
            f5i[lane] = v0; f5i[lane + 64] = v1;
            if (lane <= n) B.f5i[(size_t)sq * ld + lane] = v0;
            if (lane + 64 <= n) B.f5i[(size_t)sq * ld + lane + 64] = v1;
        } else if (w == 1) {
            // F5o[k] = F5o[k+1]*w_eu + w_ep2 * sum_{jj>=k+2} F5o[jj]*FCA[jj-2-k][k+1]      (ipp:3751-3780, pulled); F5o[n] = 1
            double v0 = lane == n ? 1.0 : 0.0, v1 = lane + 64 == n ? 1.0 : 0.0, next = 1.0;
            for (int k = n - 1; k >= 1; k--) {
                double acc = 0.0;
                { const int jj = lane; if (jj >= k + 2 && jj <= n) acc = v0 * FCA[rowoff(jj - 2 - k) + k + 1]; }
                { const int jj = lane + 64; if (jj >= k + 2 && jj <= n) acc = fma(v1, FCA[rowoff(jj - 2 - k) + k + 1], acc); }
                acc = wsum_sm(acc);
                const double val = next * L->w_eu + acc * L->w_ep2;
                if (k == lane) v0 = val;
                if (k == lane + 64) v1 = val;
                next = val;
            }
            f5o[lane] = v0; f5o[lane + 64] = v1;
            if (lane >= 1 && lane <= n) B.f5o[(size_t)sq * ld + lane] = v0;
            if (lane + 64 <= n) B.f5o[(size_t)sq * ld + lane + 64] = v1;
        }
    }
    __syncthreads();
    RH_SSTAMP(4);   // F5 chains

    // ================================================================== outside + posterior, diagonals n-2 .. 0
    // FM1[e][i-e], e = 1+q, 1+q+WPG, ... (the cells that end at this lane's column) move into registers; then FM2o takes FM1's space
    constexpr int NE = (kSmallMax + 3) / 4;
    double f1e[NE];
#pragma unroll
    for (int r = 0; r < NE; r++) {
        const int e = 1 + q + WPG * r;
        const bool ok = e <= i - 1 && i <= n - 1;      // cell (i-e, e): letters i-e .. i
        f1e[r] = ok ? FM1[rowoff(e) + i - e] : 0.0;
    }
    double* const FM2O = FM1;
    double* const RING = lds + P::OFF_RING;
    double* const FMO3 = lds + P::OFF_R3;
    double* const FM1O3 = FMO3 + 3 * P::RWS;
    double* const FCO3 = FM1O3 + 3 * P::RWS;
    __syncthreads();   // every wavefront has its FM1 values (and the F5 chains are done with FCA)
    for (int k = threadIdx.x; k < P::NR * P::RW + 9 * P::RWS; k += 512) RING[k] = 0.0;
    __syncthreads();
    const double Z = f5i[n];
    // FC of the cell (posterior) is the one operand that comes from HBM: requested one step ahead by the finishing wavefronts
    auto fc_of = [&](int d, bool fin) {
        double v = 0.0;
        if (fin && d >= 0) { const int ncell = n - 1 - d; v = tab[S_FC * ts + (size_t)d * ld + (i <= ncell ? i : ncell)]; }
        return v;
    };
    double fc_nx = fc_of(n - 2, q == 0 && grp == 0);
    for (int d = n - 2; d >= 0; d--) {
        const int ncell = n - 1 - d;
        const bool valid = i <= ncell;
        const int ic = valid ? i : ncell;
        const int j = ic + d;
        const bool active = grp == 0 || ncell > 64;
        const bool fin = q == 0 && active;
        const bool guard_m = d >= 2;
        const double o_fc = fc_nx;
        fc_nx = fc_of(d - 1, q == 0 && (grp == 0 || ncell + 1 > 64));
        double accm = 0.0, acc1 = 0.0, accc = 0.0;
        if (active) {
            if (guard_m) {
                // FMo[i,d] += FM2o[d+e][i-e] * FM1[e][i-e], e = 1..i-1                    (ipp:4046-4064, pulled)
                const int i_last = ncell < grp * 64 + 64 ? ncell : grp * 64 + 64;
#pragma unroll
                for (int r = 0; r < NE; r++) {
                    const int e = 1 + q + WPG * r;
                    // (wave-uniform guard; a lane whose column has no such term holds f1e = 0 and reads a finite neighbouring entry)
                    if (e <= i_last - 1) accm = fma(FM2O[rowoff(d + e) + i - e], f1e[r], accm);
                }
                // FM1o[i,d] += FM2o[d+e][i] * FM[e][i+d], e = 1..n-1-j
                const int emax = n - 1 - (1 + grp * 64 + d);   // the group's first column reaches furthest
                double b0 = 0.0, b1 = 0.0;
                for (int e = 1 + q; e <= emax; e += 4 * WPG) {
                    double x[4], y[4];
                    bool ok[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int ee = e + k * WPG;
                        ok[k] = valid && ee <= n - 1 - j;
                        x[k] = FM2O[ok[k] ? rowoff(d + ee) + i : 0]; y[k] = FM[ok[k] ? rowoff(ee) + j : 0];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (k & 1) b1 = fma(ok[k] ? x[k] : 0.0, ok[k] ? y[k] : 0.0, b1); else b0 = fma(ok[k] ? x[k] : 0.0, ok[k] ? y[k] : 0.0, b0);
                    }
                }
                acc1 = b0 + b1;
            }
            // enclosing single-branch loops: sum_t sum_l1 w(l1,t-l1) * FCoX[d+2+t][i-1-l1]   (ipp:4004-4024, pulled); the ring rows are
            // zero outside [1, n-1-row]
            const int room = n - 4 - d;
            if (room >= 0) {
                const int tmax = room < kMaxSingle ? room : kMaxSingle;
                // (row t: sum_l1 w[l1] * FCoX[d+2+t][i-1-l1])
                accc = filt_stream<-1>(wpad, q, WPG, tmax, [&](int t) { return (lds_cp)(RING + ((d + 2 + t) % P::NR) * P::RW + P::PADL + ic - 1); });
            }
        }
        asm volatile("" : "+v"(fc_nx));   // (pinned: left alone the optimizer reloads it where it is used, one step later)
        RH_SSTAMP(5);   // outside: term loops
        part[(w * 3 + 0) * 64 + lane] = accm;
        part[(w * 3 + 1) * 64 + lane] = acc1;
        part[(w * 3 + 2) * 64 + lane] = accc;
        lds_barrier_sm();
        RH_SSTAMP(6);   // outside: barrier
        if (fin) {
            const int s_im1 = s[ic - 1], s_i = s[ic], s_ip1 = s[ic + 1], s_j = s[j], s_jp1 = s[j + 1], s_jp2 = s[j + 2];
            const bool pairable = valid && pairs_sm(s_i, s_jp1);
            const int idx = cidx(s_i, s_ip1, s_jp1, s_j);
            const int idd = cidx(s_jp1, s_jp2, s_i, s_im1);
            const double e_tjb = CT[P::CT_TJB + idx], e_tja = CT[P::CT_TJA + idx], e_bp = CT[P::CT_BP + s_i * 5 + s_jp1];
            const double e_tjbd = CT[P::CT_TJB + idd], e_tjad = CT[P::CT_TJA + idd];
            const double e_tst = CT[P::CT_TST + cidx(s_im1, s_i, s_jp2, s_jp1)];   // (of the pair (i-1, j+2): multiplies its FCo, 0 if it is no pair)
            const double e_n01 = CT[P::CT_B01 + s_jp2], e_n10 = CT[P::CT_B10 + s_im1], e_n11 = CT[P::CT_11 + s_im1 * 5 + s_jp2];
            double sm = 0.0, s1 = 0.0, g = 0.0;
            for (int k = 0; k < WPG; k++) {
                sm += part[((grp * 4 + k) * 3 + 0) * 64 + lane]; s1 += part[((grp * 4 + k) * 3 + 1) * 64 + lane]; g += part[((grp * 4 + k) * 3 + 2) * 64 + lane];
            }
            const int r1 = ((d + 1) % 3) * P::RWS, r2 = ((d + 2) % 3) * P::RWS;
            const double o_fmo = FMO3[r1 + ic], o_fm1o = FM1O3[r1 + ic - 1];                // (zero past the end of a row and in column 0)
            const double o_fm1o_up = FM1O3[r2 + ic - 1], o_fco_up = FCO3[r2 + ic - 1];
            const double* const row3 = RING + ((d + 3) % P::NR) * P::RW + P::PADL;
            const double* const row4 = RING + ((d + 4) % P::NR) * P::RW + P::PADL;
            const double o_x01 = row3[ic - 1], o_x10 = row3[ic - 2], o_x11 = row4[ic - 2];
            const double o_f5o = f5o[j + 1], o_f5i = f5i[ic - 1];
            double fmo = 0.0, fm1o = 0.0;
            if (guard_m) {
                fmo = sm + o_fmo * w_mu;                      // ipp:3806
                fm1o = s1 + fmo + o_fm1o * w_mu;              // ipp:3809, 3833
            }
            double fco = 0.0;
            if (pairable) {
                const double ext = o_f5o * o_f5i * w_ep2;     // exterior loop, ipp:3768-3776
                const double multi = o_fm1o_up * w_mp2;       // branch of a multiloop, ipp:3828
                const double sp = w01 * e_n01 * o_x01 + w10 * e_n10 * o_x10 + w11 * e_n11 * o_x11;
                const double st = o_fco_up * lam2 * e_tst;    // stacked on (i-1,j+1)
                fco = e_bp * (e_tjad * (ext + multi) + e_tjbd * (g + sp)) + st;
            }
            const double fm2o = fmo + fco * e_tja * e_mpmb;                                                   // ipp:3803, 4027
            if (valid) {
                FM2O[rowoff(d) + i] = fm2o;
                RING[(d % P::NR) * P::RW + P::PADL + i] = fco * e_tjb;
                FMO3[(d % 3) * P::RWS + i] = fmo;
                FM1O3[(d % 3) * P::RWS + i] = fm1o;
                FCO3[(d % 3) * P::RWS + i] = fco;
                // posterior of pair (i, j+1) = FCo * FCi / Z, clipped to [0,1]                 (ipp:4689-4827)
                double p = fco * o_fc / Z;
                if (!(p == p) || p > 1e300) { atomicOr(&bad[sq], 1); p = 0.0; }
                p = p > 1.0 ? 1.0 : (p < 0.0 ? 0.0 : p);
                B.bp[(size_t)sq * B.tri_stride + tri_off_sm(n, i) + (j + 1)] = p;
            }
        }
        RH_SSTAMP(7);   // outside: epilogue
        lds_barrier_sm();
        RH_SSTAMP(8);   // outside: second barrier
    }
    RH_SSTAMP_END();
}

// wpad: the single-branch weights as zero-padded rows [31][32] (row t: w(l1, t - l1), l1 = 0 .. t)
void launch_lin_small(const McBatch& B, const LinModel* L, const double* wpad, const int* list, int nlist, int* bad, hipStream_t stream)
{
    hipLaunchKernelGGL(lin_small_fold, dim3(nlist), dim3(512), 0, stream, B, L, wpad, list, bad);
}

}  // namespace rh
