// mccaskill_strip.hip -- McCaskill inside / outside sweeps in SCALED LINEAR space, KD diagonals per launch.
//
// Same recurrences and the same diagonal-major tables as mccaskill_lin.hip (reference:
// /root/reference/src/contrafold/InferenceEngine.ipp:3356-3722 inside, 3731-4080 outside, 4498-4828 posterior).
// What changes is the unit of work.  The per-diagonal kernels re-read, for every diagonal, the ~32 most recent rows of
// FM / FM1 / FCX (their L2 hit rate is 25 %: profiles/r02_a_baseline_pmc_l2.txt) and move ~1.3 KB per cell.  Here one
// workgroup owns a STRIP: 64 columns x KD consecutive diagonals d0 .. d0+KD-1.
//   * Every operand row that is final before the launch is staged ONCE in LDS (about 75 KB per workgroup, two
//     workgroups per CU) and serves all KD diagonals: HBM traffic per cell falls by about (32+2)/2 : (32+KD)/KD.
//   * pre-phase (all W wavefronts): the terms whose operands are final, for all KD diagonals at once.  The wavefronts
//     split the TERMS, not the diagonals: a wavefront keeps FM1[m][i] of "its" m in a register and walks the KD
//     diagonals over the staged FM rows (one LDS read per FMA); the single-branch filter reads each staged FCX value
//     once and feeds up to KD diagonals from it, with wave-uniform weights from a zero-padded transposed table
//     (no masks, no per-lane branches).
//   * chain (KD short steps, one wavefront each, a workgroup barrier in between): adds the few terms that touch
//     rows d0 .. d-1 of this strip (kept in LDS), runs the cell epilogue and stores the row.
//   * cell (i, d0+k) needs columns i .. i+k of the strip's own rows, so a group's lanes shrink by one per diagonal
//     (a trapezoid): groups advance by 64-(KD-1) columns and the overlap is recomputed.
//
// Banded near/far split of the O(n^3) terms (replaces the block-aligned split of mccaskill_lin.hip for these kernels):
//   FM2[i,j] = sum_{m=1}^{31} + sum_{e=1}^{31} (m = d-e)        near: streamed here, the same for every lane
//            + FM2F[i,j] = sum_{k=i+32}^{j-32} FM1[i,k] FM[k,j]  far : block products (mccaskill_far.hip), d >= 64
// The far range is not block-aligned; the two partial blocks K = I+2 and K = J-2 of a 16x16 tile product are exactly
// the tiles of block diagonal 2, each masked upper-triangular (second index >= first), so lin_pack_tiles packs those
// tiles masked (`banded`) and the product kernels are unchanged.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "batch.h"
#include "lin_model.h"

namespace rh {

namespace {

constexpr uint32_t kPairMaskS = (1u << (0 * 5 + 3)) | (1u << (3 * 5 + 0)) | (1u << (1 * 5 + 2)) |
                                (1u << (2 * 5 + 1)) | (1u << (2 * 5 + 3)) | (1u << (3 * 5 + 2));
__device__ __forceinline__ bool pairs_s(int a, int b) { return (kPairMaskS >> (a * 5 + b)) & 1u; }
__device__ __forceinline__ size_t tri_off_s(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }

__device__ __forceinline__ double wsum_s(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int V> using IC = std::integral_constant<int, V>;
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wavefront's outstanding GLOBAL stores and
// loads (s_waitcnt vmcnt(0)): inside the chain that would expose one HBM round trip per diagonal.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// volatile LDS pointer: keeps every read a plain ds_read_b64 (256 B/clk/CU); merged into ds_read2_b64 two reads cost 8 cycles on CDNA4
typedef const volatile __attribute__((address_space(3))) double* lds_vp;

}  // namespace

#ifdef RH_STAMPS
// tuning build only (tools/build_variant.py stamps -DRH_STAMPS): per-phase cycle totals of the strip kernels, summed over workgroups
__device__ unsigned long long g_stamps[16];
// The intervals are kept in registers and added to the global totals ONCE, at the last stamp (8): an atomic (or a store) issued in
// front of the staging loads sits in the same in-order memory queue, and the wait for those loads then includes its round trip to a
// contended address -- the first form of these macros inflated the staging phase it was meant to measure (found on mccaskill_small.hip,
// where one atomic per stamp and diagonal doubled the kernel's duration).
#define RH_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc_[k] += t_ - t_prev_; t_prev_ = t_; \
        if ((k) == 8 && threadIdx.x == 0) { for (int k_ = 0; k_ < 15; k_++) if (st_acc_[k_]) atomicAdd(&g_stamps[k_], st_acc_[k_]); atomicAdd(&g_stamps[15], 1ull); } } while (0)
#define RH_STAMP_BEGIN() unsigned long long st_acc_[15] = {}; unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
extern "C" int rh_debug_stamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// -DRH_STAMPS=3: per-workgroup timeline of ONE inside launch (d0 == 256): {XCC id << 32 | HW id, realtime at start, after the staging
// barrier, after the pre-phase barrier, at the end} per workgroup, s_memrealtime (100 MHz, one clock for the whole device)
__device__ unsigned long long g_trace[5 * 16384];
extern "C" int rh_debug_trace(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(g_trace)) == hipSuccess ? 0 : -1;
}
#if RH_STAMPS == 3
#define RH_TRACE(slot) do { if (threadIdx.x == 0 && d0 == 256) { const unsigned lin_ = blockIdx.x + blockIdx.y * gridDim.x; if (lin_ < 16384) g_trace[5 * lin_ + (slot)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define RH_TRACE_ID() do { if (threadIdx.x == 0 && d0 == 256) { const unsigned lin_ = blockIdx.x + blockIdx.y * gridDim.x; if (lin_ < 16384) g_trace[5 * lin_] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#endif
#define RH_NOSTAMP(k) do { } while (0)
#define RH_NOSTAMP_BEGIN() do { } while (0)
#if RH_STAMPS == 3
#define RH_STAMPO(k) RH_NOSTAMP(k)
#define RH_STAMPO_BEGIN() RH_NOSTAMP_BEGIN()
#define RH_STAMPI(k) RH_NOSTAMP(k)
#define RH_STAMPI_BEGIN() RH_NOSTAMP_BEGIN()
#elif RH_STAMPS == 2   // -DRH_STAMPS=2: the outside kernel's phases, -DRH_STAMPS=1: the inside kernel's
#define RH_STAMPO(k) RH_STAMP(k)
#define RH_STAMPO_BEGIN() RH_STAMP_BEGIN()
#define RH_STAMPI(k) RH_NOSTAMP(k)
#define RH_STAMPI_BEGIN() RH_NOSTAMP_BEGIN()
#else
#define RH_STAMPI(k) RH_STAMP(k)
#define RH_STAMPI_BEGIN() RH_STAMP_BEGIN()
#define RH_STAMPO(k) RH_NOSTAMP(k)
#define RH_STAMPO_BEGIN() RH_NOSTAMP_BEGIN()
#endif
#else
#define RH_STAMPI(k) do { } while (0)
#define RH_STAMPI_BEGIN() do { } while (0)
#define RH_STAMPO(k) do { } while (0)
#define RH_STAMPO_BEGIN() do { } while (0)
#endif
#ifndef RH_TRACE
#define RH_TRACE(slot) do { } while (0)
#define RH_TRACE_ID() do { } while (0)
#endif

enum StripTable { S_FC = 0, S_FCX, S_FCA, S_FM1, S_FM, S_FCO, S_FCOX, S_FM2O, S_FMO, S_FM1O, S_FM2F, S_FMOF, S_FM1OF };

// ---------------------------------------------------------------------------------------------------------------
// FACTORED single-branch filter (FILT = 1; host side: strip_weights in rh_api.hip).  The weight of an interior loop (l1, l2), t = l1+l2,
// is A(t) * B(|l1-l2|) up to a sparse residual (bulge ends, centre tap, a few shapes with l1, l2 <= 4: cache_score_single,
// InferenceEngine.ipp:1161-1197).  For one staged table row x (row rho of the strip: t = rho-1+k on diagonal k) the B-weighted sum
//   S_t[i] = sum_{l1=1}^{t-1} B(|2 l1 - t|) x[i+1+l1]      obeys      S_{t+2}[i-1] = S_t[i] + B(t) (x[i+1] + x[i+1+t]),
// and x[i+1], x[i+1+t] are also the two bulge taps of (t, i).  So a wavefront takes the diagonals of ONE parity (k = k0, k0+2, k0+4,
// k0+6), computes S once per row by a full pass for k0 and then walks: per row and diagonal 2 LDS reads, 1 add, 3-4 FMAs instead of
// t+1 FMAs.  The lane that holds column i at k0 holds column i-m at diagonal k0+2m (inside; i+m outside): S never moves between
// lanes, and the shift is undone for free by the index the partial sum is written to.  Lanes that drop off the group's edge hold
// garbage; they are exactly lanes the trapezoid does not store (m <= k).
// Wavefront w: rows rho = (w & 3) + 4 q, q = 0..7, parity k0 = w >> 2.  U = (w & 3) + k0 fixes every length (t0 = U-1+4q), so the
// pass is compiled for U = 0..4 with all lengths, table offsets and LDS offsets as immediates.
//   F[(t+1)*4 + {0,1,2,3}] = A(t), wb(t), Bstep(t), Rc(t);  F[160 + 16 p + j] = B(p + 2j);  F[192 + 4 t + l1-1] = explicit residuals
template <int U, bool OUTSIDE, int CE>
__device__ __forceinline__ void filt_factored(lds_vp xb, lds_vp fw, const double* __restrict__ F, double (&g)[4])
{
    // xb: inside  = &LE[(w & 3) * CE + lane - 3]: tap l1 of the cell in lane L at step m is xb[3 - m + l1]
    //     outside = &LE[(w & 3) * CE + lane]    : tap l1 is xb[31 + m - l1]      (every offset a non-negative immediate)
    // fw: W4 in LDS (A, wb, -, Rc per length): wave-uniform addresses, i.e. broadcast reads that return in order with the taps --
    //     scalar loads would share lgkmcnt with the LDS reads and force a full drain at every use (SMEM returns out of order)
    auto X = [&](int row_off, int m, int l1) -> double { return OUTSIDE ? xb[row_off + 31 + m - l1] : xb[row_off + 3 - m + l1]; };
    constexpr int par = (U + 1) & 1;   // parity of every length this wavefront meets: t0 = U-1+4q
    double bw[15];                     // B(par + 2j): the only scalar operands of the pass, requested once
#pragma unroll
    for (int j = 0; j < 15; j++) bw[j] = F[160 + 16 * par + j];
    static_for<8>([&](auto QC) {
        constexpr int q = decltype(QC)::value, t0 = U - 1 + 4 * q, ro = q * 4 * CE;
        if constexpr (t0 + 6 >= 2 && t0 <= 30) {
            double S = 0.0, xc = 0.0;
            if constexpr (t0 >= 3) {   // full pass for the first diagonal of this parity: taps l1 = 1 .. t0-1 (the centre tap below)
                double S2 = 0.0;       // two accumulators: the pass is a dependent FMA chain otherwise
                static_for<t0 - 1>([&](auto LC) {
                    constexpr int l1 = decltype(LC)::value + 1, dist = (2 * l1 - t0) < 0 ? (t0 - 2 * l1) : (2 * l1 - t0);
                    if constexpr (2 * l1 != t0) {
                        const double x = X(ro, 0, l1);
                        if constexpr (l1 & 1) S = fma(bw[dist / 2], x, S); else S2 = fma(bw[dist / 2], x, S2);
                    }
                });
                S += S2;
            }
            if constexpr (par == 0 && t0 >= 0) {   // centre tap: the same LDS word for every step of the chain
                xc = X(ro, 0, t0 / 2);
                if constexpr (t0 >= 2) S = fma(bw[0], xc, S);
            }
            static_for<4>([&](auto MC) {
                constexpr int m = decltype(MC)::value, t = t0 + 2 * m;
                if constexpr (t >= 0 && t <= 30) {
                    const double xl = X(ro, m, 0);
                    if constexpr (t == 0) {
                        S = fma(bw[0], xl, S);   // S_2[i-1] = B(0) x[i+1]: the two end taps of t = 0 are one tap
                    } else {
                        const double e = xl + X(ro, m, t);
                        if constexpr (t >= 2) {
                            double a = g[m];
                            a = fma(fw[(t + 1) * 4 + 0], S, a);
                            a = fma(fw[(t + 1) * 4 + 1], e, a);
                            if constexpr (par == 0) a = fma(fw[(t + 1) * 4 + 3], xc, a);
                            if constexpr (t >= 3 && t <= 7)   // explicit shapes (l1, l2 <= 4): symmetric pairs share their residual
                                static_for<2>([&](auto LC) {
                                    constexpr int l1 = decltype(LC)::value + (t - 4 > 1 ? t - 4 : 1), l2 = t - l1;
                                    if constexpr (l1 < l2 && l2 <= 4) a = fma(F[192 + 4 * t + l1 - 1], X(ro, m, l1) + X(ro, m, l2), a);
                                });
                            g[m] = a;
                        }
                        if constexpr (m < 3 && t + 2 <= 30) S = fma(bw[t / 2], e, S);
                    }
                }
            });
#pragma unroll
            for (int m = 0; m < 4; m++) asm volatile("" : "+v"(g[m]));   // one row's arithmetic before the next row's (volatile) reads
        }
    });
}

template <bool OUTSIDE, int CE>
__device__ __forceinline__ void filt_factored_any(int u, lds_vp xb, lds_vp fw, const double* __restrict__ F, double (&g)[4])
{
    switch (u) {
        case 0: filt_factored<0, OUTSIDE, CE>(xb, fw, F, g); break;
        case 1: filt_factored<1, OUTSIDE, CE>(xb, fw, F, g); break;
        case 2: filt_factored<2, OUTSIDE, CE>(xb, fw, F, g); break;
        case 3: filt_factored<3, OUTSIDE, CE>(xb, fw, F, g); break;
        default: filt_factored<4, OUTSIDE, CE>(xb, fw, F, g); break;
    }
}
constexpr int kStripFiltOffD = 31 * 40;   // offset of the factored tables behind the dense wT (kWTS = 40 columns, 31 rows)

// LDS plan of the inside strip kernel (doubles)
template <int KD, int W>
struct InStripPlan {
    static constexpr int GS = 64 - (KD - 1);
    static constexpr int NM = 31;                 // near terms per end of the FM2 sum
    static constexpr int RD = NM + KD, CD = 72;   // FM rows e = 1 .. (row e-1), columns i0+d0-e .. +71; rows e > 31 are zero
    static constexpr int RA = NM + KD, CA = 96;   // FM rows d0-31 .. d0-1, columns i0+1 .. +95; KD zero rows behind them
    static constexpr int RE = 32, CE = 96;        // FCX rows d0-1-rho, rho = 0..31, columns i0+1 .. +95
    static constexpr int OFF_DUMMY = RD * CD + RA * CA + RE * CE;   // one row that is never read: the sink of the staging slots that hold no row
    // after the pre-phase everything behind the first KD rows of the fixed FM rows is dead and is reused:
    static constexpr int CS = 72;                 // row pitch of the strip's own rows
    static constexpr int TS = 4;                  // term sets: wavefront w accumulates term set w % TS for diagonals (w / TS) * KD*TS/W ..
    static constexpr bool SPLIT = W == 8 && KD == 8;   // the FM2 sums are split by END between the wavefront halves (see the kernel)
    static constexpr int PS = SPLIT ? 3 : 2;      // PART slots: FM2 partial sum (SPLIT: low end), filter, (SPLIT: high end)
    static constexpr int OFF_PART = KD * CD;      // [TS][KD][PS][64] partial sums
    static constexpr int OFF_S = OFF_PART + TS * KD * PS * 64;   // [5][KD][CS] rows FM, FM1, FCX, FC, FCA of this strip
    static constexpr int OFF_F1S = OFF_S + 5 * KD * CS;          // [KD-1][64] FM1[m][i], m < KD, handed from the pre-phase to the chain
    static constexpr int SZ = OFF_DUMMY + CA > OFF_F1S + (KD - 1) * 64 ? OFF_DUMMY + CA : OFF_F1S + (KD - 1) * 64;
};

// ---------------------------------------------------------------------------------------------------------------
// inside, diagonals d0 .. d0+KD-1 (d0 >= 32, d0 % KD == 0).  Last group: F5i[f5_lo .. d0+1] (rows <= d0-1 of FCA).
// wT[l1*kWTS + t + 1] = shape_w(l1, t-l1) for 0 <= l1 <= t <= 30, else 0.
constexpr int kWTS = 40;

template <int KD, int W, int FILT>
__global__ __launch_bounds__(64 * W, (W >= 8 ? 4 : 2)) void lin_inside_strip(McBatch B, const LinModel* __restrict__ L, const double* __restrict__ wT, int d0, int f5_lo,
                                                              double lam_d0, int pin)
{
    using P = InStripPlan<KD, W>;
    constexpr int GS = P::GS, NM = P::NM, CD = P::CD, CA = P::CA, CE = P::CE, CS = P::CS;
    constexpr int NSL = KD / W;                     // chain steps per wavefront
    constexpr int TS = P::TS, KH = KD * TS / W;     // term sets; diagonals per wavefront in the pre-phase
    static_assert(KD % W == 0 && KD <= 8 && W % TS == 0 && KD % (W / TS) == 0 && (NM + 1) % W == 0, "8 terms per end and term set, 32/W staged rows per wavefront and region");
    __shared__ double lds[P::SZ];
    __shared__ double red[W];
    constexpr bool FACT = FILT != 0 && W == 8 && KD == 8;
    __shared__ double fw[FACT ? 160 : 1];   // factored filter: W4[t+1][4] (A, wb, -, Rc), read by wave-uniform (broadcast) LDS loads
    // pin = 2 (batch size a multiple of 8): workgroups are dealt round-robin to the 8 XCDs in launch order, so sequence sq is
    // pinned to XCD sq % 8 AND the groups of one sequence are consecutive on that XCD: neighbouring groups share 40 % of their
    // staged columns, which then come out of that XCD's L2 instead of HBM.  (Placement only: any mapping gives the same result.)
    int sq = pin ? blockIdx.x : blockIdx.y, slot = pin ? blockIdx.y : blockIdx.x;
    if (pin == 2) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, t = lin >> 3;
        slot = (int)(t % gridDim.y);
        sq = (int)(t / gridDim.y) * 8 + (int)(lin & 7);
    }
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wt = w % TS, kb = (w / TS) * KH;     // term set; first diagonal of this wavefront's pre-phase sums
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const int ncell0 = n - 1 - d0 > 0 ? n - 1 - d0 : 0;
    const int ngroup = (ncell0 + GS - 1) / GS;
    if (slot > ngroup) return;

    if (slot == ngroup) {
        // F5i[jj] = F5i[jj-1]*ext_unpaired + sum_{k<=jj-2} F5i[k]*FCA[k+1,jj-1]*ext_paired   (ipp:3692-3717)
        const double* __restrict__ fca = tab + S_FCA * ts;
#pragma unroll 1
        for (int jj = f5_lo; jj <= d0 + 1; jj++) {
            if (jj < 1 || jj > n) continue;
            // fixed partition of the sum (256 threads, 4 partial sums) whatever W is: the bits of F5i do not depend on which
            // kernel -- bootstrap, strip or tail -- computes an entry, i.e. not on the other sequences of the batch
            double acc = 0.0;
            if (threadIdx.x < 256)
                for (int k = threadIdx.x; k <= jj - 2; k += 256) acc = fma(f5i[k], fca[(size_t)(jj - 2 - k) * ld + (k + 1)], acc);
            acc = wsum_s(acc);
            if (lane == 0) red[w] = acc;
            __syncthreads();
            if (threadIdx.x == 0) f5i[jj] = f5i[jj - 1] * L->w_eu + (red[0] + red[1] + red[2] + red[3]) * L->w_ep2;
            __syncthreads();   // F5i[jj] is an operand of F5i[jj+1]
        }
        return;
    }

#ifdef RH_STAGGER
    {   // tuning build: offset the second workgroup of every CU by about half a workgroup lifetime (lock-step rounds)
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        if (lin >= 256 && lin < 512) for (int q = 0; q < RH_STAGGER; q++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    RH_STAMPI_BEGIN();
    RH_TRACE_ID(); RH_TRACE(1);
    const int i0 = 1 + slot * GS;
    const int i = i0 + lane;
    const int ic = i < ld ? i : ld - 1;
    double* const LDm = lds;                        // fixed FM rows e = 1..
    double* const LA = lds + P::RD * CD;            // sliding FM rows
    double* const LE = LA + P::RA * CA;             // sliding FCX rows
    const double* __restrict__ fm = tab + S_FM * ts;
    const double* __restrict__ fm1 = tab + S_FM1 * ts;
    const double* __restrict__ fcx = tab + S_FCX * ts;

    // sequence letters of this wavefront's chain steps k = w + W*sl: the FIRST loads issued (their table gathers follow as soon as the
    // staging loads are out, and vmcnt counts in order).  Unconditional loads at clamped positions, selected afterwards: a
    // branch around a load makes the compiler drain every outstanding load at the join.
    int s_j[NSL], s_jp1[NSL], s_jp2[NSL];
    int s_im1, s_i, s_ip1;
    int rb0, rb1, rb2, r0[NSL], r1[NSL], r2[NSL];
    {
        const int top = B.lds - 1;
        const int c_i = i < top ? i : top - 1;                   // i >= 1 here
        rb0 = s[c_i - 1]; rb1 = s[c_i]; rb2 = s[c_i + 1];
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            const int j = i + d0 + w + W * sl, cj = j + 2 <= top ? j : top - 2;
            r0[sl] = s[cj]; r1[sl] = s[cj + 1]; r2[sl] = s[cj + 2];
        }
    }
    // ---- global loads of everything this wavefront stages or keeps: issued back to back (64 row segments in flight), then written to LDS.
    // Cells outside the interior of their row are staged as 0: stale bytes never enter a product.
    // No address clamps: a column past the end of its row (or a row past its table) still lies inside this sequence's table
    // block, and whatever is read there is replaced by 0 on the way into LDS.  The 32-column (8-column) tails of two (NR)
    // rows share one load: lane halves (eighths) select the row.
    constexpr int NR = (NM + 1) / W;   // staged rows per wavefront and region
    constexpr int NT = (NM + 1) / TS;  // terms per end of this wavefront's term set
    static_assert(NR % 2 == 0 && NR <= 8, "row tails are paired");
    double vA0[NR], vA1[NR / 2], vD0[NR], vD1, vE0[NR], vE1[NR / 2], a_lo[NT], a_hi[NT];
    const int lhalf = lane >> 5, lsub = lane & 31;
    const int ca0 = i0 + 1 + lane, ca1 = i0 + 65 + lsub;            // columns of the 64-wide part / of the tail
    const int dsel = lane >> 3, dsub = lane & 7;                     // D tails: row = dsel (< NR), 8 columns each
    const int eD = 1 + w + W * dsel, cD1 = i0 + d0 - eD + 64 + dsub;
    // A group whose 96-column windows lie inside every staged row (all but the last two groups of a diagonal) needs no masks: its rows
    // go from HBM straight into LDS (global_load_lds_dwordx4, 16 bytes per lane: the three regions are contiguous row-major, so one
    // instruction fills 1024 consecutive LDS bytes from per-lane addresses) -- 9 instead of 17 loads per wavefront, no registers, no
    // LDS-write phase.  RH_STRIP_DMA=0 (tuning build) keeps the register path for every group.
#ifndef RH_STRIP_DMA
#define RH_STRIP_DMA 1
#endif
    const bool full = RH_STRIP_DMA && W == 8 && i0 + 96 <= n - d0;   // (workgroup-uniform)
    if (full) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int blk = k * W + w, chunk = blk * 64 + lane;   // 16-byte chunk of the region
            if (chunk < NM * (CA / 2)) {                           // A: rows d0-31+row, columns i0+1 ..
                const int row = chunk / (CA / 2), within = chunk - row * (CA / 2);
                __builtin_amdgcn_global_load_lds((gptr_t)(fm + (unsigned)((d0 - NM + row) * ld) + (unsigned)(i0 + 1 + 2 * within)), (lptr_t)(LA + blk * 128), 16, 0, 0);
            }
            {                                                      // E: rows d0-1-rho, columns i0+1 ..  (32 x 48 chunks = 3 x 512)
                const int row = chunk / (CE / 2), within = chunk - row * (CE / 2);
                __builtin_amdgcn_global_load_lds((gptr_t)(fcx + (unsigned)((d0 - 1 - row) * ld) + (unsigned)(i0 + 1 + 2 * within)), (lptr_t)(LE + blk * 128), 16, 0, 0);
            }
            if (chunk < NM * (CD / 2)) {                           // D: rows e = 1 + row, columns i0+d0-e ..
                const int row = chunk / (CD / 2), within = chunk - row * (CD / 2), e = row + 1;
                __builtin_amdgcn_global_load_lds((gptr_t)(fm + (unsigned)(e * ld) + (unsigned)(i0 + d0 - e + 2 * within)), (lptr_t)(LDm + blk * 128), 16, 0, 0);
            }
        }
    } else {
        const unsigned oA0 = (unsigned)ca0, oA1 = (unsigned)(ca1 + lhalf * W * ld);
        const unsigned oD0 = (unsigned)(i0 + d0 + lane);
#pragma unroll
        for (int q = 0; q < NR; q++) {
            const int ra = w + W * q;                                // A row d0-31+ra (ra = 31: a zero row, anything may be read), E row d0-1-rho, rho = ra
            vA0[q] = fm[(unsigned)((d0 - NM + (ra < NM ? ra : NM - 1)) * ld) + oA0];
            vE0[q] = fcx[(unsigned)((d0 - 1 - ra) * ld) + oA0];
            const int e = 1 + w + W * q, ee = e <= NM ? e : NM;      // D row e, columns i0+d0-e ..
            vD0[q] = fm[(unsigned)(ee * (ld - 1)) + oD0];
        }
#pragma unroll
        for (int p2 = 0; p2 < NR / 2; p2++) {
            const int ra = w + W * 2 * p2;                           // lanes 0..31: row ra, lanes 32..63: row ra+W
            vA1[p2] = fm[(unsigned)((d0 - NM + ra) * ld) + oA1];                                  // (row ra+W = 31 is a zero row: masked below)
            vE1[p2] = fcx[(unsigned)((d0 - 1 - ra - W) * ld) + (unsigned)(ca1 + (1 - lhalf) * W * ld)];   // E rows run downwards: row ra in lanes 0..31
        }
        vD1 = fm[(unsigned)((eD <= NM ? eD : NM) * ld) + (unsigned)(cD1 > 0 ? cD1 : 0)];
    }
    // SPLIT (8 wavefronts): wavefronts 0..3 take the low-end sums of ALL eight diagonals and load only FM1[m][i], wavefronts 4..7 the high-end
    // sums and only FM1[d0-m][i] -- half the loads of the (term set, diagonal half) split, in which the wavefronts w and w+4 fetched the same
    // sixteen row segments; the two halves land in PART slots 0 and 2 and are added by the chain.
    constexpr bool SPLIT = P::SPLIT;
    const bool hi_role = SPLIT && w >= TS;
#pragma unroll
    for (int q = 0; q < NT; q++) {   // this term set's FM1 values: rows m = 1+wt+TS*q (low end) and d0-m (high end), column i.  Slot m = 32 is
        // no term: it is paired with zero rows below.  Per-lane validity is a select; the wave-uniform condition of the high end
        // (m <= d0-32) is a 0/1 factor applied at use -- a uniform select on a load becomes a branch with a full drain behind it.
        const int m = 1 + wt + TS * q, R = d0 - m;
        if constexpr (SPLIT) {   // one load per term: the row is selected by the wavefront's role (no branch)
            const int row = hi_role ? R : m;
            const double x = fm1[(unsigned)(row * ld) + (unsigned)i];
            a_lo[q] = i <= n - 1 - row ? x : 0.0;
            a_hi[q] = 0.0;
        } else {
            const double v = fm1[(unsigned)(m * ld) + (unsigned)i], u = fm1[(unsigned)(R * ld) + (unsigned)i];
            a_lo[q] = i <= n - 1 - m ? v : 0.0;
            a_hi[q] = i <= n - 1 - R ? u : 0.0;
        }
    }
    double fwv = 0.0;
    if constexpr (FACT) fwv = wT[kStripFiltOffD + (threadIdx.x < 160 ? threadIdx.x : 159)];   // (with the staging loads: no round trip of its own)
    // the letters are first touched HERE, behind the staging loads (fence + pins: their wait must not move above those loads)
    __builtin_amdgcn_sched_barrier(0);
    RH_STAMPI(9);    // (tuning build: every staging load is issued)
    asm volatile("" : "+v"(rb0)); asm volatile("" : "+v"(rb1)); asm volatile("" : "+v"(rb2));
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) { asm volatile("" : "+v"(r0[sl])); asm volatile("" : "+v"(r1[sl])); asm volatile("" : "+v"(r2[sl])); }
    {
        const bool vi = i <= n;
        s_im1 = vi ? rb0 : 4; s_i = vi ? rb1 : 4; s_ip1 = vi ? rb2 : 4;
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            const bool v = i <= n - 1 - (d0 + w + W * sl);
            s_j[sl] = v ? r0[sl] : 4; s_jp1[sl] = v ? r1[sl] : 4; s_jp2[sl] = v ? r2[sl] : 4;
        }
    }
    RH_STAMPI(10);   // (tuning build: the letters have arrived)
    // ---- operands of this wavefront's chain steps, raw: issued behind the staging loads, consumed after the filter
    double r_tjb[NSL], r_tja[NSL], r_tst[NSL], r_bp[NSL], r_tjbd[NSL], r_tjad[NSL], r_b01[NSL], r_b10[NSL], r_11[NSL], p_far[NSL];
    double p_x01 = 0, p_x10 = 0, p_x11 = 0, p_fc = 0, p_fca = 0, p_fm1 = 0, p_fm = 0;   // only step k < 4 reads rows < d0 here
    {
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            const int d = d0 + w + W * sl;
            const bool v = i <= n - 1 - d;
            const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1[sl] + s_j[sl];       // (i,j)   as enclosing pair
            const int idd = 25 * (5 * s_jp1[sl] + s_jp2[sl]) + 5 * s_i + s_im1;     // (j+1,i-1) as enclosed pair
            r_tjb[sl] = L->TJB[idx]; r_tja[sl] = L->TJA[idx]; r_tst[sl] = L->TST[idx];
            r_bp[sl] = L->E_bp[s_i * 5 + s_jp1[sl]];
            r_tjbd[sl] = L->TJB[idd]; r_tjad[sl] = L->TJA[idd];
            r_b01[sl] = L->E_b01[s_j[sl]]; r_b10[sl] = L->E_b10[s_ip1]; r_11[sl] = L->E_11[s_ip1 * 5 + s_j[sl]];
            const double far = tab[S_FM2F * ts + (unsigned)(d * ld) + (unsigned)i];   // (rows < 64 hold nothing: selected away)
            p_far[sl] = (v & (d >= 64)) ? far : 0.0;
        }
        // the first of its steps may still need rows < d0 (unconditional loads, selected afterwards)
        const int k0 = w, d = d0 + k0;
        const bool v = i <= n - 1 - d;
        {
            const unsigned c1 = (unsigned)(i + 1), c2 = (unsigned)(i + 2);
            // Only the wavefronts that finish steps 0..3 read rows below d0: a wave-uniform nest, behind every other load of the batch (the
            // wait that follows is for the staged rows, which are older).  Staging time follows the number of distinct 128-byte lines a
            // workgroup pulls through its CU's L1 (profiles/r03_strip_timeline.txt), and these were 35 lines per wavefront
            double x01 = 0.0, x10 = 0.0, x11 = 0.0, fc = 0.0, fca = 0.0, xm1 = 0.0, xm = 0.0;
            if (k0 < 4) {
                x11 = fcx[(unsigned)((d - 4) * ld) + c2];
                if (k0 < 3) {
                    x01 = fcx[(unsigned)((d - 3) * ld) + c1]; x10 = fcx[(unsigned)((d - 3) * ld) + c2];
                    if (k0 < 2) {
                        fc = tab[S_FC * ts + (unsigned)((d - 2) * ld) + c1]; fca = tab[S_FCA * ts + (unsigned)((d - 2) * ld) + c1];
                        if (k0 < 1) { xm1 = fm1[(unsigned)((d - 1) * ld) + c1]; xm = fm[(unsigned)((d - 1) * ld) + (unsigned)i]; }
                    }
                }
            }
            p_x01 = (v & (k0 < 3)) ? x01 : 0.0; p_x10 = (v & (k0 < 3)) ? x10 : 0.0;
            p_x11 = (v & (k0 < 4)) ? x11 : 0.0;
            p_fc = (v & (k0 < 2)) ? fc : 0.0; p_fca = (v & (k0 < 2)) ? fca : 0.0;
            p_fm1 = (v & (k0 < 1)) ? xm1 : 0.0; p_fm = (v & (k0 < 1)) ? xm : 0.0;
        }
    }
    // Every staged value is stored unconditionally (slots that hold no row go to a dummy row) and pinned here: otherwise the
    // compiler sinks a LOAD into the (branchy) select at its store and waits for it there, one round trip per row.
    double* const LX = lds + P::OFF_DUMMY;
    if (!full) {
#pragma unroll
    for (int q = 0; q < NR; q++) { asm volatile("" : "+v"(vA0[q])); asm volatile("" : "+v"(vE0[q])); asm volatile("" : "+v"(vD0[q])); }
#pragma unroll
    for (int q = 0; q < NR / 2; q++) { asm volatile("" : "+v"(vA1[q])); asm volatile("" : "+v"(vE1[q])); }
    asm volatile("" : "+v"(vD1));
    RH_STAMPI(11);   // (tuning build: the staged values have arrived)
#pragma unroll
    for (int q = 0; q < NR; q++) {   // 64-wide parts
        const int ra = w + W * q, e = 1 + w + W * q;
        double* const dA = ra < NM ? LA + ra * CA : LX;
        double* const dD = e <= NM ? LDm + (e - 1) * CD : LX;
        dA[lane] = ca0 <= n - 1 - (d0 - NM + ra) ? vA0[q] : 0.0;
        LE[ra * CE + lane] = ca0 <= n - 1 - (d0 - 1 - ra) ? vE0[q] : 0.0;
        dD[lane] = i0 + d0 - e + lane <= n - 1 - e ? vD0[q] : 0.0;
    }
#pragma unroll
    for (int p2 = 0; p2 < NR / 2; p2++) {   // tails of rows ra (lanes 0..31) and ra+W (lanes 32..63)
        const int ra = w + W * 2 * p2, rl = ra + lhalf * W;
        double* const dA = rl < NM ? LA + rl * CA : LX;
        dA[64 + lsub] = ca1 <= n - 1 - (d0 - NM + rl) ? vA1[p2] : 0.0;
        LE[rl * CE + 64 + lsub] = ca1 <= n - 1 - (d0 - 1 - rl) ? vE1[p2] : 0.0;
    }
    {
        double* const dD = (dsel < NR && eD <= NM) ? LDm + (eD - 1) * CD : LX;
        dD[64 + dsub] = cD1 <= n - 1 - eD ? vD1 : 0.0;
    }
    }
    for (int k = threadIdx.x; k < (P::RA - NM) * CA; k += 64 * W) LA[NM * CA + k] = 0.0;
    for (int k = threadIdx.x; k < (P::RD - NM) * CD; k += 64 * W) LDm[NM * CD + k] = 0.0;
    if constexpr (FACT) { asm volatile("" : "+v"(fwv)); if (threadIdx.x < 160) fw[threadIdx.x] = fwv; }
    RH_STAMPI(0);
    __syncthreads();
    RH_STAMPI(1);
    RH_TRACE(2);

    // ---- pre-phase: the terms whose operands were final before the launch, all KD diagonals
    double acc2[KH], accg[KH];
#pragma unroll
    for (int k = 0; k < KH; k++) { acc2[k] = 0.0; accg[k] = 0.0; }
    double acc8[SPLIT ? KD : 1];   // SPLIT: this wavefront's end of the FM2 sum, all eight diagonals
    if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < KD; k++) acc8[k] = 0.0;
        const lds_vp vLA = (lds_vp)LA;
        const lds_vp vLD = (lds_vp)LDm;
        double vx[KD], nx[KD];
        // (software-pipelined by hand and pinned, volatile ds_read_b64: see the unsplit form below)
        if (!hi_role) {   // low end: FM1[m][i] * FM[d-m][i+m]; FM row d0+k-m sits in staged row 31+k-m (rows >= 31 are zero; slot m = 32 reads zero rows)
            auto issue = [&](int q, double* r) {
                const int m = 1 + wt + TS * q;
                const lds_vp pa = vLA + (m <= NM ? NM - m : NM) * CA + lane + (m <= NM ? m : NM) - 1;
#pragma unroll
                for (int k = 0; k < KD; k++) r[k] = pa[k * CA];
            };
            issue(0, vx);
#pragma unroll
            for (int q = 0; q < NT; q++) {
                if (q + 1 < NT) issue(q + 1, nx);
                const double a = a_lo[q];
#pragma unroll
                for (int k = 0; k < KD; k++) acc8[k] = fma(a, vx[k], acc8[k]);
#pragma unroll
                for (int k = 0; k < KD; k++) asm volatile("" : "+v"(acc8[k]));
#pragma unroll
                for (int k = 0; k < KD; k++) vx[k] = nx[k];
            }
        } else {          // high end: FM1[d0-x][i] * FM[e][i+d0-x], e = k+x, staged row e-1 at column index lane+k (rows e > 31 are zero)
            auto issue = [&](int q, double* r) {
                const int x = 1 + wt + TS * q;
                const lds_vp pd = vLD + (x - 1) * CD + lane;
#pragma unroll
                for (int k = 0; k < KD; k++) r[k] = pd[k * (CD + 1)];
            };
            issue(0, vx);
#pragma unroll
            for (int q = 0; q < NT; q++) {
                if (q + 1 < NT) issue(q + 1, nx);
                const double b = a_lo[q] * ((1 + wt + TS * q) <= d0 - 32 ? 1.0 : 0.0);   // e = k+x <= d-32: rows m' = d-e >= 32 only
#pragma unroll
                for (int k = 0; k < KD; k++) acc8[k] = fma(b, vx[k], acc8[k]);
#pragma unroll
                for (int k = 0; k < KD; k++) asm volatile("" : "+v"(acc8[k]));
#pragma unroll
                for (int k = 0; k < KD; k++) vx[k] = nx[k];
            }
        }
    } else {
        // two LDS operands per term and diagonal; software-pipelined by hand (the reads of term q+1 are issued before the
        // FMAs of term q) and pinned, because left alone the scheduler hoists every read of this block and spills.  The
        // volatile pointer keeps the reads as ds_read_b64: merged into ds_read2_b64 they run at half the LDS rate on CDNA4.
        const lds_vp vLA = (lds_vp)LA;
        const lds_vp vLD = (lds_vp)LDm;
        double va[KH], vb[KH], na[KH], nb[KH];
        auto issue = [&](int q, double* ra, double* rb) {
            // low end: FM1[m][i] * FM[d-m][i+m]; FM row d0+k-m sits in staged row 31+k-m (rows >= 31 are zero: the strip's own rows, added by
            // the chain; slot m = 32 reads zero rows only)
            const int m = 1 + wt + TS * q;
            const lds_vp pa = vLA + ((m <= NM ? NM - m : NM) + kb) * CA + lane + (m <= NM ? m : NM) - 1;
            // high end: FM1[d0-x][i] * FM[e][i+d0-x], e = k+x, staged row e-1 at column index lane+k (rows e > 31 are zero)
            const int x = m;
            const lds_vp pd = vLD + (x - 1 + kb) * CD + lane + kb;
#pragma unroll
            for (int k = 0; k < KH; k++) { ra[k] = pa[k * CA]; rb[k] = pd[k * (CD + 1)]; }
        };
        issue(0, va, vb);
#pragma unroll
        for (int q = 0; q < NT; q++) {
            if (q + 1 < NT) issue(q + 1, na, nb);
            const double a = a_lo[q], b = a_hi[q] * ((1 + wt + TS * q) <= d0 - 32 ? 1.0 : 0.0);   // e = k+x <= d-32: rows m' = d-e >= 32 only
#pragma unroll
            for (int k = 0; k < KH; k++) acc2[k] = fma(b, vb[k], fma(a, va[k], acc2[k]));
#pragma unroll
            for (int k = 0; k < KH; k++) asm volatile("" : "+v"(acc2[k]));   // pins this term's FMAs before the next term's (volatile) reads
#pragma unroll
            for (int k = 0; k < KH; k++) { va[k] = na[k]; vb[k] = nb[k]; }
        }
    }
    RH_STAMPI(2);
    // FM1[m][i], m = 1..KD-1 (low-end operands of the chain's own-row terms) are a_lo[0], a_lo[1] of the wavefronts with kb == 0: m = 1+wt
    // and 5+wt.  They are handed to every wavefront through LDS behind the filter (F1S), instead of seven more loads per wavefront.
    const double f1keep0 = a_lo[0], f1keep1 = a_lo[1];
    RH_STAMPI(3);
    // single-branch filter: staged row rho is table row d0-1-rho = d-2-t with t = rho-1+k for diagonal d0+k; tap l1 reads column
    // i+1+l1 of it.  One LDS read feeds all KD diagonals.  The weight of (row rho, diagonal k, tap l1) is wT[l1][rho+k] (zero where
    // the shape does not exist), so for one tap the 8 rows x 8 diagonals of this wavefront are 36 consecutive entries of the
    // transposed table: wave-uniform scalar loads, 20 doubles per half (rows q = 0..3 / 4..7) and 32 FMAs behind each.
    // The first half of the term set's rows (rho <= wt + TS*3) has no tap beyond l1 = rho+kb+KH-2: skipped for larger l1.
    if constexpr (FACT) {
        // factored filter (see filt_factored): this wavefront takes the diagonals k0, k0+2, k0+4, k0+6 of rows rho = wt + 4q; accg[m] is
        // the sum of diagonal k0+2m for the cell in column i-m
        filt_factored_any<false, CE>(wt + w / TS, (lds_vp)LE + wt * CE + lane - 3, (lds_vp)fw, wT + kStripFiltOffD, accg);
    } else {
        constexpr int HR = NT / 2;
        const lds_vp xrow = (lds_vp)LE + wt * CE + lane;
#pragma unroll 1
        for (int l1 = 0; l1 <= kMaxSingle; l1++) {
            const double* __restrict__ wp = wT + l1 * kWTS + wt + kb;
            if (l1 <= wt + (HR - 1) * TS + kb + KH - 2) {
                double x[HR];
#pragma unroll
                for (int q = 0; q < HR; q++) x[q] = xrow[q * TS * CE + l1];
#pragma unroll
                for (int q = 0; q < HR; q++)
#pragma unroll
                    for (int k = 0; k < KH; k++) accg[k] = fma(wp[TS * q + k], x[q], accg[k]);
            }
            {
                double x[HR];
#pragma unroll
                for (int q = 0; q < HR; q++) x[q] = xrow[(q + HR) * TS * CE + l1];
#pragma unroll
                for (int q = 0; q < HR; q++)
#pragma unroll
                    for (int k = 0; k < KH; k++) accg[k] = fma(wp[TS * (q + HR) + k], x[q], accg[k]);
            }
        }
    }
    double cw[(KD - 1) * (KD - 2) / 2];   // single-branch weights of the chain's own-row taps (t <= KD-3), wave-uniform
#pragma unroll
    for (int k = 0; k < (KD - 1) * (KD - 2) / 2; k++) cw[k] = L->shape_w[k];
    // chain coefficients from the raw gathers (0 for a cell that is no pair: its FC is 0)
    double p_tjb[NSL], p_cst[NSL], p_ctja[NSL], p_cbx[NSL], p_cba[NSL], p_c01[NSL], p_c10[NSL], p_c11[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) {
        const int d = d0 + w + W * sl;
        const bool pr = i <= n - 1 - d && pairs_s(s_i, s_jp1[sl]);
        p_tjb[sl] = pr ? r_tjb[sl] : 0.0;
        p_ctja[sl] = pr ? r_tja[sl] * L->e_mpmb : 0.0;
        p_cst[sl] = pr ? r_tst[sl] * L->lam2 : 0.0;
        p_cbx[sl] = r_bp[sl] * r_tjbd[sl];
        p_cba[sl] = r_bp[sl] * r_tjad[sl];
        p_c01[sl] = L->w01 * r_b01[sl];
        p_c10[sl] = L->w10 * r_b10[sl];
        p_c11[sl] = L->w11 * r_11[sl];
    }
    RH_STAMPI(4);
    __syncthreads();   // every wavefront is done with the staged rows: the region behind the first KD fixed rows is reused
    RH_STAMPI(5);
    RH_TRACE(3);
    double* const PART = lds + P::OFF_PART;
    double* const SFM = lds + P::OFF_S;
    double* const SFM1 = SFM + KD * CS;
    double* const SFCX = SFM1 + KD * CS;
    double* const SFC = SFCX + KD * CS;
    double* const SFCA = SFC + KD * CS;
    double* const F1S = lds + P::OFF_F1S;   // [KD-1][64]: behind the strip's own rows (staged filter rows, dead now)
    constexpr int PS = P::PS;
    if (kb == 0) {
        F1S[wt * 64 + lane] = f1keep0;                              // m = 1 + wt
        if (4 + wt < KD - 1) F1S[(4 + wt) * 64 + lane] = f1keep1;   // m = 5 + wt <= 7
    }
    if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < KD; k++) PART[((wt * KD + k) * PS + (hi_role ? 2 : 0)) * 64 + lane] = acc8[k];
    }
#pragma unroll
    for (int k = 0; k < KH; k++) {
        if constexpr (!SPLIT) PART[((wt * KD + kb + k) * PS + 0) * 64 + lane] = acc2[k];
        if constexpr (FACT) {   // accg[k]: diagonal k0+2k of column i-k (k0 = w / TS); lanes < k hold cells of the group to the left
            if (lane >= k) PART[((wt * KD + w / TS + 2 * k) * PS + 1) * 64 + lane - k] = accg[k];
        } else {
            PART[((wt * KD + kb + k) * PS + 1) * 64 + lane] = accg[k];
        }
    }
    __syncthreads();

    double f1[KD - 1];
#pragma unroll
    for (int m = 1; m < KD; m++) f1[m - 1] = F1S[(m - 1) * 64 + lane];
    // ---- chain: step K finishes diagonal d0+K on wavefront K % W.  Only the terms that touch row K-1 (and the epilogue) are on the
    // critical path: in time slot T wavefront T % W finishes step T (fin) while the next wavefront gathers, for step T+1, the
    // partial sums and every term of rows <= T-1 (pre).
    const double w_mu = L->w_mu, w_mp2 = L->w_mp2;
    const double hp30 = L->E_hairpin[30], lam = L->lam;
    double fm2s[NSL], gs[NSL];
    // every step starts from the far sum and the four term sets' partial sums
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) {
        const int K = w + W * sl;
        double fm2 = p_far[sl], g = 0.0;
#pragma unroll
        for (int q = 0; q < TS; q++) {
            fm2 += PART[((q * KD + K) * PS + 0) * 64 + lane];
            if constexpr (SPLIT) fm2 += PART[((q * KD + K) * PS + 2) * 64 + lane];
            g += PART[((q * KD + K) * PS + 1) * 64 + lane];
        }
        fm2s[sl] = fm2; gs[sl] = g;
    }
    // the terms of step K that touch the strip's own row R <= K-2 (FM2: m = K-R at both ends; filter: t = K-2-R)
    auto addrow = [&](auto RC, auto KC) {
        constexpr int R = decltype(RC)::value, K = decltype(KC)::value, SL = K / W, m = K - R, t = K - 2 - R;
        double fm2 = fm2s[SL], g0 = 0.0, g1 = 0.0;
        fm2 = fma(f1[m - 1], SFM[R * CS + lane + m], fm2);                                            // FM1[m][i] * FM[d0+R][i+m]
        if (m <= d0 + K - 32) fm2 = fma(SFM1[R * CS + lane], LDm[(m - 1) * CD + lane + K], fm2);        // FM1[d0+R][i] * FM[e = m][i+d0+R]
#pragma unroll
        for (int l1 = 0; l1 <= t; l1++) {
            const double x = SFCX[R * CS + lane + 1 + l1];
            if (l1 & 1) g1 = fma(cw[t * (t + 1) / 2 + l1], x, g1); else g0 = fma(cw[t * (t + 1) / 2 + l1], x, g0);
        }
        fm2s[SL] = fm2; gs[SL] += g0 + g1;
    };
    auto fin = [&](auto KC) {
        constexpr int K = decltype(KC)::value, SL = K / W;
        const int d = d0 + K;
        const bool v = i <= n - 1 - d;
        double fm2 = fm2s[SL];
        const double g = gs[SL];
        if constexpr (K >= 1) {   // the two terms that touch row K-1
            fm2 = fma(f1[0], SFM[(K - 1) * CS + lane + 1], fm2);
            if (1 <= d - 32) fm2 = fma(SFM1[(K - 1) * CS + lane], LDm[lane + K], fm2);
        }
        double x01 = p_x01, x10 = p_x10, x11 = p_x11, o_fc = p_fc, o_fca = p_fca, o_fm1 = p_fm1, o_fm = p_fm;
        if constexpr (K >= 3) { x01 = SFCX[(K - 3) * CS + lane + 1]; x10 = SFCX[(K - 3) * CS + lane + 2]; }
        if constexpr (K >= 4) x11 = SFCX[(K - 4) * CS + lane + 2];
        if constexpr (K >= 2) { o_fc = SFC[(K - 2) * CS + lane + 1]; o_fca = SFCA[(K - 2) * CS + lane + 1]; }
        if constexpr (K >= 1) { o_fm1 = SFM1[(K - 1) * CS + lane + 1]; o_fm = SFM[(K - 1) * CS + lane]; }
        double lk = lam_d0;
#pragma unroll
        for (int q = 0; q < K; q++) lk *= lam;
        const double hp = lk * hp30;                                                   // ScoreHairpin, d >= 30 (ipp:2123-2152)
        const double sp = p_c01[SL] * x01 + p_c10[SL] * x10 + p_c11[SL] * x11;
        double fc = p_tjb[SL] * (g + sp + hp) + p_cst[SL] * o_fc + fm2 * p_ctja[SL];   // ipp:3573-3622 (coefficients are 0 for a non-pair)
        double fm1v = o_fca * w_mp2 + o_fm1 * w_mu;                                    // ipp:3641-3688
        double fmv = fm2 + o_fm * w_mu + fm1v;
        if (!v) { fc = 0.0; fm1v = 0.0; fmv = 0.0; }
        const double fcxv = fc * p_cbx[SL], fcav = fc * p_cba[SL];
        SFM[K * CS + lane] = fmv; SFM1[K * CS + lane] = fm1v; SFCX[K * CS + lane] = fcxv; SFC[K * CS + lane] = fc; SFCA[K * CS + lane] = fcav;
    };
    RH_STAMPI(6);
    // time slot T: wavefront T % W finishes step T (fin: the two terms of row T-1 + the epilogue); every later step K > T, on
    // its own wavefront, adds the terms of row T-1, which the previous slot completed
#define RH_SLOT(T)                                                                                  \
    if constexpr (T < KD) {                                                                         \
        if (w == T % W) fin(IC<T>{});                                                               \
        if constexpr (T >= 1 && T + 1 < KD)                                                         \
            static_for<KD>([&](auto KC) {                                                           \
                constexpr int K = decltype(KC)::value;                                              \
                if constexpr (K > T) { if (w == K % W) addrow(IC<(T >= 1 ? T - 1 : 0)>{}, KC); }    \
            });                                                                                     \
        if constexpr (T + 1 < KD) lds_barrier();                                                    \
    }
    RH_SLOT(0) RH_SLOT(1) RH_SLOT(2) RH_SLOT(3) RH_SLOT(4) RH_SLOT(5) RH_SLOT(6) RH_SLOT(7)
#undef RH_SLOT
    RH_STAMPI(7);
    lds_barrier();
    // ---- the strip's rows go to HBM now, off the chain: row r = (table, diagonal) by wavefront r % W, columns of this group only
    if (lane < GS) {
#pragma unroll 1
        for (int r = w; r < 5 * KD; r += W) {
            const int tb = r / KD, k = r - tb * KD, d = d0 + k;   // S row order: FM, FM1, FCX, FC, FCA
            const int slot_of[5] = {S_FM, S_FM1, S_FCX, S_FC, S_FCA};
            if (i <= n - 1 - d) tab[slot_of[tb] * ts + (size_t)d * ld + i] = SFM[r * CS + lane];
        }
    }
    RH_STAMPI(8);
    RH_TRACE(4);
}

// F5i[jj], jj = jlo .. n, once every row of FCA is final (after the last strip)
__global__ __launch_bounds__(256) void lin_f5i_tail(McBatch B, const LinModel* __restrict__ L, int jlo)
{
    __shared__ double red[4];
    const int sq = blockIdx.x;
    if (sq >= B.ns) return;
    const int n = B.n[sq], ld = B.ld;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const double* __restrict__ fca = B.tab + (size_t)sq * B.seq_stride + S_FCA * B.tab_stride;
#pragma unroll 1
    for (int jj = jlo < 1 ? 1 : jlo; jj <= n; jj++) {
        double acc = 0.0;
        for (int k = threadIdx.x; k <= jj - 2; k += 256) acc = fma(f5i[k], fca[(size_t)(jj - 2 - k) * ld + (k + 1)], acc);
        acc = wsum_s(acc);
        if (lane == 0) red[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) f5i[jj] = f5i[jj - 1] * L->w_eu + (red[0] + red[1] + red[2] + red[3]) * L->w_ep2;
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// outside (pull form) + posterior, diagonals d0, d0-1, .., d0-KD+1 (d0 % KD == KD-1); the mirror image of the inside strip:
//   FMo [i,d] = sum_{e=1}^{31} FM2o[d+e][i-e] * FM1[e][i-e] + FMOF [i,d]   (far: i' <= i-32, block products)     ipp:4046-4064, pulled
//   FM1o[i,d] = sum_{e=1}^{31} FM2o[d+e][i]   * FM [e][i+d] + FM1OF[i,d]   (far: j' >= j+32)
//   FCo gather: sum_t sum_l1 w(l1,t-l1) * FCoX[d+2+t][i-1-l1]                                                       ipp:4004-4024, pulled
// The sliding rows are d0+1 .. d0+32 (final before the launch), the strip's own rows are reached to the LEFT (columns i-e), so
// the trapezoid shrinks from lane 0: step k is valid for lanes >= k, a group stores lanes KD-1 .. 63 and groups advance by
// 64-(KD-1) columns.  Last group: F5o[f5_hi .. f5_lo] (descending), what the NEXT launch's cells read.
template <int KD, int W>
struct OutStripPlan {
    static constexpr int GS = 64 - (KD - 1);
    static constexpr int NM = 31;
    static constexpr int TS = 4;
    static constexpr int RD = NM + KD, CD = 72;       // FM rows e = 1.. (row e-1), columns i0+d0-(KD-1) .. +71; rows e > 31 zero
    static constexpr int RA = NM + KD, CA = 96;       // KD zero rows, then FM2o rows d0+1 .. d0+31; columns i0-31 .. i0+64
    static constexpr int RE = 32, CE = 96;            // FCoX rows d0+1+rho; columns i0-32 .. i0+63
    static constexpr int OFF_DUMMY = RD * CD + RA * CA + RE * CE;   // one row that is never read: the sink of the staging slots that hold no row
    static constexpr int SZ = OFF_DUMMY + CA + 320;
    static constexpr int PADL = 8, CS = 72;           // strip rows: lane l at index l+PADL
    static constexpr int KEEP = (KD - 1) * CD;        // fixed FM rows e = 1..KD-1 stay live through the chain
    static constexpr int OFF_PART = KEEP;             // [TS][KD][3][64] partial sums; row k is reused for the posterior of diagonal k
    static constexpr int OFF_S = OFF_PART + TS * KD * 3 * 64;   // [5][KD][CS] rows FM2o, FMo, FM1o, FCo, FCoX of this strip
    static_assert(OFF_S + 5 * KD * CS <= SZ, "overlay does not fit");
    static_assert(SZ * 8 <= 80 * 1024, "two workgroups per CU");
};

// F5o[k] = F5o[k+1]*ext_unpaired + sum_{jj>=k+2} F5o[jj]*FCA[k+1,jj-1]*ext_paired, k = khi .. klo descending   (ipp:3751-3780, pulled)
template <int W>
__device__ __forceinline__ void f5o_range(const double* __restrict__ fca_tab, double* __restrict__ f5o, const LinModel* __restrict__ L, int n, int ld,
                                          int khi, int klo, double* red)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll 1
    for (int k = khi; k >= klo && k >= 1; k--) {
        if (k > n - 1) continue;
        const double* __restrict__ fca = fca_tab + (k + 1);
        double acc = 0.0;   // fixed partition (256 threads, 4 partial sums) whatever W is: see the inside strip
        if (threadIdx.x < 256)
            for (int jj = k + 2 + threadIdx.x; jj <= n; jj += 256) acc = fma(f5o[jj], fca[(size_t)(jj - 2 - k) * ld], acc);
        acc = wsum_s(acc);
        if (lane == 0) red[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) f5o[k] = f5o[k + 1] * L->w_eu + (red[0] + red[1] + red[2] + red[3]) * L->w_ep2;
        __syncthreads();   // F5o[k] is an operand of F5o[k-1]
    }
}

// F5o[khi .. klo] of every sequence before the first outside strip
__global__ __launch_bounds__(256) void lin_f5o_head(McBatch B, const LinModel* __restrict__ L, int khi, int klo)
{
    __shared__ double red[4];
    const int sq = blockIdx.x;
    if (sq >= B.ns) return;
    f5o_range<4>(B.tab + (size_t)sq * B.seq_stride + S_FCA * B.tab_stride, B.f5o + (size_t)sq * B.ld, L, B.n[sq], B.ld, khi, klo, red);
}

template <int KD, int W, int FILT>
__global__ __launch_bounds__(64 * W, (W >= 8 ? 4 : 2)) void lin_outside_strip(McBatch B, const LinModel* __restrict__ L, const double* __restrict__ wT, int d0, int f5_hi,
                                                                              int f5_lo, int pin, int* __restrict__ bad)
{
    using P = OutStripPlan<KD, W>;
    constexpr int GS = P::GS, NM = P::NM, CD = P::CD, CA = P::CA, CE = P::CE, CS = P::CS, PADL = P::PADL;
    constexpr int NSL = KD / W;
    constexpr int TS = P::TS, KH = KD * TS / W;
    static_assert(KD % W == 0 && KD <= 8 && W % TS == 0 && KD % (W / TS) == 0 && (NM + 1) % W == 0, "8 terms per end and term set, 32/W staged rows per wavefront and region");
    __shared__ double lds[P::SZ];
    __shared__ double red[W];
    constexpr bool FACT = FILT != 0 && W == 8 && KD == 8;
    __shared__ double fw[FACT ? 160 : 1];   // factored filter: W4[t+1][4] (A, wb, -, Rc), read by wave-uniform (broadcast) LDS loads
    // pin = 2 (batch size a multiple of 8): workgroups are dealt round-robin to the 8 XCDs in launch order, so sequence sq is
    // pinned to XCD sq % 8 AND the groups of one sequence are consecutive on that XCD: neighbouring groups share 40 % of their
    // staged columns, which then come out of that XCD's L2 instead of HBM.  (Placement only: any mapping gives the same result.)
    int sq = pin ? blockIdx.x : blockIdx.y, slot = pin ? blockIdx.y : blockIdx.x;
    if (pin == 2) {
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x, t = lin >> 3;
        slot = (int)(t % gridDim.y);
        sq = (int)(t / gridDim.y) * 8 + (int)(lin & 7);
    }
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wt = w % TS, kb = (w / TS) * KH;
    const int ld = B.ld;
    const size_t ts = B.tab_stride;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    double* __restrict__ f5o = B.f5o + (size_t)sq * ld;
    const int dlow = d0 - (KD - 1);                        // the longest diagonal of the strip
    const int ncell_max = n - 1 - dlow > 0 ? n - 1 - dlow : 0;
    const int ngroup = (ncell_max + GS - 1) / GS;
    if (slot > ngroup) return;
    if (slot == ngroup) {
        f5o_range<W>(tab + S_FCA * ts, f5o, L, n, ld, f5_hi, f5_lo, red);
        return;
    }

#ifdef RH_STAGGER
    {   // tuning build: offset the second workgroup of every CU by about half a workgroup lifetime (lock-step rounds)
        const unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;
        if (lin >= 256 && lin < 512) for (int q = 0; q < RH_STAGGER; q++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    RH_STAMPO_BEGIN();
    const int i0 = 1 + slot * GS - (KD - 1);               // lane l <-> column i0+l; lanes >= KD-1 are this group's own columns
    const int i = i0 + lane;
    double* const LDm = lds;                               // fixed FM rows e = 1..
    double* const LA = lds + P::RD * CD;                   // zero rows + sliding FM2o rows
    double* const LE = LA + P::RA * CA;                    // sliding FCoX rows
    const double* __restrict__ fm = tab + S_FM * ts;
    const double* __restrict__ fm1 = tab + S_FM1 * ts;
    const double* __restrict__ fm2o = tab + S_FM2O * ts;
    const double* __restrict__ fcox = tab + S_FCOX * ts;

    // sequence letters of this wavefront's chain steps k = w + W*sl: the first loads issued, unconditional at clamped positions
    // (see the inside strip for why nothing here is loaded under a branch)
    int s_j[NSL], s_jp1[NSL], s_jp2[NSL];
    int s_im1, s_i, s_ip1;
    int rb0, rb1, rb2, r0[NSL], r1[NSL], r2[NSL];
    {
        const int top = B.lds - 1;
        const int c_i = i < 1 ? 1 : (i < top ? i : top - 1);
        rb0 = s[c_i - 1]; rb1 = s[c_i]; rb2 = s[c_i + 1];
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            const int j = i + d0 - (w + W * sl), cj = j < 0 ? 0 : (j + 2 <= top ? j : top - 2);
            r0[sl] = s[cj]; r1[sl] = s[cj + 1]; r2[sl] = s[cj + 2];
        }
    }

    // ---- global loads of everything this wavefront stages or keeps, issued back to back.  No address clamps: a column left of
    // or past its row, or a row past its table, still lies inside this sequence's table block (rows >= 1 here), and whatever
    // is read there is replaced by 0 on the way into LDS.  Row tails share loads as in the inside strip.
    constexpr int NR = (NM + 1) / W;
    constexpr int NT = (NM + 1) / TS;
    static_assert(NR % 2 == 0 && NR <= 8, "row tails are paired");
    double vA0[NR], vA1[NR / 2], vD0[NR], vD1, vE0[NR], vE1[NR / 2], a_lo[NT], a_hi[NT];
    const int lhalf = lane >> 5, lsub = lane & 31;
    const int cA0 = i0 - 31 + lane, cA1 = i0 - 31 + 64 + lsub;       // A: columns i0-31 ..
    const int cE0 = i0 - 32 + lane, cE1 = i0 - 32 + 64 + lsub;       // E: columns i0-32 ..
    const int cD0 = i0 + d0 - (KD - 1) + lane;                        // D: columns i0+d0-(KD-1) ..
    const int dsel = lane >> 3, dsub = lane & 7;                      // D tails: row = dsel (< NR), 8 columns each
    const int eD = 1 + w + W * dsel, cD1 = i0 + d0 - (KD - 1) + 64 + dsub;
    // a group whose windows lie inside every staged row goes from HBM straight into LDS (see the inside strip)
    const bool full = RH_STRIP_DMA && W == 8 && i0 >= 33 && i0 + 96 <= n - d0 && i0 + d0 - (KD - 1) >= 1;   // (workgroup-uniform)
    if (full) {
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int blk = k * W + w, chunk = blk * 64 + lane;   // 16-byte chunk of the region
            if (chunk < NM * (CA / 2)) {                           // A: rows d0+1+row behind the KD zero rows, columns i0-31 ..
                const int row = chunk / (CA / 2), within = chunk - row * (CA / 2);
                __builtin_amdgcn_global_load_lds((gptr_t)(fm2o + (unsigned)((d0 + 1 + row) * ld + i0 - 31 + 2 * within)), (lptr_t)(LA + KD * CA + blk * 128), 16, 0, 0);
            }
            {                                                      // E: rows d0+1+rho, columns i0-32 ..  (32 x 48 chunks = 3 x 512)
                const int row = chunk / (CE / 2), within = chunk - row * (CE / 2);
                __builtin_amdgcn_global_load_lds((gptr_t)(fcox + (unsigned)((d0 + 1 + row) * ld + i0 - 32 + 2 * within)), (lptr_t)(LE + blk * 128), 16, 0, 0);
            }
            if (chunk < NM * (CD / 2)) {                           // D: rows e = 1 + row, columns i0+d0-(KD-1) ..
                const int row = chunk / (CD / 2), within = chunk - row * (CD / 2), e = row + 1;
                __builtin_amdgcn_global_load_lds((gptr_t)(fm + (unsigned)(e * ld + i0 + d0 - (KD - 1) + 2 * within)), (lptr_t)(LDm + blk * 128), 16, 0, 0);
            }
        }
    } else {
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int ra = w + W * q;                                     // A row d0+1+ra (slot 31: no row), E row d0+1+rho, rho = ra
        vA0[q] = fm2o[(unsigned)((d0 + 1 + ra) * ld + cA0)];
        vE0[q] = fcox[(unsigned)((d0 + 1 + ra) * ld + cE0)];
        const int e = 1 + w + W * q, ee = e <= NM ? e : NM;           // D row e
        vD0[q] = fm[(unsigned)(ee * ld + cD0)];
    }
#pragma unroll
    for (int p2 = 0; p2 < NR / 2; p2++) {                             // lanes 0..31: row ra, lanes 32..63: row ra+W
        const int ra = w + W * 2 * p2;
        vA1[p2] = fm2o[(unsigned)((d0 + 1 + ra + lhalf * W) * ld + cA1)];
        vE1[p2] = fcox[(unsigned)((d0 + 1 + ra + lhalf * W) * ld + cE1)];
    }
    vD1 = fm[(unsigned)((eD <= NM ? eD : NM) * ld + cD1)];
    }
    // SPLIT (8 wavefronts): wavefronts 0..3 take the FMo sums of ALL eight diagonals and load only FM1[e][i-e], wavefronts 4..7 the FM1o sums and
    // only FM2o[d0+e][i] -- half the loads of the (term set, diagonal half) split, in which the wavefronts w and w+4 fetched the same sixteen row
    // segments at the same time; the partial sums land in the same PART slots.  One load per term, base and mask selected by role (no branch)
    constexpr bool SPLIT = W == 8 && KD == 8;
    const bool hi_role = SPLIT && w >= TS;
#pragma unroll
    for (int q = 0; q < NT; q++) {   // this term set's own-column values: FM1[e][i-e] (FMo terms), FM2o[d0+e][i] (FM1o terms), e = 1+wt+TS*q.
        // Slot e = 32 is no term: it is paired with zero rows below.
        const int e = 1 + wt + TS * q, R = d0 + e;
        // (eager '&': with '&&' the compiler turns the select into a branch, sinks the load into it and waits for it there --
        //  one exposed round trip per value)
        if constexpr (SPLIT) {
            a_lo[q] = 0.0; a_hi[q] = 0.0; (void)R;   // (loaded further down, as the LAST loads in front of the LDS-write phase)
        } else {
            const double v = fm1[(unsigned)(e * ld + i - e)], u = fm2o[(unsigned)(R * ld + i)];
            a_lo[q] = ((i - e >= 1) & (i <= n - 1)) ? v : 0.0;       // cell (i-e, i)
            a_hi[q] = ((i >= 1) & (i <= n - 1 - R)) ? u : 0.0;       // cell (i, i+R)
        }
    }
    double fwv = 0.0;
    if constexpr (FACT) fwv = wT[kStripFiltOffD + (threadIdx.x < 160 ? threadIdx.x : 159)];   // (with the staging loads: no round trip of its own)
    // the letters are first touched HERE, behind the staging loads
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+v"(rb0)); asm volatile("" : "+v"(rb1)); asm volatile("" : "+v"(rb2));
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) { asm volatile("" : "+v"(r0[sl])); asm volatile("" : "+v"(r1[sl])); asm volatile("" : "+v"(r2[sl])); }
    {
        const bool vi = i >= 1 && i <= n;
        s_im1 = vi ? rb0 : 4; s_i = vi ? rb1 : 4; s_ip1 = vi ? rb2 : 4;
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            const int d = d0 - (w + W * sl);
            const bool v = i >= 1 && i <= n - 1 - d;
            s_j[sl] = v ? r0[sl] : 4; s_jp1[sl] = v ? r1[sl] : 4; s_jp2[sl] = v ? r2[sl] : 4;
        }
    }
    // ---- operands of this wavefront's chain steps, raw: issued behind the staging loads, consumed after the filter
    double r_tjb[NSL], r_tja[NSL], r_tst[NSL], r_bp[NSL], r_tjbd[NSL], r_tjad[NSL], r_b01[NSL], r_b10[NSL], r_11[NSL];
    double p_far_m[NSL], p_far_1[NSL], p_fc[NSL], p_f5o[NSL];
    double p_x01, p_x10, p_x11, p_fmo, p_fm1o, p_fm1o_up, p_fco_up;   // only step k < 4 reads rows > d0 here
    const int ic = i < 1 ? 1 : i;
    double p_f5i;
    double r_z;
    {
        const double f5v = f5i[ic - 1 <= n ? ic - 1 : n];
        p_f5i = ((i >= 1) & (i - 1 <= n)) ? f5v : 0.0;
        r_z = 1.0 / f5i[n];   // (one division per lane, not one per diagonal on the chain)
#pragma unroll
        for (int sl = 0; sl < NSL; sl++) {
            const int d = d0 - (w + W * sl), j = i + d;
            const bool v = i >= 1 && i <= n - 1 - d;
            const int idx = 25 * (5 * s_i + s_ip1) + 5 * s_jp1[sl] + s_j[sl];
            const int idd = 25 * (5 * s_jp1[sl] + s_jp2[sl]) + 5 * s_i + s_im1;
            r_tjb[sl] = L->TJB[idx]; r_tja[sl] = L->TJA[idx];
            r_tst[sl] = L->TST[25 * (5 * s_im1 + s_i) + 5 * s_jp2[sl] + s_jp1[sl]];
            r_bp[sl] = L->E_bp[s_i * 5 + s_jp1[sl]];
            r_tjbd[sl] = L->TJB[idd]; r_tjad[sl] = L->TJA[idd];
            r_b01[sl] = L->E_b01[s_jp2[sl]]; r_b10[sl] = L->E_b10[s_im1]; r_11[sl] = L->E_11[s_im1 * 5 + s_jp2[sl]];
            const unsigned at = (unsigned)(d * ld + ic);
            const double fmof = tab[S_FMOF * ts + at], fm1of = tab[S_FM1OF * ts + at], fcv = tab[S_FC * ts + at];
            const double f5ov = f5o[j + 1 < 0 ? 0 : (j + 1 <= n ? j + 1 : n)];
            p_far_m[sl] = v ? fmof : 0.0;
            p_far_1[sl] = v ? fm1of : 0.0;
            p_fc[sl] = v ? fcv : 0.0;
            p_f5o[sl] = v ? f5ov : 0.0;
        }
    }
    // SPLIT: the FMo sums' own-column operands FM1[e][i-e] of wavefronts 0..3.  Wavefronts 4..7 need FM2o[d0+e][i], which is column i of
    // staged row d0+e: they read it from LDS behind the staging barrier.  These loads therefore sit under a wave-uniform condition, and
    // a conditional load is awaited where the branches join: placed among the other loads that cost 5 - 10 % of the sweep (the loads
    // behind it went out late); placed HERE, with nothing behind it but the wait for the staged rows, it costs nothing.
    if constexpr (SPLIT) {
        if (!hi_role) {
#pragma unroll
            for (int q = 0; q < NT; q++) {
                const int e = 1 + wt + TS * q;
                const double x = fm1[(unsigned)(e * ld + i - e)];
                a_lo[q] = ((i - e >= 1) & (i <= n - 1)) ? x : 0.0;
            }
        }
    }
    // ---- staged rows -> LDS (cells outside the interior of their row are staged as 0); unconditional stores, pinned values
    double* const LX = lds + P::OFF_DUMMY;
    if (!full) {
#pragma unroll
    for (int q = 0; q < NR; q++) { asm volatile("" : "+v"(vA0[q])); asm volatile("" : "+v"(vE0[q])); asm volatile("" : "+v"(vD0[q])); }
#pragma unroll
    for (int q = 0; q < NR / 2; q++) { asm volatile("" : "+v"(vA1[q])); asm volatile("" : "+v"(vE1[q])); }
    asm volatile("" : "+v"(vD1));
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int ra = w + W * q, e = 1 + w + W * q, R = d0 + 1 + ra;
        double* const dA = ra < NM ? LA + (KD + ra) * CA : LX;
        double* const dD = e <= NM ? LDm + (e - 1) * CD : LX;
        dA[lane] = ((cA0 >= 1) & (cA0 <= n - 1 - R)) ? vA0[q] : 0.0;
        LE[ra * CE + lane] = ((cE0 >= 1) & (cE0 <= n - 1 - R)) ? vE0[q] : 0.0;
        dD[lane] = ((cD0 >= 1) & (cD0 <= n - 1 - e)) ? vD0[q] : 0.0;
    }
#pragma unroll
    for (int p2 = 0; p2 < NR / 2; p2++) {
        const int rl = w + W * 2 * p2 + lhalf * W, R = d0 + 1 + rl;
        double* const dA = rl < NM ? LA + (KD + rl) * CA : LX;
        dA[64 + lsub] = ((cA1 >= 1) & (cA1 <= n - 1 - R)) ? vA1[p2] : 0.0;
        LE[rl * CE + 64 + lsub] = ((cE1 >= 1) & (cE1 <= n - 1 - R)) ? vE1[p2] : 0.0;
    }
    {
        double* const dD = (dsel < NR && eD <= NM) ? LDm + (eD - 1) * CD : LX;
        dD[64 + dsub] = ((cD1 >= 1) & (cD1 <= n - 1 - eD)) ? vD1 : 0.0;
    }
    }
    for (int k = threadIdx.x; k < KD * CA; k += 64 * W) LA[k] = 0.0;
    for (int k = threadIdx.x; k < (P::RD - NM) * CD; k += 64 * W) LDm[NM * CD + k] = 0.0;
    if constexpr (FACT) { asm volatile("" : "+v"(fwv)); if (threadIdx.x < 160) fw[threadIdx.x] = fwv; }
    RH_STAMPO(0);
    __syncthreads();
    RH_STAMPO(1);

    // ---- pre-phase: the terms whose operands were final before the launch, this wavefront's KH diagonals k = kb ..
    double accm[KH], acc1[KH], accg[KH];
#pragma unroll
    for (int k = 0; k < KH; k++) { accm[k] = 0.0; acc1[k] = 0.0; accg[k] = 0.0; }
    double acc8[SPLIT ? KD : 1];   // SPLIT: this wavefront's sum (FMo or FM1o) of all eight diagonals
    if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < KD; k++) acc8[k] = 0.0;
        const lds_vp vLA = (lds_vp)LA;
        const lds_vp vLD = (lds_vp)LDm;
        double vx[KD], nx[KD];
        if (!hi_role) {   // FMo: FM1[e][i-e] * FM2o[d0-k+e][i-e]: staged row KD + e-k-1, column index lane-e+31 (slot e = 32 reads zero rows only)
            auto issue = [&](int q, double* r) {
                const int e = 1 + wt + TS * q;
                const lds_vp pa = vLA + (e <= NM ? KD + e - 1 : KD - 1) * CA + lane - (e <= NM ? e : NM) + 31;
#pragma unroll
                for (int k = 0; k < KD; k++) r[k] = pa[-k * CA];
            };
            issue(0, vx);
#pragma unroll
            for (int q = 0; q < NT; q++) {
                if (q + 1 < NT) issue(q + 1, nx);
                const double a = a_lo[q];
#pragma unroll
                for (int k = 0; k < KD; k++) acc8[k] = fma(a, vx[k], acc8[k]);
#pragma unroll
                for (int k = 0; k < KD; k++) asm volatile("" : "+v"(acc8[k]));
#pragma unroll
                for (int k = 0; k < KD; k++) vx[k] = nx[k];
            }
        } else {          // FM1o: FM2o[d0+x][i] * FM[x+k][i+d0-k]: staged row x+k-1 (rows e > 31 are zero), column index lane + (KD-1) - k
            auto issue = [&](int q, double* r) {
                const int x = 1 + wt + TS * q;
                const lds_vp pd = vLD + (x - 1) * CD + lane + (KD - 1);
#pragma unroll
                for (int k = 0; k < KD; k++) r[k] = pd[k * (CD - 1)];
            };
            issue(0, vx);
#pragma unroll
            for (int q = 0; q < NT; q++) {
                if (q + 1 < NT) issue(q + 1, nx);
                const int x = 1 + wt + TS * q;
                const double b = x <= NM ? vLA[(KD + x - 1) * CA + lane + 31] : 0.0;   // FM2o[d0+x][i], staged (0 outside its row)
#pragma unroll
                for (int k = 0; k < KD; k++) acc8[k] = fma(b, vx[k], acc8[k]);
#pragma unroll
                for (int k = 0; k < KD; k++) asm volatile("" : "+v"(acc8[k]));
#pragma unroll
                for (int k = 0; k < KD; k++) vx[k] = nx[k];
            }
        }
    } else {
        const lds_vp vLA = (lds_vp)LA;
        const lds_vp vLD = (lds_vp)LDm;
        double va[KH], vb[KH], na[KH], nb[KH];
        auto issue = [&](int q, double* ra, double* rb) {
            // FMo: FM1[e][i-e] * FM2o[d0-k+e][i-e]; that row sits in staged row KD + e-k-1 (rows < KD are zero: the strip's own rows; slot
            // e = 32 reads zero rows only), column index lane-e+31
            const int e = 1 + wt + TS * q;
            const lds_vp pa = vLA + ((e <= NM ? KD + e - 1 : KD - 1) - kb) * CA + lane - (e <= NM ? e : NM) + 31;
            // FM1o: FM2o[d0+x][i] * FM[x+k][i+d0-k]: staged row x+k-1 (rows e > 31 are zero), column index lane + (KD-1) - k
            const int x = e;
            const lds_vp pd = vLD + (x - 1 + kb) * CD + lane + (KD - 1) - kb;
#pragma unroll
            for (int k = 0; k < KH; k++) { ra[k] = pa[-k * CA]; rb[k] = pd[k * (CD - 1)]; }
        };
        issue(0, va, vb);
#pragma unroll
        for (int q = 0; q < NT; q++) {
            if (q + 1 < NT) issue(q + 1, na, nb);
            const double a = a_lo[q], b = a_hi[q];
#pragma unroll
            for (int k = 0; k < KH; k++) { accm[k] = fma(a, va[k], accm[k]); acc1[k] = fma(b, vb[k], acc1[k]); }
#pragma unroll
            for (int k = 0; k < KH; k++) { asm volatile("" : "+v"(accm[k])); asm volatile("" : "+v"(acc1[k])); }
#pragma unroll
            for (int k = 0; k < KH; k++) { va[k] = na[k]; vb[k] = nb[k]; }
        }
    }
    RH_STAMPO(2);
    // FM1[e][i-e], e = 1..KD-1 (operands of the chain's own-row FMo terms) are a_lo[0], a_lo[1] of the wavefronts with kb == 0 (e = 1+wt, 5+wt):
    // handed to every wavefront through LDS behind the filter (F1S) instead of seven more loads per wavefront
    const double f1keep0 = a_lo[0], f1keep1 = a_lo[1];
    {   // (loaded here, behind the pre-phase, rather than with the other gathers: seven values fewer in registers while the staged rows
        //  are in flight -- the kernel is at 128 VGPRs, and a spilled gather is waited for where it is spilled)
        // the first of its steps may still need rows > d0 (unconditional loads, selected afterwards)
        const int k0 = w, d = d0 - k0, j = i + d;
        const bool v = i >= 1 && i <= n - 1 - d;
        const bool right_ok = v && j + 1 <= n - 1, left_ok = v && i - 1 >= 1;
        const unsigned a1 = (unsigned)((d + 1) * ld + ic), a2 = (unsigned)((d + 2) * ld + ic), a3 = (unsigned)((d + 3) * ld + ic), a4 = (unsigned)((d + 4) * ld + ic);
        // (only the wavefronts that finish steps 0..3 read rows above d0: a wave-uniform nest, see the inside strip)
        double l_fmo = 0.0, l_fm1o = 0.0, l_fm1o_up = 0.0, l_fco_up = 0.0, l_x01 = 0.0, l_x10 = 0.0, l_x11 = 0.0;
        if (k0 < 4) {
            l_x11 = fcox[a4 - 2];
            if (k0 < 3) {
                l_x01 = fcox[a3 - 1]; l_x10 = fcox[a3 - 2];
                if (k0 < 2) {
                    l_fm1o_up = tab[S_FM1O * ts + a2 - 1]; l_fco_up = tab[S_FCO * ts + a2 - 1];          // ipp:3828
                    if (k0 < 1) { l_fmo = tab[S_FMO * ts + a1]; l_fm1o = tab[S_FM1O * ts + a1 - 1]; }    // FMo[d+1][i], FM1o[d+1][i-1]   ipp:3806, 3833
                }
            }
        }
        p_fmo = (right_ok & (k0 < 1)) ? l_fmo : 0.0;
        p_fm1o = (left_ok & (k0 < 1)) ? l_fm1o : 0.0;
        const bool up_ok = left_ok && right_ok;                                                          // the cell (i-1, j+1) is interior
        p_fm1o_up = (up_ok & (k0 < 2)) ? l_fm1o_up : 0.0;
        p_fco_up = (up_ok & (k0 < 2)) ? l_fco_up : 0.0;
        p_x01 = (left_ok & (j + 2 <= n - 1) & (k0 < 3)) ? l_x01 : 0.0;
        p_x10 = (v & (i - 2 >= 1) & (j + 1 <= n - 1) & (k0 < 3)) ? l_x10 : 0.0;
        p_x11 = (v & (i - 2 >= 1) & (j + 2 <= n - 1) & (k0 < 4)) ? l_x11 : 0.0;
        }
    RH_STAMPO(3);
    // enclosing single-branch loops: staged row rho is table row d0+1+rho = d+2+t with t = rho-1+k for diagonal d0-k; tap l1 reads column
    // i-1-l1 of it (index lane+31-l1).  Weights as in the inside strip: wT[l1][rho+k].
    if constexpr (FACT) {
        // factored filter (see filt_factored), mirrored: accg[m] is the sum of diagonal d0-(k0+2m) for the cell in column i+m
        filt_factored_any<true, CE>(wt + w / TS, (lds_vp)LE + wt * CE + lane, (lds_vp)fw, wT + kStripFiltOffD, accg);
    } else {
        constexpr int HR = NT / 2;
        const lds_vp xrow = (lds_vp)LE + wt * CE + lane + 31;
#pragma unroll 1
        for (int l1 = 0; l1 <= kMaxSingle; l1++) {
            const double* __restrict__ wp = wT + l1 * kWTS + wt + kb;
            if (l1 <= wt + (HR - 1) * TS + kb + KH - 2) {
                double x[HR];
#pragma unroll
                for (int q = 0; q < HR; q++) x[q] = xrow[q * TS * CE - l1];
#pragma unroll
                for (int q = 0; q < HR; q++)
#pragma unroll
                    for (int k = 0; k < KH; k++) accg[k] = fma(wp[TS * q + k], x[q], accg[k]);
            }
            {
                double x[HR];
#pragma unroll
                for (int q = 0; q < HR; q++) x[q] = xrow[(q + HR) * TS * CE - l1];
#pragma unroll
                for (int q = 0; q < HR; q++)
#pragma unroll
                    for (int k = 0; k < KH; k++) accg[k] = fma(wp[TS * (q + HR) + k], x[q], accg[k]);
            }
        }
    }
    double cw[(KD - 1) * (KD - 2) / 2];   // single-branch weights of the chain's own-row taps (t <= KD-3), wave-uniform
#pragma unroll
    for (int k = 0; k < (KD - 1) * (KD - 2) / 2; k++) cw[k] = L->shape_w[k];
    // chain coefficients from the raw gathers (0 for a cell that is no pair: its FCo is 0)
    double p_cbd[NSL], p_cad[NSL], p_cst[NSL], p_ctja[NSL], p_tjbx[NSL], p_c01[NSL], p_c10[NSL], p_c11[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) {
        const int d = d0 - (w + W * sl);
        const bool pr = i >= 1 && d >= 0 && i <= n - 1 - d && pairs_s(s_i, s_jp1[sl]);
        p_cbd[sl] = pr ? r_bp[sl] * r_tjbd[sl] : 0.0;      // e_bp * e_tjbd
        p_cad[sl] = pr ? r_bp[sl] * r_tjad[sl] : 0.0;      // e_bp * e_tjad
        p_cst[sl] = pr ? r_tst[sl] * L->lam2 : 0.0;
        p_ctja[sl] = r_tja[sl] * L->e_mpmb;
        p_tjbx[sl] = r_tjb[sl];
        p_c01[sl] = L->w01 * r_b01[sl];
        p_c10[sl] = L->w10 * r_b10[sl];
        p_c11[sl] = L->w11 * r_11[sl];
    }
    RH_STAMPO(4);
    __syncthreads();   // every wavefront is done with the staged rows
    RH_STAMPO(5);
    double* const PART = lds + P::OFF_PART;
    double* const SFM2O = lds + P::OFF_S;
    double* const SFMO = SFM2O + KD * CS;
    double* const SFM1O = SFMO + KD * CS;
    double* const SFCO = SFM1O + KD * CS;
    double* const SFCOX = SFCO + KD * CS;
    if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < KD; k++) PART[((wt * KD + k) * 3 + (hi_role ? 1 : 0)) * 64 + lane] = acc8[k];
    }
#pragma unroll
    for (int k = 0; k < KH; k++) {
        if constexpr (!SPLIT) {
            PART[((wt * KD + kb + k) * 3 + 0) * 64 + lane] = accm[k];
            PART[((wt * KD + kb + k) * 3 + 1) * 64 + lane] = acc1[k];
        }
        if constexpr (FACT) {   // accg[k]: diagonal d0-(k0+2k) of column i+k; lanes > 63-k hold cells of the group to the right
            if (lane + k <= 63) PART[((wt * KD + w / TS + 2 * k) * 3 + 2) * 64 + lane + k] = accg[k];
        } else {
            PART[((wt * KD + kb + k) * 3 + 2) * 64 + lane] = accg[k];
        }
    }
    double* const F1S = lds + P::OFF_S + 5 * KD * CS;   // [KD-1][64]: behind the strip's own rows
    static_assert(P::OFF_S + 5 * KD * CS + (KD - 1) * 64 <= P::SZ, "F1S does not fit");
    if (kb == 0) {
        F1S[wt * 64 + lane] = f1keep0;                              // e = 1 + wt
        if (4 + wt < KD - 1) F1S[(4 + wt) * 64 + lane] = f1keep1;   // e = 5 + wt <= 7
    }
    for (int k = threadIdx.x; k < 5 * KD; k += 64 * W) {   // the left pad of the strip rows (read by lanes < 8 only)
#pragma unroll
        for (int c = 0; c < PADL; c++) SFM2O[k * CS + c] = 0.0;
    }
    __syncthreads();

    double f1e[KD - 1];
#pragma unroll
    for (int e = 1; e < KD; e++) f1e[e - 1] = F1S[(e - 1) * 64 + lane];
    // ---- chain (see the inside strip): time slot T: wavefront T % W finishes diagonal d0-T, the next wavefront gathers for d0-T-1
    const double w_mu = L->w_mu, w_mp2 = L->w_mp2, w_ep2 = L->w_ep2;
    double sms[NSL], s1s[NSL], gs[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) {
        const int K = w + W * sl;
        double sm = p_far_m[sl], s1 = p_far_1[sl], g = 0.0;
#pragma unroll
        for (int q = 0; q < TS; q++) {
            sm += PART[((q * KD + K) * 3 + 0) * 64 + lane];
            s1 += PART[((q * KD + K) * 3 + 1) * 64 + lane];
            g += PART[((q * KD + K) * 3 + 2) * 64 + lane];
        }
        sms[sl] = sm; s1s[sl] = s1; gs[sl] = g;
    }
    // (the posterior of diagonal K later reuses partial-sum row (0, K, 0): read above and written in fin<K> by the same wavefront)
    // the terms of step K that touch the strip's own row R <= K-2 (e = K-R in both sums; filter: t = K-2-R)
    auto addrow = [&](auto RC, auto KC) {
        constexpr int R = decltype(RC)::value, K = decltype(KC)::value, SL = K / W, e = K - R, t = K - 2 - R;
        double g0 = 0.0, g1 = 0.0;
        sms[SL] = fma(f1e[e - 1], SFM2O[R * CS + PADL + lane - e], sms[SL]);                             // FM1[e][i-e] * FM2o[d0-R][i-e]
        s1s[SL] = fma(SFM2O[R * CS + PADL + lane], LDm[(e - 1) * CD + lane + (KD - 1) - K], s1s[SL]);     // FM2o[d0-R][i] * FM[e][j]
#pragma unroll
        for (int l1 = 0; l1 <= t; l1++) {
            const double x = SFCOX[R * CS + PADL + lane - 1 - l1];
            if (l1 & 1) g1 = fma(cw[t * (t + 1) / 2 + l1], x, g1); else g0 = fma(cw[t * (t + 1) / 2 + l1], x, g0);
        }
        gs[SL] += g0 + g1;
    };
    double fco_s[NSL];
    auto fin = [&](auto KC) {
        constexpr int K = decltype(KC)::value, SL = K / W;
        const int d = d0 - K;
        const bool v = i >= 1 && d >= 0 && i <= n - 1 - d;
        double sm = sms[SL], s1 = s1s[SL];
        const double g = gs[SL];
        if constexpr (K >= 1) {   // the two terms that touch row K-1 (e = 1)
            sm = fma(f1e[0], SFM2O[(K - 1) * CS + PADL + lane - 1], sm);
            s1 = fma(SFM2O[(K - 1) * CS + PADL + lane], LDm[lane + (KD - 1) - K], s1);
        }
        double x01 = p_x01, x10 = p_x10, x11 = p_x11, o_fmo = p_fmo, o_fm1o = p_fm1o, o_fm1o_up = p_fm1o_up, o_fco_up = p_fco_up;
        if constexpr (K >= 1) { o_fmo = SFMO[(K - 1) * CS + PADL + lane]; o_fm1o = SFM1O[(K - 1) * CS + PADL + lane - 1]; }
        if constexpr (K >= 2) { o_fm1o_up = SFM1O[(K - 2) * CS + PADL + lane - 1]; o_fco_up = SFCO[(K - 2) * CS + PADL + lane - 1]; }
        if constexpr (K >= 3) { x01 = SFCOX[(K - 3) * CS + PADL + lane - 1]; x10 = SFCOX[(K - 3) * CS + PADL + lane - 2]; }
        if constexpr (K >= 4) x11 = SFCOX[(K - 4) * CS + PADL + lane - 2];
        double fmo = 0.0, fm1o = 0.0;
        if (d >= 2) {
            fmo = sm + o_fmo * w_mu;                      // ipp:3806
            fm1o = s1 + fmo + o_fm1o * w_mu;              // ipp:3809, 3833
        }
        const double ext = p_f5o[SL] * p_f5i * w_ep2;     // exterior loop, ipp:3768-3776
        const double multi = o_fm1o_up * w_mp2;           // branch of a multiloop, ipp:3828
        const double sp = p_c01[SL] * x01 + p_c10[SL] * x10 + p_c11[SL] * x11;
        double fco = p_cad[SL] * (ext + multi) + p_cbd[SL] * (g + sp) + p_cst[SL] * o_fco_up;   // coefficients are 0 for a non-pair
        double fm2o = fmo + fco * p_ctja[SL];                                                  // ipp:3803, 4027
        if (!v) { fco = 0.0; fmo = 0.0; fm1o = 0.0; fm2o = 0.0; }
        SFM2O[K * CS + PADL + lane] = fm2o; SFMO[K * CS + PADL + lane] = fmo; SFM1O[K * CS + PADL + lane] = fm1o;
        SFCO[K * CS + PADL + lane] = fco; SFCOX[K * CS + PADL + lane] = fco * p_tjbx[SL];
        fco_s[SL] = fco;   // the posterior is taken from it after the chain, off the critical path
    };
    RH_STAMPO(6);
    // time slot T: wavefront T % W finishes step T (fin: the two terms of row T-1 + the epilogue); every later step K > T, on
    // its own wavefront, adds the terms of row T-1, which the previous slot completed
#define RH_SLOT(T)                                                                                  \
    if constexpr (T < KD) {                                                                         \
        if (w == T % W) fin(IC<T>{});                                                               \
        if constexpr (T >= 1 && T + 1 < KD)                                                         \
            static_for<KD>([&](auto KC) {                                                           \
                constexpr int K = decltype(KC)::value;                                              \
                if constexpr (K > T) { if (w == K % W) addrow(IC<(T >= 1 ? T - 1 : 0)>{}, KC); }    \
            });                                                                                     \
        if constexpr (T + 1 < KD) lds_barrier();                                                    \
    }
    RH_SLOT(0) RH_SLOT(1) RH_SLOT(2) RH_SLOT(3) RH_SLOT(4) RH_SLOT(5) RH_SLOT(6) RH_SLOT(7)
#undef RH_SLOT
    RH_STAMPO(7);
    // posterior of pair (i, j+1) = FCo * FCi / Z, clipped to [0,1] (ipp:4689-4827), by the wavefront that finished the diagonal; it goes
    // to partial-sum row (term set 0, diagonal K, 0), which that wavefront read before the chain
#pragma unroll
    for (int sl = 0; sl < NSL; sl++) {
        const int K = w + W * sl, d = d0 - K;
        const bool v = i >= 1 && d >= 0 && i <= n - 1 - d;
        double p = fco_s[sl] * p_fc[sl] * r_z;
        if (v && lane >= KD - 1 && (!(p == p) || p > 1e300)) { atomicOr(&bad[sq], 1); p = 0.0; }
        p = p > 1.0 ? 1.0 : (p < 0.0 ? 0.0 : p);
        if (!(p == p)) p = 0.0;
        PART[(K * 3) * 64 + lane] = p;
    }
    lds_barrier();
    // ---- the strip's rows go to HBM now, off the chain
    if (lane >= KD - 1 && i >= 1) {
#pragma unroll 1
        for (int r = w; r < 5 * KD; r += W) {
            const int tb = r / KD, k = r - tb * KD, d = d0 - k;   // S row order: FM2o, FMo, FM1o, FCo, FCoX
            const int slot_of[5] = {S_FM2O, S_FMO, S_FM1O, S_FCO, S_FCOX};
            if (d >= 0 && i <= n - 1 - d) tab[slot_of[tb] * ts + (size_t)d * ld + i] = SFM2O[r * CS + PADL + lane];
        }
    }
    // posterior: KD consecutive entries of row i of the triangular table, written by KD adjacent threads
    for (int t = threadIdx.x; t < 64 * KD; t += 64 * W) {
        const int c = t / KD, k = t - c * KD, ii = i0 + c, d = d0 - k;
        if (c >= KD - 1 && ii >= 1 && d >= 0 && ii <= n - 1 - d)
            B.bp[(size_t)sq * B.tri_stride + tri_off_s(n, ii) + (ii + d + 1)] = PART[(k * 3) * 64 + c];
    }
    RH_STAMPO(8);
}

template __global__ void lin_inside_strip<8, 4, 0>(McBatch, const LinModel*, const double*, int, int, double, int);
template __global__ void lin_inside_strip<8, 8, 0>(McBatch, const LinModel*, const double*, int, int, double, int);
template __global__ void lin_inside_strip<8, 8, 1>(McBatch, const LinModel*, const double*, int, int, double, int);
template __global__ void lin_outside_strip<8, 4, 0>(McBatch, const LinModel*, const double*, int, int, int, int, int*);
template __global__ void lin_outside_strip<8, 8, 0>(McBatch, const LinModel*, const double*, int, int, int, int, int*);
template __global__ void lin_outside_strip<8, 8, 1>(McBatch, const LinModel*, const double*, int, int, int, int, int*);

}  // namespace rh
