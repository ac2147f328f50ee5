// mccaskill.hip -- McCaskill inside / outside / posterior sweeps for gfx950.
//
// What the reference computes (CONTRAfold model, /root/reference/src/contrafold):
//   ComputeInside    InferenceEngine.ipp:3356-3722   (FM2 3384-3411, FC 3567-3627,
//                                                     FM1 3641-3657, FM 3669-3688, F5 3692-3717)
//   ComputeOutside   InferenceEngine.ipp:3731-4080   (push form, read-modify-write)
//   ComputePosterior InferenceEngine.ipp:4498-4828
//
// How it is organised here (not a translation):
//   * span wavefront: one launch per diagonal d = j-i; every cell of a diagonal for
//     every sequence of the batch is independent, one 64-lane wavefront per cell;
//   * outside is rewritten in PULL form on decreasing spans -- every target gathers
//     from already-final sources, so there are no atomics or RMW races:
//       FMo [i,j] = FMo[i,j+1]+b  (+)  (+)_{i'<i} FM2o[i',j] + FM1i[i',i]
//       FM1o[i,j] = FMo[i,j] (+) FM1o[i-1,j]+b (+) (+)_{j'>j} FM2o[i,j'] + FMi[j,j']
//       FCo [i,j] = exterior (+) multi (+) stack (+) (+)_{enclosing loops} FCo[i',j'] + score
//       FM2o[i,j] = FMo[i,j] (+) FCo[i,j] + ScoreJunctionA(i,j) + a + c
//   * every inner loop streams two contiguous rows (transposed mirrors are written
//     by the producing cell), so lanes read 512 B coalesced segments;
//   * the posterior needs no third sweep: the reference's sum over parent contexts
//     of exp(context + FCi - Z) (ipp:4689-4817) is exp(FCo + FCi - Z) by definition
//     of FCo, so it is emitted by the outside cell as soon as FCo is final.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lse.h"
#include "score_model.h"

namespace rh {

namespace {

// AU, CG, GU both ways (InferenceEngine.ipp:391-396); code 4 (unknown letter) never pairs
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
// ScoreJunctionB(i,j) (ipp:2004-2029): letters s[i], s[j+1], s[i+1], s[j]
__device__ __forceinline__ double junction_b(const ScoreModel* M, int si, int sj1, int si1, int sj)
{
    return M->helix_closing[si * 5 + sj1] + tm4(M, si, sj1, si1, sj);
}
// ScoreJunctionA(i,j) (ipp:1927-1956); sentinel code 4 zeroes the edge dangles
__device__ __forceinline__ double junction_a(const ScoreModel* M, int si, int sj1, int si1, int sj)
{
    return M->helix_closing[si * 5 + sj1] + M->dangle_left[si * 25 + sj1 * 5 + si1] +
           M->dangle_right[si * 25 + sj1 * 5 + sj];
}
// ScoreSingleNucleotides (ipp:2290-2360) for the three shapes with a nucleotide term,
// enclosing pair (i,j) [gap indices], shape (l1,l2)
__device__ __forceinline__ double single_nucs(const ScoreModel* M, int l1, int l2, int s_ip1, int s_j)
{
    double v = 0.0;
    if (l1 == 0 && l2 == 1) v = M->bulge_0x1[s_j];
    if (l1 == 1 && l2 == 0) v = M->bulge_1x0[s_ip1];
    if (l1 == 1 && l2 == 1) v = M->internal_1x1[s_ip1 * 5 + s_j];
    return v;
}

__device__ __forceinline__ size_t tri_offset(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }

}  // namespace

// ---------------------------------------------------------------------------------
// one-off per batch: F5i[0] = 0, F5o[n] = 0 (ipp:3690, 3750)
__global__ void mc_init(McBatch B)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    B.f5i[(size_t)sq * B.ld] = 0.0;
    B.f5o[(size_t)sq * B.ld + n] = 0.0;
}

// ---------------------------------------------------------------------------------
// Work distribution shared by the two sweeps.  A launch covers every sequence of the
// batch; `pin` selects the block->(sequence, wave slot) map:
//   pin = 1: blockIdx.x = sequence (fastest varying).  Workgroups are dealt round-robin
//            over the 8 XCDs by linear id, so with ns % 8 == 0 all blocks of a sequence
//            share one XCD and its 4 MiB L2 sees only ns/8 sequences' rows (speed only).
//   pin = 0: blockIdx.y = sequence: few, large sequences are spread over all XCDs.
__device__ __forceinline__ void block_map(int pin, int* sq, int* slot)
{
    *sq = pin ? blockIdx.x : blockIdx.y;
    *slot = pin ? blockIdx.y : blockIdx.x;
}

// ---------------------------------------------------------------------------------
// inside, diagonal d: cells (i, i+d), 1 <= i <= n-1-d, one wavefront each; the wave
// after the last cell computes F5i[d+1].
__global__ __launch_bounds__(256) void mc_inside_diag(McBatch B, const ScoreModel* __restrict__ M, int d, int pin)
{
    int sq, slot;
    block_map(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int wave = slot * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: lets the cell's scores use scalar loads
    const int lane = threadIdx.x & 63;
    const int ncell = n - 1 - d > 0 ? n - 1 - d : 0;
    if (d > n - 1 || wave > ncell) return;

    const int ld = B.ld;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const size_t ts = B.tab_stride;

    if (wave == ncell) {
        // F5i[jj] = F5i[jj-1]+ext_unpaired (+) (+)_{k<=jj-2} F5i[k] + ext_paired + FCA[k+1][jj-1]   (ipp:3692-3717)
        const int jj = d + 1;
        const double* __restrict__ fcat = tab + T_FCAT * ts + (size_t)(jj - 1) * ld;
        Lse acc = lse_empty();
        lse_stream2<4>(acc, f5i, fcat + 1, 0, jj - 1, lane);
        acc.m += M->external_paired;  // constant factor of every streamed term
        if (lane == 0) lse_add(acc, f5i[jj - 1] + M->external_unpaired);
        const double v = lse_wave_finish(acc);
        if (lane == 0) f5i[jj] = v;
        return;
    }

    const int i = wave + 1, j = i + d;
    const int s_im1 = s[i - 1], s_i = s[i], s_ip1 = s[i + 1], s_j = s[j], s_jp1 = s[j + 1], s_jp2 = s[j + 2];
    const bool pairable = complementary(s_i, s_jp1);
    const bool inner = d >= 2;  // cells of span 0/1 have no FM2, FM1, FM and no enclosed pair

    // O(1) operands of FM1 / FM / the stacking term: issued before the streams so that they are in flight with them
    double op_fca = kNeg, op_fm1 = kNeg, op_fm = kNeg, op_fc = kNeg;
    if (inner) {
        op_fca = tab[T_FCA * ts + (size_t)(i + 1) * ld + (j - 1)];
        op_fm1 = tab[T_FM1 * ts + (size_t)(i + 1) * ld + j];
        op_fm = tab[T_FM * ts + (size_t)i * ld + (j - 1)];
        op_fc = tab[T_FC * ts + (size_t)(i + 1) * ld + (j - 1)];
    }

    // ---- single-branch gather of FC[i,j]: up to 8 x 64 shapes, all gathers in flight (ipp:3597-3619)
    Lse acc_c = lse_empty();
    const double jb = junction_b(M, s_i, s_jp1, s_ip1, s_j);
    if (pairable && inner) {
        const int tmax = d - 2 < kMaxSingle ? d - 2 : kMaxSingle;
        const double* __restrict__ fcx = tab + T_FCX * ts + (size_t)(i + 1) * ld + (j - 1);
        Shape sh[kMcShapeIters];
#pragma unroll
        for (int u = 0; u < kMcShapeIters; u++) sh[u] = M->mc_shape[64 * u + lane];
        double x[kMcShapeIters];
#pragma unroll
        for (int u = 0; u < kMcShapeIters; u++) {
            const bool ok = sh[u].l1 + sh[u].l2 <= tmax;
            x[u] = ok ? fcx[(ptrdiff_t)sh[u].l1 * ld - sh[u].l2] : kEmptyMax;
        }
#pragma unroll
        for (int u = 0; u < kMcShapeIters; u++) x[u] += sh[u].score + jb;
        x[0] += single_nucs(M, sh[0].l1, sh[0].l2, s_ip1, s_j);
        if (lane == 0)  // shape (0,0) is the stacking pair: FC[i+1,j-1] + ScoreBasePair(i+1,j) + ScoreHelixStacking(i,j+1)
            x[0] = op_fc + M->base_pair[s_ip1 * 5 + s_j] + hs4(M, s_i, s_jp1, s_ip1, s_j);
        lse_add_group<kMcShapeIters>(acc_c, x);
    }
    if (pairable && lane == 0 && d >= kMinHairpin) lse_add(acc_c, jb + M->hairpin_len[d < 30 ? d : 30]);  // ScoreHairpin

    // ---- FM2[i,j] = (+)_{i<k<j} FM1[i,k] + FM[k,j]        (ipp:3384-3411)
    Lse acc_2 = lse_empty();
    if (inner)
        lse_stream2<8>(acc_2, tab + T_FM1 * ts + (size_t)i * ld, tab + T_FMT * ts + (size_t)j * ld, i + 1, j, lane);

    double fm2, fc;
    lse_wave_finish2(acc_2, acc_c, fm2, fc);
    // multiloop closed by (i,j+1): FM2 + ScoreJunctionA(i,j) + multi_paired + multi_base (ipp:3622)
    fc = pairable ? lse2(fc, fm2 + junction_a(M, s_i, s_jp1, s_ip1, s_j) + M->multi_paired + M->multi_base) : kNeg;

    // ---- FM1[i,j], FM[i,j]                                  (ipp:3641-3688)
    double fm1 = kNeg, fm = kNeg;
    if (inner) {
        fm1 = lse2(op_fca + M->multi_paired, op_fm1 + M->multi_unpaired);
        fm = lse3(fm2, op_fm + M->multi_unpaired, fm1);
    }

    if (lane == 0) {
        const size_t ij = (size_t)i * ld + j, ji = (size_t)j * ld + i;
        const double bp = M->base_pair[s_i * 5 + s_jp1];
        // as the inner pair (p+1,q) = (i,j+1) of an enclosing loop: ScoreJunctionB(q,p) = JB(j+1,i-1)
        const double dec_x = bp + junction_b(M, s_jp1, s_i, s_jp2, s_im1);
        // as a branch of a multi/exterior loop: ScoreJunctionA(j+1,i-1)
        const double dec_a = bp + junction_a(M, s_jp1, s_i, s_jp2, s_im1);
        tab[T_FC * ts + ij] = fc;
        tab[T_FCX * ts + ij] = pairable ? fc + dec_x : kNeg;
        const double fca = pairable ? fc + dec_a : kNeg;
        tab[T_FCA * ts + ij] = fca;
        tab[T_FCAT * ts + ji] = fca;
        tab[T_FM1 * ts + ij] = fm1;
        tab[T_FM1T * ts + ji] = fm1;
        tab[T_FM * ts + ij] = fm;
        tab[T_FMT * ts + ji] = fm;
    }
}

// ---------------------------------------------------------------------------------
// outside (pull form) + posterior, diagonal d: cells (i, i+d), 1 <= i <= n-1-d; the
// wave after the last cell computes F5o[d+1].
__global__ __launch_bounds__(256) void mc_outside_diag(McBatch B, const ScoreModel* __restrict__ M, int d, int pin)
{
    int sq, slot;
    block_map(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int wave = slot * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: lets the cell's scores use scalar loads
    const int lane = threadIdx.x & 63;
    const int ncell = n - 1 - d;
    if (ncell < 1 || wave > ncell) return;

    const int ld = B.ld;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    double* __restrict__ f5o = B.f5o + (size_t)sq * ld;
    const size_t ts = B.tab_stride;

    if (wave == ncell) {
        // F5o[k] = F5o[k+1]+ext_unpaired (+) (+)_{jj>=k+2} F5o[jj] + ext_paired + FCA[k+1][jj-1]   (ipp:3751-3780, pulled)
        const int k = d + 1;
        const double* __restrict__ fca = tab + T_FCA * ts + (size_t)(k + 1) * ld;
        Lse acc = lse_empty();
        lse_stream2<4>(acc, f5o + 1, fca, k + 1, n, lane);  // index t = jj-1
        acc.m += M->external_paired;
        if (lane == 0) lse_add(acc, f5o[k + 1] + M->external_unpaired);
        const double v = lse_wave_finish(acc);
        if (lane == 0) f5o[k] = v;
        return;
    }

    const int i = wave + 1, j = i + d;
    const int s_im1 = s[i - 1], s_i = s[i], s_ip1 = s[i + 1], s_j = s[j], s_jp1 = s[j + 1], s_jp2 = s[j + 2];
    const bool guard_m = d >= 2;
    const bool pairable = complementary(s_i, s_jp1);

    // O(1) operands, issued up front
    double op_fmo = kNeg, op_fm1o = kNeg, op_fm1o_up = kNeg, op_fco_up = kNeg, op_f5o = kNeg, op_f5i = kNeg;
    if (guard_m) {
        if (j + 1 <= n - 1) op_fmo = tab[T_FMO * ts + (size_t)i * ld + (j + 1)];          // FMo[i,j+1]   (ipp:3806)
        if (i - 1 >= 1) op_fm1o = tab[T_FM1O * ts + (size_t)(i - 1) * ld + j];            // FM1o[i-1,j]  (ipp:3833)
    }
    if (pairable) {
        op_f5o = f5o[j + 1];
        op_f5i = f5i[i - 1];
        if (i - 1 >= 1 && j + 1 <= n - 1) {
            op_fm1o_up = tab[T_FM1O * ts + (size_t)(i - 1) * ld + (j + 1)];               // FM1o[i-1,j+1] (ipp:3828)
            op_fco_up = tab[T_FCO * ts + (size_t)(i - 1) * ld + (j + 1)];                 // FCo[i-1,j+1]  (stacking)
        }
    }
    const double fc_in = tab[T_FC * ts + (size_t)i * ld + j];
    const double Z = f5i[n];

    // ---- enclosing single-branch loops of FCo[i,j] (ipp:4004-4024 pulled): source (i',j') = (i-1-l1, j+1+l2)
    Lse acc_c = lse_empty();
    const double bp = M->base_pair[s_i * 5 + s_jp1];
    const double ja_in = junction_a(M, s_jp1, s_i, s_jp2, s_im1);  // ScoreJunctionA(j+1,i-1)
    if (pairable) {
        const double dec = bp + junction_b(M, s_jp1, s_i, s_jp2, s_im1);  // BP(p+1,q) + JB(q,p)
        const int l1max = i - 2, l2max = n - 2 - j;
        if (l1max >= 0 && l2max >= 0) {
            const double* __restrict__ fcox = tab + T_FCOX * ts + (size_t)(i - 1) * ld + (j + 1);
            Shape sh[kMcShapeIters];
#pragma unroll
            for (int u = 0; u < kMcShapeIters; u++) sh[u] = M->mc_shape[64 * u + lane];
            double x[kMcShapeIters];
#pragma unroll
            for (int u = 0; u < kMcShapeIters; u++) {
                const bool ok = sh[u].l1 <= l1max && sh[u].l2 <= l2max;
                x[u] = ok ? fcox[sh[u].l2 - (ptrdiff_t)sh[u].l1 * ld] : kEmptyMax;
            }
#pragma unroll
            for (int u = 0; u < kMcShapeIters; u++) x[u] += sh[u].score + dec;
            // nucleotide terms of the 0x1 / 1x0 / 1x1 shapes: letters i'+1 and j' of the ENCLOSING loop
            if (sh[0].l1 <= 1 && sh[0].l2 <= 1 && sh[0].l1 <= l1max && sh[0].l2 <= l2max)
                x[0] += single_nucs(M, sh[0].l1, sh[0].l2, s[i - sh[0].l1], s[j + 1 + sh[0].l2]);
            if (lane == 0)  // stacked on (i-1,j+1): FCo + ScoreBasePair(i,j+1) + ScoreHelixStacking(i-1,j+2)
                x[0] = op_fco_up + bp + hs4(M, s_im1, s_jp2, s_i, s_jp1);
            lse_add_group<kMcShapeIters>(acc_c, x);
        }
        if (lane == 0) {
            // exterior loop (ipp:3768-3776): F5o[j+1] + ext_paired + BP + JA(j+1,i-1) + F5i[i-1]
            lse_add(acc_c, op_f5o + M->external_paired + bp + ja_in + op_f5i);
            // branch of a multiloop via FM1 (ipp:3828): FM1o[i-1,j+1] + JA + multi_paired + BP
            lse_add(acc_c, op_fm1o_up + ja_in + M->multi_paired + bp);
        }
    }

    // ---- FMo[i,j] column gather, FM1o[i,j] row gather (ipp:4046-4064 pulled)
    Lse acc_m = lse_empty(), acc_1 = lse_empty();
    if (guard_m) {
        lse_stream2<4>(acc_m, tab + T_FM2OT * ts + (size_t)j * ld, tab + T_FM1T * ts + (size_t)i * ld, 1, i, lane);
        lse_stream2<4>(acc_1, tab + T_FM2O * ts + (size_t)i * ld, tab + T_FM * ts + (size_t)j * ld, j + 1, n, lane);
        if (lane == 0) {
            lse_add(acc_m, op_fmo + M->multi_unpaired);
            lse_add(acc_1, op_fm1o + M->multi_unpaired);
        }
    }
    double fmo, fm1o, fco;
    lse_wave_finish3(acc_m, acc_1, acc_c, fmo, fm1o, fco);
    if (!guard_m) { fmo = kNeg; fm1o = kNeg; }
    else fm1o = lse2(fm1o, fmo);  // FM1o[i,j] (+)= FMo[i,j]  (ipp:3809)
    if (!pairable) fco = kNeg;

    // ---- FM2o[i,j] = FMo[i,j] (+) FCo[i,j] + ScoreJunctionA(i,j) + a + c   (ipp:3803, 4027)
    const double viafc = pairable ? fco + junction_a(M, s_i, s_jp1, s_ip1, s_j) + M->multi_paired + M->multi_base : kNeg;
    const double fm2o = lse2(fmo, viafc);

    if (lane == 0) {
        const size_t ij = (size_t)i * ld + j, ji = (size_t)j * ld + i;
        tab[T_FCO * ts + ij] = fco;
        tab[T_FCOX * ts + ij] = pairable ? fco + junction_b(M, s_i, s_jp1, s_ip1, s_j) : kNeg;
        tab[T_FMO * ts + ij] = fmo;
        tab[T_FM1O * ts + ij] = fm1o;
        tab[T_FM2O * ts + ij] = fm2o;
        tab[T_FM2OT * ts + ji] = fm2o;
        // posterior of pair (i, j+1): exp(FCo + FCi - Z), clipped to [0,1] (ipp:4689-4827)
        const double e = fco + fc_in - Z;
        double p = e > kNeg / 2 ? exp(e) : 0.0;
        p = p > 1.0 ? 1.0 : p;
        B.bp[(size_t)sq * B.tri_stride + tri_offset(n, i) + (j + 1)] = pairable ? p : 0.0;
    }
}

// ---------------------------------------------------------------------------------
// width-1 accessibility, /root/reference/src/ractip.cpp:213-222:
//   up[i] = max(0, 1 - sum_{j<i} bp(j,i) - sum_{j>i} bp(i,j)),  letters 1-based
// One workgroup per 64 letters a0+1 .. a0+64.  Column part (pairs (b, a), b < a): lane <-> column a, the four wavefronts take the
// rows b = 1+w, 5+w, ..: every load is a contiguous 512-byte run of a row of the triangular table (the per-letter form read the
// column with a stride of a row: one 128-byte line per element).  Row part (pairs (a, b), b > a): wavefront w takes the letters
// a0+1+w, +4, ..; lanes run along the row.  Fixed summation order: the bits of up do not depend on the batch.
__global__ __launch_bounds__(256) void mc_unpaired(McBatch B)
{
    __shared__ double colsum[4][64];
    __shared__ double rowsum[64];
    const int sq = blockIdx.y;
    const int n = B.n[sq];
    const int a0 = blockIdx.x * 64;
    if (a0 >= n) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double* __restrict__ bp = B.bp + (size_t)sq * B.tri_stride;
    {
        const int a = a0 + 1 + lane;
        double acc = 0.0;
        const int bmax = a0 + 64 < n ? a0 + 64 : n;   // rows b < a <= bmax
        for (int b = 1 + w; b < bmax; b += 4)
            if (b < a && a <= n) acc += bp[tri_offset(n, b) + a];
        colsum[w][lane] = acc;
    }
    for (int q = w; q < 64; q += 4) {
        const int a = a0 + 1 + q;
        double acc = 0.0;
        if (a <= n)
            for (int b2 = a + 1 + lane; b2 <= n; b2 += 64) acc += bp[tri_offset(n, a) + b2];
        acc = wave_sum(acc);
        if (lane == 0) rowsum[q] = acc;
    }
    __syncthreads();
    if (w == 0 && a0 + 1 + lane <= n)
        B.up[(size_t)sq * B.ld + a0 + lane] = fmax(0.0, 1.0 - (((colsum[0][lane] + colsum[1][lane]) + (colsum[2][lane] + colsum[3][lane])) + rowsum[lane]));
}

}  // namespace rh
