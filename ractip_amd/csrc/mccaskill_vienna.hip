// mccaskill_vienna.hip -- McCaskill inside / outside / posterior and region accessibility with the Vienna energy tables
// in the pf_fold / pf_unstru semantics of ViennaRNA 1.8 or, with the tables built for it (stemE / stemM, loop kinds 3 / 4,
// tri / hexaloops: vienna_model.h), of ViennaRNA 2.x with dangles = 2 (PARITY UNPINNED, see vienna_model.h and DESIGN.md).
//
// What it replaces: the third-party calls of RactIP::rnafold, /root/reference/src/ractip.cpp:288-304, 351-367
// (pf_fold + export_bppm -> bp) and :370-375 (pf_unstru -> up[i][w] = H+I+M+E = P(i..i+w unpaired)).
//
// Organisation: the span wavefront of mccaskill.hip (one launch per diagonal, one 64-lane wavefront per cell, log
// space, pull-form outside, posterior emitted by the outside cell), with
//   * Vienna loop energies: lanes <-> the 496 (l1,l2) shapes; generic interior loops read a table that already
//     carries the inner pair's mismatchI term, long bulges add TerminalAU per pair type, and the seven small shapes
//     with joint tables (stack, 1-bulges, int11, int21, int22) are looked up explicitly;
//   * an UNAMBIGUOUS multiloop grammar (the CONTRAfold one counts a multiloop part with >= 2 branches and trailing
//     unpaired letters more than once -- fine for reproducing that engine, wrong for a partition function):
//         FM1[i,j] = leading unpaired + one branch ending at j      FMS[i,j] = FM1[i,j] (+) FMS[i,j-1]+b
//         FM2[i,j] = (+)_k FM1[i,k] + FM[k,j]                      FM[i,j]  = FM2[i,j] (+) FMS[i,j]
//   * accessibility from the finished inside/outside tables: a run of unpaired letters a..b lies in exactly one loop,
//         E: F5i[a-1] F5o[b]      H: sum_{p<a,q>b} FCo(p,q) hairpin(p,q)     I: runs inside the left / right gap of an
//         interior loop           M: FM1o[a-1,j] b^len FM1[b,j]  (before a branch),  FMSo[i,b] b^len FMS[i,a-1] (trailing)
//     all divided by Z; the sums run over probabilities in linear space;
//   * the two-molecule ensemble of co_pf_fold (/root/reference/src/ractip.cpp:400-458: hp from the joint pair matrix of
//     s1+s2): McBatch::cut marks the backbone gap that does not exist.  A loop whose backbone holds it is
//     exterior-like -- for the pair (i,j+1) around it: XS[i+1] + XP[j] + the pair's own dangles, with XS / XP the
//     exterior partition functions of s1's suffixes / s2's prefixes, one extra wavefront each per diagonal -- no
//     hairpin, interior loop side, multiloop backbone or dangle may cross it.  cut = 0 reproduces the one-molecule case.
#include <hip/hip_runtime.h>

#include "batch.h"
#include "lse.h"
#include "vienna_model.h"

namespace rh {

// square ld x ld tables per sequence; *T = stored transposed ([j][i])
enum VmTable { VM_FC = 0, VM_FCX, VM_FCA, VM_FCAT, VM_FM1, VM_FM1T, VM_FM, VM_FMT, VM_FMST,
               VM_FCO, VM_FCOX, VM_FM2O, VM_FM2OT, VM_FMSOT, VM_FM1O, VM_FCOT, VM_COUNT };
// scratch slots of the accessibility pass (their sweep contents are dead by then)
constexpr int VM_S_FCT = VM_FCAT;    // FC transposed
constexpr int VM_S_FCOT = VM_FCOT;   // FCo transposed (written by the outside cells themselves)
constexpr int VM_S_HP = VM_FCX;      // hairpin probabilities [p][q] -> exclusive prefix sums over p
static_assert((int)VM_COUNT == kViennaMcTables, "batch.h and mccaskill_vienna.hip disagree on the table count");

namespace {

__device__ __forceinline__ void block_map_v(int pin, int* sq, int* slot)
{
    *sq = pin ? blockIdx.x : blockIdx.y;
    *slot = pin ? blockIdx.y : blockIdx.x;
}
__device__ __forceinline__ size_t tri_offset_v(int n, int i) { return (size_t)i * (size_t)(2 * (n + 1) - i - 1) / 2; }
// letters g and g+1 are neighbours on one strand
__device__ __forceinline__ bool gap_ok(int cut, int g) { return g != cut; }   // cut = 0 never equals a gap index >= 1
// the missing gap lies among gaps lo..hi
__device__ __forceinline__ bool nicked(int cut, int lo, int hi) { return cut > 0 && lo <= cut && cut <= hi; }

// index of shape (l1,l2) in the row-major shape list of ViennaDx
__device__ __forceinline__ int shape_index(int l1, int l2) { return l1 * 31 - l1 * (l1 - 1) / 2 + l2; }

// hairpin closed by letters (a, b), u = b-a-1 >= 3 unpaired letters   (part_func.c expHairpinEnergy)
__device__ __forceinline__ double hairpin_w(const ViennaDx* V, const uint8_t* s, int a, int u, int type)
{
    double e = u <= 30 ? V->hairpin[u] : V->hairpin30 - V->lxc * log(u / 30.0);
    if (u == 3 || u == 4 || u == 6) {   // tabulated loops: the table holds a bonus (1.8) or the difference to the plain energy (2.x)
        int code = 0;
        bool ok = true;
        for (int k = 0; k < u + 2; k++) { const int c = s[a + k]; ok = ok && c != 0; code = code * 4 + (c - 1); }
        if (ok && u == 3) e += V->tri[code];
        if (ok && u == 4) e += V->tetra[code];
        if (ok && u == 6)
            for (int k = 0; k < V->nhexa; k++) if (V->hexa_code[k] == code) e += V->hexa[k];
    }
    if (u == 3) return e + (type > 2 ? V->tau : 0.0);
    return e + V->mmH[type * 25 + s[a + 1] * 5 + s[a + u]];
}

// log weight of the interior loop with outer pair of type `to`, inner pair of type `ti`, gaps l1 / l2;
// a1/b1 = letters after the outer 5' letter / before the outer 3' letter, p1/q1 = before the inner 5' / after the inner 3'
__device__ __forceinline__ double loop_w(const ViennaDx* V, int l1, int l2, int to, int ti, int a1, int b1, int p1, int q1)
{
    const int idx = shape_index(l1, l2);
    const int kind = V->kind[idx];
    const int rti = V->rtype[ti];
    if (kind == 1) return V->shape[idx].score + V->mmI[to * 25 + a1 * 5 + b1] + V->mmI[rti * 25 + q1 * 5 + p1];
    if (kind == 2) return V->shape[idx].score + (to > 2 ? V->tau : 0.0) + (ti > 2 ? V->tau : 0.0);
    if (kind == 3) return V->shape[idx].score + V->mm1nI[to * 25 + a1 * 5 + b1] + V->mm1nI[rti * 25 + q1 * 5 + p1];
    if (kind == 4) return V->shape[idx].score + V->mm23I[to * 25 + a1 * 5 + b1] + V->mm23I[rti * 25 + q1 * 5 + p1];
    return vienna_small_loop(V, l1, l2, to, rti, a1, b1, p1, q1);
}

}  // namespace

__global__ void mcv_init(McBatch B)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    B.f5i[(size_t)sq * B.ld] = 0.0;
    B.f5o[(size_t)sq * B.ld + n] = 0.0;
    const int cut = B.cut ? B.cut[sq] : 0;
    if (cut > 0) {
        const size_t o = (size_t)sq * B.ld;
        B.xp[o + cut] = 0.0;          // empty prefix of s2
        B.xs[o + cut + 1] = 0.0;      // empty suffix of s1
        B.xpo[o + n] = kNeg;          // no pair (i, n+1) exists
        B.xso[o + 1] = kNeg;          // no pair (0, j) exists
        B.xso[o + cut + 1] = kNeg;
    }
}

// ---------------------------------------------------------------------------------
// inside, diagonal d: cells (i, i+d), 1 <= i <= n-1-d; the wave after the last cell computes F5i[d+1]
__global__ __launch_bounds__(256) void mcv_inside_diag(McBatch B, const ViennaDx* __restrict__ V, int d, int pin)
{
    int sq, slot;
    block_map_v(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int wave = slot * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int ncell = n - 1 - d > 0 ? n - 1 - d : 0;
    const int cut = B.cut ? B.cut[sq] : 0;
    if (d > n - 1 || wave > ncell + (cut > 0 ? 2 : 0)) return;

    const int ld = B.ld;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const size_t ts = B.tab_stride;

    if (wave == ncell + 1) {
        // XP[b], b = cut+d+1: exterior partition function of s2's prefix cut+1..b (its stems have spans <= d-1)
        const int b = cut + d + 1;
        if (b > n) return;
        double* __restrict__ xp = B.xp + (size_t)sq * ld;
        const double* __restrict__ fcat = tab + VM_FCAT * ts + (size_t)(b - 1) * ld;
        Lse acc = lse_empty();
        lse_stream2<4>(acc, xp, fcat + 1, cut, b - 1, lane);
        if (lane == 0) lse_add(acc, xp[b - 1]);
        const double v = lse_wave_finish(acc);
        if (lane == 0) xp[b] = v;
        return;
    }
    if (wave == ncell + 2) {
        // XS[a], a = cut-d: exterior partition function of s1's suffix a..cut: XS[a+1] (+) FCA[a][l-1] + XS[l+1]
        const int a = cut - d;
        if (a < 1) return;
        double* __restrict__ xs = B.xs + (size_t)sq * ld;
        const double* __restrict__ fca = tab + VM_FCA * ts + (size_t)a * ld;
        Lse acc = lse_empty();
        lse_stream2<4>(acc, fca, xs + 2, a + 3, cut, lane);   // index t = l-1
        if (lane == 0) lse_add(acc, xs[a + 1]);
        const double v = lse_wave_finish(acc);
        if (lane == 0) xs[a] = v;
        return;
    }
    if (wave == ncell) {
        // q[1,jj] = q[1,jj-1] + sum_k q[1,k] * qb(k+1,jj) * dangles            (exterior stems carry no other term)
        const int jj = d + 1;
        const double* __restrict__ fcat = tab + VM_FCAT * ts + (size_t)(jj - 1) * ld;
        Lse acc = lse_empty();
        lse_stream2<4>(acc, f5i, fcat + 1, 0, jj - 1, lane);
        if (lane == 0) lse_add(acc, f5i[jj - 1]);
        const double v = lse_wave_finish(acc);
        if (lane == 0) f5i[jj] = v;
        return;
    }

    const int i = wave + 1, j = i + d;
    const int s_im1 = s[i - 1], s_i = s[i], s_ip1 = s[i + 1], s_j = s[j], s_jp1 = s[j + 1], s_jp2 = s[j + 2];
    const int type = (B.allow && !B.allow[((size_t)sq * ld + i) * ld + (j + 1)]) ? 0 : V->ptype[s_i * 5 + s_jp1];   // 0: the pair is excluded by a structure constraint
    const bool pairable = type != 0;
    const int rt = V->rtype[type];
    const bool inner = d >= 2;
    const double tau_here = type > 2 ? V->tau : 0.0;
    const bool nick_in = nicked(cut, i, j);   // the missing gap lies inside the pair (i, j+1)

    double op_fca = kNeg, op_fm1 = kNeg, op_fms = kNeg;
    if (inner) {
        // a multiloop element may not touch the missing gap: a branch checks the gaps on both of its sides, an
        // unpaired letter the two gaps next to it
        if (gap_ok(cut, i) && gap_ok(cut, j))   // the branch (i+1, j) as a multiloop stem: its neighbours s[i], s[j+1] both exist
            op_fca = tab[VM_FC * ts + (size_t)(i + 1) * ld + (j - 1)] + V->stemM[V->ptype[s_ip1 * 5 + s_j] * 25 + s_i * 5 + s_jp1];
        if (gap_ok(cut, i) && gap_ok(cut, i + 1)) op_fm1 = tab[VM_FM1 * ts + (size_t)(i + 1) * ld + j];
        if (gap_ok(cut, j - 1) && gap_ok(cut, j)) op_fms = tab[VM_FMST * ts + (size_t)(j - 1) * ld + i];
    }

    // ---- loops with one enclosed pair (p,q) = (i+1+l1, j-l2): qb(i,j+1) += qb(p,q) * expLoopEnergy
    Lse acc_c = lse_empty();
    if (pairable && inner) {
        const int tmax = d - 2 < kMaxSingle ? d - 2 : kMaxSingle;
        const double mm_out = V->mmI[type * 25 + s_ip1 * 5 + s_j];
        double x[kMcShapeIters];
#pragma unroll
        for (int u = 0; u < kMcShapeIters; u++) {
            const Shape sh = V->shape[64 * u + lane];
            const int kind = V->kind[64 * u + lane];
            x[u] = kEmptyMax;
            // both sides of the loop (letters i..p and q..j+1) must lie on one strand each
            if (sh.l1 + sh.l2 <= tmax && !nicked(cut, i, i + sh.l1) && !nicked(cut, j - sh.l2, j)) {
                const int p = i + 1 + sh.l1, q = j - sh.l2;
                const size_t at = (size_t)p * ld + (q - 1);
                if (kind == 1) x[u] = tab[VM_FCX * ts + at] + sh.score + mm_out;
                else {
                    const int t2 = V->ptype[s[p] * 5 + s[q]];
                    if (kind == 2) x[u] = tab[VM_FC * ts + at] + sh.score + tau_here + (t2 > 2 ? V->tau : 0.0);
                    else if (kind >= 3) {   // 2.x: 1xn / 2x3 loops have their own mismatch tables
                        const double* __restrict__ mm = kind == 3 ? V->mm1nI : V->mm23I;
                        x[u] = tab[VM_FC * ts + at] + sh.score + mm[type * 25 + s_ip1 * 5 + s_j] + mm[V->rtype[t2] * 25 + s[q + 1] * 5 + s[p - 1]];
                    } else x[u] = tab[VM_FC * ts + at] + vienna_small_loop(V, sh.l1, sh.l2, type, V->rtype[t2], s_ip1, s_j, s[p - 1], s[q + 1]);
                }
            }
        }
        lse_add_group<kMcShapeIters>(acc_c, x);
    }
    if (pairable && lane == 0 && d >= kMinHairpin) {
        if (!nick_in) lse_add(acc_c, hairpin_w(V, s, i, d, type));
        else   // the loop that holds the missing gap: two exterior-loop halves and the pair's own dangles
            lse_add(acc_c, B.xs[(size_t)sq * ld + i + 1] + B.xp[(size_t)sq * ld + j] +
                               V->stemE[rt * 25 + (gap_ok(cut, j) ? s_j : 0) * 5 + (gap_ok(cut, i) ? s_ip1 : 0)]);
    }

    // ---- FM2[i,j] = (+)_{i<k<j} FM1[i,k] + FM[k,j]
    Lse acc_2 = lse_empty();
    if (inner)
        lse_stream2<8>(acc_2, tab + VM_FM1 * ts + (size_t)i * ld, tab + VM_FMT * ts + (size_t)j * ld, i + 1, j, lane);

    double fm2, fc;
    lse_wave_finish2(acc_2, acc_c, fm2, fc);
    // multiloop closed by (i,j+1): expMLclosing * expMLintern[tt] * expdangle3[tt][S[i+1]] * expdangle5[tt][S[j]]
    fc = pairable ? lse2(fc, fm2 + V->ml_close + V->stemM[rt * 25 + s_j * 5 + s_ip1]) : kNeg;

    double fm1 = kNeg, fms = kNeg, fm = kNeg;
    if (inner) {
        fm1 = lse2(op_fca + V->mli, op_fm1 + V->mlb);
        fms = lse2(fm1, op_fms + V->mlb);
        fm = lse2(fm2, fms);
    }

    if (lane == 0) {
        const size_t ij = (size_t)i * ld + j, ji = (size_t)j * ld + i;
        // as the enclosed pair of a generic loop: mismatchI[rtype][S[q+1]][S[p-1]]; as a stem: dangles on both sides
        const double dec_x = V->mmI[rt * 25 + s_jp2 * 5 + s_im1];
        const double dec_a = V->stemE[type * 25 + (gap_ok(cut, i - 1) ? s_im1 : 0) * 5 + (gap_ok(cut, j + 1) ? s_jp2 : 0)];
        tab[VM_FC * ts + ij] = fc;
        tab[VM_FCX * ts + ij] = pairable ? fc + dec_x : kNeg;
        const double fca = pairable ? fc + dec_a : kNeg;
        tab[VM_FCA * ts + ij] = fca;
        tab[VM_FCAT * ts + ji] = fca;
        tab[VM_FM1 * ts + ij] = fm1;
        tab[VM_FM1T * ts + ji] = fm1;
        tab[VM_FM * ts + ij] = fm;
        tab[VM_FMT * ts + ji] = fm;
        tab[VM_FMST * ts + ji] = fms;
    }
}

// ---------------------------------------------------------------------------------
// outside (pull form) + posterior, diagonal d; the wave after the last cell computes F5o[d+1]
__global__ __launch_bounds__(256) void mcv_outside_diag(McBatch B, const ViennaDx* __restrict__ V, int d, int pin)
{
    int sq, slot;
    block_map_v(pin, &sq, &slot);
    if (sq >= B.ns) return;
    const int n = B.n[sq];
    const int wave = slot * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int ncell = n - 1 - d;
    const int cut = B.cut ? B.cut[sq] : 0;
    if (ncell < 1 || wave > ncell + (cut > 0 ? 2 : 0)) return;

    const int ld = B.ld;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    double* __restrict__ f5o = B.f5o + (size_t)sq * ld;
    const size_t ts = B.tab_stride;

    if (wave == ncell + 1) {
        // XPo[b], b = cut+d+1 <= n-1: pairs (i, b+1) around the missing gap (spans >= d+1, final) (+) XPo[b+1] (+) stems (b+1, b')
        const int b = cut + d + 1;
        if (b > n - 1) return;
        double* __restrict__ xpo = B.xpo + (size_t)sq * ld;
        const double* __restrict__ xs = B.xs + (size_t)sq * ld;
        const double* __restrict__ fcot = tab + VM_FCOT * ts + (size_t)b * ld;   // FCo[i][b] at [i]
        Lse acc = lse_empty();
        for (int i = 1 + lane; i <= cut; i += 64) {
            const int t = V->ptype[s[i] * 5 + s[b + 1]];
            if (!t || b + 1 - i < 4) continue;
            const int rt = V->rtype[t];
            lse_add(acc, fcot[i] + xs[i + 1] + V->stemE[rt * 25 + (gap_ok(cut, b) ? s[b] : 0) * 5 + (gap_ok(cut, i) ? s[i + 1] : 0)]);
        }
        lse_stream2<4>(acc, xpo + 1, tab + VM_FCA * ts + (size_t)(b + 1) * ld, b + 1, n, lane);
        if (lane == 0) lse_add(acc, xpo[b + 1]);
        const double v = lse_wave_finish(acc);
        if (lane == 0) xpo[b] = v;
        return;
    }
    if (wave == ncell + 2) {
        // XSo[a], a = cut-d >= 2: pairs (a-1, j+1) around the missing gap (+) XSo[a-1] (+) stems (a', a-1) before it
        const int a = cut - d;
        if (a < 2) return;
        double* __restrict__ xso = B.xso + (size_t)sq * ld;
        const double* __restrict__ xp = B.xp + (size_t)sq * ld;
        const double* __restrict__ fco = tab + VM_FCO * ts + (size_t)(a - 1) * ld;   // FCo[a-1][j]
        Lse acc = lse_empty();
        for (int j = cut + lane; j <= n - 1; j += 64) {
            const int t = V->ptype[s[a - 1] * 5 + s[j + 1]];
            if (!t || j + 1 - (a - 1) < 4) continue;
            const int rt = V->rtype[t];
            lse_add(acc, fco[j] + xp[j] + V->stemE[rt * 25 + (gap_ok(cut, j) ? s[j] : 0) * 5 + (gap_ok(cut, a - 1) ? s[a] : 0)]);
        }
        if (a >= 3) lse_stream2<4>(acc, xso, tab + VM_FCAT * ts + (size_t)(a - 2) * ld, 1, a - 1, lane);   // a' = 1..a-2
        if (lane == 0) lse_add(acc, xso[a - 1]);
        const double v = lse_wave_finish(acc);
        if (lane == 0) xso[a] = v;
        return;
    }
    if (wave == ncell) {
        const int k = d + 1;
        const double* __restrict__ fca = tab + VM_FCA * ts + (size_t)(k + 1) * ld;
        Lse acc = lse_empty();
        lse_stream2<4>(acc, f5o + 1, fca, k + 1, n, lane);
        if (lane == 0) lse_add(acc, f5o[k + 1]);
        const double v = lse_wave_finish(acc);
        if (lane == 0) f5o[k] = v;
        return;
    }

    const int i = wave + 1, j = i + d;
    const int s_im1 = s[i - 1], s_i = s[i], s_ip1 = s[i + 1], s_j = s[j], s_jp1 = s[j + 1], s_jp2 = s[j + 2];
    const bool guard_m = d >= 2;
    const int type = (B.allow && !B.allow[((size_t)sq * ld + i) * ld + (j + 1)]) ? 0 : V->ptype[s_i * 5 + s_jp1];   // 0: the pair is excluded by a structure constraint
    const bool pairable = type != 0;
    const int rt = V->rtype[type];
    const double tau_here = type > 2 ? V->tau : 0.0;

    double op_fmso = kNeg, op_fm1o = kNeg, op_fm1o_up = kNeg, op_f5o = kNeg, op_f5i = kNeg, op_x = kNeg;
    if (guard_m) {
        if (j + 1 <= n - 1 && gap_ok(cut, j) && gap_ok(cut, j + 1)) op_fmso = tab[VM_FMSOT * ts + (size_t)(j + 1) * ld + i];
        if (i - 1 >= 1 && gap_ok(cut, i - 1) && gap_ok(cut, i)) op_fm1o = tab[VM_FM1O * ts + (size_t)(i - 1) * ld + j];
    }
    if (pairable) {
        op_f5o = f5o[j + 1];
        op_f5i = f5i[i - 1];
        if (i - 1 >= 1 && j + 1 <= n - 1 && gap_ok(cut, i - 1) && gap_ok(cut, j + 1))
            op_fm1o_up = tab[VM_FM1O * ts + (size_t)(i - 1) * ld + (j + 1)];
        // stem of one of the exterior-loop halves of a loop around the missing gap
        if (cut > 0 && i > cut) op_x = B.xpo[(size_t)sq * ld + j + 1] + B.xp[(size_t)sq * ld + i - 1];
        if (cut > 0 && j + 1 <= cut) op_x = B.xso[(size_t)sq * ld + i] + B.xs[(size_t)sq * ld + j + 2];
    }
    const double fc_in = tab[VM_FC * ts + (size_t)i * ld + j];
    const double Z = f5i[n];

    // ---- enclosing loops: outer pair (i-1-l1, j+2+l2) [letters], gap cell (i-1-l1, j+1+l2)
    Lse acc_c = lse_empty();
    const double dec_a = V->stemE[type * 25 + (gap_ok(cut, i - 1) ? s_im1 : 0) * 5 + (gap_ok(cut, j + 1) ? s_jp2 : 0)];
    if (pairable) {
        const double dec_in = V->mmI[rt * 25 + s_jp2 * 5 + s_im1];
        const int l1max = i - 2, l2max = n - 2 - j;
        if (l1max >= 0 && l2max >= 0) {
            double x[kMcShapeIters];
#pragma unroll
            for (int u = 0; u < kMcShapeIters; u++) {
                const Shape sh = V->shape[64 * u + lane];
                const int kind = V->kind[64 * u + lane];
                x[u] = kEmptyMax;
                if (sh.l1 <= l1max && sh.l2 <= l2max && !nicked(cut, i - 1 - sh.l1, i - 1) && !nicked(cut, j + 1, j + 1 + sh.l2)) {
                    const int io = i - 1 - sh.l1, jo = j + 1 + sh.l2;
                    const size_t at = (size_t)io * ld + jo;
                    if (kind == 1) x[u] = tab[VM_FCOX * ts + at] + sh.score + dec_in;
                    else {
                        const int to = V->ptype[s[io] * 5 + s[jo + 1]];
                        if (kind == 2) x[u] = tab[VM_FCO * ts + at] + sh.score + tau_here + (to > 2 ? V->tau : 0.0);
                        else if (kind >= 3) {
                            const double* __restrict__ mm = kind == 3 ? V->mm1nI : V->mm23I;
                            x[u] = tab[VM_FCO * ts + at] + sh.score + mm[to * 25 + s[io + 1] * 5 + s[jo]] + mm[rt * 25 + s_jp2 * 5 + s_im1];
                        } else x[u] = tab[VM_FCO * ts + at] + vienna_small_loop(V, sh.l1, sh.l2, to, rt, s[io + 1], s[jo], s_im1, s_jp2);
                    }
                }
            }
            lse_add_group<kMcShapeIters>(acc_c, x);
        }
        if (lane == 0) {
            lse_add(acc_c, op_f5o + op_f5i + dec_a);            // stem of the exterior loop
            lse_add(acc_c, op_fm1o_up + V->mli + V->stemM[type * 25 + s_im1 * 5 + s_jp2]);   // branch of a multiloop (both neighbours exist)
            lse_add(acc_c, op_x + dec_a);
        }
    }

    // ---- FMo[i,j] = (+)_{i'<i} FM2o[i',j] + FM1[i',i];  FM1o[i,j] gathers FM2o[i,j'] + FM[j,j'], j' > j
    Lse acc_m = lse_empty(), acc_1 = lse_empty();
    if (guard_m) {
        lse_stream2<4>(acc_m, tab + VM_FM2OT * ts + (size_t)j * ld, tab + VM_FM1T * ts + (size_t)i * ld, 1, i, lane);
        lse_stream2<4>(acc_1, tab + VM_FM2O * ts + (size_t)i * ld, tab + VM_FM * ts + (size_t)j * ld, j + 1, n, lane);
        if (lane == 0) lse_add(acc_1, op_fm1o + V->mlb);
    }
    double fmo, fm1o, fco;
    lse_wave_finish3(acc_m, acc_1, acc_c, fmo, fm1o, fco);
    double fmso = kNeg;
    if (!guard_m) { fmo = kNeg; fm1o = kNeg; }
    else {
        fmso = lse2(fmo, op_fmso + V->mlb);   // FMS[i,j+1] -> FMS[i,j] + b
        fm1o = lse2(fm1o, fmso);              // FMS[i,j] -> FM1[i,j]
    }
    if (!pairable) fco = kNeg;
    const double viafc = pairable ? fco + V->ml_close + V->stemM[rt * 25 + s_j * 5 + s_ip1] : kNeg;
    const double fm2o = lse2(fmo, viafc);

    if (lane == 0) {
        const size_t ij = (size_t)i * ld + j, ji = (size_t)j * ld + i;
        tab[VM_FCO * ts + ij] = fco;
        tab[VM_FCOT * ts + ji] = fco;
        tab[VM_FCOX * ts + ij] = pairable ? fco + V->mmI[type * 25 + s_ip1 * 5 + s_j] : kNeg;
        tab[VM_FM2O * ts + ij] = fm2o;
        tab[VM_FM2OT * ts + ji] = fm2o;
        tab[VM_FMSOT * ts + ji] = fmso;
        tab[VM_FM1O * ts + ij] = fm1o;
        const double e = fco + fc_in - Z;
        double p = e > kNeg / 2 ? exp(e) : 0.0;
        p = p > 1.0 ? 1.0 : p;
        B.bp[(size_t)sq * B.tri_stride + tri_offset_v(n, i) + (j + 1)] = (pairable && d >= kMinHairpin) ? p : 0.0;
    }
}

// =================================================================================
// accessibility pass (after both sweeps)

// z = 0: FC -> FC^T, z = 1: FCo -> FCo^T (32x32 LDS tiles); z = 2: hairpin probabilities Hp[p][q] (letters), 0 elsewhere
__global__ __launch_bounds__(256) void mcv_acc_prep(McBatch B, const ViennaDx* __restrict__ V)
{
    __shared__ double tile[32][33];
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int tiles = (ld + 31) / 32;
    const int tr = blockIdx.x / tiles, tc = blockIdx.x % tiles;
    double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    if (blockIdx.z == 1) return;   // FCo^T is written by the outside cells
    if (blockIdx.z == 0) {
        const double* __restrict__ src = tab + VM_FC * ts;
        double* __restrict__ dst = tab + VM_S_FCT * ts;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
            // only interior cells 1 <= r <= c <= n-1 were written by the sweeps
            tile[ty + 8 * k][tx] = (r >= 1 && r <= c && c <= n - 1) ? src[(size_t)r * ld + c] : kNeg;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = tc * 32 + ty + 8 * k, c = tr * 32 + tx;
            if (r < ld && c < ld) dst[(size_t)r * ld + c] = tile[tx][ty + 8 * k];
        }
        return;
    }
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    const double Z = B.f5i[(size_t)sq * ld + n];
    double* __restrict__ hp = tab + VM_S_HP * ts;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = tr * 32 + ty + 8 * k, q = tc * 32 + tx;
        if (p >= ld || q >= ld) continue;
        double v = 0.0;
        if (p >= 1 && q <= n && q - p - 1 >= kMinHairpin) {
            const int type = V->ptype[s[p] * 5 + s[q]];
            if (type) {
                const double e = tab[VM_FCO * ts + (size_t)p * ld + (q - 1)] + hairpin_w(V, s, p, q - p - 1, type) - Z;
                v = e > kNeg / 2 ? exp(e) : 0.0;
            }
        }
        hp[(size_t)p * ld + q] = v;
    }
}

// exclusive prefix sums over p, in place: C[a][q] = sum_{p<a} Hp[p][q]; one thread per column q
__global__ __launch_bounds__(256) void mcv_acc_hscan(McBatch B, int slot)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n) return;
    double* __restrict__ hp = B.tab + (size_t)sq * B.seq_stride + (size_t)slot * B.tab_stride + q;
    double run = 0.0;
    for (int a = 0; a <= n; a += 4) {
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = a + k <= n ? hp[(size_t)(a + k) * ld] : 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (a + k <= n) { hp[(size_t)(a + k) * ld] = run; run += v[k]; }
    }
}

// gap probabilities of interior loops.  z = 0: GL[p][l1] = P(loop with outer 5' letter p and l1 >= 1 unpaired letters
// p+1..p+l1 before the enclosed pair), one wavefront per (p,l1), lanes over the outer 3' letter q;
// z = 1: GR[q][l2] likewise for the 3' gap q-l2..q-1, lanes over p (reads the transposed copies).
// a loop whose outside x inside weight is below e^-70 of Z cannot reach e^-60 even with the most favourable loop energy
// (a stack is worth < e^6): skipping it changes no accessibility by more than 1e-20
constexpr double kSkipLog = -70.0;
__global__ __launch_bounds__(256) void mcv_acc_gaps(McBatch B, const ViennaDx* __restrict__ V, double* __restrict__ gaps)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int wave = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int pos = wave / 30 + 1, g = wave % 30 + 1;   // outer letter (p or q), own gap length 1..30
    if (pos > n) return;
    const uint8_t* __restrict__ s = B.seq + (size_t)sq * B.lds;
    const double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const double Z = B.f5i[(size_t)sq * ld + n];
    const bool right = blockIdx.z != 0;
    double acc = 0.0;
    if (!right) {
        const int p = pos, l1 = g, k = p + 1 + l1;   // inner 5' letter
        const double* __restrict__ fco = tab + VM_FCO * ts + (size_t)p * ld;      // FCo[p][q-1]
        const double* __restrict__ fc = tab + VM_FC * ts + (size_t)k * ld;        // FC[k][l-1]
        for (int q0 = k + 5; q0 <= n; q0 += 64) {            // inner hairpin needs l >= k+4, so q >= k+5
            const int q = q0 + lane;
            if (q > n) continue;
            const int to = V->ptype[s[p] * 5 + s[q]];
            if (!to) continue;
            const double o = fco[q - 1];
            if (!(o > kNeg / 2)) continue;
            for (int l2 = 0; l2 <= kMaxSingle - l1; l2++) {
                const int l = q - 1 - l2;
                if (l < k + 4) break;
                const int ti = V->ptype[s[k] * 5 + s[l]];
                if (!ti) continue;
                const double in = fc[l - 1];
                if (!(o + in - Z > kSkipLog)) continue;
                acc += exp(o + in - Z + loop_w(V, l1, l2, to, ti, s[p + 1], s[q - 1], s[k - 1], s[l + 1]));
            }
        }
    } else {
        const int q = pos, l2 = g, l = q - 1 - l2;   // inner 3' letter
        if (l >= 5) {
            const double* __restrict__ fcot = tab + VM_S_FCOT * ts + (size_t)(q - 1) * ld;   // FCo[p][q-1] at [p]
            const double* __restrict__ fct = tab + VM_S_FCT * ts + (size_t)(l - 1) * ld;     // FC[k][l-1] at [k]
            for (int p0 = 1; p0 <= l - 5; p0 += 64) {        // k = p+1+l1 <= l-4
                const int p = p0 + lane;
                if (p > l - 5) continue;
                const int to = V->ptype[s[p] * 5 + s[q]];
                if (!to) continue;
                const double o = fcot[p];
                if (!(o > kNeg / 2)) continue;
                for (int l1 = 0; l1 <= kMaxSingle - l2; l1++) {
                    const int k = p + 1 + l1;
                    if (k > l - 4) break;
                    const int ti = V->ptype[s[k] * 5 + s[l]];
                    if (!ti) continue;
                    const double in = fct[k];
                    if (!(o + in - Z > kSkipLog)) continue;
                    acc += exp(o + in - Z + loop_w(V, l1, l2, to, ti, s[p + 1], s[q - 1], s[k - 1], s[l + 1]));
                }
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) gaps[((size_t)(2 * sq + (right ? 1 : 0)) * ld + pos) * 32 + g] = acc;
}

// up[(a-1)*max_w + w] = P(letters a..a+w unpaired); one wavefront per letter a
__global__ __launch_bounds__(256) void mcv_acc_final(McBatch B, const ViennaDx* __restrict__ V, const double* __restrict__ gaps, int max_w)
{
    const int sq = blockIdx.y;
    const int n = B.n[sq], ld = B.ld;
    const int a = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 1;
    const int lane = threadIdx.x & 63;
    if (a > n) return;
    const double* __restrict__ tab = B.tab + (size_t)sq * B.seq_stride;
    const size_t ts = B.tab_stride;
    const double* __restrict__ f5i = B.f5i + (size_t)sq * ld;
    const double* __restrict__ f5o = B.f5o + (size_t)sq * ld;
    const double Z = f5i[n];
    const double* __restrict__ gl = gaps + (size_t)(2 * sq) * ld * 32;
    const double* __restrict__ gr = gaps + (size_t)(2 * sq + 1) * ld * 32;
    const double* __restrict__ C = tab + VM_S_HP * ts + (size_t)a * ld;   // C[a][q] = sum_{p<a} Hp[p][q]
    double* __restrict__ up = B.up + ((size_t)sq * ld + (a - 1)) * max_w;
    for (int w = 0; w < max_w; w++) {
        const int b = a + w, len = w + 1;
        if (b > n) { if (lane == 0) up[w] = 0.0; continue; }
        double acc = 0.0;
        if (lane == 0) acc += exp(f5i[a - 1] + f5o[b] - Z);                              // E
        for (int q = b + 1 + lane; q <= n; q += 64) acc += C[q];                          // H
        // I: 5' gaps p+1..p+l1 covering a..b (p <= a-1, p+l1 >= b), 3' gaps q-l2..q-1 covering it
        for (int c = lane; c < 32 * 32; c += 64) {
            const int dp = c >> 5, l = c & 31;   // distance of the outer letter from the run, gap length
            const int p = a - 1 - dp, q = b + 1 + dp;
            if (l >= 1 && l <= kMaxSingle) {
                if (p >= 1 && l >= b - p) acc += gl[(size_t)p * 32 + l];
                if (q <= n && l >= q - a) acc += gr[(size_t)q * 32 + l];
            }
        }
        const double lb = len * V->mlb - Z;
        if (a >= 2 && b <= n - 3) {                                                       // M, before a branch
            const double* __restrict__ x = tab + VM_FM1O * ts + (size_t)(a - 1) * ld;
            const double* __restrict__ y = tab + VM_FM1 * ts + (size_t)b * ld;
            for (int j = b + 2 + lane; j <= n - 1; j += 64) acc += exp(x[j] + y[j] + lb);
        }
        if (a >= 4 && b <= n - 1) {                                                       // M, after the last branch
            const double* __restrict__ x = tab + VM_FMSOT * ts + (size_t)b * ld;
            const double* __restrict__ y = tab + VM_FMST * ts + (size_t)(a - 1) * ld;
            for (int i = 1 + lane; i <= a - 3; i += 64) acc += exp(x[i] + y[i] + lb);
        }
        acc = wave_sum(acc);
        if (lane == 0) up[w] = acc > 1.0 ? 1.0 : acc;
    }
}

// two-molecule batch: hp[p][i][j] = P(letter i of s1 pairs letter j of s2) = joint pair matrix entry (i, cut+j), the copy
// of /root/reference/src/ractip.cpp:451-454 without its threshold (the host adapter applies th_hy); logz[p] = F5i[n]
// lin_s >= 0: the tables are in scaled linear space (mccaskill_vlin.hip): log Z = log F5i~[n] + lin_s*n, and a Z~ outside the
// safe double range flags the pair for the log-space recomputation
__global__ __launch_bounds__(256) void mcv_extract_hp(McBatch B, double* __restrict__ hp, size_t hp_stride, int ldd, double* __restrict__ logz,
                                                      double lin_s, int* __restrict__ bad)
{
    const int p = blockIdx.y;
    const int n = B.n[p], n1 = B.cut[p], n2 = n - n1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) {
        const double z = B.f5i[(size_t)p * B.ld + n];
        if (lin_s >= 0.0) {
            if (!(z > 1e-200 && z < 1e200)) atomicOr(&bad[p], 1);
            logz[p] = log(z) + lin_s * (double)n;
        } else logz[p] = z;
    }
    if (c >= n1 * n2) return;
    const int i = c / n2 + 1, j = c % n2 + 1;
    hp[(size_t)p * hp_stride + (size_t)i * ldd + j] = B.bp[(size_t)p * B.tri_stride + tri_offset_v(n, i) + n1 + j];
}

// logZ = F5i[n]
__global__ void mcv_finish(McBatch B, double* __restrict__ logz)
{
    const int sq = blockIdx.x * blockDim.x + threadIdx.x;
    if (sq < B.ns) logz[sq] = B.f5i[(size_t)sq * B.ld + B.n[sq]];
}

}  // namespace rh
