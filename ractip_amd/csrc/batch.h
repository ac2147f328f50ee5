// batch.h -- HBM layout of one batch of independent DP problems.
//
// McCaskill side: NS sequences (2 per pair).  Every DP table is a square
// ld x ld array of doubles per sequence (ld = nmax+2, gap indices 0..n on both
// axes, element [r*ld+c]); tables that are read along a column by some
// recurrence are ALSO stored transposed so that every inner loop of every kernel
// streams two contiguous rows.  Only interior cells 1 <= i <= j <= n-1 are ever
// read, and each is fully (re)written by the sweep, so tables need no clearing
// between batches.
//
// Duplex side: NP pairs, tables (n1max+2) x ldd, 1-based letters.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rh {

// ---- McCaskill tables (gap-indexed cell (i,j): letters i+1..j inside; FC assumes
//      letters (i, j+1) paired -- InferenceEngine.ipp:3556-3567)
enum McTable {
    T_FC = 0,   // FCi[i][j]
    T_FCX,      // FCi + ScoreBasePair(i,j+1) + ScoreJunctionB(j+1,i-1): operand of single-branch loops
    T_FCA,      // FCi + ScoreBasePair(i,j+1) + ScoreJunctionA(j+1,i-1): operand of FM1 / F5
    T_FCAT,     // transpose of T_FCA  ([j][i])
    T_FM1,      // FM1i[i][j]
    T_FM1T,     // [j][i]
    T_FM,       // FMi[i][j]
    T_FMT,      // [j][i]
    T_FCO,      // FCo[i][j]
    T_FCOX,     // FCo + ScoreJunctionB(i,j): operand of the outside single-branch gather
    T_FM2O,     // FM2o[i][j]
    T_FM2OT,    // [j][i]
    T_FMO,      // FMo[i][j]
    T_FM1O,     // FM1o[i][j]
    T_COUNT
};

constexpr int kViennaMcTables = 16;   // VmTable of mccaskill_vienna.hip (own enum, same square layout)

struct McBatch {
    const uint8_t* seq;  // [NS][lds] nucleotide codes, seq[0] = seq[n+1] = 4
    const int* n;        // [NS]
    double* tab;         // [NS][T_COUNT][ld*ld]
    double* f5i;         // [NS][ld]  F5i[0..n]
    double* f5o;         // [NS][ld]  F5o[0..n]
    double* bp;          // [NS][tri_stride] posterior, reference triangular layout
    double* up;          // [NS][ld*max_w]  up[i*max_w+w] = P(letters i+1..i+1+w unpaired); max_w = 1 for the CONTRAfold model
    // two-molecule form (co_pf_fold semantics, mccaskill_vienna.hip only): cut[sq] = n1 > 0 means the sequence is s1+s2
    // and the backbone gap after letter n1 does not exist; xp/xs = exterior partition functions of s2's prefixes
    // cut+1..b and s1's suffixes a..cut, xpo/xso their outside counterparts.  cut == nullptr: one molecule each.
    // structure constraints (fold_constrained, Vienna-BL kernels only): allow[sq*ld*ld + a*ld + b] != 0 iff letters a < b may
    // pair; nullptr = unconstrained
    const uint8_t* allow;
    const int* cut;      // [NS] or nullptr
    double* xp;          // [NS][ld]
    double* xs;
    double* xpo;
    double* xso;
    int ns, nmax, ld, lds;
    size_t tab_stride;   // doubles per table  (ld*ld)
    size_t seq_stride;   // doubles per sequence (T_COUNT*ld*ld)
    size_t tri_stride;   // doubles per bp table
    // operand tiles of the block products (mccaskill_far.hip): kPkCopies re-laid copies of the 16x16 tiles (P <= Q) of
    // FM1 / FM / FM2o, each tile 256 doubles in MFMA fragment order; [NS][kPkCopies][pk_stride]
    double* rowp;        // [NS][2][ld] look-ahead partial sums of the next inside diagonal (mccaskill_lin.hip, MODE 1 -> 2)
    double* pk;
    size_t pk_stride;    // doubles per copy = nb*(nb+1)/2 * 256
    int nb;              // 16-blocks per axis = (nmax-1)/16 + 1
    // two-molecule batch, scaled linear kernels: the one-strand cells of the inside tables were copied from the single-molecule
    // folds of the same upload (vlin_co_seed), so the inside sweep computes only the groups that touch both strands
    int seeded;
};
constexpr int kPkCopies = 6;
// sequences of kSmallMin .. kSmallMax letters are folded by mccaskill_small.hip (one workgroup each, tables in LDS); the upper bound is
// what three triangles of doubles plus the partial-sum buffers leave of 160 KB
constexpr int kSmallMin = 8, kSmallMax = 109;
// the strip kernels (mccaskill_strip.hip) run when the longest sequence the sweeps see has at least this many letters (one strip behind the
// 32 bootstrap diagonals); shorter sequences of such a batch get a pass of their own (rh_api.hip: launch_mc_lin)
constexpr int kStripMinN = 40;
// layout of rh_ctx::d_wT: transposed weights of the strip kernels [31][40], their factored tables, zero-padded rows [31][32] for mccaskill_small.hip
constexpr int kStripFiltOff = 31 * 40, kStripFiltLen = 232, kSmallWLen = 31 * 32;

enum DxTable {
    D_IN = 0,  // inside[i][j]
    D_INX,     // inside + terminal_mismatch[s1[i]][s2[j]][s1[i+1]][s2[j-1]]: as upstream pair of a loop
    D_OUT,     // outside[i][j]
    D_OUTX,    // outside + terminal_mismatch[s2[j]][s1[i]][s2[j+1]][s1[i-1]] + base_pair[s1[i]][s2[j]]
    D_COUNT
};

struct DxBatch {
    const uint8_t* seq;  // the McCaskill sequence buffer: pair p = sequences 2p (s1) and 2p+1 (s2)
    const int* n;        // [2*NP]
    double* tab;         // [NP][D_COUNT][rows*ldd]
    double* hp;          // [NP][rows*ldd]  posterior, 1-based, row/col 0 zero
    double* logz;        // [NP]
    int np, n1max, n2max, ldd, lds;
    size_t tab_stride;   // rows*ldd
    size_t pair_stride;  // D_COUNT*rows*ldd
};

// ---- duplex, scaled linear path: anti-diagonal-major tables [sd*lda + kDxPad + a], a = i, sd = i + (L2+1-j)
constexpr int kDxPad = 32;   // zero columns on both sides of every row (>= 29: the longest window reach)
enum DxLinTable { DL_IN = 0, DL_INX, DL_OUT, DL_OUTX, DL_COUNT };

struct DxLinBatch {
    const uint8_t* seq;
    const int* n;        // [2*NP]
    double* tab;         // [NP][DL_COUNT][rows*lda]
    double* hp;          // same posterior buffer / layout as DxBatch::hp
    int np, n1max, n2max, lda, lds, ldd;
    size_t tab_stride;   // rows*lda (+ slack)
    size_t pair_stride;  // DL_COUNT*tab_stride
    size_t hp_stride;
    double pw_in[2];     // (lam*e^eu)^(sd-2) * lam^2 for the two inside diagonals of this launch
    double pw_out[2];    // (lam*e^eu)^(L1+L2-sd) * lam^2 for the two outside diagonals
    double pw4[4];       // dxl_sweep4: (lam*e^eu)^(4*step+k) * lam^2, k = 0..3 (the same for both directions)
    double pw8[8];       // dxl_strip8: (lam*e^eu)^(8*step+k) * lam^2, k = 0..7
};

}  // namespace rh
