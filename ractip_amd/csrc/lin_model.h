// lin_model.h -- the scoring model in SCALED LINEAR space (the fast path).
//
// Every DP quantity Q of span d = j-i is stored as Q * lambda^d (inside) or
// Q * lambda^(n-d) (outside), lambda = exp(-s): the log-semiring recurrences of
// /root/reference/src/contrafold/InferenceEngine.ipp:3356-4080 become plain
// multiply-adds (a hop that changes the span by D carries a factor lambda^D), and
// posterior = FCo~ * FCi~ / F5i~[n] needs no rescaling.  s ~ logZ/n keeps all values
// within a few e^+-20 of 1 for ordinary sequences (measured: |log Q~| < 25 at n=500);
// a batch whose partition function leaves the double range is detected
// (non-finite / non-positive Z~) and recomputed by the log-space kernels.
//
// Sequence-dependent scores are pre-multiplied into 625-entry tables indexed by two
// per-position codes u[i] = 5*s[i]+s[i+1], v[j] = 5*s[j+1]+s[j]:
//   junction_b(a,b) = TJB[25*u[a]+v[b]]  (ScoreJunctionB, ipp:2004-2029)
//   junction_a(a,b) = TJA[25*u[a]+v[b]]  (ScoreJunctionA, ipp:1927-1956)
//   stack(a,b)      = TST[25*u[a]+v[b]]  (ScoreBasePair(a+1,b)+ScoreHelixStacking(a,b+1), ipp:3595)
#pragma once
#include "score_model.h"

namespace rh {

constexpr int kLinShapes = 496;  // sorted by t = l1+l2 (then l1): shapes valid for a span are a prefix

struct LinModel {
    double TJB[625], TJA[625], TST[625];
    double E_bp[25];      // exp(base_pair[a][b])
    double E_11[25];      // exp(internal_1x1[a][b])
    double E_b01[8], E_b10[8];
    double E_hairpin[32]; // exp(hairpin_len[min(d,30)])
    // shape weights exp(cache_score_single[l1][l2]) * lambda^(l1+l2+2); the stacking shape and
    // the three shapes with nucleotide terms have weight 0 here and are added in the epilogue
    double shape_w[kLinShapes];
    int shape_l1[kLinShapes];
    int shape_l2[kLinShapes];
    int shape_cnt[32];    // number of shapes with l1+l2 <= t
    double w01, w10, w11; // exp(cs)*lambda^(t+2) of shapes (0,1), (1,0), (1,1)
    double s;             // scale exponent per unit span
    double lam, lam2;     // lambda, lambda^2
    double w_mu;          // lambda   * exp(multi_unpaired)
    double w_mp2;         // lambda^2 * exp(multi_paired)
    double w_eu;          // lambda   * exp(external_unpaired)
    double w_ep2;         // lambda^2 * exp(external_paired)
    double e_mpmb;        // exp(multi_paired + multi_base)
};

void build_lin_model(const ScoreModel& m, double s, LinModel* out);

// duplex scores in scaled linear space (duplex_lin.hip); lam = exp(-s) per unit of a+b
struct DxLinModel {
    double E_tm[625], E_hs[625];
    double E_bp[25], E_hc[25], E_11[25];
    double E_dl[125], E_dr[125];
    double E_b01[8], E_b10[8];
    double lam_pow[32];   // lam^k
    double lam_eu;        // lam * exp(external_unpaired)
    double s;
};
void build_dx_lin_model(const ScoreModel& m, double s, DxLinModel* out);

}  // namespace rh
