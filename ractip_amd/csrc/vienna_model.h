// vienna_model.h -- Vienna loop energies for the duplex and McCaskill sweeps (PARITY UNPINNED), in two semantics:
//   kViennaSem18: ViennaRNA-1.8 LoopEnergy()/dangles, what the 1.8 branch of pf_duplex.c:209-433 calls (BL* default);
//   kViennaSem20: ViennaRNA-2.x E_IntLoop / E_ExtLoop / E_MLstem / E_Hairpin (pf_duplex.c:128-206 is written against them):
//                 mismatch1nI for 1xn loops, mismatch23I for 2x3 loops, mismatchExt / mismatchM instead of dangle sums where
//                 both neighbours exist, tri/tetra/hexaloop energies that REPLACE the hairpin energy.  Log-space kernels only.
//
// Energy tables: /root/reference/src/boltzmann_param.c (BL* values, shipped as data in
// ractip_amd/data/vienna_bl_star.params); energy function: the ViennaRNA-1.8 LoopEnergy() that the 1.8 branch of
// /root/reference/src/pf_duplex.c:209-433 calls.  ViennaRNA itself is a third-party dependency that is absent and
// unversioned, and the reference holds no output for this path, so results are validated by invariants and by
// brute-force enumeration only.  All entries are log Boltzmann weights  -E*10/kT  (E in 10 cal/mol).
#pragma once
#include <hip/hip_runtime.h>
#include "score_model.h"

namespace rh {

struct ViennaDx {
    double stack[64];        // [t1*8+t2]            stack[type][type_2]
    double bulge1[64];       //                      bulge[1] + stack[type][type_2]
    double int11[64 * 25];   // [(t1*8+t2)*25 + a*5+b]
    double int21[64 * 125];  // [(t1*8+t2)*125 + (a*5+b)*5+c]
    double int22[64 * 625];  // [(t1*8+t2)*625 + ((a*5+b)*5+c)*5+d]
    double mmI[8 * 25];      // mismatchI[t][a][b]
    double dangle5[8 * 5], dangle3[8 * 5];   // clipped to <= 0 kcal as scale_parameters() does at 37 C
    double tau;              // TerminalAU
    double duplex_init;      // DuplexInit (410, ViennaRNA-1.8 constant)
    double pad_[2];
    Shape shape[kMcShapes];  // row-major (l1,l2), l1+l2 <= 30 (MAXLOOP); score = length-dependent part
    int kind[kMcShapes];     // 0 = explicit small loop, 1 = generic interior loop, 2 = bulge of length >= 2;
                             // kViennaSem20 only: 3 = 1xn loop (n >= 3, mismatch1nI), 4 = 2x3 loop (mismatch23I)
    int ptype[25];           // pair type of two nucleotide codes (A,C,G,U = 1..4, other 0): CG=1 GC=2 GU=3 UG=4 AU=5 UA=6
    int rtype[8];
    // ---- McCaskill part: pf_fold semantics of ViennaRNA-1.8 part_func.c (what src/ractip.cpp:288-304, 351-375 calls)
    double mmH[8 * 25];      // mismatchH[t][a][b]
    double hairpin[32];      // hairpin[u], u <= 30
    double hairpin30, lxc;   // u > 30: hairpin30 - lxc*log(u/30)   (lxc already scaled by 10/kT)
    double tetra[4096];      // tetraloop bonus by 6-letter code (closing pair + 4 loop letters, base 4), 0 elsewhere
    double d5x[8 * 5], d3x[8 * 5];   // smoothed stem dangles; TerminalAU folded into d3x, code 0 = no neighbour
    double ml_close;         // -(ML_closing + ML_intern)
    double mli, mlb;         // -ML_intern, -ML_base
    // ---- both semantics through the same lookups (built by build_vienna_dx):
    int semantics;           // kViennaSem18 / kViennaSem20
    int nhexa;               // entries of hexa_code / hexa
    double mm1nI[8 * 25], mm23I[8 * 25];   // kViennaSem20: mismatch1nI / mismatch23I[t][a][b] (zero under kViennaSem18: no shape of kind 3 / 4)
    // exterior-loop term of a duplex end, pf_duplex.c:146,158,185,200 (2.x E_ExtLoop) / :321-326,337-340 (1.8 dangles):
    // [t*36 + a*6 + b], a = 5' neighbour letter, b = 3' neighbour letter, 5 = there is none; TerminalAU included
    double dxE[8 * 36];
    // stem of an exterior loop / of a multiloop in pf_fold: [t*25 + a*5 + b], a = 5' neighbour, b = 3' neighbour, 0 = none (or unknown
    // letter); TerminalAU included, MLintern not.  1.8: smoothed dangle5 + dangle3 for both; 2.x: mismatchExt / mismatchM where both exist
    double stemE[8 * 25], stemM[8 * 25];
    double tri[1024];        // triloop term by 5-letter code (closing pair + 3 loop letters), added to the plain hairpin energy
    int hexa_code[40];       // hexaloops by 8-letter code, base 4
    double hexa[40];
    // special hairpins: under kViennaSem20 the tabulated energy replaces the whole hairpin energy; tetra / tri / hexa hold the
    // DIFFERENCE to the plain energy of that very loop (fixed by the letters of the code), so the kernels add them in both semantics
};
constexpr int kViennaSem18 = 1, kViennaSem20 = 2;

// pair type of two nucleotide codes (A,C,G,U = 1..4, other 0): CG=1 GC=2 GU=3 UG=4 AU=5 UA=6, and the type of the reversed
// pair, in arithmetic: the kernels' first use of a letter is its pair type, and a table lookup there is one more memory round
// trip in front of everything that depends on it (ViennaDx::ptype / rtype hold the same values)
__host__ __device__ inline int vienna_ptype(int a, int b)
{
    const int s = a + b;
    return s == 5 ? (a == 2 ? 1 : (a == 3 ? 2 : (a == 1 ? 5 : (a == 4 ? 6 : 0)))) : (s == 7 ? (a == 3 ? 3 : (a == 4 ? 4 : 0)) : 0);
}
__host__ __device__ inline int vienna_rtype(int t) { return (t == 0 || t == 7) ? t : (((t - 1) ^ 1) + 1); }

// LoopEnergy for the seven shapes with joint tables (stack, 1-bulges, int11, int21, int22): type = pair type of the
// pair that closes the loop seen from outside, type_2 = rtype of the other pair; si1/sj1 = the unpaired letters next
// to the first pair inside the loop (5' side / 3' side), sp1/sq1 = those next to the second pair
__host__ __device__ inline double vienna_small_loop(const ViennaDx* V, int l1, int l2, int t1, int t2, int si1, int sj1, int sp1, int sq1)
{
    const int tt = t1 * 8 + t2;
    if (l1 == 0 && l2 == 0) return V->stack[tt];
    if (l1 + l2 == 1) return V->bulge1[tt];
    if (l1 == 1 && l2 == 1) return V->int11[tt * 25 + si1 * 5 + sj1];
    if (l1 == 1 && l2 == 2) return V->int21[tt * 125 + (si1 * 5 + sq1) * 5 + sj1];
    if (l1 == 2 && l2 == 1) return V->int21[(t2 * 8 + t1) * 125 + (sq1 * 5 + si1) * 5 + sp1];
    return V->int22[tt * 625 + ((si1 * 5 + sp1) * 5 + sq1) * 5 + sj1];
}

// param_file: the flat BL* dump (ractip_amd/data/vienna_bl_star.params), or a ViennaRNA parameter file ("## RNAfold parameter
// file", 1.x or v2.0 layout); semantics 0 = by the file (v2.0 file -> kViennaSem20, anything else kViennaSem18).
// defaults_file (may be null): a parameter file read FIRST, i.e. the library's built-in tables that RactIP leaves in place where
// neither copy_boltzmann_parameters() nor read_parameter_file() writes (src/ractip.cpp:1563-1567); use_bl: install the bundled
// BL* tables between the two, as RactIP::run does by default.  Tables no source provides stay 0.
bool load_vienna_dx_ex(const char* defaults_file, bool use_bl, const char* bl_path, const char* param_file, int semantics, ViennaDx* out,
                       char* err, int errlen);
bool load_vienna_dx(const char* path, ViennaDx* out, char* err, int errlen);

// ---- the same model in SCALED LINEAR space (mccaskill_vlin.hip): every DP quantity of span d is stored as Q*lam^d
// (inside) / Q*lam^(n-d) (outside), lam = exp(-s), exactly as lin_model.h does for the CONTRAfold model; all entries
// are Boltzmann weights exp(-E*10/kT), loop weights already carry lam^(l1+l2+2).
// Sequence-dependent factors are tabulated over two per-position letter pairs (x, x1 | y1, y), index 25*(5x+x1)+(5y1+y):
//   pair (i,j+1) seen from INSIDE its loop : x = s[i], x1 = s[i+1], y1 = s[j+1], y = s[j]
//   pair (i,j+1) seen from OUTSIDE         : x = s[j+1], x1 = s[j+2], y1 = s[i], y = s[i-1]
struct VLinModel {
    double TXO[625];      // mismatchI[type][x1][y]: closing pair of a generic interior loop (0 where the letters do not pair)
    double TMC[625];      // multiloop closure: MLclosing + MLintern + dangle3[rt][x1] + dangle5[rt][y]
    double TMH[625];      // hairpin mismatchH[type][x1][y]
    double TXI[625];      // enclosed pair of a generic interior loop: mismatchI[rt][x1][y]
    double TSA[625];      // stem of a multi / exterior loop: dangle5[type][y] + dangle3[type][x1] (TerminalAU inside dangle3)
    double TNC[625];      // pair closing the exterior-like loop around the gap between two molecules: dangle3[rt][x1] + dangle5[rt][y]
    double E_tau[8];      // TerminalAU by pair type (1 for CG/GC)
    double E_stack[64], E_bulge1[64];
    double E_int11[64 * 25], E_int21[64 * 125], E_int22[64 * 625];
    double E_tetra[4096];
    double E_hairpin[32];
    double shape_w[496];  // generic interior loops (l1, l2 >= 1, not 1x1 1x2 2x1 2x2), sorted by t = l1+l2, then l1; 0 elsewhere
    double WB[32];        // bulge of length l >= 2
    int ptype[25], rtype[8];
    double s, lam, lam2;
    double w_mu;          // lam   * exp(-ML_base)
    double w_mp2;         // lam^2 * exp(-ML_intern)
    double hairpin30, lxc;   // log-space pieces for hairpins longer than 30
};
void build_vlin_model(const ViennaDx& V, double s, VLinModel* out);

// ---- pf_duplex in SCALED LINEAR space (duplex_vlin.hip): IN~ = IN * lam^(a+b), OUT~ = OUT * lam^((L1+1-a)+(L2+1-b)) with
// a = i, b = L2+1-j; loop weights come from a VLinModel built with the duplex scale (a loop of l1+l2 unpaired letters changes
// a+b by l1+l2+2, the power the McCaskill tables already carry); what the duplex needs on top of it:
struct VDxLin {
    double E_mmI[8 * 25];   // exp(mismatchI[t][a][b])
    double E_d5[8 * 5], E_d3[8 * 5];
    double E_init;          // exp(DuplexInit)
    double s, lam;
    double pad_;
};
void build_vdx_lin(const ViennaDx& V, double s, VDxLin* out);

}  // namespace rh
