// vienna_model.h -- BL*/ViennaRNA-1.8-semantics loop energies for the duplex sweeps (PARITY UNPINNED).
//
// Energy tables: /root/reference/src/boltzmann_param.c (BL* values, shipped as data in
// ractip_amd/data/vienna_bl_star.params); energy function: the ViennaRNA-1.8 LoopEnergy() that the 1.8 branch of
// /root/reference/src/pf_duplex.c:209-433 calls.  ViennaRNA itself is a third-party dependency that is absent and
// unversioned, and the reference holds no output for this path, so results are validated by invariants and by
// brute-force enumeration only.  All entries are log Boltzmann weights  -E*10/kT  (E in 10 cal/mol).
#pragma once
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
    int kind[kMcShapes];     // 0 = explicit small loop, 1 = generic interior loop, 2 = bulge of length >= 2
    int ptype[25];           // pair type of two nucleotide codes (A,C,G,U = 1..4, other 0): CG=1 GC=2 GU=3 UG=4 AU=5 UA=6
    int rtype[8];
};

bool load_vienna_dx(const char* path, ViennaDx* out, char* err, int errlen);

}  // namespace rh
