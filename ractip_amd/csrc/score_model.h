// score_model.h -- device-visible scoring tables of the CONTRAfold log-linear model.
//
// Physical tables the reference binds its 708 logical weights to
// (/root/reference/src/contrafold/InferenceEngine.ipp:419-938) plus the two
// sequence-independent caches it derives from them (ipp:1120-1124, 1161-1197).
// Nucleotide codes: A,C,G,U -> 0..3, anything else -> 4 (ipp:379-384); every table
// is 5-wide per nucleotide dimension and slot 4 is zero (ipp:440-443), so sequence
// sentinels (code 4 at positions 0 and n+1) make the edge conditions of
// ScoreJunctionA (ipp:1927-1956) fall out without branches.
//
// The single-branch loop shapes (l1,l2) are flattened row-major (l1 major, l2 minor) so
// that consecutive lanes gather consecutive columns of one table row (coalesced), and
// padded to a multiple of 64 with never-valid shapes:
//   McCaskill: l1+l2 <= 30 (496 shapes -> 512, ipp:3601-3606)
//   duplex   : l1+l2 <= 28 (435 shapes -> 448, DuplexEngine.ipp:1038-1042)
#pragma once
#include <stdint.h>

#define RH_NEG_INF (-2e20)  // LogSpace.hpp:12 (same finite sentinel as the reference)

namespace rh {

constexpr int kMaxSingle = 30;   // Config.hpp:213
constexpr int kMinHairpin = 3;   // Config.hpp:212
constexpr int kMcShapes = 512;   // 496 real
constexpr int kDxShapes = 448;   // 435 real
constexpr int kMcShapeIters = kMcShapes / 64;
constexpr int kDxShapeIters = kDxShapes / 64;

struct alignas(16) Shape {      // 16 bytes: one dwordx4 load
    double score;   // cache_score_single[l1][l2] (McCaskill; 0 for the duplex, which has no length term)
    int l1, l2;     // padding entries carry l1 = l2 = 1000
};

struct ScoreModel {
    double base_pair[25];         // [a*5+b]
    double helix_closing[25];     // [a*5+b]
    double internal_1x1[25];      // [a*5+b]
    double bulge_0x1[8];          // [a] (padded)
    double bulge_1x0[8];
    double hairpin_len[32];       // prefix-summed, index min(len,30)
    double dangle_left[125];      // [a*25+b*5+c]
    double dangle_right[125];
    double terminal_mismatch[625];  // [((a*5+b)*5+c)*5+d]
    double helix_stacking[625];
    double multi_base, multi_unpaired, multi_paired;
    double external_unpaired, external_paired;
    double pad_[4];   // keeps the shape tables 16-byte aligned
    // flattened single-branch shapes
    Shape mc_shape[kMcShapes];
    Shape dx_shape[kDxShapes];
    int mc_iter_l1[kMcShapeIters];   // l1 of the first shape of each 64-wide iteration (early exit)
    int dx_iter_l1[kDxShapeIters];
};

static_assert(sizeof(Shape) == 16, "Shape must be one 16-byte load");

// host: parse a CONTRAfold "name value" file and bind it (param_loader.cpp)
bool load_score_model(const char* path, ScoreModel* out, char* err, int errlen);

}  // namespace rh
