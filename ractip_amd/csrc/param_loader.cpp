// param_loader.cpp -- bind a CONTRAfold "name value" weight file to rh::ScoreModel.
//
// Host-side equivalent of RegisterParameters + LoadValues + InitializeCache of the
// reference engines (/root/reference/src/contrafold/InferenceEngine.ipp:419-938,
// 1386-1397, 1106-1197; DuplexEngine.ipp:83-602 registers the identical set).
// Binding is by logical NAME, so the file may list the weights in any order.
#include "score_model.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>

namespace rh {
namespace {

const char kAlpha[] = "ACGU";

struct Weights {
    std::unordered_map<std::string, double> kv;
    int missing = 0;
    std::string first_missing;
    double get(const std::string& name)
    {
        auto it = kv.find(name);
        if (it == kv.end()) {
            if (!missing++) first_missing = name;
            return 0.0;
        }
        return it->second;
    }
    // two spellings are tied to the lexicographically smaller one (ipp:446-451, 758-763, 844-849)
    double tied(const std::string& a, const std::string& b) { return get(std::min(a, b)); }
};

std::string nucs(std::initializer_list<int> codes)
{
    std::string s;
    for (int c : codes) s.push_back(kAlpha[c]);
    return s;
}

// running sums of "<prefix>_at_least_k" weights, k = first..last (ipp:1120-1159)
void prefix_sum(Weights& w, const char* prefix, int first, int last, double* out)
{
    double acc = 0.0;
    for (int k = 0; k <= last; k++) {
        if (k >= first) acc += w.get(std::string(prefix) + std::to_string(k));
        out[k] = acc;
    }
}

}  // namespace

bool load_score_model(const char* path, ScoreModel* m, char* err, int errlen)
{
    std::ifstream f(path);
    if (!f) {
        snprintf(err, errlen, "cannot open parameter file %s", path);
        return false;
    }
    Weights w;
    std::string name;
    double v;
    while (f >> name >> v) w.kv[name] = v;
    std::memset(m, 0, sizeof(*m));

    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
            m->base_pair[a * 5 + b] = w.tied("base_pair_" + nucs({a, b}), "base_pair_" + nucs({b, a}));
            m->helix_closing[a * 5 + b] = w.get("helix_closing_" + nucs({a, b}));
            m->internal_1x1[a * 5 + b] = w.tied("internal_1x1_nucleotides_" + nucs({a, b}),
                                                "internal_1x1_nucleotides_" + nucs({b, a}));
            for (int c = 0; c < 4; c++) {
                m->dangle_left[a * 25 + b * 5 + c] = w.get("dangle_left_" + nucs({a, b, c}));
                m->dangle_right[a * 25 + b * 5 + c] = w.get("dangle_right_" + nucs({a, b, c}));
                for (int d = 0; d < 4; d++) {
                    const int idx = ((a * 5 + b) * 5 + c) * 5 + d;
                    m->terminal_mismatch[idx] = w.get("terminal_mismatch_" + nucs({a, b, c, d}));
                    m->helix_stacking[idx] =
                        w.tied("helix_stacking_" + nucs({a, b, c, d}), "helix_stacking_" + nucs({d, c, b, a}));
                }
            }
        }
    for (int a = 0; a < 4; a++)  // one weight feeds both orientations (ipp:688-690)
        m->bulge_0x1[a] = m->bulge_1x0[a] = w.get("bulge_0x1_nucleotides_" + nucs({a}));

    prefix_sum(w, "hairpin_length_at_least_", 0, 30, m->hairpin_len);
    m->multi_base = w.get("multi_base");
    m->multi_unpaired = w.get("multi_unpaired");
    m->multi_paired = w.get("multi_paired");
    m->external_unpaired = w.get("external_unpaired");
    m->external_paired = w.get("external_paired");

    // cache_score_single[l1][l2] (ipp:1161-1197)
    double bulge[31], inter[31], sym[16], asym[29], expl[5][5] = {};
    prefix_sum(w, "bulge_length_at_least_", 1, 30, bulge);
    prefix_sum(w, "internal_length_at_least_", 2, 30, inter);
    prefix_sum(w, "internal_symmetric_length_at_least_", 1, 15, sym);
    prefix_sum(w, "internal_asymmetry_at_least_", 1, 28, asym);
    for (int a = 1; a <= 4; a++)
        for (int b = 1; b <= 4; b++)
            expl[a][b] = w.get("internal_explicit_" + std::to_string(std::min(a, b)) + "_" +
                               std::to_string(std::max(a, b)));
    auto single_len = [&](int l1, int l2) {
        if (l1 == 0 && l2 == 0) return 0.0;
        if (l1 == 0 || l2 == 0) return bulge[std::min(30, l1 + l2)];
        double s = inter[std::min(30, l1 + l2)] + asym[std::min(28, std::abs(l1 - l2))];
        if (l1 <= 4 && l2 <= 4) s += expl[l1][l2];
        if (l1 == l2) s += sym[std::min(15, l1)];
        return s;
    };

    // shapes sorted by total unpaired length t = l1+l2, then l1 ascending, so the
    // shapes admissible for a span (t <= tmax) are a prefix of (tmax+1)(tmax+2)/2 entries
    int k = 0;
    for (int t = 0; t <= kMaxSingle; t++)
        for (int l1 = 0; l1 <= t; l1++, k++) {
            m->mc_combo_len[k] = (uint16_t)(l1 | ((t - l1) << 8));
            m->mc_combo_score[k] = single_len(l1, t - l1);
        }
    k = 0;
    for (int t = 0; t <= 28; t++)  // DuplexEngine.ipp:1038-1042 works out to l1+l2 <= 28
        for (int l1 = 0; l1 <= t; l1++, k++) m->dx_combo_len[k] = (uint16_t)(l1 | ((t - l1) << 8));

    if (w.missing) {
        snprintf(err, errlen, "%d weights missing in %s (first: %s)", w.missing, path, w.first_missing.c_str());
        return false;
    }
    return true;
}

}  // namespace rh
