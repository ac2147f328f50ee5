// param_loader.cpp -- bind a CONTRAfold "name value" weight file to rh::ScoreModel.
//
// Host-side equivalent of RegisterParameters + LoadValues + InitializeCache of the
// reference engines (/root/reference/src/contrafold/InferenceEngine.ipp:419-938,
// 1386-1397, 1106-1197; DuplexEngine.ipp:83-602 registers the identical set).
// Binding is by logical NAME, so the file may list the weights in any order.
#include "score_model.h"
#include "lin_model.h"
#include <cmath>

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

    // shapes row-major in (l1, l2): consecutive entries are consecutive columns of one row
    int k = 0;
    for (int l1 = 0; l1 <= kMaxSingle; l1++)
        for (int l2 = 0; l1 + l2 <= kMaxSingle; l2++, k++) m->mc_shape[k] = Shape{single_len(l1, l2), l1, l2};
    for (; k < kMcShapes; k++) m->mc_shape[k] = Shape{0.0, 1000, 1000};
    k = 0;
    for (int l1 = 0; l1 <= 28; l1++)  // DuplexEngine.ipp:1038-1042 works out to l1+l2 <= 28
        for (int l2 = 0; l1 + l2 <= 28; l2++, k++) m->dx_shape[k] = Shape{0.0, l1, l2};
    for (; k < kDxShapes; k++) m->dx_shape[k] = Shape{0.0, 1000, 1000};
    for (int u = 0; u < kMcShapeIters; u++) m->mc_iter_l1[u] = m->mc_shape[64 * u].l1;
    for (int u = 0; u < kDxShapeIters; u++) m->dx_iter_l1[u] = m->dx_shape[64 * u].l1;

    if (w.missing) {
        snprintf(err, errlen, "%d weights missing in %s (first: %s)", w.missing, path, w.first_missing.c_str());
        return false;
    }
    return true;
}

void build_lin_model(const ScoreModel& m, double s, LinModel* L)
{
    std::memset(L, 0, sizeof(*L));
    const double lam = std::exp(-s);
    L->s = s; L->lam = lam; L->lam2 = lam * lam;
    L->w_mu = lam * std::exp(m.multi_unpaired);
    L->w_mp2 = lam * lam * std::exp(m.multi_paired);
    L->w_eu = lam * std::exp(m.external_unpaired);
    L->w_ep2 = lam * lam * std::exp(m.external_paired);
    L->e_mpmb = std::exp(m.multi_paired + m.multi_base);
    for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) {
            L->E_bp[a * 5 + b] = std::exp(m.base_pair[a * 5 + b]);
            L->E_11[a * 5 + b] = std::exp(m.internal_1x1[a * 5 + b]);
        }
    for (int a = 0; a < 5; a++) { L->E_b01[a] = std::exp(m.bulge_0x1[a]); L->E_b10[a] = std::exp(m.bulge_1x0[a]); }
    for (int k = 0; k <= 30; k++) L->E_hairpin[k] = std::exp(m.hairpin_len[k]);
    L->E_hairpin[31] = L->E_hairpin[30];
    // letters: x = s[a], x1 = s[a+1], y1 = s[b+1], y = s[b];  index 25*(5x+x1) + (5y1+y)
    for (int x = 0; x < 5; x++) for (int x1 = 0; x1 < 5; x1++) for (int y1 = 0; y1 < 5; y1++) for (int y = 0; y < 5; y++) {
        const int idx = 25 * (5 * x + x1) + (5 * y1 + y);
        const double hc = m.helix_closing[x * 5 + y1];
        L->TJB[idx] = std::exp(hc + m.terminal_mismatch[((x * 5 + y1) * 5 + x1) * 5 + y]);
        L->TJA[idx] = std::exp(hc + m.dangle_left[x * 25 + y1 * 5 + x1] + m.dangle_right[x * 25 + y1 * 5 + y]);
        L->TST[idx] = std::exp(m.base_pair[x1 * 5 + y] + m.helix_stacking[((x * 5 + y1) * 5 + x1) * 5 + y]);
    }
    // shapes sorted by total length: recover cache_score_single from the row-major table
    auto cs = [&](int l1, int l2) {
        for (int k = 0; k < kMcShapes; k++)
            if (m.mc_shape[k].l1 == l1 && m.mc_shape[k].l2 == l2) return m.mc_shape[k].score;
        return 0.0;
    };
    int k = 0;
    for (int t = 0; t <= kMaxSingle; t++) {
        for (int l1 = 0; l1 <= t; l1++, k++) {
            const int l2 = t - l1;
            const double w = std::exp(cs(l1, l2)) * std::pow(lam, t + 2);
            L->shape_l1[k] = l1; L->shape_l2[k] = l2;
            const bool special = (l1 <= 1 && l2 <= 1);
            L->shape_w[k] = special ? 0.0 : w;
            if (l1 == 0 && l2 == 1) L->w01 = w;
            if (l1 == 1 && l2 == 0) L->w10 = w;
            if (l1 == 1 && l2 == 1) L->w11 = w;
        }
        L->shape_cnt[t] = k;
    }
    L->shape_cnt[31] = k;
}

void build_dx_lin_model(const ScoreModel& m, double s, DxLinModel* D)
{
    std::memset(D, 0, sizeof(*D));
    D->s = s;
    const double lam = std::exp(-s);
    for (int k = 0; k < 32; k++) D->lam_pow[k] = std::pow(lam, k);
    D->lam_eu = lam * std::exp(m.external_unpaired);
    for (int k = 0; k < 625; k++) { D->E_tm[k] = std::exp(m.terminal_mismatch[k]); D->E_hs[k] = std::exp(m.helix_stacking[k]); }
    for (int k = 0; k < 25; k++) {
        D->E_bp[k] = std::exp(m.base_pair[k]); D->E_hc[k] = std::exp(m.helix_closing[k]); D->E_11[k] = std::exp(m.internal_1x1[k]);
    }
    for (int k = 0; k < 125; k++) { D->E_dl[k] = std::exp(m.dangle_left[k]); D->E_dr[k] = std::exp(m.dangle_right[k]); }
    for (int k = 0; k < 5; k++) { D->E_b01[k] = std::exp(m.bulge_0x1[k]); D->E_b10[k] = std::exp(m.bulge_1x0[k]); }
}

}  // namespace rh
