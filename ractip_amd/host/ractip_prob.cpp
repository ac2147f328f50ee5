// ractip_prob.cpp -- see ractip_prob.hpp.  Pure host code above the C ABI.
#include "ractip_prob.hpp"

#include <algorithm>

#include "../../include/ractip_hot.h"

namespace ractip_amd {

VI make_offsets(uint L)
{
    VI off(L + 1);
    for (uint i = 0; i <= L; ++i) off[i] = (int)(i * ((L + 1) + (L + 1) - i - 1) / 2);
    return off;
}

ProbabilityEngine::ProbabilityEngine(int device, float th_hy, const char* param_file)
    : ctx_(rh_create(device, RH_MODEL_CONTRAFOLD, param_file)), th_hy_(th_hy)
{
    if (!ctx_) throw std::logic_error(std::string("ractip_amd: ") + rh_last_error(nullptr));
}

ProbabilityEngine::~ProbabilityEngine() { rh_destroy(ctx_); }

void ProbabilityEngine::raise(const char* where) const
{
    throw std::logic_error(std::string("ractip_amd::") + where + ": " + rh_last_error(ctx_));
}

namespace {
// narrow to float exactly where the reference does (VF containers, src/ractip.cpp:82-83)
void narrow_bp(const std::vector<double>& d, VF& bp)
{
    bp.resize(d.size());
    std::transform(d.begin(), d.end(), bp.begin(), [](double v) { return (float)v; });
}
void narrow_up(const std::vector<double>& d, VVF& up)
{
    up.assign(d.size(), VF(1));
    for (size_t i = 0; i < d.size(); ++i) up[i][0] = (float)d[i];
}
}  // namespace

void ProbabilityEngine::contrafold(const std::string& seq, VF& bp, VI& offset, VVF& up) const
{
    const uint L = seq.size();
    std::vector<double> dbp((size_t)(L + 1) * (L + 2) / 2, 0.0), dup(L, 1.0);
    offset = make_offsets(L);
    if (L > 0 && rh_fold(ctx_, seq.c_str(), (int)L, dbp.data(), dup.data(), nullptr) != RH_OK) raise("contrafold");
    narrow_bp(dbp, bp);
    narrow_up(dup, up);
}

void ProbabilityEngine::contraduplex(const std::string& seq1, const std::string& seq2, VVF& hp) const
{
    const uint n1 = seq1.size(), n2 = seq2.size();
    hp.assign(n1 + 1, VF(n2 + 1, 0.0f));
    if (n1 == 0 || n2 == 0) return;
    std::vector<double> d((size_t)(n1 + 1) * (n2 + 1));
    if (rh_duplex(ctx_, seq1.c_str(), (int)n1, seq2.c_str(), (int)n2, d.data(), nullptr) != RH_OK) raise("contraduplex");
    for (uint i = 1; i <= n1; ++i)
        for (uint j = 1; j <= n2; ++j) {
            const float p = (float)d[(size_t)i * (n2 + 1) + j];
            hp[i][j] = p >= th_hy_ ? p : 0.0f;  // GetPosterior(th_hy_, ip), DuplexEngine.ipp:1189-1196
        }
}

void ProbabilityEngine::rnaduplex(const std::string& seq1, const std::string& seq2, VVF& hp) const
{
    const uint n1 = seq1.size(), n2 = seq2.size();
    hp.assign(n1 + 1, VF(n2 + 1, 0.0f));  // hp.resize(s1.size()+1, VF(s2.size()+1)), :393
    if (n1 == 0 || n2 == 0) return;
    std::vector<double> d((size_t)(n1 + 1) * (n2 + 1));
    if (rh_duplex(ctx_, seq1.c_str(), (int)n1, seq2.c_str(), (int)n2, d.data(), nullptr) != RH_OK) raise("rnaduplex");
    for (uint i = 1; i <= n1; ++i)
        for (uint j = 1; j <= n2; ++j) hp[i][j] = (float)d[(size_t)i * (n2 + 1) + j];  // :395-397, no threshold
}

std::vector<PairProbabilities> ProbabilityEngine::solve_probabilities(
    const std::vector<std::pair<std::string, std::string>>& pairs) const
{
    const int np = (int)pairs.size();
    std::vector<PairProbabilities> out(np);
    if (np == 0) return out;
    std::vector<const char*> a(np), b(np);
    std::vector<int> na(np), nb(np);
    for (int p = 0; p < np; ++p) {
        a[p] = pairs[p].first.c_str(); b[p] = pairs[p].second.c_str();
        na[p] = (int)pairs[p].first.size(); nb[p] = (int)pairs[p].second.size();
    }
    if (rh_batch_upload(ctx_, np, a.data(), na.data(), b.data(), nb.data()) != RH_OK) raise("solve_probabilities");
    if (rh_batch_compute(ctx_) != RH_OK) raise("solve_probabilities");
    for (int p = 0; p < np; ++p) {
        const uint n1 = na[p], n2 = nb[p];
        std::vector<double> bp1((size_t)(n1 + 1) * (n1 + 2) / 2), bp2((size_t)(n2 + 1) * (n2 + 2) / 2), up1(n1), up2(n2),
            hp((size_t)(n1 + 1) * (n2 + 1));
        double z[3];
        if (rh_batch_results(ctx_, p, bp1.data(), bp2.data(), up1.data(), up2.data(), hp.data(), z) != RH_OK)
            raise("solve_probabilities");
        PairProbabilities& r = out[p];
        narrow_bp(bp1, r.bp1); narrow_bp(bp2, r.bp2);
        narrow_up(up1, r.up1); narrow_up(up2, r.up2);
        r.offset1 = make_offsets(n1); r.offset2 = make_offsets(n2);
        r.hp.assign(n1 + 1, VF(n2 + 1));
        for (uint i = 0; i <= n1; ++i)
            for (uint j = 0; j <= n2; ++j) r.hp[i][j] = (float)hp[(size_t)i * (n2 + 1) + j];
        r.logZ1 = z[0]; r.logZ2 = z[1]; r.logZd = z[2];
    }
    return out;
}

}  // namespace ractip_amd
