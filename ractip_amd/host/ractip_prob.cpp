// ractip_prob.cpp -- see ractip_prob.hpp.  Pure host code above the C ABI.
#include "ractip_prob.hpp"

#include <algorithm>
#include <thread>

#include "../../include/ractip_hot.h"

namespace ractip_amd {

VI make_offsets(uint L)
{
    VI off(L + 1);
    for (uint i = 0; i <= L; ++i) off[i] = (int)(i * ((L + 1) + (L + 1) - i - 1) / 2);
    return off;
}

ProbabilityEngine::ProbabilityEngine(int device, float th_hy, const char* param_file)
    : ProbabilityEngine(std::vector<int>(1, device), th_hy, param_file) {}

ProbabilityEngine::ProbabilityEngine(const std::vector<int>& devices, float th_hy, const char* param_file)
    : ctx_(nullptr), devices_(devices.empty() ? std::vector<int>(1, 0) : devices), device_(devices_[0]), th_hy_(th_hy)
{
    for (int d : devices_) {
        rh_ctx* c = rh_create(d, RH_MODEL_CONTRAFOLD, param_file);
        if (!c) {
            const std::string why = rh_last_error(nullptr);
            for (rh_ctx* k : ctxs_) rh_destroy(k);
            throw std::logic_error("ractip_amd: " + why);
        }
        ctxs_.push_back(c);
    }
    ctx_ = ctxs_[0];
    vctxs_.assign(devices_.size(), nullptr);
}

ProbabilityEngine::~ProbabilityEngine()
{
    for (rh_ctx* c : ctxs_) rh_destroy(c);
    for (rh_ctx* c : vctxs_) rh_destroy(c);
}

rh_ctx* ProbabilityEngine::vienna() const
{
    if (!vctx_) {
        for (size_t k = 0; k < devices_.size(); ++k) {
            if (!vctxs_[k])
                vctxs_[k] = rh_create_vienna(devices_[k], v_defaults_.empty() ? nullptr : v_defaults_.c_str(), v_use_bl_ ? 1 : 0,
                                             v_param_.empty() ? nullptr : v_param_.c_str(), v_semantics_);
            if (!vctxs_[k]) throw std::logic_error(std::string("ractip_amd: ") + rh_last_error(nullptr));
        }
        vctx_ = vctxs_[0];
    }
    return vctx_;
}

void ProbabilityEngine::set_vienna_parameters(const std::string& defaults_file, bool use_bl_param, const std::string& param_file, int semantics)
{
    v_defaults_ = defaults_file; v_use_bl_ = use_bl_param; v_param_ = param_file; v_semantics_ = semantics;
    for (rh_ctx*& c : vctxs_) { if (c) rh_destroy(c); c = nullptr; }
    vctx_ = nullptr;
}

std::pair<int, int> ProbabilityEngine::shard_bounds(int num, int k, int parts)
{
    const int base = num / parts, extra = num % parts;
    const int lo = k * base + std::min(k, extra);
    return {lo, lo + base + (k < extra ? 1 : 0)};
}

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

void ProbabilityEngine::rnafold(const std::string& seq, VF& bp, VI& offset) const
{
    const uint L = seq.size();
    std::vector<double> dbp((size_t)(L + 1) * (L + 2) / 2, 0.0);
    offset = make_offsets(L);
    rh_ctx* v = vienna();
    if (L > 0 && rh_bpp(v, seq.c_str(), (int)L, nullptr, dbp.data(), nullptr) != RH_OK)
        throw std::logic_error(std::string("ractip_amd::rnafold: ") + rh_last_error(v));
    narrow_bp(dbp, bp);
}

void ProbabilityEngine::rnafold(const std::string& seq, VF& bp, VI& offset, VVF& up, uint max_w) const
{
    const uint L = seq.size();
    std::vector<double> dbp((size_t)(L + 1) * (L + 2) / 2, 0.0), dup((size_t)L * max_w, 0.0);
    offset = make_offsets(L);
    rh_ctx* v = vienna();
    if (L > 0 && (rh_set_max_w(v, (int)max_w) != RH_OK || rh_fold(v, seq.c_str(), (int)L, dbp.data(), dup.data(), nullptr) != RH_OK))
        throw std::logic_error(std::string("ractip_amd::rnafold: ") + rh_last_error(v));
    narrow_bp(dbp, bp);
    up.assign(L, VF(max_w));   // up.resize(L, VF(max_w)), :370
    for (uint i = 0; i < L; ++i)
        for (uint w = 0; w < max_w; ++w) up[i][w] = (float)dup[(size_t)i * max_w + w];
}

void ProbabilityEngine::rnafold(const std::string& seq, const std::string& str, VF& bp, VI& offset, VVF& up, uint max_w) const
{
    const uint L = seq.size();
    std::string c(L, '.');   // src/ractip.cpp:275-287
    for (uint i = 0; i != str.size() && i != L; ++i) c[i] = (str[i] == '[' || str[i] == ']' || str[i] == 'e') ? 'x' : str[i];
    std::vector<double> dbp((size_t)(L + 1) * (L + 2) / 2, 0.0), dup((size_t)L * max_w, 0.0);
    offset = make_offsets(L);
    rh_ctx* v = vienna();
    if (L > 0 && (rh_set_max_w(v, (int)max_w) != RH_OK ||
                  rh_fold_constrained(v, seq.c_str(), (int)L, c.c_str(), dbp.data(), dup.data(), nullptr) != RH_OK))
        throw std::logic_error(std::string("ractip_amd::rnafold: ") + rh_last_error(v));
    narrow_bp(dbp, bp);
    up.assign(L, VF(max_w));
    for (uint i = 0; i < L; ++i)
        for (uint w = 0; w < max_w; ++w) up[i][w] = (float)dup[(size_t)i * max_w + w];
}

void ProbabilityEngine::rnaduplex_cofold(const std::string& seq1, const std::string& seq2, VVF& hp) const
{
    const uint n1 = seq1.size(), n2 = seq2.size();
    hp.assign(n1 + 1, VF(n2 + 1, 0.0f));   // hp.resize(s1.size()+1, VF(s2.size()+1, 0.0)), :403-404
    if (n1 == 0 || n2 == 0) return;
    rh_ctx* v = vienna();
    std::vector<double> d((size_t)(n1 + 1) * (n2 + 1));
    bool ok = rh_set_hybrid(v, RH_HYBRID_COFOLD) == RH_OK &&
              rh_duplex(v, seq1.c_str(), (int)n1, seq2.c_str(), (int)n2, d.data(), nullptr) == RH_OK;
    const std::string why = ok ? "" : rh_last_error(v);
    rh_set_hybrid(v, RH_HYBRID_DUPLEX);
    if (!ok) throw std::logic_error("ractip_amd::rnaduplex: " + why);
    for (uint i = 1; i <= n1; ++i)
        for (uint j = 1; j <= n2; ++j) {
            const float p = (float)d[(size_t)i * (n2 + 1) + j];   // pair_info::p is a float
            if (p > th_hy_) hp[i][j] = p;                          // :451-454
        }
}

void ProbabilityEngine::rnaduplex_cofold(const std::string& seq1, const std::string& str1, const std::string& seq2,
                                         const std::string& str2, VVF& hp) const
{
    const uint n1 = seq1.size(), n2 = seq2.size();
    hp.assign(n1 + 1, VF(n2 + 1, 0.0f));
    if (n1 == 0 || n2 == 0) return;
    std::string c(n1 + n2, '.');   // src/ractip.cpp:409-440
    for (uint i = 0; i != str1.size() && i != n1; ++i)
        switch (str1[i]) {
            case '[': c[i] = '('; break;
            case '(': case ')': case 'l': case 'x': c[i] = 'x'; break;
            default: break;
        }
    for (uint i = 0; i != str2.size() && i != n2; ++i)
        switch (str2[i]) {
            case ']': c[n1 + i] = ')'; break;
            case '(': case ')': case 'l': case 'x': c[n1 + i] = 'x'; break;
            default: break;
        }
    rh_ctx* v = vienna();
    std::vector<double> d((size_t)(n1 + 1) * (n2 + 1));
    if (rh_cofold_constrained(v, seq1.c_str(), (int)n1, seq2.c_str(), (int)n2, c.c_str(), d.data(), nullptr) != RH_OK)
        throw std::logic_error(std::string("ractip_amd::rnaduplex: ") + rh_last_error(v));
    for (uint i = 1; i <= n1; ++i)
        for (uint j = 1; j <= n2; ++j) {
            const float p = (float)d[(size_t)i * (n2 + 1) + j];
            if (p > th_hy_) hp[i][j] = p;
        }
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
    // the --duplex branch (src/ractip.cpp:390-398): pf_duplex() with the BL* energies, i.e. the Vienna-BL context with hp from
    // the duplex-only ensemble -- the same numbers the pf_duplex shim and solve_probabilities_default(duplex = true) return
    const uint n1 = seq1.size(), n2 = seq2.size();
    hp.assign(n1 + 1, VF(n2 + 1, 0.0f));  // hp.resize(s1.size()+1, VF(s2.size()+1)), :393
    if (n1 == 0 || n2 == 0) return;
    rh_ctx* v = vienna();
    std::vector<double> d((size_t)(n1 + 1) * (n2 + 1));
    if (rh_set_hybrid(v, RH_HYBRID_DUPLEX) != RH_OK || rh_duplex(v, seq1.c_str(), (int)n1, seq2.c_str(), (int)n2, d.data(), nullptr) != RH_OK)
        throw std::logic_error(std::string("ractip_amd::rnaduplex: ") + rh_last_error(v));
    for (uint i = 1; i <= n1; ++i)
        for (uint j = 1; j <= n2; ++j) hp[i][j] = (float)d[(size_t)i * (n2 + 1) + j];  // :395-397, no threshold
}

std::vector<PairProbabilities> ProbabilityEngine::solve_probabilities(
    const std::vector<std::pair<std::string, std::string>>& pairs) const
{
    return batch(ctx_, pairs, 1, false);
}

std::vector<PairProbabilities> ProbabilityEngine::solve_probabilities_default(
    const std::vector<std::pair<std::string, std::string>>& pairs, uint max_w, bool duplex) const
{
    vienna();
    for (rh_ctx* v : vctxs_)
        if (rh_set_max_w(v, (int)std::max(1u, max_w)) != RH_OK || rh_set_hybrid(v, duplex ? RH_HYBRID_DUPLEX : RH_HYBRID_COFOLD) != RH_OK)
            throw std::logic_error(std::string("ractip_amd::solve_probabilities_default: ") + rh_last_error(v));
    std::vector<PairProbabilities> out = batch(vctx_, pairs, std::max(1u, max_w), !duplex);
    for (rh_ctx* v : vctxs_) rh_set_hybrid(v, RH_HYBRID_DUPLEX);
    return out;
}

// contiguous blocks of the pairs, one per listed device, each on its own context and host thread; results in iteration order
std::vector<PairProbabilities> ProbabilityEngine::batch(rh_ctx* ctx, const std::vector<std::pair<std::string, std::string>>& pairs,
                                                        uint max_w, bool threshold_hp) const
{
    const int np = (int)pairs.size();
    const std::vector<rh_ctx*>& all = (ctx == ctx_) ? ctxs_ : vctxs_;
    const int parts = std::min<int>((int)all.size(), std::max(1, np));
    if (parts <= 1) return batch_one(ctx, pairs, 0, np, max_w, threshold_hp);
    std::vector<std::vector<PairProbabilities>> part(parts);
    std::vector<std::string> errors(parts);
    std::vector<std::thread> workers;
    for (int k = 0; k < parts; ++k)
        workers.emplace_back([&, k] {
            try {
                const auto b = shard_bounds(np, k, parts);
                part[k] = batch_one(all[k], pairs, b.first, b.second, max_w, threshold_hp);
            } catch (const std::exception& e) { errors[k] = e.what(); }
        });
    for (std::thread& t : workers) t.join();
    std::vector<PairProbabilities> out;
    out.reserve(np);
    for (int k = 0; k < parts; ++k) {
        if (!errors[k].empty()) throw std::logic_error(errors[k]);
        for (PairProbabilities& r : part[k]) out.push_back(std::move(r));
    }
    return out;
}

std::vector<PairProbabilities> ProbabilityEngine::batch_one(rh_ctx* ctx, const std::vector<std::pair<std::string, std::string>>& all_pairs,
                                                            int lo, int hi, uint max_w, bool threshold_hp) const
{
    const std::vector<std::pair<std::string, std::string>> pairs(all_pairs.begin() + lo, all_pairs.begin() + hi);
    const int np = (int)pairs.size();
    std::vector<PairProbabilities> out(np);
    if (np == 0) return out;
    std::vector<const char*> a(np), b(np);
    std::vector<int> na(np), nb(np);
    for (int p = 0; p < np; ++p) {
        a[p] = pairs[p].first.c_str(); b[p] = pairs[p].second.c_str();
        na[p] = (int)pairs[p].first.size(); nb[p] = (int)pairs[p].second.size();
    }
    auto raise_ctx = [&](const char* where) { throw std::logic_error(std::string("ractip_amd::") + where + ": " + rh_last_error(ctx)); };
    if (rh_batch_upload(ctx, np, a.data(), na.data(), b.data(), nb.data()) != RH_OK) raise_ctx("solve_probabilities");
    if (rh_batch_compute(ctx) != RH_OK) raise_ctx("solve_probabilities");
    // three device-to-host copies for the whole batch, then unpack the padded layout on the host
    size_t tri_stride = 0, hp_stride = 0;
    int up_ld = 0, hp_ld = 0;
    if (rh_batch_layout(ctx, &tri_stride, &up_ld, &hp_stride, &hp_ld) != RH_OK) raise_ctx("solve_probabilities");
    std::vector<double> bp((size_t)2 * np * tri_stride), up((size_t)2 * np * up_ld), hp((size_t)np * hp_stride), z((size_t)3 * np);
    if (rh_batch_results_all(ctx, bp.data(), up.data(), hp.data(), z.data()) != RH_OK) raise_ctx("solve_probabilities");
    for (int p = 0; p < np; ++p) {
        const uint n1 = na[p], n2 = nb[p];
        PairProbabilities& r = out[p];
        const double* b1 = bp.data() + (size_t)(2 * p) * tri_stride;
        const double* b2 = bp.data() + (size_t)(2 * p + 1) * tri_stride;
        r.bp1.resize((size_t)(n1 + 1) * (n1 + 2) / 2);
        r.bp2.resize((size_t)(n2 + 1) * (n2 + 2) / 2);
        std::transform(b1, b1 + r.bp1.size(), r.bp1.begin(), [](double v) { return (float)v; });
        std::transform(b2, b2 + r.bp2.size(), r.bp2.begin(), [](double v) { return (float)v; });
        r.up1.assign(n1, VF(max_w)); r.up2.assign(n2, VF(max_w));
        for (uint i = 0; i < n1; ++i)
            for (uint w = 0; w < max_w; ++w) r.up1[i][w] = (float)up[(size_t)(2 * p) * up_ld + (size_t)i * max_w + w];
        for (uint i = 0; i < n2; ++i)
            for (uint w = 0; w < max_w; ++w) r.up2[i][w] = (float)up[(size_t)(2 * p + 1) * up_ld + (size_t)i * max_w + w];
        r.offset1 = make_offsets(n1); r.offset2 = make_offsets(n2);
        r.hp.assign(n1 + 1, VF(n2 + 1));
        const double* h = hp.data() + (size_t)p * hp_stride;
        for (uint i = 0; i <= n1; ++i)
            for (uint j = 0; j <= n2; ++j) {
                const float v = (float)h[(size_t)i * hp_ld + j];
                r.hp[i][j] = (!threshold_hp || v > th_hy_) ? v : 0.0f;   // plist entries with p > th_hy (:451-454)
            }
        r.logZ1 = z[3 * p]; r.logZ2 = z[3 * p + 1]; r.logZd = z[3 * p + 2];
    }
    return out;
}

}  // namespace ractip_amd
