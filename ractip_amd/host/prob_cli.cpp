// prob_cli.cpp -- drives the C++ adapters exactly as RactIP::solve does
// (/root/reference/src/ractip.cpp:536-548) and prints the float matrices it would
// hand to the ILP, one value per line, for tests/test_gpu_host_adapter.py.
//   [RACTIP_DEVICES=0,1,..] prob_cli contrafold SEQ | rnafold SEQ MAX_W | contraduplex S1 S2 TH | rnaduplex S1 S2 | cofold S1 S2 | pfduplex S1 S2 | solve S1 S2 [S1 S2 ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ractip_prob.hpp"

extern "C" {
extern double** pr_duplex;
double pf_duplex(const char*, const char*);
void free_pf_duplex();
}

using namespace ractip_amd;

static void dump_bp(const VF& bp, const VI& off, const VVF& up)
{
    std::printf("offset %zu\n", off.size());
    for (int o : off) std::printf("%d\n", o);
    std::printf("bp %zu\n", bp.size());
    for (float v : bp) std::printf("%.9g\n", v);
    std::printf("up %zu\n", up.size());
    for (const VF& r : up) std::printf("%.9g\n", r[0]);
}
static void dump_hp(const VVF& hp)
{
    std::printf("hp %zu %zu\n", hp.size(), hp.empty() ? (size_t)0 : hp[0].size());
    for (const VF& r : hp)
        for (float v : r) std::printf("%.9g\n", v);
}

int main(int argc, char** argv)
{
    try {
        if (argc < 3) throw std::logic_error("usage");
        const std::string mode = argv[1];
        if (mode == "bounds") {   // bounds NUM PARTS: the contiguous blocks of the in-process shard (no GPU needed)
            const int num = std::atoi(argv[2]), parts = argc > 3 ? std::atoi(argv[3]) : 1;
            for (int k = 0; k < parts; k++) { const auto b = ProbabilityEngine::shard_bounds(num, k, parts); std::printf("%d %d\n", b.first, b.second); }
            return 0;
        }
        if (mode == "pfduplex") {
            const double z = pf_duplex(argv[2], argv[3]);
            const size_t n1 = std::strlen(argv[2]), n2 = std::strlen(argv[3]);
            std::printf("logZ %.17g\nhp %zu %zu\n", z, n1 + 1, n2 + 1);
            for (size_t j = 0; j <= n2; j++) std::printf("0\n");
            for (size_t i = 1; i <= n1; i++)
                for (size_t j = 0; j <= n2; j++) std::printf("%.17g\n", j ? pr_duplex[i][j] : 0.0);
            free_pf_duplex();
            return 0;
        }
        // RACTIP_DEVICES=0,1,... : one context and host thread per listed device for the batched modes (the in-process shard)
        std::vector<int> devices;
        if (const char* e = std::getenv("RACTIP_DEVICES"))
            for (const char* p = e; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        if (devices.empty()) devices.push_back(0);
        ProbabilityEngine en(devices, mode == "contraduplex" ? (float)std::atof(argv[4]) : 0.1f);
        {   // RactIP's -P / --no-bl-param (src/ractip.cpp:1563-1567), through the environment the pf_duplex shim reads too
            const char* dflt = std::getenv("RACTIP_AMD_VIENNA_DEFAULTS");
            const char* par = std::getenv("RACTIP_AMD_VIENNA_PARAMS");
            const char* nobl = std::getenv("RACTIP_AMD_NO_BL_PARAM");
            const char* sem = std::getenv("RACTIP_AMD_VIENNA_SEMANTICS");
            if (dflt || par || nobl || sem)
                en.set_vienna_parameters(dflt ? dflt : "", !(nobl && std::atoi(nobl)), par ? par : "", sem ? std::atoi(sem) : 0);
        }
        if (mode == "contrafold") {
            VF bp; VI off; VVF up;
            en.contrafold(argv[2], bp, off, up);
            dump_bp(bp, off, up);
        } else if (mode == "rnafold") {   // rnafold SEQ MAX_W
            VF bp; VI off; VVF up;
            const unsigned mw = argc > 3 ? (unsigned)std::atoi(argv[3]) : 1u;
            if (argc > 4) en.rnafold(argv[2], argv[4], bp, off, up, mw);   // rnafold SEQ MAX_W STRUCTURE-LINE
            else en.rnafold(argv[2], bp, off, up, mw);
            std::printf("offset %zu\n", off.size());
            for (int o : off) std::printf("%d\n", o);
            std::printf("bp %zu\n", bp.size());
            for (float v : bp) std::printf("%.9g\n", v);
            std::printf("up %zu\n", up.size() * mw);
            for (const VF& r : up) for (float v : r) std::printf("%.9g\n", v);
        } else if (mode == "contraduplex") {
            VVF hp; en.contraduplex(argv[2], argv[3], hp); dump_hp(hp);
        } else if (mode == "cofold") {
            VVF hp;
            if (argc > 5) en.rnaduplex_cofold(argv[2], argv[4], argv[3], argv[5], hp);   // cofold S1 S2 STR1 STR2
            else en.rnaduplex_cofold(argv[2], argv[3], hp);
            dump_hp(hp);
        } else if (mode == "rnaduplex") {
            VVF hp; en.rnaduplex(argv[2], argv[3], hp); dump_hp(hp);
        } else if (mode == "solve_default" || mode == "solve_default_duplex") {   // solve_default[_duplex] MAX_W S1 S2 [S1 S2 ...]
            const unsigned mw = (unsigned)std::atoi(argv[2]);
            std::vector<std::pair<std::string, std::string>> pairs;
            for (int k = 3; k + 1 < argc; k += 2) pairs.emplace_back(argv[k], argv[k + 1]);
            for (const PairProbabilities& r : en.solve_probabilities_default(pairs, mw, mode == "solve_default_duplex")) {
                std::printf("pair %.17g %.17g %.17g\n", r.logZ1, r.logZ2, r.logZd);
                std::printf("bp %zu\n", r.bp1.size());
                for (float v : r.bp1) std::printf("%.9g\n", v);
                std::printf("up %zu\n", r.up2.size() * mw);
                for (const VF& row : r.up2) for (float v : row) std::printf("%.9g\n", v);
                dump_hp(r.hp);
            }
        } else if (mode == "solve") {
            std::vector<std::pair<std::string, std::string>> pairs;
            for (int k = 2; k + 1 < argc; k += 2) pairs.emplace_back(argv[k], argv[k + 1]);
            for (const PairProbabilities& r : en.solve_probabilities(pairs)) {
                std::printf("pair %.17g %.17g %.17g\n", r.logZ1, r.logZ2, r.logZd);
                dump_bp(r.bp1, r.offset1, r.up1);
                dump_bp(r.bp2, r.offset2, r.up2);
                dump_hp(r.hp);
            }
        } else throw std::logic_error("unknown mode");
    } catch (const std::logic_error& e) {  // RactIP's main() handler, src/ractip.cpp:1684-1687
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
