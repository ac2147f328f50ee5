// pf_duplex_shim.cpp -- source-compatible stand-in for /root/reference/src/pf_duplex.h:25-28
//
//   extern double **pr_duplex;
//   double pf_duplex(const char *s1, const char *s2);
//   void free_pf_duplex();
//
// so that RactIP::rnaduplex's --duplex branch (/root/reference/src/ractip.cpp:390-398)
// links unchanged against libractip_pfduplex.so.  Same ownership as the original:
// pf_duplex allocates pr_duplex (rows 1..n1, each n2+1 doubles, row 0 NULL;
// pf_duplex.c:94-95) and returns log Z; free_pf_duplex releases it (:119-126).
//
// Scoring: RH_MODEL_VIENNA_BL -- the BL* tables that RactIP installs by default
// (src/boltzmann_param.c, copy_boltzmann_parameters at src/ractip.cpp:1566-1567) evaluated with
// ViennaRNA-1.8 loop-energy semantics, i.e. the 1.8 branch of src/pf_duplex.c:209-433.  PARITY UNPINNED:
// ViennaRNA itself is a third-party dependency that is absent and unversioned.
// The HAVE_VIENNA20 branch (src/pf_duplex.c:128-206, E_ExtLoop / E_IntLoop) needs tables only RNAlib holds; whoever has them
// selects it through the environment, which stands for the process-global parameter state of the original:
//   RACTIP_AMD_VIENNA_DEFAULTS = ViennaRNA parameter file with the library's built-in tables (e.g. rna_turner2004.par)
//   RACTIP_AMD_VIENNA_PARAMS   = the -P file (read_parameter_file, src/ractip.cpp:1567)
//   RACTIP_AMD_NO_BL_PARAM=1   = --no-bl-param            RACTIP_AMD_VIENNA_SEMANTICS = 1 (1.8) | 2 (2.x); default by the files
// RACTIP_AMD_DUPLEX_MODEL=contrafold selects the CONTRAfold duplex scores instead (what
// RactIP::contraduplex computes, src/ractip.cpp:225-245).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ractip_hot.h"

extern "C" {

double** pr_duplex = nullptr;
static int g_n1 = 0;
static rh_ctx* g_ctx = nullptr;
// the original keeps process-global state too (pr_duplex); the context is released when the process exits or the library is unloaded
static void release_context() { if (g_ctx) { rh_destroy(g_ctx); g_ctx = nullptr; } }

double pf_duplex(const char* s1, const char* s2)
{
    const int n1 = (int)std::strlen(s1), n2 = (int)std::strlen(s2);
    if (!g_ctx) {
        const char* dev = std::getenv("RACTIP_AMD_DEVICE");
        const char* mdl = std::getenv("RACTIP_AMD_DUPLEX_MODEL");
        const int model = (mdl && !std::strcmp(mdl, "contrafold")) ? RH_MODEL_CONTRAFOLD : RH_MODEL_VIENNA_BL;
        const char* dflt = std::getenv("RACTIP_AMD_VIENNA_DEFAULTS");
        const char* par = std::getenv("RACTIP_AMD_VIENNA_PARAMS");
        const char* nobl = std::getenv("RACTIP_AMD_NO_BL_PARAM");
        const char* sem = std::getenv("RACTIP_AMD_VIENNA_SEMANTICS");
        g_ctx = model == RH_MODEL_VIENNA_BL
                    ? rh_create_vienna(dev ? std::atoi(dev) : 0, dflt, !(nobl && std::atoi(nobl)), par, sem ? std::atoi(sem) : 0)
                    : rh_create(dev ? std::atoi(dev) : 0, model, nullptr);
        if (!g_ctx) {  // the original aborts inside ViennaRNA's space() on failure; do the same, loudly
            std::fprintf(stderr, "pf_duplex (ractip_amd): %s\n", rh_last_error(nullptr));
            std::abort();
        }
        std::atexit(release_context);
    }
    std::vector<double> hp((size_t)(n1 + 1) * (n2 + 1), 0.0);
    double logz = 0.0;
    if (n1 > 0 && n2 > 0 && rh_duplex(g_ctx, s1, n1, s2, n2, hp.data(), &logz) != RH_OK) {
        std::fprintf(stderr, "pf_duplex (ractip_amd): %s\n", rh_last_error(g_ctx));
        std::abort();
    }
    pr_duplex = (double**)std::calloc((size_t)n1 + 1, sizeof(double*));
    for (int i = 1; i <= n1; i++) {
        pr_duplex[i] = (double*)std::malloc(sizeof(double) * ((size_t)n2 + 1));
        std::memcpy(pr_duplex[i], hp.data() + (size_t)i * (n2 + 1), sizeof(double) * ((size_t)n2 + 1));
    }
    g_n1 = n1;
    return logz;
}

void free_pf_duplex()
{
    if (!pr_duplex) return;
    for (int i = 1; i <= g_n1; i++) std::free(pr_duplex[i]);
    std::free(pr_duplex);
    pr_duplex = nullptr;
}

}  // extern "C"
