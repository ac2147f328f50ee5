// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI driver around the reference's own, unmodified CONTRAfold engines
// (InferenceEngine<RealT>, DuplexEngine<RealT>), compiled from the sources where
// they lie under /root/reference/src/contrafold (see oracle/Makefile).  Nothing
// from the reference is copied into this repository: this file only #includes
// the reference headers by path at build time and calls their public API in the
// same order RactIP does:
//   RactIP::contrafold    /root/reference/src/ractip.cpp:195-223
//   RactIP::contraduplex  /root/reference/src/ractip.cpp:225-245
//
// Built into oracle/_ref/libref_contrafold.so (git-ignored, travels to the GPU
// box with the snapshot).  Used (a) to pin the CPU restatement in
// oracle/cf_oracle.c, (b) to generate tests/golden/*.json via
// oracle/gen_golden.py, (c) optionally as bench.py's cpu_baseline
// ("kind": "reference").
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <list>
#include <map>
#include <queue>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

// The intermediate DP tables are private members of the reference classes.
// For table-level parity checks of the restatement / kernels this translation
// unit is compiled with g++ -fno-access-control (see oracle/Makefile).
#include "contrafold/SStruct.hpp"
#include "contrafold/InferenceEngine.hpp"
#include "contrafold/DuplexEngine.hpp"
#include "contrafold/Defaults.ipp"

namespace {

template <class RealT>
double run_inference(const char* seq, double* post, double* tables8, double* f5)
{
    std::string s(seq);
    SStruct ss("unknown", s);
    ParameterManager<RealT> pm;
    InferenceEngine<RealT> en(false);
    std::vector<RealT> w = GetDefaultComplementaryValues<RealT>();
    en.RegisterParameters(pm);
    en.LoadValues(w);
    en.LoadSequence(ss);
    en.ComputeInside();
    en.ComputeOutside();
    en.ComputePosterior();
    const int L = (int)s.size();
    const int SIZE = (L + 1) * (L + 2) / 2;
    if (post) {
        std::vector<RealT> p;
        en.GetPosterior(RealT(0), p);
        for (int i = 0; i < SIZE; i++) post[i] = (double)p[i];
    }
    if (tables8) {
        const std::vector<RealT>* t[6] = {&en.FCi, &en.FMi, &en.FM1i, &en.FCo, &en.FMo, &en.FM1o};
        for (int k = 0; k < 6; k++)
            for (int i = 0; i < SIZE; i++) tables8[(size_t)k * SIZE + i] = (double)(*t[k])[i];
    }
    if (f5) {
        for (int i = 0; i <= L; i++) { f5[i] = (double)en.F5i[i]; f5[L + 1 + i] = (double)en.F5o[i]; }
    }
    return (double)en.ComputeLogPartitionCoefficient();
}

template <class RealT>
void run_duplex(const char* s1, const char* s2, double* post, double* inside, double* outside, double* logz2)
{
    std::string a(s1), b(s2);
    SStruct ss1("unknown", a), ss2("unknown", b);
    ParameterManager<RealT> pm;
    DuplexEngine<RealT> en(false);
    std::vector<RealT> w = GetDefaultComplementaryValues<RealT>();
    en.RegisterParameters(pm);
    en.LoadValues(w);
    en.LoadSequence(ss1, ss2);
    RealT zi = en.ComputeInside();
    RealT zo = en.ComputeOutside();
    en.ComputePosterior();
    const int SIZE = ((int)a.size() + 1) * ((int)b.size() + 1);
    if (post) {
        std::vector<RealT> p;
        en.GetPosterior(RealT(0), p);
        for (int i = 0; i < SIZE; i++) post[i] = (double)p[i];
    }
    if (inside)  for (int i = 0; i < SIZE; i++) inside[i] = (double)en.inside[i];
    if (outside) for (int i = 0; i < SIZE; i++) outside[i] = (double)en.outside[i];
    logz2[0] = (double)zi;
    logz2[1] = (double)zo;
}

}  // namespace

extern "C" {

// number of logical parameters (708) of the active feature set
int ref_num_params()
{
    ParameterManager<double> pm;
    InferenceEngine<double> en(false);
    en.RegisterParameters(pm);
    return (int)pm.GetNumLogicalParameters();
}

// name (<=63 chars) and default "complementary" value of logical parameter idx
void ref_param(int idx, char* name64, double* value)
{
    ParameterManager<double> pm;
    InferenceEngine<double> en(false);
    en.RegisterParameters(pm);
    std::vector<double> w = GetDefaultComplementaryValues<double>();
    std::vector<std::string> names = pm.GetNames();
    std::strncpy(name64, names[idx].c_str(), 63);
    name64[63] = 0;
    *value = w[idx];
}

// McCaskill inside/outside/posterior, reference engine.  post: T(n) doubles
// (reference triangular layout), tables: 6*T(n) doubles (FCi,FMi,FM1i,FCo,FMo,FM1o)
// or NULL, f5: 2*(n+1) doubles (F5i,F5o) or NULL.  Returns logZ.
double ref_inference(const char* seq, int use_float, double* post, double* tables, double* f5)
{
    return use_float ? run_inference<float>(seq, post, tables, f5)
                     : run_inference<double>(seq, post, tables, f5);
}

// Duplex inside/outside/posterior, reference engine.  All arrays (n1+1)*(n2+1)
// row-major or NULL.  logz2[0] = inside logZ, logz2[1] = outside logZ.
void ref_duplex(const char* s1, const char* s2, int use_float,
                double* post, double* inside, double* outside, double* logz2)
{
    if (use_float) run_duplex<float>(s1, s2, post, inside, outside, logz2);
    else           run_duplex<double>(s1, s2, post, inside, outside, logz2);
}

}  // extern "C"
