// lse.h -- log-semiring primitives for 64-lane wavefronts (gfx950).
//
// The reference accumulates log(sum exp) one term at a time with
// Fast_LogPlusEquals (/root/reference/src/contrafold/LogSpace.hpp:232-237: exact
// log(exp(x-y)+1)+y in double, terms more than 30 below the running value or at
// the -2e20 sentinel are skipped).  Here every lane keeps a running (max, scaled
// sum) pair -- one exp per term, no log -- and the wavefront folds the 64 pairs
// with a max butterfly, one rescale and an add butterfly.  The two differ only in
// rounding (and in the reference's e^-30 truncation), far below the 1e-6 bar.
#pragma once
#include <hip/hip_runtime.h>
#include "score_model.h"

namespace rh {

constexpr double kNeg = RH_NEG_INF;
constexpr double kEmptyMax = -1e300;  // "no term yet": below every value incl. sums of sentinels

struct Lse {
    double m, s;  // represents m + log(s)
};

__device__ __forceinline__ Lse lse_empty() { return Lse{kEmptyMax, 0.0}; }

// fold one term: exactly one exp
__device__ __forceinline__ void lse_add(Lse& a, double x)
{
    const double d = x - a.m;
    const double e = exp(-fabs(d));
    const bool up = d > 0.0;
    a.s = up ? fma(a.s, e, 1.0) : a.s + e;
    a.m = up ? x : a.m;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sentinel normalisation: anything at or below NEG_INF/2 is "log 0" (LogSpace.hpp:234)
__device__ __forceinline__ double lse_norm(double v) { return v > kNeg / 2 ? v : kNeg; }

// wave-wide result, identical in every lane
__device__ __forceinline__ double lse_wave_finish(const Lse& a)
{
    const double M = wave_max(a.m);
    const double S = wave_sum(a.s * exp(a.m - M));
    return lse_norm(M + log(S));
}

// scalar helpers for the O(1)-term recurrences (uniform across the wave)
__device__ __forceinline__ double lse2(double a, double b)
{
    const double hi = fmax(a, b), lo = fmin(a, b);
    return lse_norm(hi + log1p(exp(lo - hi)));
}
__device__ __forceinline__ double lse3(double a, double b, double c)
{
    const double hi = fmax(a, fmax(b, c));
    return lse_norm(hi + log(exp(a - hi) + exp(b - hi) + exp(c - hi)));
}

}  // namespace rh
