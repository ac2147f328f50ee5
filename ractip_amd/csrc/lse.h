// lse.h -- log-semiring primitives for 64-lane wavefronts (gfx950).
//
// The reference accumulates log(sum exp) one term at a time with
// Fast_LogPlusEquals (/root/reference/src/contrafold/LogSpace.hpp:232-237: exact
// log(exp(x-y)+1)+y in double, terms more than 30 below the running value or at
// the -2e20 sentinel are skipped).  Here every lane keeps a running (max, scaled
// sum) pair and folds terms in register-resident groups of U: all U operand loads
// are issued before any arithmetic (memory-level parallelism), the group maximum
// is taken with U fmax, and U independent exps follow -- one exp per term, no log.
// The wavefront then folds the 64 pairs with a max butterfly, one rescale and an
// add butterfly.  The result differs from the reference's only in rounding (and in
// its e^-30 truncation), far below the 1e-6 bar.
#pragma once
#include <hip/hip_runtime.h>
#include "score_model.h"

namespace rh {

constexpr double kNeg = RH_NEG_INF;
constexpr double kEmptyMax = -1e300;  // "no term": below every value incl. sums of sentinels

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

// fold U register-resident terms (absent terms = kEmptyMax): U+1 exps, all independent
template <int U>
__device__ __forceinline__ void lse_add_group(Lse& a, const double (&x)[U])
{
    double gm = x[0];
#pragma unroll
    for (int u = 1; u < U; u++) gm = fmax(gm, x[u]);
    const double nm = fmax(a.m, gm);
    double s = a.s * exp(a.m - nm);
#pragma unroll
    for (int u = 0; u < U; u++) s += exp(x[u] - nm);
    a.m = nm;
    a.s = s;
}

// stream x_k = r1[k] + r2[k], k in [k0,k1), lanes strided, U loads of each row in flight
template <int U>
__device__ __forceinline__ void lse_stream2(Lse& a, const double* __restrict__ r1, const double* __restrict__ r2,
                                            int k0, int k1, int lane)
{
    for (int base = k0 + lane; base < k1; base += 64 * U) {
        double p[U], q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = base + 64 * u;
            const bool ok = k < k1;
            p[u] = ok ? r1[k] : kEmptyMax;
            q[u] = ok ? r2[k] : 0.0;
        }
        double x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = p[u] + q[u];
        lse_add_group<U>(a, x);
    }
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

// three independent accumulators folded together: the butterflies interleave, so the
// cross-lane latency of one hides behind the others
__device__ __forceinline__ void lse_wave_finish3(const Lse& a, const Lse& b, const Lse& c, double& ra, double& rb, double& rc)
{
    double ma = a.m, mb = b.m, mc = c.m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ta = __shfl_xor(ma, o, 64), tb = __shfl_xor(mb, o, 64), tc = __shfl_xor(mc, o, 64);
        ma = fmax(ma, ta); mb = fmax(mb, tb); mc = fmax(mc, tc);
    }
    double sa = a.s * exp(a.m - ma), sb = b.s * exp(b.m - mb), sc = c.s * exp(c.m - mc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ta = __shfl_xor(sa, o, 64), tb = __shfl_xor(sb, o, 64), tc = __shfl_xor(sc, o, 64);
        sa += ta; sb += tb; sc += tc;
    }
    ra = lse_norm(ma + log(sa));
    rb = lse_norm(mb + log(sb));
    rc = lse_norm(mc + log(sc));
}
__device__ __forceinline__ void lse_wave_finish2(const Lse& a, const Lse& b, double& ra, double& rb)
{
    double ma = a.m, mb = b.m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ta = __shfl_xor(ma, o, 64), tb = __shfl_xor(mb, o, 64);
        ma = fmax(ma, ta); mb = fmax(mb, tb);
    }
    double sa = a.s * exp(a.m - ma), sb = b.s * exp(b.m - mb);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ta = __shfl_xor(sa, o, 64), tb = __shfl_xor(sb, o, 64);
        sa += ta; sb += tb;
    }
    ra = lse_norm(ma + log(sa));
    rb = lse_norm(mb + log(sb));
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
