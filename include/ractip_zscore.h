/* ractip_zscore.h -- host-side helpers of the z-score shard.
 *
 * The z-score loop of the reference (/root/reference/src/ractip.cpp:1624-1670) draws
 * `num_shuffling` dinucleotide-preserving shuffles from libc random(); the shuffles do not
 * depend on any DP or ILP result, so rank 0 can replay the RNG stream first and hand all
 * pairs to the GPUs at once.  These entry points are pure host code (libractip_prob.so).
 */
#ifndef RACTIP_ZSCORE_H
#define RACTIP_ZSCORE_H

#ifdef __cplusplus
extern "C" {
#endif

/* One k=2 (dinucleotide-preserving) shuffle of s[0..l) into t, consuming libc random() in
 * exactly the order uShuffle::shuffle(s, t, l, 2) does (src/ushuffle.c:127-268: Euler graph
 * on the letters, Wilson's random arborescence, edge permutation, walk). */
void rh_dishuffle(const char* s, char* t, int l);

/* Replays src/ractip.cpp:1636-1643: srandom(seed); for it in 0..num-1: shuffle s1 if
 * mode in {1,12}, shuffle s2 if mode in {2,12} (always from the ORIGINAL sequences).
 * out1/out2: num rows of (n1+1)/(n2+1) chars, NUL-terminated.  Returns 0, or -1 on a bad mode. */
int rh_zscore_shuffles(const char* s1, const char* s2, int mode, int num, unsigned seed,
                       char* out1, char* out2);

/* Float accumulation of the z-score statistics in iteration order, bit-for-bit as
 * src/ractip.cpp:1655-1669: returns (e - m)/sqrt(v) for the gathered per-iteration energies. */
float rh_zscore_from_energies(const float* ee, int num, float e_native);

#ifdef __cplusplus
}
#endif
#endif
