/* ractip_hot.h -- C ABI of the MI355X-native probability-matrix engine for RactIP.
 *
 * This is the drop-in boundary of SURVEY.md section 8(b): plain C, no C++ types,
 * no exceptions, caller-owned output buffers, one context per host thread / GPU.
 * Every entry point names the reference interface it replaces (paths relative
 * to /root/reference).  The C++ adapters that rebuild the reference's
 * VF/VI/VVF containers on top of this ABI live in ractip_amd/host/.
 *
 * Status codes: 0 = ok, negative = error (see rh_last_error).
 * Layouts (identical to the reference):
 *   bp  : T(n) = (n+1)(n+2)/2 doubles, element (i,j), 1 <= i < j <= n, at
 *         offset[i]+j with offset[i] = i*(2(n+1)-i-1)/2   (src/ractip.cpp:254-257,
 *         src/contrafold/InferenceEngine.ipp:316); all other entries 0.
 *   up  : n*max_w doubles row-major, up[i*max_w+w] = P(i..i+w unpaired), 0-based i
 *         (src/ractip.cpp:213-222 for max_w=1; src/ractip.cpp:370-375).
 *   hp  : (n1+1)*(n2+1) doubles row-major, 1-based, row 0 / column 0 zero
 *         (src/ractip.cpp:393-397, 236-244).
 */
#ifndef RACTIP_HOT_H
#define RACTIP_HOT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rh_ctx rh_ctx;

/* scoring model selector */
#define RH_MODEL_CONTRAFOLD 0 /* --contrafold path: src/ractip.cpp:195-246 */
#define RH_MODEL_VIENNA_BL 1  /* default CLI path: pf_fold bp + pf_unstru up (src/ractip.cpp:248-382) and the --duplex
                                 pf_duplex hp (src/ractip.cpp:390-398) with the BL* energies of src/boltzmann_param.c in
                                 ViennaRNA-1.8 semantics; PARITY UNPINNED (ViennaRNA is absent and unversioned) */

#define RH_OK 0
#define RH_ERR_ARG (-1)
#define RH_ERR_HIP (-2)
#define RH_ERR_OOM (-3)
#define RH_ERR_PARAM (-4)
#define RH_ERR_UNSUPPORTED (-5)

/* Create a context bound to HIP device `device`.  `param_file` = NULL loads the
 * bundled default weights (ractip_amd/data/contrafold_complementary.params, the
 * values of GetDefaultComplementaryValues, src/contrafold/Defaults.ipp:7-723; for
 * RH_MODEL_VIENNA_BL ractip_amd/data/vienna_bl_star.params, the BL* tables of
 * src/boltzmann_param.c).
 * Replaces the per-call engine construction of src/ractip.cpp:199-206, 229-234.
 * Returns NULL on failure; rh_last_error(NULL) then describes why. */
rh_ctx* rh_create(int device, int model, const char* param_file);
/* RH_MODEL_VIENNA_BL with the energy tables installed the way RactIP::run installs them (src/ractip.cpp:1563-1567):
 *   library defaults  ->  copy_boltzmann_parameters() unless --no-bl-param (use_bl_param)  ->  read_parameter_file(param_file).
 * defaults_file (or NULL): a ViennaRNA parameter file standing for the tables compiled into the user's RNAlib -- ViennaRNA is
 * not part of the reference, so its built-in Turner tables are not in this library; tables no source provides are ZERO.
 * param_file (or NULL): the -P file, ViennaRNA "## RNAfold parameter file" (1.x or v2.0 layout) or the flat dump format of
 * ractip_amd/data/vienna_bl_star.params.  rh_create(dev, RH_MODEL_VIENNA_BL, f) = rh_create_vienna(dev, NULL, 1, f, 0).
 * semantics: how the loop energies are evaluated --
 *   RH_VIENNA_SEM_18: ViennaRNA-1.8 LoopEnergy + dangle sums, the 1.8 branch of src/pf_duplex.c:209-433 (all kernels);
 *   RH_VIENNA_SEM_20: ViennaRNA-2.x E_IntLoop / E_ExtLoop / E_MLstem / E_Hairpin, the HAVE_VIENNA20 branch the reference's
 *     CMake selects (src/pf_duplex.c:128-206; CMakeLists.txt:28): mismatch_interior_1n / _23, mismatch_exterior / _multi,
 *     tri/tetra/hexaloop energies replacing the hairpin energy.  Log-space kernels only (rh_set_mode(LINEAR) is refused);
 *   RH_VIENNA_SEM_AUTO: 2.x if defaults_file or param_file is a v2.0 parameter file, else 1.8.
 * PARITY UNPINNED in both semantics. */
#define RH_VIENNA_SEM_AUTO 0
#define RH_VIENNA_SEM_18 1
#define RH_VIENNA_SEM_20 2
rh_ctx* rh_create_vienna(int device, const char* defaults_file, int use_bl_param, const char* param_file, int semantics);
/* RH_VIENNA_SEM_18 / RH_VIENNA_SEM_20 of a RH_MODEL_VIENNA_BL context, 0 for the CONTRAfold model */
int rh_vienna_semantics(const rh_ctx* ctx);
void rh_destroy(rh_ctx* ctx);
const char* rh_last_error(const rh_ctx* ctx);

/* Arithmetic path of the McCaskill sweeps.  AUTO (default): scaled linear-space kernels,
 * and if any sequence of the batch leaves the double range (detected on device) the whole
 * batch is recomputed by the log-space kernels, which follow the reference's own
 * log-space arithmetic (src/contrafold/LogSpace.hpp).  LOG / LINEAR force one path
 * (LINEAR reports RH_OK even if values overflowed: testing only). */
#define RH_MODE_AUTO 0
#define RH_MODE_LOG 1
#define RH_MODE_LINEAR 2
int rh_set_mode(rh_ctx* ctx, int mode);
/* path taken by the last compute: 1 = linear, 2 = log-space, 3 = linear, then log-space fallback */
int rh_last_path(const rh_ctx* ctx);
/* the same for the sweeps that produced hp (duplex or two-molecule ensemble) */
int rh_last_hybrid_path(const rh_ctx* ctx);

/* Base-pairing probabilities of one sequence.  Replaces the body of
 * RactIP::contrafold up to GetPosterior (src/ractip.cpp:199-211:
 * ComputeInside/ComputeOutside/ComputePosterior/GetPosterior(0,...)) and, for
 * RH_MODEL_VIENNA_BL, pf_fold + export_bppm of RactIP::rnafold
 * (src/ractip.cpp:288-304, 351-367).  `constraint` (RH_MODEL_VIENNA_BL only, or NULL): the string RactIP hands to
 * pf_fold when use_constraint_ is set (src/ractip.cpp:271-291), in ViennaRNA's fold_constrained alphabet:
 * 'x' never pairs, '<' / '>' pairs only with a later / earlier letter, matched '(' ')' keeps that pair and removes every
 * pair inconsistent with it, '|' and '.' do not restrict the partition function; shorter strings are padded with '.'.
 * A forced pair of non-complementary letters is rejected (RH_ERR_ARG). */
int rh_bpp(rh_ctx* ctx, const char* seq, int n, const char* constraint,
           double* bp_tri, double* logZ);

/* Accessibility, up[i*max_w+w] = P(letters i..i+w unpaired), n*max_w doubles.  RH_MODEL_CONTRAFOLD: max_w must be 1
 * (src/ractip.cpp:213-222, up[i] = max(0, 1 - sum_j bp(i,j))).  RH_MODEL_VIENNA_BL: any 1 <= max_w <= 64; replaces
 * pf_unstru and the H+I+M+E sum of src/ractip.cpp:370-375.  Sets the context's max_w (see rh_set_max_w). */
int rh_unpaired(rh_ctx* ctx, const char* seq, int n, int max_w, double* up);

/* Number of accessibility widths the batched / rh_fold forms compute (RactIP's --max-w, src/ractip.cpp:546-547:
 * rnafold(..., std::max(1, max_w_))).  Default 1 for RH_MODEL_CONTRAFOLD (only value allowed), 15 for
 * RH_MODEL_VIENNA_BL.  Takes effect at the next upload / single-sequence call. */
int rh_set_max_w(rh_ctx* ctx, int max_w);
int rh_get_max_w(const rh_ctx* ctx);

/* Source of the hybridization matrix hp under RH_MODEL_VIENNA_BL (rh_duplex and the batched form):
 *   RH_HYBRID_DUPLEX (default): pf_duplex -- duplexes only, the --duplex branch (src/ractip.cpp:390-398);
 *   RH_HYBRID_COFOLD: the joint ensemble of both molecules, the default branch (src/ractip.cpp:400-458):
 *     co_pf_fold(s1+s2) with cut_point = n1+1, hp[i][j] = pr(i, n1+j).  The reference keeps only entries
 *     with p > th_hy (assign_plist_from_pr); this ABI returns the dense block and the adapter thresholds.
 *     logZ = log partition function of the two-molecule ensemble.  PARITY UNPINNED like the rest of this model. */
#define RH_HYBRID_DUPLEX 0
#define RH_HYBRID_COFOLD 1
int rh_set_hybrid(rh_ctx* ctx, int hybrid);

/* Both of the above from ONE inside/outside pass -- the whole of RactIP::contrafold (src/ractip.cpp:199-222) or of
 * the accessibility overload of RactIP::rnafold (src/ractip.cpp:308-382).  bp_tri, up (n*max_w doubles, max_w as set
 * by rh_set_max_w) or logZ may be NULL. */
int rh_fold(rh_ctx* ctx, const char* seq, int n, double* bp_tri, double* up, double* logZ);
/* rh_fold under a structure constraint (see rh_bpp): the whole accessibility overload of RactIP::rnafold with
 * use_constraint_ (src/ractip.cpp:308-382). */
int rh_fold_constrained(rh_ctx* ctx, const char* seq, int n, const char* constraint, double* bp_tri, double* up, double* logZ);

/* Hybridization probabilities of a pair.  Replaces RactIP::contraduplex
 * (src/ractip.cpp:225-245: DuplexEngine ComputeInside/Outside/Posterior) and, for
 * RH_MODEL_VIENNA_BL, pf_duplex + pr_duplex of RactIP::rnaduplex (src/ractip.cpp:390-398). */
int rh_duplex(rh_ctx* ctx, const char* s1, int n1, const char* s2, int n2,
              double* hp, double* logZ);

/* The two-molecule ensemble under a structure constraint: what the default branch of RactIP::rnaduplex hands to
 * co_pf_fold when use_constraint_ is set (src/ractip.cpp:405-447): `constraint` has n1+n2 characters over s1+s2 in the
 * fold_constrained alphabet of rh_bpp (RactIP writes '(' for '[' of s1, ')' for ']' of s2 and 'x' for every letter that
 * is paired inside its own molecule).  RH_MODEL_VIENNA_BL only; independent of rh_set_hybrid. */
int rh_cofold_constrained(rh_ctx* ctx, const char* s1, int n1, const char* s2, int n2, const char* constraint,
                          double* hp, double* logZ);

/* ---- batched, device-resident form (z-score loop src/ractip.cpp:1638-1657 and bench) ----
 * A batch is `npairs` independent (s1,s2) pairs; for each the engine computes
 * bp(s1), bp(s2), up(s1), up(s2) and hp(s1,s2) -- everything RactIP::solve needs
 * before the ILP (src/ractip.cpp:536-548).
 *   rh_batch_upload   : encode + copy the sequences to HBM, (re)allocate tables
 *   rh_batch_compute  : run all DP kernels (inputs already resident); blocks until done
 *   rh_batch_results  : copy dense results of pair `p` to caller buffers (any may be NULL)
 *   rh_batch_candidates: thresholded sparse results of pair `p` -- the scans of
 *                       src/ractip.cpp:557-568, 578-589, 598-608, 621-627 done on device.
 */
int rh_batch_upload(rh_ctx* ctx, int npairs,
                    const char* const* s1, const int* n1,
                    const char* const* s2, const int* n2);
int rh_batch_compute(rh_ctx* ctx);
int rh_batch_results(rh_ctx* ctx, int p,
                     double* bp1_tri, double* bp2_tri, double* up1, double* up2,
                     double* hp, double* logZ3 /* logZ(s1), logZ(s2), logZ(duplex) */);

/* Per-pair scalars of the whole batch in one copy: out[3p..3p+2] = logZ(s1), logZ(s2),
 * logZ(duplex) -- what a z-score shard gathers across GPUs (src/ractip.cpp:1655-1663). */
int rh_batch_logz(rh_ctx* ctx, double* out);

typedef struct rh_cand {
    int i, j; /* 1-based letters; for `up`: i = 0-based position, j = width index (region i..i+j, src/ractip.cpp:621-627) */
    float p;  /* probability narrowed to float exactly as the reference does (src/ractip.cpp:82-83) */
} rh_cand;
/* which: 0 = bp1 (p > th), 1 = bp2, 2 = hp, 3 = up1, 4 = up2.  Writes at most `cap`
 * entries in row-major scan order and returns the total number found (>= 0) or an error. */
int rh_batch_candidates(rh_ctx* ctx, int p, int which, float threshold, rh_cand* out, int cap);

/* The same scan for EVERY pair of the batch in two kernels and three copies (the z-score loop needs 5 scans
 * for each of 1000 pairs): out holds the concatenated lists, pair p owns out[first[p] .. first[p+1]) (first has
 * npairs+1 entries).  Returns the total number found (>= 0; only min(total, cap) entries are written) or an error. */
int rh_batch_candidates_all(rh_ctx* ctx, int which, float threshold, rh_cand* out, int cap, int* first);

/* Dense results of the whole batch in three copies, in the padded device layout described by rh_batch_layout:
 *   bp [2*npairs][tri_stride]  (sequence 2p = s1 of pair p; each table in the reference's triangular layout for ITS n)
 *   up [2*npairs][up_ld]       (n*max_w entries used, row-major position x width)
 *   hp [npairs][hp_stride] with row pitch hp_ld       logz [3*npairs]
 * Any pointer may be NULL. */
int rh_batch_layout(rh_ctx* ctx, size_t* tri_stride, int* up_ld, size_t* hp_stride, int* hp_ld);
int rh_batch_results_all(rh_ctx* ctx, double* bp, double* up, double* hp, double* logz);

/* Page-locked host memory for result buffers (hipHostMalloc): device-to-host copies into it run at the full DMA rate and,
 * issued from two contexts, overlap the other context's kernels (the reference's result containers are plain std::vector,
 * src/ractip.cpp:82-83; a host that wants the dense matrices at PCIe speed hands these buffers to rh_batch_results_all).
 * rh_host_alloc returns NULL on failure; any context's device is fine. */
void* rh_host_alloc(rh_ctx* ctx, size_t bytes);
void rh_host_free(rh_ctx* ctx, void* p);

/* Device time (ms, HIP events on the context's streams) spent by the last
 * rh_batch_compute in: [0] McCaskill inside sweep, [1] McCaskill outside sweep
 * (+posterior), [2] duplex sweeps, [3] whole compute.  Launch counts in n_launch[0..2]. */
int rh_batch_timings(rh_ctx* ctx, double ms[4], int n_launch[3]);

/* Names (as a kernel trace prints them) of the sweep kernels the last rh_batch_compute launched, per phase
 * [0] McCaskill inside, [1] outside (+posterior), [2] duplex: the per-diagonal kernel, the block-product kernel ("" if
 * none) and how many of the phase's n_launch were block-product launches. */
int rh_batch_kernels(rh_ctx* ctx, const char* fine[3], const char* far[3], int n_far[3]);

/* Measurement aid: rh_set_overlap(ctx, 0) makes rh_batch_compute run the duplex sweeps, the McCaskill inside sweep and
 * the outside sweep one after the other instead of overlapping the duplex stream with the McCaskill stream, so that
 * rh_batch_timings returns each phase's device time with nothing else on the GPU (what a kernel trace reports per
 * kernel).  Results are unchanged.  Default: overlap on. */
int rh_set_overlap(rh_ctx* ctx, int on);

/* Measurement aid: rh_set_kernel_timing(ctx, cls) brackets every launch of ONE class of sweep kernels by a HIP event pair on its
 * stream (cls 0 = McCaskill inside sweep kernel, 1 = its block products incl. operand packing, 2 = outside sweep kernel, 3 = its
 * block products, 4 = duplex sweep kernel; -1 = off, the default) -- launch graphs are bypassed meanwhile.  After rh_batch_compute,
 * rh_kernel_times returns how many launches of that class ran and the sum of their durations: the live counterpart of the
 * per-kernel average of a rocprofv3 kernel trace (bench.py: roofline.avg_launch_us).  Results are unchanged. */
int rh_set_kernel_timing(rh_ctx* ctx, int cls);
int rh_kernel_times(rh_ctx* ctx, int* n_launches, double* total_ms);

/* Which problems of the last rh_batch_compute left the double range on the scaled linear path and were recomputed by the
 * log-space kernels (rh_last_path / rh_last_hybrid_path == 3): which = 0: sequence indices (2p = s1 of pair p), 1: pair
 * indices of the duplex.  Only those problems are recomputed; every other problem keeps its linear-path result.  Writes at
 * most `cap` indices, returns how many there are.  (The reference's log-space arithmetic, LogSpace.hpp:232-244, never
 * leaves its range; this is how the fast path keeps that guarantee.)
 * which = 2: the sequence indices that left the range with the default scale exponent and were recomputed on the linear kernels
 * with another one (CONTRAfold model: 0.45, 1.5 or 0 per unit span instead of 0.12) -- they are NOT in the list of which = 0.
 * With rh_set_scale_memory(ctx, 1): when one exponent held more than half of a batch of at least eight sequences, the next
 * rh_batch_compute starts on it.
 * Vienna-BL model: the whole batch is run again with another exponent (0.7, 1.8 or 0 instead of 0.28); which = 2 then lists the
 * sequences that made it necessary.
 * which = 3: the pair indices whose DUPLEX sweeps left the range with the default exponent (0.65 per unit of i + (L2+1-j)) and were
 * recomputed on the linear duplex kernels with another one (1.3, 2.2, 0.3 or 0; CONTRAfold model) -- they are NOT in the list of
 * which = 1. */
int rh_batch_fallbacks(rh_ctx* ctx, int which, int* out, int cap);

/* Scale-exponent memory (default OFF).  Off: every rh_batch_compute starts on the default exponent, so a sequence's result bits
 * depend on its own letters only (the reference, src/ractip.cpp:195-245, is deterministic per input) and sharded + gathered
 * results equal single-context results bit for bit whatever each context computed before.  On: a stream of batches of one kind
 * (e.g. long, very stable RNAs) pays the failed first pass once instead of once per batch; results then differ by a few ulp
 * with the context's history, which voids the bit-for-bit shard guarantee.  Turning it off also forgets the remembered exponent. */
int rh_set_scale_memory(rh_ctx* ctx, int on);

/* Device pointers of the last batch (for callers that keep results on the GPU):
 * bp tables [2*npairs][tri_stride] (sequence 2p = s1 of pair p, 2p+1 = s2),
 * hp tables [npairs][hp_stride]. */
int rh_batch_device_views(rh_ctx* ctx, const double** bp, size_t* tri_stride,
                          const double** hp, size_t* hp_stride, int* hp_ld);

/* ---- loader inspection (host only, no GPU, no context) ----
 * What the parameter loader holds after binding its sources, for checks of the binding itself (tests/test_bl_cells.py,
 * tests/test_vienna_par.py); not needed by a caller of the probability path.  Return 0, -1 (bad index / NULL) or
 * RH_ERR_PARAM (-4: the files could not be loaded).
 * rh_debug_vienna_cell: energy in the file's 10 cal/mol units of one cell of table 0 = stack[i][j], 1 = int11[i][j][k][l],
 * 2 = int21[i][j][k][l][m], 3 = int22[i][j][k][l][m][n] (pair types 0..7, nucleotides 0..4) of a flat BL* dump or
 * ViennaRNA parameter file (the conventions of src/boltzmann_param.c:5908-5971).
 * rh_debug_vienna_value: one entry (log Boltzmann weight) of the model built from (defaults_file, use_bl_param, bl_path,
 * param_file, semantics) in the install order of src/ractip.cpp:1563-1567; `table` codes 10..27 are listed next to the
 * function in ractip_amd/csrc/vienna_loader.cpp. */
int rh_debug_vienna_cell(const char* param_file, int table, int i, int j, int k, int l, int m, int n, double* energy);
int rh_debug_vienna_value(const char* defaults_file, int use_bl_param, const char* bl_path, const char* param_file, int semantics,
                          int table, int i, int j, int k, int l, double* out);

/* ---- source-compatible pf_duplex surface (src/pf_duplex.h:25-28) ----
 * double pf_duplex(const char*, const char*); extern double** pr_duplex; void free_pf_duplex();
 * are provided by libractip_pfduplex (ractip_amd/csrc/pf_duplex_shim.cpp) on top of rh_duplex. */

#ifdef __cplusplus
}
#endif
#endif /* RACTIP_HOT_H */
