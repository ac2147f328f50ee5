/* oracle/cf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's CONTRAfold-model hot path
 * (SURVEY.md section 8 rows a4-a8, a10, a11 and the width-1 `up` of a1).
 * It is the checker for the HIP path: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library
 * (ractip_amd/libractip_hot.so) never links, loads or calls anything here.
 *
 * Parity status: PINNED -- checked against the reference's own engines
 * compiled from /root/reference (oracle/_ref/libref_contrafold.so) on every
 * bundled data/X.fa sequence and random sequences (tests/test_oracle.py), and
 * against the committed golden fixtures generated from them (tests/golden/).
 */
#ifndef CF_ORACLE_H
#define CF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define CFO_NEG_INF (-2e20) /* /root/reference/src/contrafold/LogSpace.hpp:12 */

typedef struct cfo_model cfo_model;

/* Load a CONTRAfold "name value" parameter file (708 logical parameters) and
 * bind it to the physical score tables with the reference's tying rules
 * (InferenceEngine.ipp:419-938) and caches (InferenceEngine.ipp:1106-1197).
 * Returns NULL on error. */
cfo_model* cfo_load_params(const char* path);
void cfo_free_model(cfo_model* m);

/* T(n) = (n+1)(n+2)/2, offset[i] = i*(2(n+1)-i-1)/2  (InferenceEngine.ipp:316) */
long cfo_tri_size(int n);
long cfo_tri_offset(int n, int i);

/* McCaskill inside (ipp:3356-3722), outside (ipp:3731-4080) and posterior
 * (ipp:4498-4828) for one sequence, double precision, same loop order, same
 * sequential Fast_LogPlusEquals semantics (LogSpace.hpp:232-237).
 *   post   : T(n) doubles, reference layout (posterior[offset[i]+j], letters i<j), or NULL
 *   tables : 6*T(n) doubles FCi,FMi,FM1i,FCo,FMo,FM1o, or NULL
 *   f5     : 2*(n+1) doubles F5i,F5o, or NULL
 * Returns logZ = F5i[n]. */
double cfo_inference(const cfo_model* m, const char* seq, int n,
                     double* post, double* tables, double* f5);

/* Duplex inside/outside/posterior (DuplexEngine.ipp:1015-1169).  Arrays are
 * (n1+1)*(n2+1) row-major or NULL; logz2[0]=inside logZ, logz2[1]=outside logZ. */
void cfo_duplex(const cfo_model* m, const char* s1, int n1, const char* s2, int n2,
                double* post, double* inside, double* outside, double* logz2);

/* RactIP::contrafold's width-1 accessibility, /root/reference/src/ractip.cpp:213-222:
 * float accumulation in the reference's order over a float-narrowed bp. */
void cfo_up_float(int n, const float* bp_tri, float* up);

/* Algorithmic-bytes instrumentation: when enabled, every
 * DP-table element load/store of the recurrences increments a counter. */
void cfo_count_enable(int on);
void cfo_count_reset(void);
unsigned long long cfo_count_loads(void);
unsigned long long cfo_count_stores(void);

#ifdef __cplusplus
}
#endif
#endif
