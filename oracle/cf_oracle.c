/* oracle/cf_oracle.c -- TEST INFRASTRUCTURE ONLY (see cf_oracle.h).
 *
 * CPU restatement, in plain C, of the reference's CONTRAfold-model partition
 * function DPs.  Written from the recurrences, not copied: every function cites
 * the reference lines it follows (paths relative to /root/reference/src/contrafold).
 *
 * Conventions kept from the reference so tables can be compared element-wise:
 *   - gap indexing 0..L, triangular storage offset[i]+j          (InferenceEngine.ipp:316)
 *   - nucleotide codes A,C,G,U -> 0..3, anything else -> 4        (ipp:379-384, 1034-1039)
 *   - -inf sentinel NEG_INF=-2e20, "zero" test y > NEG_INF/2      (LogSpace.hpp:12,232-237)
 *   - sequential log-add with the x-y<30 cut-off                  (LogSpace.hpp:232-237)
 */
#include "cf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NEG CFO_NEG_INF
#define MAXSINGLE 30 /* Config.hpp:213 C_MAX_SINGLE_LENGTH */
#define MINHAIRPIN 3 /* Config.hpp:212 C_MIN_HAIRPIN_LENGTH */

struct cfo_model {
    double base_pair[5][5];
    double terminal_mismatch[5][5][5][5];
    double hairpin_len[31];     /* prefix-summed cache, ipp:1120-1124 */
    double single_len[31][31];  /* cache_score_single, ipp:1161-1197 */
    double bulge_0x1[5], bulge_1x0[5];
    double internal_1x1[5][5];
    double helix_stacking[5][5][5][5];
    double helix_closing[5][5];
    double multi_base, multi_unpaired, multi_paired;
    double dangle_left[5][5][5], dangle_right[5][5][5];
    double external_unpaired, external_paired;
};

/* ---------------------------------------------------------------- counters */
static int g_count_on = 0;
static unsigned long long g_loads = 0, g_stores = 0;
void cfo_count_enable(int on) { g_count_on = on; }
void cfo_count_reset(void) { g_loads = g_stores = 0; }
unsigned long long cfo_count_loads(void) { return g_loads; }
unsigned long long cfo_count_stores(void) { return g_stores; }
#define LD(n) do { if (g_count_on) g_loads += (n); } while (0)
#define ST(n) do { if (g_count_on) g_stores += (n); } while (0)

/* ------------------------------------------------------------ log-space add */
/* LogSpace.hpp:232-237 (double): x <- log(exp(x)+exp(y)), skipping negligible terms */
static inline void lpe(double* x, double y)
{
    double hi = *x, lo = y;
    if (hi < lo) { double t = hi; hi = lo; lo = t; }
    if (lo > NEG / 2 && hi - lo < 30.0)
        hi = log(exp(hi - lo) + 1.0) + lo;
    *x = hi;
}
/* LogSpace.hpp:218-223 */
static inline double ladd(double x, double y) { lpe(&x, y); return x; }
/* LogSpace.hpp:21-25 */
static inline double fexp(double x) { return x <= NEG / 2 ? 0.0 : exp(x); }

/* ------------------------------------------------------- parameter loading */
typedef struct { char name[64]; double v; } kv_t;
typedef struct { kv_t* kv; int n; } kvtab_t;

static double kv_get(const kvtab_t* t, const char* name, int* missing)
{
    for (int i = 0; i < t->n; i++)
        if (strcmp(t->kv[i].name, name) == 0) return t->kv[i].v;
    (*missing)++;
    return 0.0;
}

static const char ALPHA[] = "ACGU";

cfo_model* cfo_load_params(const char* path)
{
    FILE* f = fopen(path, "r");
    if (!f) return NULL;
    kvtab_t t; t.n = 0; t.kv = (kv_t*)malloc(sizeof(kv_t) * 4096);
    while (t.n < 4096 && fscanf(f, "%63s %lf", t.kv[t.n].name, &t.kv[t.n].v) == 2) t.n++;
    fclose(f);
    cfo_model* m = (cfo_model*)calloc(1, sizeof(cfo_model));
    int miss = 0;
    char a[80], b[80];
    /* base_pair_XY tied with YX under the lexicographically smaller name (ipp:446-451) */
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        snprintf(a, sizeof a, "base_pair_%c%c", ALPHA[i], ALPHA[j]);
        snprintf(b, sizeof b, "base_pair_%c%c", ALPHA[j], ALPHA[i]);
        m->base_pair[i][j] = kv_get(&t, strcmp(a, b) < 0 ? a : b, &miss);
    }
    /* terminal_mismatch_WXYZ, untied (ipp:486-487) */
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
    for (int k = 0; k < 4; k++) for (int l = 0; l < 4; l++) {
        snprintf(a, sizeof a, "terminal_mismatch_%c%c%c%c", ALPHA[i], ALPHA[j], ALPHA[k], ALPHA[l]);
        m->terminal_mismatch[i][j][k][l] = kv_get(&t, a, &miss);
    }
    /* hairpin_length_at_least_k, prefix sum (ipp:1120-1124) */
    {
        double acc = 0;
        for (int k = 0; k <= 30; k++) {
            snprintf(a, sizeof a, "hairpin_length_at_least_%d", k);
            acc += kv_get(&t, a, &miss);
            m->hairpin_len[k] = acc;
        }
    }
    /* single-branch length cache (ipp:1126-1197) */
    {
        double expl[5][5] = {{0}}, bulge[31] = {0}, inter[31] = {0}, sym[16] = {0}, asym[29] = {0};
        for (int i = 1; i <= 4; i++) for (int j = 1; j <= 4; j++) {
            snprintf(a, sizeof a, "internal_explicit_%d_%d", i < j ? i : j, i < j ? j : i);
            expl[i][j] = kv_get(&t, a, &miss);
        }
        for (int k = 1; k <= 30; k++) { snprintf(a, sizeof a, "bulge_length_at_least_%d", k); bulge[k] = bulge[k - 1] + kv_get(&t, a, &miss); }
        for (int k = 2; k <= 30; k++) { snprintf(a, sizeof a, "internal_length_at_least_%d", k); inter[k] = inter[k - 1] + kv_get(&t, a, &miss); }
        for (int k = 1; k <= 15; k++) { snprintf(a, sizeof a, "internal_symmetric_length_at_least_%d", k); sym[k] = sym[k - 1] + kv_get(&t, a, &miss); }
        for (int k = 1; k <= 28; k++) { snprintf(a, sizeof a, "internal_asymmetry_at_least_%d", k); asym[k] = asym[k - 1] + kv_get(&t, a, &miss); }
        for (int l1 = 0; l1 <= 30; l1++) for (int l2 = 0; l1 + l2 <= 30; l2++) {
            double v = 0;
            if (l1 == 0 && l2 == 0) { m->single_len[l1][l2] = 0; continue; }
            if (l1 == 0 || l2 == 0) {
                v += bulge[l1 + l2 < 30 ? l1 + l2 : 30];
            } else {
                if (l1 <= 4 && l2 <= 4) v += expl[l1][l2];
                v += inter[l1 + l2 < 30 ? l1 + l2 : 30];
                if (l1 == l2) v += sym[l1 < 15 ? l1 : 15];
                int d = l1 > l2 ? l1 - l2 : l2 - l1;
                v += asym[d < 28 ? d : 28];
            }
            m->single_len[l1][l2] = v;
        }
    }
    /* bulge_0x1_nucleotides_X feeds both the 0x1 and the 1x0 table (ipp:688-690) */
    for (int i = 0; i < 4; i++) {
        snprintf(a, sizeof a, "bulge_0x1_nucleotides_%c", ALPHA[i]);
        m->bulge_0x1[i] = m->bulge_1x0[i] = kv_get(&t, a, &miss);
    }
    /* internal_1x1_nucleotides_XY symmetric (ipp:758-763) */
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        snprintf(a, sizeof a, "internal_1x1_nucleotides_%c%c", ALPHA[i], ALPHA[j]);
        snprintf(b, sizeof b, "internal_1x1_nucleotides_%c%c", ALPHA[j], ALPHA[i]);
        m->internal_1x1[i][j] = kv_get(&t, strcmp(a, b) < 0 ? a : b, &miss);
    }
    /* helix_stacking_WXYZ tied with ZYXW (ipp:844-849) */
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
    for (int k = 0; k < 4; k++) for (int l = 0; l < 4; l++) {
        snprintf(a, sizeof a, "helix_stacking_%c%c%c%c", ALPHA[i], ALPHA[j], ALPHA[k], ALPHA[l]);
        snprintf(b, sizeof b, "helix_stacking_%c%c%c%c", ALPHA[l], ALPHA[k], ALPHA[j], ALPHA[i]);
        m->helix_stacking[i][j][k][l] = kv_get(&t, strcmp(a, b) < 0 ? a : b, &miss);
    }
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        snprintf(a, sizeof a, "helix_closing_%c%c", ALPHA[i], ALPHA[j]);
        m->helix_closing[i][j] = kv_get(&t, a, &miss);
    }
    m->multi_base = kv_get(&t, "multi_base", &miss);
    m->multi_unpaired = kv_get(&t, "multi_unpaired", &miss);
    m->multi_paired = kv_get(&t, "multi_paired", &miss);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) {
        snprintf(a, sizeof a, "dangle_left_%c%c%c", ALPHA[i], ALPHA[j], ALPHA[k]);
        m->dangle_left[i][j][k] = kv_get(&t, a, &miss);
        snprintf(a, sizeof a, "dangle_right_%c%c%c", ALPHA[i], ALPHA[j], ALPHA[k]);
        m->dangle_right[i][j][k] = kv_get(&t, a, &miss);
    }
    m->external_unpaired = kv_get(&t, "external_unpaired", &miss);
    m->external_paired = kv_get(&t, "external_paired", &miss);
    free(t.kv);
    if (miss) { fprintf(stderr, "cf_oracle: %d parameters missing in %s\n", miss, path); free(m); return NULL; }
    return m;
}

void cfo_free_model(cfo_model* m) { free(m); }

long cfo_tri_size(int n) { return (long)(n + 1) * (n + 2) / 2; }
long cfo_tri_offset(int n, int i) { return (long)i * (2 * (n + 1) - i - 1) / 2; }

/* ----------------------------------------------------------------- encoding */
static int nuc_code(char c)
{
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'U': case 'u': return 3;
        default: return 4;
    }
}
static int complementary(int a, int b)
{ /* AU, GU, CG both ways (ipp:391-396) */
    return (a == 0 && b == 3) || (a == 3 && b == 0) || (a == 2 && b == 3) ||
           (a == 3 && b == 2) || (a == 1 && b == 2) || (a == 2 && b == 1);
}

/* ------------------------------------------------------------ scoring terms */
typedef struct {
    const cfo_model* m;
    const int* s; /* s[0]=4 (sentinel), s[1..L] */
    int L;
    const long* off;
    const unsigned char* pairable; /* T(n): letters (i,j) may pair */
} ctx_t;

/* ScoreJunctionA, ipp:1927-1956 */
static double junction_a(const ctx_t* c, int i, int j)
{
    const cfo_model* m = c->m; const int* s = c->s;
    double v = m->helix_closing[s[i]][s[j + 1]];
    if (i < c->L) v += m->dangle_left[s[i]][s[j + 1]][s[i + 1]];
    if (j > 0) v += m->dangle_right[s[i]][s[j + 1]][s[j]];
    return v;
}
/* ScoreJunctionB, ipp:2004-2029 */
static double junction_b(const ctx_t* c, int i, int j)
{
    const cfo_model* m = c->m; const int* s = c->s;
    return m->helix_closing[s[i]][s[j + 1]] + m->terminal_mismatch[s[i]][s[j + 1]][s[i + 1]][s[j]];
}
/* ScoreBasePair, ipp:2060-2083 */
static double base_pair(const ctx_t* c, int i, int j) { return c->m->base_pair[c->s[i]][c->s[j]]; }
/* ScoreHelixStacking, ipp:219-230 */
static double helix_stack(const ctx_t* c, int i, int j)
{ return c->m->helix_stacking[c->s[i]][c->s[j]][c->s[i + 1]][c->s[j - 1]]; }
/* ScoreHairpin, ipp:2123-2152 */
static double hairpin(const ctx_t* c, int i, int j)
{ return junction_b(c, i, j) + c->m->hairpin_len[j - i < 30 ? j - i : 30]; }
/* ScoreSingleNucleotides, ipp:2290-2360 (active groups: bulge 0x1/1x0, internal 1x1) */
static double single_nucs(const ctx_t* c, int i, int j, int p, int q)
{
    int l1 = p - i, l2 = j - q;
    if (l1 == 0 && l2 == 1) return c->m->bulge_0x1[c->s[j]];
    if (l1 == 1 && l2 == 0) return c->m->bulge_1x0[c->s[i + 1]];
    if (l1 == 1 && l2 == 1) return c->m->internal_1x1[c->s[i + 1]][c->s[j]];
    return 0.0;
}
/* one single-branch term without the FC operand, ipp:3601-3616 */
static double single_term(const ctx_t* c, int i, int j, int p, int q, double jb_ij)
{
    if (p == i && q == j) return base_pair(c, i + 1, j) + helix_stack(c, i, j + 1);
    return jb_ij + c->m->single_len[p - i][j - q] + base_pair(c, p + 1, q) + junction_b(c, q, p) +
           single_nucs(c, i, j, p, q);
}

/* ------------------------------------------------------------- McCaskill DP */
double cfo_inference(const cfo_model* m, const char* seq, int n,
                     double* post, double* tables, double* f5)
{
    const int L = n;
    const long T = cfo_tri_size(L);
    int* s = (int*)malloc(sizeof(int) * (L + 2));
    long* off = (long*)malloc(sizeof(long) * (L + 2));
    unsigned char* pairable = (unsigned char*)calloc(T, 1);
    double* buf = (double*)malloc(sizeof(double) * (6 * T + 2 * (L + 1)));
    double *FCi = buf, *FMi = buf + T, *FM1i = buf + 2 * T, *FCo = buf + 3 * T, *FMo = buf + 4 * T,
           *FM1o = buf + 5 * T, *F5i = buf + 6 * T, *F5o = buf + 6 * T + (L + 1);
    s[0] = 4;
    for (int i = 1; i <= L; i++) s[i] = nuc_code(seq[i - 1]);
    s[L + 1] = 4;
    for (int i = 0; i <= L; i++) off[i] = cfo_tri_offset(L, i);
    /* allow_paired, ipp:1060-1096 */
    for (int i = 1; i <= L; i++)
        for (int j = i + 1; j <= L; j++) pairable[off[i] + j] = (unsigned char)complementary(s[i], s[j]);
    for (long k = 0; k < 6 * T; k++) buf[k] = NEG;
    for (int k = 0; k <= L; k++) F5i[k] = F5o[k] = NEG;

    ctx_t c = {m, s, L, off, pairable};
    const double mp = m->multi_paired, mb = m->multi_base, mu = m->multi_unpaired;
    const double eu = m->external_unpaired, ep = m->external_paired;

    /* ---- inside, ipp:3376-3717 */
    for (int i = L; i >= 0; i--) {
        for (int j = i; j <= L; j++) {
            double fm2 = NEG; /* ipp:3384-3411 */
            for (int k = i + 1; k < j; k++) { LD(2); lpe(&fm2, FM1i[off[i] + k] + FMi[off[k] + j]); }
            if (0 < i && j < L && pairable[off[i] + j + 1]) { /* FC, ipp:3567-3627 */
                double acc = NEG;
                if (j - i >= MINHAIRPIN) lpe(&acc, hairpin(&c, i, j));
                double jb = junction_b(&c, i, j);
                int pmax = i + MAXSINGLE < j ? i + MAXSINGLE : j;
                for (int p = i; p <= pmax; p++) {
                    int qmin = p + 2 > p - i + j - MAXSINGLE ? p + 2 : p - i + j - MAXSINGLE;
                    for (int q = j; q >= qmin; q--) {
                        if (!pairable[off[p + 1] + q]) continue;
                        LD(1);
                        lpe(&acc, FCi[off[p + 1] + q - 1] + single_term(&c, i, j, p, q, jb));
                    }
                }
                lpe(&acc, fm2 + junction_a(&c, i, j) + mp + mb);
                FCi[off[i] + j] = acc; ST(1);
            }
            if (0 < i && i + 2 <= j && j < L) { /* FM1 ipp:3641-3657, FM ipp:3669-3688 */
                double acc = NEG;
                if (pairable[off[i + 1] + j]) { LD(1); lpe(&acc, FCi[off[i + 1] + j - 1] + junction_a(&c, j, i) + mp + base_pair(&c, i + 1, j)); }
                LD(1); lpe(&acc, FM1i[off[i + 1] + j] + mu);
                FM1i[off[i] + j] = acc; ST(1);
                double accm = NEG;
                lpe(&accm, fm2);
                LD(1); lpe(&accm, FMi[off[i] + j - 1] + mu);
                lpe(&accm, acc);
                FMi[off[i] + j] = accm; ST(1);
            }
        }
    }
    F5i[0] = 0.0; /* ipp:3692-3717 */
    for (int j = 1; j <= L; j++) {
        double acc = NEG;
        lpe(&acc, F5i[j - 1] + eu);
        for (int k = 0; k < j; k++)
            if (pairable[off[k + 1] + j]) { LD(2); lpe(&acc, F5i[k] + FCi[off[k + 1] + j - 1] + ep + base_pair(&c, k + 1, j) + junction_a(&c, j, k)); }
        F5i[j] = acc; ST(1);
    }
    const double Z = F5i[L];

    /* ---- outside (push form, as the reference), ipp:3751-4064 */
    F5o[L] = 0.0;
    for (int j = L; j >= 1; j--) {
        LD(1); ST(1); lpe(&F5o[j - 1], F5o[j] + eu);
        for (int k = 0; k < j; k++) {
            if (!pairable[off[k + 1] + j]) continue;
            double t = F5o[j] + ep + base_pair(&c, k + 1, j) + junction_a(&c, j, k);
            LD(4); ST(2);
            lpe(&F5o[k], t + FCi[off[k + 1] + j - 1]);
            lpe(&FCo[off[k + 1] + j - 1], t + F5i[k]);
        }
    }
    for (int i = 0; i <= L; i++) {
        for (int j = L; j >= i; j--) {
            double fm2o = NEG;
            if (0 < i && i + 2 <= j && j < L) {
                LD(1); lpe(&fm2o, FMo[off[i] + j]);                               /* ipp:3803 */
                LD(1); ST(1); lpe(&FMo[off[i] + j - 1], FMo[off[i] + j] + mu);    /* ipp:3806 */
                LD(1); ST(1); lpe(&FM1o[off[i] + j], FMo[off[i] + j]);            /* ipp:3809 */
                if (pairable[off[i + 1] + j]) {                                   /* ipp:3828 */
                    LD(1); ST(1);
                    lpe(&FCo[off[i + 1] + j - 1], FM1o[off[i] + j] + junction_a(&c, j, i) + mp + base_pair(&c, i + 1, j));
                }
                LD(1); ST(1); lpe(&FM1o[off[i + 1] + j], FM1o[off[i] + j] + mu);  /* ipp:3833 */
            }
            if (0 < i && j < L && pairable[off[i] + j + 1]) { /* ipp:3979-4031 */
                double fco = FCo[off[i] + j]; LD(1);
                double jb = junction_b(&c, i, j);
                int pmax = i + MAXSINGLE < j ? i + MAXSINGLE : j;
                for (int p = i; p <= pmax; p++) {
                    int qmin = p + 2 > p - i + j - MAXSINGLE ? p + 2 : p - i + j - MAXSINGLE;
                    for (int q = j; q >= qmin; q--) {
                        if (!pairable[off[p + 1] + q]) continue;
                        LD(1); ST(1);
                        lpe(&FCo[off[p + 1] + q - 1], fco + single_term(&c, i, j, p, q, jb));
                    }
                }
                lpe(&fm2o, fco + junction_a(&c, i, j) + mp + mb);
            }
            for (int k = i + 1; k < j; k++) { /* ipp:4046-4064 */
                LD(4); ST(2);
                lpe(&FM1o[off[i] + k], fm2o + FMi[off[k] + j]);
                lpe(&FMo[off[k] + j], fm2o + FM1i[off[i] + k]);
            }
        }
    }

    /* ---- posterior, ipp:4498-4828 */
    if (post) {
        for (long k = 0; k < T; k++) post[k] = 0.0;
        for (int i = L; i >= 0; i--) {
            for (int j = i; j <= L; j++) {
                if (0 < i && j < L && pairable[off[i] + j + 1]) {
                    double outside = FCo[off[i] + j] - Z; LD(1);
                    double jb = junction_b(&c, i, j);
                    int pmax = i + MAXSINGLE < j ? i + MAXSINGLE : j;
                    for (int p = i; p <= pmax; p++) {
                        int qmin = p + 2 > p - i + j - MAXSINGLE ? p + 2 : p - i + j - MAXSINGLE;
                        for (int q = j; q >= qmin; q--) {
                            if (!pairable[off[p + 1] + q]) continue;
                            LD(2); ST(1);
                            post[off[p + 1] + q] += fexp(outside + single_term(&c, i, j, p, q, jb) + FCi[off[p + 1] + q - 1]);
                        }
                    }
                }
                if (0 < i && i + 2 <= j && j < L && pairable[off[i + 1] + j]) {
                    LD(3); ST(1);
                    post[off[i + 1] + j] += fexp(FM1o[off[i] + j] + FCi[off[i + 1] + j - 1] + junction_a(&c, j, i) + mp + base_pair(&c, i + 1, j) - Z);
                }
            }
        }
        for (int j = 1; j <= L; j++) {
            double outside = F5o[j] - Z;
            for (int k = 0; k < j; k++)
                if (pairable[off[k + 1] + j]) {
                    LD(3); ST(1);
                    post[off[k + 1] + j] += fexp(outside + F5i[k] + FCi[off[k + 1] + j - 1] + ep + base_pair(&c, k + 1, j) + junction_a(&c, j, k));
                }
        }
        for (int i = 1; i <= L; i++)
            for (int j = i + 1; j <= L; j++) {
                double v = post[off[i] + j];
                post[off[i] + j] = v < 0 ? 0 : (v > 1 ? 1 : v);
            }
    }
    if (tables) memcpy(tables, buf, sizeof(double) * 6 * T);
    if (f5) memcpy(f5, F5i, sizeof(double) * 2 * (L + 1));
    free(buf); free(pairable); free(off); free(s);
    return Z;
}

/* ------------------------------------------------------------------- duplex */
/* LoopScore, DuplexEngine.ipp:974-1012 with (i,j,p,q) = (p,q,i,j) of the caller:
 * upstream pair (p,q), downstream pair (i,j), l1=i-p-1, l2=q-j-1 */
static double dup_loop_nucs(const cfo_model* m, const int* s1, const int* s2, int p, int q, int i, int j)
{
    int l1 = i - p - 1, l2 = q - j - 1;
    if (l1 == 0 && l2 == 1) return m->bulge_0x1[s2[q - 1]];
    if (l1 == 1 && l2 == 0) return m->bulge_1x0[s1[p + 1]];
    if (l1 == 1 && l2 == 1) return m->internal_1x1[s1[p + 1]][s2[q - 1]];
    return 0.0;
}
/* score of extending from pair (p,q) to pair (i,j), DuplexEngine.ipp:1047-1061 */
static double dup_step(const cfo_model* m, const int* s1, const int* s2, int p, int q, int i, int j)
{
    if (i - p - 1 == 0 && q - j - 1 == 0)
        return m->base_pair[s1[i]][s2[j]] + m->helix_stacking[s1[p]][s2[q]][s1[i]][s2[j]];
    return m->terminal_mismatch[s1[p]][s2[q]][s1[p + 1]][s2[q - 1]] +
           m->terminal_mismatch[s2[j]][s1[i]][s2[j + 1]][s1[i - 1]] + m->base_pair[s1[i]][s2[j]] +
           dup_loop_nucs(m, s1, s2, p, q, i, j);
}
/* left end (duplex starts at (i,j)), DuplexEngine.ipp:1029-1035 */
static double dup_open(const cfo_model* m, const int* s1, const int* s2, int L2, int i, int j)
{
    double v = m->external_unpaired * (double)(i - 1 + L2 - j);
    if (i > 1) v += m->dangle_right[s2[j]][s1[i]][s1[i - 1]];
    if (j < L2) v += m->dangle_left[s2[j]][s1[i]][s2[j + 1]];
    return v + m->base_pair[s2[j]][s1[i]] + m->helix_closing[s2[j]][s1[i]];
}
/* right end (duplex stops at (i,j)), DuplexEngine.ipp:1066-1073 */
static double dup_close(const cfo_model* m, const int* s1, const int* s2, int L1, int i, int j)
{
    double v = m->external_unpaired * (double)(L1 - i + j - 1);
    if (i < L1) v += m->dangle_left[s1[i]][s2[j]][s1[i + 1]];
    if (j > 1) v += m->dangle_right[s1[i]][s2[j]][s2[j - 1]];
    return v + m->helix_closing[s1[i]][s2[j]];
}

void cfo_duplex(const cfo_model* m, const char* a, int n1, const char* b, int n2,
                double* post, double* inside_out, double* outside_out, double* logz2)
{
    const int L1 = n1, L2 = n2, W = L2 + 1;
    const long SZ = (long)(L1 + 1) * W;
    int* s1 = (int*)malloc(sizeof(int) * (L1 + 2));
    int* s2 = (int*)malloc(sizeof(int) * (L2 + 2));
    s1[0] = s2[0] = 4; s1[L1 + 1] = s2[L2 + 1] = 4;
    for (int i = 1; i <= L1; i++) s1[i] = nuc_code(a[i - 1]);
    for (int j = 1; j <= L2; j++) s2[j] = nuc_code(b[j - 1]);
    double* in = (double*)malloc(sizeof(double) * 2 * SZ);
    double* out = in + SZ;
    for (long k = 0; k < 2 * SZ; k++) in[k] = NEG;
#define PAIR(i, j) complementary(s1[i], s2[j])
    /* inside, DuplexEngine.ipp:1015-1077 */
    double Zi = NEG;
    for (int i = 1; i <= L1; i++) {
        for (int j = L2; j > 0; j--) {
            if (!PAIR(i, j)) continue;
            double acc = NEG;
            acc = ladd(acc, dup_open(m, s1, s2, L2, i, j));
            int pmin = i - MAXSINGLE > 1 ? i - MAXSINGLE : 1;
            for (int p = i - 1; p >= pmin; p--) {
                int qmax = MAXSINGLE + p - i + j < L2 ? MAXSINGLE + p - i + j : L2;
                for (int q = j + 1; q <= qmax; q++) {
                    if (!PAIR(p, q)) continue;
                    LD(1);
                    acc = ladd(acc, in[(long)p * W + q] + dup_step(m, s1, s2, p, q, i, j));
                }
            }
            in[(long)i * W + j] = acc; ST(1);
            Zi = ladd(Zi, acc + dup_close(m, s1, s2, L1, i, j));
        }
    }
    /* outside (push form), DuplexEngine.ipp:1080-1143 */
    double Zo = NEG;
    for (int i = L1; i > 0; i--) {
        for (int j = 1; j <= L2; j++) {
            if (!PAIR(i, j)) continue;
            LD(1); ST(1);
            out[(long)i * W + j] = ladd(out[(long)i * W + j], dup_close(m, s1, s2, L1, i, j));
            double cur = out[(long)i * W + j];
            int pmin = i - MAXSINGLE > 1 ? i - MAXSINGLE : 1;
            for (int p = i - 1; p >= pmin; p--) {
                int qmax = MAXSINGLE + p - i + j < L2 ? MAXSINGLE + p - i + j : L2;
                for (int q = j + 1; q <= qmax; q++) {
                    if (!PAIR(p, q)) continue;
                    LD(1); ST(1);
                    out[(long)p * W + q] = ladd(out[(long)p * W + q], cur + dup_step(m, s1, s2, p, q, i, j));
                }
            }
            Zo = ladd(Zo, cur + dup_open(m, s1, s2, L2, i, j));
        }
    }
    /* posterior, DuplexEngine.ipp:1146-1169 (uses the inside logZ) */
    if (post) {
        for (long k = 0; k < SZ; k++) post[k] = 0.0;
        for (int i = 1; i <= L1; i++)
            for (int j = L2; j > 0; j--)
                if (PAIR(i, j)) { LD(2); ST(1); post[(long)i * W + j] = fexp(in[(long)i * W + j] + out[(long)i * W + j] - Zi); }
    }
#undef PAIR
    if (inside_out) memcpy(inside_out, in, sizeof(double) * SZ);
    if (outside_out) memcpy(outside_out, out, sizeof(double) * SZ);
    if (logz2) { logz2[0] = Zi; logz2[1] = Zo; }
    free(in); free(s1); free(s2);
}

/* ----------------------------------------------------------- width-1 `up` */
/* /root/reference/src/ractip.cpp:213-222 */
void cfo_up_float(int n, const float* bp, float* up)
{
    for (int i = 0; i < n; i++) {
        float u = 1.0f;
        for (int j = 0; j < i; j++) u -= bp[cfo_tri_offset(n, j + 1) + (i + 1)];
        for (int j = i + 1; j < n; j++) u -= bp[cfo_tri_offset(n, i + 1) + (j + 1)];
        up[i] = u > 0.0f ? u : 0.0f;
    }
}
