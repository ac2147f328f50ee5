/* oracle/vienna_oracle.c -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement of the duplex partition function of /root/reference/src/pf_duplex.c with the BL* energy
 * tables of /root/reference/src/boltzmann_param.c, in the semantics of the file's ViennaRNA-1.8 branch
 * (pf_duplex.c:209-433: P->DuplexInit, P->dangle5/dangle3, P->TerminalAU and LoopEnergy()).
 *
 * Why "unpinned": the loop energies live in ViennaRNA (RNAlib2), a third-party dependency that is NOT vendored
 * and NOT installed (CMakeLists.txt:25; README ">= 2.2.0", no pinned version), and the reference has no test
 * or golden output for this path.  LoopEnergy() below restates the published ViennaRNA-1.8 function
 * (stack / bulge / int11 / int21 / int22 / generic interior loop with Ninio asymmetry and mismatchI); DuplexInit
 * = 410 and rtype = {0,2,1,4,3,6,5,7} are the 1.8 constants.  The branch the reference's CMake actually builds
 * (HAVE_VIENNA20, pf_duplex.c:42-207) additionally needs ViennaRNA-2.x-only tables (mismatchExt, mismatch1nI,
 * mismatch23I) that the BL* file does not provide, so it cannot be restated from the repository at all.
 * What IS checked: the DP against brute-force enumeration of all duplexes under the same energy function,
 * forward logZ == backward logZ, and the GPU kernels against this file.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXLOOP 30
#define NBP 7
#define VINF 1000000

typedef struct vo_model {
    int stack[NBP + 1][NBP + 1];
    int mismatchI[NBP + 1][5][5];
    int dangle5[NBP + 1][5], dangle3[NBP + 1][5];
    int int11[NBP + 1][NBP + 1][5][5];
    int int21[NBP + 1][NBP + 1][5][5][5];
    int int22[NBP + 1][NBP + 1][5][5][5][5];
    int bulge[31], internal_loop[31];
    int TerminalAU, ninio, max_ninio, DuplexInit;
    double kT; /* cal/mol */
} vo_model;

static const int RTYPE[8] = {0, 2, 1, 4, 3, 6, 5, 7};

static int read_table(FILE* f, const char* want, int* dst, int count)
{
    char name[64];
    int n;
    rewind(f);
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        if (sscanf(line, "%63s %d", name, &n) == 2 && strcmp(name, want) == 0 && !(name[0] == '-' || (name[0] >= '0' && name[0] <= '9'))) {
            if (n != count) return -1;
            for (int k = 0; k < n; k++)
                if (fscanf(f, "%d", &dst[k]) != 1) return -1;
            return 0;
        }
    }
    return -1;
}

vo_model* vo_load(const char* path)
{
    FILE* f = fopen(path, "r");
    if (!f) return NULL;
    vo_model* m = (vo_model*)calloc(1, sizeof(vo_model));
    int* buf = (int*)malloc(sizeof(int) * 13000);
    int bad = 0, p;
    /* index conventions of the copy_* loops, boltzmann_param.c:5908-5971 */
    bad |= read_table(f, "stack37", buf, 49); p = 0;
    for (int i = 1; i <= NBP; i++) for (int j = 1; j <= NBP; j++) m->stack[i][j] = buf[p++];
    bad |= read_table(f, "mismatchI37", buf, 175); p = 0;
    for (int i = 1; i <= NBP; i++) for (int j = 0; j < 5; j++) for (int k = 0; k < 5; k++) m->mismatchI[i][j][k] = buf[p++];
    bad |= read_table(f, "dangle5_37", buf, 40); p = 0;
    for (int i = 0; i <= NBP; i++) for (int j = 0; j < 5; j++) m->dangle5[i][j] = buf[p++];
    bad |= read_table(f, "dangle3_37", buf, 40); p = 0;
    for (int i = 0; i <= NBP; i++) for (int j = 0; j < 5; j++) m->dangle3[i][j] = buf[p++];
    bad |= read_table(f, "int11_37", buf, 1225); p = 0;
    for (int i = 1; i <= NBP; i++) for (int j = 1; j <= NBP; j++) for (int k = 0; k < 5; k++) for (int l = 0; l < 5; l++)
        m->int11[i][j][k][l] = buf[p++];
    bad |= read_table(f, "int21_37", buf, 6125); p = 0;
    for (int i = 1; i <= NBP; i++) for (int j = 1; j <= NBP; j++) for (int k = 0; k < 5; k++) for (int l = 0; l < 5; l++)
        for (int q = 0; q < 5; q++) m->int21[i][j][k][l][q] = buf[p++];
    bad |= read_table(f, "int22_37", buf, 12544); p = 0;
    for (int i = 1; i <= NBP; i++) for (int j = 1; j <= NBP; j++) for (int k = 1; k < 5; k++) for (int l = 1; l < 5; l++)
        for (int q = 1; q < 5; q++) for (int r = 1; r < 5; r++) m->int22[i][j][k][l][q][r] = buf[p++];
    bad |= read_table(f, "bulge37", m->bulge, 31);
    bad |= read_table(f, "internal_loop37", m->internal_loop, 31);
    bad |= read_table(f, "MLparams", buf, 4);
    m->TerminalAU = buf[3];
    bad |= read_table(f, "ninio", buf, 2);
    m->ninio = buf[0]; m->max_ninio = buf[1];
    free(buf);
    fclose(f);
    if (bad) { free(m); return NULL; }
    /* scale_parameters() at 37 C leaves the *37 tables unchanged except that dangles are clipped to <= 0 */
    for (int i = 0; i <= NBP; i++) for (int j = 0; j < 5; j++) {
        if (m->dangle5[i][j] > 0) m->dangle5[i][j] = 0;
        if (m->dangle3[i][j] > 0) m->dangle3[i][j] = 0;
    }
    m->DuplexInit = 410;
    m->kT = (37.0 + 273.15) * 1.98717; /* (temperature+K0)*GASCONST, pf_duplex.c:73, ractip.cpp:262 */
    return m;
}
void vo_free(vo_model* m) { free(m); }

/* encode_char: A,C,G,U -> 1..4 (T as U), anything else 0 */
static int vcode(char c)
{
    switch (c) {
        case 'A': case 'a': return 1;
        case 'C': case 'c': return 2;
        case 'G': case 'g': return 3;
        case 'U': case 'u': case 'T': case 't': return 4;
        default: return 0;
    }
}
/* pair types CG=1 GC=2 GU=3 UG=4 AU=5 UA=6 (boltzmann_param.c:22) */
static int ptype(int a, int b)
{
    static const int T[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, 1, 0}, {0, 0, 2, 0, 3}, {0, 6, 0, 4, 0}};
    return T[a][b];
}

/* ViennaRNA-1.8 LoopEnergy(n1, n2, type, type_2, si1, sj1, sp1, sq1) */
static int loop_energy(const vo_model* P, int n1, int n2, int type, int type_2, int si1, int sj1, int sp1, int sq1)
{
    int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1, energy;
    if (nl == 0) return P->stack[type][type_2];
    if (ns == 0) {
        energy = P->bulge[nl];
        if (nl == 1) energy += P->stack[type][type_2];
        else {
            if (type > 2) energy += P->TerminalAU;
            if (type_2 > 2) energy += P->TerminalAU;
        }
        return energy;
    }
    if (ns == 1) {
        if (nl == 1) return P->int11[type][type_2][si1][sj1];
        if (nl == 2) return n1 == 1 ? P->int21[type][type_2][si1][sq1][sj1] : P->int21[type_2][type][sq1][si1][sp1];
    } else if (n1 == 2 && n2 == 2) return P->int22[type][type_2][si1][sp1][sq1][sj1];
    energy = P->internal_loop[n1 + n2];
    int asym = (nl - ns) * P->ninio;
    energy += asym < P->max_ninio ? asym : P->max_ninio;
    energy += P->mismatchI[type][si1][sj1] + P->mismatchI[type_2][sq1][sp1];
    return energy;
}

static double logadd(double x, double y) /* pf_duplex.c:34-40 */
{
    if (x <= -INFINITY) return y;
    if (y <= -INFINITY) return x;
    return x > y ? log1p(exp(y - x)) + x : log1p(exp(x - y)) + y;
}

static int e_open(const vo_model* P, const int* S1, const int* S2, int n2, int i, int j, int type)
{ /* pf_duplex.c:321-325 */
    int E = P->DuplexInit;
    if (i > 1) E += P->dangle5[type][S1[i - 1]];
    if (j < n2) E += P->dangle3[type][S2[j + 1]];
    if (type > 2) E += P->TerminalAU;
    return E;
}
static int e_close(const vo_model* P, const int* S1, const int* S2, int n1, int i, int j, int type)
{ /* pf_duplex.c:337-340 */
    int E = 0;
    if (i < n1) E += P->dangle3[RTYPE[type]][S1[i + 1]];
    if (j > 1) E += P->dangle5[RTYPE[type]][S2[j - 1]];
    if (type > 2) E += P->TerminalAU;
    return E;
}

/* pf_duplex (pf_duplex.c:241-291), fw (:304-345), bk (:347-394).  Arrays (n1+1)*(n2+1) row-major or NULL.
 * Returns the forward log Z; *esum_bk receives the backward one. */
double vo_pf_duplex(const vo_model* P, const char* s1, int n1, const char* s2, int n2,
                    double* pr, double* fw_out, double* bk_out, double* esum_bk)
{
    const int W = n2 + 1;
    int* S1 = (int*)calloc(n1 + 2, sizeof(int));
    int* S2 = (int*)calloc(n2 + 2, sizeof(int));
    for (int i = 1; i <= n1; i++) S1[i] = vcode(s1[i - 1]);
    for (int j = 1; j <= n2; j++) S2[j] = vcode(s2[j - 1]);
    double* fw = (double*)malloc(sizeof(double) * 2 * (size_t)(n1 + 1) * W);
    double* bk = fw + (size_t)(n1 + 1) * W;
    for (size_t k = 0; k < 2 * (size_t)(n1 + 1) * W; k++) fw[k] = -INFINITY;
    const double sc = 10.0 / P->kT;
    double Esum = -INFINITY;
    for (int i = 1; i <= n1; i++)
        for (int j = n2; j > 0; j--) {
            const int type = ptype(S1[i], S2[j]);
            if (!type) continue;
            double v = -e_open(P, S1, S2, n2, i, j, type) * sc;
            for (int k = i - 1; k > 0 && k > i - MAXLOOP - 2; k--)
                for (int l = j + 1; l <= n2; l++) {
                    if (i - k + l - j - 2 > MAXLOOP) break;
                    const int type2 = ptype(S1[k], S2[l]);
                    if (!type2) continue;
                    const int E = loop_energy(P, i - k - 1, l - j - 1, type2, RTYPE[type], S1[k + 1], S2[l - 1], S1[i - 1], S2[j + 1]);
                    v = logadd(v, fw[(size_t)k * W + l] - E * sc);
                }
            fw[(size_t)i * W + j] = v;
            Esum = logadd(Esum, v - e_close(P, S1, S2, n1, i, j, type) * sc);
        }
    double Ebk = -INFINITY;
    for (int i = n1; i > 0; i--)
        for (int j = 1; j <= n2; j++) {
            const int type = ptype(S1[i], S2[j]);
            if (!type) continue;
            double* cur = &bk[(size_t)i * W + j];
            *cur = logadd(*cur, -e_close(P, S1, S2, n1, i, j, type) * sc);
            for (int k = i - 1; k > 0 && k > i - MAXLOOP - 2; k--)
                for (int l = j + 1; l <= n2; l++) {
                    if (i - k + l - j - 2 > MAXLOOP) break;
                    const int type2 = ptype(S1[k], S2[l]);
                    if (!type2) continue;
                    const int E = loop_energy(P, i - k - 1, l - j - 1, type2, RTYPE[type], S1[k + 1], S2[l - 1], S1[i - 1], S2[j + 1]);
                    bk[(size_t)k * W + l] = logadd(bk[(size_t)k * W + l], *cur - E * sc);
                }
            Ebk = logadd(Ebk, *cur - e_open(P, S1, S2, n2, i, j, type) * sc);
        }
    if (pr)
        for (int i = 0; i <= n1; i++)
            for (int j = 0; j <= n2; j++) {
                const double a = fw[(size_t)i * W + j], b = bk[(size_t)i * W + j];
                pr[(size_t)i * W + j] = (i && j && a > -INFINITY && b > -INFINITY) ? exp(a + b - Esum) : 0.0;
            }
    if (fw_out) memcpy(fw_out, fw, sizeof(double) * (size_t)(n1 + 1) * W);
    if (bk_out) memcpy(bk_out, bk, sizeof(double) * (size_t)(n1 + 1) * W);
    if (esum_bk) *esum_bk = Ebk;
    free(fw); free(S1); free(S2);
    return Esum;
}

/* ---- brute force: enumerate every duplex (chain of pairs, i increasing, j decreasing, loops within the
 *      DP's budget) and sum Boltzmann weights; independent check of the recurrences above for tiny inputs. */
typedef struct { const vo_model* P; const int *S1, *S2; int n1, n2; double sc, Z; double* marg; int pi[64], pj[64]; } bf_t;

static void bf_extend(bf_t* b, int depth, double logw)
{
    const int i = b->pi[depth - 1], j = b->pj[depth - 1];
    const int type = ptype(b->S1[i], b->S2[j]);
    /* stop here */
    {
        const double w = exp(logw - e_close(b->P, b->S1, b->S2, b->n1, i, j, type) * b->sc);
        b->Z += w;
        for (int d = 0; d < depth; d++) b->marg[(size_t)b->pi[d] * (b->n2 + 1) + b->pj[d]] += w;
    }
    for (int i2 = i + 1; i2 <= b->n1 && i2 < i + MAXLOOP + 2; i2++)
        for (int j2 = j - 1; j2 >= 1; j2--) {
            if ((i2 - i - 1) + (j - j2 - 1) > MAXLOOP) break;
            const int t2 = ptype(b->S1[i2], b->S2[j2]);
            if (!t2) continue;
            /* the DP extends from (k,l)=(i,j) to (i2,j2): LoopEnergy(i2-i-1, j-j2-1, type(k,l), rtype[type(i2,j2)], ...) */
            const int E = loop_energy(b->P, i2 - i - 1, j - j2 - 1, type, RTYPE[t2], b->S1[i + 1], b->S2[j - 1], b->S1[i2 - 1], b->S2[j2 + 1]);
            b->pi[depth] = i2; b->pj[depth] = j2;
            bf_extend(b, depth + 1, logw - E * b->sc);
        }
}

double vo_bruteforce(const vo_model* P, const char* s1, int n1, const char* s2, int n2, double* pr)
{
    int* S1 = (int*)calloc(n1 + 2, sizeof(int));
    int* S2 = (int*)calloc(n2 + 2, sizeof(int));
    for (int i = 1; i <= n1; i++) S1[i] = vcode(s1[i - 1]);
    for (int j = 1; j <= n2; j++) S2[j] = vcode(s2[j - 1]);
    bf_t b;
    b.P = P; b.S1 = S1; b.S2 = S2; b.n1 = n1; b.n2 = n2; b.sc = 10.0 / P->kT; b.Z = 0.0;
    b.marg = (double*)calloc((size_t)(n1 + 1) * (n2 + 1), sizeof(double));
    for (int i = 1; i <= n1; i++)
        for (int j = n2; j >= 1; j--) {
            const int type = ptype(S1[i], S2[j]);
            if (!type) continue;
            b.pi[0] = i; b.pj[0] = j;
            bf_extend(&b, 1, -e_open(P, S1, S2, n2, i, j, type) * b.sc);
        }
    if (pr)
        for (size_t k = 0; k < (size_t)(n1 + 1) * (n2 + 1); k++) pr[k] = b.Z > 0 ? b.marg[k] / b.Z : 0.0;
    const double z = b.Z > 0 ? log(b.Z) : -INFINITY;
    free(b.marg); free(S1); free(S2);
    return z;
}
