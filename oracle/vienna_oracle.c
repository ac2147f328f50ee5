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
    /* McCaskill part (pf_fold semantics) */
    int hairpin[31], mismatchH[NBP + 1][5][5];
    int ML_base, ML_closing, ML_intern;
    int n_tetra; char tetra[64][8]; int tetra_e[64];
    double d5x[NBP + 1][5], d3x[NBP + 1][5]; /* stem dangles (10 cal/mol, smoothed) with the end-of-sequence rule folded into code 0 */
    double lxc;
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

static double vo_smooth(double X)
{
    const double x = X / 10.0;
    if (x < -1.2283697) return 0.0;
    if (x > 0.8660254) return X;
    const double t = sin(x - 0.34242663) + 1.0;
    return 10.0 * 0.38490018 * t * t;
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
    bad |= read_table(f, "hairpin37", m->hairpin, 31);
    bad |= read_table(f, "mismatchH37", buf, 175); p = 0;
    for (int i = 1; i <= NBP; i++) for (int j = 0; j < 5; j++) for (int k = 0; k < 5; k++) m->mismatchH[i][j][k] = buf[p++];
    bad |= read_table(f, "MLparams", buf, 4);
    m->ML_base = buf[0]; m->ML_closing = buf[1]; m->ML_intern = buf[2];   /* copy_MLparams, boltzmann_param.c:5973-5983 */
    m->TerminalAU = buf[3];
    {   /* tetraloops: "SEQ energy" lines after the header */
        rewind(f);
        char line2[256], nm[64]; int cnt;
        while (fgets(line2, sizeof line2, f))
            if (sscanf(line2, "%63s %d", nm, &cnt) == 2 && strcmp(nm, "tetraloops") == 0) {
                m->n_tetra = cnt;
                for (int k = 0; k < cnt && k < 64; k++)
                    if (fscanf(f, "%7s %d", m->tetra[k], &m->tetra_e[k]) != 2) bad = 1;
                break;
            }
    }
    m->lxc = 107.856;  /* lxc37, ViennaRNA constant (not overridden by BL*) */
    bad |= read_table(f, "ninio", buf, 2);
    m->ninio = buf[0]; m->max_ninio = buf[1];
    free(buf);
    fclose(f);
    if (bad) { free(m); return NULL; }
    int raw5[NBP + 1][5], raw3[NBP + 1][5];
    memcpy(raw5, m->dangle5, sizeof raw5); memcpy(raw3, m->dangle3, sizeof raw3);
    /* scale_parameters() at 37 C leaves the *37 tables unchanged except that dangles are clipped to <= 0 */
    for (int i = 0; i <= NBP; i++) for (int j = 0; j < 5; j++) {
        if (m->dangle5[i][j] > 0) m->dangle5[i][j] = 0;
        if (m->dangle3[i][j] > 0) m->dangle3[i][j] = 0;
    }
    /* stems of multi/exterior loops in pf_fold (1.8): dangles on both sides whenever the neighbour exists; TerminalAU is
       folded into the 3' dangle, and at the 3' end of the sequence only TerminalAU remains.  Code 0 = no neighbour. */
    /* part_func.c (1.8) does not clip the dangles: expdangle = exp(SMOOTH(-GT)*10/kT) with the C2-smooth ramp
       SMOOTH(X) = 0 for X/10 < -1.2283697, X for X/10 > 0.8660254, else 10*0.38490018*(sin(X/10-0.34242663)+1)^2 */
    for (int t = 0; t <= NBP; t++) for (int x = 0; x < 5; x++) {
        const double tau = t > 2 ? m->TerminalAU : 0;
        m->d5x[t][x] = x ? -vo_smooth(-(double)raw5[t][x]) : 0.0;
        m->d3x[t][x] = (x ? -vo_smooth(-(double)raw3[t][x]) : 0.0) + tau;
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


/* ===================================================================================================
 * McCaskill partition function with the same tables (pf_fold semantics of ViennaRNA 1.8 part_func.c: dangles on
 * both sides of every multi/exterior stem, TerminalAU folded into the 3' dangle, smoothed dangles, tetraloop
 * bonuses, hairpins > 30 extrapolated with lxc) -- PARITY UNPINNED like the duplex above: it is what
 * /root/reference/src/ractip.cpp:288-304, 351-367 obtains from the third-party pf_fold()/export_bppm(), and
 * what :370-375 obtains from pf_unstru() (P(i..i+w unpaired) = H+I+M+E).  Written with the gap-indexed tables of
 * the CONTRAfold engine (cell (i,j): letters i and j+1 paired) so that the HIP kernels can be compared table by
 * table, but NOT with its multiloop grammar: FM = FM2 | FM[i,j-1]+b | FM1 with FM2 = FM1 x FM
 * (InferenceEngine.ipp:3384-3411, 3669-3688) derives a multiloop part with >= 2 branches and trailing unpaired
 * letters in more than one way, which is what the reference computes for the CONTRAfold model but is not a
 * partition function.  Here trailing unpaired letters belong to the LAST branch only (as qqm does in part_func.c):
 *     FM1[i,j] = leading unpaired + one branch ending at j        FMS[i,j] = FM1[i,j] (+) FMS[i,j-1]+b
 *     FM2[i,j] = (+)_k FM1[i,k] + FM[k,j]                        FM[i,j]  = FM2[i,j] (+) FMS[i,j]
 * Validated independently by brute-force enumeration of all secondary structures.
 * =================================================================================================== */
/* cut = 0: one molecule.  cut = n1 > 0: two molecules s1+s2 concatenated (co_pf_fold, src/ractip.cpp:400-458): the
 * backbone gap between letters cut and cut+1 does not exist.  A loop whose backbone holds that gap is an exterior-like
 * loop (only the stems' own dangle/TerminalAU terms count), dangles are never taken across it. */
typedef struct { const vo_model* P; const int* S; int n; double sc; int cut; const unsigned char* allow; } mc_t;
/* allow: (n+1)x(n+1) bytes, allow[a*(n+1)+b] != 0 iff letters a < b may pair (structure constraints of pf_fold); NULL = no constraint */
#define GAP_OK(c, g) ((c)->cut == 0 || (g) != (c)->cut)   /* letters g and g+1 are neighbours on one strand */

static int tetra_bonus(const vo_model* P, const int* S, int a)
{
    static const char L[] = "_ACGU";
    char key[7];
    for (int k = 0; k < 6; k++) key[k] = L[S[a + k]];
    key[6] = 0;
    for (int k = 0; k < P->n_tetra; k++) if (strcmp(key, P->tetra[k]) == 0) return P->tetra_e[k];
    return 0;
}
/* all energies below in 10 cal/mol; letters a < b paired */
static double e_hairpin(const mc_t* c, int a, int b)
{   /* part_func.c (1.8) expHairpinEnergy: exphairpin[u] (u > 30: hairpin[30] + lxc*log(u/30), not truncated),
       tetraloop bonus, u == 3: TerminalAU only, else mismatchH */
    const vo_model* P = c->P; const int* S = c->S;
    const int u = b - a - 1, type = ptype(S[a], S[b]);
    double E = u <= 30 ? (double)P->hairpin[u] : P->hairpin[30] + P->lxc * log(u / 30.0);
    if (u == 4) E += tetra_bonus(P, S, a);
    if (u == 3) { if (type > 2) E += P->TerminalAU; }
    else E += P->mismatchH[type][S[a + 1]][S[b - 1]];
    return E;
}
static double e_interior(const mc_t* c, int a, int b, int p, int q)
{   /* outer pair (a,b), inner pair (p,q) */
    const int* S = c->S;
    return loop_energy(c->P, p - a - 1, b - q - 1, ptype(S[a], S[b]), RTYPE[ptype(S[p], S[q])], S[a + 1], S[b - 1], S[p - 1], S[q + 1]);
}
static double e_mlclose(const mc_t* c, int a, int b)
{
    const vo_model* P = c->P; const int* S = c->S;
    const int tt = RTYPE[ptype(S[a], S[b])];
    return P->ML_closing + P->ML_intern + P->d3x[tt][S[a + 1]] + P->d5x[tt][S[b - 1]];
}
static double e_stem(const mc_t* c, int p, int q)
{   /* stem (p,q) in a multi or exterior loop, without ML_intern; no dangle from a letter of the other molecule */
    const vo_model* P = c->P; const int* S = c->S;
    const int t = ptype(S[p], S[q]);
    return P->d5x[t][GAP_OK(c, p - 1) ? S[p - 1] : 0] + P->d3x[t][GAP_OK(c, q) ? S[q + 1] : 0];
}
static double e_nickclose(const mc_t* c, int a, int b)
{   /* pair (a,b) closing the loop that holds the nick: an exterior stem seen from inside */
    const vo_model* P = c->P; const int* S = c->S;
    const int tt = RTYPE[ptype(S[a], S[b])];
    return P->d3x[tt][GAP_OK(c, a) ? S[a + 1] : 0] + P->d5x[tt][GAP_OK(c, b - 1) ? S[b - 1] : 0];
}

/* post: T(n) triangular (reference layout) or NULL; up: n*max_w row-major, up[i*max_w+w] = P(letters i+1..i+1+w
 * unpaired) (0 where the region runs off the end) or NULL; tabs: 8*T log-space tables FCi,FMi,FM1i,FCo,FMo,FM1o,
 * FMSi,FMSo + f5: 2*(n+1) (F5i, F5o) or NULL.  Returns log Z (inside); *logz_out = outside log Z. */
/* structure constraints (fold_constrained): the next DP / enumeration call honours this mask (see mc_t::allow); set NULL to clear */
static const unsigned char* vo_allow_mask = NULL;
void vo_set_allow_mask(const unsigned char* m) { vo_allow_mask = m; }

double vo_mccaskill_cut(const vo_model* P, const char* seq, int n, int cut, double* post, double* logz_out,
                        double* up, int max_w, double* tabs, double* f5);
double vo_mccaskill(const vo_model* P, const char* seq, int n, double* post, double* logz_out,
                    double* up, int max_w, double* tabs, double* f5)
{
    return vo_mccaskill_cut(P, seq, n, 0, post, logz_out, up, max_w, tabs, f5);
}
/* cut > 0: co_pf_fold semantics for s1 = seq[0..cut), s2 = seq[cut..n) (accessibility is not defined then: up must be NULL) */
double vo_mccaskill_cut(const vo_model* P, const char* seq, int n, int cut, double* post, double* logz_out,
                        double* up, int max_w, double* tabs, double* f5)
{
    const int L = n;
    const long T = (long)(L + 1) * (L + 2) / 2;
    int* S = (int*)calloc(L + 3, sizeof(int));
    for (int i = 1; i <= L; i++) S[i] = vcode(seq[i - 1]);
    long* off = (long*)malloc(sizeof(long) * (L + 2));
    for (int i = 0; i <= L; i++) off[i] = (long)i * (2 * (L + 1) - i - 1) / 2;
    double* buf = (double*)malloc(sizeof(double) * (8 * T + 6 * (L + 2)));
    double *FCi = buf, *FMi = buf + T, *FM1i = buf + 2 * T, *FCo = buf + 3 * T, *FMo = buf + 4 * T, *FM1o = buf + 5 * T,
           *FMSi = buf + 6 * T, *FMSo = buf + 7 * T, *F5i = buf + 8 * T, *F5o = buf + 8 * T + (L + 1),
           *XP = buf + 8 * T + 2 * (L + 2), *XS = XP + (L + 2), *XPo = XS + (L + 2), *XSo = XPo + (L + 2);
    for (long k = 0; k < 8 * T + 6 * (L + 2); k++) buf[k] = -INFINITY;
    mc_t c = {P, S, L, 10.0 / P->kT, cut, vo_allow_mask};
    if (cut > 0 && up) { free(buf); free(off); free(S); return NAN; }
#define NICKED(i, j) (cut > 0 && (i) <= cut && cut <= (j))   /* the gap lies inside the pair (i, j+1) */
    const double sc = c.sc, mlb = -P->ML_base * sc, mli = -P->ML_intern * sc;
#define PAIR(a, b) ((a) >= 1 && (b) <= L && ptype(S[a], S[b]) && (!c.allow || c.allow[(size_t)(a) * (L + 1) + (b)]))
    /* inside */
    for (int i = L; i >= 0; i--) {
        if (cut > 0 && i == cut) {   /* rows > cut are final: exterior partition function of s2's prefixes cut+1..b */
            XP[cut] = 0.0;
            for (int b = cut + 1; b <= L; b++) {
                double acc = XP[b - 1];
                for (int k = cut; k < b; k++)
                    if (PAIR(k + 1, b)) acc = logadd(acc, XP[k] + FCi[off[k + 1] + b - 1] - e_stem(&c, k + 1, b) * sc);
                XP[b] = acc;
            }
            XS[cut + 1] = 0.0;
        }
        for (int j = i; j <= L; j++) {
            double fm2 = -INFINITY;
            for (int k = i + 1; k < j; k++) fm2 = logadd(fm2, FM1i[off[i] + k] + FMi[off[k] + j]);
            if (0 < i && j < L && j + 1 - i >= 4 && PAIR(i, j + 1)) {
                double acc = -INFINITY;
                if (j - i >= 3 && !NICKED(i, j)) acc = logadd(acc, -e_hairpin(&c, i, j + 1) * sc);
                for (int p = i; p <= (i + MAXLOOP < j ? i + MAXLOOP : j); p++) {
                    if (NICKED(i, p)) break;                       /* 5' side i..p+1 must be one strand */
                    int qmin = p + 2 > p - i + j - MAXLOOP ? p + 2 : p - i + j - MAXLOOP;
                    for (int q = j; q >= qmin; q--) {
                        if (NICKED(q, j)) continue;                /* 3' side q..j+1 */
                        if (!PAIR(p + 1, q)) continue;
                        acc = logadd(acc, FCi[off[p + 1] + q - 1] - e_interior(&c, i, j + 1, p + 1, q) * sc);
                    }
                }
                acc = logadd(acc, fm2 - e_mlclose(&c, i, j + 1) * sc);
                if (NICKED(i, j)) acc = logadd(acc, XS[i + 1] + XP[j] - e_nickclose(&c, i, j + 1) * sc);
                FCi[off[i] + j] = acc;
            }
            if (0 < i && i + 2 <= j && j < L) {
                double acc = -INFINITY;
                if (PAIR(i + 1, j) && GAP_OK(&c, i) && GAP_OK(&c, j))
                    acc = logadd(acc, FCi[off[i + 1] + j - 1] + mli - e_stem(&c, i + 1, j) * sc);
                if (GAP_OK(&c, i) && GAP_OK(&c, i + 1)) acc = logadd(acc, FM1i[off[i + 1] + j] + mlb);
                FM1i[off[i] + j] = acc;
                double fms = acc;
                if (GAP_OK(&c, j - 1) && GAP_OK(&c, j)) fms = logadd(fms, FMSi[off[i] + j - 1] + mlb);
                FMSi[off[i] + j] = fms;
                FMi[off[i] + j] = logadd(fm2, fms);
            }
        }
        if (cut > 0 && i >= 1 && i <= cut) {   /* row i is final: exterior partition function of s1's suffix i..cut */
            double acc = XS[i + 1];
            for (int l = i + 4; l <= cut; l++)
                if (PAIR(i, l)) acc = logadd(acc, FCi[off[i] + l - 1] - e_stem(&c, i, l) * sc + XS[l + 1]);
            XS[i] = acc;
        }
    }
    F5i[0] = 0.0;
    for (int j = 1; j <= L; j++) {
        double acc = F5i[j - 1];
        for (int k = 0; k < j; k++)
            if (PAIR(k + 1, j)) acc = logadd(acc, F5i[k] + FCi[off[k + 1] + j - 1] - e_stem(&c, k + 1, j) * sc);
        F5i[j] = acc;
    }
    const double Z = F5i[L];
    /* outside, push form */
    F5o[L] = 0.0;
    for (int j = L; j >= 1; j--) {
        F5o[j - 1] = logadd(F5o[j - 1], F5o[j]);
        for (int k = 0; k < j; k++) {
            if (!PAIR(k + 1, j)) continue;
            const double t = F5o[j] - e_stem(&c, k + 1, j) * sc;
            F5o[k] = logadd(F5o[k], t + FCi[off[k + 1] + j - 1]);
            FCo[off[k + 1] + j - 1] = logadd(FCo[off[k + 1] + j - 1], t + F5i[k]);
        }
    }
    for (int i = 0; i <= L; i++) {
        if (cut > 0 && i == cut + 1)   /* every pair around the nick has pushed into XPo: unwind the prefix recursion of s2 */
            for (int b = L; b > cut; b--) {
                if (!(XPo[b] > -INFINITY)) continue;
                XPo[b - 1] = logadd(XPo[b - 1], XPo[b]);
                for (int k = cut; k < b; k++) {
                    if (!PAIR(k + 1, b)) continue;
                    const double t = XPo[b] - e_stem(&c, k + 1, b) * sc;
                    XPo[k] = logadd(XPo[k], t + FCi[off[k + 1] + b - 1]);
                    FCo[off[k + 1] + b - 1] = logadd(FCo[off[k + 1] + b - 1], t + XP[k]);
                }
            }
        if (cut > 0 && i >= 1 && i <= cut && XSo[i] > -INFINITY) {   /* XSo[i] is final (pushed by rows < i): unwind the suffix recursion */
            XSo[i + 1] = logadd(XSo[i + 1], XSo[i]);
            for (int l = i + 4; l <= cut; l++) {
                if (!PAIR(i, l)) continue;
                const double t = XSo[i] - e_stem(&c, i, l) * sc;
                XSo[l + 1] = logadd(XSo[l + 1], t + FCi[off[i] + l - 1]);
                FCo[off[i] + l - 1] = logadd(FCo[off[i] + l - 1], t + XS[l + 1]);
            }
        }
        for (int j = L; j >= i; j--) {
            double fm2o = -INFINITY;
            if (0 < i && i + 2 <= j && j < L) {
                fm2o = logadd(fm2o, FMo[off[i] + j]);
                FMSo[off[i] + j] = logadd(FMSo[off[i] + j], FMo[off[i] + j]);
                if (GAP_OK(&c, j - 1) && GAP_OK(&c, j))
                    FMSo[off[i] + j - 1] = logadd(FMSo[off[i] + j - 1], FMSo[off[i] + j] + mlb);
                FM1o[off[i] + j] = logadd(FM1o[off[i] + j], FMSo[off[i] + j]);
                if (PAIR(i + 1, j) && GAP_OK(&c, i) && GAP_OK(&c, j))
                    FCo[off[i + 1] + j - 1] = logadd(FCo[off[i + 1] + j - 1], FM1o[off[i] + j] + mli - e_stem(&c, i + 1, j) * sc);
                if (GAP_OK(&c, i) && GAP_OK(&c, i + 1))
                    FM1o[off[i + 1] + j] = logadd(FM1o[off[i + 1] + j], FM1o[off[i] + j] + mlb);
            }
            if (0 < i && j < L && j + 1 - i >= 4 && PAIR(i, j + 1)) {
                const double fco = FCo[off[i] + j];
                if (NICKED(i, j)) {
                    const double t = fco - e_nickclose(&c, i, j + 1) * sc;
                    XSo[i + 1] = logadd(XSo[i + 1], t + XP[j]);
                    XPo[j] = logadd(XPo[j], t + XS[i + 1]);
                }
                for (int p = i; p <= (i + MAXLOOP < j ? i + MAXLOOP : j); p++) {
                    if (NICKED(i, p)) break;
                    int qmin = p + 2 > p - i + j - MAXLOOP ? p + 2 : p - i + j - MAXLOOP;
                    for (int q = j; q >= qmin; q--) {
                        if (NICKED(q, j)) continue;
                        if (!PAIR(p + 1, q)) continue;
                        FCo[off[p + 1] + q - 1] = logadd(FCo[off[p + 1] + q - 1], fco - e_interior(&c, i, j + 1, p + 1, q) * sc);
                    }
                }
                fm2o = logadd(fm2o, fco - e_mlclose(&c, i, j + 1) * sc);
            }
            for (int k = i + 1; k < j; k++) {
                FM1o[off[i] + k] = logadd(FM1o[off[i] + k], fm2o + FMi[off[k] + j]);
                FMo[off[k] + j] = logadd(FMo[off[k] + j], fm2o + FM1i[off[i] + k]);
            }
        }
    }
    if (post) {
        for (long k = 0; k < T; k++) post[k] = 0.0;
        for (int i = 1; i <= L; i++)
            for (int j = i; j < L; j++)
                if (j + 1 - i >= 4 && PAIR(i, j + 1)) {
                    const double e = FCo[off[i] + j] + FCi[off[i] + j] - Z;
                    post[off[i] + j + 1] = e > -INFINITY ? exp(e) : 0.0;
                }
    }
    if (up && max_w > 0) {
        /* P(letters a..b unpaired) by the loop that holds the run (the H, I, M, E split of pf_unstru):
         *  E: F5i[a-1] * F5o[b]                                   H: sum_{p<a, q>b} FCo(p,q) * hairpin(p,q)
         *  I: sum over loops (p,q;k,l) with the run in the left gap (p<a, b<k) or the right gap (l<a, b<q)
         *  M: leading run of a branch  FM1[a-1,j] -> ... -> FM1[b,j]   : sum_j FM1o[a-1,j] * mlb^len * FM1i[b,j]
         *     trailing run            FMS[i,b]  -> ... -> FMS[i,a-1]  : sum_i FMSo[i,b]   * mlb^len * FMSi[i,a-1]
         * probabilities are summed in linear space. */
        double* Hs = (double*)calloc((size_t)(L + 2) * (L + 2), sizeof(double));   /* Hs[a][b] = sum_{p<a,q>b} */
        double* GL = (double*)calloc((size_t)(L + 2) * (MAXLOOP + 2), sizeof(double)); /* GL[p][l1]: left gap p+1..p+l1 */
        double* GR = (double*)calloc((size_t)(L + 2) * (MAXLOOP + 2), sizeof(double)); /* GR[q][l2]: right gap q-l2..q-1 */
#define HS(a, b) Hs[(size_t)(a) * (L + 2) + (b)]
        for (int p = 1; p <= L; p++)
            for (int q = p + 4; q <= L; q++)
                if (PAIR(p, q)) {
                    const double e = FCo[off[p] + q - 1] - e_hairpin(&c, p, q) * sc - Z;
                    if (e > -INFINITY) HS(p + 1, q - 1) += exp(e);   /* counted at its own corner, swept below */
                    const double fco = FCo[off[p] + q - 1];
                    if (!(fco > -INFINITY)) continue;
                    for (int k = p + 1; k <= p + MAXLOOP + 1 && k < q; k++)
                        for (int l = q - 1; l > k && (k - p - 1) + (q - l - 1) <= MAXLOOP; l--) {
                            if (!PAIR(k, l)) continue;
                            const double t = fco - e_interior(&c, p, q, k, l) * sc + FCi[off[k] + l - 1] - Z;
                            if (!(t > -INFINITY)) continue;
                            GL[(size_t)p * (MAXLOOP + 2) + (k - p - 1)] += exp(t);
                            GR[(size_t)q * (MAXLOOP + 2) + (q - l - 1)] += exp(t);
                        }
                }
        /* dominance sums: Hs[a][b] = sum_{p+1<=a, q-1>=b} h(p+1,q-1) */
        for (int a = 1; a <= L; a++)
            for (int b = L; b >= 1; b--)
                HS(a, b) += HS(a - 1, b) + HS(a, b + 1) - HS(a - 1, b + 1);
        for (int a = 1; a <= L; a++)
            for (int w = 0; w < max_w; w++) {
                const int b = a + w, len = w + 1;
                double pu = 0.0;
                if (b <= L) {
                    pu += exp(F5i[a - 1] + F5o[b] - Z);
                    pu += HS(a, b);
                    for (int p = a - 1; p >= 1 && p >= b - MAXLOOP; p--)       /* left gaps p+1..p+l1 covering a..b */
                        for (int l1 = b - p; l1 <= MAXLOOP; l1++) pu += GL[(size_t)p * (MAXLOOP + 2) + l1];
                    for (int q = b + 1; q <= L && q <= a + MAXLOOP; q++)
                        for (int l2 = q - a; l2 <= MAXLOOP; l2++) pu += GR[(size_t)q * (MAXLOOP + 2) + l2];
                    if (a >= 2)
                        for (int j = b + 2; j < L; j++) {
                            const double t = FM1o[off[a - 1] + j] + len * mlb + FM1i[off[b] + j] - Z;
                            if (t > -INFINITY) pu += exp(t);
                        }
                    for (int i = 1; i + 2 <= a - 1; i++)
                        if (b < L) {
                            const double t = FMSo[off[i] + b] + len * mlb + FMSi[off[i] + a - 1] - Z;
                            if (t > -INFINITY) pu += exp(t);
                        }
                }
                up[(size_t)(a - 1) * max_w + w] = pu;
            }
#undef HS
        free(Hs); free(GL); free(GR);
    }
#undef PAIR
#undef NICKED
    if (logz_out) *logz_out = F5o[0];
    if (tabs) memcpy(tabs, buf, sizeof(double) * 8 * T);
    if (f5) memcpy(f5, F5i, sizeof(double) * 2 * (L + 1));
    free(buf); free(off); free(S);
    return Z;
}

/* ---- brute force over all secondary structures (hairpins >= 3, every loop decomposed explicitly) */
typedef struct { mc_t c; int* pt; double Z; double* marg; double* up; int max_w; } sbf_t;

static double loop_logw(const sbf_t* b, int a, int bb)
{   /* log weight of the loop closed by (a,bb) and of everything nested in it */
    const mc_t* c = &b->c;
    int stems = 0, unpaired = 0, sp[64], sq[64];
    int nick = 0;   /* does the loop's own backbone hold the gap between the two molecules? */
    for (int x = a + 1; x < bb;) {
        if (c->cut > 0 && x - 1 == c->cut) nick = 1;          /* gap x-1 precedes the element starting at x */
        if (b->pt[x] > x) { sp[stems] = x; sq[stems] = b->pt[x]; stems++; x = b->pt[x] + 1; }
        else { unpaired++; x++; }
    }
    if (c->cut > 0 && bb - 1 == c->cut) nick = 1;
    double lw = 0.0;
    if (nick) {
        lw = -e_nickclose(c, a, bb) * c->sc;
        for (int k = 0; k < stems; k++) lw += -e_stem(c, sp[k], sq[k]) * c->sc;
    } else if (stems == 0) lw = -e_hairpin(c, a, bb) * c->sc;
    else if (stems == 1) {
        if (unpaired > MAXLOOP) return -INFINITY;
        lw = -e_interior(c, a, bb, sp[0], sq[0]) * c->sc;
    } else {
        lw = -(e_mlclose(c, a, bb) + c->P->ML_base * unpaired) * c->sc;
        for (int k = 0; k < stems; k++) lw += -(c->P->ML_intern + e_stem(c, sp[k], sq[k])) * c->sc;
    }
    for (int k = 0; k < stems; k++) lw += loop_logw(b, sp[k], sq[k]);
    return lw;
}
static void sbf_finish(sbf_t* b)
{
    const int n = b->c.n;
    double lw = 0.0;
    for (int x = 1; x <= n;) {
        if (b->pt[x] > x) { lw += -e_stem(&b->c, x, b->pt[x]) * b->c.sc + loop_logw(b, x, b->pt[x]); x = b->pt[x] + 1; }
        else x++;
    }
    if (lw == -INFINITY) return;
    const double w = exp(lw);
    b->Z += w;
    for (int x = 1; x <= n; x++)
        if (b->pt[x] > x) b->marg[(long)x * (2 * (n + 1) - x - 1) / 2 + b->pt[x]] += w;
    if (b->up)
        for (int a = 1; a <= n; a++)
            for (int e = a; e <= n && e - a < b->max_w && b->pt[e] < 0; e++) b->up[(size_t)(a - 1) * b->max_w + (e - a)] += w;
}
static void sbf_rec(sbf_t* b, int pos)
{   /* decide position pos: unpaired (-1), or paired with a later free position q; everything strictly between must
       still be undecided, which keeps the structures non-crossing (only openers left of pos can own a closer there) */
    const int n = b->c.n;
    while (pos <= n && b->pt[pos] != 0) pos++;
    if (pos > n) { sbf_finish(b); return; }
    b->pt[pos] = -1;
    sbf_rec(b, pos + 1);
    for (int q = pos + 4; q <= n; q++) {
        if (b->pt[q] != 0 || !ptype(b->c.S[pos], b->c.S[q])) continue;
        if (b->c.allow && !b->c.allow[(size_t)pos * (n + 1) + q]) continue;
        int ok = 1;
        for (int x = pos + 1; x < q && ok; x++) if (b->pt[x] > 0) ok = 0;
        if (!ok) continue;
        b->pt[pos] = q; b->pt[q] = pos;
        sbf_rec(b, pos + 1);
        b->pt[q] = 0;
    }
    b->pt[pos] = 0;
}
double vo_fold_bruteforce_cut(const vo_model* P, const char* seq, int n, int cut, double* post, double* up, int max_w);
double vo_fold_bruteforce(const vo_model* P, const char* seq, int n, double* post, double* up, int max_w)
{
    return vo_fold_bruteforce_cut(P, seq, n, 0, post, up, max_w);
}
double vo_fold_bruteforce_cut(const vo_model* P, const char* seq, int n, int cut, double* post, double* up, int max_w)
{
    int* S = (int*)calloc(n + 3, sizeof(int));
    for (int i = 1; i <= n; i++) S[i] = vcode(seq[i - 1]);
    sbf_t b;
    b.c.P = P; b.c.S = S; b.c.n = n; b.c.sc = 10.0 / P->kT; b.c.cut = cut; b.c.allow = vo_allow_mask;
    b.pt = (int*)calloc(n + 2, sizeof(int));
    b.Z = 0.0;
    const long T = (long)(n + 1) * (n + 2) / 2;
    b.marg = (double*)calloc(T, sizeof(double));
    b.up = up; b.max_w = max_w;
    if (up) for (long k = 0; k < (long)n * max_w; k++) up[k] = 0.0;
    sbf_rec(&b, 1);
    if (post) for (long k = 0; k < T; k++) post[k] = b.marg[k] / b.Z;
    if (up) for (long k = 0; k < (long)n * max_w; k++) up[k] /= b.Z;
    const double z = log(b.Z);
    free(b.marg); free(b.pt); free(S);
    return z;
}
