// vienna_loader.cpp -- read ractip_amd/data/vienna_bl_star.params and build rh::ViennaDx (see vienna_model.h).
// The flat arrays keep the order of /root/reference/src/boltzmann_param.c; their index conventions are those of
// its copy_* loops (:5908-5971): pair types 1..7 for stack/mismatch/int11/int21/int22, nucleotides 0..4 except
// int22 (1..4), dangles 0..7 x 0..4.
#include "vienna_model.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace rh {

bool load_vienna_dx(const char* path, ViennaDx* V, char* err, int errlen)
{
    std::ifstream f(path);
    if (!f) { snprintf(err, errlen, "cannot open parameter file %s", path); return false; }
    std::map<std::string, std::vector<int>> tab;
    std::vector<std::pair<std::string, int>> tetra;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream hs(line);
        std::string name;
        int count = 0;
        if (!(hs >> name >> count) || count <= 0) { snprintf(err, errlen, "bad header '%s' in %s", line.c_str(), path); return false; }
        if (name == "tetraloops") {   // `count` lines "CGAAAG -160": closing pair + 4 loop letters, bonus energy
            for (int k = 0; k < count; k++) {
                std::string sq; int e;
                if (!(f >> sq >> e) || sq.size() != 6) { snprintf(err, errlen, "tetraloop list truncated in %s", path); return false; }
                tetra.emplace_back(sq, e);
            }
            std::getline(f, line);
            continue;
        }
        std::vector<int>& v = tab[name];
        v.resize(count);
        for (int k = 0; k < count; k++)
            if (!(f >> v[k])) { snprintf(err, errlen, "table %s truncated in %s", name.c_str(), path); return false; }
        std::getline(f, line);  // rest of the last value line
    }
    const struct { const char* name; size_t n; } need[] = {{"stack37", 49}, {"mismatchI37", 175}, {"dangle5_37", 40}, {"dangle3_37", 40},
        {"int11_37", 1225}, {"int21_37", 6125}, {"int22_37", 12544}, {"bulge37", 31}, {"internal_loop37", 31}, {"MLparams", 4}, {"ninio", 2},
        {"hairpin37", 31}, {"mismatchH37", 175}};
    for (auto& t : need)
        if (tab[t.name].size() != t.n) { snprintf(err, errlen, "table %s missing or of wrong size in %s", t.name, path); return false; }

    std::memset(V, 0, sizeof(*V));
    const double kT = (37.0 + 273.15) * 1.98717;   // (temperature+K0)*GASCONST, pf_duplex.c:73
    auto w = [&](int E) { return -E * 10.0 / kT; };
    const std::vector<int>&stack = tab["stack37"], &mmI = tab["mismatchI37"], &d5 = tab["dangle5_37"], &d3 = tab["dangle3_37"],
                     &i11 = tab["int11_37"], &i21 = tab["int21_37"], &i22 = tab["int22_37"], &bulge = tab["bulge37"],
                     &il = tab["internal_loop37"];
    const int tau = tab["MLparams"][3], ninio = tab["ninio"][0], max_ninio = tab["ninio"][1];
    int p = 0;
    for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) {
        V->stack[i * 8 + j] = w(stack[p]);
        V->bulge1[i * 8 + j] = w(bulge[1] + stack[p]);
        p++;
    }
    p = 0;
    for (int i = 1; i <= 7; i++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) V->mmI[i * 25 + a * 5 + b] = w(mmI[p++]);
    p = 0;
    for (int i = 0; i <= 7; i++) for (int a = 0; a < 5; a++, p++) {
        V->dangle5[i * 5 + a] = w(std::min(d5[p], 0));   // dangles are clipped to <= 0 by scale_parameters()
        V->dangle3[i * 5 + a] = w(std::min(d3[p], 0));
    }
    p = 0;
    for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++)
        V->int11[(i * 8 + j) * 25 + a * 5 + b] = w(i11[p++]);
    p = 0;
    for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) for (int c = 0; c < 5; c++)
        V->int21[(i * 8 + j) * 125 + (a * 5 + b) * 5 + c] = w(i21[p++]);
    p = 0;
    for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 1; a < 5; a++) for (int b = 1; b < 5; b++)
        for (int c = 1; c < 5; c++) for (int d = 1; d < 5; d++)
            V->int22[(i * 8 + j) * 625 + ((a * 5 + b) * 5 + c) * 5 + d] = w(i22[p++]);
    V->tau = w(tau);
    V->duplex_init = w(410);
    // loop shapes, row-major (l1, l2); length-dependent part of LoopEnergy (ViennaRNA 1.8)
    int k = 0;
    for (int l1 = 0; l1 <= 30; l1++)
        for (int l2 = 0; l1 + l2 <= 30; l2++, k++) {
            const int nl = std::max(l1, l2), ns = std::min(l1, l2);
            int kind, E = 0;
            if (nl <= 2 && !(ns == 0 && nl == 2)) kind = 0;   // (0,0) (0,1) (1,0) (1,1) (1,2) (2,1) (2,2): explicit tables
            else if (ns == 0) { kind = 2; E = bulge[nl]; }
            else { kind = 1; E = il[l1 + l2] + std::min(max_ninio, (nl - ns) * ninio); }
            V->shape[k] = Shape{w(E), l1, l2};
            V->kind[k] = kind;
        }
    for (; k < kMcShapes; k++) { V->shape[k] = Shape{0.0, 1000, 1000}; V->kind[k] = 1; }
    const int T[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, 1, 0}, {0, 0, 2, 0, 3}, {0, 6, 0, 4, 0}};
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) V->ptype[a * 5 + b] = T[a][b];
    const int R[8] = {0, 2, 1, 4, 3, 6, 5, 7};
    for (int t = 0; t < 8; t++) V->rtype[t] = R[t];

    // ---- McCaskill part (part_func.c of ViennaRNA 1.8: scale_pf_params / expHairpinEnergy / the qm, qqm, q recurrences)
    const std::vector<int>&hp = tab["hairpin37"], &mmH = tab["mismatchH37"], &ml = tab["MLparams"];
    p = 0;
    for (int i = 1; i <= 7; i++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) V->mmH[i * 25 + a * 5 + b] = w(mmH[p++]);
    for (int u = 0; u <= 30; u++) V->hairpin[u] = w(hp[u]);
    V->hairpin30 = w(hp[30]);
    V->lxc = 107.856 * 10.0 / kT;   // lxc37 (ViennaRNA constant, not overridden by BL*)
    for (auto& t : tetra) {
        int code = 0;
        bool ok = true;
        for (char ch : t.first) {
            const char* q = std::strchr("ACGU", ch);
            if (!q) { ok = false; break; }
            code = code * 4 + (int)(q - "ACGU");
        }
        if (ok) V->tetra[code] = w(t.second);
    }
    // stems of exterior / multi loops: dangles on both sides whenever the neighbour exists, not clipped but smoothed
    // (SMOOTH of part_func.c), TerminalAU folded into the 3' dangle; nucleotide code 0 = no neighbour
    auto smooth = [](double X) {
        const double x = X / 10.0;
        if (x < -1.2283697) return 0.0;
        if (x > 0.8660254) return X;
        const double t = std::sin(x - 0.34242663) + 1.0;
        return 10.0 * 0.38490018 * t * t;
    };
    p = 0;
    for (int i = 0; i <= 7; i++) for (int a = 0; a < 5; a++, p++) {
        const double tau_e = i > 2 ? (double)tau : 0.0;
        V->d5x[i * 5 + a] = a ? smooth(-(double)d5[p]) * 10.0 / kT : 0.0;
        V->d3x[i * 5 + a] = (a ? smooth(-(double)d3[p]) * 10.0 / kT : 0.0) - tau_e * 10.0 / kT;
    }
    V->ml_close = w(ml[1] + ml[2]);
    V->mli = w(ml[2]);
    V->mlb = w(ml[0]);
    return true;
}

void build_vlin_model(const ViennaDx& V, double s, VLinModel* L)
{
    std::memset(L, 0, sizeof(*L));
    const double lam = std::exp(-s);
    L->s = s; L->lam = lam; L->lam2 = lam * lam;
    L->w_mu = lam * std::exp(V.mlb);
    L->w_mp2 = lam * lam * std::exp(V.mli);
    L->hairpin30 = V.hairpin30; L->lxc = V.lxc;
    std::memcpy(L->ptype, V.ptype, sizeof L->ptype);
    std::memcpy(L->rtype, V.rtype, sizeof L->rtype);
    for (int t = 0; t < 8; t++) L->E_tau[t] = t > 2 ? std::exp(V.tau) : 1.0;
    for (int x = 0; x < 5; x++) for (int x1 = 0; x1 < 5; x1++) for (int y1 = 0; y1 < 5; y1++) for (int y = 0; y < 5; y++) {
        const int idx = 25 * (5 * x + x1) + (5 * y1 + y);
        const int ti = V.ptype[x * 5 + y1];          // seen from inside: letters (i, j+1) = (x, y1)
        if (ti) {
            const int rt = V.rtype[ti];
            L->TXO[idx] = std::exp(V.mmI[ti * 25 + x1 * 5 + y]);
            L->TMC[idx] = std::exp(V.ml_close + V.d3x[rt * 5 + x1] + V.d5x[rt * 5 + y]);
            L->TMH[idx] = std::exp(V.mmH[ti * 25 + x1 * 5 + y]);
            L->TNC[idx] = std::exp(V.d3x[rt * 5 + x1] + V.d5x[rt * 5 + y]);
        }
        const int to = V.ptype[y1 * 5 + x];          // seen from outside: letters (i, j+1) = (y1, x)
        if (to) {
            const int rt = V.rtype[to];
            L->TXI[idx] = std::exp(V.mmI[rt * 25 + x1 * 5 + y]);
            L->TSA[idx] = std::exp(V.d5x[to * 5 + y] + V.d3x[to * 5 + x1]);
        }
    }
    const double l2 = lam * lam, l3 = l2 * lam, l4 = l2 * l2, l5 = l4 * lam, l6 = l4 * l2;
    for (int k = 0; k < 64; k++) { L->E_stack[k] = std::exp(V.stack[k]) * l2; L->E_bulge1[k] = std::exp(V.bulge1[k]) * l3; }
    for (int k = 0; k < 64 * 25; k++) L->E_int11[k] = std::exp(V.int11[k]) * l4;
    for (int k = 0; k < 64 * 125; k++) L->E_int21[k] = std::exp(V.int21[k]) * l5;
    for (int k = 0; k < 64 * 625; k++) L->E_int22[k] = std::exp(V.int22[k]) * l6;
    // type 0 rows/columns of the joint tables hold log-weight 0 -> weight 1; they are only ever multiplied by an FC of 0
    for (int k = 0; k < 4096; k++) L->E_tetra[k] = std::exp(V.tetra[k]);
    for (int u = 0; u <= 30; u++) L->E_hairpin[u] = std::exp(V.hairpin[u]);
    L->E_hairpin[31] = L->E_hairpin[30];
    // generic loops and long bulges from the row-major (l1,l2) shape list
    auto at = [&](int l1, int l2) { return l1 * 31 - l1 * (l1 - 1) / 2 + l2; };
    int k = 0;
    for (int t = 0; t <= kMaxSingle; t++)
        for (int l1 = 0; l1 <= t; l1++, k++) {
            const int q = at(l1, t - l1);
            L->shape_w[k] = V.kind[q] == 1 ? std::exp(V.shape[q].score) * std::pow(lam, t + 2) : 0.0;
        }
    for (int l = 2; l <= kMaxSingle; l++) L->WB[l] = std::exp(V.shape[at(l, 0)].score) * std::pow(lam, l + 2);
}

void build_vdx_lin(const ViennaDx& V, double s, VDxLin* D)
{
    std::memset(D, 0, sizeof(*D));
    for (int k = 0; k < 200; k++) D->E_mmI[k] = std::exp(V.mmI[k]);
    for (int k = 0; k < 40; k++) { D->E_d5[k] = std::exp(V.dangle5[k]); D->E_d3[k] = std::exp(V.dangle3[k]); }
    D->E_init = std::exp(V.duplex_init);
    D->s = s; D->lam = std::exp(-s);
}

}  // namespace rh

// Host-only inspection hook for tests/test_bl_cells.py (no GPU involved): the energy, in the file's 10 cal/mol units, that
// the product's loader holds for one table cell after binding the flat BL* arrays -- so that the binding can be checked
// against cells labelled independently from the reference's block comments (tests/golden/bl_star_cells.json).
// table: 0 = stack[i][j], 1 = int11[i][j][k][l], 2 = int21[i][j][k][l][m], 3 = int22[i][j][k][l][m][n].
extern "C" int rh_debug_vienna_cell(const char* param_file, int table, int i, int j, int k, int l, int m, int n, double* energy)
{
    static rh::ViennaDx* V = nullptr;
    static std::string loaded;
    if (!param_file || !energy) return -1;
    if (!V || loaded != param_file) {
        delete V;
        V = new rh::ViennaDx;
        char err[256];
        if (!rh::load_vienna_dx(param_file, V, err, sizeof err)) { delete V; V = nullptr; return -4; }
        loaded = param_file;
    }
    const double kT = (37.0 + 273.15) * 1.98717;
    if (i < 0 || i > 7 || j < 0 || j > 7) return -1;
    const int tt = i * 8 + j;
    double w;
    switch (table) {
        case 0: w = V->stack[tt]; break;
        case 1: if (k < 0 || k > 4 || l < 0 || l > 4) return -1; w = V->int11[tt * 25 + k * 5 + l]; break;
        case 2: if (k < 0 || k > 4 || l < 0 || l > 4 || m < 0 || m > 4) return -1; w = V->int21[tt * 125 + (k * 5 + l) * 5 + m]; break;
        case 3: if (k < 0 || k > 4 || l < 0 || l > 4 || m < 0 || m > 4 || n < 0 || n > 4) return -1; w = V->int22[tt * 625 + ((k * 5 + l) * 5 + m) * 5 + n]; break;
        default: return -1;
    }
    *energy = -w * kT / 10.0;
    return 0;
}
