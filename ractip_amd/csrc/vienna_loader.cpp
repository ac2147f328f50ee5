// vienna_loader.cpp -- read the Vienna energy tables and build rh::ViennaDx (see vienna_model.h).  Sources, applied in the order
// RactIP::run installs them (/root/reference/src/ractip.cpp:1563-1567): the library defaults (a ViennaRNA parameter file, if the
// caller has one), the BL* tables (ractip_amd/data/vienna_bl_star.params, the flat dump of src/boltzmann_param.c), the -P file.
#include "vienna_model.h"

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace rh {

namespace {

constexpr int kInf = 1000000;   // INF of the parameter files

// every table of either file layout, in the files' 10 cal/mol integers, indexed as ViennaRNA indexes them
struct ViennaInts {
    int stack[8][8];
    int mmH[8][5][5], mmI[8][5][5], mm1nI[8][5][5], mm23I[8][5][5], mmM[8][5][5], mmExt[8][5][5];
    int d5[8][5], d3[8][5];
    int int11[8][8][5][5], int21[8][8][5][5][5], int22[8][8][5][5][5][5];
    int hairpin[31], bulge[31], il[31];
    int ml_base, ml_closing, ml_intern, ninio, max_ninio, tau, duplex_init;
};
struct ViennaTables : ViennaInts {
    double lxc;
    // special hairpins: (letters, energy).  `special_total`: the energies are whole hairpin energies (v2.0 files) rather than bonuses
    std::vector<std::pair<std::string, int>> tetra, tri, hexa;
    bool special_total;
    bool v20;   // a v2.0 parameter file was among the sources
    unsigned provided;   // bit k: table kCore[k] was supplied by at least one source (completeness check of load_vienna_dx_ex)
};
// the tables every kernel family reads under either semantics; the 2.x-only mismatch tables and the special hairpins may be
// absent (then zero / none: DESIGN.md section 2)
const char* const kCore[] = {"stack", "mismatch_interior", "mismatch_hairpin", "dangle5", "dangle3", "int11", "int21", "int22",
                             "bulge", "interior", "hairpin", "ML_params", "NINIO"};
constexpr int kNCore = sizeof(kCore) / sizeof(kCore[0]);
void mark(ViennaTables* T, const char* core) { for (int k = 0; k < kNCore; k++) if (!std::strcmp(kCore[k], core)) T->provided |= 1u << k; }

void default_tables(ViennaTables* T)
{
    std::memset(static_cast<ViennaInts*>(T), 0, sizeof(ViennaInts));
    T->duplex_init = 410;    // DuplexInit, ViennaRNA constant (BL* does not override it)
    T->lxc = 107.856;        // lxc37
    T->max_ninio = 300;
    T->tetra.clear(); T->tri.clear(); T->hexa.clear();
    T->special_total = false; T->v20 = false; T->provided = 0;
}

bool fail_msg(char* err, int errlen, const char* fmt, const char* a, const char* b = "")
{
    snprintf(err, errlen, fmt, a, b);
    return false;
}

// ---- the flat dump of the BL* tables (ractip_amd/data/vienna_bl_star.params): "name count" + values, in the order of boltzmann_param.c; index
// conventions of its copy_* loops (:5908-5971): pair types 1..7 for stack/mismatch/int11/int21/int22, nucleotides 0..4 except
// int22 (1..4), dangles 0..7 x 0..4.  Tables the file does not hold keep their value.
bool read_flat(const char* path, ViennaTables* T, char* err, int errlen)
{
    std::ifstream f(path);
    if (!f) return fail_msg(err, errlen, "cannot open parameter file %s", path);
    std::map<std::string, std::vector<int>> tab;
    std::map<std::string, std::vector<std::pair<std::string, int>>> lists;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream hs(line);
        std::string name;
        int count = 0;
        if (!(hs >> name >> count) || count <= 0) return fail_msg(err, errlen, "bad header '%s' in %s", line.c_str(), path);
        if (name == "tetraloops" || name == "triloops" || name == "hexaloops") {   // `count` lines "CGAAAG -160": closing pair + loop letters, energy
            for (int k = 0; k < count; k++) {
                std::string sq; int e;
                if (!(f >> sq >> e)) return fail_msg(err, errlen, "%s list truncated in %s", name.c_str(), path);
                lists[name].emplace_back(sq, e);
            }
            std::getline(f, line);
            continue;
        }
        std::vector<int>& v = tab[name];
        v.resize(count);
        for (int k = 0; k < count; k++)
            if (!(f >> v[k])) return fail_msg(err, errlen, "table %s truncated in %s", name.c_str(), path);
        std::getline(f, line);  // rest of the last value line
    }
    const struct { const char* name; size_t n; } sizes[] = {{"stack37", 49}, {"mismatchI37", 175}, {"mismatchH37", 175}, {"mismatch1nI37", 175},
        {"mismatch23I37", 175}, {"mismatchM37", 175}, {"mismatchExt37", 175}, {"dangle5_37", 40}, {"dangle3_37", 40}, {"int11_37", 1225},
        {"int21_37", 6125}, {"int22_37", 12544}, {"bulge37", 31}, {"internal_loop37", 31}, {"hairpin37", 31}, {"MLparams", 4}, {"ninio", 2}};
    for (auto& t : sizes)
        if (tab.count(t.name) && tab[t.name].size() != t.n) return fail_msg(err, errlen, "table %s of wrong size in %s", t.name, path);
    {
        const struct { const char* flat; const char* core; } names[] = {{"stack37", "stack"}, {"mismatchI37", "mismatch_interior"}, {"mismatchH37", "mismatch_hairpin"},
            {"dangle5_37", "dangle5"}, {"dangle3_37", "dangle3"}, {"int11_37", "int11"}, {"int21_37", "int21"}, {"int22_37", "int22"}, {"bulge37", "bulge"},
            {"internal_loop37", "interior"}, {"hairpin37", "hairpin"}, {"MLparams", "ML_params"}, {"ninio", "NINIO"}};
        for (auto& nm : names) if (tab.count(nm.flat)) mark(T, nm.core);
    }
    auto mism = [&](const char* name, int (*dst)[5][5]) {
        if (!tab.count(name)) return;
        const std::vector<int>& v = tab[name];
        int p = 0;
        for (int i = 1; i <= 7; i++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) dst[i][a][b] = v[p++];
    };
    int p;
    if (tab.count("stack37")) { p = 0; for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) T->stack[i][j] = tab["stack37"][p++]; }
    mism("mismatchI37", T->mmI); mism("mismatchH37", T->mmH); mism("mismatch1nI37", T->mm1nI); mism("mismatch23I37", T->mm23I);
    mism("mismatchM37", T->mmM); mism("mismatchExt37", T->mmExt);
    if (tab.count("dangle5_37")) { p = 0; for (int i = 0; i <= 7; i++) for (int a = 0; a < 5; a++) T->d5[i][a] = tab["dangle5_37"][p++]; }
    if (tab.count("dangle3_37")) { p = 0; for (int i = 0; i <= 7; i++) for (int a = 0; a < 5; a++) T->d3[i][a] = tab["dangle3_37"][p++]; }
    if (tab.count("int11_37")) {
        p = 0;
        for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) T->int11[i][j][a][b] = tab["int11_37"][p++];
    }
    if (tab.count("int21_37")) {
        p = 0;
        for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) for (int c = 0; c < 5; c++)
            T->int21[i][j][a][b][c] = tab["int21_37"][p++];
    }
    if (tab.count("int22_37")) {
        p = 0;
        for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 1; a < 5; a++) for (int b = 1; b < 5; b++)
            for (int c = 1; c < 5; c++) for (int d = 1; d < 5; d++) T->int22[i][j][a][b][c][d] = tab["int22_37"][p++];
    }
    if (tab.count("bulge37")) std::copy(tab["bulge37"].begin(), tab["bulge37"].end(), T->bulge);
    if (tab.count("internal_loop37")) std::copy(tab["internal_loop37"].begin(), tab["internal_loop37"].end(), T->il);
    if (tab.count("hairpin37")) std::copy(tab["hairpin37"].begin(), tab["hairpin37"].end(), T->hairpin);
    if (tab.count("MLparams")) {   // copy_MLparams, boltzmann_param.c:5973-5983: cu, cc, ci, TerminalAU
        T->ml_base = tab["MLparams"][0]; T->ml_closing = tab["MLparams"][1]; T->ml_intern = tab["MLparams"][2]; T->tau = tab["MLparams"][3];
    }
    if (tab.count("ninio")) { T->ninio = tab["ninio"][0]; T->max_ninio = tab["ninio"][1]; }
    if (lists.count("tetraloops")) T->tetra = lists["tetraloops"];
    if (lists.count("triloops")) T->tri = lists["triloops"];
    if (lists.count("hexaloops")) T->hexa = lists["hexaloops"];
    return true;
}

// ---- ViennaRNA parameter files ("## RNAfold parameter file" / "... v2.0"): sections "# name", C comments anywhere, INF and DEF
// tokens (DEF = leave the value).  The layout of a table follows from the number of values the section holds: with or without
// the row of pair type 0 (NP), int22 over 6x6 or 7x7 pair types (nucleotides A..U only).  *_enthalpies sections are skipped
// (37 C only, as RactIP runs).  The semantics of the sections follows ViennaRNA's read_epars.c as published; nothing in the
// reference repository pins it (PARITY UNPINNED).
bool read_par(const char* path, const std::string& text_in, ViennaTables* T, char* err, int errlen)
{
    std::string text = text_in;
    for (size_t a; (a = text.find("/*")) != std::string::npos;) {   // comments may span lines
        const size_t b = text.find("*/", a + 2);
        text.erase(a, b == std::string::npos ? std::string::npos : b + 2 - a);
    }
    std::istringstream in(text);
    std::string line, section;
    std::map<std::string, std::vector<std::string>> sec;       // numeric sections: tokens
    std::map<std::string, std::vector<std::pair<std::string, int>>> loops;
    bool v20 = false;
    while (std::getline(in, line)) {
        const size_t h = line.find_first_not_of(" \t\r");
        if (h == std::string::npos) continue;
        if (line[h] == '#') {
            if (line.compare(h, 2, "##") == 0) { if (line.find("v2.0") != std::string::npos) v20 = true; section.clear(); continue; }
            std::istringstream hs(line.substr(h + 1));
            section.clear();
            hs >> section;
            continue;
        }
        if (section.empty()) continue;
        std::istringstream ls(line);
        if (section == "Tetraloops" || section == "Triloops" || section == "Hexaloops") {
            std::string sq, e;
            if (ls >> sq >> e) loops[section].emplace_back(sq, e == "INF" ? kInf : std::atoi(e.c_str()));
            continue;
        }
        for (std::string tok; ls >> tok;) sec[section].push_back(tok);
    }
    // values of a section into dst[k] for the positions `slots` lists (DEF keeps the old value)
    bool ok = true;
    auto fill = [&](const std::string& name, const std::vector<int*>& slots) {
        const std::vector<std::string>& v = sec[name];
        if (v.size() != slots.size()) { snprintf(err, errlen, "section %s of %s holds %zu values, expected %zu", name.c_str(), path, v.size(), slots.size()); ok = false; return; }
        for (size_t k = 0; k < v.size(); k++) {
            if (v[k] == "DEF") continue;
            if (v[k] == "INF") { *slots[k] = kInf; continue; }
            char* end = nullptr;
            const long x = std::strtol(v[k].c_str(), &end, 10);
            if (end == v[k].c_str()) { snprintf(err, errlen, "bad value '%s' in section %s of %s", v[k].c_str(), name.c_str(), path); ok = false; return; }
            *slots[k] = (int)x;
        }
    };
    auto first_of = [&](std::initializer_list<const char*> names) -> std::string {
        for (const char* n : names) if (sec.count(n)) return n;
        return "";
    };
    auto mism = [&](std::initializer_list<const char*> names, int (*dst)[5][5]) {
        const std::string n = first_of(names);
        if (n.empty() || !ok) return;
        const int t0 = sec[n].size() == 200 ? 0 : 1;
        std::vector<int*> slots;
        for (int i = t0; i <= 7; i++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) slots.push_back(&dst[i][a][b]);
        fill(n, slots);
    };
    {
        const std::string n = first_of({"stack", "stack_energies"});
        if (!n.empty()) {
            const int t0 = sec[n].size() == 64 ? 0 : 1;
            std::vector<int*> slots;
            for (int i = t0; i <= 7; i++) for (int j = t0; j <= 7; j++) slots.push_back(&T->stack[i][j]);
            fill(n, slots);
        }
    }
    mism({"mismatch_hairpin"}, T->mmH); mism({"mismatch_interior"}, T->mmI); mism({"mismatch_interior_1n"}, T->mm1nI);
    mism({"mismatch_interior_23"}, T->mm23I); mism({"mismatch_multi"}, T->mmM); mism({"mismatch_exterior"}, T->mmExt);
    for (int which = 0; which < 2 && ok; which++) {
        const char* n = which ? "dangle3" : "dangle5";
        if (!sec.count(n)) continue;
        int (*dst)[5] = which ? T->d3 : T->d5;
        const int t0 = sec[n].size() == 40 ? 0 : 1;
        std::vector<int*> slots;
        for (int i = t0; i <= 7; i++) for (int a = 0; a < 5; a++) slots.push_back(&dst[i][a]);
        fill(n, slots);
    }
    if (ok) {
        const std::string n = first_of({"int11", "int11_energies"});
        if (!n.empty()) {
            std::vector<int*> slots;
            for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) slots.push_back(&T->int11[i][j][a][b]);
            fill(n, slots);
        }
    }
    if (ok) {
        const std::string n = first_of({"int21", "int21_energies"});
        if (!n.empty()) {
            std::vector<int*> slots;
            for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) for (int c = 0; c < 5; c++)
                slots.push_back(&T->int21[i][j][a][b][c]);
            fill(n, slots);
        }
    }
    if (ok) {
        const std::string n = first_of({"int22", "int22_energies"});
        if (!n.empty()) {
            const int tmax = sec[n].size() == 12544 ? 7 : 6;   // the files leave the non-standard pair type to the built-in defaults
            std::vector<int*> slots;
            for (int i = 1; i <= tmax; i++) for (int j = 1; j <= tmax; j++) for (int a = 1; a < 5; a++) for (int b = 1; b < 5; b++)
                for (int c = 1; c < 5; c++) for (int d = 1; d < 5; d++) slots.push_back(&T->int22[i][j][a][b][c][d]);
            fill(n, slots);
        }
    }
    auto lens = [&](std::initializer_list<const char*> names, int* dst) {
        const std::string n = first_of(names);
        if (n.empty() || !ok) return;
        std::vector<int*> slots;
        for (int u = 0; u <= 30; u++) slots.push_back(&dst[u]);
        fill(n, slots);
    };
    lens({"hairpin"}, T->hairpin); lens({"bulge"}, T->bulge); lens({"interior", "internal_loop"}, T->il);
    int dummy = 0;
    if (ok && sec.count("NINIO")) {
        if (sec["NINIO"].size() == 3) fill("NINIO", {&T->ninio, &dummy, &T->max_ninio});        // m, m_dH, max
        else fill("NINIO", {&T->ninio, &T->max_ninio});                                          // 1.x: m, max
    }
    if (ok && sec.count("ML_params")) {
        if (sec["ML_params"].size() == 6) fill("ML_params", {&T->ml_base, &dummy, &T->ml_closing, &dummy, &T->ml_intern, &dummy});   // cu cu_dH cc cc_dH ci ci_dH
        else fill("ML_params", {&T->ml_base, &T->ml_closing, &T->ml_intern, &T->tau});                                              // 1.x: cu cc ci TerminalAU
    }
    if (ok && sec.count("Misc")) {   // DuplexInit DuplexInit_dH TerminalAU TerminalAU_dH [lxc lxc_dH]
        const std::vector<std::string>& v = sec["Misc"];
        if (v.size() >= 4) {
            if (v[0] != "DEF") T->duplex_init = std::atoi(v[0].c_str());
            if (v[2] != "DEF") T->tau = std::atoi(v[2].c_str());
            if (v.size() >= 5 && v[4] != "DEF") T->lxc = std::atof(v[4].c_str());
        }
    }
    if (!ok) return false;
    {
        const struct { const char* a; const char* b; const char* core; } names[] = {{"stack", "stack_energies", "stack"},
            {"mismatch_interior", "", "mismatch_interior"}, {"mismatch_hairpin", "", "mismatch_hairpin"}, {"dangle5", "", "dangle5"}, {"dangle3", "", "dangle3"},
            {"int11", "int11_energies", "int11"}, {"int21", "int21_energies", "int21"}, {"int22", "int22_energies", "int22"}, {"bulge", "", "bulge"},
            {"interior", "internal_loop", "interior"}, {"hairpin", "", "hairpin"}, {"ML_params", "", "ML_params"}, {"NINIO", "", "NINIO"}};
        for (auto& nm : names) if (sec.count(nm.a) || sec.count(nm.b)) mark(T, nm.core);
    }
    if (loops.count("Tetraloops")) T->tetra = loops["Tetraloops"];
    if (loops.count("Triloops")) T->tri = loops["Triloops"];
    if (loops.count("Hexaloops")) T->hexa = loops["Hexaloops"];
    if (v20) { T->v20 = true; T->special_total = true; }
    return true;
}

bool read_any(const char* path, ViennaTables* T, char* err, int errlen)
{
    std::ifstream f(path);
    if (!f) return fail_msg(err, errlen, "cannot open parameter file %s", path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    if (text.find("## RNAfold parameter file") != std::string::npos) return read_par(path, text, T, err, errlen);
    return read_flat(path, T, err, errlen);
}

int letter_code4(char ch)
{
    const char* q = std::strchr("ACGU", ch == 'T' ? 'U' : ch);
    return (q && ch) ? (int)(q - "ACGU") : -1;
}

void build_vienna_dx(const ViennaTables& T, int semantics, ViennaDx* V)
{
    std::memset(static_cast<void*>(V), 0, sizeof(*V));
    V->semantics = semantics;
    const bool s20 = semantics == kViennaSem20;
    const double kT = (37.0 + 273.15) * 1.98717;   // (temperature+K0)*GASCONST, pf_duplex.c:73
    auto w = [&](int E) { return -E * 10.0 / kT; };
    for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) {
        V->stack[i * 8 + j] = w(T.stack[i][j]);
        V->bulge1[i * 8 + j] = w(T.bulge[1] + T.stack[i][j]);
    }
    for (int i = 1; i <= 7; i++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
        V->mmI[i * 25 + a * 5 + b] = w(T.mmI[i][a][b]);
        V->mmH[i * 25 + a * 5 + b] = w(T.mmH[i][a][b]);
        if (s20) { V->mm1nI[i * 25 + a * 5 + b] = w(T.mm1nI[i][a][b]); V->mm23I[i * 25 + a * 5 + b] = w(T.mm23I[i][a][b]); }
    }
    for (int i = 0; i <= 7; i++) for (int a = 0; a < 5; a++) {
        V->dangle5[i * 5 + a] = w(std::min(T.d5[i][a], 0));   // dangles are clipped to <= 0 by scale_parameters()
        V->dangle3[i * 5 + a] = w(std::min(T.d3[i][a], 0));
    }
    for (int i = 1; i <= 7; i++) for (int j = 1; j <= 7; j++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
        V->int11[(i * 8 + j) * 25 + a * 5 + b] = w(T.int11[i][j][a][b]);
        for (int c = 0; c < 5; c++) {
            V->int21[(i * 8 + j) * 125 + (a * 5 + b) * 5 + c] = w(T.int21[i][j][a][b][c]);
            for (int d = 0; d < 5; d++) V->int22[(i * 8 + j) * 625 + ((a * 5 + b) * 5 + c) * 5 + d] = w(T.int22[i][j][a][b][c][d]);
        }
    }
    V->tau = w(T.tau);
    V->duplex_init = w(T.duplex_init);
    // loop shapes, row-major (l1, l2); length-dependent part of LoopEnergy (1.8) / E_IntLoop (2.x)
    int k = 0;
    for (int l1 = 0; l1 <= 30; l1++)
        for (int l2 = 0; l1 + l2 <= 30; l2++, k++) {
            const int nl = std::max(l1, l2), ns = std::min(l1, l2);
            int kind, E = 0;
            if (nl <= 2 && !(ns == 0 && nl == 2)) kind = 0;   // (0,0) (0,1) (1,0) (1,1) (1,2) (2,1) (2,2): explicit tables
            else if (ns == 0) { kind = 2; E = T.bulge[nl]; }
            else if (s20 && ns == 1) { kind = 3; E = T.il[nl + 1] + std::min(T.max_ninio, (nl - ns) * T.ninio); }   // 1xn, n >= 3
            else if (s20 && ns == 2 && nl == 3) { kind = 4; E = T.il[5] + T.ninio; }                                 // 2x3
            else { kind = 1; E = T.il[l1 + l2] + std::min(T.max_ninio, (nl - ns) * T.ninio); }
            V->shape[k] = Shape{w(E), l1, l2};
            V->kind[k] = kind;
        }
    for (; k < kMcShapes; k++) { V->shape[k] = Shape{0.0, 1000, 1000}; V->kind[k] = 1; }
    const int PT[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 5}, {0, 0, 0, 1, 0}, {0, 0, 2, 0, 3}, {0, 6, 0, 4, 0}};
    for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) V->ptype[a * 5 + b] = PT[a][b];
    const int R[8] = {0, 2, 1, 4, 3, 6, 5, 7};
    for (int t = 0; t < 8; t++) V->rtype[t] = R[t];

    // ---- ends of a duplex (integer energies, as pf_duplex.c evaluates them): 1.8 = clipped dangles on the sides that exist
    // (:321-326, 337-340); 2.x = E_ExtLoop (:146,158,185,200): mismatchExt where both neighbours exist, else the one dangle;
    // scale_parameters() clips dangles and mismatchExt to <= 0
    for (int t = 1; t <= 7; t++) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
        int E;
        const int d5c = a < 5 ? std::min(T.d5[t][a], 0) : 0, d3c = b < 5 ? std::min(T.d3[t][b], 0) : 0;
        if (!s20) E = d5c + d3c;
        else if (a < 5 && b < 5) E = std::min(T.mmExt[t][a][b], 0);
        else E = a < 5 ? d5c : d3c;
        if (t > 2) E += T.tau;
        V->dxE[t * 36 + a * 6 + b] = w(E);
    }

    // ---- McCaskill part.  Boltzmann factors of dangles and of mismatchM / mismatchExt are smoothed, not clipped (SMOOTH of
    // part_func.c 1.8 / params.c 2.x): 0 for X/10 < -1.2283697, X for X/10 > 0.8660254, else 10*0.38490018*(sin(X/10-0.34242663)+1)^2
    for (int u = 0; u <= 30; u++) V->hairpin[u] = w(T.hairpin[u]);
    V->hairpin30 = w(T.hairpin[30]);
    V->lxc = T.lxc * 10.0 / kT;
    auto smooth = [](double X) {
        const double x = X / 10.0;
        if (x < -1.2283697) return 0.0;
        if (x > 0.8660254) return X;
        const double t = std::sin(x - 0.34242663) + 1.0;
        return 10.0 * 0.38490018 * t * t;
    };
    auto sm = [&](int E) { return smooth(-(double)E) * 10.0 / kT; };
    for (int i = 0; i <= 7; i++) for (int a = 0; a < 5; a++) {   // nucleotide code 0 = no neighbour; TerminalAU folded into the 3' dangle
        const double tau_e = i > 2 ? (double)T.tau : 0.0;
        V->d5x[i * 5 + a] = a ? sm(T.d5[i][a]) : 0.0;
        V->d3x[i * 5 + a] = (a ? sm(T.d3[i][a]) : 0.0) - tau_e * 10.0 / kT;
    }
    for (int t = 1; t <= 7; t++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
        const double tau_w = t > 2 ? w(T.tau) : 0.0;
        double e, m;
        if (!s20 || !(a && b)) e = m = (a ? sm(T.d5[t][a]) : 0.0) + (b ? sm(T.d3[t][b]) : 0.0);   // 2.x with one neighbour: that dangle alone
        else { e = sm(T.mmExt[t][a][b]); m = sm(T.mmM[t][a][b]); }
        V->stemE[t * 25 + a * 5 + b] = e + tau_w;
        V->stemM[t * 25 + a * 5 + b] = m + tau_w;
    }
    V->ml_close = w(T.ml_closing + T.ml_intern);
    V->mli = w(T.ml_intern);
    V->mlb = w(T.ml_base);
    // special hairpins (letters: closing 5' letter, loop letters, closing 3' letter).  Bonus files (1.x, BL*): the energy is
    // added to the plain one.  v2.0 files: the energy replaces it, so the table holds the difference to the plain energy
    // of that very loop (hairpin[u] + mismatchH, or hairpin[3] + TerminalAU for a triloop)
    auto encode = [&](const std::string& sq, int len, int* code, int* type, int* first, int* last) {
        if ((int)sq.size() != len) return false;
        int c = 0;
        for (char ch : sq) { const int q = letter_code4(ch); if (q < 0) return false; c = c * 4 + q; }
        *code = c;
        *first = letter_code4(sq[1]) + 1; *last = letter_code4(sq[len - 2]) + 1;
        *type = PT[letter_code4(sq[0]) + 1][letter_code4(sq[len - 1]) + 1];
        return true;
    };
    auto special = [&](int E, int u, int type, int first, int last) {
        if (!T.special_total) return w(E);
        if (!type) return 0.0;
        const double plain = u == 3 ? w(T.hairpin[3]) + (type > 2 ? w(T.tau) : 0.0) : w(T.hairpin[u]) + w(T.mmH[type][first][last]);
        return w(E) - plain;
    };
    int code, type, first, last;
    for (auto& t : T.tetra) if (encode(t.first, 6, &code, &type, &first, &last)) V->tetra[code] = special(t.second, 4, type, first, last);
    for (auto& t : T.tri) if (encode(t.first, 5, &code, &type, &first, &last)) V->tri[code] = special(t.second, 3, type, first, last);
    for (auto& t : T.hexa)
        if (V->nhexa < 40 && encode(t.first, 8, &code, &type, &first, &last)) {
            V->hexa_code[V->nhexa] = code;
            V->hexa[V->nhexa++] = special(t.second, 6, type, first, last);
        }
}

}  // namespace

bool load_vienna_dx_ex(const char* defaults_file, bool use_bl, const char* bl_path, const char* param_file, int semantics, ViennaDx* V,
                       char* err, int errlen)
{
    ViennaTables* T = new ViennaTables;
    default_tables(T);
    bool ok = true;
    if (defaults_file) ok = read_any(defaults_file, T, err, errlen);
    if (ok && use_bl) ok = read_any(bl_path, T, err, errlen);
    const bool v20_before = T->v20;
    if (ok && param_file) ok = read_any(param_file, T, err, errlen);
    (void)v20_before;
    if (ok && T->provided != (1u << kNCore) - 1) {   // a truncated or edited file must not leave a table at zero energy silently
        std::string missing;
        for (int k = 0; k < kNCore; k++) if (!(T->provided >> k & 1)) missing += std::string(missing.empty() ? "" : ", ") + kCore[k];
        snprintf(err, errlen, "Vienna parameter tables missing from every source (defaults, BL*, -P): %s", missing.c_str());
        ok = false;
    }
    if (ok) {
        if (semantics != 0 && semantics != kViennaSem18 && semantics != kViennaSem20) {
            snprintf(err, errlen, "unknown Vienna semantics %d", semantics);
            ok = false;
        } else {
            build_vienna_dx(*T, semantics ? semantics : (T->v20 ? kViennaSem20 : kViennaSem18), V);
        }
    }
    delete T;
    return ok;
}

bool load_vienna_dx(const char* path, ViennaDx* V, char* err, int errlen)
{
    return load_vienna_dx_ex(nullptr, false, nullptr, path, 0, V, err, errlen);
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

// Host-only inspection hook for tests/test_vienna_par.py (no GPU involved): one entry of the ViennaDx the loader builds from
// (defaults_file, use_bl, param_file, semantics), as the log Boltzmann weight the kernels read.  table:
//   10 mmI  11 mmH  12 mm1nI  13 mm23I [t=i][a=j][b=k]     14 dxE [t=i][a=j][b=k], a, b in 0..5      15 stemE  16 stemM [t][a][b]
//   17 shape score (l1=i, l2=j)   18 shape kind (l1=i, l2=j)   19 tetra[code=i]   20 tri[code=i]   21 hexaloop term of code i (0: none)
//   22 semantics   23 scalars: i = 0 tau, 1 duplex_init, 2 ml_close, 3 mli, 4 mlb, 5 lxc, 6 hairpin[j], 7 bulge1[j*8+k]
//   24 stack[i][j]   25 int11[i][j][k][l]   26 dangle5[t=i][a=j]   27 dangle3[t=i][a=j]
extern "C" int rh_debug_vienna_value(const char* defaults_file, int use_bl, const char* bl_path, const char* param_file, int semantics,
                                     int table, int i, int j, int k, int l, double* out)
{
    if (!out) return -1;
    static rh::ViennaDx* V = nullptr;   // the model of the last argument set is kept: tests read many entries of one model
    static std::string loaded;
    const std::string key = std::string(defaults_file ? defaults_file : "") + "|" + (use_bl ? "1" : "0") + "|" + (bl_path ? bl_path : "") + "|" +
                            (param_file ? param_file : "") + "|" + std::to_string(semantics);
    if (!V || key != loaded) {
        delete V;
        V = new rh::ViennaDx;
        loaded.clear();
        char err[256];
        if (!rh::load_vienna_dx_ex(defaults_file, use_bl != 0, bl_path, param_file, semantics, V, err, sizeof err)) { delete V; V = nullptr; return -4; }
        loaded = key;
    }
    auto in = [](int v, int hi) { return v >= 0 && v < hi; };
    int rc = 0;
    switch (table) {
        case 10: case 11: case 12: case 13: case 15: case 16:
            if (!in(i, 8) || !in(j, 5) || !in(k, 5)) { rc = -1; break; }
            *out = (table == 10 ? V->mmI : table == 11 ? V->mmH : table == 12 ? V->mm1nI : table == 13 ? V->mm23I : table == 15 ? V->stemE : V->stemM)[i * 25 + j * 5 + k];
            break;
        case 14: if (!in(i, 8) || !in(j, 6) || !in(k, 6)) { rc = -1; break; } *out = V->dxE[i * 36 + j * 6 + k]; break;
        case 17: case 18: {
            if (!in(i, 31) || !in(j, 31) || i + j > 30) { rc = -1; break; }
            const int idx = i * 31 - i * (i - 1) / 2 + j;
            *out = table == 17 ? V->shape[idx].score : (double)V->kind[idx];
            break;
        }
        case 19: if (!in(i, 4096)) { rc = -1; break; } *out = V->tetra[i]; break;
        case 20: if (!in(i, 1024)) { rc = -1; break; } *out = V->tri[i]; break;
        case 21: *out = 0.0; for (int q = 0; q < V->nhexa; q++) if (V->hexa_code[q] == i) *out = V->hexa[q]; break;
        case 22: *out = (double)V->semantics; break;
        case 23:
            switch (i) {
                case 0: *out = V->tau; break;
                case 1: *out = V->duplex_init; break;
                case 2: *out = V->ml_close; break;
                case 3: *out = V->mli; break;
                case 4: *out = V->mlb; break;
                case 5: *out = V->lxc; break;
                case 6: if (!in(j, 31)) rc = -1; else *out = V->hairpin[j]; break;
                case 7: if (!in(j, 8) || !in(k, 8)) rc = -1; else *out = V->bulge1[j * 8 + k]; break;
                default: rc = -1;
            }
            break;
        case 24: if (!in(i, 8) || !in(j, 8)) { rc = -1; break; } *out = V->stack[i * 8 + j]; break;
        case 25: if (!in(i, 8) || !in(j, 8) || !in(k, 5) || !in(l, 5)) { rc = -1; break; } *out = V->int11[(i * 8 + j) * 25 + k * 5 + l]; break;
        case 26: case 27: if (!in(i, 8) || !in(j, 5)) { rc = -1; break; } *out = (table == 26 ? V->dangle5 : V->dangle3)[i * 5 + j]; break;
        default: rc = -1;
    }
    return rc;
}
