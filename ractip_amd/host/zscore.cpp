// zscore.cpp -- host side of the z-score shard (include/ractip_zscore.h).
//
// A restatement, specialised to k = 2, of what uShuffle::shuffle does
// (/root/reference/src/ushuffle.c:127-268) with the SAME sequence of random() calls, so
// that the shuffled pairs -- and therefore every DP input -- are those of the reference
// for a given --seed (checked in tests/test_zscore.py against rows produced by the reference's
// own ushuffle.c).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ractip_zscore.h"

namespace {

struct Node {               // one distinct letter of the sequence
    std::vector<int> out;   // successor nodes, one per dinucleotide starting here, in sequence order
    int next = 0;           // Wilson: chosen exit edge
    bool in_tree = false;
    char letter = 0;
};

void permute(int* v, int n)  // ushuffle.c:198-207
{
    for (int i = n - 1; i > 0; i--) {
        const int j = (int)(random() % (i + 1));
        const int tmp = v[i]; v[i] = v[j]; v[j] = tmp;
    }
}

}  // namespace

extern "C" void rh_dishuffle(const char* s, char* t, int l)
{
    if (l <= 2) {  // k >= l: exact copy, no random numbers drawn (ushuffle.c:214-217)
        std::memcpy(t, s, (size_t)l);
        return;
    }
    // nodes in order of first appearance (the hash table of ushuffle.c:95-110 assigns ids that way)
    int id_of[256];
    for (int& v : id_of) v = -1;
    std::vector<Node> g;
    std::vector<int> node_at(l);
    for (int i = 0; i < l; i++) {
        const unsigned char c = (unsigned char)s[i];
        if (id_of[c] < 0) { id_of[c] = (int)g.size(); g.emplace_back(); g.back().letter = s[i]; }
        node_at[i] = id_of[c];
    }
    for (int i = 0; i + 1 < l; i++) g[node_at[i]].out.push_back(node_at[i + 1]);
    const int root = node_at[l - 1];
    const int nv = (int)g.size();

    // Wilson's algorithm: a uniformly random arborescence into the root (ushuffle.c:227-241)
    g[root].in_tree = true;
    for (int i = 0; i < nv; i++) {
        int u = i;
        while (!g[u].in_tree) {
            g[u].next = (int)(random() % (long)g[u].out.size());
            u = g[u].out[g[u].next];
        }
        u = i;
        while (!g[u].in_tree) {
            g[u].in_tree = true;
            u = g[u].out[g[u].next];
        }
    }
    // the tree edge leaves last; all other edges in random order (ushuffle.c:243-254)
    for (int i = 0; i < nv; i++) {
        Node& u = g[i];
        const int n = (int)u.out.size();
        if (i != root) {
            const int last = u.out[n - 1];
            u.out[n - 1] = u.out[u.next];
            u.out[u.next] = last;
            permute(u.out.data(), n - 1);
        } else {
            permute(u.out.data(), n);
        }
    }
    // Euler walk from the first letter (ushuffle.c:256-267)
    std::vector<int> used(nv, 0);
    t[0] = s[0];
    int u = 0, pos = 1;
    while (used[u] < (int)g[u].out.size()) {
        const int v = g[u].out[used[u]++];
        t[pos++] = g[v].letter;
        u = v;
    }
}

extern "C" int rh_zscore_shuffles(const char* s1, const char* s2, int mode, int num, unsigned seed, char* out1, char* out2)
{
    if (mode != 1 && mode != 2 && mode != 12) return -1;
    const int n1 = (int)std::strlen(s1), n2 = (int)std::strlen(s2);
    std::vector<char> a(s1, s1 + n1), b(s2, s2 + n2);  // the reference starts from copies of the natives
    srandom(seed);                                      // src/ractip.cpp:1636
    for (int it = 0; it < num; it++) {
        if (mode == 1 || mode == 12) rh_dishuffle(s1, a.data(), n1);
        if (mode == 2 || mode == 12) rh_dishuffle(s2, b.data(), n2);
        std::memcpy(out1 + (size_t)it * (n1 + 1), a.data(), n1); out1[(size_t)it * (n1 + 1) + n1] = 0;
        std::memcpy(out2 + (size_t)it * (n2 + 1), b.data(), n2); out2[(size_t)it * (n2 + 1) + n2] = 0;
    }
    return 0;
}

extern "C" float rh_zscore_from_energies(const float* ee, int num, float e_native)
{
    float sum = 0.0f, sum2 = 0.0f;                      // src/ractip.cpp:1626, 1655
    for (int i = 0; i < num; i++) { sum += ee[i]; sum2 += ee[i] * ee[i]; }
    const float m = sum / num;
    float v = sum2 / num - m * m;
    if (v < 0.0f) v = 0.0f;
    return (e_native - m) / std::sqrt(v);               // :1667-1669
}
