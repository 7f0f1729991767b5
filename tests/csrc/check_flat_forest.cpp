// The device layout of a forest (csrc/forest.h: blocked and chained variants) walked on the HOST against the model it
// was flattened from: every tree gives the model's leaf for random feature vectors, the chain links of a chained forest
// lead through the trees t, t + chain, ... to the resting leaf, and the resting leaf points at itself.  Host only
// (ASan / UBSan in tests/test_forest_fuzz.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "forest.h"

static unsigned long long st = 88172645463325252ull;
static unsigned long long rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static float unit() { return (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

// a random binary tree of about `target` nodes appended to m; returns its root
static int grow(kpl::ForestModel &m, int target, int max_depth, bool fractional) {
    struct Open { int node, depth; };
    std::vector<Open> open;
    auto add = [&]() { m.var.push_back(-1); m.thr.push_back(0.f); m.left.push_back(-1); m.right.push_back(-1); m.value.push_back(0.0); return (int)m.var.size() - 1; };
    const int root = add();
    open.push_back({root, 1});
    int count = 1;
    while (!open.empty()) {
        const size_t pick = rnd() % open.size();
        const Open o = open[pick];
        open[pick] = open.back();
        open.pop_back();
        const bool split = count + 2 <= target && o.depth < max_depth && (count < 3 || rnd() % 8 != 0);
        if (!split) {
            m.value[o.node] = fractional ? (double)(float)(unit() * 3.0f - 1.5f) : (double)(rnd() % 3);
            continue;
        }
        m.var[o.node] = (int)(rnd() % m.var_count);
        m.thr[o.node] = unit();
        const int l = add(), r = add();
        m.left[o.node] = l;
        m.right[o.node] = r;
        open.push_back({l, o.depth + 1});
        open.push_back({r, o.depth + 1});
        count += 2;
    }
    return root;
}

static double model_leaf(const kpl::ForestModel &m, int t, const std::vector<float> &x, int &depth) {
    int n = m.root[t];
    depth = 1;
    while (m.var[n] >= 0) { n = x[m.var[n]] <= m.thr[n] ? m.left[n] : m.right[n]; ++depth; }
    return m.value[n];
}

// the walk the kernels do on the flat layout: returns the slot of the leaf
static uint32_t flat_leaf(const kpl::FlatForest &f, int t, const std::vector<float> &x, int &depth) {
    const uint32_t line0 = (f.ntop + 15u) & ~15u;
    uint32_t nd = (uint32_t)t;
    depth = 1;
    for (;;) {
        const kpl::FlatNode n = f.nodes[nd];
        const uint32_t var = n.y >> 24;
        if (var == kpl::kLeafVar) return nd;
        float thr;
        memcpy(&thr, &n.x, 4);
        const uint32_t child = n.y & 0x00ffffffu;
        // siblings: adjacent, except where the children start blocks of their own (blocked layout only)
        const uint32_t stride = (f.chain == 0 && child >= f.ntop && ((child - line0) & 15u) == 0u) ? kpl::kBlockSlots : 1u;
        uint32_t next;
        if (f.chain == 0 && nd >= f.ntop && ((nd - line0) % kpl::kBlockSlots) < 3) {
            const uint32_t base = nd - (nd - line0) % kpl::kBlockSlots, sl = nd - base;       // inside a block: no child index needed
            next = base + 1 + 2 * sl + (x[var] <= thr ? 0u : 1u);
            if (next != child + (x[var] <= thr ? 0u : 1u)) { fprintf(stderr, "block slot rule and child index disagree\n"); exit(2); }
        } else {
            next = child + (x[var] <= thr ? 0u : stride);
        }
        if (next >= f.nodes.size()) { fprintf(stderr, "child out of range\n"); exit(2); }
        nd = next;
        ++depth;
    }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    int chained = 0, blocked = 0;
    for (int it = 0; it < rounds; ++it) {
        kpl::ForestModel m;
        const int shape = it % 5;
        m.var_count = shape == 0 ? 80 : shape == 1 ? 32 : shape == 2 ? 30 : shape == 3 ? 255 : 40;
        const int ntrees = shape == 2 ? 3 + (int)(rnd() % 60) : shape == 4 ? 1 + (int)(rnd() % 39) : 40 + (int)(rnd() % 70);
        const bool fractional = it % 7 == 6;
        const int target = shape == 3 ? 40 : 20 + (int)(rnd() % 1500), max_depth = 2 + (int)(rnd() % 24);
        for (int t = 0; t < ntrees; ++t) m.root.push_back(grow(m, target, max_depth, fractional));
        kpl::FlatForest f;
        std::string err;
        if (!kpl::flatten_forest(m, f, err)) { fprintf(stderr, "flatten: %s\n", err.c_str()); return 1; }
        const bool want_chain = !fractional && ntrees >= kpl::kChainMinTrees && m.var_count >= kpl::kChainMinVars;
        if ((f.chain != 0) != want_chain || (f.chain != 0 && f.chain != (int)kpl::kChainStride)) { fprintf(stderr, "chain = %d, expected %d\n", f.chain, want_chain); return 1; }
        if (f.order_free == fractional) { fprintf(stderr, "order_free wrong\n"); return 1; }
        (f.chain ? chained : blocked)++;
        if (f.chain && f.ntop != f.nodes.size()) { fprintf(stderr, "chained forest with blocks\n"); return 1; }
        // the resting leaf: slot ntrees, value 0, points at itself
        const kpl::FlatNode rest = f.nodes[ntrees];
        if (rest.y != ((kpl::kLeafVar << 24) | (uint32_t)ntrees) || rest.x != 0u) { fprintf(stderr, "resting leaf\n"); return 1; }
        std::vector<float> x(m.var_count);
        for (int q = 0; q < 40; ++q) {
            for (auto &v : x) v = unit();
            for (int t = 0; t < ntrees; ++t) {
                int dm, df;
                const double want = model_leaf(m, t, x, dm);
                const uint32_t leaf = flat_leaf(f, t, x, df);
                float got;
                memcpy(&got, &f.nodes[leaf].x, 4);
                if ((double)got != want || dm != df) { fprintf(stderr, "tree %d: leaf %g depth %d, model %g depth %d\n", t, got, df, want, dm); return 1; }
                const uint32_t link = f.nodes[leaf].y & 0x00ffffffu;
                const uint32_t expect = f.chain ? (uint32_t)(t + f.chain < ntrees ? t + f.chain : ntrees) : 0u;
                if (link != expect) { fprintf(stderr, "tree %d: chain link %u, expected %u\n", t, link, expect); return 1; }
            }
        }
        if (f.max_depth > max_depth) { fprintf(stderr, "max_depth\n"); return 1; }
    }
    printf("chained %d blocked %d\n", chained, blocked);
    return 0;
}
