// mutation fuzz of the forest reader under ASan/UBSan (host only)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "forest.h"
int main(int argc, char **argv) {
    std::string raw, err;
    if (!kpl::read_maybe_gzip(argv[1], raw, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    unsigned long long st = 1234567;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    int ok = 0, rej = 0;
    const int iters = atoi(argv[2]);
    for (int it = 0; it < iters; ++it) {
        std::string t = raw;
        const int kind = (int)(rnd() % 4);
        if (kind == 0) t.resize(rnd() % (t.size() + 1));                       // truncation
        else if (kind == 1) for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) t[rnd() % t.size()] = (char)(rnd() & 0xff);   // byte flips
        else if (kind == 2) { size_t a = rnd() % t.size(), n = rnd() % 64; t.erase(a, n); }                        // deletion
        else { size_t a = rnd() % t.size(); t.insert(a, t.substr(rnd() % t.size(), rnd() % 64)); }                 // duplication
        kpl::ForestModel m; kpl::FlatForest f;
        if (kpl::parse_forest_yaml(t.data(), t.size(), m, err) && kpl::flatten_forest(m, f, err)) ++ok; else ++rej;
    }
    printf("accepted %d rejected %d\n", ok, rej);
    return 0;
}
