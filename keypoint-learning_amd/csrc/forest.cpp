// forest.cpp -- OpenCV-YAML(.gz) random forest reader + device flattening (see forest.h).
//
// The dialect handled is cv::FileStorage's YAML 1.0 as written by OpenCV 3.x
// DTreesImpl::write (what /root/reference/src/main_train_detector.cpp:512 `forest_->save`
// produces and hpp:162 `cv::ml::RTrees::load` consumes), restated from memory of OpenCV 3.2 --
// no sample file survives in the reference checkout, so the dialect is "parity unpinned":
//   %YAML:1.0 directive, optional ---, ONE top-level mapping whose first value is the model;
//   block mappings / sequences by indentation, flow maps `{ var:3, quality:1., le:5.e-01 }`
//   (no space after the colon), flow lists wrapped over several lines, `!!opencv-matrix` tags,
//   floats spelled `1.`, `.5`, `.Inf`, `-.Inf`, `.Nan`; legacy (OpenCV 2.4 CvRTrees) files
//   carry extra per-node keys which are ignored.
// Trees are rebuilt from the pre-order node list exactly as DTreesImpl::readTree does: a node
// with `splits` becomes the current parent, a new node fills the parent's left slot first,
// then its right; after a leaf, climb while the parent's right slot is filled.
#include "forest.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <algorithm>
#include <memory>

namespace kpl {
namespace {

struct YNode {
    enum Type { Null, Scalar, Map, Seq } type = Null;
    std::string s;
    std::vector<std::pair<std::string, YNode>> map;
    std::vector<YNode> seq;

    const YNode *get(const char *key) const {
        if (type != Map) return nullptr;
        for (auto &kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool number(double &v) const {
        if (type != Scalar || s.empty()) return false;
        const char *c = s.c_str();
        bool neg = false;
        if (*c == '-') { neg = true; ++c; } else if (*c == '+') { ++c; }
        if (*c == '.' && (c[1] == 'I' || c[1] == 'i') ) { v = neg ? -INFINITY : INFINITY; return true; }
        if (*c == '.' && (c[1] == 'N' || c[1] == 'n') ) { v = NAN; return true; }
        char *end = nullptr;
        v = strtod(s.c_str(), &end);
        return end && end != s.c_str() && *end == '\0';
    }
    bool integer(long &v) const {
        double d;
        if (!number(d) || d != std::floor(d) || std::fabs(d) > 2.0e9) return false;
        v = (long)d;
        return true;
    }
};

struct Line {
    int indent;
    const char *b, *e;  // trimmed content
    bool dash() const { return b < e && *b == '-' && (e - b == 1 || b[1] == ' '); }
};

static std::string trim(const char *b, const char *e) {
    while (b < e && (*b == ' ' || *b == '\t' || *b == '\r')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
    return std::string(b, e);
}

static std::string unquote(std::string t) {
    if (t.size() >= 2 && (t.front() == '"' || t.front() == '\'') && t.back() == t.front())
        return t.substr(1, t.size() - 2);
    return t;
}

// flow collections --------------------------------------------------------------------------
struct Flow {
    const std::string &s;
    size_t i = 0;
    std::string &err;
    Flow(const std::string &str, std::string &e) : s(str), err(e) {}
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }
    std::string token(const char *stops) {
        ws();
        size_t j = i;
        if (i < s.size() && (s[i] == '"' || s[i] == '\'')) {
            size_t k = s.find(s[i], i + 1);
            if (k == std::string::npos) k = s.size() - 1;
            i = k + 1;
            return s.substr(j, i - j);
        }
        while (i < s.size() && !strchr(stops, s[i])) ++i;
        return trim(s.data() + j, s.data() + i);
    }
    bool value(YNode &out, int depth = 0) {
        ws();
        if (i >= s.size() || depth > 64) { err = "truncated flow collection"; return false; }
        if (s[i] == '[') {
            out.type = YNode::Seq;
            ++i;
            for (;;) {
                ws();
                if (i >= s.size()) { err = "unterminated '['"; return false; }
                if (s[i] == ']') { ++i; return true; }
                out.seq.emplace_back();
                if (!value(out.seq.back(), depth + 1)) return false;
                ws();
                if (i < s.size() && s[i] == ',') ++i;
            }
        }
        if (s[i] == '{') {
            out.type = YNode::Map;
            ++i;
            for (;;) {
                ws();
                if (i >= s.size()) { err = "unterminated '{'"; return false; }
                if (s[i] == '}') { ++i; return true; }
                std::string key = unquote(token(":,}"));
                ws();
                out.map.emplace_back(key, YNode());
                if (i < s.size() && s[i] == ':') {
                    ++i;
                    if (!value(out.map.back().second, depth + 1)) return false;
                }
                ws();
                if (i < s.size() && s[i] == ',') ++i;
            }
        }
        out.type = YNode::Scalar;
        out.s = unquote(token(",]}"));
        return true;
    }
};

static bool balanced(const std::string &s) {
    int depth = 0;
    char q = 0;
    for (char c : s) {
        if (q) { if (c == q) q = 0; }
        else if (c == '"' || c == '\'') q = c;
        else if (c == '[' || c == '{') ++depth;
        else if (c == ']' || c == '}') --depth;
    }
    return depth <= 0;
}

// block structure ---------------------------------------------------------------------------
struct BlockParser {
    std::vector<Line> lines;
    size_t k = 0;
    std::string err;

    explicit BlockParser(const char *text, size_t len) {
        const char *p = text, *end = text + len;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *le = nl ? nl : end;
            const char *b = p;
            int ind = 0;
            while (b < le && *b == ' ') { ++b; ++ind; }
            const char *e = le;
            while (e > b && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
            bool skip = (b == e) || *b == '#' || *b == '%' ||
                        (e - b == 3 && (!memcmp(b, "---", 3) || !memcmp(b, "...", 3)));
            if (!skip) lines.push_back(Line{ind, b, e});
            p = nl ? nl + 1 : end;
        }
    }
    bool more() const { return k < lines.size(); }

    static std::string strip_tag(std::string rest) {
        if (!rest.empty() && rest[0] == '!') {
            size_t sp = rest.find(' ');
            return sp == std::string::npos ? std::string() : trim(rest.data() + sp, rest.data() + rest.size());
        }
        return rest;
    }
    bool inline_value(std::string rest, YNode &out) {
        if (rest[0] == '[' || rest[0] == '{') {
            while (!balanced(rest)) {
                if (!more()) { err = "unterminated flow collection"; return false; }
                rest.push_back(' ');
                rest.append(lines[k].b, lines[k].e);
                ++k;
            }
            Flow f(rest, err);
            return f.value(out);
        }
        out.type = YNode::Scalar;
        out.s = unquote(rest);
        return true;
    }
    bool block(YNode &out, int depth) {
        if (depth > 64) { err = "nesting too deep"; return false; }
        if (!more()) return true;
        return lines[k].dash() ? seq(lines[k].indent, out, depth) : map(lines[k].indent, out, depth);
    }
    bool after_key(int indent, std::string rest, YNode &out, int depth) {
        rest = strip_tag(rest);
        if (!rest.empty()) return inline_value(rest, out);
        if (!more()) return true;
        if (lines[k].indent > indent) return block(out, depth + 1);
        if (lines[k].indent == indent && lines[k].dash()) return seq(indent, out, depth + 1);
        return true;
    }
    bool map(int indent, YNode &out, int depth) {
        out.type = YNode::Map;
        while (more() && lines[k].indent == indent && !lines[k].dash()) {
            const Line &ln = lines[k];
            const char *colon = (const char *)memchr(ln.b, ':', (size_t)(ln.e - ln.b));
            if (!colon) { err = "expected 'key:' near '" + std::string(ln.b, ln.e) + "'"; return false; }
            std::string key = unquote(trim(ln.b, colon));
            std::string rest = trim(colon + 1, ln.e);
            ++k;
            out.map.emplace_back(key, YNode());
            if (!after_key(indent, rest, out.map.back().second, depth)) return false;
        }
        return true;
    }
    bool seq(int indent, YNode &out, int depth) {
        out.type = YNode::Seq;
        while (more() && lines[k].indent == indent && lines[k].dash()) {
            Line ln = lines[k];
            std::string rest = trim(ln.b + 1, ln.e);
            out.seq.emplace_back();
            YNode &item = out.seq.back();
            if (rest.empty()) {
                ++k;
                if (more() && lines[k].indent > indent)
                    if (!block(item, depth + 1)) return false;
            } else if (rest[0] == '[' || rest[0] == '{') {
                ++k;
                if (!inline_value(rest, item)) return false;
            } else if (rest[0] != '"' && rest[0] != '\'' && rest.find(':') != std::string::npos) {
                // compact "- key: value": re-read the remainder as a mapping further in
                int off = indent + (int)((ln.e - ln.b) - (ptrdiff_t)rest.size());
                lines[k].indent = off;
                lines[k].b = ln.e - rest.size();
                if (!map(off, item, depth + 1)) return false;
            } else {
                ++k;
                item.type = YNode::Scalar;
                item.s = unquote(rest);
            }
        }
        return true;
    }
};

}  // namespace

bool parse_forest_yaml(const char *text, size_t len, ForestModel &out, std::string &err) {
    BlockParser bp(text, len);
    if (bp.lines.empty()) { err = "empty document"; return false; }
    YNode doc;
    if (!bp.block(doc, 0)) { err = "YAML: " + bp.err; return false; }
    if (doc.type != YNode::Map || doc.map.empty()) { err = "YAML: top level is not a mapping"; return false; }
    // cv::Algorithm::load takes the first top-level node whatever its name
    const YNode *model = &doc.map.front().second;
    if (model->type != YNode::Map || !model->get("trees")) {
        // tolerate a document that IS the model (no wrapping key)
        if (doc.get("trees")) model = &doc;
        else { err = "no 'trees' sequence in the model node"; return false; }
    }
    long var_count = 0;
    if (const YNode *vc = model->get("var_count")) vc->integer(var_count);
    if (var_count <= 0)
        if (const YNode *va = model->get("var_all")) { va->integer(var_count); }
    if (var_count <= 0) { err = "missing var_count"; return false; }
    if (const YNode *ic = model->get("is_classifier")) {
        long v = 1;
        if (ic->integer(v) && v == 0) { err = "regression forests are not supported"; return false; }
    }
    const YNode *trees = model->get("trees");
    if (trees->type != YNode::Seq || trees->seq.empty()) { err = "'trees' is empty"; return false; }
    out = ForestModel();
    out.var_count = (int)var_count;
    std::vector<int> parent;
    std::vector<int> inversed;
    for (const YNode &tree : trees->seq) {
        const YNode *nodes = tree.get("nodes");
        if (!nodes || nodes->type != YNode::Seq || nodes->seq.empty()) { err = "tree without 'nodes'"; return false; }
        const int base = (int)out.var.size();
        out.root.push_back(base);
        int pidx = -1;
        for (const YNode &nd : nodes->seq) {
            if (nd.type != YNode::Map) { err = "node is not a mapping"; return false; }
            const int nidx = (int)out.var.size();
            double value = 0.0;
            if (const YNode *v = nd.get("value")) {
                if (!v->number(value)) { err = "bad node value"; return false; }
            }
            out.value.push_back(value);
            out.left.push_back(-1);
            out.right.push_back(-1);
            parent.push_back(pidx);
            const YNode *splits = nd.get("splits");
            bool has_split = splits && splits->type == YNode::Seq && !splits->seq.empty();
            if (has_split) {
                const YNode &sp = splits->seq.front();   // primary split only (no surrogates)
                long var = -1;
                const YNode *v = sp.get("var");
                if (!v || !v->integer(var) || var < 0) { err = "split without a valid 'var'"; return false; }
                const YNode *c = sp.get("le");
                bool inv = false;
                if (!c) { c = sp.get("gt"); inv = c != nullptr; }
                if (!c) { err = "categorical ('in'/'not_in') splits are not supported"; return false; }
                double thr;
                if (!c->number(thr)) { err = "bad split threshold"; return false; }
                out.var.push_back((int)var);
                out.thr.push_back((float)thr);
                if (inv) inversed.push_back(nidx);
            } else {
                out.var.push_back(-1);
                out.thr.push_back(0.0f);
            }
            if (pidx >= 0) {
                if (out.left[pidx] < 0) out.left[pidx] = nidx;
                else if (out.right[pidx] < 0) out.right[pidx] = nidx;
                else { err = "node list is not a pre-order binary tree"; return false; }
            } else if (nidx != base) {
                err = "tree has nodes after it is complete";
                return false;
            }
            if (has_split) pidx = nidx;
            else while (pidx >= 0 && out.right[pidx] >= 0) pidx = parent[pidx];
        }
        if (pidx >= 0) { err = "tree ends with an unfilled split node"; return false; }
    }
    for (int nd : inversed) std::swap(out.left[nd], out.right[nd]);   // 'gt': children swap roles
    if (const YNode *nt = model->get("ntrees")) {
        long v;
        if (nt->integer(v) && v != (long)out.root.size()) { err = "ntrees does not match 'trees'"; return false; }
    }
    return true;
}

bool inflate_if_gzip(const void *data, size_t len, std::string &out, std::string &err) {
    const unsigned char *p = (const unsigned char *)data;
    if (len < 2 || p[0] != 0x1f || p[1] != 0x8b) {
        out.assign((const char *)data, len);
        return true;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) { err = "zlib init failed"; return false; }
    zs.next_in = (Bytef *)p;
    zs.avail_in = (uInt)len;
    out.clear();
    std::vector<char> buf(1 << 20);
    int rc;
    do {
        zs.next_out = (Bytef *)buf.data();
        zs.avail_out = (uInt)buf.size();
        rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); err = "corrupt gzip stream"; return false; }
        out.append(buf.data(), buf.size() - zs.avail_out);
    } while (rc != Z_STREAM_END);
    inflateEnd(&zs);
    return true;
}

bool read_maybe_gzip(const char *path, std::string &out, std::string &err) {
    FILE *f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::string raw;
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) raw.append(buf, got);
    fclose(f);
    return inflate_if_gzip(raw.data(), raw.size(), out, err);
}

bool flatten_forest(const ForestModel &m, FlatForest &out, std::string &err) {
    const int64_t nn = m.nnodes();
    if (m.ntrees() <= 0 || nn <= 0) { err = "forest has no trees"; return false; }
    if (m.var_count <= 0 || m.var_count > (int)kLeafVar) { err = "var_count must be in 1..255"; return false; }
    if ((int64_t)m.thr.size() != nn || (int64_t)m.left.size() != nn || (int64_t)m.right.size() != nn ||
        (int64_t)m.value.size() != nn) { err = "node arrays differ in length"; return false; }
    if (nn >= (int64_t)kMaxFlatNodes) { err = "forest has more than 2^24 nodes"; return false; }
    out = FlatForest();
    out.ntrees = m.ntrees();
    out.var_count = m.var_count;
    out.nodes.reserve((size_t)nn);
    out.order_free = m.ntrees() <= (1 << 15);
    for (int64_t i = 0; i < nn && out.order_free; ++i)
        if (m.var[i] < 0 && (!(std::fabs(m.value[i]) <= 32768.0) || m.value[i] != std::floor(m.value[i]))) out.order_free = false;
    // the forests the chained kernel takes (kernels.hip): level-major throughout, leaf records linked
    out.chain = out.order_free && m.ntrees() >= kChainMinTrees && m.var_count >= kChainMinVars ? (int)kChainStride : 0;
    std::vector<char> seen((size_t)nn, 0);
    // Top part: ONE breadth-first pass over the whole forest, level by level: the roots of all trees
    // first (root of tree t = node t), then the second level of all trees, and so on, while the whole
    // next level still fits kTopNodes slots.  A node's record is written when it is visited; its two
    // children get adjacent slots reserved at that moment.
    struct Item { int src; uint32_t dst; int depth; int tree; };
    auto emit = [&](const Item &it, uint32_t lpos, std::string &e) -> int {   // 1 = internal, 0 = leaf, -1 = error
        if (seen[it.src]) { e = "node reachable twice (not a tree)"; return -1; }
        seen[it.src] = 1;
        if (it.depth > out.max_depth) out.max_depth = it.depth;
        FlatNode fn;
        if (m.var[it.src] < 0) {
            const float v = (float)m.value[it.src];
            // (double)v != value also rejects NaN; +-Inf leaves would make the tree sum Inf - Inf = NaN
            if (!std::isfinite(m.value[it.src]) || (double)v != m.value[it.src]) { e = "leaf value is not a finite float"; return -1; }
            memcpy(&fn.x, &v, 4);
            // chained: a leaf points at the root of the tree that follows its own in the chain (forest.h), or at the resting leaf
            fn.y = (kLeafVar << 24) | (out.chain ? (uint32_t)std::min(it.tree + out.chain, m.ntrees()) : 0u);
            out.nodes[it.dst] = fn;
            return 0;
        }
        const int l = m.left[it.src], rr = m.right[it.src];
        if (m.var[it.src] >= m.var_count) { e = "split variable >= var_count"; return -1; }
        if (l < 0 || l >= nn || rr < 0 || rr >= nn) { e = "split node without two children"; return -1; }
        if (!std::isfinite(m.thr[it.src])) { e = "non-finite split threshold"; return -1; }
        memcpy(&fn.x, &m.thr[it.src], 4);
        fn.y = ((uint32_t)m.var[it.src] << 24) | lpos;
        out.nodes[it.dst] = fn;
        return 1;
    };
    out.nnodes = nn;
    std::vector<Item> level, next;
    for (int t = 0; t < m.ntrees(); ++t) {
        const int r = m.root[t];
        if (r < 0 || r >= nn) { err = "root index out of range"; return false; }
        level.push_back(Item{r, (uint32_t)out.nodes.size(), 1, t});      // the root of tree t is slot t
        out.nodes.push_back(FlatNode{0, 0});
    }
    // the resting leaf (value 0) right behind the roots: where a walk without a tree sits (kernels.hip)
    {
        FlatNode rest;
        const float zero = 0.0f;
        memcpy(&rest.x, &zero, 4);
        rest.y = (kLeafVar << 24) | (uint32_t)m.ntrees();      // it follows itself
        out.nodes.push_back(rest);
    }
    std::vector<Item> pairs;          // internal nodes of the last top level: their children start the blocked part
    while (!level.empty()) {
        size_t internal = 0;
        for (const Item &it : level) internal += m.var[it.src] >= 0 ? 1 : 0;
        const bool top = out.chain != 0 || out.nodes.size() + 2 * internal <= (size_t)kTopNodes;
        next.clear();
        for (const Item &it : level) {
            if (m.var[it.src] >= 0 && !top) {       // its children start a block: slot known only then
                pairs.push_back(it);
                continue;
            }
            const uint32_t lpos = (uint32_t)out.nodes.size();
            const int kind = emit(it, lpos, err);
            if (kind < 0) return false;
            if (kind == 1) {
                out.nodes.push_back(FlatNode{0, 0});
                out.nodes.push_back(FlatNode{0, 0});
                next.push_back(Item{m.left[it.src], lpos, it.depth + 1, it.tree});
                next.push_back(Item{m.right[it.src], lpos + 1, it.depth + 1, it.tree});
            }
        }
        if (!top) break;
        level.swap(next);
    }
    out.ntop = (uint32_t)out.nodes.size();
    // Blocked part: the two children of every pending internal node get one 128-byte line: two blocks of
    // kBlockSlots = 8 slots, the left child's at `base`, the right child's at base + 8.  A block holds a
    // subtree of three levels: slot 0 its root, slots 1 2 the root's children, slots 3 4 the children of
    // slot 1, slots 5 6 those of slot 2 (slot 7 stays empty).  Inside a block the walk needs no child
    // index; a node of the third level points at the line of ITS children, as the pending node does.
    while (!pairs.empty()) {
        const Item parent = pairs.back();
        pairs.pop_back();
        size_t base = out.nodes.size();
        base = (base + 2 * kBlockSlots - 1) / (2 * kBlockSlots) * (2 * kBlockSlots);
        if (base + 2 * kBlockSlots > (size_t)kMaxFlatNodes) { err = "forest needs more than 2^24 node slots"; return false; }
        out.nodes.resize(base + 2 * kBlockSlots, FlatNode{0, 0});
        if (emit(parent, (uint32_t)base, err) < 0) return false;                  // the parent's record points at the left block
        for (uint32_t side = 0; side < 2; ++side) {
            const uint32_t b = (uint32_t)base + side * kBlockSlots;
            int src[kBlockSlots];
            int depth[kBlockSlots];
            for (uint32_t k = 0; k < kBlockSlots; ++k) src[k] = -1;
            src[0] = side == 0 ? m.left[parent.src] : m.right[parent.src];
            depth[0] = parent.depth + 1;
            for (uint32_t sl = 0; sl < 7; ++sl) {
                if (src[sl] < 0) continue;
                const Item it{src[sl], b + sl, depth[sl], parent.tree};
                if (sl < 3) {
                    const uint32_t c = 1 + 2 * sl;                                // children of slot sl: slots 1 + 2 sl, 2 + 2 sl
                    const int kind = emit(it, b + c, err);
                    if (kind < 0) return false;
                    if (kind == 1) {
                        src[c] = m.left[it.src];
                        src[c + 1] = m.right[it.src];
                        depth[c] = depth[c + 1] = it.depth + 1;
                    }
                } else if (m.var[it.src] >= 0) {
                    pairs.push_back(it);                 // emitted when the line of its children is placed
                } else if (emit(it, 0u, err) < 0) {
                    return false;
                }
            }
        }
    }
    return true;
}

}  // namespace kpl
