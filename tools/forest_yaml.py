"""OpenCV-3.x `cv::ml::RTrees` YAML(.gz) reader and writer in pure Python.

Tooling / test-side code: it generates the forest fixtures the product (libkpl's own C++ reader,
keypoint-learning_amd/csrc/forest_yaml.cpp) has to load, and it is the independent second
implementation the C++ reader is checked against.  The format is restated from memory of
OpenCV 3.2 `DTreesImpl::write/read` (the reference only calls `cv::ml::RTrees::load`,
/root/reference/include/impl/KeypointLearning.hpp:162, and `forest_->save`,
/root/reference/src/main_train_detector.cpp:512); no sample file survives in the reference
checkout (.MISSING_LARGE_BLOBS), so the dialect is "parity unpinned".
"""
import gzip
import io
import math

import numpy as np


# ------------------------------------------------------------------------------------------
# tolerant YAML-subset parser (OpenCV FileStorage dialect)
# ------------------------------------------------------------------------------------------
def _scalar(tok):
    t = tok.strip()
    if len(t) >= 2 and t[0] == t[-1] and t[0] in "\"'":
        return t[1:-1]
    low = t.lower()
    if low in (".inf", "+.inf"):
        return math.inf
    if low == "-.inf":
        return -math.inf
    if low == ".nan":
        return math.nan
    try:
        return int(t)
    except ValueError:
        pass
    try:
        return float(t)
    except ValueError:
        return t


class _Flow:
    """Recursive-descent parser of a flow collection held in one string."""

    def __init__(self, s):
        self.s = s
        self.i = 0

    def ws(self):
        while self.i < len(self.s) and self.s[self.i] in " \t\r\n":
            self.i += 1

    def value(self):
        self.ws()
        c = self.s[self.i]
        if c == "[":
            return self.seq()
        if c == "{":
            return self.map()
        return _scalar(self.token(",]}"))

    def token(self, stops):
        self.ws()
        j = self.i
        if j < len(self.s) and self.s[j] in "\"'":
            q = self.s[j]
            k = self.s.index(q, j + 1)
            self.i = k + 1
            return self.s[j:k + 1]
        while self.i < len(self.s) and self.s[self.i] not in stops:
            self.i += 1
        return self.s[j:self.i]

    def seq(self):
        out = []
        self.i += 1
        while True:
            self.ws()
            if self.s[self.i] == "]":
                self.i += 1
                return out
            out.append(self.value())
            self.ws()
            if self.s[self.i] == ",":
                self.i += 1

    def map(self):
        out = {}
        self.i += 1
        while True:
            self.ws()
            if self.s[self.i] == "}":
                self.i += 1
                return out
            key = self.token(":,}").strip()
            self.ws()
            if self.i < len(self.s) and self.s[self.i] == ":":
                self.i += 1
                out[_scalar(key)] = self.value()
            else:
                out[_scalar(key)] = None
            self.ws()
            if self.s[self.i] == ",":
                self.i += 1


def _balanced(s):
    depth = 0
    q = None
    for c in s:
        if q:
            if c == q:
                q = None
        elif c in "\"'":
            q = c
        elif c in "[{":
            depth += 1
        elif c in "]}":
            depth -= 1
    return depth <= 0


class _Block:
    def __init__(self, text):
        self.lines = []
        for raw in text.splitlines():
            line = raw.rstrip()
            st = line.strip()
            if not st or st.startswith("#") or st.startswith("%") or st in ("---", "..."):
                continue
            self.lines.append((len(line) - len(line.lstrip(" ")), st))
        self.k = 0

    def peek(self):
        return self.lines[self.k] if self.k < len(self.lines) else (-1, "")

    def inline(self, rest):
        """value that starts on the current line (already consumed): flow or scalar."""
        if rest[0] in "[{":
            while not _balanced(rest):
                rest += " " + self.lines[self.k][1]
                self.k += 1
            return _Flow(rest).value()
        return _scalar(rest)

    def block(self, indent):
        ind, st = self.peek()
        if st == "-" or st.startswith("- "):
            return self.seq(ind)
        return self.map(ind)

    @staticmethod
    def strip_tag(rest):
        if rest.startswith("!!") or rest.startswith("!"):
            parts = rest.split(None, 1)
            return parts[1].strip() if len(parts) > 1 else ""
        return rest

    def after_key(self, indent, rest):
        rest = self.strip_tag(rest.strip())
        if rest:
            return self.inline(rest)
        ind, st = self.peek()
        if ind > indent:
            return self.block(ind)
        if ind == indent and (st == "-" or st.startswith("- ")):
            return self.seq(ind)
        return None

    def map(self, indent):
        out = {}
        while True:
            ind, st = self.peek()
            if ind != indent or st == "-" or st.startswith("- "):
                return out
            self.k += 1
            key, _, rest = st.partition(":")
            out[_scalar(key)] = self.after_key(indent, rest)

    def seq(self, indent):
        out = []
        while True:
            ind, st = self.peek()
            if ind != indent or not (st == "-" or st.startswith("- ")):
                return out
            rest = st[1:].strip()
            if not rest:
                self.k += 1
                nind, _ = self.peek()
                out.append(self.block(nind) if nind > indent else None)
            elif rest[0] in "[{":
                self.k += 1
                out.append(self.inline(rest))
            elif ":" in rest and not rest[0] in "\"'":
                # compact "- key: value" mapping: re-read the line as a mapping two columns in
                off = indent + (len(st) - len(rest))
                self.lines[self.k] = (off, rest)
                out.append(self.map(off))
            else:
                self.k += 1
                out.append(_scalar(rest))


def parse_yaml(text):
    b = _Block(text)
    if not b.lines:
        return {}
    return b.block(b.lines[0][0])


def read_text(path_or_bytes):
    """Returns the YAML text; gzip is sniffed by magic, never by extension."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        raw = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return raw.decode("utf-8", errors="replace")


# ------------------------------------------------------------------------------------------
# forest model
# ------------------------------------------------------------------------------------------
class ForestArrays:
    """Flat node arrays with global node numbering (the oracle's kplo_forest layout)."""

    def __init__(self, root, var, thr, left, right, value, var_count, depth=None,
                 class_idx=None, quality=None):
        self.root = np.asarray(root, dtype=np.int32)
        self.var = np.asarray(var, dtype=np.int32)
        self.thr = np.asarray(thr, dtype=np.float32)
        self.left = np.asarray(left, dtype=np.int32)
        self.right = np.asarray(right, dtype=np.int32)
        self.value = np.asarray(value, dtype=np.float64)
        self.var_count = int(var_count)
        self.depth = None if depth is None else np.asarray(depth, dtype=np.int32)
        self.class_idx = None if class_idx is None else np.asarray(class_idx, dtype=np.int32)
        self.quality = None if quality is None else np.asarray(quality, dtype=np.float64)

    @property
    def ntrees(self):
        return len(self.root)

    @property
    def nnodes(self):
        return len(self.var)

    def tree_ranges(self):
        """Node ranges per tree; valid when trees are stored one after another (as read/written)."""
        ends = list(self.root[1:]) + [self.nnodes]
        return list(zip(self.root.tolist(), [int(e) for e in ends]))


def forest_from_yaml(text):
    doc = parse_yaml(text)
    if not isinstance(doc, dict) or not doc:
        raise ValueError("not an OpenCV ml YAML document")
    top = next(iter(doc.values()))          # first top-level node, like cv::Algorithm::load
    if not isinstance(top, dict) or "trees" not in top:
        raise ValueError("no 'trees' in the model node")
    var_count = int(top.get("var_count", top.get("var_all", 0)))
    root, var, thr, left, right, value, depth, cidx, qual = ([] for _ in range(9))
    inversed = []
    for tree in top["trees"]:
        nodes = tree["nodes"]
        base = len(var)
        root.append(base)
        parent = {}
        pidx = -1
        for k, nd in enumerate(nodes):
            nidx = base + k
            splits = nd.get("splits")
            depth.append(int(nd.get("depth", 0)))
            value.append(float(nd.get("value", 0.0)))
            cidx.append(int(nd.get("norm_class_idx", 0)))
            left.append(-1)
            right.append(-1)
            if splits:
                sp = splits[0]              # only the primary split is used (no surrogates)
                inv = "gt" in sp
                if "le" not in sp and "gt" not in sp:
                    raise ValueError("categorical splits are not supported")
                var.append(int(sp["var"]))
                thr.append(np.float32(sp["gt"] if inv else sp["le"]))
                qual.append(float(sp.get("quality", 0.0)))
                if inv:
                    inversed.append(nidx)
            else:
                var.append(-1)
                thr.append(np.float32(0))
                qual.append(0.0)
            parent[nidx] = pidx
            if pidx >= 0:
                if left[pidx] < 0:
                    left[pidx] = nidx
                else:
                    right[pidx] = nidx
            if splits:
                pidx = nidx
            else:
                while pidx >= 0 and right[pidx] >= 0:
                    pidx = parent[pidx]
    for nd in inversed:                     # 'gt' = inversed split: children swap roles
        left[nd], right[nd] = right[nd], left[nd]
    ntrees = int(top.get("ntrees", len(root)))
    if ntrees != len(root):
        raise ValueError("ntrees does not match the number of trees")
    return ForestArrays(root, var, thr, left, right, value, var_count, depth, cidx, qual)


def load_forest(path_or_bytes):
    return forest_from_yaml(read_text(path_or_bytes))


# ------------------------------------------------------------------------------------------
# writer (OpenCV 3.x layout: block-style nodes, flow-style split maps, 3-space indent)
# ------------------------------------------------------------------------------------------
def _fmt_float(v, digits):
    v = float(v)
    if math.isnan(v):
        return ".Nan"
    if math.isinf(v):
        return ".Inf" if v > 0 else "-.Inf"
    if v == int(v) and abs(v) < 1e9:
        return "%d." % int(v)
    return ("%." + str(digits) + "e") % v


def fmt_f32(v):
    return _fmt_float(np.float32(v), 8)


def fmt_f64(v):
    return _fmt_float(v, 16)


def _wrap_flow_list(key, items, indent, width=80):
    """`key: [ a, b, ... ]` wrapped across lines like cv::FileStorage does."""
    pad = " " * indent
    lines = []
    cur = pad + key + ": [ "
    for k, it in enumerate(items):
        tok = it + ("," if k + 1 < len(items) else "")
        if len(cur) + len(tok) + 1 > width and cur.strip():
            lines.append(cur.rstrip())
            cur = pad + "    "
        cur += tok + " "
    lines.append(cur + "]")
    return lines


def forest_to_yaml(fa, max_depth=25, min_sample_count=1, nactive_vars=0,
                   top_key="opencv_ml_rtrees", legacy_keys=False):
    """Serialises `fa` (trees must be stored contiguously, pre-order not required)."""
    F = fa.var_count
    out = io.StringIO()
    w = out.write
    w("%YAML:1.0\n---\n")
    w(top_key + ":\n")
    w("   format: 3\n   is_classifier: 1\n")
    w("   var_all: %d\n   var_count: %d\n   ord_var_count: %d\n   cat_var_count: 1\n"
      % (F + 1, F, F))
    w("   training_params:\n      use_surrogates: 0\n      max_categories: 10\n"
      "      regression_accuracy: 0.\n      max_depth: %d\n      min_sample_count: %d\n"
      "      cross_validation_folds: 0\n" % (max_depth, min_sample_count))
    w("      priors: !!opencv-matrix\n         rows: 1\n         cols: 2\n         dt: d\n"
      "         data: [ 1., 1. ]\n")
    w("      nactive_vars: %d\n" % (nactive_vars or max(1, int(math.sqrt(F)))))
    w("   global_var_idx: 1\n")
    for ln in _wrap_flow_list("var_idx", [str(i) for i in range(F)], 3):
        w(ln + "\n")
    for ln in _wrap_flow_list("var_type", ["0"] * F + ["1"], 3):
        w(ln + "\n")
    for ln in _wrap_flow_list("cat_ofs", ["0"] * (2 * F) + ["0", "2"], 3):
        w(ln + "\n")
    w("   class_labels: [ 0, 1 ]\n")
    for ln in _wrap_flow_list("missing_subst", ["0."] * (F + 1), 3):
        w(ln + "\n")
    w("   oob_error: 0.\n")
    w("   ntrees: %d\n" % fa.ntrees)
    w("   trees:\n")
    for t in range(fa.ntrees):
        w("      -\n")
        if legacy_keys:
            w("         best_tree_idx: -1\n")
        w("         nodes:\n")
        # pre-order, left child first
        stack = [(int(fa.root[t]), 0)]
        while stack:
            nd, d = stack.pop()
            w("            -\n")
            w("               depth: %d\n" % d)
            if legacy_keys:
                w("               sample_count: 1\n")
            w("               value: %s\n" % fmt_f64(fa.value[nd]))
            cidx = int(fa.class_idx[nd]) if fa.class_idx is not None else int(fa.value[nd])
            w("               norm_class_idx: %d\n" % cidx)
            if legacy_keys:
                w("               Tn: 0\n               complexity: 0\n               alpha: 0.\n"
                  "               node_risk: 0.\n               tree_risk: 0.\n"
                  "               tree_error: 0.\n")
            if fa.var[nd] >= 0:
                q = float(fa.quality[nd]) if fa.quality is not None else 1.0
                w("               splits:\n")
                w("                  - { var:%d, quality:%s, le:%s }\n"
                  % (int(fa.var[nd]), fmt_f32(q), fmt_f32(fa.thr[nd])))
                stack.append((int(fa.right[nd]), d + 1))
                stack.append((int(fa.left[nd]), d + 1))
    return out.getvalue()


def save_forest(fa, path, gz=None, **kw):
    text = forest_to_yaml(fa, **kw).encode("utf-8")
    if gz is None:
        gz = str(path).endswith(".gz")
    if gz:
        # mtime=0 keeps the fixture byte-stable
        with open(path, "wb") as f:
            with gzip.GzipFile(fileobj=f, mode="wb", mtime=0, filename="") as g:
                g.write(text)
    else:
        with open(path, "wb") as f:
            f.write(text)
