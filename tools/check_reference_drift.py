#!/usr/bin/env python3
"""Drift guard between the reference's own code on the hot path and its restatement in oracle/kpl_oracle.c.

The oracle cannot be compiled against the reference (PCL / OpenCV / Eigen / FLANN are absent and no stand-in headers are
written), so the parts of it that restate the reference's OWN loops -- the feature loop, the row normalisation, the score
formula and the NMS / draws logic -- are tied to the reference text instead:

  1. every "/* :NNN */" line citation of those oracle functions must land on a reference line that still holds the
     construct it cites (a token that must appear in that line, whitespace ignored);
  2. the four bilinear histogram updates must be the SAME expressions token for token once the reference's names are
     mapped onto the oracle's (annulus_weight -> aw, bin_pair -> bp, ...);
  3. the cited reference ranges must still hash to what they hashed to when the oracle was written (any edit of the
     reference inside them is reported with the range, so that the restatement is re-read against it).

It reads /root/reference (build container only) and copies nothing from it: this file holds line numbers, short tokens
and hashes.  `python tools/check_reference_drift.py` prints a report and exits non-zero on drift; `--update-hashes`
prints the current hashes (for a deliberate re-pin after re-reading the restatement).
tests/test_reference_drift.py runs it when /root/reference is present.
"""
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
HPP = "include/impl/KeypointLearning.hpp"
CPP = "src/KeypointLearning.cpp"
ORACLE = os.path.join(ROOT, "oracle", "kpl_oracle.c")

# (reference file, first line, last line, sha256 of the lines with all whitespace removed)
PINNED = [
    (HPP, 116, 156, "initCompute (normal fallbacks)"),
    (HPP, 179, 263, "detectKeypoints"),
    (HPP, 267, 296, "runForest"),
    (HPP, 321, 376, "computePointFeatures"),
    (CPP, 41, 92, "findAnnulusPair / findBinPair"),
]
HASHES = {
    (HPP, 116, 156): "1b1c29a3b76c3fce65acad1a0d7cb47d78dc51a130f58bcc66fe3e87d8734dad",   # initCompute (normal fallbacks)
    (HPP, 179, 263): "4397a7b8d4c81732719f2f35f1a1b3e8bdd54bba36dd4b12c0f14f6abb7198b2",   # detectKeypoints
    (HPP, 267, 296): "5829a53d374d0834076bfe58b9a833fdb46451592af603f28e1f5f17e2ffe5d9",   # runForest
    (HPP, 321, 376): "36c4d414169943d6709fea080b17ab1df7ffe6871b985d3bf674c69ebca68e89",   # computePointFeatures
    (CPP, 41, 92): "1e3c33160857498bea2d88023811dbf923ffbfd835debf6f509d6e55651136c6",   # findAnnulusPair / findBinPair
}

# oracle line citation -> token the cited reference line must contain (whitespace-insensitive)
LINE_TOKENS = {
    # computePointFeatures
    325: "MatrixXf::Zero", 332: "getNormalVector3fMap", 336: "neigh_indx=1", 338: "isFinite<pcl::Normal>",
    342: "1-point_vector.dot(normal_vector)", 345: "findAnnulusPair(this->n_annulus_,sqrt(distances[neigh_indx]),this->search_radius_",
    348: "findBinPair(this->n_bins_,cosine", 350: "histograms(annulus_index,bin_index)+=", 351: "histograms(annulus_index,bin_pair)+=",
    354: "histograms(annulus_pair,bin_index)+=", 355: "histograms(annulus_pair,bin_pair)+=",
    362: "histograms.row(i).norm()>0", 364: "histograms.row(i).normalize()", 368: "(i*this->n_bins_)+k",
    # runForest
    277: "isFinite", 279: "computePointFeatures(pIdx)", 281: "PREDICT_SUM", 287: "1-(sum/(forest_size*1.0f))",
    # initCompute: which estimator, with which settings (oracle: kplo_estimate_normals radius branch, kplo_integral_image_normals)
    130: "!this->surface_->isOrganized()", 135: "setRadiusSearch(this->search_radius_)", 140: "IntegralImageNormalEstimation<PointInT,NormalT>",
    141: "::SIMPLE_3D_GRADIENT", 143: "setNormalSmoothingSize(5.0)",
    # detectKeypoints
    205: "!isFinite(response->points[idx])", 206: "!pcl_isfinite(response->points[idx].intensity)", 207: "intensity<this->prediction_th_",
    213: "radiusSearch(idx,this->non_maxima_radius_", 219: "points[idx].intensity<response->points[*iIt].intensity",
    222: "break", 224: "points[idx].intensity==response->points[*iIt].intensity", 225: "idx!=*iIt", 227: "draws.push_back(*iIt)",
    233: "non_maxima_draws_remove_&&has_draw", 234: "std::find(skipList.begin(),skipList.end(),idx)==skipList.end()",
    239: ".norm()", 240: "distance<non_maxima_draws_threshold_", 242: "skipList.push_back(draws[i])",
    247: "keypoints_indices_->indices.push_back(idx)", 253: "keypoints_indices_->indices.push_back(idx)",
}

# the four updates: reference line -> oracle citation; names mapped onto the oracle's
RENAME = {"annulus_weight": "aw", "bin_weight": "bw", "annulus_index": "a", "annulus_pair": "ap", "bin_index": "bi", "bin_pair": "bp"}
UPDATES = [350, 351, 354, 355]


def ref_lines(rel):
    with open(os.path.join(REF, rel), "rb") as f:
        return f.read().decode("latin-1").replace("\r", "").split("\n")


def squeeze(s):
    return re.sub(r"\s+", "", s)


def block_hash(lines, a, b):
    return hashlib.sha256(squeeze("\n".join(lines[a - 1:b])).encode()).hexdigest()


def rhs_tokens(expr):
    expr = expr.split("+=", 1)[1]
    expr = expr.split(";", 1)[0]
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+|[()*+\-/]", expr)
    return [RENAME.get(t, t) for t in toks]


def main():
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to check")
        return 0
    hpp, cpp = ref_lines(HPP), ref_lines(CPP)
    files = {HPP: hpp, CPP: cpp}
    oracle = open(ORACLE).read().split("\n")
    problems = []
    if "--update-hashes" in sys.argv:
        for rel, a, b, what in PINNED:
            print('    (%s, %d, %d): "%s",   # %s' % ("HPP" if rel == HPP else "CPP", a, b, block_hash(files[rel], a, b), what))
        return 0
    # 3. pinned ranges
    for rel, a, b, what in PINNED:
        h = block_hash(files[rel], a, b)
        if HASHES[(rel, a, b)] != h:
            problems.append("%s:%d-%d (%s) changed since the oracle was written against it: re-read the restatement "
                            "(sha256 %s...)" % (rel, a, b, what, h[:12]))
    # 1. line citations of the oracle
    cited = set()
    for ln in oracle:
        for m in re.finditer(r"/\*\s*:(\d+)(?:-(\d+))?[^*]*\*/", ln):
            cited.add(int(m.group(1)))
    for line, token in sorted(LINE_TOKENS.items()):
        if squeeze(token) not in squeeze(hpp[line - 1]):
            problems.append("%s:%d no longer holds `%s`" % (HPP, line, token))
    missing = [l for l in (336, 338, 342, 345, 348, 350, 351, 354, 355, 279, 281, 287, 213, 219, 224, 234, 240, 242) if l not in cited]
    if missing:
        problems.append("oracle/kpl_oracle.c lost its citation(s) of hpp lines %s" % missing)
    # 2. the four updates, token for token
    for line in UPDATES:
        ref_t = rhs_tokens(hpp[line - 1])
        mine = [ln for ln in oracle if re.search(r"/\*\s*:%d\s*\*/" % line, ln) and "+=" in ln]
        if len(mine) != 1:
            problems.append("oracle/kpl_oracle.c: expected exactly one '+=' statement citing :%d, found %d" % (line, len(mine)))
            continue
        if rhs_tokens(mine[0]) != ref_t:
            problems.append("update :%d differs: reference %s, oracle %s" % (line, " ".join(ref_t), " ".join(rhs_tokens(mine[0]))))
        # and the cell it goes to: histograms(x, y) vs H[x * B + y]
        rm = re.search(r"histograms\((\w+),\s*(\w+)\)", hpp[line - 1])
        om = re.search(r"H\[(\w+) \* B \+ (\w+)\]", mine[0])
        if not rm or not om or (RENAME[rm.group(1)], RENAME[rm.group(2)]) != (om.group(1), om.group(2)):
            problems.append("update :%d goes to another cell in the oracle" % line)
    for p in problems:
        print("DRIFT:", p)
    if not problems:
        print("oracle/kpl_oracle.c and the cited reference lines agree (%d line tokens, %d updates, %d pinned ranges)"
              % (len(LINE_TOKENS), len(UPDATES), len(PINNED)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
