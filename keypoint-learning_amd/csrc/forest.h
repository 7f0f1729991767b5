// forest.h -- host-side random forest model of libkpl.
//
// Replaces what the reference gets from cv::ml::RTrees (load / getRoots / predict; call sites
// /root/reference/include/impl/KeypointLearning.hpp:162,172,271,281): an OpenCV-YAML(.gz)
// reader, the tree structure as node arrays, and the flattened device layout.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace kpl {

// Trees as parsed: global node numbering, children by index, var < 0 marks a leaf.
struct ForestModel {
    int var_count = 0;
    std::vector<int> root;
    std::vector<int> var;
    std::vector<float> thr;
    std::vector<int> left;
    std::vector<int> right;
    std::vector<double> value;
    int ntrees() const { return (int)root.size(); }
    int64_t nnodes() const { return (int64_t)var.size(); }
};

// Device layout: one 8-byte record per node.
//   x = threshold bits (internal) or leaf value as float bits (leaf)
//   y = [31:24] split variable (0..254; 255 = leaf)   [23:0] index of the LEFT child (of a leaf: 0, or its chain link)
// The forest is laid out breadth first, level by level ACROSS the trees: node t is the root of tree t, then come the
// second levels of all trees, and so on -- the first k nodes are the top of every tree (what the forest kernels stage in
// LDS); siblings are adjacent (right = left + 1).  Slot ntrees, right behind the roots, is a "resting leaf" of value 0
// that belongs to no tree and points at itself (kernels.hip parks idle walks there).
//
// Two variants of what lies below the top:
//  * blocked (FlatForest::chain == 0): level-major for as many whole levels as fit kTopNodes slots (FlatForest::ntop
//    slots in all).  What lies below is laid out for the cache line and for one lane fetching a whole subtree at once:
//    blocks of 8 slots (64 bytes) holding a node, its 2 children and its 4 grandchildren (slots 0 | 1 2 | 3 4 5 6), the
//    blocks of two siblings side by side in one 128-byte line.  A walk below the top part costs one 64-byte fetch per
//    three levels instead of one line per level.  A record whose children start blocks holds the slot of the LEFT
//    child's block; the right child's block is 8 slots further.  Slots that hold no node are never referenced.
//  * chained (chain == kChainStride; forests of >= kChainMinTrees trees and >= kChainMinVars variables whose sum is
//    exact in any order -- what forest_split_kernel takes): level-major THROUGHOUT (ntop = all slots), and the record
//    of a leaf of tree t holds the root of tree t + chain as its "child" (the resting leaf past the last tree): a walk
//    that just follows the records goes through the trees t, t + chain, t + 2 chain, ... and then rests.
struct FlatNode {
    uint32_t x;
    uint32_t y;
};
constexpr uint32_t kLeafVar = 255u;
constexpr uint32_t kMaxFlatNodes = 1u << 24;   // slots, padding of the blocked part included
constexpr uint32_t kTopNodes = 8192;           // = the forest kernel's LDS node budget (64 KB)
constexpr uint32_t kBlockSlots = 8;
constexpr uint32_t kChainStride = 16;        // = lanes per point x walks per lane of the chained forest kernel (kernels.hip)
constexpr int kChainMinTrees = 40, kChainMinVars = 32;

struct FlatForest {
    int ntrees = 0;
    int var_count = 0;
    int max_depth = 0;               // longest root->leaf path, counted in nodes
    bool order_free = false;         // every leaf value is an integer of magnitude <= 2^15 and there are at most 2^15
                                     // trees: the sum of leaf values is exact in any order (and fits an int32)
    int chain = 0;                   // kChainStride: chained layout (see above); 0: blocked layout
    std::vector<FlatNode> nodes;     // slots
    uint32_t ntop = 0;               // slots of the level-major part; every slot >= ntop belongs to a block
    int64_t nnodes = 0;              // nodes of the model (nodes.size() counts padding slots too)
};

// Parses OpenCV FileStorage YAML text of a cv::ml::RTrees / DTrees / legacy CvRTrees model.
// Returns false and sets err on failure.
bool parse_forest_yaml(const char *text, size_t len, ForestModel &out, std::string &err);

// Reads a file, inflating it when it starts with the gzip magic.
bool read_maybe_gzip(const char *path, std::string &out, std::string &err);
bool inflate_if_gzip(const void *data, size_t len, std::string &out, std::string &err);

// Validates the structure (every tree a proper binary tree, indices in range, thresholds finite,
// leaf values exactly representable as float) and produces the device layout.
bool flatten_forest(const ForestModel &m, FlatForest &out, std::string &err);

}  // namespace kpl
