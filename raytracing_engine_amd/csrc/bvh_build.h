// bvh_build.h — host BVH builder for path B: binned-SAH binary tree -> compressed 8-wide nodes.
//
// Node = 80 bytes = 5 x 16-byte fetches for 8 children (20 little-endian words):
//   w0..w2  p.xyz (f32)        origin of the node's quantisation frame (= box minimum)
//   w3      ex | ey<<8 | ez<<16 | imask<<24    per-axis scale = 2^(e-127) as a float exponent byte;
//                                               imask bit s = child slot s is an inner node
//   w4      child_base         index of the first inner child; inner child in slot s lives at
//                              child_base + popcount(imask & ((1<<s)-1))
//   w5      tri_base           leaf-order index of the node's first leaf triangle
//   w6      leafmask           bit s (0..7) = child slot s is a leaf.  A leaf is exactly ONE triangle, the one at
//                              tri_base + popcount(leafmask & ((1<<s)-1)); a slot in neither imask nor leafmask is empty
//   w7      0                  reserved
//   w8..w19 qlo.x[8] qlo.y[8] qlo.z[8] qhi.x[8] qhi.y[8] qhi.z[8]   child boxes, 8 bits per plane,
//                              box = p + q * scale, rounded outward (conservative); empty slots hold an inverted box
//                              (lo 255, hi 0) that no ray hits
// One-triangle leaves: with a quantised box per triangle the traversal's hit bits ARE the work lists (inner children
// to enter = hits & imask, triangles to test = hits & leafmask), no per-child count / offset decoding in the node step;
// on the 1 M-triangle soup the optimal-cut collapse chose single-triangle leaves for 98 % of the leaves anyway.
// Child slots are assigned so that slot ^ (7 - ray octant) enumerates children roughly front to back.
#pragma once
#include <cstdint>
#include <vector>

namespace rt {

struct BvhResult {
    std::vector<uint32_t> nodes;  // 20 words (80 B) per compressed 8-wide node
    std::vector<uint32_t> order;  // leaf-order position -> original triangle index
    float cost_prim = 0.8f;       // in: SAH cost of a triangle test relative to a node visit (collapse DP);
                                  // measured on the bench scene: 0.3 -> 1615, 0.6 -> 1794, 1.0 -> 1799 Mrays/s
    uint32_t n_nodes = 0;
    uint32_t depth = 0;           // levels of 8-wide inner nodes (root = 1)
    uint32_t stack_need = 0;      // worst-case traversal stack occupancy (entries)
    float pad = 0.0f;             // conservative box padding that was applied
    double sah_area = 0.0;        // sum of child half-areas (quality metric)
};

constexpr uint32_t kBvhMaxDepth = 30;  // depth cap of the binary tree before it is collapsed

// v0/e1/e2: n*3 floats each (edges already formed in fp32).  Returns false on invalid input.
bool build_bvh(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t max_depth, BvhResult* out);

}  // namespace rt
