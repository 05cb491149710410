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
    float pad_in = -1.0f;         // in: >= 0: use this box padding instead of 2e-5 * max(largest |coordinate|, 1) (the chunks of a two-level
                                  // build share the whole mesh's padding)
    int max_threads = 0;          // in: > 0: use at most this many threads (the chunks of a two-level build are built side by side)
    uint32_t n_nodes = 0;
    uint32_t depth = 0;           // levels of 8-wide inner nodes (root = 1)
    uint32_t stack_need = 0;      // worst-case traversal stack occupancy (entries)
    float pad = 0.0f;             // conservative box padding that was applied
    float maxabs = 1.0f;          // out: max(1, largest |vertex coordinate|): what the padding was chosen for (pad_in < 0: pad = 2e-5 * maxabs)
    double sah_area = 0.0;        // sum of child half-areas (quality metric)
};

constexpr uint32_t kBvhMaxDepth = 30;  // depth cap of the binary tree before it is collapsed

// v0/e1/e2: n*3 floats each (edges already formed in fp32).  Returns false on invalid input.
bool build_bvh(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t max_depth, BvhResult* out);

// Two-level build ("TLAS over BLAS chunks", BASELINE.json configs[2]): the triangle set is cut into `chunks` leaves of a
// top-down binned-SAH split (the fullest leaf is split next, so chunk sizes stay within a small factor of each other and
// chunk boxes do not overlap the way runs of a Morton order do), each chunk gets a BVH of its own (a bottom-level
// structure, build_bvh), and a top-level BVH8 is built over the chunk boxes.  The result is flattened into ONE node array of the same format - a
// top-level leaf simply becomes an inner child that is the chunk's root - so the traversal kernels do not know the
// difference; what the split buys is that a chunk whose triangles moved is rebuilt alone (rebuild_chunk): 1/chunks of the
// binned-SAH work plus a microscopic top level, then a re-flatten.
struct TwoLevelBvh {
    std::vector<BvhResult> blas;            // per chunk, local node / triangle indices
    std::vector<uint32_t> sorted;           // chunk order -> original triangle index; chunk c owns sorted[first[c] .. first[c + 1]), ascending inside a chunk
    std::vector<uint32_t> first;            // chunks + 1 entries
    float pad = 0.0f;
    float maxabs = 1.0f;                    // max(1, largest |vertex coordinate|) at build time: rebuild_chunk's coordinate range
    uint32_t tlas_nodes = 0, tlas_depth = 0;
    double ms_cut = 0.0, ms_blas = 0.0, ms_tlas = 0.0, ms_flatten = 0.0;  // wall time of the last (re)build's phases
};
bool build_bvh_two_level(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t chunks, uint32_t max_depth, TwoLevelBvh* tl, BvhResult* out);
// Rebuild bottom-level structure `chunk` from the current vertex data (v0/e1/e2 of the WHOLE mesh, original triangle order;
// the chunk keeps its triangles), rebuild the top level over the new chunk boxes and flatten again into `out`.
// Transactional: on failure (false, or an exception) `tl` is as it was.  On success `displaced` (may be NULL) receives the
// chunk's previous structure, so a caller whose own later steps fail can put it back (std::swap(tl->blas[chunk], *displaced)).
bool rebuild_chunk(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t chunk, uint32_t max_depth, TwoLevelBvh* tl, BvhResult* out,
                   BvhResult* displaced = nullptr);

}  // namespace rt
