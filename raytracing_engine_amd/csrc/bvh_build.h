// bvh_build.h — host BVH2 builder for path B (see bvh_build.cpp for the layout).
#pragma once
#include <cstdint>
#include <vector>

namespace rt {

struct BvhResult {
    std::vector<float> nodes;     // 32 floats (128 B) per 4-wide node
    std::vector<uint32_t> order;  // leaf-order position -> original triangle index
    uint32_t leaf_max = 2;        // in: triangles per leaf (1..4); 2 measured best on the bench scene
    uint32_t n_nodes = 0;
    uint32_t depth = 0;           // levels of 4-wide inner nodes (root = 1)
    uint32_t stack_need = 0;      // worst-case traversal stack occupancy (entries)
    float pad = 0.0f;             // conservative box padding that was applied
    double sah_area = 0.0;        // sum of child half-areas (quality metric)
};

constexpr uint32_t kBvhMaxDepth = 30;  // depth cap of the binary tree before it is collapsed

// v0/e1/e2: n*3 floats each (edges already formed in fp32).  Returns false on invalid input.
bool build_bvh(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t max_depth, BvhResult* out);

}  // namespace rt
