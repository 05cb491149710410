// bvh_check.cpp — host-side structural check of the compressed 8-wide BVH (csrc/bvh_build.cpp),
// built with -fsanitize=address,undefined by tests/test_bvh_build_host.py.  No GPU involved.
//   bvh_check <n_tris> <seed> <edge> [chunks [chunk-to-rebuild]]
// Verifies: the leaf order is a permutation; every triangle is reachable exactly once; inner-child
// indexing (child_base + popcount(imask below slot)) and leaf indexing (one triangle per leaf slot, tri_base +
// popcount(leafmask below slot)) are consistent; empty slots hold inverted boxes; every leaf triangle lies inside its de-quantised child box; depth <= stack_need - 1.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../raytracing_engine_amd/csrc/bvh_build.h"

static uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

int main(int argc, char** argv) {
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 1000;
    const uint32_t seed = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1;
    const float edge = argc > 3 ? (float)std::atof(argv[3]) : 0.5f;
    std::vector<float> v0(3 * (size_t)n), e1(3 * (size_t)n), e2(3 * (size_t)n);
    uint32_t s = hash32(seed);
    auto u = [&]() { s = hash32(s + 0x9e3779b9u); return (float)(s >> 8) * 0x1p-24f; };
    for (size_t i = 0; i < (size_t)n * 3; i++) {
        v0[i] = u() * 20.0f - 10.0f;
        e1[i] = (u() * 2.0f - 1.0f) * edge;
        e2[i] = (u() * 2.0f - 1.0f) * edge;
    }
    if (n > 10) {  // degenerate and duplicate triangles must survive
        for (int a = 0; a < 3; a++) e1[3 * 5 + a] = e2[3 * 5 + a] = 0.0f;
        for (int a = 0; a < 3; a++) { v0[3 * 7 + a] = v0[3 * 6 + a]; e1[3 * 7 + a] = e1[3 * 6 + a]; e2[3 * 7 + a] = e2[3 * 6 + a]; }
    }
    const uint32_t chunks = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 0;  // > 0: two-level build (top level over `chunks` bottom-level BVHs, flattened)
    rt::BvhResult b;
    rt::TwoLevelBvh tl;
    if (chunks) {
        if (!rt::build_bvh_two_level(v0.data(), e1.data(), e2.data(), n, chunks, rt::kBvhMaxDepth, &tl, &b)) { std::puts("FAIL two-level build"); return 1; }
        if (argc > 5) {  // rebuild one chunk after moving its triangles: the flattened result must pass the same checks
            const uint32_t c = (uint32_t)std::atoi(argv[5]) % (uint32_t)tl.blas.size();
            for (uint32_t i = tl.first[c]; i < tl.first[c + 1]; i++)
                for (int a = 0; a < 3; a++) v0[3 * (size_t)tl.sorted[i] + a] = v0[3 * (size_t)tl.sorted[i] + a] * 0.5f + 0.25f;
            if (!rt::rebuild_chunk(v0.data(), e1.data(), e2.data(), n, c, rt::kBvhMaxDepth, &tl, &b)) { std::puts("FAIL chunk rebuild"); return 1; }
        }
    } else if (!rt::build_bvh(v0.data(), e1.data(), e2.data(), n, rt::kBvhMaxDepth, &b)) { std::puts("FAIL build"); return 1; }
    if (b.order.size() != n || b.nodes.size() != (size_t)b.n_nodes * 20) { std::puts("FAIL sizes"); return 1; }
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t t : b.order) {
        if (t >= n || seen[t]) { std::puts("FAIL order is not a permutation"); return 1; }
        seen[t] = 1;
    }
    std::vector<uint32_t> reached(n, 0);
    struct Item { uint32_t node, level; };
    std::vector<Item> stack{{0, 1}};
    uint32_t max_level = 0, n_visited = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.node >= b.n_nodes) { std::puts("FAIL node index out of range"); return 1; }
        n_visited++;
        max_level = it.level > max_level ? it.level : max_level;
        const uint32_t* w = &b.nodes[(size_t)it.node * 20];
        float p[3];
        std::memcpy(p, w, 12);
        float scale[3];
        for (int a = 0; a < 3; a++) { const uint32_t bits = ((w[3] >> (8 * a)) & 0xffu) << 23; std::memcpy(&scale[a], &bits, 4); }
        const uint32_t imask = w[3] >> 24, child_base = w[4], tri_base = w[5];
        const uint32_t leafmask = w[6];
        if (leafmask > 0xffu || w[7] != 0u || (leafmask & imask)) { std::puts("FAIL leafmask / reserved word"); return 1; }
        const uint8_t* q = reinterpret_cast<const uint8_t*>(&w[8]);  // [6][8]
        for (int slot = 0; slot < 8; slot++) {
            const bool inner = (imask >> slot) & 1u, leaf = (leafmask >> slot) & 1u;
            if (!inner && !leaf) {  // empty slot: inverted box that no ray can hit
                for (int a = 0; a < 3; a++)
                    if (q[a * 8 + slot] != 255 || q[(3 + a) * 8 + slot] != 0) { std::puts("FAIL empty slot without an inverted box"); return 1; }
                continue;
            }
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) { lo[a] = p[a] + (float)q[a * 8 + slot] * scale[a]; hi[a] = p[a] + (float)q[(3 + a) * 8 + slot] * scale[a]; }
            if (inner) {
                const uint32_t rel = (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u));
                stack.push_back({child_base + rel, it.level + 1});
            } else {  // a leaf is one triangle: tri_base + rank of the slot among the node's leaf slots
                const uint32_t li = tri_base + (uint32_t)__builtin_popcount(leafmask & ((1u << slot) - 1u));
                if (li >= n) { std::puts("FAIL leaf triangle index"); return 1; }
                const uint32_t t = b.order[li];
                reached[t]++;
                for (int a = 0; a < 3; a++) {
                    const float x0 = v0[3 * (size_t)t + a], x1 = x0 + e1[3 * (size_t)t + a], x2 = x0 + e2[3 * (size_t)t + a];
                    const float mn = std::fmin(x0, std::fmin(x1, x2)), mx = std::fmax(x0, std::fmax(x1, x2));
                    if (mn < lo[a] || mx > hi[a]) { std::printf("FAIL triangle %u outside its leaf box on axis %d\n", t, a); return 1; }
                }
            }
        }
    }
    for (uint32_t t = 0; t < n; t++)
        if (reached[t] != 1) { std::printf("FAIL triangle %u reached %u times\n", t, reached[t]); return 1; }
    if (n_visited != b.n_nodes) { std::puts("FAIL unreachable nodes"); return 1; }
    if (max_level != b.depth || b.stack_need != b.depth + 1) { std::printf("FAIL depth %u vs %u\n", max_level, b.depth); return 1; }
    unsigned long long h = 1469598103934665603ull;  // FNV-1a over the node words and the leaf order: the builder's whole output
    for (uint32_t w : b.nodes) h = (h ^ w) * 1099511628211ull;
    for (uint32_t w : b.order) h = (h ^ w) * 1099511628211ull;
    std::printf("OK n=%u nodes=%u depth=%u tris/node=%.2f hash=%016llx\n", n, b.n_nodes, b.depth, (double)n / b.n_nodes, h);
    return 0;
}
