// bvh_build.cpp — host-side BVH2 builder for path B (binned SAH, child-pair nodes).
//
// No reference counterpart: the reference has no triangles or BVH; this is the
// build-defined extension of DESIGN.md §6.  The renderer's results do not depend on the tree
// (boxes are padded conservatively, closest hit = lexicographic (t, triangle id) minimum), so the
// builder is free to optimise for traversal cost only.
//
// The binary SAH tree is collapsed into a 4-wide BVH: traversal on gfx950 is bound by the latency of
// dependent node fetches (L2 / Infinity-Cache round trips), so halving the number of levels matters
// more than the extra box tests per step.
//
// Output layout, sized for per-lane gathers on gfx950:
//   nodes: one 128-byte record (= one L2 line) per inner node = 8 x float4, children planar (SoA):
//       q0 = lo.x[0..3]  q1 = lo.y[0..3]  q2 = lo.z[0..3]  q3 = hi.x[0..3]  q4 = hi.y[0..3]  q5 = hi.z[0..3]
//       q6 = ref[0..3]   q7 = unused
//       ref >= 0: inner node index; ref < 0: leaf, ~ref = (first << 2) | (count - 1), count <= 4;
//       ref == 0x80000000: empty slot (its box is a far-away point no ray reaches)
//   order: leaf-order position -> original triangle index (triangle records are stored in leaf order
//       so a leaf's triangles are contiguous)
#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace rt {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box& b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    void grow(const float p[3]) {
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 16;

struct Builder {
    const std::vector<Box>& tri_box;
    const std::vector<float>& centroid;  // n*3
    std::vector<uint32_t>& order;
    std::vector<float>& nodes;  // 16 floats per node
    float pad;
    uint32_t max_depth;
    uint32_t kLeafMax = 4;
    uint32_t depth_reached = 0;
    double sah = 0.0;

    Box range_box(uint32_t first, uint32_t count) const {
        Box b;
        b.reset();
        for (uint32_t i = 0; i < count; i++) b.grow(tri_box[order[first + i]]);
        return b;
    }

    static int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 2) | (count - 1u)); }

    // smallest depth a balanced tree needs for `count` triangles with leaves of kLeafMax
    uint32_t min_depth(uint32_t count) const {
        uint32_t d = 0;
        uint64_t cap = kLeafMax;
        while (cap < count) {
            cap <<= 1;
            d++;
        }
        return d;
    }

    // returns child ref; depth = depth of the node that would be created
    int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
        if (count <= kLeafMax) return leaf_ref(first, count);
        depth_reached = std::max(depth_reached, depth);
        const uint32_t me = (uint32_t)(nodes.size() / 16);
        nodes.resize(nodes.size() + 16);

        // centroid bounds
        float clo[3], chi[3];
        for (int a = 0; a < 3; a++) {
            clo[a] = std::numeric_limits<float>::infinity();
            chi[a] = -clo[a];
        }
        for (uint32_t i = 0; i < count; i++) {
            const float* c = &centroid[3 * (size_t)order[first + i]];
            for (int a = 0; a < 3; a++) {
                clo[a] = std::min(clo[a], c[a]);
                chi[a] = std::max(chi[a], c[a]);
            }
        }

        uint32_t mid = 0;
        // depth budget: once the remaining levels are only just enough for a balanced split, stop using SAH
        const bool must_balance = depth + 1 + min_depth((count + 1) / 2) >= max_depth;
        if (!must_balance) {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            for (int a = 0; a < 3; a++) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0.0f)) continue;
                const float k = (float)kBins / ext;
                Box bb[kBins];
                uint32_t bc[kBins] = {};
                for (auto& b : bb) b.reset();
                for (uint32_t i = 0; i < count; i++) {
                    const uint32_t t = order[first + i];
                    int bin = (int)((centroid[3 * (size_t)t + a] - clo[a]) * k);
                    bin = std::min(std::max(bin, 0), kBins - 1);
                    bb[bin].grow(tri_box[t]);
                    bc[bin]++;
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                acc.reset();
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; b--) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    right_area[b] = cnt ? acc.half_area() : 0.0f;
                    right_cnt[b] = cnt;
                }
                acc.reset();
                cnt = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bb[b]);
                    cnt += bc[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = b;
                    }
                }
            }
            if (best_axis >= 0) {
                const float k = (float)kBins / (chi[best_axis] - clo[best_axis]);
                const float lo = clo[best_axis];
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                    int bin = (int)((centroid[3 * (size_t)t + best_axis] - lo) * k);
                    bin = std::min(std::max(bin, 0), kBins - 1);
                    return bin <= best_bin;
                });
                mid = (uint32_t)(it - (order.begin() + first));
                // keep both subtrees within the depth budget
                if (depth + 1 + min_depth(std::max(mid, count - mid)) >= max_depth) mid = 0;
            }
        }
        if (mid == 0 || mid == count) {  // median split on the longest centroid axis
            int axis = 0;
            if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
            if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
            mid = count / 2;
            std::nth_element(order.begin() + first, order.begin() + first + mid, order.begin() + first + count,
                             [&](uint32_t x, uint32_t y) {
                                 const float cx = centroid[3 * (size_t)x + axis], cy = centroid[3 * (size_t)y + axis];
                                 return cx < cy || (cx == cy && x < y);
                             });
        }

        Box b0 = range_box(first, mid), b1 = range_box(first + mid, count - mid);
        sah += (double)b0.half_area() + (double)b1.half_area();
        const int32_t r0 = build(first, mid, depth + 1);
        const int32_t r1 = build(first + mid, count - mid, depth + 1);
        float* n = &nodes[(size_t)me * 16];
        const float p = pad;
        n[0] = b0.lo[0] - p; n[1] = b0.lo[1] - p; n[2] = b0.lo[2] - p; n[3] = b0.hi[0] + p;
        n[4] = b0.hi[1] + p; n[5] = b0.hi[2] + p; n[6] = b1.lo[0] - p; n[7] = b1.lo[1] - p;
        n[8] = b1.lo[2] - p; n[9] = b1.hi[0] + p; n[10] = b1.hi[1] + p; n[11] = b1.hi[2] + p;
        std::memcpy(&n[12], &r0, 4);
        std::memcpy(&n[13], &r1, 4);
        n[14] = n[15] = 0.0f;
        return (int32_t)me;
    }
};

}  // namespace

namespace {

constexpr int32_t kEmptyRef = (int32_t)0x80000000;

struct Child {
    float box[6];  // lo.xyz, hi.xyz (already padded)
    int32_t ref;   // BVH2 ref
    float area() const {
        const float dx = box[3] - box[0], dy = box[4] - box[1], dz = box[5] - box[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Collapser {
    const std::vector<float>& n2;  // 16 floats per binary node
    std::vector<float>& n4;        // 32 floats per 4-wide node
    uint32_t stack_need = 0, depth = 0;

    void children_of(int32_t node2, Child out[2]) const {
        const float* n = &n2[(size_t)node2 * 16];
        std::memcpy(out[0].box, n, 24);
        std::memcpy(out[1].box, n + 6, 24);
        std::memcpy(&out[0].ref, n + 12, 4);
        std::memcpy(&out[1].ref, n + 13, 4);
    }

    // below = stack entries already held when this node is entered (worst case)
    int32_t collapse(int32_t node2, uint32_t below, uint32_t level) {
        Child ch[4];
        int k = 2;
        children_of(node2, ch);
        if (ch[0].ref < 0 && ch[0].ref == ch[1].ref) k = 1;  // tiny mesh: both binary slots name one leaf
        while (k < 4) {  // open the inner child with the largest surface until 4 children
            int best = -1;
            for (int i = 0; i < k; i++)
                if (ch[i].ref >= 0 && (best < 0 || ch[i].area() > ch[best].area())) best = i;
            if (best < 0) break;
            Child two[2];
            children_of(ch[best].ref, two);
            ch[best] = two[0];
            ch[k++] = two[1];
        }
        const uint32_t me = (uint32_t)(n4.size() / 32);
        n4.resize(n4.size() + 32);
        const uint32_t held = below + (uint32_t)(k - 1);
        stack_need = std::max(stack_need, held);
        depth = std::max(depth, level + 1);
        int32_t refs[4];
        for (int i = 0; i < 4; i++) refs[i] = i < k ? (ch[i].ref < 0 ? ch[i].ref : collapse(ch[i].ref, held, level + 1)) : kEmptyRef;
        float* n = &n4[(size_t)me * 32];
        for (int i = 0; i < 4; i++)
            for (int a = 0; a < 6; a++) n[a * 4 + i] = i < k ? ch[i].box[a] : 3.0e38f;
        std::memcpy(n + 24, refs, 16);
        n[28] = n[29] = n[30] = n[31] = 0.0f;
        return (int32_t)me;
    }
};

}  // namespace

bool build_bvh(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t max_depth, BvhResult* out) {
    if (!v0 || !e1 || !e2 || n == 0 || n >= (1u << 29) || !out) return false;
    std::vector<Box> tri_box(n);
    std::vector<float> centroid(3 * (size_t)n);
    float maxabs = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        Box b;
        b.reset();
        float p0[3], p1[3], p2[3];
        for (int a = 0; a < 3; a++) {
            p0[a] = v0[3 * (size_t)i + a];
            p1[a] = p0[a] + e1[3 * (size_t)i + a];
            p2[a] = p0[a] + e2[3 * (size_t)i + a];
            centroid[3 * (size_t)i + a] = p0[a] + (e1[3 * (size_t)i + a] + e2[3 * (size_t)i + a]) * (1.0f / 3.0f);
            maxabs = std::max(maxabs, std::max(std::fabs(p0[a]), std::max(std::fabs(p1[a]), std::fabs(p2[a]))));
        }
        b.grow(p0);
        b.grow(p1);
        b.grow(p2);
        tri_box[i] = b;
    }
    out->order.resize(n);
    for (uint32_t i = 0; i < n; i++) out->order[i] = i;
    out->nodes.clear();
    out->nodes.reserve(16 * (size_t)(n / 2 + 16));
    // conservative padding: absorbs the rounding of the slab test and of Moeller-Trumbore's t
    out->pad = 2e-5f * std::max(maxabs, 1.0f);
    const uint32_t leaf_max = std::min<uint32_t>(std::max<uint32_t>(out->leaf_max, 1u), 4u);
    Builder b{tri_box, centroid, out->order, out->nodes, out->pad, max_depth, leaf_max};
    if (n <= leaf_max) {
        // one node whose two slots name the same leaf (testing it twice is idempotent)
        out->nodes.resize(16);
        Box bx = b.range_box(0, n);
        float* nd = out->nodes.data();
        const float p = out->pad;
        const float box[6] = {bx.lo[0] - p, bx.lo[1] - p, bx.lo[2] - p, bx.hi[0] + p, bx.hi[1] + p, bx.hi[2] + p};
        std::memcpy(nd, box, 24);
        std::memcpy(nd + 6, box, 24);
        const int32_t r = Builder::leaf_ref(0, n);
        std::memcpy(nd + 12, &r, 4);
        std::memcpy(nd + 13, &r, 4);
        nd[14] = nd[15] = 0.0f;
        out->depth = 1;
    } else {
        b.build(0, n, 0);
        out->depth = b.depth_reached + 1;
    }
    out->sah_area = b.sah;
    // collapse to 4-wide nodes
    std::vector<float> binary;
    binary.swap(out->nodes);
    out->nodes.reserve(binary.size());
    Collapser col{binary, out->nodes};
    col.collapse(0, 0, 0);
    out->n_nodes = (uint32_t)(out->nodes.size() / 32);
    out->depth = col.depth;
    out->stack_need = col.stack_need;
    return true;
}

}  // namespace rt
