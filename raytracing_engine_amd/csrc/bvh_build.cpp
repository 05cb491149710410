// bvh_build.cpp — host-side BVH2 builder for path B (binned SAH, child-pair nodes).
//
// No reference counterpart: the reference has no triangles or BVH; this is the
// build-defined extension of DESIGN.md §6.  The renderer's results do not depend on the tree
// (boxes are padded conservatively, closest hit = lexicographic (t, triangle id) minimum), so the
// builder is free to optimise for traversal cost only.
//
// The binary SAH tree is collapsed into a compressed 8-wide BVH (after Ylitie, Karras, Laine,
// "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", HPG 2017): traversal on
// gfx950 is bound by the vector L1's request rate (one lane-address per cycle; a divergent 16-byte
// load is 64 of them), so the node format minimises 16-byte fetches per ray: 8 children in 5 fetches.
// Layout: bvh_build.h.
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

struct Child {
    float box[6];  // lo.xyz, hi.xyz (already padded)
    int32_t ref;   // >= 0: binary inner node that becomes an 8-wide node; < 0: leaf ~((first << 2) | (count - 1)), count <= 3
};

inline float box_area(const float b[6]) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}

// Binary SAH tree (one triangle per leaf) -> compressed 8-wide BVH (layout in bvh_build.h).
// Which binary nodes become 8-wide nodes, which subtrees become <= 3-triangle leaves and how the 8
// child slots of every node are spent is chosen by the dynamic program of Ylitie et al. 2017 (§4.1):
//   C(n,1)   = min( A(n) P(n) c_prim  [P(n) <= 3],   A(n) c_node + D(n,8) )
//   C(n,i>1) = min( D(n,i), C(n,i-1) ),   D(n,i) = min_{0<k<i} C(left,k) + C(right,i-k)
// Nodes are emitted breadth-first so that the inner children of a node are consecutive
// (child_base + rank among inner slots) and the triangles of its leaf children are consecutive
// (tri_base + offset < 24).
struct Cw8Builder {
    const std::vector<float>& n2;           // 16 floats per binary node
    const std::vector<uint32_t>& order2;    // binary leaf order -> triangle id
    std::vector<uint32_t>& nodes;           // 20 words per node
    std::vector<uint32_t>& order8;          // final leaf order -> triangle id
    uint32_t depth = 0;

    static constexpr float kCostNode = 1.0f;
    float kCostPrim = 0.8f;
    struct Dp {
        float cost[8];     // cost[i], i = 1..7: subtree as a forest of <= i roots
        uint8_t split[9];  // split[i], i = 2..8: roots given to the left child in D(n,i)
        uint8_t prev;      // bit i set: C(n,i) = C(n,i-1)
        uint8_t is_leaf;   // C(n,1) chose the leaf
        uint32_t first, prims;
        float box[6];
    };
    std::vector<Dp> dp;

    void child_box(int32_t node2, int side, float out[6]) const { std::memcpy(out, &n2[(size_t)node2 * 16 + 6 * side], 24); }
    int32_t child_ref(int32_t node2, int side) const {
        int32_t r;
        std::memcpy(&r, &n2[(size_t)node2 * 16 + 12 + side], 4);
        return r;
    }

    void solve() {
        const size_t n_nodes = n2.size() / 16;
        dp.resize(n_nodes);
        for (size_t jj = n_nodes; jj-- > 0;) {  // children have larger indices than their parent
            const int32_t j = (int32_t)jj;
            Dp& d = dp[jj];
            float cb[2][6], carea[2];
            int32_t cr[2];
            uint32_t cprims[2], cfirst[2];
            for (int s = 0; s < 2; s++) {
                child_box(j, s, cb[s]);
                cr[s] = child_ref(j, s);
                carea[s] = box_area(cb[s]);
                if (cr[s] < 0) {
                    const uint32_t ref = ~(uint32_t)cr[s];
                    cfirst[s] = ref >> 2;
                    cprims[s] = (ref & 3u) + 1u;
                } else {
                    cfirst[s] = dp[cr[s]].first;
                    cprims[s] = dp[cr[s]].prims;
                }
            }
            auto C = [&](int s, int i) -> float {
                if (cr[s] < 0) return carea[s] * (float)cprims[s] * kCostPrim;
                return dp[cr[s]].cost[i > 7 ? 7 : i];
            };
            for (int a = 0; a < 3; a++) {
                d.box[a] = std::min(cb[0][a], cb[1][a]);
                d.box[3 + a] = std::max(cb[0][3 + a], cb[1][3 + a]);
            }
            const bool same = cr[0] < 0 && cr[0] == cr[1];  // tiny mesh: both slots name one leaf
            d.first = std::min(cfirst[0], cfirst[1]);
            d.prims = same ? cprims[0] : cprims[0] + cprims[1];
            float D[9];
            D[1] = std::numeric_limits<float>::infinity();
            for (int i = 2; i <= 8; i++) {
                float best = std::numeric_limits<float>::infinity();
                int bk = 1;
                for (int k = 1; k < i; k++) {
                    const float c = C(0, k) + C(1, i - k);
                    if (c < best) {
                        best = c;
                        bk = k;
                    }
                }
                D[i] = best;
                d.split[i] = (uint8_t)bk;
            }
            const float area = box_area(d.box);
            const float c_leaf = d.prims <= 3 ? area * (float)d.prims * kCostPrim : std::numeric_limits<float>::infinity();
            const float c_int = D[8] + area * kCostNode;
            d.is_leaf = c_leaf <= c_int;
            d.cost[1] = d.is_leaf ? c_leaf : c_int;
            d.prev = 0;
            for (int i = 2; i <= 7; i++) {
                if (d.cost[i - 1] <= D[i]) {
                    d.cost[i] = d.cost[i - 1];
                    d.prev |= (uint8_t)(1u << i);
                } else {
                    d.cost[i] = D[i];
                }
            }
            d.cost[0] = 0.0f;
        }
    }

    // represent binary subtree `ref` (box `box`) as a forest of at most `budget` roots
    void expand(int32_t ref, const float box[6], int budget, Child* out, int& k) const {
        if (ref < 0) {
            std::memcpy(out[k].box, box, 24);
            out[k++].ref = ref;
            return;
        }
        const Dp& d = dp[ref];
        if (budget > 7) budget = 7;
        while (budget > 1 && ((d.prev >> budget) & 1)) budget--;
        if (budget == 1) {
            std::memcpy(out[k].box, d.box, 24);
            out[k++].ref = d.is_leaf ? ~(int32_t)((d.first << 2) | (d.prims - 1u)) : ref;
            return;
        }
        distribute(ref, budget, out, k);
    }
    void distribute(int32_t node2, int budget, Child* out, int& k) const {
        const int kl = dp[node2].split[budget];
        float b0[6], b1[6];
        child_box(node2, 0, b0);
        child_box(node2, 1, b1);
        const int32_t r0 = child_ref(node2, 0), r1 = child_ref(node2, 1);
        if (r0 < 0 && r0 == r1) {  // tiny mesh
            expand(r0, b0, 1, out, k);
            return;
        }
        expand(r0, b0, kl, out, k);
        expand(r1, b1, budget - kl, out, k);
    }

    struct Pending {
        int32_t node2;
        uint32_t index, level;
    };

    void emit(const Pending& pd, std::vector<Pending>& queue) {
        Child ch[8];
        int k = 0;
        if (dp[pd.node2].is_leaf && pd.level == 0) {  // whole mesh fits one leaf: root node with a single leaf child
            const Dp& d = dp[pd.node2];
            std::memcpy(ch[0].box, d.box, 24);
            ch[0].ref = ~(int32_t)((d.first << 2) | (d.prims - 1u));
            k = 1;
        } else {
            distribute(pd.node2, 8, ch, k);
        }
        depth = std::max(depth, pd.level + 1);

        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = ch[0].box[a];
            hi[a] = ch[0].box[3 + a];
            for (int i = 1; i < k; i++) {
                lo[a] = std::min(lo[a], ch[i].box[a]);
                hi[a] = std::max(hi[a], ch[i].box[3 + a]);
            }
        }
        // slot assignment: slot bits (x,y,z) = which side of the node centre the child sits on, so that
        // slot ^ (7 - ray octant) orders children front to back; greedy on dot(child centre - node centre, slot dir)
        int slot_of[8], child_in[8];
        for (int i = 0; i < 8; i++) slot_of[i] = child_in[i] = -1;
        for (int round = 0; round < k; round++) {
            float best = -std::numeric_limits<float>::infinity();
            int bi = -1, bs = -1;
            for (int i = 0; i < k; i++) {
                if (slot_of[i] >= 0) continue;
                for (int s = 0; s < 8; s++) {
                    if (child_in[s] >= 0) continue;
                    float c = 0.0f;
                    for (int a = 0; a < 3; a++) {
                        const float off = 0.5f * (ch[i].box[a] + ch[i].box[3 + a]) - 0.5f * (lo[a] + hi[a]);
                        c += ((s >> (2 - a)) & 1) ? off : -off;
                    }
                    if (c > best) {
                        best = c;
                        bi = i;
                        bs = s;
                    }
                }
            }
            slot_of[bi] = bs;
            child_in[bs] = bi;
        }

        // quantisation frame: p = lo, per-axis power-of-two scale with 255 * scale >= extent
        uint32_t e_byte[3];
        double scale[3];
        for (int a = 0; a < 3; a++) {
            const double ext = (double)hi[a] - (double)lo[a];
            int e = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -126;
            e = std::min(std::max(e, -126), 127);
            while (e < 127 && std::ldexp(255.0, e) < ext) e++;
            e_byte[a] = (uint32_t)(e + 127);
            scale[a] = std::ldexp(1.0, e);
        }

        uint32_t* w = &nodes[(size_t)pd.index * 20];
        std::memcpy(w, lo, 12);
        uint32_t imask = 0, n_inner = 0;
        for (int s = 0; s < 8; s++)
            if (child_in[s] >= 0 && ch[child_in[s]].ref >= 0) {
                imask |= 1u << s;
                n_inner++;
            }
        w[3] = e_byte[0] | (e_byte[1] << 8) | (e_byte[2] << 16) | (imask << 24);
        const uint32_t child_base = (uint32_t)(nodes.size() / 20);
        const uint32_t tri_base = (uint32_t)order8.size();
        w[4] = child_base;
        w[5] = tri_base;
        nodes.resize(nodes.size() + (size_t)n_inner * 20);
        w = &nodes[(size_t)pd.index * 20];  // resize may have moved the storage
        uint8_t meta[8] = {}, q[6][8];
        for (int a = 0; a < 6; a++)
            for (int s = 0; s < 8; s++) q[a][s] = a < 3 ? 255 : 0;  // empty slot: inverted box, meta 0
        uint32_t rank = 0;
        for (int s = 0; s < 8; s++) {
            const int i = child_in[s];
            if (i < 0) continue;
            if (ch[i].ref >= 0) {
                meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
                queue.push_back(Pending{ch[i].ref, child_base + rank, pd.level + 1});
                rank++;
            } else {
                const uint32_t ref = ~(uint32_t)ch[i].ref, first = ref >> 2, cnt = (ref & 3u) + 1u;  // cnt <= 3
                const uint32_t off = (uint32_t)order8.size() - tri_base;
                meta[s] = (uint8_t)((((1u << cnt) - 1u) << 5) | off);
                for (uint32_t t = 0; t < cnt; t++) order8.push_back(order2[first + t]);
            }
            for (int a = 0; a < 3; a++) {
                double ql = std::floor(((double)ch[i].box[a] - (double)lo[a]) / scale[a]);
                double qh = std::ceil(((double)ch[i].box[3 + a] - (double)lo[a]) / scale[a]);
                ql = std::min(std::max(ql, 0.0), 255.0);
                qh = std::min(std::max(qh, 0.0), 255.0);
                q[a][s] = (uint8_t)ql;
                q[3 + a][s] = (uint8_t)qh;
            }
        }
        std::memcpy(&w[6], meta, 8);
        for (int a = 0; a < 6; a++) std::memcpy(&w[8 + 2 * a], q[a], 8);
    }

    void build() {
        solve();
        nodes.assign(20, 0u);
        std::vector<Pending> queue;
        queue.push_back(Pending{0, 0, 0});
        for (size_t head = 0; head < queue.size(); head++) {
            const Pending pd = queue[head];
            emit(pd, queue);
        }
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
    std::vector<float> binary;
    binary.reserve(16 * (size_t)(n / 2 + 16));
    // conservative padding: absorbs the rounding of the slab test and of Moeller-Trumbore's t
    out->pad = 2e-5f * std::max(maxabs, 1.0f);
    const uint32_t leaf_max = 1;  // the binary tree goes down to single triangles; the collapse forms the <= 3-triangle leaves
    Builder b{tri_box, centroid, out->order, binary, out->pad, max_depth, leaf_max};
    if (n <= leaf_max) {
        // one node whose two slots name the same leaf (testing it twice is idempotent)
        binary.resize(16);
        Box bx = b.range_box(0, n);
        float* nd = binary.data();
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
    // collapse to compressed 8-wide nodes; triangles are re-ordered so every node's leaf triangles are contiguous
    std::vector<uint32_t> order2;
    order2.swap(out->order);
    out->order.reserve(n);
    Cw8Builder cw{binary, order2, out->nodes, out->order, 0, out->cost_prim, {}};
    cw.build();
    if (out->order.size() != n) return false;
    out->n_nodes = (uint32_t)(out->nodes.size() / 20);
    out->depth = cw.depth;
    out->stack_need = cw.depth + 1;  // at most one pending sibling group per level
    return true;
}

}  // namespace rt
