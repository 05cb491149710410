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

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <exception>
#include <limits>
#include <system_error>
#include <thread>

#ifdef RT_BVH_TIMING
#include <chrono>
#include <cstdio>
#define RT_BVH_T(x) const auto x = std::chrono::steady_clock::now()
#define RT_BVH_REPORT(what, a, b) std::fprintf(stderr, "bvh: %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>((b) - (a)).count())
#else
#define RT_BVH_T(x) (void)0
#define RT_BVH_REPORT(what, a, b) (void)0
#endif

namespace rt {
namespace {

// CPUs this process may run on (the GPU boxes grant a slice of the machine), at most 32
int host_threads() {
    cpu_set_t set;
    int n = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    return std::min(std::max(n, 1), 32);
}

// fn(i) for i in [0, n) on up to `threads` threads (contiguous chunks; fn must only touch item i's data).
// Nothing escapes a worker thread: an exception inside fn is carried back and rethrown here after every
// thread has been joined, and chunks whose thread could not be created (std::system_error: thread or
// process limit of the box) run on the calling thread, so the result never depends on how many started.
template <typename F>
void parallel_for(size_t n, int threads, size_t kMinPerThread, F fn) {
    const size_t want = std::min<size_t>((size_t)std::max(threads, 1), (n + kMinPerThread - 1) / kMinPerThread);
    if (want <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(want - 1);
    std::vector<std::exception_ptr> errs(want);
    const size_t chunk = (n + want - 1) / want;
    auto run = [&](size_t t) noexcept {
        try {
            for (size_t i = t * chunk; i < std::min(n, (t + 1) * chunk); i++) fn(i);
        } catch (...) {
            errs[t] = std::current_exception();
        }
    };
    for (size_t t = 1; t < want; t++) {
        try {
            pool.emplace_back(run, t);
        } catch (const std::system_error&) {
            break;
        }
    }
    run(0);
    for (size_t t = pool.size() + 1; t < want; t++) run(t);
    for (auto& th : pool) th.join();
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
}

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
    // Subtrees are independent (disjoint ranges of `order`, disjoint node indices: with single-triangle
    // leaves a range of `count` triangles owns exactly count - 1 nodes, numbered in preorder), so big
    // ones are built by their own threads.  Results do not depend on the thread count.
    std::atomic<int>* spare_threads = nullptr;
    int n_threads = 1;

    struct Sub {
        int32_t ref;
        uint32_t depth_reached;
        double sah;
    };

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

    // returns child ref; depth = depth of the node that would be created; `me` = its node index
    // (kLeafMax == 1: preorder numbering, the left subtree takes me + 1 .. me + mid - 1, the right one starts at me + mid)
    Sub build(uint32_t first, uint32_t count, uint32_t depth, uint32_t me) {
        if (count <= kLeafMax) return Sub{leaf_ref(first, count), 0u, 0.0};

        // centroid bounds (big nodes: in parallel chunks; min / max do not depend on the order)
        const int par = count >= kParallelBinMin ? n_threads : 1;
        const size_t n_chunks = (size_t)std::max(par, 1);
        const uint32_t chunk = (uint32_t)((count + n_chunks - 1) / n_chunks);
        float clo[3], chi[3];
        {
            // (no heap traffic for the many small nodes: one chunk lives on the stack)
            float one_lo[3], one_hi[3];
            std::vector<float> many_lo, many_hi;
            if (n_chunks > 1) {
                many_lo.resize(3 * n_chunks);
                many_hi.resize(3 * n_chunks);
            }
            float* plo = n_chunks > 1 ? many_lo.data() : one_lo;
            float* phi = n_chunks > 1 ? many_hi.data() : one_hi;
            for (size_t i = 0; i < 3 * n_chunks; i++) {
                plo[i] = std::numeric_limits<float>::infinity();
                phi[i] = -std::numeric_limits<float>::infinity();
            }
            parallel_for(n_chunks, par, 1, [&](size_t c) {
                float lo[3] = {plo[3 * c], plo[3 * c + 1], plo[3 * c + 2]}, hi[3] = {phi[3 * c], phi[3 * c + 1], phi[3 * c + 2]};
                const uint32_t i1 = std::min<uint32_t>(count, (uint32_t)(c + 1) * chunk);
                for (uint32_t i = (uint32_t)c * chunk; i < i1; i++) {
                    const float* ce = &centroid[3 * (size_t)order[first + i]];
                    for (int a = 0; a < 3; a++) {
                        lo[a] = std::min(lo[a], ce[a]);
                        hi[a] = std::max(hi[a], ce[a]);
                    }
                }
                for (int a = 0; a < 3; a++) {
                    plo[3 * c + a] = lo[a];
                    phi[3 * c + a] = hi[a];
                }
            });
            for (int a = 0; a < 3; a++) {
                clo[a] = plo[a];
                chi[a] = phi[a];
                for (size_t c = 1; c < n_chunks; c++) {
                    clo[a] = std::min(clo[a], plo[3 * c + a]);
                    chi[a] = std::max(chi[a], phi[3 * c + a]);
                }
            }
        }

        uint32_t mid = 0;
        // depth budget: once the remaining levels are only just enough for a balanced split, stop using SAH
        const bool must_balance = depth + 1 + min_depth((count + 1) / 2) >= max_depth;
        if (!must_balance) {
            float best_cost = std::numeric_limits<float>::infinity();
            int best_axis = -1, best_bin = -1;
            // all three axes binned in one pass over the triangles (per chunk, then merged: box unions and counts
            // are order independent)
            struct Bins {
                Box bb[3][kBins];
                uint32_t bc[3][kBins];
            };
            Bins one_part;
            std::vector<Bins> many_part;
            if (n_chunks > 1) many_part.resize(n_chunks);
            Bins* part = n_chunks > 1 ? many_part.data() : &one_part;
            float kk[3];
            for (int a = 0; a < 3; a++) kk[a] = chi[a] - clo[a] > 0.0f ? (float)kBins / (chi[a] - clo[a]) : 0.0f;
            parallel_for(n_chunks, par, 1, [&](size_t c) {
                Bins& P = part[c];
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < kBins; b++) {
                        P.bb[a][b].reset();
                        P.bc[a][b] = 0;
                    }
                const uint32_t i1 = std::min<uint32_t>(count, (uint32_t)(c + 1) * chunk);
                for (uint32_t i = (uint32_t)c * chunk; i < i1; i++) {
                    const uint32_t t = order[first + i];
                    for (int a = 0; a < 3; a++) {
                        if (!(kk[a] > 0.0f)) continue;
                        int bin = (int)((centroid[3 * (size_t)t + a] - clo[a]) * kk[a]);
                        bin = std::min(std::max(bin, 0), kBins - 1);
                        P.bb[a][bin].grow(tri_box[t]);
                        P.bc[a][bin]++;
                    }
                }
            });
            for (int a = 0; a < 3; a++) {
                if (!(kk[a] > 0.0f)) continue;
                Box bb[kBins];
                uint32_t bc[kBins];
                for (int b = 0; b < kBins; b++) {
                    bb[b] = part[0].bb[a][b];
                    bc[b] = part[0].bc[a][b];
                    for (size_t c = 1; c < n_chunks; c++) {
                        bb[b].grow(part[c].bb[a][b]);
                        bc[b] += part[c].bc[a][b];
                    }
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
        Sub s0, s1;
        bool forked = false;
        if (spare_threads && count >= kForkMin) {
            if (spare_threads->fetch_sub(1) > 0) {
                std::exception_ptr left_err;
                std::thread left;
                try {
                    left = std::thread([&]() noexcept {
                        try {
                            s0 = build(first, mid, depth + 1, me + 1u);
                        } catch (...) {
                            left_err = std::current_exception();
                        }
                    });
                    forked = true;
                } catch (const std::system_error&) {  // thread limit reached: this subtree is built on the calling thread
                }
                if (forked) {
                    {
                        struct Join {  // the right-hand build may throw while `left` is still running
                            std::thread& t;
                            ~Join() {
                                if (t.joinable()) t.join();
                            }
                        } join{left};
                        try {
                            s1 = build(first + mid, count - mid, depth + 1, me + mid);
                        } catch (...) {
                            spare_threads->fetch_add(1);
                            throw;
                        }
                    }
                    spare_threads->fetch_add(1);
                    if (left_err) std::rethrow_exception(left_err);
                } else {
                    spare_threads->fetch_add(1);
                }
            } else {
                spare_threads->fetch_add(1);  // undo the failed reservation
            }
        }
        if (!forked) {
            s0 = build(first, mid, depth + 1, me + 1u);
            s1 = build(first + mid, count - mid, depth + 1, me + mid);
        }
        const int32_t r0 = s0.ref, r1 = s1.ref;
        float* n = &nodes[(size_t)me * 16];
        const float p = pad;
        n[0] = b0.lo[0] - p; n[1] = b0.lo[1] - p; n[2] = b0.lo[2] - p; n[3] = b0.hi[0] + p;
        n[4] = b0.hi[1] + p; n[5] = b0.hi[2] + p; n[6] = b1.lo[0] - p; n[7] = b1.lo[1] - p;
        n[8] = b1.lo[2] - p; n[9] = b1.hi[0] + p; n[10] = b1.hi[1] + p; n[11] = b1.hi[2] + p;
        std::memcpy(&n[12], &r0, 4);
        std::memcpy(&n[13], &r1, 4);
        n[14] = n[15] = 0.0f;
        // tree-shaped sums: the same value whatever the thread count
        return Sub{(int32_t)me, std::max(depth, std::max(s0.depth_reached, s1.depth_reached)),
                   ((double)b0.half_area() + (double)b1.half_area()) + (s0.sah + s1.sah)};
    }
    static constexpr uint32_t kForkMin = 1u << 14;        // triangles below which a subtree is not worth a thread
    static constexpr uint32_t kParallelBinMin = 1u << 17;  // triangles from which one node's binning is split over threads
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
// Which binary nodes become 8-wide nodes and how the 8 (every leaf is one triangle: bvh_build.h)
// child slots of every node are spent is chosen by the dynamic program of Ylitie et al. 2017 (§4.1):
//   C(n,1)   = min( A(n) P(n) c_prim  [P(n) == 1],   A(n) c_node + D(n,8) )
//   C(n,i>1) = min( D(n,i), C(n,i-1) ),   D(n,i) = min_{0<k<i} C(left,k) + C(right,i-k)
// Nodes are emitted breadth-first so that the inner children of a node are consecutive
// (child_base + rank among inner slots) and the triangles of its leaf children are consecutive
// (tri_base + rank among leaf slots).
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
    int max_threads = 0;  // 0 = every CPU the process may use
    int thread_count() const { return max_threads > 0 ? std::min(max_threads, host_threads()) : host_threads(); }

    void child_box(int32_t node2, int side, float out[6]) const { std::memcpy(out, &n2[(size_t)node2 * 16 + 6 * side], 24); }
    int32_t child_ref(int32_t node2, int side) const {
        int32_t r;
        std::memcpy(&r, &n2[(size_t)node2 * 16 + 12 + side], 4);
        return r;
    }

    // The binary nodes are numbered in preorder (a subtree is a contiguous index range, children have
    // larger indices than their parent), so disjoint subtrees are solved by different threads and the few
    // nodes above them afterwards.
    void solve() {
        const size_t n_nodes = n2.size() / 16;
        dp.resize(n_nodes);
        const int threads = thread_count();
        struct Range {
            size_t lo, hi;
        };
        std::vector<Range> subtrees;
        std::vector<size_t> top;
        const size_t grain = std::max<size_t>(n_nodes / (size_t)(8 * threads), 4096);
        std::vector<Range> todo{Range{0, n_nodes}};
        while (!todo.empty()) {
            const Range r = todo.back();
            todo.pop_back();
            if (r.hi - r.lo <= grain || threads == 1) {
                subtrees.push_back(r);
                continue;
            }
            top.push_back(r.lo);
            const int32_t r0 = child_ref((int32_t)r.lo, 0), r1 = child_ref((int32_t)r.lo, 1);
            if (r0 >= 0 && r1 >= 0) {
                todo.push_back(Range{(size_t)r0, (size_t)r1});
                todo.push_back(Range{(size_t)r1, r.hi});
            } else if (r0 >= 0) {
                todo.push_back(Range{(size_t)r0, r.hi});
            } else if (r1 >= 0) {
                todo.push_back(Range{(size_t)r1, r.hi});
            }
        }
        parallel_for(subtrees.size(), threads, 1, [&](size_t i) {
            for (size_t jj = subtrees[i].hi; jj-- > subtrees[i].lo;) solve_node(jj);
        });
        std::sort(top.begin(), top.end());
        for (size_t i = top.size(); i-- > 0;) solve_node(top[i]);
    }

    void solve_node(size_t jj) {
        {
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
            const float c_leaf = d.prims <= 1 ? area * (float)d.prims * kCostPrim : std::numeric_limits<float>::infinity();  // leaves are single triangles
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
    // One 8-wide node being formed: its children, their slots, and where its inner children and leaf
    // triangles go.  Nodes are emitted level by level: plan() (children + slots, independent per node) in
    // parallel, a prefix sum over the level in index order for child_base / tri_base, write() in
    // parallel.  The layout is the breadth-first one a sequential queue produces, whatever the thread count.
    struct Work {
        Pending pd;
        Child ch[8];
        int k;
        int child_in[8];  // slot -> child, -1 = empty
        float lo[3], hi[3];
        uint32_t n_inner, n_tris;
        uint32_t child_base, tri_base;
    };

    void plan(const Pending& pd, Work& w) const {
        w.pd = pd;
        Child* ch = w.ch;
        int k = 0;
        if (dp[pd.node2].is_leaf && pd.level == 0) {  // whole mesh fits one leaf: root node with a single leaf child
            const Dp& d = dp[pd.node2];
            std::memcpy(ch[0].box, d.box, 24);
            ch[0].ref = ~(int32_t)((d.first << 2) | (d.prims - 1u));
            k = 1;
        } else {
            distribute(pd.node2, 8, ch, k);
        }
        w.k = k;
        float* lo = w.lo;
        float* hi = w.hi;
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
        int slot_of[8];
        int* child_in = w.child_in;
        for (int i = 0; i < 8; i++) slot_of[i] = child_in[i] = -1;
        float score[8][8];
        for (int i = 0; i < k; i++) {
            float off[3];
            for (int a = 0; a < 3; a++) off[a] = 0.5f * (ch[i].box[a] + ch[i].box[3 + a]) - 0.5f * (lo[a] + hi[a]);
            for (int s = 0; s < 8; s++) {
                float c = 0.0f;
                for (int a = 0; a < 3; a++) c += ((s >> (2 - a)) & 1) ? off[a] : -off[a];
                score[i][s] = c;
            }
        }
        for (int round = 0; round < k; round++) {
            float best = -std::numeric_limits<float>::infinity();
            int bi = -1, bs = -1;
            for (int i = 0; i < k; i++) {
                if (slot_of[i] >= 0) continue;
                for (int s = 0; s < 8; s++) {
                    if (child_in[s] >= 0) continue;
                    if (score[i][s] > best) {
                        best = score[i][s];
                        bi = i;
                        bs = s;
                    }
                }
            }
            slot_of[bi] = bs;
            child_in[bs] = bi;
        }
        w.n_inner = w.n_tris = 0;
        for (int s = 0; s < 8; s++) {
            const int i = child_in[s];
            if (i < 0) continue;
            if (ch[i].ref >= 0) w.n_inner++;
            else w.n_tris += (~(uint32_t)ch[i].ref & 3u) + 1u;
        }
    }

    // node words, leaf triangle order and the next level's entries of one planned node (storage is sized by the caller)
    void write(const Work& wk, Pending* next_level, uint32_t next_level_base) {
        const Child* ch = wk.ch;
        const int* child_in = wk.child_in;
        const float* lo = wk.lo;
        const float* hi = wk.hi;
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
        uint32_t* w = &nodes[(size_t)wk.pd.index * 20];
        std::memcpy(w, lo, 12);
        uint32_t imask = 0;
        for (int s = 0; s < 8; s++)
            if (child_in[s] >= 0 && ch[child_in[s]].ref >= 0) imask |= 1u << s;
        w[3] = e_byte[0] | (e_byte[1] << 8) | (e_byte[2] << 16) | (imask << 24);
        w[4] = wk.child_base;
        w[5] = wk.tri_base;
        uint8_t q[6][8];
        uint32_t leafmask = 0;
        for (int a = 0; a < 6; a++)
            for (int s = 0; s < 8; s++) q[a][s] = a < 3 ? 255 : 0;  // empty slot: inverted box, in neither mask
        uint32_t rank = 0, off = 0;
        for (int s = 0; s < 8; s++) {
            const int i = child_in[s];
            if (i < 0) continue;
            if (ch[i].ref >= 0) {
                next_level[wk.child_base + rank - next_level_base] = Pending{ch[i].ref, wk.child_base + rank, wk.pd.level + 1};
                rank++;
            } else {
                const uint32_t ref = ~(uint32_t)ch[i].ref, first = ref >> 2;  // one triangle per leaf
                leafmask |= 1u << s;
                order8[(size_t)wk.tri_base + off] = order2[first];
                off += 1;
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
        w[6] = leafmask;
        w[7] = 0;
        for (int a = 0; a < 6; a++) std::memcpy(&w[8 + 2 * a], q[a], 8);
    }

    void build() {
        RT_BVH_T(t0);
        solve();
        RT_BVH_T(t1);
        RT_BVH_REPORT("  collapse: DP", t0, t1);
        const int threads = thread_count();
        nodes.clear();
        order8.clear();
        std::vector<Pending> level{Pending{0, 0, 0}}, next;
        std::vector<Work> work;
        uint32_t node_count = 1, tri_count = 0;
        depth = 0;
        while (!level.empty()) {
            depth = level[0].level + 1;
            work.resize(level.size());
            parallel_for(level.size(), threads, 256, [&](size_t i) { plan(level[i], work[i]); });
            const uint32_t next_base = node_count;
            for (Work& w : work) {  // index order = the order a sequential breadth-first queue visits them
                w.child_base = node_count;
                w.tri_base = tri_count;
                node_count += w.n_inner;
                tri_count += w.n_tris;
            }
            nodes.resize((size_t)node_count * 20, 0u);
            order8.resize(tri_count);
            next.resize(node_count - next_base);
            parallel_for(work.size(), threads, 256, [&](size_t i) { write(work[i], next.data(), next_base); });
            level.swap(next);
        }
    }
};

}  // namespace

bool build_bvh(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t max_depth, BvhResult* out) {
    if (!v0 || !e1 || !e2 || n == 0 || n >= (1u << 29) || !out) return false;
    RT_BVH_T(t_start);
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
    // conservative padding: absorbs the rounding of the slab test and of Moeller-Trumbore's t
    out->maxabs = std::max(maxabs, 1.0f);
    out->pad = out->pad_in >= 0.0f ? out->pad_in : 2e-5f * out->maxabs;
    const uint32_t leaf_max = 1;  // the binary tree goes down to single triangles; the collapse keeps them as one-triangle leaves
    const int threads = out->max_threads > 0 ? std::min(out->max_threads, host_threads()) : host_threads();
    std::atomic<int> spare_threads{threads - 1};
    Builder b{tri_box, centroid, out->order, binary, out->pad, max_depth, leaf_max, &spare_threads, threads};
    out->sah_area = 0.0;
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
        RT_BVH_T(t_prep);
        RT_BVH_REPORT("prepare", t_start, t_prep);
        binary.resize(16 * (size_t)(n - 1));  // single-triangle leaves: exactly n - 1 inner nodes
        const Builder::Sub top = b.build(0, n, 0, 0);
        out->depth = top.depth_reached + 1;
        out->sah_area = top.sah;
        RT_BVH_T(t_bin);
        RT_BVH_REPORT("binary SAH build", t_prep, t_bin);
    }
    // collapse to compressed 8-wide nodes; triangles are re-ordered so every node's leaf triangles are contiguous
    std::vector<uint32_t> order2;
    order2.swap(out->order);
    out->order.reserve(n);
    Cw8Builder cw{binary, order2, out->nodes, out->order, 0, out->cost_prim, {}, threads};
    RT_BVH_T(t_c0);
    cw.build();
    RT_BVH_T(t_c1);
    RT_BVH_REPORT("collapse + emit", t_c0, t_c1);
    if (out->order.size() != n) return false;
    out->n_nodes = (uint32_t)(out->nodes.size() / 20);
    out->depth = cw.depth;
    out->stack_need = cw.depth + 1;  // at most one pending sibling group per level
    return true;
}

}  // namespace rt
