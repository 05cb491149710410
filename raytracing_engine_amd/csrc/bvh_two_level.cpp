// bvh_two_level.cpp — two-level BVH build for path B: a top-level BVH8 over per-chunk bottom-level BVH8s
// (BASELINE.json configs[2] "2-level BVH"; SURVEY.md section 8d config 3: TLAS over 64 BLAS chunks).  No reference
// counterpart (the reference has no triangles).
//
// Chunks are the leaves of a binned-SAH cut: the triangle set is split top-down with the same 16-bin SAH rule the
// bottom-level builder uses, always splitting the leaf that holds the most triangles, until there are `chunks` leaves.
// (Round 2 cut the centroids' Morton order into equal runs: those chunk boxes overlap, a shadow ray entered 20 % more
// nodes and traversal was 16-21 % slower than on the single SAH tree; profiles/r02_two_level_bvh.txt.)
//
// The two levels are FLATTENED into the one node array the traversal kernels read (layout: bvh_build.h): a top-level
// leaf (= a chunk) becomes an inner child slot whose node is the chunk's root, bottom-level nodes are copied with their
// child / triangle bases relocated.  Frames cannot differ from the single-level build's (results do not depend on the
// tree); what differs is build cost structure: rebuild_chunk() redoes one chunk's binned-SAH build
// (1/chunks of the triangles), the top level over the chunk boxes and the flatten copy.
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <system_error>
#include <thread>

#include "bvh_build.h"

namespace rt {
namespace {

using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

int host_cpus() {
    cpu_set_t set;
    int hw = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::max(1, std::min(CPU_COUNT(&set), 32));
    return hw;
}

// fn(t) for t in [0, parts) on up to `parts` threads; a thread that cannot be created runs on the caller
template <typename F>
void run_parts(uint32_t parts, F fn) {
    std::vector<std::thread> pool;
    uint32_t started = 1;
    for (uint32_t t = 1; t < parts; t++) {
        try {
            pool.emplace_back(fn, t);
            started++;
        } catch (const std::system_error&) {
            break;
        }
    }
    fn(0u);
    for (auto& th : pool) th.join();
    for (uint32_t t = started; t < parts; t++) fn(t);
}

// One binned-SAH split of order[first, first + count) (16 bins per axis over the centroid bounds, cost = half-area x
// count on either side): partitions the range and returns the size of the left part (1 .. count - 1).  Box unions,
// counts and minima are order independent, so the result does not depend on the number of threads.
constexpr int kCutBins = 16;
uint32_t sah_split(std::vector<uint32_t>& order, uint32_t first, uint32_t count, const std::vector<float>& tb /* 6 per triangle */,
                   const std::vector<float>& cen, int threads) {
    const uint32_t parts = count >= (1u << 16) ? (uint32_t)std::max(1, threads) : 1u;
    const uint32_t slice = (count + parts - 1) / parts;
    struct Acc {
        float clo[3], chi[3];
        float lo[3][kCutBins][3], hi[3][kCutBins][3];
        uint32_t cnt[3][kCutBins];
    };
    std::vector<Acc> acc(parts);
    run_parts(parts, [&](uint32_t t) {
        Acc& A = acc[t];
        for (int a = 0; a < 3; a++) {
            A.clo[a] = INFINITY;
            A.chi[a] = -INFINITY;
        }
        const uint32_t i1 = std::min(count, (t + 1) * slice);
        for (uint32_t i = t * slice; i < i1; i++) {
            const float* c = &cen[3 * (size_t)order[first + i]];
            for (int a = 0; a < 3; a++) {
                A.clo[a] = std::min(A.clo[a], c[a]);
                A.chi[a] = std::max(A.chi[a], c[a]);
            }
        }
    });
    float clo[3], chi[3], kk[3];
    for (int a = 0; a < 3; a++) {
        clo[a] = acc[0].clo[a];
        chi[a] = acc[0].chi[a];
        for (uint32_t t = 1; t < parts; t++) {
            clo[a] = std::min(clo[a], acc[t].clo[a]);
            chi[a] = std::max(chi[a], acc[t].chi[a]);
        }
        kk[a] = chi[a] - clo[a] > 0.0f ? (float)kCutBins / (chi[a] - clo[a]) : 0.0f;
    }
    auto bin_of = [&](uint32_t tri, int a) {
        const int b = (int)((cen[3 * (size_t)tri + a] - clo[a]) * kk[a]);
        return std::min(std::max(b, 0), kCutBins - 1);
    };
    run_parts(parts, [&](uint32_t t) {
        Acc& A = acc[t];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < kCutBins; b++) {
                A.cnt[a][b] = 0;
                for (int k = 0; k < 3; k++) {
                    A.lo[a][b][k] = INFINITY;
                    A.hi[a][b][k] = -INFINITY;
                }
            }
        const uint32_t i1 = std::min(count, (t + 1) * slice);
        for (uint32_t i = t * slice; i < i1; i++) {
            const uint32_t tri = order[first + i];
            const float* bx = &tb[6 * (size_t)tri];
            for (int a = 0; a < 3; a++) {
                if (!(kk[a] > 0.0f)) continue;
                const int b = bin_of(tri, a);
                A.cnt[a][b]++;
                for (int k = 0; k < 3; k++) {
                    A.lo[a][b][k] = std::min(A.lo[a][b][k], bx[k]);
                    A.hi[a][b][k] = std::max(A.hi[a][b][k], bx[3 + k]);
                }
            }
        }
    });
    float best = INFINITY;
    int best_axis = -1, best_bin = -1;
    auto half_area = [](const float lo[3], const float hi[3]) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    for (int a = 0; a < 3; a++) {
        if (!(kk[a] > 0.0f)) continue;
        float lo[kCutBins][3], hi[kCutBins][3];
        uint32_t cnt[kCutBins];
        for (int b = 0; b < kCutBins; b++) {
            cnt[b] = 0;
            for (int k = 0; k < 3; k++) {
                lo[b][k] = INFINITY;
                hi[b][k] = -INFINITY;
            }
            for (uint32_t t = 0; t < parts; t++) {
                cnt[b] += acc[t].cnt[a][b];
                for (int k = 0; k < 3; k++) {
                    lo[b][k] = std::min(lo[b][k], acc[t].lo[a][b][k]);
                    hi[b][k] = std::max(hi[b][k], acc[t].hi[a][b][k]);
                }
            }
        }
        float r_area[kCutBins];
        uint32_t r_cnt[kCutBins];
        float alo[3] = {INFINITY, INFINITY, INFINITY}, ahi[3] = {-INFINITY, -INFINITY, -INFINITY};
        uint32_t c = 0;
        for (int b = kCutBins - 1; b > 0; b--) {
            for (int k = 0; k < 3; k++) {
                alo[k] = std::min(alo[k], lo[b][k]);
                ahi[k] = std::max(ahi[k], hi[b][k]);
            }
            c += cnt[b];
            r_area[b] = c ? half_area(alo, ahi) : 0.0f;
            r_cnt[b] = c;
        }
        for (int k = 0; k < 3; k++) {
            alo[k] = INFINITY;
            ahi[k] = -INFINITY;
        }
        c = 0;
        for (int b = 0; b < kCutBins - 1; b++) {
            for (int k = 0; k < 3; k++) {
                alo[k] = std::min(alo[k], lo[b][k]);
                ahi[k] = std::max(ahi[k], hi[b][k]);
            }
            c += cnt[b];
            if (c == 0 || r_cnt[b + 1] == 0) continue;
            const float cost = half_area(alo, ahi) * (float)c + r_area[b + 1] * (float)r_cnt[b + 1];
            if (cost < best) {
                best = cost;
                best_axis = a;
                best_bin = b;
            }
        }
    }
    uint32_t mid = 0;
    if (best_axis >= 0) {
        auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t tri) { return bin_of(tri, best_axis) <= best_bin; });
        mid = (uint32_t)(it - (order.begin() + first));
    }
    if (mid == 0 || mid == count) {  // all centroids in one bin: median split on the longest centroid axis, ties by index
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        mid = count / 2;
        std::nth_element(order.begin() + first, order.begin() + first + mid, order.begin() + first + count, [&](uint32_t x, uint32_t y) {
            const float cx = cen[3 * (size_t)x + axis], cy = cen[3 * (size_t)y + axis];
            return cx < cy || (cx == cy && x < y);
        });
    }
    return mid;
}

// conservative world-space box of a built BVH = the root node's quantisation frame [p, p + 255 * scale]
void root_box(const BvhResult& b, float lo[3], float hi[3]) {
    const uint32_t* w = b.nodes.data();
    for (int a = 0; a < 3; a++) {
        float p, s;
        std::memcpy(&p, &w[a], 4);
        const uint32_t bits = ((w[3] >> (8 * a)) & 0xffu) << 23;
        std::memcpy(&s, &bits, 4);
        lo[a] = p;
        hi[a] = p + 255.0f * s;
    }
}

bool build_one_chunk(const float* v0, const float* e1, const float* e2, const TwoLevelBvh& tl, uint32_t c, uint32_t max_depth, int max_threads, BvhResult* out) {
    const uint32_t lo = tl.first[c], n = tl.first[c + 1] - lo;
    std::vector<float> cv0(3 * (size_t)n), ce1(3 * (size_t)n), ce2(3 * (size_t)n);
    for (uint32_t i = 0; i < n; i++) {
        const size_t t = tl.sorted[lo + i];
        std::memcpy(&cv0[3 * (size_t)i], &v0[3 * t], 12);
        std::memcpy(&ce1[3 * (size_t)i], &e1[3 * t], 12);
        std::memcpy(&ce2[3 * (size_t)i], &e2[3 * t], 12);
    }
    *out = BvhResult{};
    out->pad_in = tl.pad;  // every chunk is padded for the whole mesh's coordinate range
    out->max_threads = max_threads;
    return build_bvh(cv0.data(), ce1.data(), ce2.data(), n, max_depth, out);
}

// top level over the chunk boxes + flatten everything into `out`
bool assemble(TwoLevelBvh* tl, uint32_t n, uint32_t max_depth, BvhResult* out) {
    const uint32_t chunks = (uint32_t)tl->blas.size();
    auto t0 = Clock::now();
    // a chunk enters the top-level builder as a degenerate "triangle" whose bounding box is the chunk's box
    std::vector<float> bv0(3 * (size_t)chunks), be1(3 * (size_t)chunks), be2(3 * (size_t)chunks, 0.0f);
    for (uint32_t c = 0; c < chunks; c++) {
        float lo[3], hi[3];
        root_box(tl->blas[c], lo, hi);
        for (int a = 0; a < 3; a++) {
            bv0[3 * (size_t)c + a] = lo[a];
            // v0 + e1 must not fall short of hi after rounding
            float e = hi[a] - lo[a];
            while (lo[a] + e < hi[a]) e = std::nextafter(e, std::numeric_limits<float>::infinity());
            be1[3 * (size_t)c + a] = e;
        }
    }
    BvhResult top;
    top.pad_in = 0.0f;  // the chunk boxes are padded already; quantisation rounds outward
    top.max_threads = 1;
    if (!build_bvh(bv0.data(), be1.data(), be2.data(), chunks, max_depth, &top)) return false;
    tl->tlas_nodes = top.n_nodes;
    tl->tlas_depth = top.depth;
    tl->ms_tlas = ms_since(t0);

    t0 = Clock::now();
    // global placement, breadth-first over the top level: the children of a top-level node - further top-level nodes and
    // chunk roots alike - occupy consecutive indices in slot order; the non-root nodes of each chunk follow in chunk order
    struct Item {
        uint32_t is_root, index;  // top-level node index | chunk whose root goes here
    };
    std::vector<Item> place{Item{0u, 0u}};
    std::vector<uint32_t> level{1u};  // tree level of place[g] (root = 1)
    std::vector<uint32_t> top_child_base(top.n_nodes, 0u);
    uint32_t depth = 0;  // levels of 8-wide nodes on the longest root-to-leaf path of the flattened tree
    for (size_t g = 0; g < place.size(); g++) {
        if (place[g].is_root) {
            depth = std::max(depth, level[g] + tl->blas[place[g].index].depth - 1u);
            continue;
        }
        depth = std::max(depth, level[g]);
        const uint32_t* w = &top.nodes[(size_t)place[g].index * 20];
        const uint32_t imask = w[3] >> 24, leafmask = w[6] & 0xffu;
        top_child_base[place[g].index] = (uint32_t)place.size();
        for (uint32_t s = 0; s < 8; s++) {
            if ((imask >> s) & 1u) place.push_back(Item{0u, w[4] + (uint32_t)__builtin_popcount(imask & ((1u << s) - 1u))});
            else if ((leafmask >> s) & 1u) place.push_back(Item{1u, top.order[w[5] + (uint32_t)__builtin_popcount(leafmask & ((1u << s) - 1u))]});
            else continue;
            level.push_back(level[g] + 1u);
        }
    }
    std::vector<uint32_t> rest_base(chunks), tri_off(chunks);
    uint32_t n_nodes = (uint32_t)place.size(), n_tris = 0;
    for (uint32_t c = 0; c < chunks; c++) {
        rest_base[c] = n_nodes;
        n_nodes += tl->blas[c].n_nodes - 1u;
        tri_off[c] = n_tris;
        n_tris += (uint32_t)tl->blas[c].order.size();
    }
    if (n_tris != n) return false;
    out->nodes.assign((size_t)n_nodes * 20, 0u);
    out->order.resize(n);
    auto copy_blas_node = [&](uint32_t c, uint32_t local, uint32_t global) {
        const uint32_t* src = &tl->blas[c].nodes[(size_t)local * 20];
        uint32_t* dst = &out->nodes[(size_t)global * 20];
        std::memcpy(dst, src, 80);
        dst[4] = rest_base[c] + src[4] - 1u;  // local child indices are >= 1 (the root is local node 0)
        dst[5] = tri_off[c] + src[5];
    };
    for (size_t g = 0; g < place.size(); g++) {
        if (place[g].is_root) {
            copy_blas_node(place[g].index, 0u, (uint32_t)g);
        } else {
            const uint32_t* src = &top.nodes[(size_t)place[g].index * 20];
            uint32_t* dst = &out->nodes[g * 20];
            std::memcpy(dst, src, 80);
            const uint32_t imask = src[3] >> 24, leafmask = src[6] & 0xffu;
            dst[3] = (src[3] & 0x00ffffffu) | ((imask | leafmask) << 24);  // chunk roots are inner children
            dst[4] = top_child_base[place[g].index];
            dst[5] = 0u;
            dst[6] = 0u;
        }
    }
    for (uint32_t c = 0; c < chunks; c++) {
        const BvhResult& b = tl->blas[c];
        for (uint32_t i = 1; i < b.n_nodes; i++) copy_blas_node(c, i, rest_base[c] + i - 1u);
        for (size_t li = 0; li < b.order.size(); li++) out->order[tri_off[c] + li] = tl->sorted[tl->first[c] + b.order[li]];
    }
    out->n_nodes = n_nodes;
    out->depth = depth;
    out->stack_need = out->depth + 1u;
    out->pad = tl->pad;
    out->maxabs = tl->maxabs;
    out->sah_area = 0.0;
    tl->ms_flatten = ms_since(t0);
    return true;
}

}  // namespace

bool build_bvh_two_level(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t chunks, uint32_t max_depth, TwoLevelBvh* tl, BvhResult* out) {
    if (!v0 || !e1 || !e2 || !tl || !out || n == 0 || chunks == 0) return false;
    chunks = std::min(chunks, std::max(1u, n / 4u));  // on average at least four triangles per chunk
    auto t_cut = Clock::now();
    std::vector<float> cen(3 * (size_t)n), tb(6 * (size_t)n);
    const uint32_t pre_parts = n >= (1u << 16) ? (uint32_t)host_cpus() : 1u;
    std::vector<float> part_max(pre_parts, 0.0f);
    run_parts(pre_parts, [&](uint32_t w) {
        const uint32_t slice = (n + pre_parts - 1) / pre_parts, i1 = std::min(n, (w + 1) * slice);
        float m = 0.0f;
        for (uint32_t i = w * slice; i < i1; i++)
            for (int a = 0; a < 3; a++) {
                const float p0 = v0[3 * (size_t)i + a], p1 = p0 + e1[3 * (size_t)i + a], p2 = p0 + e2[3 * (size_t)i + a];
                cen[3 * (size_t)i + a] = p0 + (e1[3 * (size_t)i + a] + e2[3 * (size_t)i + a]) * (1.0f / 3.0f);
                tb[6 * (size_t)i + a] = std::min(p0, std::min(p1, p2));
                tb[6 * (size_t)i + 3 + a] = std::max(p0, std::max(p1, p2));
                m = std::max(m, std::max(std::fabs(p0), std::max(std::fabs(p1), std::fabs(p2))));
            }
        part_max[w] = m;
    });
    float maxabs = 0.0f;
    for (float m : part_max) maxabs = std::max(maxabs, m);
    // the SAH cut, in rounds: with L leaves, the min(L, chunks - L) fullest ones (ties: the leftmost) are split side by side,
    // until there are `chunks` leaves (ranges of `sorted`, kept in left-to-right order).  A power-of-two chunk count gives
    // the complete binary SAH tree of that depth; the rule does not mention threads, so neither does the result.
    tl->sorted.resize(n);
    for (uint32_t i = 0; i < n; i++) tl->sorted[i] = i;
    std::vector<uint32_t> bounds{0u, n};  // leaf k = [bounds[k], bounds[k + 1])
    const int cpus = host_cpus();
    while (bounds.size() - 1 < chunks) {
        const size_t leaves = bounds.size() - 1;
        std::vector<size_t> pick(leaves);
        for (size_t k = 0; k < leaves; k++) pick[k] = k;
        std::stable_sort(pick.begin(), pick.end(), [&](size_t x, size_t y) { return bounds[x + 1] - bounds[x] > bounds[y + 1] - bounds[y]; });
        size_t take = std::min<size_t>(leaves, chunks - leaves);
        while (take > 0 && bounds[pick[take - 1] + 1] - bounds[pick[take - 1]] < 2) take--;  // a one-triangle leaf cannot be split
        if (take == 0) break;
        pick.resize(take);
        std::vector<uint32_t> mids(take);
        // few big leaves: one after the other, each binned by all threads; many leaves: side by side, one thread each
        const uint32_t side = (uint32_t)std::min<size_t>(take, (size_t)cpus);
        const int inner = std::max(1, cpus / (int)side);
        run_parts(side, [&](uint32_t w) {
            for (size_t j = w; j < take; j += side) {
                const uint32_t first = bounds[pick[j]], count = bounds[pick[j] + 1] - first;
                mids[j] = first + sah_split(tl->sorted, first, count, tb, cen, inner);
            }
        });
        bounds.insert(bounds.end(), mids.begin(), mids.end());
        std::sort(bounds.begin(), bounds.end());
    }
    chunks = (uint32_t)bounds.size() - 1u;
    tl->first = bounds;
    {  // a chunk lists its triangles in ascending index
        const uint32_t side = (uint32_t)std::min<uint32_t>(chunks, (uint32_t)cpus);
        run_parts(side, [&](uint32_t w) {
            for (uint32_t c = w; c < chunks; c += side) std::sort(tl->sorted.begin() + tl->first[c], tl->sorted.begin() + tl->first[c + 1]);
        });
    }
    tl->maxabs = std::max(maxabs, 1.0f);
    tl->pad = 2e-5f * tl->maxabs;
    tl->ms_cut = ms_since(t_cut);

    auto t0 = Clock::now();
    tl->blas.assign(chunks, BvhResult{});
    std::vector<char> ok(chunks, 0);
    {  // chunks side by side, each built by one thread (its result does not depend on threads anyway)
        const uint32_t workers = std::min<uint32_t>((uint32_t)host_cpus(), chunks);
        std::vector<std::thread> pool;
        auto run = [&](uint32_t w) {
            for (uint32_t c = w; c < chunks; c += workers) {
                try {
                    ok[c] = build_one_chunk(v0, e1, e2, *tl, c, max_depth, 1, &tl->blas[c]) ? 1 : 0;
                } catch (...) {
                    ok[c] = 0;
                }
            }
        };
        uint32_t started = 1;
        for (uint32_t w = 1; w < workers; w++) {
            try {
                pool.emplace_back(run, w);
                started++;
            } catch (const std::system_error&) {
                break;
            }
        }
        run(0);
        for (auto& th : pool) th.join();
        for (uint32_t w = started; w < workers; w++) run(w);  // workers whose thread could not be created
    }
    for (uint32_t c = 0; c < chunks; c++)
        if (!ok[c]) return false;
    tl->ms_blas = ms_since(t0);
    return assemble(tl, n, max_depth, out);
}

bool rebuild_chunk(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t chunk, uint32_t max_depth, TwoLevelBvh* tl, BvhResult* out,
                   BvhResult* displaced) {
    if (!v0 || !e1 || !e2 || !tl || !out || chunk >= tl->blas.size() || tl->sorted.size() != n) return false;
    // the padding was chosen for the mesh's coordinate range at build time (maxabs, stored: 2e-5f * M / 2e-5f does not
    // round-trip to M in fp32): moved vertices must stay inside it
    const float reach = tl->maxabs;
    for (uint32_t i = tl->first[chunk]; i < tl->first[chunk + 1]; i++) {
        const size_t t = tl->sorted[i];
        for (int a = 0; a < 3; a++) {
            const float p0 = v0[3 * t + a], p1 = p0 + e1[3 * t + a], p2 = p0 + e2[3 * t + a];
            if (!(std::fabs(p0) <= reach && std::fabs(p1) <= reach && std::fabs(p2) <= reach)) return false;
        }
    }
    auto t0 = Clock::now();
    BvhResult fresh;
    if (!build_one_chunk(v0, e1, e2, *tl, chunk, max_depth, 0, &fresh)) return false;
    // transactional: the chunk's old structure comes back if the top level or the flatten step fails (or throws)
    std::swap(tl->blas[chunk], fresh);
    const double ms_blas = ms_since(t0);
    bool ok = false;
    try {
        ok = assemble(tl, n, max_depth, out);
    } catch (...) {
        std::swap(tl->blas[chunk], fresh);
        throw;
    }
    if (!ok) {
        std::swap(tl->blas[chunk], fresh);
        return false;
    }
    tl->ms_blas = ms_blas;
    if (displaced) *displaced = std::move(fresh);  // the chunk's previous structure: swap it back in to undo this rebuild
    return true;
}

}  // namespace rt
