// bvh_two_level.cpp — two-level BVH build for path B: a top-level BVH8 over per-chunk bottom-level BVH8s
// (BASELINE.json configs[2] "2-level BVH"; SURVEY.md section 8d config 3: TLAS over 64 BLAS chunks, chunks = runs of the
// centroid Morton order).  No reference counterpart (the reference has no triangles).
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

uint32_t spread10(uint32_t x) {  // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
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
    std::vector<uint32_t> top_child_base(top.n_nodes, 0u);
    for (size_t g = 0; g < place.size(); g++) {
        if (place[g].is_root) continue;
        const uint32_t* w = &top.nodes[(size_t)place[g].index * 20];
        const uint32_t imask = w[3] >> 24, leafmask = w[6] & 0xffu;
        top_child_base[place[g].index] = (uint32_t)place.size();
        for (uint32_t s = 0; s < 8; s++) {
            if ((imask >> s) & 1u) place.push_back(Item{0u, w[4] + (uint32_t)__builtin_popcount(imask & ((1u << s) - 1u))});
            else if ((leafmask >> s) & 1u) place.push_back(Item{1u, top.order[w[5] + (uint32_t)__builtin_popcount(leafmask & ((1u << s) - 1u))]});
        }
    }
    std::vector<uint32_t> rest_base(chunks), tri_off(chunks);
    uint32_t n_nodes = (uint32_t)place.size(), n_tris = 0, depth = 0;
    for (uint32_t c = 0; c < chunks; c++) {
        rest_base[c] = n_nodes;
        n_nodes += tl->blas[c].n_nodes - 1u;
        tri_off[c] = n_tris;
        n_tris += (uint32_t)tl->blas[c].order.size();
        depth = std::max(depth, tl->blas[c].depth);
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
    out->depth = top.depth + depth;
    out->stack_need = out->depth + 1u;
    out->pad = tl->pad;
    out->sah_area = 0.0;
    tl->ms_flatten = ms_since(t0);
    return true;
}

}  // namespace

bool build_bvh_two_level(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t chunks, uint32_t max_depth, TwoLevelBvh* tl, BvhResult* out) {
    if (!v0 || !e1 || !e2 || !tl || !out || n == 0 || chunks == 0) return false;
    chunks = std::min(chunks, std::max(1u, n / 4u));  // at least four triangles per chunk
    // Morton order of the centroids over the mesh's bounding box (10 bits per axis), ties by triangle index
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, maxabs = 0.0f;
    std::vector<float> cen(3 * (size_t)n);
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            const float p0 = v0[3 * (size_t)i + a], p1 = p0 + e1[3 * (size_t)i + a], p2 = p0 + e2[3 * (size_t)i + a];
            const float c = p0 + (e1[3 * (size_t)i + a] + e2[3 * (size_t)i + a]) * (1.0f / 3.0f);
            cen[3 * (size_t)i + a] = c;
            lo[a] = std::min(lo[a], c);
            hi[a] = std::max(hi[a], c);
            maxabs = std::max(maxabs, std::max(std::fabs(p0), std::max(std::fabs(p1), std::fabs(p2))));
        }
    std::vector<unsigned long long> keys(n);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t code = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = hi[a] - lo[a];
            const float u = ext > 0.0f ? (cen[3 * (size_t)i + a] - lo[a]) / ext : 0.0f;
            const uint32_t q = (uint32_t)std::min(1023.0f, std::max(0.0f, u * 1024.0f));
            code |= spread10(q) << (2 - a);
        }
        keys[i] = ((unsigned long long)code << 32) | i;
    }
    std::sort(keys.begin(), keys.end());
    tl->sorted.resize(n);
    for (uint32_t i = 0; i < n; i++) tl->sorted[i] = (uint32_t)keys[i];
    tl->first.resize(chunks + 1);
    for (uint32_t c = 0; c <= chunks; c++) tl->first[c] = (uint32_t)((unsigned long long)n * c / chunks);
    tl->pad = 2e-5f * std::max(maxabs, 1.0f);

    auto t0 = Clock::now();
    tl->blas.assign(chunks, BvhResult{});
    std::vector<char> ok(chunks, 0);
    {  // chunks side by side, each built by one thread (its result does not depend on threads anyway)
        cpu_set_t set;
        int hw = 1;
        if (sched_getaffinity(0, sizeof set, &set) == 0) hw = std::max(1, std::min(CPU_COUNT(&set), 32));
        const uint32_t workers = std::min<uint32_t>((uint32_t)hw, chunks);
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

bool rebuild_chunk(const float* v0, const float* e1, const float* e2, uint32_t n, uint32_t chunk, uint32_t max_depth, TwoLevelBvh* tl, BvhResult* out) {
    if (!v0 || !e1 || !e2 || !tl || !out || chunk >= tl->blas.size() || tl->sorted.size() != n) return false;
    // the padding was chosen for the mesh's coordinate range at build time: moved vertices must stay inside it
    const float reach = tl->pad / 2e-5f;
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
    tl->blas[chunk] = std::move(fresh);
    tl->ms_blas = ms_since(t0);
    return assemble(tl, n, max_depth, out);
}

}  // namespace rt
