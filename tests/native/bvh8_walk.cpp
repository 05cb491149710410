// bvh8_walk.cpp — TEST-ONLY host-side walk of the compressed 8-wide BVH (layout: csrc/bvh_build.h), written
// from the layout and the traversal rules of DESIGN.md section 6.3 / 6.7 with scalar per-child code.  It
// builds the tree with the product's own builder (csrc/bvh_build.cpp, deterministic) from the same
// triangles, traces the same rays and writes, per ray, how many nodes it fetched and how many
// triangles it tested, plus the hit.  tests/test_gpu_path_b.py compares these numbers with
// rt_trace_rays_counted: the device's traversal statistics (rt_pt_stats.nodes_visited / tris_tested, the
// counts bench.py's roofline quotes) are cross-checked by an independent program, not self-certified.
// No reference counterpart (the reference has no triangles or BVH).
//
//   bvh8_walk <in.bin> <out.bin>
//   in : u32 n_tris, u32 n_rays, u32 any_hit | chunks << 8 (chunks > 0: two-level build), f32 verts[n_tris*9], f32 origins[n_rays*3], f32 dirs[n_rays*3]
//   out: per ray  u32 nodes, u32 tris, f32 t (closest: distance or inf; any: 1/0), i32 tri (id or -1; any: 1/0)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../raytracing_engine_amd/csrc/bvh_build.h"

namespace {

struct V3 { float x, y, z; };
inline float dot3(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
inline V3 cross3(V3 a, V3 b) { return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))}; }
inline float safe(float d) { return std::fabs(d) > 1e-20f ? d : std::copysign(1e-20f, d); }

// Moeller-Trumbore with the division deferred
bool tri_test(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float* t) {
    const V3 pvec = cross3(d, e2);
    const float det = dot3(e1, pvec);
    if (det == 0.0f) return false;
    const V3 tvec = {o.x - v0.x, o.y - v0.y, o.z - v0.z};
    const float u = dot3(tvec, pvec);
    const V3 qvec = cross3(tvec, e1);
    const float v = dot3(d, qvec);
    if (det > 0.0f) {
        if (u < 0.0f || v < 0.0f || u + v > det) return false;
    } else {
        if (u > 0.0f || v > 0.0f || u + v < det) return false;
    }
    *t = dot3(e2, qvec) / det;
    return true;
}

struct Tri { V3 v0, e1, e2; uint32_t id; };

struct Pending {  // sibling inner children still to visit: node index and entry order key
    uint32_t child_base, imask;
    uint8_t order[8];  // slots in visiting order
    int n, next;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3) return 2;
    const uint32_t n = hdr[0], n_rays = hdr[1], any_hit = hdr[2] & 0xffu, chunks = hdr[2] >> 8;  // chunks > 0: two-level build
    std::vector<float> verts((size_t)n * 9), org((size_t)n_rays * 3), dir((size_t)n_rays * 3);
    if (std::fread(verts.data(), 4, verts.size(), f) != verts.size() || std::fread(org.data(), 4, org.size(), f) != org.size() ||
        std::fread(dir.data(), 4, dir.size(), f) != dir.size())
        return 2;
    std::fclose(f);

    std::vector<float> v0((size_t)n * 3), e1((size_t)n * 3), e2((size_t)n * 3);
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {  // edges are formed once, in fp32
            v0[3 * i + a] = verts[9 * i + a];
            e1[3 * i + a] = verts[9 * i + 3 + a] - verts[9 * i + a];
            e2[3 * i + a] = verts[9 * i + 6 + a] - verts[9 * i + a];
        }
    rt::BvhResult bvh;
    rt::TwoLevelBvh tl;
    if (chunks ? !rt::build_bvh_two_level(v0.data(), e1.data(), e2.data(), n, chunks, rt::kBvhMaxDepth, &tl, &bvh)
               : !rt::build_bvh(v0.data(), e1.data(), e2.data(), n, rt::kBvhMaxDepth, &bvh))
        return 3;
    std::vector<Tri> tris(n);
    for (size_t li = 0; li < n; li++) {
        const uint32_t t = bvh.order[li];
        tris[li] = {{v0[3 * t], v0[3 * t + 1], v0[3 * t + 2]}, {e1[3 * t], e1[3 * t + 1], e1[3 * t + 2]}, {e2[3 * t], e2[3 * t + 1], e2[3 * t + 2]}, t};
    }

    FILE* out = std::fopen(argv[2], "wb");
    if (!out) return 2;
    const float kShadowTmax = 0.999f;
    for (uint32_t r = 0; r < n_rays; r++) {
        const V3 o = {org[3 * r], org[3 * r + 1], org[3 * r + 2]}, d = {dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]};
        const V3 inv = {1.0f / safe(d.x), 1.0f / safe(d.y), 1.0f / safe(d.z)};
        const V3 noi = {-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)};
        const bool pos[3] = {!std::signbit(d.x), !std::signbit(d.y), !std::signbit(d.z)};  // by sign bit: 1 / -0.0 is negative
        const uint32_t oct_inv = (pos[0] ? 4u : 0u) | (pos[1] ? 2u : 0u) | (pos[2] ? 1u : 0u);  // 7 - octant
        float tmax = any_hit ? kShadowTmax : INFINITY;
        float best_t = INFINITY;
        uint32_t best_id = 0xffffffffu;
        bool found = false, occluded = false;
        uint32_t n_nodes = 0, n_tris = 0;

        std::vector<Pending> stack;
        uint32_t node = 0;  // the root
        bool have_node = true;
        while (have_node && !occluded) {
            const uint32_t* w = &bvh.nodes[(size_t)node * 20];
            n_nodes++;
            float p[3], s[3];
            std::memcpy(p, w, 12);
            for (int a = 0; a < 3; a++) {
                const uint32_t bits = ((w[3] >> (8 * a)) & 0xffu) << 23;
                std::memcpy(&s[a], &bits, 4);
            }
            const uint32_t imask = w[3] >> 24, child_base = w[4], tri_base = w[5];
            const uint32_t leafmask = w[6] & 0xffu;  // bit s: slot s is a leaf = the one triangle tri_base + popcount(leafmask below s)
            const uint8_t* q = reinterpret_cast<const uint8_t*>(&w[8]);  // qlo.x[8] qlo.y[8] qlo.z[8] qhi.x[8] qhi.y[8] qhi.z[8]
            const float iv[3] = {inv.x, inv.y, inv.z}, nv[3] = {noi.x, noi.y, noi.z};
            float a_[3], b_[3];
            for (int a = 0; a < 3; a++) {  // plane t = q * (s * inv) + (p * inv - o * inv)
                a_[a] = s[a] * iv[a];
                b_[a] = fmaf(p[a], iv[a], nv[a]);
            }
            const float tlim = tmax;  // the box padding is the slack (see node_step in csrc/path_b.hip)
            uint32_t inner_hit = 0;   // bit (slot ^ oct_inv): inner child in `slot` was hit
            uint32_t leaf_hit = 0;    // bit slot: the leaf in `slot` was hit
            for (int slot = 0; slot < 8; slot++) {
                const bool inner = (imask >> slot) & 1u, leaf = (leafmask >> slot) & 1u;
                if (!inner && !leaf) continue;  // empty
                float tn = 0.0f, tf = tlim;
                for (int a = 0; a < 3; a++) {
                    const float lo = (float)q[8 * a + slot], hi = (float)q[24 + 8 * a + slot];
                    const float t_near = fmaf(pos[a] ? lo : hi, a_[a], b_[a]), t_far = fmaf(pos[a] ? hi : lo, a_[a], b_[a]);
                    tn = std::fmax(tn, t_near);
                    tf = std::fmin(tf, t_far);
                }
                if (std::signbit(tf - tn)) continue;  // the kernels collect sign bits of tf - tn (node_step in csrc/path_b.hip)
                if (inner) inner_hit |= 1u << ((uint32_t)slot ^ oct_inv);
                else leaf_hit |= 1u << slot;
            }
            // leaf triangles first, in ascending slot order
            for (uint32_t k = 0; k < 8 && !occluded; k++) {
                if (!((leaf_hit >> k) & 1u)) continue;
                const Tri& t = tris[tri_base + (uint32_t)__builtin_popcount(leafmask & ((1u << k) - 1u))];
                n_tris++;
                float tt;
                if (tri_test(o, d, t.v0, t.e1, t.e2, &tt) && tt > 0.0f) {
                    if (any_hit) {
                        if (tt < kShadowTmax) occluded = true;
                    } else if (tt < best_t || (tt == best_t && t.id < best_id)) {
                        best_t = tt;
                        best_id = t.id;
                        found = true;
                        tmax = tt;
                    }
                }
            }
            if (occluded) break;
            // inner children front to back: descending key (slot ^ oct_inv)
            Pending pd{child_base, imask, {}, 0, 0};
            for (int key = 7; key >= 0; key--)
                if ((inner_hit >> key) & 1u) pd.order[pd.n++] = (uint8_t)((uint32_t)key ^ oct_inv);
            if (pd.n) stack.push_back(pd);
            have_node = false;
            while (!stack.empty()) {
                Pending& top = stack.back();
                const uint32_t slot = top.order[top.next++];
                node = top.child_base + (uint32_t)__builtin_popcount(top.imask & ((1u << slot) - 1u));
                if (top.next == top.n) stack.pop_back();
                have_node = true;
                break;
            }
        }
        const float t_out = any_hit ? (occluded ? 1.0f : 0.0f) : best_t;
        const int32_t tri_out = any_hit ? (occluded ? 1 : 0) : (found ? (int32_t)best_id : -1);
        std::fwrite(&n_nodes, 4, 1, out);
        std::fwrite(&n_tris, 4, 1, out);
        std::fwrite(&t_out, 4, 1, out);
        std::fwrite(&tri_out, 4, 1, out);
    }
    std::fclose(out);
    return 0;
}
