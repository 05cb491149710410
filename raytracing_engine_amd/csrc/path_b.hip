// path_b.hip — wavefront path tracer over a triangle BVH (BASELINE.json configs[2..4]).
//
// NO REFERENCE COUNTERPART: the reference has no triangles, BVH, RNG, spp or bounces (SURVEY.md
// §0); only the camera model is the reference's (shaders/fragment.glsl:129-133,
// shaders/utilities.glsl:26-29).  This file implements the specification of DESIGN.md §6; parity
// is against oracle B and is "unpinned by the reference".
//
// Structure (one launch per ray stage, all queue sizes stay on the device):
//   pt_generate       camera rays for every (pixel, sample) of the owned tiles -> path state + queue 0
//   pt_trace<closest> persistent waves with per-lane refill from the device-resident queue; traversal
//                     of a compressed 8-wide BVH (80-byte nodes, five 16-byte fetches per node) with a
//                     per-lane stack of node groups in LDS, ray/triangle tests, writes (t, triangle)
//   pt_shade          emission / sky / next-event estimation / cosine bounce; survivors are appended to
//                     the next queue and shadow rays to the shadow queue with wave ballot +
//                     prefix-popcount compaction (one atomic per 1024-thread workgroup)
//   pt_trace<any>     shadow rays: any-hit traversal, unoccluded contributions added to the path
//   pt_resolve        per pixel: samples summed in index order, divided by spp
// Memory: path state is SoA of float4 (16 B per lane per array = widest coalesced access), BVH nodes
// are 80-byte quantised records (bvh_build.h), triangles 48-byte records in leaf order.
#include "rt_device_math.h"
#include "rt_internal.h"

namespace rt {
using namespace rtk;

constexpr float kShadowTmax = 0.999f;

// ---- spec §6.2: counter-based RNG -------------------------------------------------------------
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t path_key(uint32_t pixel, uint32_t sample, uint32_t seed) {
    return hash32(hash32(pixel + hash32(seed)) + sample);
}
__device__ __forceinline__ float rnd(uint32_t key, uint32_t depth, uint32_t dim) {
    const uint32_t h = hash32(key + (depth * 8u + dim + 1u) * 0x9e3779b9U);
    return (float)(h >> 8) * 0x1p-24f;
}

// ---- spec §6.5: sin/cos(2*pi*u) from fma polynomials only ----------------------------------------
__device__ __forceinline__ void sincos_2pi(float u, float& s_out, float& c_out) {
    const float t = u * 4.0f;
    uint32_t q = (uint32_t)t;
    if (q > 3u) q = 3u;
    const float th = ((t - (float)q) - 0.5f) * 1.57079632679f;
    const float th2 = th * th;
    const float ps = __builtin_fmaf(th2, __builtin_fmaf(th2, __builtin_fmaf(th2, __builtin_fmaf(th2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
    const float s = th * ps;
    const float c = __builtin_fmaf(th2, __builtin_fmaf(th2, __builtin_fmaf(th2, __builtin_fmaf(th2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
    const float R = 0.70710678f;
    const float cA = (q == 0u || q == 3u) ? R : -R;
    const float sA = (q < 2u) ? R : -R;
    c_out = __builtin_fmaf(cA, c, -(sA * s));
    s_out = __builtin_fmaf(sA, c, cA * s);
}

__device__ __forceinline__ v3 cosine_dir(v3 n, float u1, float u2) {
    const float r = __builtin_sqrtf(u1);
    float s, c;
    sincos_2pi(u2, s, c);
    const float x = r * c, y = r * s, z = __builtin_sqrtf(fmax_(0.0f, 1.0f - u1));
    const float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    const float a = -1.0f / (sign + n.z);
    const float b = (n.x * n.y) * a;
    const v3 b1 = mk(__builtin_fmaf(sign, (n.x * n.x) * a, 1.0f), sign * b, -sign * n.x);
    const v3 b2 = mk(b, __builtin_fmaf(n.y * n.y, a, sign), -n.y);
    return mk(__builtin_fmaf(n.x, z, __builtin_fmaf(b2.x, y, b1.x * x)), __builtin_fmaf(n.y, z, __builtin_fmaf(b2.y, y, b1.y * x)),
              __builtin_fmaf(n.z, z, __builtin_fmaf(b2.z, y, b1.z * x)));
}

// ---- spec §6.3: ray / triangle -------------------------------------------------------------------
__device__ __forceinline__ bool tri_test(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float& t_out) {
    const v3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    if (det == 0.0f) return false;
    const v3 tvec = o - v0;
    const float u = dot(tvec, pvec);
    const v3 qvec = cross(tvec, e1);
    const float v = dot(d, qvec);
    if (det > 0.0f) {
        if (u < 0.0f || v < 0.0f || u + v > det) return false;
    } else {
        if (u > 0.0f || v > 0.0f || u + v < det) return false;
    }
    t_out = dot(e2, qvec) / det;
    return true;
}

// tri_test() for the lanes of a wave that hold a triangle, as straight-line code: the same operations in the same order, the
// per-lane early returns replaced by lane masks combined on the scalar unit and ONE wave-uniform way out before the division.
__device__ __forceinline__ bool tri_test_flat(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float& t_out) {
    const v3 pvec = cross(d, e2);
    const float det = dot(e1, pvec);
    const v3 tvec = o - v0;
    const float u = dot(tvec, pvec);
    const v3 qvec = cross(tvec, e1);
    const float v = dot(d, qvec);
    const float uv = u + v;
    const unsigned long long m_pos = __builtin_amdgcn_ballot_w64(det > 0.0f), m_nz = __builtin_amdgcn_ballot_w64(det != 0.0f);
    const unsigned long long out_pos = __builtin_amdgcn_ballot_w64(u < 0.0f) | __builtin_amdgcn_ballot_w64(v < 0.0f) | __builtin_amdgcn_ballot_w64(uv > det);
    const unsigned long long out_neg = __builtin_amdgcn_ballot_w64(u > 0.0f) | __builtin_amdgcn_ballot_w64(v > 0.0f) | __builtin_amdgcn_ballot_w64(uv < det);
    const unsigned long long inside = m_nz & ((m_pos & ~out_pos) | (~m_pos & ~out_neg));  // (ballots hold the active lanes only)
    if (inside == 0ull) return false;
    t_out = dot(e2, qvec) / det;
    return __builtin_amdgcn_inverse_ballot_w64(inside);
}

__device__ __forceinline__ v3 safe_inv(v3 d) {
    const float x = __builtin_fabsf(d.x) > 1e-20f ? d.x : __builtin_copysignf(1e-20f, d.x);
    const float y = __builtin_fabsf(d.y) > 1e-20f ? d.y : __builtin_copysignf(1e-20f, d.y);
    const float z = __builtin_fabsf(d.z) > 1e-20f ? d.z : __builtin_copysignf(1e-20f, d.z);
    return mk(1.0f / x, 1.0f / y, 1.0f / z);
}

// A ray in traversal form.  The slab test uses t = plane*inv - o*inv (one fma per plane); boxes are
// padded at build time and quantised outward, so this test only has to be conservative, not
// bit-identical to anything (results do not depend on which boxes are visited).
struct TRay {
    v3 o, d, inv, noi;  // noi = -(o * inv)
    float tmax;
    uint32_t oct_inv;   // 7 - octant: slot ^ oct_inv enumerates a node's children front to back
};
__device__ __forceinline__ TRay make_tray(v3 o, v3 d, float tmax) {
    TRay r;
    r.o = o;
    r.d = d;
    r.inv = safe_inv(d);
    r.noi = mk(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
    r.tmax = tmax;
    // by sign BIT, like safe_inv's copysign: a -0.0 component has a negative reciprocal and must take the far plane first
    r.oct_inv = ((__float_as_uint(d.x) >> 31) ? 0u : 4u) | ((__float_as_uint(d.y) >> 31) ? 0u : 2u) | ((__float_as_uint(d.z) >> 31) ? 0u : 1u);
    return r;
}

struct Hit {
    float t;
    int li;       // leaf-order triangle index, -1 = none
    uint32_t id;  // original triangle index (tie-break)
};

struct TravCounters {
    uint32_t nodes, tris, overflow;
    uint32_t flushes = 0;  // TRI_POOL, COUNT: pool_test passes of this wave (wave-uniform)
};

// A traversal work item (Ylitie et al. 2017): either a node group  x = child_base,
// y = hit bits of inner children in 31..24 (bit 24 + (slot ^ oct_inv): front to back) | the parent's imask in 7..0;
// or a triangle group  x = tri_base, y = hit leaf slots in 7..0 | the node's leafmask in 15..8
// (the triangle of leaf slot s is tri_base + popcount(leafmask below s), bvh_build.h).
struct Group {
    uint32_t x, y;
};
__device__ __forceinline__ bool has_nodes(const Group& g) { return g.y > 0x00ffffffu; }
__device__ __forceinline__ bool has_tris(const Group& t) { return (t.y & 0xffu) != 0u; }

typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

// Per-lane traversal stack of 8-byte groups.  The first `lds_cap` entries live in LDS (column of
// this thread, stride 256 entries: conflict-free); the tree pushes at most one pending sibling
// group per level, the builder reports the depth and the host sizes lds_cap + spill_cap to it;
// entries beyond lds_cap spill to a global column (entry-major, coalesced across a wave).
struct TravStack {
    // The LDS part is an address-space-qualified pointer on purpose: with two generic pointers the compiler folds pop()'s two
    // loads into ONE flat_load on a selected address - the flat path, both address computations and a vmcnt(0) + lgkmcnt(0) wait
    // for every pop, even when nothing ever spills.
    lds_u64* lds;
    unsigned long long* spill;
    size_t spill_stride;
    int lds_cap, spill_cap;
    int sp;
    __device__ __forceinline__ void push(Group g, uint32_t& overflow) {
        const unsigned long long v = ((unsigned long long)g.y << 32) | g.x;
        if (sp < lds_cap) lds[sp * 256] = v;
        else if (sp - lds_cap < spill_cap) spill[(size_t)(sp - lds_cap) * spill_stride] = v;
        else {
            overflow = 1;
            return;
        }
        sp++;
    }
    __device__ __forceinline__ Group pop() {  // caller checks sp > 0
        --sp;
        unsigned long long v;
        if (sp < lds_cap) v = lds[sp * 256];
        else v = spill[(size_t)(sp - lds_cap) * spill_stride];
        return Group{(uint32_t)v, (uint32_t)(v >> 32)};
    }
};

__device__ __forceinline__ float ubyte_f32(uint32_t w, int byte) {  // v_cvt_f32_ubyteN
    return (float)((w >> (8 * byte)) & 0xffu);
}

// Visit the nearest pending inner child of node group G: fetch its 80-byte record (five 16-byte
// loads for eight children), slab-test the eight quantised boxes and turn the hits into a new node
// group (inner children, ordered by ray octant) and a triangle group (leaf triangles).
template <bool COUNT, bool UNORDERED = false>
__device__ __forceinline__ void node_step(const float4* __restrict__ nodes, const uint8_t* perm_lut, const TRay& r, Group& G, Group& T, TravStack& stk, TravCounters& tc) {
    const uint32_t hits = G.y;
    const uint32_t bit = 31u - (uint32_t)__builtin_clz(hits);
    G.y &= ~(1u << bit);
    if (has_nodes(G)) stk.push(G, tc.overflow);  // remaining siblings
    const uint32_t slot = UNORDERED ? bit - 24u : (bit - 24u) ^ r.oct_inv;  // UNORDERED (any-hit rays of an all-shadow launch): children in slot order, no re-keying
    const uint32_t rel = (uint32_t)__builtin_popcount(hits & ~(0xffffffffu << slot));  // low byte of hits = imask
    const float4* nd = nodes + (size_t)(G.x + rel) * 5;
    const float4 n0 = nd[0], n1 = nd[1], n2 = nd[2], n3 = nd[3], n4 = nd[4];
    if (COUNT) tc.nodes++;

    const uint32_t w3 = __float_as_uint(n0.w);
    const float sx = __uint_as_float((w3 & 0xffu) << 23), sy = __uint_as_float(((w3 >> 8) & 0xffu) << 23), sz = __uint_as_float(((w3 >> 16) & 0xffu) << 23);
    const uint32_t imask = w3 >> 24;
    // plane t = (p + q*s - o) * inv = q * (s*inv) + (p*inv - o*inv)
    const float ax = sx * r.inv.x, ay = sy * r.inv.y, az = sz * r.inv.z;
    const float bx = __builtin_fmaf(n0.x, r.inv.x, r.noi.x), by = __builtin_fmaf(n0.y, r.inv.y, r.noi.y), bz = __builtin_fmaf(n0.z, r.inv.z, r.noi.z);
    // entry / exit planes per axis are chosen once per node from the ray octant (no per-child min/max)
    const bool px = (r.oct_inv & 4u) != 0u, py = (r.oct_inv & 2u) != 0u, pz = (r.oct_inv & 1u) != 0u;  // direction >= 0
    const uint32_t lx[2] = {__float_as_uint(n2.x), __float_as_uint(n2.y)}, ly[2] = {__float_as_uint(n2.z), __float_as_uint(n2.w)};
    const uint32_t lz[2] = {__float_as_uint(n3.x), __float_as_uint(n3.y)}, hx[2] = {__float_as_uint(n3.z), __float_as_uint(n3.w)};
    const uint32_t hy[2] = {__float_as_uint(n4.x), __float_as_uint(n4.y)}, hz[2] = {__float_as_uint(n4.z), __float_as_uint(n4.w)};
    const uint32_t nx[2] = {px ? lx[0] : hx[0], px ? lx[1] : hx[1]}, fx[2] = {px ? hx[0] : lx[0], px ? hx[1] : lx[1]};
    const uint32_t ny[2] = {py ? ly[0] : hy[0], py ? ly[1] : hy[1]}, fy[2] = {py ? hy[0] : ly[0], py ? hy[1] : ly[1]};
    const uint32_t nz[2] = {pz ? lz[0] : hz[0], pz ? lz[1] : hz[1]}, fz[2] = {pz ? hz[0] : lz[0], pz ? hz[1] : lz[1]};
    // No relative slack on the comparison: the build pads every box by 2e-5 * M (M = largest |coordinate|), at least five
    // times the rounding error of these fmas for ray origins within 32 M (render_pt_common checks the camera), so a box
    // that holds the ray's hit - or a (t, id) tie - always passes tn <= tf and tn <= tmax.
    const float tlim = r.tmax;
    // The eight results are collected as SIGN BITS: miss = (miss << 1) | sign(tf - tn), one v_alignbit_b32 behind one
    // subtraction per child (instead of compare + select + or), children 7 .. 0 so that slot s ends in bit s.  tf - tn < 0 is
    // tn > tf except where tf = -0 meets tn = +0 (a box that ends exactly at the ray's origin and holds no hit with t > 0:
    // missing it is as good as entering it, results do not depend on which boxes are visited); no NaN reaches this point
    // (finite planes, |inv| <= 1e20, tmax = +inf only as the last argument of a minimum).  Empty slots hold inverted boxes.
    uint32_t miss = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const int w = i >> 2, bsel = i & 3;
        const float tnx = __builtin_fmaf(ubyte_f32(nx[w], bsel), ax, bx), tfx = __builtin_fmaf(ubyte_f32(fx[w], bsel), ax, bx);
        const float tny = __builtin_fmaf(ubyte_f32(ny[w], bsel), ay, by), tfy = __builtin_fmaf(ubyte_f32(fy[w], bsel), ay, by);
        const float tnz = __builtin_fmaf(ubyte_f32(nz[w], bsel), az, bz), tfz = __builtin_fmaf(ubyte_f32(fz[w], bsel), az, bz);
        const float tn = fmax_(fmax_(tnx, tny), fmax_(tnz, 0.0f));
        const float tf = fmin_(fmin_(tfx, tfy), fmin_(tfz, tlim));
        miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31u);
    }
    const uint32_t h8 = ~miss & 0xffu;  // bit s: the box in child slot s is hit
    // the hit bits are the work lists: inner children to enter, re-keyed front to back (bit slot -> bit slot ^ oct_inv,
    // one byte from a 2 KiB LDS table), and the leaf slots whose single triangle is to be tested
    const uint32_t leafmask = __float_as_uint(n1.z) & 0xffu;
    const uint32_t keyed = UNORDERED ? (h8 & imask) : perm_lut[r.oct_inv * 256u + (h8 & imask)];
    G.x = __float_as_uint(n1.x);
    G.y = (keyed << 24) | imask;
    T.x = __float_as_uint(n1.y);
    T.y = (h8 & leafmask) | (leafmask << 8);
}

// perm_lut[o * 256 + m] = the byte m with bit s moved to bit s ^ o (8 octants x 256 masks), built once per workgroup
__device__ __forceinline__ void build_perm_lut(uint8_t* lut) {
    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) {
        const uint32_t o = i >> 8, m = i & 0xffu;
        uint32_t out = 0;
#pragma unroll
        for (uint32_t sl = 0; sl < 8; sl++) out |= ((m >> sl) & 1u) << (sl ^ o);
        lut[i] = (uint8_t)out;
    }
    __syncthreads();
}

// Test the next pending triangle of triangle group T.  Returns true when an any-hit ray found an occluder.
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool tri_step(const float4* __restrict__ tris, TRay& r, Hit& best, Group& T, TravCounters& tc) {
    const uint32_t bit = (uint32_t)__builtin_ctz(T.y);  // lowest pending leaf slot (caller checked has_tris)
    T.y &= T.y - 1u;
    const uint32_t li = T.x + (uint32_t)__builtin_popcount((T.y >> 8) & ~(0xffffffffu << bit));  // rank of the slot among the node's leaves
    const float4* tp = tris + (size_t)li * 3;
    const float4 a = tp[0], b = tp[1], c = tp[2];
    if (COUNT) tc.tris++;
    float t;
    if (tri_test_flat(r.o, r.d, mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), t) && t > 0.0f) {
        if (ANY) return t < kShadowTmax;
        const uint32_t id = __float_as_uint(c.y);
        if (t < best.t || (t == best.t && id < best.id)) {
            best.t = t;
            best.li = (int)li;
            best.id = id;
            r.tmax = t;
        }
    }
    return false;
}

// tri_step for a wave whose lanes carry both kinds of ray (trace_queue_mixed): the kind is a per-lane flag.
template <bool COUNT>
__device__ __forceinline__ bool tri_step_mixed(const float4* __restrict__ tris, TRay& r, Hit& best, Group& T, TravCounters& tc, bool is_any) {
    const uint32_t bit = (uint32_t)__builtin_ctz(T.y);
    T.y &= T.y - 1u;
    const uint32_t li = T.x + (uint32_t)__builtin_popcount((T.y >> 8) & ~(0xffffffffu << bit));
    const float4* tp = tris + (size_t)li * 3;
    const float4 a = tp[0], b = tp[1], c = tp[2];
    if (COUNT) tc.tris++;
    float t;
    if (tri_test_flat(r.o, r.d, mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), t) && t > 0.0f) {
        if (is_any) return t < kShadowTmax;
        const uint32_t id = __float_as_uint(c.y);
        if (t < best.t || (t == best.t && id < best.id)) {
            best.t = t;
            best.li = (int)li;
            best.id = id;
            r.tmax = t;
        }
    }
    return false;
}

__device__ __forceinline__ Group root_group() { return Group{0u, 0x80000000u}; }  // "child 0 of nothing" = node 0

// Whole-ray traversal for one lane (used by the rt_trace_rays test hook; the render kernels drive
// the same step functions from a refilling persistent loop).
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse(const float4* __restrict__ nodes, const float4* __restrict__ tris, const uint8_t* perm_lut, v3 o, v3 d, TravStack& stk,
                                         Hit& best, TravCounters& tc) {
    TRay r = make_tray(o, d, ANY ? kShadowTmax : best.t);
    stk.sp = 0;
    Group G = root_group(), T{0u, 0u};
    for (;;) {
        if (!has_nodes(G)) {
            if (stk.sp == 0) return false;
            G = stk.pop();
        }
        node_step<COUNT>(nodes, perm_lut, r, G, T, stk, tc);
        while (has_tris(T))
            if (tri_step<ANY, COUNT>(tris, r, best, T, tc)) return true;
    }
}

// ---- pixel slots ---------------------------------------------------------------------------------
// slot = owned_tile * 4096 + m, m = Morton code of (lx, ly) inside the 64x64 tile: 64 consecutive
// paths cover a compact pixel block, so camera rays of a wave stay coherent.
__device__ __forceinline__ uint32_t compact1by1(uint32_t x) {
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
__device__ __forceinline__ bool slot_pixel(const PtFrame& f, uint32_t slot, uint32_t& px, uint32_t& py, uint32_t& lx, uint32_t& ly, uint32_t& k) {
    k = slot >> 12;
    const uint32_t m = slot & 4095u;
    lx = compact1by1(m);
    ly = compact1by1(m >> 1);
    const uint32_t tile = f.part.rank + k * f.part.n_ranks;
    const uint32_t ty = tile / f.part.tiles_x, tx = tile - ty * f.part.tiles_x;
    px = tx * RT_TILE + lx;
    py = ty * RT_TILE + ly;
    return px < f.width && py < f.height;
}

// Queue append with workgroup-level aggregation: ballot + prefix popcount inside each wave, the wave
// totals meet in LDS, ONE atomic per workgroup reserves the range (a single hot counter serves only
// ~90 atomics/us chip-wide, so one atomic per wave made the compaction kernels atomic-bound).
// Must be called by every thread of the workgroup (two barriers inside).
constexpr uint32_t kAppendThreads = 1024;
__device__ __forceinline__ uint32_t block_append(bool want, uint32_t* counter, uint32_t* lds /* >= 17 words */) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const unsigned long long mask = __ballot(want);
    if (lane == 0) lds[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < n_waves; w++) total += lds[w];
        uint32_t base = total ? atomicAdd(counter, total) : 0u;
        for (uint32_t w = 0; w < n_waves; w++) {
            const uint32_t c = lds[w];
            lds[w] = base;
            base += c;
        }
    }
    __syncthreads();
    const uint32_t idx = lds[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // lds is reused by the next append
    return idx;
}

// Queue append with a sort inside the workgroup ("rays compacted and sorted in LDS"): the rays a workgroup
// emits are reserved as one contiguous range (one atomic, as block_append) and placed inside it in key
// order by an LDS counting sort (histogram with returning LDS atomics -> rank in bin, exclusive scan over
// the bins, position = bin start + rank).  key < kSortBins; the order inside a bin is arrival order (not
// deterministic, and irrelevant: the queue order never changes a result).  Rays that
// sit next to each other in the queue are picked up by the same wave of pt_trace.
// Must be called by every thread of the workgroup.  lds: kSortBins + 40 words.
constexpr uint32_t kSortBins = 512;
template <uint32_t BINS = kSortBins>
__device__ __forceinline__ uint32_t block_append_sorted(bool want, uint32_t key, uint32_t* counter, uint32_t* lds) {
    static_assert(BINS % 64u == 0u && BINS <= kSortBins, "whole waves of bins");
    uint32_t* hist = lds;               // BINS
    uint32_t* wsum = lds + kSortBins;   // 16 wave sums of the scan + 1 base
    for (uint32_t i = threadIdx.x; i < BINS; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    uint32_t rank = 0;
    if (want) rank = atomicAdd(&hist[key], 1u);
    __syncthreads();
    // exclusive scan of the BINS counts by the first BINS threads (wave scan + wave sums)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t v = 0, incl = 0;
    if (threadIdx.x < BINS) {
        v = hist[threadIdx.x];
        incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if ((int)lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < BINS / 64u; w++) {
            const uint32_t c = wsum[w];
            wsum[w] = total;
            total += c;
        }
        wsum[16] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < BINS) hist[threadIdx.x] = wsum[wave] + incl - v;  // bin start inside the workgroup's range
    __syncthreads();
    const uint32_t idx = want ? wsum[16] + hist[key] + rank : 0u;
    __syncthreads();  // lds is reused by the next append
    return idx;
}

// Sort key of a ray: direction octant (3 bits, major) and the cell of its origin in a 4 x 4 x 4 grid over the
// BVH root's quantisation frame (6 bits).
__device__ __forceinline__ uint32_t ray_sort_key(const float4* __restrict__ nodes, v3 o, v3 d) {
    const float4 n0 = nodes[0];  // root: p.xyz, exponent bytes
    const uint32_t w3 = __float_as_uint(n0.w);
    const float ex = __uint_as_float((((w3 & 0xffu) + 6u) & 0xffu) << 23), ey = __uint_as_float(((((w3 >> 8) & 0xffu) + 6u) & 0xffu) << 23),
                ez = __uint_as_float(((((w3 >> 16) & 0xffu) + 6u) & 0xffu) << 23);  // 64 quantisation steps = a quarter of the frame
    const int cx = (int)((o.x - n0.x) / ex), cy = (int)((o.y - n0.y) / ey), cz = (int)((o.z - n0.z) / ez);
    const uint32_t ux = (uint32_t)(cx < 0 ? 0 : cx > 3 ? 3 : cx), uy = (uint32_t)(cy < 0 ? 0 : cy > 3 ? 3 : cy), uz = (uint32_t)(cz < 0 ? 0 : cz > 3 ? 3 : cz);
    const uint32_t oct = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    return (oct << 6) | (uz << 4) | (uy << 2) | ux;
}

// Sort keys that predict WORK rather than locality (tune_sort_rays = 2; 64 bins): rays that sit next to each other in the queue are
// taken by the same wave's refill, and lanes whose rays end at about the same time leave fewer lanes waiting for the next refill.
// Shadow rays: the segment's length in quarter-octaves (a segment is traversed end to end unless it is occluded; its node count
// grows with its length).  Bounce rays: how steeply the direction leaves the scene's long axis (rays along the soup's slab cross
// more of it) - the largest |component| of the direction picks the axis, its magnitude 16 steps.
__device__ __forceinline__ uint32_t shadow_work_key(v3 d) {
    const float l2 = dot(d, d);                                             // squared length: two quarter-octave steps per octave of length
    const int e = (int)((__float_as_uint(l2) >> 21) & 0x3ffu) - (125 << 2);  // exponent and two mantissa bits, offset so that |d| = 0.5 .. 128 maps to 0 .. 63
    return (uint32_t)(e < 0 ? 0 : e > 63 ? 63 : e);
}
__device__ __forceinline__ uint32_t bounce_work_key(v3 d) {
    const float ax = __builtin_fabsf(d.x), ay = __builtin_fabsf(d.y), az = __builtin_fabsf(d.z);
    const uint32_t axis = ax >= ay && ax >= az ? 0u : ay >= az ? 1u : 2u;
    const float m = axis == 0u ? ax : axis == 1u ? ay : az;  // 0.577 .. 1
    const int q = (int)((m - 0.5f) * 32.0f);
    return axis * 16u + (uint32_t)(q < 0 ? 0 : q > 15 ? 15 : q);
}

// camera ray of sample s of pixel (px, py): fragment.glsl:129-133 with the pixel-centre 0.5 replaced by a random offset
__device__ __forceinline__ v3 camera_dir(const PtFrame& f, uint32_t px, uint32_t py, uint32_t s) {
    const uint32_t key = path_key(py * f.width + px, s, f.seed);
    const float nx = ((((float)px + rnd(key, 0, 0)) * 2.0f) / (float)f.width - 1.0f) * f.cam.ratio[0];
    const float ny = ((((float)py + rnd(key, 0, 1)) * 2.0f) / (float)f.height - 1.0f) * f.cam.ratio[1];
    return normalize(rotate_q(f.cam.rot[0], f.cam.rot[1], f.cam.rot[2], f.cam.rot[3], mk(nx, 1.0f, ny)));
}

// ---- generate -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAppendThreads) void pt_generate(const PtFrame f, PtState st, uint32_t* __restrict__ queue, uint32_t* __restrict__ ctr) {
    __shared__ uint32_t lds[32];
    const uint32_t pid = blockIdx.x * kAppendThreads + threadIdx.x;
    bool alive = false;
    if (pid < f.n_paths) {
        const uint32_t slot = pid / f.spp_batch, s = f.sample0 + (pid - slot * f.spp_batch);
        uint32_t px, py, lx, ly, k;
        if (slot_pixel(f, slot, px, py, lx, ly, k)) {
            alive = true;
            const v3 d = camera_dir(f, px, py, s);
            st.ray_o[pid] = make_float4(f.cam.pos[0], f.cam.pos[1], f.cam.pos[2], 0.0f);
            st.ray_d[pid] = make_float4(d.x, d.y, d.z, 0.0f);
            st.thr[pid] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        }
        st.rad[pid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t idx = block_append(alive, &ctr[PT_CTR_COUNT], lds);
    if (alive) queue[idx] = pid;
}

// ---- wave-pooled triangle tests --------------------------------------------------------------------
// In the per-lane loop a triangle phase runs with the 5-7 lanes that happen to hold a leaf hit (0.12 triangles per node
// visit on the 1 M soup), at the price of ~65 vector instructions for the whole wave, every round.  Pooled mode takes
// the phase out of the round: a lane whose node step hit leaf slots appends ONE 8-byte group (tri_base, hit bits,
// leafmask, owner lane) to a per-wave ring in LDS and keeps traversing; when the ring holds enough groups the whole
// wave tests one triangle per lane - ray origin / direction of the owner through ds_bpermute, the result merged into
// the owner's slot with a 64-bit LDS minimum on (t bits, triangle id), which is exactly tri_step's tie-break rule
// (t > 0, so the float's bits order like the float) - and groups with further hit slots go back to the ring.
// The owner learns its new tmax / its occlusion after the flush; until then it may enter nodes a tighter tmax would
// have culled, which never changes a result (boxes are conservative, the hit is a minimum over
// every triangle tested).  The ring is drained before any lane retires, so a ray's result is complete when it is stored.
constexpr uint32_t kPoolRing = 128;  // groups; a flush is due at <= 64, a round adds <= 64
// The pool is read and written by different lanes of ONE wave: DS operations of a wave execute in program order, so no
// barrier instruction is needed; pool_sync() only keeps the COMPILER from moving LDS accesses across the phase boundaries
// (the pointers are not volatile: volatile accesses would stay on generic pointers and become flat_* instructions).
struct TriPool {
    lds_u64* ring;  // kPoolRing groups: x = tri_base, y = hit slots 7..0 | leafmask 15..8 | owner lane 21..16
    lds_u64* best;  // 64 owners: closest = (t bits << 32) | triangle id; any-hit: != 0 = occluded
    lds_u32* li;    // 64 owners: leaf-order index of the best triangle
    uint32_t head, count;  // wave-uniform
};
__device__ __forceinline__ void pool_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
constexpr unsigned long long kPoolNoHit = (0x7f800000ull << 32) | 0xffffffffull;  // (inf, no id)

__device__ __forceinline__ float lane_read(float v, uint32_t src_lane) {  // ds_bpermute_b32: every lane reads lane src_lane's v
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}

// One pass over (at most) the 64 oldest groups of the ring: lane i tests the first pending triangle of group i.
// Must be called by the whole wave in convergent code.
template <bool ANY, bool COUNT>
__device__ __forceinline__ void pool_test(const float4* __restrict__ tris, const TRay& r, TriPool& P, uint32_t lane, TravCounters& tc) {
    const uint32_t n = P.count < 64u ? P.count : 64u;
    const bool has = lane < n;
    if (COUNT) tc.flushes++;
    const unsigned long long e = has ? P.ring[(P.head + lane) & (kPoolRing - 1u)] : 0x100ull << 32;
    const uint32_t ex = (uint32_t)e, ey = (uint32_t)(e >> 32);
    P.head = uniform(P.head + n);
    P.count = uniform(P.count - n);
    const uint32_t bit = (uint32_t)__builtin_ctz(ey);  // lowest pending leaf slot (the idle lanes' dummy has bit 8 set)
    const uint32_t rest = ey & (ey - 1u);
    const uint32_t li = ex + (uint32_t)__builtin_popcount((ey >> 8) & 0xffu & ~(0xffffffffu << bit));
    const uint32_t owner = (ey >> 16) & 63u;
    const v3 o = mk(lane_read(r.o.x, owner), lane_read(r.o.y, owner), lane_read(r.o.z, owner));
    const v3 d = mk(lane_read(r.d.x, owner), lane_read(r.d.y, owner), lane_read(r.d.z, owner));
    bool hit = false;
    float t = 0.0f;
    uint32_t id = 0;
    if (has) {
        const float4* tp = tris + (size_t)li * 3;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        if (COUNT) tc.tris++;
        hit = tri_test(o, d, mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), t) && t > 0.0f;
        id = __float_as_uint(c.y);
    }
    if (ANY) {
        if (hit && t < kShadowTmax) P.best[owner] = 1ull;
    } else {
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | id;
        if (hit) __hip_atomic_fetch_min(&P.best[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        pool_sync();
        if (hit && P.best[owner] == key) P.li[owner] = li;  // ids are unique: at most one lane per owner sees its own key
    }
    // groups with further hit slots go back to the ring
    const bool more = has && (rest & 0xffu) != 0u;
    const unsigned long long mm = __ballot(more);
    if (mm) {
        const uint32_t pos = P.head + P.count + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
        if (more) P.ring[pos & (kPoolRing - 1u)] = ((unsigned long long)rest << 32) | ex;
        P.count = uniform(P.count + (uint32_t)__popcll(mm));
    }
    pool_sync();
}

// ---- trace ----------------------------------------------------------------------------------------
// Persistent waves with per-lane refill.  Traversal lengths are heavy-tailed (a ray may end after 3
// nodes or after 500), so a wave that waits for its slowest ray idles most lanes.  Instead every
// lane carries its own ray: whenever at least `refill_min` lanes have finished, the wave retires
// their results and hands them the next rays of the device-resident queue (one atomic per refill on
// one of PT_HEADS interleaved stream heads, ballot + prefix-popcount to assign entries; a wave whose
// stream runs dry moves on to the next one, so the tail of the queue is shared by all waves).  The wave exits when the queue is drained and all its
// lanes are done, so the grid is sized for the machine, not for the queue length.
// Between two refill checks every lane visits one node and tests up to `kTrisPerRound` triangles.
constexpr int kTrisPerRound = 1;

// How the triangle tests of the per-lane kernels are scheduled (rt_pt_params.tune_tri_mode):
//   TRI_INLINE  every round ends with a triangle phase for the lanes that hold a leaf hit (rounds 1 and 2 of the build)
//   TRI_POOL    wave-pooled tests: leaf hits go to a per-wave LDS ring, the wave tests 64 of them at once (above)
//   TRI_INLINE_PF  the inline phase with the software-pipelined refill (trace_queue_pf below): rays wait in a per-wave LDS ring
//   TRI_DEFER   postponed tests: a lane parks up to two leaf-hit groups in registers and keeps visiting nodes; the triangle
//               phase runs when enough lanes hold a group (or enough of them can do nothing else)
enum { TRI_INLINE = TRI_MODE_INLINE, TRI_POOL = TRI_MODE_POOL, TRI_DEFER = TRI_MODE_DEFER, TRI_INLINE_PF = TRI_MODE_INLINE_PF };

struct PoolMem {  // LDS of one wave's pool (TRI_POOL kernels only)
    lds_u64* ring;
    lds_u64* best;
    lds_u32* li;
};

template <bool ANY, bool COUNT, int MODE>
__device__ __forceinline__ void trace_queue(const PtScene& sc, const PtState& st, const uint32_t* __restrict__ queue,
                                            const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ head,
                                            unsigned long long* __restrict__ stats, TravStack& stk, const uint8_t* perm_lut, uint32_t refill_min,
                                            const PoolMem& pm, uint32_t tri_cfg /* TRI_POOL: byte 0 = groups that trigger a flush, byte 1 = rounds a group may wait */,
                                            uint32_t shadow_stat /* word of the shadow-ray node counter: 4 alone, 11 inside the fused launch */ = 4u) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n = uniform(*count_ptr);
    const int tris_per_round = (int)((refill_min >> 8) & 0xffu) ? (int)((refill_min >> 8) & 0xffu) : kTrisPerRound;
    refill_min &= 0xffu;
    TravCounters tc{0, 0, 0};

    TRay r = make_tray(mk(0.0f, 0.0f, 0.0f), mk(0.0f, 1.0f, 0.0f), 0.0f);
    Hit best{0.0f, -1, 0u};
    Group G{0u, 0u}, T{0u, 0u}, T2{0u, 0u};  // T2: TRI_DEFER's second parking slot
    uint32_t slot = 0;  // closest: path id; any-hit: shadow-queue index
    bool has_ray = false, occluded = false;
    bool exhausted = n == 0;  // wave-uniform: every stream of the queue has been found dry
    // (readfirstlane: threadIdx.x >> 6 is wave-uniform, but only this tells the compiler, and everything the stream index touches -
    // the dry-stream test, `exhausted`, the refill branch - would otherwise live in vector registers under lane masks)
    uint32_t stream = uniform((blockIdx.x * 4u + (threadIdx.x >> 6)) & (PT_HEADS - 1u));  // wave-uniform: the stream this wave pulls from
    uint32_t dry_streams = 0;                                                    // wave-uniform
    uint32_t rounds = 0, alive_rounds = 0;                // COUNT only
    bool alive = false;       // this lane still has traversal work for its ray

    // TRI_POOL state (all wave-uniform)
    TriPool P{pm.ring, pm.best, pm.li, 0u, 0u};
    const lds_u32* best32 = reinterpret_cast<const lds_u32*>(pm.best);
    // TRI_POOL: groups in the ring that trigger a flush / rounds a group may wait;  TRI_DEFER: holding lanes / stuck lanes that trigger the phase
    const uint32_t flush_at = (tri_cfg & 0xffu) ? ((tri_cfg & 0xffu) < 64u ? (tri_cfg & 0xffu) : 64u) : (MODE == TRI_DEFER ? 32u : 40u);
    const uint32_t wait_max = ((tri_cfg >> 8) & 0xffu) ? ((tri_cfg >> 8) & 0xffu) : (MODE == TRI_DEFER ? 8u : 6u);
    uint32_t waited = 0;
    bool flushed = false;

    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle == ~0ull || (!exhausted && (uint32_t)__popcll(idle) >= refill_min)) {
            if (MODE == TRI_POOL) {  // a ray's result is complete only when none of its triangles is pending
                while (P.count) {
                    pool_test<ANY, COUNT>(sc.tris, r, P, lane, tc);
                    flushed = true;
                }
                waited = 0;
            }
            if (!alive && has_ray) {  // retire
                if (MODE == TRI_POOL) {
                    if (ANY) occluded = best32[2u * lane] != 0u;
                    else {
                        best.t = __uint_as_float(best32[2u * lane + 1u]);
                        best.li = (int)pm.li[lane];
                    }
                }
                if (ANY) {
                    if (!occluded) {
                        const uint32_t pid = __float_as_uint(st.sh_o[slot].w);
                        const float4 c = st.sh_c[slot];
                        float4 L = st.rad[pid];
                        L.x += c.x;
                        L.y += c.y;
                        L.z += c.z;
                        st.rad[pid] = L;
                    }
                } else {
                    st.hit[slot] = make_float2(best.t, __int_as_float(best.li));
                }
                has_ray = false;
            }
            if (!exhausted) {
                // One returning atomic on the wave's stream head reserves exactly what the idle lanes
                // need.  Stream-local entry j of stream k is queue entry ((j / 64) * PT_HEADS + k) * 64 + j % 64.
                const uint32_t want = (uint32_t)__popcll(idle);
                const uint32_t my_rank = (uint32_t)__popcll(idle & lt_mask);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head + stream * PT_HEAD_STRIDE, want);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t j = base + my_rank, je = base + want;
                const uint32_t i = !alive ? ((((j >> 6) * PT_HEADS + stream) << 6) | (j & 63u)) : n;  // >= n: nothing for this lane
                if (((((je >> 6) * PT_HEADS + stream) << 6) | (je & 63u)) >= n) {  // this stream is dry (entries grow with j): move on
                    stream = (stream + 1u) & (PT_HEADS - 1u);
                    exhausted = ++dry_streams >= PT_HEADS;
                }
                if (!alive && i < n) {
                    if (ANY) {
                        const float4 so = st.sh_o[i], sd = st.sh_d[i];
                        r = make_tray(mk(so.x, so.y, so.z), mk(sd.x, sd.y, sd.z), kShadowTmax);
                        slot = i;
                        occluded = false;
                        if (MODE == TRI_POOL) pm.best[lane] = 0ull;
                    } else {
                        slot = queue[i];
                        const float4 ro = st.ray_o[slot], rd = st.ray_d[slot];
                        r = make_tray(mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), __builtin_inff());
                        best = Hit{__builtin_inff(), -1, 0xffffffffu};
                        if (MODE == TRI_POOL) {
                            pm.best[lane] = kPoolNoHit;
                            pm.li[lane] = 0xffffffffu;
                        }
                    }
                    has_ray = true;
                    alive = true;
                    G = root_group();
                    T = Group{0u, 0u};
                    T2 = Group{0u, 0u};
                    stk.sp = 0;
                }
            }
            if (__ballot(alive) == 0ull) break;  // queue drained and every lane retired
        }
        if (COUNT) {  // occupancy of the round: wave-rounds and alive lane-rounds
            rounds++;
            alive_rounds += (uint32_t)__popcll(__ballot(alive));
        }
        if (MODE == TRI_POOL) {
            if (flushed) {  // owners pick up what the pool found for them (wave-uniform branch)
                flushed = false;
                if (ANY) {
                    if (alive && best32[2u * lane] != 0u) alive = false;  // occluded: the ray is done
                } else if (alive) {
                    r.tmax = __uint_as_float(best32[2u * lane + 1u]);
                }
            }
            // node phase: every lane with traversal work visits its next node
            T.y = 0u;
            if (alive) {
                if (!has_nodes(G)) {
                    if (stk.sp) G = stk.pop();
                    else alive = false;
                }
                if (alive) node_step<COUNT>(sc.nodes, perm_lut, r, G, T, stk, tc);
            }
            // leaf hits of this round -> the ring (ballot + prefix popcount, no atomic: head / count are wave-uniform)
            const bool add = has_tris(T);
            const unsigned long long am = __ballot(add);
            if (am) {
                const uint32_t pos = P.head + P.count + (uint32_t)__popcll(am & lt_mask);
                if (add) P.ring[pos & (kPoolRing - 1u)] = ((unsigned long long)(T.y | (lane << 16)) << 32) | T.x;
                P.count = uniform(P.count + (uint32_t)__popcll(am));
                pool_sync();
            }
            waited = P.count ? waited + 1u : 0u;
            if (P.count >= flush_at || waited >= wait_max) {
                pool_test<ANY, COUNT>(sc.tris, r, P, lane, tc);
                flushed = true;
                waited = 0;
            }
        } else if (MODE == TRI_DEFER) {
            // node phase: a lane visits its next node as long as it has somewhere to park a leaf-hit group
            bool stuck = false;  // holds a group and cannot visit a node: out of nodes, or both parking slots taken
            if (alive) {
                if (!has_nodes(G) && stk.sp) G = stk.pop();
                if (!has_nodes(G)) {
                    if (has_tris(T)) stuck = true;
                    else alive = false;  // no nodes left, nothing parked: the ray is done
                } else if (has_tris(T2)) {
                    stuck = true;
                } else {
                    Group N{0u, 0u};
                    node_step<COUNT>(sc.nodes, perm_lut, r, G, N, stk, tc);
                    if (has_tris(N)) {
                        if (has_tris(T)) T2 = N;
                        else T = N;
                    }
                }
            }
            // triangle phase: one test per holding lane, when enough lanes hold a group or enough of them are stuck
            const unsigned long long hold = __ballot(alive && has_tris(T));
            const unsigned long long stuck_m = __ballot(stuck);
            const uint32_t n_hold = (uint32_t)__popcll(hold), n_stuck = (uint32_t)__popcll(stuck_m);
            if (n_hold >= flush_at || n_stuck >= wait_max || (n_stuck != 0u && n_stuck == (uint32_t)__popcll(__ballot(alive)))) {
                if (alive && has_tris(T)) {
                    if (tri_step<ANY, COUNT>(sc.tris, r, best, T, tc)) {
                        occluded = true;
                        alive = false;
                    }
                    if (!has_tris(T)) {
                        T = T2;
                        T2 = Group{0u, 0u};
                    }
                }
                if (COUNT) tc.flushes++;
            }
        } else {
            // node phase: lanes without pending triangles visit their next node
            if (alive && !has_tris(T)) {
                if (!has_nodes(G)) {
                    if (stk.sp) G = stk.pop();
                    else alive = false;
                }
                if (alive) node_step<COUNT, ANY>(sc.nodes, perm_lut, r, G, T, stk, tc);
            }
            // triangle phase
#pragma unroll 1
            for (int it = 0; it < tris_per_round; it++) {
                if (alive && has_tris(T)) {
                    if (tri_step<ANY, COUNT>(sc.tris, r, best, T, tc)) {
                        occluded = true;
                        alive = false;
                    }
                }
            }
        }
    }
    if (COUNT) {
        // wave reduction of the traversal counters, one atomic per wave
        unsigned long long a = tc.nodes, b = tc.tris;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            if (!ANY) {
                atomicAdd(&stats[6], (unsigned long long)rounds);
                atomicAdd(&stats[7], (unsigned long long)alive_rounds);
            }
            atomicAdd(&stats[13], (unsigned long long)tc.flushes);
            atomicAdd(&stats[14], (unsigned long long)rounds);
            atomicAdd(&stats[ANY ? shadow_stat : 0u], a);
            atomicAdd(&stats[ANY ? shadow_stat + 1u : 1u], b);
        }
    }
    if (tc.overflow) atomicOr((unsigned int*)&stats[2], 1u);
}

// ---- one loop for both queues of a fused launch ---------------------------------------------------
// trace_queue<closest> followed by trace_queue<any> makes every wave DRAIN between the two: once the closest-hit queue is dry a
// wave gets no refills, its lanes run out one by one (a tenth of its rounds, at nine of 64 lanes alive) and only then does it
// turn to the shadow queue.  The node step is the same for both kinds of ray and the triangle step differs only in what a hit
// means, so here the kind is a per-lane flag: when the closest-hit queue is dry the wave's idle lanes are refilled from the shadow
// queue while its last closest-hit rays are still walking.  One tail per wave and launch instead of two.  Every ray is traced by
// exactly the step functions of the separate loops, so frames and counts are unchanged.
template <bool COUNT>
__device__ __forceinline__ void trace_queue_mixed(const PtScene& sc, const PtState& st, const uint32_t* __restrict__ queue,
                                                  const uint32_t* __restrict__ closest_count, uint32_t* __restrict__ closest_head,
                                                  const uint32_t* __restrict__ shadow_count, uint32_t* __restrict__ shadow_head,
                                                  unsigned long long* __restrict__ stats, TravStack& stk, const uint8_t* perm_lut, uint32_t refill_min) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_closest = uniform(*closest_count), n_shadow = uniform(*shadow_count);
    const int tris_per_round = (int)((refill_min >> 8) & 0xffu) ? (int)((refill_min >> 8) & 0xffu) : kTrisPerRound;
    refill_min &= 0xffu;
    TravCounters tc{0, 0, 0};                          // COUNT: the current ray of this lane
    uint32_t cl_nodes = 0, cl_tris = 0, any_nodes = 0, any_tris = 0;  // COUNT: retired rays of this lane, by kind

    TRay r = make_tray(mk(0.0f, 0.0f, 0.0f), mk(0.0f, 1.0f, 0.0f), 0.0f);
    Hit best{0.0f, -1, 0u};
    Group G{0u, 0u}, T{0u, 0u};
    uint32_t slot = 0;  // closest: path id; any-hit: shadow-queue index
    bool has_ray = false, occluded = false, alive = false;
    bool is_any = false;  // the kind of this lane's ray
    // wave-uniform: which queue the wave refills from, and its state
    uint32_t phase = n_closest == 0u ? 1u : 0u;  // 0 = closest-hit queue, 1 = shadow queue
    uint32_t n = phase == 0u ? n_closest : n_shadow;
    uint32_t* head = phase == 0u ? closest_head : shadow_head;
    bool exhausted = n == 0u;  // both queues dry
    const uint32_t stream0 = uniform((blockIdx.x * 4u + (threadIdx.x >> 6)) & (PT_HEADS - 1u));
    uint32_t stream = stream0, dry_streams = 0;
    uint32_t rounds = 0, alive_rounds = 0;  // COUNT only

    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle == ~0ull || (!exhausted && (uint32_t)__popcll(idle) >= refill_min)) {
            if (!alive && has_ray) {  // retire
                if (is_any) {
                    if (!occluded) {
                        const uint32_t pid = __float_as_uint(st.sh_o[slot].w);
                        const float4 c = st.sh_c[slot];
                        float4 L = st.rad[pid];
                        L.x += c.x;
                        L.y += c.y;
                        L.z += c.z;
                        st.rad[pid] = L;
                    }
                    if (COUNT) {
                        any_nodes += tc.nodes;
                        any_tris += tc.tris;
                    }
                } else {
                    st.hit[slot] = make_float2(best.t, __int_as_float(best.li));
                    if (COUNT) {
                        cl_nodes += tc.nodes;
                        cl_tris += tc.tris;
                    }
                }
                if (COUNT) tc.nodes = tc.tris = 0;
                has_ray = false;
            }
            if (!exhausted) {
                // (as trace_queue: one returning atomic on the wave's stream head reserves what the idle lanes need)
                const uint32_t want = (uint32_t)__popcll(idle);
                const uint32_t my_rank = (uint32_t)__popcll(idle & lt_mask);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(head + stream * PT_HEAD_STRIDE, want);
                base = __builtin_amdgcn_readfirstlane(base);
                const uint32_t j = base + my_rank, je = base + want;
                const uint32_t i = !alive ? ((((j >> 6) * PT_HEADS + stream) << 6) | (j & 63u)) : n;
                const bool take = !alive && i < n;
                if (take) {
                    if (phase != 0u) {
                        const float4 so = st.sh_o[i], sd = st.sh_d[i];
                        r = make_tray(mk(so.x, so.y, so.z), mk(sd.x, sd.y, sd.z), kShadowTmax);
                        slot = i;
                        occluded = false;
                        is_any = true;
                    } else {
                        slot = queue[i];
                        const float4 ro = st.ray_o[slot], rd = st.ray_d[slot];
                        r = make_tray(mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), __builtin_inff());
                        best = Hit{__builtin_inff(), -1, 0xffffffffu};
                        is_any = false;
                    }
                    has_ray = true;
                    alive = true;
                    G = root_group();
                    T = Group{0u, 0u};
                    stk.sp = 0;
                }
                if (((((je >> 6) * PT_HEADS + stream) << 6) | (je & 63u)) >= n) {  // this stream is dry: move on
                    stream = (stream + 1u) & (PT_HEADS - 1u);
                    if (++dry_streams >= PT_HEADS) {  // this queue is dry: on to the shadow queue, or done
                        if (phase == 0u && n_shadow != 0u) {
                            phase = 1u;
                            n = n_shadow;
                            head = shadow_head;
                            stream = stream0;
                            dry_streams = 0u;
                        } else {
                            exhausted = true;
                        }
                    }
                }
            }
            if (__ballot(alive) == 0ull && exhausted) break;  // both queues drained and every lane retired
        }
        if (COUNT) {
            rounds++;
            alive_rounds += (uint32_t)__popcll(__ballot(alive));
        }
        // node phase: lanes without pending triangles visit their next node
        if (alive && !has_tris(T)) {
            if (!has_nodes(G)) {
                if (stk.sp) G = stk.pop();
                else alive = false;
            }
            if (alive) node_step<COUNT>(sc.nodes, perm_lut, r, G, T, stk, tc);
        }
        // triangle phase: one test, then what a hit means to this lane's kind of ray
#pragma unroll 1
        for (int it = 0; it < tris_per_round; it++) {
            if (alive && has_tris(T)) {
                if (tri_step_mixed<COUNT>(sc.tris, r, best, T, tc, is_any)) {
                    occluded = true;
                    alive = false;
                }
            }
        }
    }
    if (COUNT) {
        unsigned long long a = cl_nodes, b = cl_tris, c = any_nodes, d = any_tris;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
            c += __shfl_down(c, off);
            d += __shfl_down(d, off);
        }
        if (lane == 0) {
            atomicAdd(&stats[6], (unsigned long long)rounds);
            atomicAdd(&stats[7], (unsigned long long)alive_rounds);
            atomicAdd(&stats[14], (unsigned long long)rounds);
            atomicAdd(&stats[0], a);
            atomicAdd(&stats[1], b);
            atomicAdd(&stats[11], c);
            atomicAdd(&stats[12], d);
        }
    }
    if (tc.overflow) atomicOr((unsigned int*)&stats[2], 1u);
}

// ---- software-pipelined refill (TRI_INLINE_PF) ------------------------------------------------------
// The blocking refill above is three dependent memory round trips (head atomic -> queue entry -> ray) during which the whole
// wave stands still, plus ~190 vector instructions, and that is why its threshold sits at 24 idle lanes: measured, a refill event
// costs what 2.2 traversal rounds cost, so on average 16 of a wave's 64 lanes wait for the next one (47.8 alive per round).
// Here the fetch is taken out of the lanes' way: the wave keeps a ring of kPfRing ready rays in LDS and a four-stage pipeline
// that advances ONE stage per traversal round - reserve kPfBatch queue entries (returning atomic, result not awaited), read the
// queue entries, read the rays, park them in the ring - so every load was issued a round earlier and its data has arrived behind
// the node fetches in between (vector memory returns in order).  A lane that finishes takes the next ray out of LDS as soon as
// `pop_min` lanes are idle (default 8).  The rays in flight chip-wide grow by at most kPfRing per wave (25 %).
constexpr uint32_t kPfRing = 16, kPfBatch = 8;
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));  // native vector: HIP's float4 class has no address-space-qualified members
typedef __attribute__((address_space(3))) f4v lds_f4;

template <bool ANY, bool COUNT>
__device__ __forceinline__ void trace_queue_pf(const PtScene& sc, const PtState& st, const uint32_t* __restrict__ queue,
                                               const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ head,
                                               unsigned long long* __restrict__ stats, TravStack& stk, const uint8_t* perm_lut, uint32_t refill_min,
                                               lds_f4* ring /* 2 x kPfRing: origin as loaded, (direction, slot bits) */, uint32_t shadow_stat = 4u) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n = uniform(*count_ptr);
    const int tris_per_round = (int)((refill_min >> 8) & 0xffu) ? (int)((refill_min >> 8) & 0xffu) : kTrisPerRound;
    const uint32_t pop_min = (refill_min & 0xffu) ? ((refill_min & 0xffu) < 64u ? (refill_min & 0xffu) : 64u) : 8u;
    TravCounters tc{0, 0, 0};

    TRay r = make_tray(mk(0.0f, 0.0f, 0.0f), mk(0.0f, 1.0f, 0.0f), 0.0f);
    Hit best{0.0f, -1, 0u};
    Group G{0u, 0u}, T{0u, 0u};
    uint32_t slot = 0;  // closest: path id; any-hit: shadow-queue index
    bool has_ray = false, occluded = false, alive = false;
    uint32_t rounds = 0, alive_rounds = 0;  // COUNT only

    // fetch pipeline (wave-uniform unless noted)
    uint32_t stream = uniform((blockIdx.x * 4u + (threadIdx.x >> 6)) & (PT_HEADS - 1u));
    uint32_t dry_streams = 0;
    uint32_t fetch_done = n == 0u ? 1u : 0u;  // every stream of the queue has been found dry
    uint32_t pf_stage = 0;                // 0 idle, 1 entries reserved, 2 queue entries read (closest-hit only), 3 rays read
    uint32_t pf_base = 0;                 // lane 0: what the head atomic returned
    uint32_t pf_idx = n, pf_slot = 0;     // per lane < kPfBatch: queue index (>= n: none), path id / shadow index
    // per lane: the ray, kept as the two 16-byte tuples the loads deliver and the LDS stores take (scalars would be copied out of the
    // load's registers as soon as it is issued, and the copies wait for the data)
    f4v pf_o = {0.0f, 0.0f, 0.0f, 0.0f}, pf_d = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t ring_head = 0, ring_count = 0;

    for (;;) {
        // (the wave-uniform state, pinned to scalar registers: without this the divergence analysis taints it through the lane-
        // dependent code around it and every branch below becomes a lane-masked region whose merges wait for the loads just issued)
        pf_stage = uniform(pf_stage);
        ring_head = uniform(ring_head);
        ring_count = uniform(ring_count);
        stream = uniform(stream);
        dry_streams = uniform(dry_streams);
        fetch_done = uniform(fetch_done);
        // ---- one pipeline stage per round; every value used here was requested a round ago ----
        if (pf_stage == 3u) {
            const bool valid = pf_idx < n;
            const unsigned long long vm = __ballot(valid);
            if (valid) {
                const uint32_t pos = (ring_head + ring_count + (uint32_t)__popcll(vm & lt_mask)) & (kPfRing - 1u);
                f4v d4 = pf_d;
                d4.w = __uint_as_float(pf_slot);
                ring[2u * pos] = pf_o;
                ring[2u * pos + 1u] = d4;
            }
            ring_count = uniform(ring_count + (uint32_t)__popcll(vm));
            pf_stage = 0u;
            pool_sync();
        } else if (pf_stage == 2u) {
            // (every lane loads, lanes without an entry a clamped address: a load under a lane predicate is merged into the live
            // registers with copies, and the copies would wait for the data right here)
            pf_o = *reinterpret_cast<const f4v*>(&st.ray_o[pf_slot]);
            pf_d = *reinterpret_cast<const f4v*>(&st.ray_d[pf_slot]);
            pf_stage = 3u;
        } else if (pf_stage == 1u) {
            // Stream-local entry j of stream k is queue entry ((j / 64) * PT_HEADS + k) * 64 + j % 64.
            const uint32_t base = uniform(pf_base);
            const uint32_t j = base + lane, je = base + kPfBatch;
            pf_idx = lane < kPfBatch ? ((((j >> 6) * PT_HEADS + stream) << 6) | (j & 63u)) : n;
            if (((((je >> 6) * PT_HEADS + stream) << 6) | (je & 63u)) >= n) {  // this stream is dry (entries grow with j): move on
                stream = (stream + 1u) & (PT_HEADS - 1u);
                fetch_done = ++dry_streams >= PT_HEADS ? 1u : 0u;
            }
            const uint32_t safe = pf_idx < n ? pf_idx : n - 1u;  // n > 0 here
            if (ANY) {
                pf_o = *reinterpret_cast<const f4v*>(&st.sh_o[safe]);
                pf_d = *reinterpret_cast<const f4v*>(&st.sh_d[safe]);
                pf_slot = safe;
                pf_stage = 3u;
            } else {
                pf_slot = queue[safe];
                pf_stage = 2u;
            }
        }
        if (pf_stage == 0u && !fetch_done && ring_count + kPfBatch <= kPfRing) {
            if (lane == 0) pf_base = atomicAdd(head + stream * PT_HEAD_STRIDE, kPfBatch);
            pf_stage = 1u;
        }

        // ---- retire finished lanes and hand them rays from the ring ----
        const unsigned long long idle = __ballot(!alive);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        const bool drained = fetch_done && pf_stage == 0u && ring_count == 0u;
        if ((ring_count != 0u && n_idle >= pop_min) || (n_idle == 64u && (ring_count != 0u || drained))) {
            if (!alive && has_ray) {  // retire
                if (ANY) {
                    if (!occluded) {
                        const uint32_t pid = __float_as_uint(st.sh_o[slot].w);
                        const float4 c = st.sh_c[slot];
                        float4 L = st.rad[pid];
                        L.x += c.x;
                        L.y += c.y;
                        L.z += c.z;
                        st.rad[pid] = L;
                    }
                } else {
                    st.hit[slot] = make_float2(best.t, __int_as_float(best.li));
                }
                has_ray = false;
            }
            const uint32_t m = n_idle < ring_count ? n_idle : ring_count;
            const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
            if (!alive && rank < m) {
                const uint32_t pos = (ring_head + rank) & (kPfRing - 1u);
                const f4v ro = ring[2u * pos], rd = ring[2u * pos + 1u];
                slot = __float_as_uint(rd.w);
                if (ANY) {
                    r = make_tray(mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), kShadowTmax);
                    occluded = false;
                } else {
                    r = make_tray(mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), __builtin_inff());
                    best = Hit{__builtin_inff(), -1, 0xffffffffu};
                }
                has_ray = true;
                alive = true;
                G = root_group();
                T = Group{0u, 0u};
                stk.sp = 0;
            }
            ring_head = uniform(ring_head + m);
            ring_count = uniform(ring_count - m);
            pool_sync();
            if (m == 0u && drained) break;  // queue and ring empty, every lane retired
        }
        if (COUNT) {
            rounds++;
            alive_rounds += (uint32_t)__popcll(__ballot(alive));
        }
        // node phase: lanes without pending triangles visit their next node
        if (alive && !has_tris(T)) {
            if (!has_nodes(G)) {
                if (stk.sp) G = stk.pop();
                else alive = false;
            }
            if (alive) node_step<COUNT>(sc.nodes, perm_lut, r, G, T, stk, tc);
        }
        // triangle phase
#pragma unroll 1
        for (int it = 0; it < tris_per_round; it++) {
            if (alive && has_tris(T)) {
                if (tri_step<ANY, COUNT>(sc.tris, r, best, T, tc)) {
                    occluded = true;
                    alive = false;
                }
            }
        }
    }
    if (COUNT) {
        unsigned long long a = tc.nodes, b = tc.tris;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            if (!ANY) {
                atomicAdd(&stats[6], (unsigned long long)rounds);
                atomicAdd(&stats[7], (unsigned long long)alive_rounds);
            }
            atomicAdd(&stats[14], (unsigned long long)rounds);
            atomicAdd(&stats[ANY ? shadow_stat : 0u], a);
            atomicAdd(&stats[ANY ? shadow_stat + 1u : 1u], b);
        }
    }
    if (tc.overflow) atomicOr((unsigned int*)&stats[2], 1u);
}

// LDS of a 256-thread workgroup of the per-lane kernels: the traversal stacks (dynamic), the octant table and, for TRI_POOL
// kernels, four pools of 1.75 KiB.
template <int MODE>
struct PoolLds {
    __device__ __forceinline__ PoolMem get(uint32_t) { return PoolMem{nullptr, nullptr, nullptr}; }
};
template <>
struct PoolLds<TRI_POOL> {
    unsigned long long ring[4][kPoolRing];
    unsigned long long best[4][64];
    uint32_t li[4][64];
    __device__ __forceinline__ PoolMem get(uint32_t wave) {
        return PoolMem{(lds_u64*)ring[wave], (lds_u64*)best[wave], (lds_u32*)li[wave]};
    }
};
template <>
struct PoolLds<TRI_INLINE_PF> {
    f4v rays[4][2 * kPfRing];
    __device__ __forceinline__ PoolMem get(uint32_t) { return PoolMem{nullptr, nullptr, nullptr}; }
    __device__ __forceinline__ lds_f4* ring(uint32_t wave) { return (lds_f4*)rays[wave]; }
};
constexpr uint32_t kPoolLdsBytes = 4u * (kPoolRing * 8u + 64u * 8u + 64u * 4u);
constexpr uint32_t kPfLdsBytes = 4u * 2u * kPfRing * 16u;
constexpr int kInlineWaves = 8;  // (7 = 72 VGPRs compiles to the same instruction count)
constexpr int kPfWaves = 6;    // TRI_INLINE_PF: the prefetch registers (two 16-byte tuples, index, slot) do not fit 72 VGPRs without spills in the loop
constexpr int kPoolWaves = 7;  // waves per SIMD the TRI_POOL / TRI_DEFER kernels are compiled for (72 VGPRs; the default stack split leaves room for seven workgroups per CU anyway)

template <bool ANY, bool COUNT, int MODE>
__global__ __launch_bounds__(256, MODE == TRI_INLINE ? kInlineWaves : MODE == TRI_INLINE_PF ? kPfWaves : kPoolWaves) void pt_trace(const PtScene sc, PtState st, const uint32_t* __restrict__ queue,
                                                const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ head,
                                                unsigned long long* __restrict__ stats, const StackCfg sk, uint32_t refill_min, uint32_t tri_cfg) {
    extern __shared__ unsigned long long lds_stack[];  // sk.lds_cap x 256 entries
    __shared__ uint8_t perm_lut[2048];
    __shared__ PoolLds<MODE> pool;
    build_perm_lut(perm_lut);
    const size_t gtid = (size_t)blockIdx.x * 256u + threadIdx.x;
    TravStack stk{(lds_u64*)&lds_stack[threadIdx.x], sk.spill + gtid, sk.spill_stride, sk.lds_cap, sk.spill_cap, 0};
    if constexpr (MODE == TRI_INLINE_PF) trace_queue_pf<ANY, COUNT>(sc, st, queue, count_ptr, head, stats, stk, perm_lut, refill_min, pool.ring(threadIdx.x >> 6));
    else trace_queue<ANY, COUNT, MODE>(sc, st, queue, count_ptr, head, stats, stk, perm_lut, refill_min, pool.get(threadIdx.x >> 6), tri_cfg);
}

// closest-hit rays of depth d + 1 and the shadow rays of depth d in ONE persistent launch: the two are independent (the
// shadow rays only add to the paths' radiance, the closest-hit rays only read rays), so every wave first pulls from the
// closest-hit queue - the frame's critical path: shade(d + 1) waits for it - and moves on to the shadow queue when that one
// is dry, instead of leaving the machine to the few long rays of a launch's tail.  One tail per bounce instead of two.
template <bool COUNT, int MODE>
__global__ __launch_bounds__(256, MODE == TRI_INLINE ? kInlineWaves : MODE == TRI_INLINE_PF ? kPfWaves : kPoolWaves) void pt_trace_fused(const PtScene sc, PtState st, const uint32_t* __restrict__ queue,
                                                      const uint32_t* __restrict__ closest_count, uint32_t* __restrict__ closest_head,
                                                      const uint32_t* __restrict__ shadow_count, uint32_t* __restrict__ shadow_head,
                                                      unsigned long long* __restrict__ stats, const StackCfg sk, uint32_t refill_min, uint32_t tri_cfg) {
    extern __shared__ unsigned long long lds_stack[];
    __shared__ uint8_t perm_lut[2048];
    __shared__ PoolLds<MODE> pool;
    build_perm_lut(perm_lut);
    const size_t gtid = (size_t)blockIdx.x * 256u + threadIdx.x;
    TravStack stk{(lds_u64*)&lds_stack[threadIdx.x], sk.spill + gtid, sk.spill_stride, sk.lds_cap, sk.spill_cap, 0};
    if constexpr (MODE == TRI_INLINE_PF) {
        trace_queue_pf<false, COUNT>(sc, st, queue, closest_count, closest_head, stats, stk, perm_lut, refill_min, pool.ring(threadIdx.x >> 6));
        trace_queue_pf<true, COUNT>(sc, st, nullptr, shadow_count, shadow_head, stats, stk, perm_lut, refill_min, pool.ring(threadIdx.x >> 6), 11u);
    } else if constexpr (MODE == TRI_INLINE) {
        if (tri_cfg & 1u) {  // tuning: the two loops one after the other (every wave drains between the queues)
            const PoolMem pm = pool.get(threadIdx.x >> 6);
            trace_queue<false, COUNT, MODE>(sc, st, queue, closest_count, closest_head, stats, stk, perm_lut, refill_min, pm, tri_cfg);
            trace_queue<true, COUNT, MODE>(sc, st, nullptr, shadow_count, shadow_head, stats, stk, perm_lut, refill_min, pm, tri_cfg, 11u);
        } else {
            trace_queue_mixed<COUNT>(sc, st, queue, closest_count, closest_head, shadow_count, shadow_head, stats, stk, perm_lut, refill_min);
        }
    } else {
        const PoolMem pm = pool.get(threadIdx.x >> 6);
        trace_queue<false, COUNT, MODE>(sc, st, queue, closest_count, closest_head, stats, stk, perm_lut, refill_min, pm, tri_cfg);
        trace_queue<true, COUNT, MODE>(sc, st, nullptr, shadow_count, shadow_head, stats, stk, perm_lut, refill_min, pm, tri_cfg, 11u);
    }
}

// ---- packet trace (camera rays) ---------------------------------------------------------------------
// Camera rays share their origin and the 64 paths of a wave cover a 4x4-pixel block (Morton slots), so the
// whole wave walks the tree TOGETHER: one wave-uniform traversal (stack of node groups in LDS, node header and
// triangle records through scalar loads), the 48 quantised planes of a node decoded ONCE per wave - lane k
// converts plane k and parks it in LDS, ordered near / far for the packet's direction octant - and every
// lane then only runs the six slab fmas per child against planes broadcast from LDS.  A child is entered
// when ANY lane's ray hits its box, a leaf's triangles are tested by all lanes (testing more boxes or
// triangles than a ray needs never changes its (t, id)-minimal hit).  Per node step
// this costs about 100 vector instructions and one 48-byte vector load for 64 rays, against about 300
// instructions and 64 x 5 sixteen-byte gathers in pt_trace.  Lanes whose octant differs from the packet
// leader's (blocks that straddle a sign change of the direction) are walked in a further pass.
constexpr int kPkStack = (int)kPacketStackEntries;  // one pending sibling group per tree level; render_pt_common sends trees whose stack_need exceeds it
                                                   // (single-level: depth <= kBvhMaxDepth / 3 + 2; a flattened two-level tree adds its top level) to the per-lane kernel

template <bool COUNT>
__global__ __launch_bounds__(256) void pt_trace_packet(const PtScene sc, const PtFrame f, PtState st, unsigned long long* __restrict__ stats) {
    // (PtState is passed by value: its pointers are written through)
    __shared__ float s_planes[4][64];
    __shared__ unsigned long long s_stack[4][kPkStack];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    float* planes = s_planes[wv];
    unsigned long long* stack = s_stack[wv];
    const uint32_t pid = (blockIdx.x * 4u + wv) * 64u + lane;

    bool alive = false;
    v3 d = mk(0.0f, 1.0f, 0.0f);
    if (pid < f.n_paths) {  // the generate stage, fused: this kernel makes the camera rays it traces (pt_shade(0) needs only the direction)
        const uint32_t slot = pid / f.spp_batch;
        uint32_t px, py, lx, ly, k;
        alive = slot_pixel(f, slot, px, py, lx, ly, k);
        if (alive) {
            d = camera_dir(f, px, py, f.sample0 + (pid - slot * f.spp_batch));
            st.ray_d[pid] = make_float4(d.x, d.y, d.z, 0.0f);
        }
    }
    const v3 o = mk(f.cam.pos[0], f.cam.pos[1], f.cam.pos[2]);  // wave-uniform
    const v3 inv = safe_inv(d);
    const v3 noi = mk(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    const uint32_t oct_inv = ((__float_as_uint(d.x) >> 31) ? 0u : 4u) | ((__float_as_uint(d.y) >> 31) ? 0u : 2u) | ((__float_as_uint(d.z) >> 31) ? 0u : 1u);
    Hit best{__builtin_inff(), -1, 0xffffffffu};

    // lane k < 48 decodes plane k of a node: byte 32 + k = qlo.x[8] qlo.y[8] qlo.z[8] qhi.x[8] qhi.y[8] qhi.z[8]
    const uint32_t pl = lane < 48u ? lane : 47u;
    const uint32_t p_axis = (pl >> 3) % 3u, p_child = pl & 7u;
    const bool p_hi = pl >= 24u;
    uint32_t n_nodes = 0, n_tris = 0, overflow = 0;  // wave-uniform

    unsigned long long remaining = __ballot(alive);
    while (remaining) {
        const uint32_t oct = (uint32_t)__builtin_amdgcn_readlane((int)oct_inv, (int)__builtin_ctzll(remaining));
        const bool act = alive && oct_inv == oct;
        const unsigned long long act_mask = __ballot(act);  // wave-uniform
        remaining &= ~act_mask;
        // LDS slot of this lane's plane: child * 8 + {near x, near y, near z, far x, far y, far z}
        const bool dir_pos = ((oct >> (2u - p_axis)) & 1u) != 0u;  // oct bit 4 = x, 2 = y, 1 = z: direction >= 0
        const uint32_t lds_idx = p_child * 8u + (p_hi == dir_pos ? 3u : 0u) + p_axis;

        int sp = 0;
        uint32_t gx = 0u, gy = 0x80000000u;  // the root group
        for (;;) {
            if (gy <= 0x00ffffffu) {
                if (sp == 0) break;
                const unsigned long long e = stack[--sp];
                gx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)e);
                gy = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(e >> 32));
            }
            const uint32_t bit = 31u - (uint32_t)__builtin_clz(gy);
            const uint32_t hits = gy;
            gy &= ~(1u << bit);
            if (gy > 0x00ffffffu) {  // remaining siblings
                if (sp < kPkStack) {
                    if (lane == 0) stack[sp] = ((unsigned long long)gy << 32) | gx;
                    sp++;
                } else {
                    overflow = 1;
                }
            }
            const uint32_t slot = (bit - 24u) ^ oct;
            const uint32_t node = gx + (uint32_t)__builtin_popcount(hits & ~(0xffffffffu << slot));
            const uint32_t* __restrict__ nd = reinterpret_cast<const uint32_t*>(sc.nodes) + (size_t)uniform(node) * 20u;
            if (COUNT) n_nodes++;
            // header: the address is wave-uniform, so the 32 bytes come through the SCALAR cache into scalar registers.  Written as an
            // s_load: left to itself the compiler issues vector loads of the one address (it cannot rule out that the kernel's own
            // stores alias the node array) and moves the seven words to scalar registers with seven v_readfirstlane
            u32x8 hdr;
            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hdr) : "s"(nd) : "memory");
            const float px_ = __uint_as_float(hdr[0]), py_ = __uint_as_float(hdr[1]), pz_ = __uint_as_float(hdr[2]);
            const uint32_t w3 = hdr[3], child_base = hdr[4], tri_base = hdr[5], leafmask = hdr[6] & 0xffu;
            const float sx = __uint_as_float((w3 & 0xffu) << 23), sy = __uint_as_float(((w3 >> 8) & 0xffu) << 23), sz = __uint_as_float(((w3 >> 16) & 0xffu) << 23);
            const uint32_t imask = w3 >> 24;
            // cooperative decode: one 48-byte vector load for the wave, world-space plane = p + q * scale
            const uint32_t q = reinterpret_cast<const uint8_t*>(nd)[32u + pl];
            const float ps = p_axis == 0u ? sx : p_axis == 1u ? sy : sz, pp = p_axis == 0u ? px_ : p_axis == 1u ? py_ : pz_;
            const float plane = __builtin_fmaf((float)q, ps, pp);
            __builtin_amdgcn_wave_barrier();  // the previous node's plane reads are done (one wave: DS ops run in order)
            if (lane < 48u) planes[lds_idx] = plane;
            __builtin_amdgcn_wave_barrier();
            uint32_t any = 0;  // bit c: some ray of the pass hits child slot c (empty slots hold inverted boxes and never hit)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float4 a = *reinterpret_cast<const float4*>(&planes[c * 8]);      // near x, y, z, far x
                const float2 b = *reinterpret_cast<const float2*>(&planes[c * 8 + 4]);  // far y, z
                const float tn = fmax_(fmax_(__builtin_fmaf(a.x, inv.x, noi.x), __builtin_fmaf(a.y, inv.y, noi.y)), fmax_(__builtin_fmaf(a.z, inv.z, noi.z), 0.0f));
                const float tf = fmin_(fmin_(__builtin_fmaf(a.w, inv.x, noi.x), __builtin_fmaf(b.x, inv.y, noi.y)), fmin_(__builtin_fmaf(b.y, inv.z, noi.z), best.t));
                // (scalar arithmetic on the ballot, no bool: a uniform i1 is kept as a lane mask and the select comes back through
                // v_cndmask + v_readfirstlane, two vector instructions per child)
                const uint32_t hit_lanes = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(tn <= tf) & act_mask);
                any |= (hit_lanes < 1u ? hit_lanes : 1u) << c;
            }
            // wave-uniform bookkeeping, branch-free on the scalar unit: inner children to enter, keyed by
            // slot ^ octant (front to back), and the triangles of the leaf children that were hit
            uint32_t ih = any & imask;
            ih = (oct & 1u) ? (((ih & 0x55u) << 1) | ((ih >> 1) & 0x55u)) : ih;
            ih = (oct & 2u) ? (((ih & 0x33u) << 2) | ((ih >> 2) & 0x33u)) : ih;
            ih = (oct & 4u) ? (((ih & 0x0fu) << 4) | ((ih >> 4) & 0x0fu)) : ih;
            const uint32_t inner_hits = ih << 24;
            for (uint32_t lh = any & leafmask; lh; lh &= lh - 1u) {  // the single triangles of the leaf slots some ray hit
                const uint32_t c = (uint32_t)__builtin_ctz(lh);
                const uint32_t li = tri_base + (uint32_t)__builtin_popcount(leafmask & ~(0xffffffffu << c));
                const float* __restrict__ tp = reinterpret_cast<const float*>(sc.tris) + (size_t)li * 12u;  // wave-uniform: scalar loads
                if (COUNT) n_tris++;
                float t;
                if (act && tri_test(o, d, mk(tp[0], tp[1], tp[2]), mk(tp[3], tp[4], tp[5]), mk(tp[6], tp[7], tp[8]), t) && t > 0.0f) {
                    const uint32_t id = __float_as_uint(tp[9]);
                    if (t < best.t || (t == best.t && id < best.id)) {
                        best.t = t;
                        best.li = (int)li;
                        best.id = id;
                    }
                }
            }
            gx = child_base;
            gy = inner_hits | imask;
        }
    }
    if (alive) st.hit[pid] = make_float2(best.t, __int_as_float(best.li));
    if (COUNT && lane == 0) {  // records fetched once per wave
        atomicAdd(&stats[9], (unsigned long long)n_nodes);
        atomicAdd(&stats[10], (unsigned long long)n_tris);
        atomicAdd(&stats[8], 1ull);
    }
    if (overflow && lane == 0) atomicOr((unsigned int*)&stats[2], 1u);
}

// ---- packet trace, interval form ---------------------------------------------------------------------
// The same walk with the node test in two steps.  (1) ONE interval test per child for the whole pass, on 8 lanes per child:
// the rays of a pass share their origin and the signs of their direction, so with [imin, imax] the range of 1 / d over the
// pass's lanes (per axis) every lane's slab distance (plane - o) * inv lies between the products with the two ends; lane
// 8c + k holds plane k of child c (k = 0..2 near x y z, 4..6 far x y z, 3 / 7 the constants 0 and -(largest best.t)),
// turns it into a LOWER bound of t_near resp. of -t_far (widened by the rounding of the lanes' own fma form), and two
// quad-wide DPP maxima + one half-row mirror add give  max(lower bounds of t_near, 0) - min(upper bounds of t_far, best)  in
// lane 8c; one ds_bpermute hands the eight verdicts to lanes 0..15 in the two orders the bookkeeping wants (front to back for
// the inner children, slot order for the leaves), so ONE ballot is the next node group and the leaf list: about ten vector
// instructions for all eight children.  (2) Only children that pass - 1.2 of 8 on the metric's scene are hit by any ray -
// get the per-ray slab test of the kernel above (planes parked in LDS as there); PURE skips (2) and enters every child that
// passes (1).  Both walk a superset of the boxes each ray would visit alone and test every triangle with the ray's own
// arithmetic, so frames are unchanged.  The traversal stack lives in two VGPRs (entry i in lane i).
typedef __attribute__((address_space(3))) float lds_f32;
// v = max(v, v of the lane the DPP control names); written out because the builtin form (v_mov_dpp, then fmaxf) pays a
// canonicalising v_max per operand.  s_nop 1: a DPP read of a VGPR needs two wait states behind the VALU write, which the
// compiler does not insert for text it does not parse.
#define RT_DPP_MAX(v, ctrl) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
// largest of a wave-uniform set of NON-NEGATIVE floats, one per lane (integer order = float order): rows by DPP, then the scalar unit
__device__ __forceinline__ float wave_max_nonneg(float v) {
    RT_DPP_MAX(v, "quad_perm:[1,0,3,2]");
    RT_DPP_MAX(v, "quad_perm:[2,3,0,1]");
    RT_DPP_MAX(v, "row_half_mirror");
    RT_DPP_MAX(v, "row_mirror");
    const uint32_t b = __float_as_uint(v);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)b, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)b, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)b, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)b, 48);
    const uint32_t m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
    return __uint_as_float(m01 > m23 ? m01 : m23);
}
#define RT_DPP_MIN(v, ctrl) asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float wave_min_nonneg(float v) {
    RT_DPP_MIN(v, "quad_perm:[1,0,3,2]");
    RT_DPP_MIN(v, "quad_perm:[2,3,0,1]");
    RT_DPP_MIN(v, "row_half_mirror");
    RT_DPP_MIN(v, "row_mirror");
    const uint32_t b = __float_as_uint(v);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)b, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)b, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)b, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)b, 48);
    const uint32_t m01 = r0 < r1 ? r0 : r1, m23 = r2 < r3 ? r2 : r3;
    return __uint_as_float(m01 < m23 ? m01 : m23);
}
template <bool COUNT, bool PURE, bool FARCAP>
__global__ __launch_bounds__(256) void pt_trace_packet_ia(const PtScene sc, const PtFrame f, PtState st, unsigned long long* __restrict__ stats) {
    __shared__ f4v s_planes[4][16];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    lds_f32* planes = (lds_f32*)s_planes[wv];
    lds_f4* planes4 = (lds_f4*)s_planes[wv];
    const uint32_t pid = (blockIdx.x * 4u + wv) * 64u + lane;

    bool alive = false;
    v3 d = mk(0.0f, 1.0f, 0.0f);
    if (pid < f.n_paths) {
        const uint32_t slot = pid / f.spp_batch;
        uint32_t px, py, lx, ly, k;
        alive = slot_pixel(f, slot, px, py, lx, ly, k);
        if (alive) {
            d = camera_dir(f, px, py, f.sample0 + (pid - slot * f.spp_batch));
            st.ray_d[pid] = make_float4(d.x, d.y, d.z, 0.0f);
        }
    }
    const v3 o = mk(f.cam.pos[0], f.cam.pos[1], f.cam.pos[2]);  // wave-uniform
    const v3 inv = safe_inv(d);
    const v3 noi = mk(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
    const uint32_t oct_inv = ((__float_as_uint(d.x) >> 31) ? 0u : 4u) | ((__float_as_uint(d.y) >> 31) ? 0u : 2u) | ((__float_as_uint(d.z) >> 31) ? 0u : 1u);
    Hit best{__builtin_inff(), -1, 0xffffffffu};

    // lane 8c + k: plane k of child c
    const uint32_t p_child = lane >> 3, pk = lane & 7u, p_axis = pk & 3u;
    const bool p_plane = p_axis != 3u, p_far = pk >= 4u;
    const float o_ax = p_axis == 0u ? o.x : p_axis == 1u ? o.y : o.z;
    uint32_t n_nodes = 0, n_tris = 0, overflow = 0;  // wave-uniform

    unsigned long long remaining = __ballot(alive);
    while (remaining) {
        const uint32_t oct = (uint32_t)__builtin_amdgcn_readlane((int)oct_inv, (int)__builtin_ctzll(remaining));
        const bool act = alive && oct_inv == oct;
        const unsigned long long act_mask = __ballot(act);  // wave-uniform
        remaining &= ~act_mask;
        // the pass's range of 1 / d per axis (one sign per axis: the octant is shared), and this lane's share of it
        const float inf = __builtin_inff();
        // (by magnitude - rows by DPP, the four rows on the scalar unit in integer order - and the pass's sign put back; which end is
        // "min" does not matter: the bounds below take the smaller of the two products)
        const float ax = __builtin_fabsf(inv.x), ay = __builtin_fabsf(inv.y), az = __builtin_fabsf(inv.z);
        const float ix0 = wave_min_nonneg(act ? ax : inf), ix1 = wave_max_nonneg(act ? ax : 0.0f);
        const float iy0 = wave_min_nonneg(act ? ay : inf), iy1 = wave_max_nonneg(act ? ay : 0.0f);
        const float iz0 = wave_min_nonneg(act ? az : inf), iz1 = wave_max_nonneg(act ? az : 0.0f);
        const float i_sign = ((oct >> (2u - (p_axis < 2u ? p_axis : 2u))) & 1u) != 0u ? 1.0f : -1.0f;  // oct bit set: direction >= 0
        const float imin = i_sign * (p_axis == 0u ? ix0 : p_axis == 1u ? iy0 : iz0), imax = i_sign * (p_axis == 0u ? ix1 : p_axis == 1u ? iy1 : iz1);
        // near lanes bound t from below: min(u * imin, u * imax); far lanes bound -t from below: min(u * -imin, u * -imax)
        const float ia_a = p_plane ? (p_far ? -imin : imin) : 0.0f, ia_b = p_plane ? (p_far ? -imax : imax) : 0.0f;
        // widening: a lane computes fma(plane, inv, -(o * inv)), off the exact (plane - o) * inv by at most 2^-24 (|o * inv| + |t|)
        float ia_m = p_plane ? (__builtin_fabsf(o_ax) * fmax_(__builtin_fabsf(imin), __builtin_fabsf(imax))) * 0x1p-22f : (pk == 3u ? 0.0f : inf);
        const bool dir_pos = p_plane && ((oct >> (2u - p_axis)) & 1u) != 0u;  // oct bit 4 = x, 2 = y, 1 = z: direction >= 0
        const uint32_t q_off = p_plane ? 32u + (p_far == dir_pos ? 24u : 0u) + p_axis * 8u + p_child : 32u + p_child;
        // Lanes 0..15 collect the children's verdicts (one ds_bpermute of lane 8c's value): lane p < 8 reads child slot p ^ oct - the
        // ballot's bits 0..7 are the hit INNER children in front-to-back order, what the group word wants - and lane 8 + c reads
        // slot c for the leaves.  v_sh moves the lane's bit of  imask | leafmask << 8  to the sign.
        const uint32_t v_slot = lane < 8u ? lane ^ oct : lane & 7u;
        const uint32_t v_addr = lane < 16u ? v_slot * 32u : 0u;
        const uint32_t v_sh = lane < 8u ? 31u - v_slot : lane < 16u ? 31u - lane : 0u;  // lanes >= 16: bit 31 of the word, always 0
        const uint32_t p_sh = lane < 8u ? 31u - (lane ^ oct) : 0u;  // (hybrid) bit p ^ oct of a slot-ordered mask to the sign; other lanes: see the & 0xff
        const uint32_t s_sh = 23u - 8u * (p_axis < 2u ? p_axis : 2u);  // this lane's scale exponent of header word 3 to the exponent field
        const float w_x = p_axis == 0u ? 1.0f : 0.0f, w_y = p_axis == 1u ? 1.0f : 0.0f, w_z = p_axis >= 2u ? 1.0f : 0.0f;

        // traversal stack in two VGPRs, entry i in lane i (v_writelane / v_readlane with a scalar index: no LDS, no exec games);
        // render_pt_common sends trees that may need more than kPacketStackEntries (< 64) entries to the per-lane kernel
        int stx = 0, sty = 0;                // entry 0 = (0, 0): the end marker
        uint32_t sp = 1, sp_max = 0;
        uint32_t gx = 0u, gy = 0x80000000u;  // the root group
        for (;;) {
        do {  // (gx, gy) holds at least one child; the inner loop descends while some child is entered
            const uint32_t lz = (uint32_t)__builtin_clz(gy);  // 0..7: the front-most pending child is bit 31 - lz
            const uint32_t hits = gy;
            gy &= ~(0x80000000u >> lz);
            if (gy > 0x00ffffffu) {  // remaining siblings
                // (no builtin for v_writelane in this compiler; below gfx10 the lane select has to come through m0 when the value is a
                // scalar register - one constant-bus operand.  m0 is a reserved register that cannot be named as a clobber, so the
                // statement puts back what it found there)
                uint32_t m0_saved;
                asm("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %5\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\ts_mov_b32 m0, %2"
                    : "+v"(stx), "+v"(sty), "=&s"(m0_saved)
                    : "s"(gx), "s"(gy), "s"(sp));
                sp++;
                sp_max = sp_max > sp ? sp_max : sp;
            }
            const uint32_t slot = (7u - lz) ^ oct;
            const uint32_t node = gx + (uint32_t)__builtin_popcount(hits & ((1u << slot) - 1u));
            const uint32_t* __restrict__ nd = reinterpret_cast<const uint32_t*>(sc.nodes) + (size_t)uniform(node) * 20u;
            if (COUNT) n_nodes++;
            const uint32_t q = reinterpret_cast<const uint8_t*>(nd)[q_off];  // issued ahead of the header's wait
            u32x8 hdr;
            asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hdr) : "s"(nd) : "memory");
            const float px_ = __uint_as_float(hdr[0]), py_ = __uint_as_float(hdr[1]), pz_ = __uint_as_float(hdr[2]);
            const uint32_t w3 = hdr[3], child_base = hdr[4], tri_base = hdr[5], leafmask = hdr[6] & 0xffu;
            const uint32_t imask = w3 >> 24;
            // this lane's plane: p[axis] + q * 2^e[axis]; the axis is picked with 0 / 1 weights (three fast multiply-adds, one scalar
            // operand each) instead of selects (the constant bus takes one scalar register per instruction)
            const float ps = __uint_as_float((w3 << s_sh) & 0x7f800000u);
            const float pp = __builtin_fmaf(w_z, pz_, __builtin_fmaf(w_y, py_, w_x * px_));
            const float plane = __builtin_fmaf((float)q, ps, pp);
            if (!PURE) {
                __builtin_amdgcn_wave_barrier();  // the previous node's plane reads are done (one wave: DS ops run in order)
                planes[lane] = plane;             // child c: [near x y z, -, far x y z, -]
                __builtin_amdgcn_wave_barrier();
            }
            // (1) interval test: lane 8c + k bounds its plane, quad maxima, near + (-far) <= 0 in lanes 8c..8c+3
            const float u = plane - o_ax;
            const float lo = fmin_(u * ia_a, u * ia_b);
            float qm = __builtin_fmaf(__builtin_fabsf(lo), -0x1p-21f, lo - ia_m);
            RT_DPP_MAX(qm, "quad_perm:[1,0,3,2]");
            RT_DPP_MAX(qm, "quad_perm:[2,3,0,1]");  // every lane holds its quad's maximum
            float gap;  // max(lower bounds of t_near, 0) - min(upper bounds of t_far, cap): all finite
            asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(gap) : "v"(qm));
            const float gap_c = __int_as_float(__builtin_amdgcn_ds_bpermute((int)v_addr, __float_as_int(gap)));
            const uint32_t masks = imask | ((PURE ? leafmask : imask | leafmask) << 8);  // bits 8..15: what step (2) / the triangle loop may be handed
            // (two ballots and a scalar AND: the AND of two i1 would go through v_cndmask and a third compare)
            const uint32_t verdict = (uint32_t)__builtin_amdgcn_ballot_w64(gap_c <= 0.0f) & (uint32_t)__builtin_amdgcn_ballot_w64((int)(masks << v_sh) < 0);
            uint32_t any;  // bit c: leaf / inner child slot c is entered (slot order)
            uint32_t inner_hits;
            if (PURE) {
                any = verdict >> 8;
                inner_hits = verdict << 24;
            } else {  // (2) the rays' own slab tests of the children that passed
                any = 0;
                for (uint32_t m = (verdict >> 8) & 0xffu; m; m &= m - 1u) {
                    const uint32_t c = (uint32_t)__builtin_ctz(m);
                    const f4v a = planes4[2u * c], b = planes4[2u * c + 1u];
                    const float tn = fmax_(fmax_(__builtin_fmaf(a.x, inv.x, noi.x), __builtin_fmaf(a.y, inv.y, noi.y)), fmax_(__builtin_fmaf(a.z, inv.z, noi.z), 0.0f));
                    const float tf = fmin_(fmin_(__builtin_fmaf(b.x, inv.x, noi.x), __builtin_fmaf(b.y, inv.y, noi.y)), fmin_(__builtin_fmaf(b.z, inv.z, noi.z), best.t));
                    // (scalar text: a uniform i1 is kept as a lane mask and comes back through v_cndmask + v_readfirstlane)
                    const unsigned long long hit_mask = __builtin_amdgcn_ballot_w64(tn <= tf) & act_mask;
                    uint32_t some;
                    asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(some) : "s"(hit_mask) : "scc");
                    any |= some << c;
                }
                // front-to-back order of the inner hits: lane p < 8 looks at bit p ^ oct, the ballot is the permuted byte
                inner_hits = (uint32_t)__builtin_amdgcn_ballot_w64((int)((any & imask) << p_sh) < 0 && lane < 8u) << 24;
            }
            unsigned long long improved = 0ull;  // wave-uniform: lanes whose best hit moved at this node
            for (uint32_t lh = any & leafmask; lh; lh &= lh - 1u) {  // the single triangles of the leaf slots that are entered
                const uint32_t c = (uint32_t)__builtin_ctz(lh);
                const uint32_t li = tri_base + (uint32_t)__builtin_popcount(leafmask & ((1u << c) - 1u));
                const float* __restrict__ tp = reinterpret_cast<const float*>(sc.tris) + (size_t)li * 12u;  // wave-uniform address
                if (COUNT) n_tris++;
                // tri_test() with the same operations in the same order, but as straight-line code with lane masks and ONE wave-uniform
                // way out: its per-lane early returns cost about sixty scalar instructions per triangle in exec-mask bookkeeping
                const v3 v0 = mk(tp[0], tp[1], tp[2]), e1 = mk(tp[3], tp[4], tp[5]), e2 = mk(tp[6], tp[7], tp[8]);
                const v3 pvec = cross(d, e2);
                const float det = dot(e1, pvec);
                const v3 tvec = o - v0;
                const float u = dot(tvec, pvec);
                const v3 qvec = cross(tvec, e1);
                const float v = dot(d, qvec);
                const float uv = u + v;
                // (ballots, combined as 64-bit scalars: boolean expressions come back as branches or through v_cndmask + v_cmp)
                const unsigned long long m_pos = __builtin_amdgcn_ballot_w64(det > 0.0f), m_nz = __builtin_amdgcn_ballot_w64(det != 0.0f);
                const unsigned long long out_pos = __builtin_amdgcn_ballot_w64(u < 0.0f) | __builtin_amdgcn_ballot_w64(v < 0.0f) | __builtin_amdgcn_ballot_w64(uv > det);
                const unsigned long long out_neg = __builtin_amdgcn_ballot_w64(u > 0.0f) | __builtin_amdgcn_ballot_w64(v > 0.0f) | __builtin_amdgcn_ballot_w64(uv < det);
                const unsigned long long inside = act_mask & m_nz & ((m_pos & ~out_pos) | (~m_pos & ~out_neg));
                if (inside == 0ull) continue;
                const float t = dot(e2, qvec) / det;
                const uint32_t id = __float_as_uint(tp[9]);
                const unsigned long long closer = __builtin_amdgcn_ballot_w64(t < best.t) | (__builtin_amdgcn_ballot_w64(t == best.t) & __builtin_amdgcn_ballot_w64(id < best.id));
                const unsigned long long take = inside & __builtin_amdgcn_ballot_w64(t > 0.0f) & closer;
                if (FARCAP) improved |= take;
                const bool mine = __builtin_amdgcn_inverse_ballot_w64(take);
                best.t = mine ? t : best.t;
                best.li = mine ? (int)li : best.li;
                best.id = mine ? id : best.id;
            }
            if (FARCAP && improved != 0ull) {
                const float maxbest = wave_max_nonneg(act ? best.t : 0.0f);
                if (pk == 7u) ia_m = maxbest;
            }
            gx = child_base;
            gy = inner_hits | imask;
        } while (gy > 0x00ffffffu);
            // no child entered: on with the newest pending group
            sp--;
            gx = (uint32_t)__builtin_amdgcn_readlane(stx, (int)sp);
            gy = (uint32_t)__builtin_amdgcn_readlane(sty, (int)sp);
            if (gy == 0u) break;
        }
        if (sp_max > 63u) overflow = 1;
    }
    if (alive) st.hit[pid] = make_float2(best.t, __int_as_float(best.li));
    if (COUNT && lane == 0) {  // records fetched once per wave
        atomicAdd(&stats[9], (unsigned long long)n_nodes);
        atomicAdd(&stats[10], (unsigned long long)n_tris);
        atomicAdd(&stats[8], 1ull);
    }
    if (overflow && lane == 0) atomicOr((unsigned int*)&stats[2], 1u);
}

// ---- shade ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAppendThreads, 8) void pt_shade(const PtScene sc, const PtFrame f, PtState st, const uint32_t* __restrict__ queue,
                                                           const uint32_t* __restrict__ count_ptr, uint32_t depth, uint32_t* __restrict__ next_queue,
                                                           uint32_t* __restrict__ next_ctr, uint32_t sort_rays) {
    __shared__ uint32_t lds[kSortBins + 40];
    // queue == nullptr (depth 0 behind the packet kernel, which has no generate stage and no queue): the items are the
    // path ids themselves; origin = camera, throughput 1, radiance 0 are known and not read from the path state
    const bool direct = queue == nullptr;
    const uint32_t n = direct ? f.n_paths : *count_ptr;
    const uint32_t stride = gridDim.x * kAppendThreads;
    // grid-stride over whole workgroups: the trip count is workgroup-uniform (barriers in block_append)
    for (uint32_t base = blockIdx.x * kAppendThreads; base < n; base += stride) {
        const uint32_t i = base + threadIdx.x;
        bool bounce = false, shadow = false;
        uint32_t pid = 0;
        float4 so = {}, sd = {}, scn = {};
        bool item = i < n;
        if (item && direct) {
            uint32_t px, py, lx, ly, k;
            item = slot_pixel(f, i / f.spp_batch, px, py, lx, ly, k);
        }
        if (item) {
            pid = direct ? i : queue[i];
            const float2 hrec = st.hit[pid];
            const int li = __float_as_int(hrec.y);
            const float4 rd = st.ray_d[pid];
            const float4 ro = direct ? make_float4(f.cam.pos[0], f.cam.pos[1], f.cam.pos[2], 0.0f) : st.ray_o[pid];
            const float4 T = direct ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : st.thr[pid];
            const v3 o = mk(ro.x, ro.y, ro.z), d = mk(rd.x, rd.y, rd.z);
            float4 L = direct ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : st.rad[pid];
            if (direct) st.rad[pid] = L;  // the shadow stage and pt_resolve read it; the branches below overwrite it where they add light
            if (li < 0) {  // left the scene
                L.x = __builtin_fmaf(T.x, f.sky[0], L.x);
                L.y = __builtin_fmaf(T.y, f.sky[1], L.y);
                L.z = __builtin_fmaf(T.z, f.sky[2], L.z);
                st.rad[pid] = L;
            } else {
                const float4 ta = sc.tris[(size_t)li * 3 + 0], tb = sc.tris[(size_t)li * 3 + 1], tc = sc.tris[(size_t)li * 3 + 2];
                if (__float_as_uint(tc.z) != 0u) {  // emissive (flag in the triangle record); lights are seen directly only by camera rays
                    if (depth == 0) {
                        const float4 em = sc.emission[li];
                        L.x = __builtin_fmaf(T.x, em.x, L.x);
                        L.y = __builtin_fmaf(T.y, em.y, L.y);
                        L.z = __builtin_fmaf(T.z, em.z, L.z);
                        st.rad[pid] = L;
                    }
                } else {
                    const float4 alb = sc.albedo[li];
                    v3 nrm = normalize(cross(mk(ta.w, tb.x, tb.y), mk(tb.z, tb.w, tc.x)));
                    if (dot(nrm, d) > 0.0f) nrm = -nrm;
                    const v3 pt = fma3(d, hrec.x, o);
                    const v3 po = fma3(nrm, f.ray_eps, pt);
                    // path id -> rng key
                    const uint32_t slot = pid / f.spp_batch, s = f.sample0 + (pid - slot * f.spp_batch);
                    uint32_t px, py, lx, ly, k;
                    slot_pixel(f, slot, px, py, lx, ly, k);
                    const uint32_t key = path_key(py * f.width + px, s, f.seed);
                    if (sc.n_lights > 0) {  // next-event estimation
                        uint32_t kk = (uint32_t)(rnd(key, depth, 2) * (float)sc.n_lights);
                        if (kk > sc.n_lights - 1) kk = sc.n_lights - 1;
                        const uint32_t lt = sc.lights[kk];
                        const float su = __builtin_sqrtf(rnd(key, depth, 3)), u2 = rnd(key, depth, 4);
                        const float b1 = su * (1.0f - u2), b2 = su * u2;
                        const float4 la = sc.tris[(size_t)lt * 3 + 0], lb = sc.tris[(size_t)lt * 3 + 1], lc = sc.tris[(size_t)lt * 3 + 2];
                        const v3 lv0 = mk(la.x, la.y, la.z), le1 = mk(la.w, lb.x, lb.y), le2 = mk(lb.z, lb.w, lc.x);
                        const v3 q = mk(__builtin_fmaf(le2.x, b2, __builtin_fmaf(le1.x, b1, lv0.x)), __builtin_fmaf(le2.y, b2, __builtin_fmaf(le1.y, b1, lv0.y)),
                                        __builtin_fmaf(le2.z, b2, __builtin_fmaf(le1.z, b1, lv0.z)));
                        const v3 wi = q - po;
                        const float d2 = dot(wi, wi);
                        const v3 nl = cross(le1, le2);
                        const float cs = dot(nrm, wi), cl = __builtin_fabsf(dot(nl, wi));
                        if (cs > 0.0f && cl > 0.0f && d2 > 0.0f) {
                            const float w = ((cs * cl) * ((float)sc.n_lights * 0.15915494f)) / (d2 * d2);
                            const float4 le = sc.emission[lt];
                            shadow = true;
                            so = make_float4(po.x, po.y, po.z, __uint_as_float(pid));
                            sd = make_float4(wi.x, wi.y, wi.z, 0.0f);
                            scn = make_float4(((T.x * alb.x) * le.x) * w, ((T.y * alb.y) * le.y) * w, ((T.z * alb.z) * le.z) * w, 0.0f);
                        }
                    }
                    if (depth < f.bounces) {
                        const v3 nd = cosine_dir(nrm, rnd(key, depth, 5), rnd(key, depth, 6));
                        bounce = true;
                        st.ray_o[pid] = make_float4(po.x, po.y, po.z, 0.0f);
                        st.ray_d[pid] = make_float4(nd.x, nd.y, nd.z, 0.0f);
                        st.thr[pid] = make_float4(T.x * alb.x, T.y * alb.y, T.z * alb.z, 0.0f);
                    }
                }
            }
        }
        uint32_t bi, si;
        if (sort_rays == 2u) {  // wave-uniform
            uint32_t kb = 0, ks = 0;
            if (bounce) {
                const float4 rd = st.ray_d[pid];
                kb = bounce_work_key(mk(rd.x, rd.y, rd.z));
            }
            if (shadow) ks = shadow_work_key(mk(sd.x, sd.y, sd.z));
            bi = block_append_sorted<64>(bounce, kb, &next_ctr[PT_CTR_COUNT], lds);
            si = block_append_sorted<64>(shadow, ks, &next_ctr[PT_CTR_SHADOW_COUNT], lds);
        } else if (sort_rays) {
            uint32_t kb = 0, ks = 0;
            if (bounce) {
                const float4 ro = st.ray_o[pid], rd = st.ray_d[pid];
                kb = ray_sort_key(sc.nodes, mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z));
            }
            if (shadow) ks = ray_sort_key(sc.nodes, mk(so.x, so.y, so.z), mk(sd.x, sd.y, sd.z));
            bi = block_append_sorted(bounce, kb, &next_ctr[PT_CTR_COUNT], lds);
            si = block_append_sorted(shadow, ks, &next_ctr[PT_CTR_SHADOW_COUNT], lds);
        } else {
            bi = block_append(bounce, &next_ctr[PT_CTR_COUNT], lds);
            si = block_append(shadow, &next_ctr[PT_CTR_SHADOW_COUNT], lds);
        }
        if (bounce) next_queue[bi] = pid;
        if (shadow) {
            st.sh_o[si] = so;
            st.sh_d[si] = sd;
            st.sh_c[si] = scn;
        }
    }
}

// ---- resolve --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_resolve(const PtFrame f, PtState st, float* __restrict__ acc, float* __restrict__ dst, uint32_t tile_major) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= f.n_slots) return;
    uint32_t px, py, lx, ly, k;
    if (!slot_pixel(f, slot, px, py, lx, ly, k)) return;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (f.sample0 > 0) {
        r = acc[(size_t)slot * 3 + 0];
        g = acc[(size_t)slot * 3 + 1];
        b = acc[(size_t)slot * 3 + 2];
    }
    for (uint32_t s = 0; s < f.spp_batch; s++) {  // spec §6.6: samples are summed in index order
        const float4 L = st.rad[(size_t)slot * f.spp_batch + s];
        r += L.x;
        g += L.y;
        b += L.z;
    }
    if (f.sample0 + f.spp_batch < f.spp_total) {
        acc[(size_t)slot * 3 + 0] = r;
        acc[(size_t)slot * 3 + 1] = g;
        acc[(size_t)slot * 3 + 2] = b;
        return;
    }
    const float inv = (float)f.spp_total;
    const size_t idx = tile_major ? ((size_t)k * (RT_TILE * RT_TILE) + (size_t)ly * RT_TILE + lx) : ((size_t)py * f.width + px);
    dst[idx * 3 + 0] = r / inv;
    dst[idx * 3 + 1] = g / inv;
    dst[idx * 3 + 2] = b / inv;
}

// ---- test hook: trace a batch of caller-supplied rays ---------------------------------------------
template <bool COUNT>
__device__ __forceinline__ void trace_one_ray(const PtScene& sc, const float* __restrict__ origins, const float* __restrict__ dirs, uint32_t i,
                                              int any_hit, float* __restrict__ t_out, int* __restrict__ tri_out, TravStack& stk, const uint8_t* perm_lut,
                                              TravCounters& tc) {
    const v3 o = mk(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]), d = mk(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]);
    if (any_hit) {
        Hit h{kShadowTmax, -1, 0u};
        const bool occ = traverse<true, COUNT>(sc.nodes, sc.tris, perm_lut, o, d, stk, h, tc);
        t_out[i] = occ ? 1.0f : 0.0f;
        tri_out[i] = occ ? 1 : 0;
    } else {
        Hit h{__builtin_inff(), -1, 0xffffffffu};
        traverse<false, COUNT>(sc.nodes, sc.tris, perm_lut, o, d, stk, h, tc);
        t_out[i] = h.t;
        tri_out[i] = h.li < 0 ? -1 : (int)h.id;
    }
}

// counts != nullptr: per-ray node fetches and triangle tests (counts[2i], counts[2i+1]) of the very step
// functions the render kernels run, for the host-side cross-check of the traversal statistics
template <bool COUNT>
__global__ __launch_bounds__(256) void pt_trace_rays(const PtScene sc, const float* __restrict__ origins, const float* __restrict__ dirs, uint32_t n,
                                                     int any_hit, float* __restrict__ t_out, int* __restrict__ tri_out, uint32_t* __restrict__ counts,
                                                     const StackCfg sk) {
    extern __shared__ unsigned long long lds_stack[];
    __shared__ uint8_t perm_lut[2048];
    build_perm_lut(perm_lut);
    // grid-stride so the spill columns (one per launched thread) stay within sk.spill_stride
    const size_t gtid = (size_t)blockIdx.x * 256u + threadIdx.x;
    TravStack stk{(lds_u64*)&lds_stack[threadIdx.x], sk.spill + gtid, sk.spill_stride, sk.lds_cap, sk.spill_cap, 0};
    for (uint32_t i = (uint32_t)gtid; i < n; i += gridDim.x * 256u) {
        TravCounters tc{0, 0, 0};
        trace_one_ray<COUNT>(sc, origins, dirs, i, any_hit, t_out, tri_out, stk, perm_lut, tc);
        if (COUNT) {
            counts[2 * (size_t)i] = tc.nodes;
            counts[2 * (size_t)i + 1] = tc.tris;
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------
int launch_pt_generate(Ctx* c, const PtFrame& f, const PtState& st, uint32_t* queue, uint32_t* ctr) {
    hipLaunchKernelGGL(pt_generate, dim3((f.n_paths + kAppendThreads - 1u) / kAppendThreads), dim3(kAppendThreads), 0, c->stream, f, st, queue, ctr);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_trace(Ctx* c, const PtScene& sc, const PtState& st, const uint32_t* queue, const uint32_t* count_ptr, uint32_t* head,
                    unsigned long long* stats, bool any_hit, bool count, uint32_t grid, const StackCfg& stack_cap, uint32_t refill_min, uint32_t tri_mode,
                    uint32_t tri_cfg) {
    if (stack_cap.lds_cap < 1 || stack_cap.lds_cap > 160 || (size_t)grid * 256u > stack_cap.spill_stride)
        return c->fail(RT_ERR_INVALID, "bad traversal stack configuration");
    const dim3 g(grid), b(256);
    const size_t lds = (size_t)stack_cap.lds_cap * 256 * sizeof(unsigned long long);
#define RT_LAUNCH_TRACE(ANY, COUNT, MODE) \
    hipLaunchKernelGGL((pt_trace<ANY, COUNT, MODE>), g, b, lds, c->stream, sc, st, queue, count_ptr, head, stats, stack_cap, refill_min, tri_cfg)
#define RT_LAUNCH_TRACE_MODE(ANY, COUNT)                         \
    do {                                                         \
        if (tri_mode == TRI_POOL) RT_LAUNCH_TRACE(ANY, COUNT, TRI_POOL); \
        else if (tri_mode == TRI_DEFER) RT_LAUNCH_TRACE(ANY, COUNT, TRI_DEFER); \
        else if (tri_mode == TRI_INLINE_PF) RT_LAUNCH_TRACE(ANY, COUNT, TRI_INLINE_PF); \
        else RT_LAUNCH_TRACE(ANY, COUNT, TRI_INLINE);            \
    } while (0)
    if (any_hit) {
        if (count) RT_LAUNCH_TRACE_MODE(true, true);
        else RT_LAUNCH_TRACE_MODE(true, false);
    } else {
        if (count) RT_LAUNCH_TRACE_MODE(false, true);
        else RT_LAUNCH_TRACE_MODE(false, false);
    }
#undef RT_LAUNCH_TRACE_MODE
#undef RT_LAUNCH_TRACE
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_trace_fused(Ctx* c, const PtScene& sc, const PtState& st, const uint32_t* queue, const uint32_t* closest_count, uint32_t* closest_head,
                          const uint32_t* shadow_count, uint32_t* shadow_head, unsigned long long* stats, bool count, uint32_t grid,
                          const StackCfg& stack_cap, uint32_t refill_min, uint32_t tri_mode, uint32_t tri_cfg) {
    if (stack_cap.lds_cap < 1 || stack_cap.lds_cap > 160 || (size_t)grid * 256u > stack_cap.spill_stride)
        return c->fail(RT_ERR_INVALID, "bad traversal stack configuration");
    const size_t lds = (size_t)stack_cap.lds_cap * 256 * sizeof(unsigned long long);
#define RT_LAUNCH_FUSED(COUNT, MODE)                                                                                                                  \
    hipLaunchKernelGGL((pt_trace_fused<COUNT, MODE>), dim3(grid), dim3(256), lds, c->stream, sc, st, queue, closest_count, closest_head, shadow_count, \
                       shadow_head, stats, stack_cap, refill_min, tri_cfg)
    if (tri_mode == TRI_POOL) {
        if (count) RT_LAUNCH_FUSED(true, TRI_POOL);
        else RT_LAUNCH_FUSED(false, TRI_POOL);
    } else if (tri_mode == TRI_DEFER) {
        if (count) RT_LAUNCH_FUSED(true, TRI_DEFER);
        else RT_LAUNCH_FUSED(false, TRI_DEFER);
    } else if (tri_mode == TRI_INLINE_PF) {
        if (count) RT_LAUNCH_FUSED(true, TRI_INLINE_PF);
        else RT_LAUNCH_FUSED(false, TRI_INLINE_PF);
    } else {
        if (count) RT_LAUNCH_FUSED(true, TRI_INLINE);
        else RT_LAUNCH_FUSED(false, TRI_INLINE);
    }
#undef RT_LAUNCH_FUSED
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

uint32_t pt_pool_lds_bytes(uint32_t tri_mode) { return tri_mode == TRI_POOL ? kPoolLdsBytes : tri_mode == TRI_INLINE_PF ? kPfLdsBytes : 0u; }

int launch_pt_trace_packet(Ctx* c, const PtScene& sc, const PtFrame& f, const PtState& st, unsigned long long* stats, bool count, uint32_t mode) {
    const dim3 g((f.n_paths + 255u) / 256u), b(256);
#define RT_LAUNCH_PACKET(K) hipLaunchKernelGGL(K, g, b, 0, c->stream, sc, f, st, stats)
    if (mode == PACKET_EXACT) {
        if (count) RT_LAUNCH_PACKET(pt_trace_packet<true>);
        else RT_LAUNCH_PACKET(pt_trace_packet<false>);
    } else if (mode == PACKET_INTERVAL_ONLY) {
        if (count) RT_LAUNCH_PACKET((pt_trace_packet_ia<true, true, true>));
        else RT_LAUNCH_PACKET((pt_trace_packet_ia<false, true, true>));
    } else if (mode == PACKET_INTERVAL_NOCAP) {
        if (count) RT_LAUNCH_PACKET((pt_trace_packet_ia<true, false, false>));
        else RT_LAUNCH_PACKET((pt_trace_packet_ia<false, false, false>));
    } else {
        if (count) RT_LAUNCH_PACKET((pt_trace_packet_ia<true, false, true>));
        else RT_LAUNCH_PACKET((pt_trace_packet_ia<false, false, true>));
    }
#undef RT_LAUNCH_PACKET
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_shade(Ctx* c, const PtScene& sc, const PtFrame& f, const PtState& st, const uint32_t* queue, const uint32_t* count_ptr,
                    uint32_t depth, uint32_t* next_queue, uint32_t* next_ctr, uint32_t grid, uint32_t sort_rays) {
    hipLaunchKernelGGL(pt_shade, dim3(grid), dim3(kAppendThreads), 0, c->stream, sc, f, st, queue, count_ptr, depth, next_queue, next_ctr,
                       sort_rays);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_resolve(Ctx* c, const PtFrame& f, const PtState& st, float* acc, float* dst, int tile_major) {
    hipLaunchKernelGGL(pt_resolve, dim3((f.n_slots + 255u) / 256u), dim3(256), 0, c->stream, f, st, acc, dst, (uint32_t)(tile_major ? 1 : 0));
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_trace_rays(Ctx* c, const PtScene& sc, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out, int* tri_out,
                         uint32_t* counts, const StackCfg& sk, uint32_t grid) {
    if ((size_t)grid * 256u > sk.spill_stride) return c->fail(RT_ERR_INVALID, "bad traversal stack configuration");
    const size_t lds = (size_t)sk.lds_cap * 256 * sizeof(unsigned long long);
    if (counts) hipLaunchKernelGGL(pt_trace_rays<true>, dim3(grid), dim3(256), lds, c->stream, sc, origins, dirs, n, any_hit, t_out, tri_out, counts, sk);
    else hipLaunchKernelGGL(pt_trace_rays<false>, dim3(grid), dim3(256), lds, c->stream, sc, origins, dirs, n, any_hit, t_out, tri_out, counts, sk);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

}  // namespace rt
