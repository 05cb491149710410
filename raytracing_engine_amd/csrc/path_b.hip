// path_b.hip — wavefront path tracer over a triangle BVH (BASELINE.json configs[2..4]).
//
// NO REFERENCE COUNTERPART: the reference has no triangles, BVH, RNG, spp or bounces (SURVEY.md
// §0); only the camera model is the reference's (shaders/fragment.glsl:129-133,
// shaders/utilities.glsl:26-29).  This file implements the specification of DESIGN.md §6; parity
// is against oracle B and is "unpinned by the reference".
//
// Structure (one launch per ray stage, all queue sizes stay on the device):
//   pt_generate       camera rays for every (pixel, sample) of the owned tiles -> path state + queue 0
//   pt_trace<closest> persistent waves pull 64-ray chunks off the queue, BVH2 traversal with a
//                     per-lane stack in LDS, ray/triangle tests, writes (t, triangle)
//   pt_shade          emission / sky / next-event estimation / cosine bounce; survivors are appended to
//                     the next queue and shadow rays to the shadow queue with wave ballot +
//                     prefix-popcount compaction (one atomic per wave)
//   pt_trace<any>     shadow rays: any-hit traversal, unoccluded contributions added to the path
//   pt_resolve        per pixel: samples summed in index order, divided by spp
// Memory: path state is SoA of float4 (16 B per lane per array = widest coalesced access), BVH nodes
// are 64-byte child-pair records (one fetch per traversal step), triangles 48-byte records in leaf
// order.
#include "rt_device_math.h"
#include "rt_internal.h"

namespace rt {
using namespace rtk;

constexpr int kStack = 32;  // >= kBvhMaxDepth + 1 (bvh_build.h)
constexpr int kSentinel = (int)0x80000000;
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

__device__ __forceinline__ v3 safe_inv(v3 d) {
    const float x = __builtin_fabsf(d.x) > 1e-20f ? d.x : __builtin_copysignf(1e-20f, d.x);
    const float y = __builtin_fabsf(d.y) > 1e-20f ? d.y : __builtin_copysignf(1e-20f, d.y);
    const float z = __builtin_fabsf(d.z) > 1e-20f ? d.z : __builtin_copysignf(1e-20f, d.z);
    return mk(1.0f / x, 1.0f / y, 1.0f / z);
}

// conservative slab test against a padded box
__device__ __forceinline__ bool box_test(float lox, float loy, float loz, float hix, float hiy, float hiz, v3 o, v3 inv, float tmax,
                                         float& tn) {
    const float t0x = (lox - o.x) * inv.x, t1x = (hix - o.x) * inv.x;
    const float t0y = (loy - o.y) * inv.y, t1y = (hiy - o.y) * inv.y;
    const float t0z = (loz - o.z) * inv.z, t1z = (hiz - o.z) * inv.z;
    tn = fmax_(fmax_(fmin_(t0x, t1x), fmin_(t0y, t1y)), fmax_(fmin_(t0z, t1z), 0.0f));
    const float tf = fmin_(fmin_(fmax_(t0x, t1x), fmax_(t0y, t1y)), fmin_(fmax_(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

struct Hit {
    float t;
    int li;       // leaf-order triangle index, -1 = none
    uint32_t id;  // original triangle index (tie-break)
};

// BVH2 traversal, "while-while": descend through inner nodes until a leaf reference comes up, then
// test its triangles.  `stack` points at this thread's column of the LDS stack (stride 256).
template <bool ANY, bool COUNT>
__device__ __forceinline__ bool traverse(const float4* __restrict__ nodes, const float4* __restrict__ tris, v3 o, v3 d, int* stack,
                                         Hit& best, uint32_t& n_nodes, uint32_t& n_tris, uint32_t& overflow) {
    const v3 inv = safe_inv(d);
    float tmax = ANY ? kShadowTmax : best.t;
    int sp = 0;
    int cur = 0;
    while (cur != kSentinel) {
        while (cur >= 0) {
            const float4 q0 = nodes[(size_t)cur * 4 + 0], q1 = nodes[(size_t)cur * 4 + 1], q2 = nodes[(size_t)cur * 4 + 2],
                         q3 = nodes[(size_t)cur * 4 + 3];
            if (COUNT) n_nodes++;
            float tn0, tn1;
            const bool h0 = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, inv, tmax, tn0);
            const bool h1 = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, inv, tmax, tn1);
            const int r0 = __float_as_int(q3.x), r1 = __float_as_int(q3.y);
            if (h0 && h1) {
                const bool swap = tn1 < tn0;
                const int near = swap ? r1 : r0, far = swap ? r0 : r1;
                if (sp < kStack) stack[(sp++) * 256] = far;
                else overflow = 1;
                cur = near;
            } else if (h0) {
                cur = r0;
            } else if (h1) {
                cur = r1;
            } else {
                cur = sp ? stack[(--sp) * 256] : kSentinel;
            }
        }
        if (cur != kSentinel) {
            const uint32_t ref = ~(uint32_t)cur;
            const uint32_t first = ref >> 2, cnt = (ref & 3u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                const uint32_t li = first + i;
                const float4 a = tris[(size_t)li * 3 + 0], b = tris[(size_t)li * 3 + 1], c = tris[(size_t)li * 3 + 2];
                if (COUNT) n_tris++;
                float t;
                if (tri_test(o, d, mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), t) && t > 0.0f) {
                    if (ANY) {
                        if (t < kShadowTmax) return true;
                    } else {
                        const uint32_t id = __float_as_uint(c.y);
                        if (t < best.t || (t == best.t && id < best.id)) {
                            best.t = t;
                            best.li = (int)li;
                            best.id = id;
                            tmax = t;
                        }
                    }
                }
            }
            cur = sp ? stack[(--sp) * 256] : kSentinel;
        }
    }
    return false;
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

__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t* counter) {
    // active-lane compaction: ballot + prefix popcount, one atomic per wave
    const unsigned long long mask = __ballot(want);
    uint32_t base = 0;
    const uint32_t lane = threadIdx.x & 63u;
    if (mask) {
        if (lane == (uint32_t)__builtin_ctzll(mask)) base = atomicAdd(counter, (uint32_t)__popcll(mask));
        base = __shfl(base, __builtin_ctzll(mask));
    }
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// ---- generate -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_generate(const PtFrame f, PtState st, uint32_t* __restrict__ queue, uint32_t* __restrict__ ctr) {
    const uint32_t pid = blockIdx.x * 256u + threadIdx.x;
    bool alive = false;
    if (pid < f.n_paths) {
        const uint32_t slot = pid / f.spp_batch, s = f.sample0 + (pid - slot * f.spp_batch);
        uint32_t px, py, lx, ly, k;
        if (slot_pixel(f, slot, px, py, lx, ly, k)) {
            alive = true;
            const uint32_t key = path_key(py * f.width + px, s, f.seed);
            // camera ray: fragment.glsl:129-133 with the pixel-centre 0.5 replaced by a random offset
            const float nx = ((((float)px + rnd(key, 0, 0)) * 2.0f) / (float)f.width - 1.0f) * f.cam.ratio[0];
            const float ny = ((((float)py + rnd(key, 0, 1)) * 2.0f) / (float)f.height - 1.0f) * f.cam.ratio[1];
            const v3 d = normalize(rotate_q(f.cam.rot[0], f.cam.rot[1], f.cam.rot[2], f.cam.rot[3], mk(nx, 1.0f, ny)));
            st.ray_o[pid] = make_float4(f.cam.pos[0], f.cam.pos[1], f.cam.pos[2], 0.0f);
            st.ray_d[pid] = make_float4(d.x, d.y, d.z, 0.0f);
            st.thr[pid] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        }
        st.rad[pid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const uint32_t idx = wave_append(alive, &ctr[PT_CTR_COUNT]);
    if (alive) queue[idx] = pid;
}

// ---- trace ----------------------------------------------------------------------------------------
// Persistent waves: every wave pulls chunks of 64 queue entries until the queue is drained, so the
// grid is sized for the machine, not for the (device-resident) queue length.
template <bool ANY, bool COUNT>
__global__ __launch_bounds__(256) void pt_trace(const PtScene sc, PtState st, const uint32_t* __restrict__ queue,
                                                const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ head,
                                                unsigned long long* __restrict__ stats) {
    __shared__ int lds_stack[kStack * 256];
    int* stack = &lds_stack[threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = *count_ptr;
    uint32_t n_nodes = 0, n_tris = 0, overflow = 0;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;  // wave-uniform exit: every wave reaches it once the queue is drained
        const uint32_t i = base + lane;
        if (i < n) {
            if (ANY) {
                const float4 so = st.sh_o[i], sd = st.sh_d[i];
                Hit h{kShadowTmax, -1, 0u};
                const bool occ = traverse<true, COUNT>(sc.nodes, sc.tris, mk(so.x, so.y, so.z), mk(sd.x, sd.y, sd.z), stack, h, n_nodes, n_tris, overflow);
                if (!occ) {
                    const uint32_t pid = __float_as_uint(so.w);
                    const float4 c = st.sh_c[i];
                    float4 L = st.rad[pid];
                    L.x += c.x;
                    L.y += c.y;
                    L.z += c.z;
                    st.rad[pid] = L;
                }
            } else {
                const uint32_t pid = queue[i];
                const float4 ro = st.ray_o[pid], rd = st.ray_d[pid];
                Hit h{__builtin_inff(), -1, 0xffffffffu};
                traverse<false, COUNT>(sc.nodes, sc.tris, mk(ro.x, ro.y, ro.z), mk(rd.x, rd.y, rd.z), stack, h, n_nodes, n_tris, overflow);
                st.hit[pid] = make_float2(h.t, __int_as_float(h.li));
            }
        }
    }
    if (COUNT) {
        // wave reduction of the traversal counters, one atomic per wave
        unsigned long long a = n_nodes, b = n_tris;
        for (int off = 32; off > 0; off >>= 1) {
            a += __shfl_down(a, off);
            b += __shfl_down(b, off);
        }
        if (lane == 0) {
            atomicAdd(&stats[ANY ? 4 : 0], a);
            atomicAdd(&stats[ANY ? 5 : 1], b);
        }
    }
    if (overflow) atomicOr((unsigned int*)&stats[2], 1u);
}

// ---- shade ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_shade(const PtScene sc, const PtFrame f, PtState st, const uint32_t* __restrict__ queue,
                                                const uint32_t* __restrict__ count_ptr, uint32_t depth, uint32_t* __restrict__ next_queue,
                                                uint32_t* __restrict__ next_ctr) {
    const uint32_t n = *count_ptr;
    const uint32_t stride = gridDim.x * 256u;
    // grid-stride over whole waves so the ballots below always see a full, converged wave
    for (uint32_t base = blockIdx.x * 256u + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + (threadIdx.x & 63u);
        bool bounce = false, shadow = false;
        uint32_t pid = 0;
        float4 so = {}, sd = {}, scn = {};
        if (i < n) {
            pid = queue[i];
            const float2 hrec = st.hit[pid];
            const int li = __float_as_int(hrec.y);
            const float4 ro = st.ray_o[pid], rd = st.ray_d[pid], T = st.thr[pid];
            const v3 o = mk(ro.x, ro.y, ro.z), d = mk(rd.x, rd.y, rd.z);
            float4 L = st.rad[pid];
            if (li < 0) {  // left the scene
                L.x = __builtin_fmaf(T.x, f.sky[0], L.x);
                L.y = __builtin_fmaf(T.y, f.sky[1], L.y);
                L.z = __builtin_fmaf(T.z, f.sky[2], L.z);
                st.rad[pid] = L;
            } else {
                const float4 em = sc.emission[li];
                if (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) {  // lights are seen directly only by camera rays
                    if (depth == 0) {
                        L.x = __builtin_fmaf(T.x, em.x, L.x);
                        L.y = __builtin_fmaf(T.y, em.y, L.y);
                        L.z = __builtin_fmaf(T.z, em.z, L.z);
                        st.rad[pid] = L;
                    }
                } else {
                    const float4 alb = sc.albedo[li];
                    const float4 ta = sc.tris[(size_t)li * 3 + 0], tb = sc.tris[(size_t)li * 3 + 1], tc = sc.tris[(size_t)li * 3 + 2];
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
        const uint32_t bi = wave_append(bounce, &next_ctr[PT_CTR_COUNT]);
        if (bounce) next_queue[bi] = pid;
        const uint32_t si = wave_append(shadow, &next_ctr[PT_CTR_SHADOW_COUNT]);
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
__global__ __launch_bounds__(256) void pt_trace_rays(const PtScene sc, const float* __restrict__ origins, const float* __restrict__ dirs, uint32_t n,
                                                     int any_hit, float* __restrict__ t_out, int* __restrict__ tri_out) {
    __shared__ int lds_stack[kStack * 256];
    int* stack = &lds_stack[threadIdx.x];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const v3 o = mk(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]), d = mk(dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]);
    uint32_t a = 0, b = 0, ov = 0;
    if (any_hit) {
        Hit h{kShadowTmax, -1, 0u};
        const bool occ = traverse<true, false>(sc.nodes, sc.tris, o, d, stack, h, a, b, ov);
        t_out[i] = occ ? 1.0f : 0.0f;
        tri_out[i] = occ ? 1 : 0;
    } else {
        Hit h{__builtin_inff(), -1, 0xffffffffu};
        traverse<false, false>(sc.nodes, sc.tris, o, d, stack, h, a, b, ov);
        t_out[i] = h.t;
        tri_out[i] = h.li < 0 ? -1 : (int)h.id;
    }
}

// ---- launchers ------------------------------------------------------------------------------------
int launch_pt_generate(Ctx* c, const PtFrame& f, const PtState& st, uint32_t* queue, uint32_t* ctr) {
    hipLaunchKernelGGL(pt_generate, dim3((f.n_paths + 255u) / 256u), dim3(256), 0, c->stream, f, st, queue, ctr);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_trace(Ctx* c, const PtScene& sc, const PtState& st, const uint32_t* queue, const uint32_t* count_ptr, uint32_t* head,
                    unsigned long long* stats, bool any_hit, bool count, uint32_t grid) {
    const dim3 g(grid), b(256);
    if (any_hit) {
        if (count) hipLaunchKernelGGL((pt_trace<true, true>), g, b, 0, c->stream, sc, st, queue, count_ptr, head, stats);
        else hipLaunchKernelGGL((pt_trace<true, false>), g, b, 0, c->stream, sc, st, queue, count_ptr, head, stats);
    } else {
        if (count) hipLaunchKernelGGL((pt_trace<false, true>), g, b, 0, c->stream, sc, st, queue, count_ptr, head, stats);
        else hipLaunchKernelGGL((pt_trace<false, false>), g, b, 0, c->stream, sc, st, queue, count_ptr, head, stats);
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_shade(Ctx* c, const PtScene& sc, const PtFrame& f, const PtState& st, const uint32_t* queue, const uint32_t* count_ptr,
                    uint32_t depth, uint32_t* next_queue, uint32_t* next_ctr, uint32_t grid) {
    hipLaunchKernelGGL(pt_shade, dim3(grid), dim3(256), 0, c->stream, sc, f, st, queue, count_ptr, depth, next_queue, next_ctr);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_resolve(Ctx* c, const PtFrame& f, const PtState& st, float* acc, float* dst, int tile_major) {
    hipLaunchKernelGGL(pt_resolve, dim3((f.n_slots + 255u) / 256u), dim3(256), 0, c->stream, f, st, acc, dst, (uint32_t)(tile_major ? 1 : 0));
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_pt_trace_rays(Ctx* c, const PtScene& sc, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out, int* tri_out) {
    hipLaunchKernelGGL(pt_trace_rays, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, sc, origins, dirs, n, any_hit, t_out, tri_out);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

}  // namespace rt
