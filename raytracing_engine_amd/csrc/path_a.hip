// path_a.hip — the reference's hot path as gfx950 kernels: hierarchical sphere-SDF cone marching
// (one launch per pyramid level) and the shading pass with soft-shadow marching.
//
// Replaces (file:line in the reference repo):
//   shaders/compute.glsl:70-87 (main) + :34-68 (traceCone)      -> cone_level_kernel
//   shaders/fragment.glsl:127-187 (main) + :89-121 (shadowRay)  -> shade_kernel
//   shaders/utilities.glsl:26-29, 36-38                          -> rt_device_math.h, sphere_sdf
//
// Mapping to CDNA4: the reference's 8x8 workgroup (compute.glsl:5) is exactly one wave64, so one
// wave owns one 8x8 pixel tile (compact footprint = similar march trip counts = less divergence);
// a workgroup is 4 such waves.  The scene (<= 8 spheres) travels in the kernarg segment and lives
// in SGPRs; per-lane state is the 8 lazily refreshed distance bounds (VGPRs, fully unrolled, so
// no scratch).  These kernels are VALU/sqrt-bound (about 8 B of HBM traffic per thread); see
// DESIGN.md §5.
#include "rt_device_math.h"
#include "rt_internal.h"

namespace rt {
using namespace rtk;

// shaders/utilities.glsl:36-38   distance(p, s.pos) - s.size
// shaders/utilities.glsl:31-34  repeat(p, r) = mod(p + 0.5*r, r) - 0.5*r, mod(x, y) = x - y*floor(x/y),
// per axis where the period is > 0.  The reference defines it and never calls it: it is applied to the
// position of every SDF evaluation and of the surface normal (build-defined).  REP = false
// (the reference as shipped) compiles to nothing.
struct Rep {
    float x, y, z;
};
__device__ __forceinline__ float repeat1(float p, float r) {
    if (!(r > 0.0f)) return p;
    const float h = 0.5f * r, a = p + h;
    return (a - r * __builtin_floorf(a / r)) - h;
}
template <bool REP>
__device__ __forceinline__ v3 domain(v3 p, Rep r) {
    if (!REP) return p;
    return mk(repeat1(p.x, r.x), repeat1(p.y, r.y), repeat1(p.z, r.z));
}
template <bool REP>
__device__ __forceinline__ float sphere_sdf(v3 p, float4 s, Rep r) { return length(domain<REP>(p, r) - mk(s.x, s.y, s.z)) - s.w; }

// Does this rank own the RT_TILE^2 tile containing full-res pixel (x0, y0)?  Off-screen -> false.
__device__ __forceinline__ bool owns_pixel_tile(const Partition& part, uint32_t x0, uint32_t y0, uint32_t width,
                                                uint32_t height) {
    if (x0 >= width || y0 >= height) return false;
    const uint32_t t = (y0 / RT_TILE) * part.tiles_x + (x0 / RT_TILE);
    return t % part.n_ranks == part.rank;
}

// ---- shaders/compute.glsl:34-68 ---------------------------------------------------------------
// ALG 3 = the loop body compute.glsl ships (:46-65); ALG 1, 2 = shaders/tracing_algorithms.txt:2-13,
// :16-37 placed in the same loop (oracle_a.c trace_cone1/2 are the definitions).
template <int N, int ALG, bool REP>
__device__ __forceinline__ float trace_cone(const float4 (&sph)[RT_MAX_OBJECTS], v3 origin, v3 step, float threshold, float render_dist,
                                            uint32_t max_steps, Rep rep) {
    float len = 0.0f;
    uint32_t it = 0;
    if (ALG == 1) {  // every SDF every step; the radius is taken after the step
        while (len < render_dist) {
            if (max_steps && it++ >= max_steps) break;
            const v3 position = fma3(step, len, origin);
            float dist = sphere_sdf<REP>(position, sph[0], rep);
#pragma unroll
            for (int i = 1; i < N; i++) dist = fmin_(dist, sphere_sdf<REP>(position, sph[i], rep));
            len += dist;
            const float radius = (len + 1.0f) * threshold;
            if (dist <= radius) {
                len -= radius;
                break;
            }
        }
        return len;
    }
    float distances[N];
#pragma unroll
    for (int i = 0; i < N; i++) distances[i] = sphere_sdf<REP>(origin, sph[i], rep);  // :37-39
    if (ALG == 2) {  // one SDF per step: the object with the smallest cached bound, at the loop-top position
        uint32_t closest = 0;
        float nearest = 0.0f;
        float dc = distances[0];  // distances[closest], kept current (no dynamically indexed register array)
        while (len < render_dist) {
            if (max_steps && it++ >= max_steps) break;
            const v3 position = fma3(step, len, origin);
#pragma unroll
            for (int i = 0; i < N; i++) {
                distances[i] -= nearest;
                if ((uint32_t)i == closest) dc = distances[i];
                if (distances[i] < dc) {
                    closest = (uint32_t)i;
                    dc = distances[i];
                }
            }
            nearest = dc;
            len += nearest;
            float4 sc = sph[0];
#pragma unroll
            for (int i = 1; i < N; i++)
                if ((uint32_t)i == closest) sc = sph[i];
            dc = sphere_sdf<REP>(position, sc, rep);
#pragma unroll
            for (int i = 0; i < N; i++)
                if ((uint32_t)i == closest) distances[i] = dc;
            const float radius = (len + 1.0f) * threshold;
            if (dc <= radius) {
                len += dc - radius;
                break;
            }
        }
        return len;
    }
    float last = 0.0f;
    while (len < render_dist) {  // :44
        if (max_steps && it++ >= max_steps) break;
        const v3 position = fma3(step, len, origin);    // :45
        float dist = render_dist;                       // :49
        const float radius = (len + 1.0f) * threshold;  // :50
#pragma unroll
        for (int i = 0; i < N; i++) {  // :51-57
            // kept as a branch: evaluating all N distances and selecting (more ILP) was measured 1.5x
            // slower - the correctly rounded sqrt sequences are not free
            distances[i] -= last;
            if (distances[i] <= radius) distances[i] = sphere_sdf<REP>(position, sph[i], rep);
            dist = fmin_(dist, distances[i]);
        }
        last = fmax_(dist, 0.0f);  // :59
        len += last;               // :60
        if (dist <= radius) {      // :62-65
            len -= radius;
            break;
        }
    }
    return len;
}

// ---- shaders/compute.glsl:70-87 ---------------------------------------------------------------
// One invocation of compute.glsl:main for level pixel (gx, gy); `len0` is 1.0 at level 0 (:79) or the
// parent texel (:80-82).
template <int N, int ALG = 3, bool REP = false>
__device__ __forceinline__ float cone_pixel(const float4 (&sph)[RT_MAX_OBJECTS], const Camera cam, float isx, float isy, uint32_t gx, uint32_t gy, float len0,
                                            float render_dist, uint32_t max_steps, Rep rep = Rep{0.0f, 0.0f, 0.0f}) {
    // :71-72  (gid*2 + 1) * imageSize - 1, then * ratio   (jitter = 0 for the reference's sample)
    float nx = __builtin_fmaf((float)(gx * 2u + 1u), isx, -1.0f) + cam.jitter[0];
    float ny = __builtin_fmaf((float)(gy * 2u + 1u), isy, -1.0f) + cam.jitter[1];
    nx *= cam.ratio[0];
    ny *= cam.ratio[1];
    const float threshold = (1.4142135f * 8.0f) * isx;  // :75
    const v3 step = normalize(rotate_q(cam.rot[0], cam.rot[1], cam.rot[2], cam.rot[3], mk(nx, 1.0f, ny)));  // :77
    const v3 pos = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
    const float len = len0 + trace_cone<N, ALG, REP>(sph, fma3(step, len0, pos), step, threshold, render_dist, max_steps, rep);  // :84
    return fmax_(len, 0.0f);                                                                                    // :86
}

template <int N, int ALG, bool REP>
__global__ __launch_bounds__(256) void cone_level_kernel(const SphereSet S, const ConeLevelParams p,
                                                         const float* __restrict__ parent, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles_w = p.w >> 3;
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);  // one 8x8 tile per wave
    if (t >= tiles_w * (p.h >> 3)) return;                    // wave-uniform
    const uint32_t ty = t / tiles_w, tx = t - ty * tiles_w;
    if (p.partitioned && (8u << p.shift) <= RT_TILE) {
        // this level tile lies inside one framebuffer tile: trace it only on the owning rank
        if (!owns_pixel_tile(p.part, (tx * 8u) << p.shift, (ty * 8u) << p.shift, p.width, p.height)) return;
    }
    const uint32_t gx = tx * 8u + (lane & 7u), gy = ty * 8u + (lane >> 3);
    const uint32_t b = blockIdx.y;  // sample of the batch: its own jitter, its own level images
    Camera cam = p.cam;
    sample_jitter(p.sample0 + b, p.n_strata, p.width, p.height, &cam.jitter[0], &cam.jitter[1]);
    const float len0 = p.level > 0 ? parent[(size_t)b * p.parent_stride + (size_t)(gy >> 1) * p.parent_w + (gx >> 1)] : 1.0f;  // :79-82
    out[(size_t)b * p.level_stride + (size_t)gy * p.w + gx] = cone_pixel<N, ALG, REP>(S.s, cam, p.image_size[0], p.image_size[1], gx, gy, len0, p.render_dist, p.max_steps,
                                                                              Rep{p.repeat[0], p.repeat[1], p.repeat[2]});
}

// ---- shaders/fragment.glsl:89-121 -------------------------------------------------------------
template <int N, bool REP>
__device__ __forceinline__ float shadow_ray(const float4 (&sphere)[RT_MAX_OBJECTS], v3 origin, v3 step, float end,
                                            float ray_radius, uint32_t max_steps, Rep rep) {
    float distances[N];
#pragma unroll
    for (int i = 0; i < N; i++) distances[i] = sphere_sdf<REP>(origin, sphere[i], rep);  // :92-94

    float last = 0.0f, nearest = 1.0f;  // :96-97
    uint32_t it = 0;
    for (float len = 0.0f; len < end; len += last + ray_radius) {  // :99
        if (max_steps && it++ >= max_steps) break;
        const v3 position = fma3(step, len, origin);  // :100
        float dist = end;                             // :104
#pragma unroll
        for (int i = 0; i < N; i++) {  // :105-111
            distances[i] -= last;
            if (distances[i] <= nearest) distances[i] = sphere_sdf<REP>(position, sphere[i], rep);
            dist = fmin_(dist, distances[i]);
        }
        if (dist <= ray_radius) return 0.0f;  // :113-115
        last = fmax_(dist, 0.0f);             // :117
        nearest = fmin_(nearest, dist);       // :118
    }
    return nearest;  // :120
}

// Element i (wave-uniform) of an 8-entry kernarg array by compare/select over constant indices.  A
// dynamically indexed by-value kernel argument makes the compiler copy the whole struct to scratch
// (576 B per lane, +40 VGPRs); constant indices keep it in SGPRs.  For the same reason the inlined
// per-pixel functions take these structs BY VALUE: a reference to a by-value kernel argument pins
// the copy in memory, a value is split into scalars again after inlining.
__device__ __forceinline__ float4 pick8(const float4 (&a)[8], uint32_t i) {
    float4 v = a[0];
#pragma unroll
    for (uint32_t k = 1; k < 8; k++)
        if (i == k) v = a[k];
    return v;
}

// ---- shaders/fragment.glsl:144-186 ------------------------------------------------------------
// Nearest sphere, material and light loop for surface point `position` seen from `eye` along the unit direction `step`
// (camera ray: eye = push_constants.pos; mirror bounce: the previous hit).  Returns the normal and the material's
// specular coefficient for a following mirror bounce.
struct Surface {  // what a following mirror / transmission step needs of the surface just shaded
    v3 normal;
    float specular, diffuse;  // mat.specular / mat.diffuse: per-material scales of the mirror / transmission chains (the reference reads neither)
    float4 obj;               // the sphere: centre, radius
};

template <int N, bool REP>
__device__ __forceinline__ void shade_point(const ShadeSet& S, const ShadeParams& p, v3 position, v3 eye, v3 step, float& r, float& g, float& b, Surface& surf) {
    // :144-156 nearest sphere (strict '<', first wins ties); material index = object index
    const Rep rep{p.repeat[0], p.repeat[1], p.repeat[2]};
    float dist = sphere_sdf<REP>(position, S.sphere[0], rep);
    float4 obj = S.sphere[0];
    float4 mat = S.mat_color_ambient[0];
    float shine = S.mat_shine[0], specular = S.mat_specular[0], mat_diffuse = S.mat_diffuse[0];
#pragma unroll
    for (int i = 1; i < N; i++) {
        const float nd = sphere_sdf<REP>(position, S.sphere[i], rep);
        if (nd < dist) {
            dist = nd;
            obj = S.sphere[i];
            mat = S.mat_color_ambient[i];
            shine = S.mat_shine[i];
            specular = S.mat_specular[i];
            mat_diffuse = S.mat_diffuse[i];
        }
    }

    const float cam_dist = length(position - eye);                                                   // :162
    const float cam_fall = fmax_(p.cam_fall_off * __builtin_fmaf(cam_dist, cam_dist, 1.0f), 1.0f);   // :163
    const v3 normal = normalize(domain<REP>(position, rep) - mk(obj.x, obj.y, obj.z));                // :166
    const v3 cam_dir = -step;
    const float normal_fall = fmax_(dot(normal, cam_dir), 0.0f);  // :167

    r = g = b = 0.0f;
    for (uint32_t i = 0; i < S.light_count; i++) {  // :170-186
        const float4 lp = pick8(S.light_pos, i), lc = pick8(S.light_color, i);
        const v3 lpos = mk(lp.x, lp.y, lp.z);
        const v3 light_dir = normalize(lpos - position);   // :173
        const float light_dist = length(position - lpos);  // :174
        const float soft = fmin_(shadow_ray<N, REP>(S.sphere, position + light_dir, light_dir, light_dist, p.ray_radius, p.max_steps, rep), 1.0f);  // :176
        const float light_fall = fmax_((p.light_fall_off * light_dist) * light_dist, 1.0f);                                               // :178
        const float diffuse = fmax_(dot(normal, light_dir), 0.0f);                                                                        // :180
        // :181, :47-50  reflect(I,N) = I - 2*dot(N,I)*N with I = -lightDir
        const v3 inc = -light_dir;
        const float kk = 2.0f * dot(normal, inc);
        const v3 refl = mk(__builtin_fmaf(-kk, normal.x, inc.x), __builtin_fmaf(-kk, normal.y, inc.y), __builtin_fmaf(-kk, normal.z, inc.z));
        const float base = dot(refl, cam_dir);
        // pow(x<=0, y) is undefined in GLSL; defined as 0 here
        const float spec = base > 0.0f ? fmax_(diffuse * __builtin_powf(base, shine), 0.0f) : 0.0f;
        const float s = fmax_(diffuse + spec, 0.0f);  // :183
        const float dr = ((s * lc.x) / light_fall) * soft;
        const float dg = ((s * lc.y) / light_fall) * soft;
        const float db = ((s * lc.z) / light_fall) * soft;
        // :185  (ambient + direct) / camDistFallOff * normalFallOff * mat.color
        r = __builtin_fmaf(((mat.w + dr) / cam_fall) * normal_fall, mat.x, r);
        g = __builtin_fmaf(((mat.w + dg) / cam_fall) * normal_fall, mat.y, g);
        b = __builtin_fmaf(((mat.w + db) / cam_fall) * normal_fall, mat.z, b);
    }
    surf.normal = normal;
    surf.specular = specular;
    surf.diffuse = mat_diffuse;
    surf.obj = obj;
}

// ---- shaders/fragment.glsl:127-187 ------------------------------------------------------------
// One invocation of fragment.glsl:main for full-resolution pixel (px, py) whose depth is total_dist.
// Returns true for a hit pixel; rgb = 0 for a miss (:137-140).  REFL: mirror reflections (fragment.glsl:125 is a TODO in the
// reference; build-defined, specification at rt_config.reflections / oracle.h): r = reflect(step, normal), the ray starts one
// unit off the surface (the shadowRay idiom of :176) and is marched by compute.glsl's own loop with cone threshold RAY_RADIUS;
// a hit is shaded with the previous hit as the eye and added with weight prod(reflectivity * mat.specular).
// n_points / n_refl count the shaded reflection hits and the mirror rays of this lane.
template <int N, bool REP, bool REFL>
__device__ __forceinline__ bool shade_pixel(const ShadeSet S, const ShadeParams p, float jx, float jy, uint32_t px, uint32_t py, float total_dist, float& r,
                                            float& g, float& b, uint32_t& n_points, uint32_t& n_refl, uint32_t& n_trans) {
    r = g = b = 0.0f;
    if (!(total_dist < p.render_dist)) return false;  // :137-140
    // :129-133  gl_FragCoord.xy * 2 / cs.view - 1.0   (gl_FragCoord = pixel + 0.5)
    float nx = (((float)px + 0.5f) * 2.0f) / p.view[0] - 1.0f + jx;
    float ny = (((float)py + 0.5f) * 2.0f) / p.view[1] - 1.0f + jy;
    nx *= p.cam.ratio[0];
    ny *= p.cam.ratio[1];
    v3 step = normalize(rotate_q(p.cam.rot[0], p.cam.rot[1], p.cam.rot[2], p.cam.rot[3], mk(nx, 1.0f, ny)));
    const v3 pos = mk(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
    v3 position = fma3(step, total_dist, pos);  // :142
    Surface first;
    shade_point<N, REP>(S, p, position, pos, step, r, g, b, first);
    if (REFL) {
        const Rep rep{p.repeat[0], p.repeat[1], p.repeat[2]};
        const v3 first_position = position, first_step = step;
        float weight = 1.0f;
        Surface surf = first;
        for (uint32_t bounce = 0; bounce < p.reflections; bounce++) {
            weight *= p.reflectivity * surf.specular;
            const float kk = 2.0f * dot(surf.normal, step);
            const v3 rd = mk(__builtin_fmaf(-kk, surf.normal.x, step.x), __builtin_fmaf(-kk, surf.normal.y, step.y), __builtin_fmaf(-kk, surf.normal.z, step.z));
            n_refl++;
            const float len = 1.0f + trace_cone<N, 3, REP>(S.sphere, position + rd, rd, p.ray_radius, p.render_dist, p.max_steps, rep);
            if (!(len < p.render_dist)) break;
            const v3 hit = fma3(rd, fmax_(len, 0.0f), position);
            float rr, rg, rb;
            shade_point<N, REP>(S, p, hit, position, rd, rr, rg, rb, surf);
            n_points++;
            r = __builtin_fmaf(weight, rr, r);
            g = __builtin_fmaf(weight, rg, g);
            b = __builtin_fmaf(weight, rb, b);
            position = hit;
            step = rd;
        }
        // transmission (fragment.glsl:124 "TODO: transparency", :126 "TODO: refraction"; build-defined, specification at
        // rt_config.transmissions / oracle.h): enter the sphere (straight on, or bent by Snell's law), cross it to its far side in
        // closed form, leave it, march on like a mirror ray; weight prod(transparency * mat.diffuse).  Starts at the camera ray's hit.
        weight = 1.0f;
        surf = first;
        v3 P = first_position, I = first_step;
        const float idx = p.refraction_index;
        const bool bend = idx != 1.0f;
        for (uint32_t pass = 0; pass < p.transmissions; pass++) {
            weight *= p.transparency * surf.diffuse;
            v3 T = I;
            if (bend) {  // T = normalize(refract(I, n, 1 / index))
                const float eta = 1.0f / idx;
                const float ci = dot(surf.normal, I);
                const float k = __builtin_fmaf(-(eta * eta), __builtin_fmaf(-ci, ci, 1.0f), 1.0f);
                if (k < 0.0f) break;
                const float sn = __builtin_fmaf(eta, ci, sqrt_cr(k));
                T = normalize(mk(__builtin_fmaf(-sn, surf.normal.x, eta * I.x), __builtin_fmaf(-sn, surf.normal.y, eta * I.y), __builtin_fmaf(-sn, surf.normal.z, eta * I.z)));
            }
            const v3 oc = domain<REP>(P, rep) - mk(surf.obj.x, surf.obj.y, surf.obj.z);
            const float bq = dot(oc, T);
            const float cc = __builtin_fmaf(-surf.obj.w, surf.obj.w, dot(oc, oc));
            const float disc = __builtin_fmaf(bq, bq, -cc);
            float t = disc > 0.0f ? sqrt_cr(disc) - bq : 0.0f;
            if (!(t > 0.0f)) t = 0.0f;
            const v3 Q = fma3(T, t, P);
            v3 D = T;
            if (bend) {  // D = normalize(refract(T, -n2, index)), n2 = outward normal at the exit point
                const v3 n2 = normalize(fma3(T, t, oc));
                const float ce = dot(n2, T);
                const float k2 = __builtin_fmaf(-(idx * idx), __builtin_fmaf(-ce, ce, 1.0f), 1.0f);
                if (k2 < 0.0f) break;  // total internal reflection
                const float s2 = __builtin_fmaf(-idx, ce, sqrt_cr(k2));
                D = normalize(mk(__builtin_fmaf(s2, n2.x, idx * T.x), __builtin_fmaf(s2, n2.y, idx * T.y), __builtin_fmaf(s2, n2.z, idx * T.z)));
            }
            n_trans++;
            const float len = 1.0f + trace_cone<N, 3, REP>(S.sphere, Q + D, D, p.ray_radius, p.render_dist, p.max_steps, rep);
            if (!(len < p.render_dist)) break;
            const v3 hit = fma3(D, fmax_(len, 0.0f), Q);
            float rr, rg, rb;
            shade_point<N, REP>(S, p, hit, Q, D, rr, rg, rb, surf);
            n_points++;
            r = __builtin_fmaf(weight, rr, r);
            g = __builtin_fmaf(weight, rg, g);
            b = __builtin_fmaf(weight, rb, b);
            P = hit;
            I = D;
        }
    }
    return true;
}

// value of lane j of this lane's quad (v_mov_b32_dpp quad_perm:[j,j,j,j])
__device__ __forceinline__ float quad_bcast(float v, int j) {
    const int x = __float_as_int(v);
    switch (j) {
        case 0: return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x00, 0xf, 0xf, true));
        case 1: return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0x55, 0xf, 0xf, true));
        case 2: return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xaa, 0xf, 0xf, true));
        default: return __int_as_float(__builtin_amdgcn_mov_dpp(x, 0xff, 0xf, 0xf, true));
    }
}

// Grid: 16 workgroups per owned framebuffer tile; every wave shades one 8x8 block of the tile.
// SP (sample-parallel, batches of a multiple of four samples): 64 workgroups per tile, a wave shades a 4x4 block with FOUR lanes
// per pixel, one sample each per round of four; lane 0 of the quad then adds the four colours in sample order (the other lanes'
// values through DPP quad broadcasts) - the same sum, bit for bit, from four times as many waves of a quarter the length: a frame's
// last waves (the pixels with the longest shadow marches) hold the machine for a quarter of the time.
template <int N, bool REP, bool REFL, bool SP = false>
__global__ __launch_bounds__(256) void shade_kernel(const ShadeSet S, const ShadeParams p, const float* __restrict__ depth,
                                                    float* __restrict__ dst, uint64_t* __restrict__ counters) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = SP ? blockIdx.x >> 6 : blockIdx.x >> 4;                                           // owned-tile index
    const uint32_t sub = SP ? ((blockIdx.x & 63u) << 2) + (threadIdx.x >> 6) : ((blockIdx.x & 15u) << 2) + (threadIdx.x >> 6);  // 4x4 / 8x8 block inside the tile
    const uint32_t tile = p.part.rank + k * p.part.n_ranks;
    const uint32_t tile_y = tile / p.part.tiles_x, tile_x = tile - tile_y * p.part.tiles_x;
    const uint32_t pix = lane >> 2, smp = lane & 3u;  // SP: pixel of the 4x4 block, sample of the round
    const uint32_t lx = SP ? ((sub & 15u) << 2) + (pix & 3u) : ((sub & 7u) << 3) + (lane & 7u);
    const uint32_t ly = SP ? ((sub >> 4) << 2) + (pix >> 2) : ((sub >> 3) << 3) + (lane >> 3);
    const uint32_t px = tile_x * RT_TILE + lx, py = tile_y * RT_TILE + ly;
    const bool inside = px < p.width && py < p.height;

    // the samples of the batch in index order, continuing the running sum of earlier batches:
    // ((s0 + s1) + s2) + ... exactly what one launch per sample gives
    const size_t idx = p.tile_major ? ((size_t)k * (RT_TILE * RT_TILE) + (size_t)ly * RT_TILE + lx) : ((size_t)py * p.width + px);
    float r = 0.0f, g = 0.0f, b = 0.0f;
    bool have_sum = false;
    if ((p.mode & 1u) && inside) {
        r = dst[idx * 3];
        g = dst[idx * 3 + 1];
        b = dst[idx * 3 + 2];
        have_sum = true;
    }
    uint32_t n_hits = 0, n_points = 0, n_refl = 0, n_trans = 0;
    for (uint32_t sb0 = 0; sb0 < p.n_batch; sb0 += SP ? 4u : 1u) {
        const uint32_t sb = SP ? sb0 + smp : sb0;
        float jx, jy, sr = 0.0f, sg = 0.0f, sbl = 0.0f;
        sample_jitter(p.sample0 + sb, p.n_strata, p.width, p.height, &jx, &jy);
        const bool hit = inside && shade_pixel<N, REP, REFL>(S, p, jx, jy, px, py, depth[(size_t)sb * p.depth_stride + (size_t)py * p.depth_w + px], sr, sg, sbl, n_points, n_refl, n_trans);  // :135
#pragma unroll
        for (int j = 0; j < (SP ? 4 : 1); j++) {  // the round's samples in index order (SP: sample j sits in lane j of the quad)
            const float cr = SP ? quad_bcast(sr, j) : sr, cg = SP ? quad_bcast(sg, j) : sg, cb = SP ? quad_bcast(sbl, j) : sbl;
            if (have_sum) {
                r += cr;
                g += cg;
                b += cb;
            } else {
                r = cr;
                g = cg;
                b = cb;
                have_sum = true;
            }
        }
        n_hits += (uint32_t)__popcll(__ballot(hit));
    }
    // hit-pixel statistics: one atomic per wave, spread over 1024 slots (a single hot word serves only
    // ~90 atomics/us chip-wide and made this kernel atomic-bound); the host sums the slots
    if (lane == 0 && n_hits) atomicAdd((unsigned long long*)&counters[blockIdx.x & 1023u], (unsigned long long)n_hits);
    if (REFL) {  // secondary hits shaded / mirror rays marched / transmitted rays marched: slots 1024.., 2048.. and 3072..
        unsigned long long pts = n_points, rays = n_refl, trs = n_trans;
        for (int off = 32; off > 0; off >>= 1) {
            pts += __shfl_down(pts, off);
            rays += __shfl_down(rays, off);
            trs += __shfl_down(trs, off);
        }
        if (lane == 0 && (rays | trs)) {
            atomicAdd((unsigned long long*)&counters[1024u + (blockIdx.x & 1023u)], pts);
            atomicAdd((unsigned long long*)&counters[2048u + (blockIdx.x & 1023u)], rays);
            atomicAdd((unsigned long long*)&counters[3072u + (blockIdx.x & 1023u)], trs);
        }
    }

    if (inside && (!SP || smp == 0u)) {
        if (p.mode & 2u) {
            r = r / p.spp;
            g = g / p.spp;
            b = b / p.spp;
        }
        dst[idx * 3] = r;
        dst[idx * 3 + 1] = g;
        dst[idx * 3 + 2] = b;
    }
}

// ---- self test of the arithmetic contract's sqrt ------------------------------------------------
// every one of the 2^32 fp32 bit patterns: sqrt_cr must return the bits of the compiler's correctly rounded sqrt
__global__ __launch_bounds__(256) void selftest_sqrt_kernel(unsigned long long* __restrict__ mismatches) {
    unsigned long long bad = 0;
    for (uint32_t k = 0; k < 256u; k++) {
        const uint32_t bits = (blockIdx.x * 256u + threadIdx.x) * 256u + k;
        const float x = __uint_as_float(bits);
        const float a = sqrt_cr(x), b = __builtin_sqrtf(x);
        const bool same = (a != a && b != b) || __float_as_uint(a) == __float_as_uint(b);
        bad += same ? 0u : 1u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---- fused pyramid kernel -------------------------------------------------------------------------
// The reference records one dispatch per pyramid level (src/main.rs:300-316) because Vulkan needs a
// barrier between levels.  A pixel's chain of ancestors is private to its 32x32 neighbourhood, so on
// gfx950 the whole pyramid CAN be one launch: a 256-thread workgroup owns a 32x32 block of
// full-resolution pixels and walks the pyramid top-down for just that block (1,1,..,2x2,4x4,8x8,
// 16x16,32x32 texels; coarse texels shared with neighbouring blocks are recomputed, they are a
// handful), keeping the parent level in LDS.  Seven kernel boundaries disappear.
// MEASURED (1920x1080, 8 spheres): 0.35 ms against 0.28 ms for the eight per-level launches - the
// frame's critical path is the sum of the per-level march latencies either way, and the per-level
// launches spread each level over the whole chip while a workgroup spends its coarse levels with
// one lane busy.  So this schedule is OFF by default (rt_config.fuse_levels) and kept as the
// tested alternative.  (Fusing the shading pass as well needs more scalar registers than a wave has
// - scene, lights, materials and nine levels of parameters are all wave-uniform - and spilled.)
// Level images are written only where a texel has a descendant inside the frame; the others are
// never read by anything and stay 0.
constexpr uint32_t kFusedTile = 32;
__device__ __forceinline__ void tile_coords(uint32_t idx, uint32_t ext, uint32_t& lx, uint32_t& ly) {
    if (ext >= 8u) {  // 8x8 blocks: one wave iteration = one compact block (similar march lengths)
        const uint32_t blk = idx >> 6, in = idx & 63u, bw = ext >> 3;
        lx = (blk % bw) * 8u + (in & 7u);
        ly = (blk / bw) * 8u + (in >> 3);
    } else {
        lx = idx % ext;
        ly = idx / ext;
    }
}

template <int N>
__global__ __launch_bounds__(256) void pyramid_tile_kernel(const SphereSet S, const PyramidParams fp) {
    __shared__ float lds[2][kFusedTile * kFusedTile];  // ping-pong: parent level / current level
    const uint32_t k = blockIdx.x >> 2, quad = blockIdx.x & 3u;
    const uint32_t tile = fp.part.rank + k * fp.part.n_ranks;
    const uint32_t tile_y = tile / fp.part.tiles_x, tile_x = tile - tile_y * fp.part.tiles_x;
    const uint32_t X0 = tile_x * RT_TILE + (quad & 1u) * kFusedTile, Y0 = tile_y * RT_TILE + (quad >> 1) * kFusedTile;
    if (X0 >= fp.width || Y0 >= fp.height) return;  // whole block outside the frame (workgroup-uniform)

    const uint32_t last = fp.count - 1u;
    for (uint32_t lvl = 0; lvl <= last; lvl++) {
        const uint32_t shift = last - lvl;
        const uint32_t ext = (kFusedTile >> shift) ? (kFusedTile >> shift) : 1u, pext = (ext >> 1) ? (ext >> 1) : 1u;
        const uint32_t ox = X0 >> shift, oy = Y0 >> shift;
        float* cur = lds[lvl & 1u];
        const float* par = lds[(lvl & 1u) ^ 1u];
        // per-level parameters by compare/select over constant indices (see pick8)
        float isx = fp.image_size[0][0], isy = fp.image_size[0][1];
        float* img = fp.level[0];
        uint32_t pitch = fp.level_w[0];
#pragma unroll
        for (uint32_t q = 1; q < RT_MAX_LEVELS; q++)
            if (lvl == q) {
                isx = fp.image_size[q][0];
                isy = fp.image_size[q][1];
                img = fp.level[q];
                pitch = fp.level_w[q];
            }
        for (uint32_t idx = threadIdx.x; idx < ext * ext; idx += 256u) {
            uint32_t lx, ly;
            tile_coords(idx, ext, lx, ly);
            const uint32_t gx = ox + lx, gy = oy + ly;
            float v = 0.0f;
            if ((gx << shift) < fp.width && (gy << shift) < fp.height) {  // has a descendant inside the frame
                const float len0 = lvl ? par[(ly >> 1) * pext + (lx >> 1)] : 1.0f;
                v = cone_pixel<N>(S.s, fp.cam, isx, isy, gx, gy, len0, fp.render_dist, fp.max_steps);
                img[(size_t)gy * pitch + gx] = v;
            }
            cur[ly * ext + lx] = v;
        }
        __syncthreads();
    }
}

// Scatter rank-major, tile-major gathered buffers into a full frame (one thread per pixel).
__global__ __launch_bounds__(256) void detile_kernel(const float* __restrict__ tiles, uint32_t n_ranks, uint32_t tiles_per_rank,
                                                     uint32_t tiles_x, uint32_t tiles_y, uint32_t width, uint32_t height,
                                                     float* __restrict__ rgb) {
    const uint32_t px = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t py = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (px >= width || py >= height) return;
    const uint32_t tile = (py / RT_TILE) * tiles_x + (px / RT_TILE);
    const uint32_t rank = tile % n_ranks, k = tile / n_ranks;
    const size_t src = ((size_t)rank * tiles_per_rank + k) * (RT_TILE * RT_TILE) + (size_t)(py % RT_TILE) * RT_TILE + (px % RT_TILE);
    const size_t dst = (size_t)py * width + px;
    rgb[dst * 3 + 0] = tiles[src * 3 + 0];
    rgb[dst * 3 + 1] = tiles[src * 3 + 1];
    rgb[dst * 3 + 2] = tiles[src * 3 + 2];
    (void)tiles_y;
}

// linear f32 -> UNORM8 (clamp, *255, round-to-nearest-even), alpha = 255
__global__ __launch_bounds__(256) void to_rgba8_kernel(const float* __restrict__ rgb, uint32_t* __restrict__ rgba, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t packed = 0xff000000u;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = rgb[i * 3 + c];
        v = v > 0.0f ? v : 0.0f;
        v = v < 1.0f ? v : 1.0f;
        packed |= (uint32_t)__builtin_rintf(v * 255.0f) << (8 * c);
    }
    rgba[i] = packed;
}

// ---- launchers ----------------------------------------------------------------------------------
template <int N>
static void cone_launch_n(hipStream_t st, dim3 grid, const SphereSet& S, const ConeLevelParams& p, const float* parent, float* out) {
    const bool rep = p.repeat[0] > 0.0f || p.repeat[1] > 0.0f || p.repeat[2] > 0.0f;
    // the reference as shipped (algorithm 3, no repetition) first; the sketched variants behind it
    if (p.alg == 1) {
        if (rep) hipLaunchKernelGGL((cone_level_kernel<N, 1, true>), grid, dim3(256), 0, st, S, p, parent, out);
        else hipLaunchKernelGGL((cone_level_kernel<N, 1, false>), grid, dim3(256), 0, st, S, p, parent, out);
    } else if (p.alg == 2) {
        if (rep) hipLaunchKernelGGL((cone_level_kernel<N, 2, true>), grid, dim3(256), 0, st, S, p, parent, out);
        else hipLaunchKernelGGL((cone_level_kernel<N, 2, false>), grid, dim3(256), 0, st, S, p, parent, out);
    } else if (rep) {
        hipLaunchKernelGGL((cone_level_kernel<N, 3, true>), grid, dim3(256), 0, st, S, p, parent, out);
    } else {
        hipLaunchKernelGGL((cone_level_kernel<N, 3, false>), grid, dim3(256), 0, st, S, p, parent, out);
    }
}
template <int N>
static void shade_launch_n(hipStream_t st, dim3 grid, const ShadeSet& S, const ShadeParams& p, const float* depth, float* dst,
                           uint64_t* counters) {
    const bool rep = p.repeat[0] > 0.0f || p.repeat[1] > 0.0f || p.repeat[2] > 0.0f;
    // the reference as shipped (no reflections, no transmission, no repetition) first; the sketched variants behind it
    if (p.reflections == 0 && p.transmissions == 0) {
        if (rep) hipLaunchKernelGGL((shade_kernel<N, true, false>), grid, dim3(256), 0, st, S, p, depth, dst, counters);
        else if ((p.n_batch & 3u) == 0u) hipLaunchKernelGGL((shade_kernel<N, false, false, true>), dim3(grid.x * 4u), dim3(256), 0, st, S, p, depth, dst, counters);  // four lanes per pixel
        else hipLaunchKernelGGL((shade_kernel<N, false, false>), grid, dim3(256), 0, st, S, p, depth, dst, counters);
    } else {
        if (rep) hipLaunchKernelGGL((shade_kernel<N, true, true>), grid, dim3(256), 0, st, S, p, depth, dst, counters);
        else hipLaunchKernelGGL((shade_kernel<N, false, true>), grid, dim3(256), 0, st, S, p, depth, dst, counters);
    }
}

int launch_cone_level(Ctx* c, const SphereSet& S, uint32_t n_obj, const ConeLevelParams& p, const float* parent, float* out, uint32_t batch) {
    if (n_obj < 1 || n_obj > RT_MAX_OBJECTS) return c->fail(RT_ERR_INVALID, "objCount %u out of [1,8]", n_obj);
    if ((p.w & 7u) || (p.h & 7u) || p.w == 0 || p.h == 0) return c->fail(RT_ERR_INVALID, "level dims %ux%u not multiples of 8", p.w, p.h);
    if (batch < 1 || batch > 65535u || p.n_strata < 1) return c->fail(RT_ERR_INVALID, "sample batch %u / strata %u", batch, p.n_strata);
    if (p.alg != 1 && p.alg != 2 && p.alg != 3) return c->fail(RT_ERR_INVALID, "march algorithm %u", p.alg);
    if (p.level > 0 && (parent == nullptr || p.parent_w * 2u < p.w)) return c->fail(RT_ERR_INVALID, "level %u: bad parent image", p.level);
    const uint32_t tiles = (p.w >> 3) * (p.h >> 3);
    const dim3 grid((tiles + 3u) / 4u, batch);
    switch (n_obj) {
        case 1: cone_launch_n<1>(c->stream, grid, S, p, parent, out); break;
        case 2: cone_launch_n<2>(c->stream, grid, S, p, parent, out); break;
        case 3: cone_launch_n<3>(c->stream, grid, S, p, parent, out); break;
        case 4: cone_launch_n<4>(c->stream, grid, S, p, parent, out); break;
        case 5: cone_launch_n<5>(c->stream, grid, S, p, parent, out); break;
        case 6: cone_launch_n<6>(c->stream, grid, S, p, parent, out); break;
        case 7: cone_launch_n<7>(c->stream, grid, S, p, parent, out); break;
        default: cone_launch_n<8>(c->stream, grid, S, p, parent, out); break;
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_shade(Ctx* c, const ShadeSet& S, uint32_t n_obj, const ShadeParams& p, const float* depth, float* dst, uint64_t* counters) {
    if (n_obj < 1 || n_obj > RT_MAX_OBJECTS) return c->fail(RT_ERR_INVALID, "objCount %u out of [1,8]", n_obj);
    if (p.depth_w < p.width) return c->fail(RT_ERR_INVALID, "depth pitch %u < width %u", p.depth_w, p.width);
    if (p.n_batch < 1 || p.n_strata < 1) return c->fail(RT_ERR_INVALID, "sample batch %u / strata %u", p.n_batch, p.n_strata);
    const uint32_t total = p.part.tiles_x * p.part.tiles_y;
    if (p.part.n_ranks == 0 || p.part.rank >= p.part.n_ranks) return c->fail(RT_ERR_INVALID, "bad partition");
    const uint32_t owned = total > p.part.rank ? (total - p.part.rank + p.part.n_ranks - 1u) / p.part.n_ranks : 0u;
    if (owned == 0) return RT_OK;
    const dim3 grid(owned * 16u);
    switch (n_obj) {
        case 1: shade_launch_n<1>(c->stream, grid, S, p, depth, dst, counters); break;
        case 2: shade_launch_n<2>(c->stream, grid, S, p, depth, dst, counters); break;
        case 3: shade_launch_n<3>(c->stream, grid, S, p, depth, dst, counters); break;
        case 4: shade_launch_n<4>(c->stream, grid, S, p, depth, dst, counters); break;
        case 5: shade_launch_n<5>(c->stream, grid, S, p, depth, dst, counters); break;
        case 6: shade_launch_n<6>(c->stream, grid, S, p, depth, dst, counters); break;
        case 7: shade_launch_n<7>(c->stream, grid, S, p, depth, dst, counters); break;
        default: shade_launch_n<8>(c->stream, grid, S, p, depth, dst, counters); break;
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

template <int N>
static void pyramid_launch_n(hipStream_t st, dim3 grid, const SphereSet& S, const PyramidParams& fp) {
    hipLaunchKernelGGL(pyramid_tile_kernel<N>, grid, dim3(256), 0, st, S, fp);
}

int launch_pyramid_fused(Ctx* c, const SphereSet& S, uint32_t n_obj, const PyramidParams& fp) {
    if (n_obj < 1 || n_obj > RT_MAX_OBJECTS) return c->fail(RT_ERR_INVALID, "objCount %u out of [1,8]", n_obj);
    const Partition& part = fp.part;
    if (part.n_ranks == 0 || part.rank >= part.n_ranks || fp.count < 1 || fp.count > RT_MAX_LEVELS) return c->fail(RT_ERR_INVALID, "bad frame parameters");
    for (uint32_t i = 0; i < fp.count; i++)
        if (!fp.level[i] || fp.level_w[i] == 0) return c->fail(RT_ERR_INVALID, "level %u has no image", i);
    const uint32_t total = part.tiles_x * part.tiles_y;
    const uint32_t owned = total > part.rank ? (total - part.rank + part.n_ranks - 1u) / part.n_ranks : 0u;
    if (owned == 0) return RT_OK;
    const dim3 grid(owned * 4u);  // four 32x32 blocks per 64x64 framebuffer tile
    switch (n_obj) {
        case 1: pyramid_launch_n<1>(c->stream, grid, S, fp); break;
        case 2: pyramid_launch_n<2>(c->stream, grid, S, fp); break;
        case 3: pyramid_launch_n<3>(c->stream, grid, S, fp); break;
        case 4: pyramid_launch_n<4>(c->stream, grid, S, fp); break;
        case 5: pyramid_launch_n<5>(c->stream, grid, S, fp); break;
        case 6: pyramid_launch_n<6>(c->stream, grid, S, fp); break;
        case 7: pyramid_launch_n<7>(c->stream, grid, S, fp); break;
        default: pyramid_launch_n<8>(c->stream, grid, S, fp); break;
    }
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_selftest_sqrt(Ctx* c, unsigned long long* mismatches_dev) {
    hipLaunchKernelGGL(selftest_sqrt_kernel, dim3(1u << 16), dim3(256), 0, c->stream, mismatches_dev);  // 65536 x 256 x 256 = 2^32 inputs
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_detile(Ctx* c, const float* tiles, uint32_t n_ranks, uint32_t tiles_per_rank, float* rgb) {
    const dim3 grid((c->width + 63u) / 64u, (c->height + 3u) / 4u);
    hipLaunchKernelGGL(detile_kernel, grid, dim3(256), 0, c->stream, tiles, n_ranks, tiles_per_rank, c->part.tiles_x, c->part.tiles_y,
                       c->width, c->height, rgb);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

int launch_to_rgba8(Ctx* c, const float* rgb, uint8_t* rgba, uint64_t n_pixels) {
    const dim3 grid((unsigned)((n_pixels + 255u) / 256u));
    hipLaunchKernelGGL(to_rgba8_kernel, grid, dim3(256), 0, c->stream, rgb, (uint32_t*)rgba, n_pixels);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

}  // namespace rt
