/*
 * oracle.h — CPU oracle for the per-pixel ray/scene intersection + shading hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and there
 * only as the checker / the timed CPU baseline.  The product library (librt_amd.so) never
 * links, loads or calls anything declared here.
 *
 * PARITY STATUS: "parity unpinned by the reference".  The reference
 * (IvoteSligte/raytracing_engine) ships no tests, golden vectors or fixtures, cannot be built
 * in this environment (no Rust, no GLSL compiler, no Vulkan), and is not
 * Python.  Oracle A below is a line-by-line restatement of the reference GLSL + host launch
 * logic and is pinned by analytic known-answer tests (tests/test_oracle_a.py) and by an
 * independent brute-force cross-check (shaders/tracing_algorithms.txt:2-13 "algorithm 1").
 * Oracle B (triangles + BVH + path tracing) has no reference counterpart at all.
 *
 * ARITHMETIC CONTRACT (identical in oracle and HIP kernels):
 *   fp32 only, round-to-nearest-even, denormals kept, no implicit contraction
 *   (-ffp-contract=off); every fused multiply-add is written explicitly as fmaf();
 *   sqrtf and '/' are the correctly rounded IEEE operations;
 *   dot(a,b)     = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x))
 *   length(a)    = sqrtf(dot(a,a));   distance(a,b) = length(a-b)
 *   normalize(a) = a * (1.0f / length(a))
 *   cross(a,b).x = fmaf(a.y,b.z, -(a.z*b.y))   (cyclic)
 *   min/max      = fminf/fmaxf (NaN inputs are outside the contract)
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_MATERIALS 8u /* shaders/utilities.glsl:2 */
#define ORA_MAX_OBJECTS 8u   /* shaders/utilities.glsl:3 */
#define ORA_MAX_LIGHTS 8u    /* shaders/utilities.glsl:4 */
#define ORA_MAX_LEVELS 9u    /* shaders/compute.glsl:14-15, src/main.rs:359 */

/* std140 images of shaders/utilities.glsl:8-24 as the vulkano shader! macro lays them out
 * (src/main.rs:524-591 shows the _dummy padding fields). */
typedef struct { float color[3]; float diffuse; float specular; float shine; float ambient; uint32_t pad; } ora_material; /* 32 B */
typedef struct { float pos[3]; float size; } ora_object;                                                          /* 16 B */
typedef struct { float pos[3]; uint32_t pad0; float color[3]; uint32_t pad1; } ora_light;                          /* 32 B */
/* MutableData, shaders/compute.glsl:17-24 == shaders/fragment.glsl:17-24; 656 B */
typedef struct {
    uint32_t matCount, objCount, lightCount, pad;
    ora_material mats[8];
    ora_object objs[8];
    ora_light lights[8];
} ora_scene;

typedef struct {
    float render_dist;    /* RENDER_DIST = 1000, src/main.rs:362, spec-const compute.glsl:32 */
    float cam_fall_off;   /* CAM_FALL_OFF = 0.01, fragment.glsl:35 */
    float light_fall_off; /* LIGHT_FALL_OFF = 0.01, fragment.glsl:36 */
    float ray_radius;     /* RAY_RADIUS = 0.01, fragment.glsl:37 */
    uint32_t max_steps;   /* safety cap on either march loop (build-side; 0 = unlimited) */
    /* SDF feature growth the author sketched; 0 / {0,0,0} = the reference as shipped */
    uint32_t march_algorithm; /* cone march loop body: 0 or 3 = compute.glsl:46-65 ("algorithm 3"),
                                 1, 2 = shaders/tracing_algorithms.txt:2-13 / :16-37 */
    float repeat[3];          /* > 0: domain repetition period on that axis, utilities.glsl:31-34 */
    /* mirror reflections (fragment.glsl:125 "TODO: reflection"; build-defined): 0 = off = the reference
     * as shipped.  After shading a hit point P seen along the unit direction `step`: r = reflect(step, normal); the ray
     * starts one unit off the surface (the reference's shadowRay idiom, fragment.glsl:176) and is marched by compute.glsl's
     * own loop (:44-66) with cone threshold RAY_RADIUS: len = 1 + traceCone(P + r, r, RAY_RADIUS); a hit (len < RENDER_DIST)
     * at P' = P + r * max(len, 0) is shaded by fragment.glsl:144-186 with P as the eye, and added with weight
     * prod(reflectivity * mat.specular) over the surfaces passed; at most `reflections` bounces. */
    uint32_t reflections;
    float reflectivity;
    /* transmission (fragment.glsl:124 "TODO: transparency", :126 "TODO: refraction"; build-defined): 0 = off =
     * the reference as shipped.  After shading a hit point P on sphere S seen along the unit direction I (outward normal n):
     * the ray enters the sphere - straight on for refraction_index == 1 (transparency), bent by Snell's law otherwise
     * (T = normalize(refract(I, n, 1 / index)), GLSL's refract) - and crosses it to the far side in closed form:
     * oc = domain(P) - S.pos, b = oc.T, disc = b*b - (oc.oc - S.size^2), t = disc > 0 ? sqrt(disc) - b : 0, clamped at 0,
     * Q = P + T t.  There it leaves along D = T (transparency) or D = normalize(refract(T, -n2, index)) with
     * n2 = normalize(oc + T t); total internal reflection (k < 0) ends the chain.  From Q the scene is marched like a mirror
     * ray: len = 1 + traceCone(Q + D, D, RAY_RADIUS) (compute.glsl:44-66); a hit (len < RENDER_DIST) at Q + D * max(len, 0) is
     * shaded by fragment.glsl:144-186 with Q as the eye and added with weight prod(transparency * mat.diffuse) over the
     * surfaces passed (mat.diffuse is a field the reference uploads and never reads; 1 in its scene); at most `transmissions`
     * spheres are crossed.  The chain starts at the camera ray's hit and is independent of the mirror chain. */
    uint32_t transmissions;
    float transparency;      /* 0..1, default 0.5 */
    float refraction_index;  /* >= 1, default 1 (straight through) */
} ora_config;

typedef struct {
    uint64_t cone_threads;  /* compute.glsl invocations over all levels */
    uint64_t cone_steps;    /* iterations of compute.glsl:44-66 */
    uint64_t cone_sdf;      /* sphereSDF evaluations inside traceCone (init + refresh) */
    uint64_t hit_pixels;    /* full-res pixels with depth < RENDER_DIST */
    uint64_t shadow_rays;   /* shadowRay calls (= hit_pixels * lightCount) */
    uint64_t shadow_steps;  /* iterations of fragment.glsl:99-119 */
    uint64_t shadow_sdf;    /* sphereSDF evaluations inside shadowRay */
    uint64_t reflection_rays; /* mirror rays marched (reflections > 0) */
    uint64_t transmission_rays; /* rays marched behind a crossed sphere (transmissions > 0) */
} ora_counters;

void ora_default_config(ora_config* cfg);
/* default scene of src/main.rs:524-591 (4 materials, 4 spheres, 2 lights) */
void ora_default_scene(ora_scene* s);
/* level count (src/main.rs:639, floor form, capped at 9) and level dims (src/main.rs:203-234) */
uint32_t ora_level_count(uint32_t width);
void ora_level_dims(uint32_t width, uint32_t height, uint32_t count, uint32_t level, uint32_t* w, uint32_t* h);
/* camera quaternion of src/main.rs:402-404: Rz(-yaw) * Rx(pitch), array order x,y,z,w */
void ora_camera_quat(float yaw, float pitch, float out[4]);

/* One frame of path A = the launch schedule of src/main.rs:300-316 + the draw of :335.
 *   levels[i] (optional, may be NULL / contain NULLs): receives level i (w_i*h_i floats)
 *   rgb (optional): width*height*3 floats, row-major, origin = gl_FragCoord origin
 *   jitter: NDC offset added to normCoord before *ratio (build-side "spp" extension,
 *           {0,0} reproduces the reference exactly); may be NULL
 *   threads: OpenMP threads (<=0: all)
 * returns 0, or <0 on invalid arguments */
int ora_render_a(const ora_scene* scene, const ora_config* cfg, uint32_t width, uint32_t height,
                 const float ratio[2], const float rot[4], const float pos[3], const float jitter[2],
                 float* const* levels, float* rgb, ora_counters* counters, int threads);

/* brute-force marcher ("algorithm 1", shaders/tracing_algorithms.txt:2-13) for one ray:
 * independent cross-check of traceCone; returns the accumulated length */
float ora_trace_bruteforce(const ora_scene* scene, const ora_config* cfg, const float origin[3],
                           const float dir[3], float threshold);
/* single-ray entry points used by the known-answer tests */
float ora_trace_cone(const ora_scene* scene, const ora_config* cfg, const float origin[3],
                     const float dir[3], float threshold);
float ora_shadow_ray(const ora_scene* scene, const ora_config* cfg, const float origin[3],
                     const float dir[3], float end);
void ora_rotate(const float q[4], const float v[3], float out[3]);
/* linear float -> UNORM8 as a *_UNORM swapchain stores it (src/main.rs:471-486): clamp, *255, rint */
void ora_to_unorm8(const float* rgb, uint64_t n_pixels, uint8_t* rgba);

/* =========================================================================================
 * Oracle B — build-defined extension (BASELINE.json configs[2..4]): triangles + BVH + path
 * tracing.  NO REFERENCE COUNTERPART (SURVEY.md §0, §8a last row): the reference has no
 * triangles, BVH, RNG, spp or bounces, so parity for this path is "HIP kernels vs this
 * oracle" only — "parity unpinned by the reference".  The specification both sides implement
 * is DESIGN.md §6.
 * ========================================================================================= */
typedef struct {
    uint32_t n_tris;
    const float* verts;    /* n_tris * 9: v0, v1, v2 */
    const float* albedo;   /* n_tris * 3 */
    const float* emission; /* n_tris * 3; a triangle with any component > 0 is a light */
} orb_mesh;

typedef struct {
    uint32_t width, height;
    uint32_t spp;      /* samples per pixel */
    uint32_t bounces;  /* indirect bounces after the camera ray */
    uint32_t seed;
    float ratio[2];
    float rot[4];
    float pos[3];
    float sky[3];      /* radiance of rays that leave the scene */
    float ray_eps;     /* origin offset along the shading normal */
} orb_params;

typedef struct {
    uint64_t camera_rays, bounce_rays, shadow_rays; /* rays handed to BVH traversal */
    uint64_t nodes_visited;  /* BVH2 child-pair nodes fetched, all rays */
    uint64_t tris_tested;    /* ray/triangle tests, all rays */
    uint64_t n_nodes;        /* size of the oracle's own BVH */
} orb_counters;

typedef struct orb_scene orb_scene; /* mesh + the oracle's own BVH */
orb_scene* orb_scene_create(const orb_mesh* mesh);
void orb_scene_destroy(orb_scene* s);

/* rgb: width*height*3, row 0 = py 0.  use_bvh = 0: brute force over all triangles (small
 * scenes only; independent cross-check of the BVH path).  threads <= 0: all. */
int orb_render(const orb_scene* s, const orb_params* p, float* rgb, orb_counters* ct, int use_bvh, int threads);
/* rows [row0,row1) only; rgb: (row1-row0)*width*3 */
int orb_render_rows(const orb_scene* s, const orb_params* p, uint32_t row0, uint32_t row1, float* rgb, orb_counters* ct, int use_bvh, int threads);

/* single-ray hooks for the known-answer tests: closest hit (returns triangle index or -1,
 * *t_out = distance) and occlusion of the open segment (origin, origin + dir) */
int32_t orb_closest_hit(const orb_scene* s, const float origin[3], const float dir[3], float* t_out, int use_bvh);
int orb_occluded(const orb_scene* s, const float origin[3], const float dir[3], int use_bvh);
/* the spec's RNG and direction sampler, exposed for KATs */
float orb_rand(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim, uint32_t seed);
void orb_cosine_dir(const float n[3], float u1, float u2, float out[3]);
void orb_sincos_2pi(float u, float* s, float* c);

#ifdef __cplusplus
}
#endif
#endif
