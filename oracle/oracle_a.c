/*
 * oracle_a.c — Oracle A: CPU restatement of the reference's sphere-SDF cone marcher + shading.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  "parity unpinned by the reference": the reference
 * holds no golden vectors; this file follows the GLSL/Rust text cited at each function.
 *
 * All file:line citations are relative to the reference repository root.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -mfma -fopenmp (oracle/Makefile).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

/* ---- arithmetic contract (oracle.h header comment) ------------------------------------ */
static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* a*s + b, one fma per component */
static inline v3 v3_fma(v3 a, float s, v3 b) { return v3_make(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { float s = 1.0f / v3_length(a); return v3_scale(a, s); }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 v3_load(const float* p) { return v3_make(p[0], p[1], p[2]); }

/* shaders/utilities.glsl:26-29   t = cross(q.xyz, v) + q.w*v;  return v + 2*cross(q.xyz, t) */
static inline v3 rotate_q(const float q[4], v3 v) {
    v3 qv = v3_make(q[0], q[1], q[2]);
    v3 c = v3_cross(qv, v);
    v3 t = v3_make(fmaf(q[3], v.x, c.x), fmaf(q[3], v.y, c.y), fmaf(q[3], v.z, c.z));
    v3 c2 = v3_cross(qv, t);
    return v3_make(fmaf(2.0f, c2.x, v.x), fmaf(2.0f, c2.y, v.y), fmaf(2.0f, c2.z, v.z));
}

/* shaders/utilities.glsl:31-34   repeat(p, r) = mod(p + 0.5*r, r) - 0.5*r with GLSL's
 * mod(x, y) = x - y*floor(x/y); applied per axis where the period is > 0 (the reference defines the
 * function and never calls it: which positions it applies to is build-defined) */
static inline float repeat1(float p, float r) {
    if (!(r > 0.0f)) return p;
    const float h = 0.5f * r, a = p + h;
    return (a - r * floorf(a / r)) - h;
}
static inline v3 domain(v3 p, const ora_config* cfg) {
    return v3_make(repeat1(p.x, cfg->repeat[0]), repeat1(p.y, cfg->repeat[1]), repeat1(p.z, cfg->repeat[2]));
}

/* shaders/utilities.glsl:36-38   distance(p, s.pos) - s.size   (p in the repeated domain) */
static inline float sphere_sdf(v3 p, const ora_object* s, const ora_config* cfg) {
    return v3_length(v3_sub(domain(p, cfg), v3_load(s->pos))) - s->size;
}

/* ---- public helpers ---------------------------------------------------------------------- */
void ora_default_config(ora_config* cfg) {
    cfg->render_dist = 1000.0f;  /* src/main.rs:362 */
    cfg->cam_fall_off = 0.01f;   /* shaders/fragment.glsl:35 */
    cfg->light_fall_off = 0.01f; /* shaders/fragment.glsl:36 */
    cfg->ray_radius = 0.01f;     /* shaders/fragment.glsl:37 */
    cfg->max_steps = 1u << 20;
    cfg->march_algorithm = 0;
    cfg->repeat[0] = cfg->repeat[1] = cfg->repeat[2] = 0.0f;
    cfg->reflections = 0;    /* the reference has none (fragment.glsl:125 is a TODO) */
    cfg->reflectivity = 0.5f;
    cfg->transmissions = 0;  /* the reference has none (fragment.glsl:124,126 are TODOs) */
    cfg->transparency = 0.5f;
    cfg->refraction_index = 1.0f;
}

/* src/main.rs:524-591 */
void ora_default_scene(ora_scene* s) {
    memset(s, 0, sizeof *s);
    static const float mat_color[4][3] = {{0.2f, 0.2f, 1.0f}, {0.1f, 1.0f, 0.1f}, {1.0f, 1.0f, 0.1f}, {1.0f, 0.1f, 0.1f}};
    static const float mat_shine[4] = {1.0f, 10.0f, 1.0f, 1.0f};
    static const float obj[4][4] = {{5.0f, 5.0f, -1.0f, 3.0f}, {5.0f, 4.0f, 10.0f, 6.0f}, {-3.0f, 3.0f, -3.0f, 1.0f}, {4.0f, -1.0f, 0.0f, 2.0f}};
    static const float light_pos[2][3] = {{-1.0f, 0.0f, -3.0f}, {8.0f, -5.0f, 10.0f}};
    static const float light_col[2][3] = {{0.1f, 0.5f, 0.6f}, {1.2f, 0.2f, 0.3f}};
    s->matCount = 4; s->objCount = 4; s->lightCount = 2;
    for (int i = 0; i < 4; i++) {
        memcpy(s->mats[i].color, mat_color[i], 12);
        s->mats[i].diffuse = 1.0f; s->mats[i].specular = 1.0f;
        s->mats[i].shine = mat_shine[i]; s->mats[i].ambient = 0.05f;
        memcpy(s->objs[i].pos, obj[i], 12); s->objs[i].size = obj[i][3];
    }
    for (int i = 0; i < 2; i++) {
        memcpy(s->lights[i].pos, light_pos[i], 12);
        memcpy(s->lights[i].color, light_col[i], 12);
    }
}

/* src/main.rs:639  (view.x / 8.0).log2() as usize + 1   — floor form; `as usize` saturates
 * negatives to 0; capped at COMPUTE_IMAGE_COUNT (src/main.rs:359). floor(log2(w/8)) equals
 * the index of the top bit of floor(w/8) for w >= 8. */
uint32_t ora_level_count(uint32_t width) {
    uint32_t q = width / 8u, l = 0;
    while (q > 1u) { q >>= 1; l++; }
    uint32_t c = l + 1u;
    return c > ORA_MAX_LEVELS ? ORA_MAX_LEVELS : c;
}

/* src/main.rs:209-213  ratio = res / (4 << count); dims = ceil((1<<i) * ratio) * 8
 * (4<<count and 1<<i are powers of two, so the f32 expression is exact integer ceil-div) */
void ora_level_dims(uint32_t width, uint32_t height, uint32_t count, uint32_t level, uint32_t* w, uint32_t* h) {
    uint64_t den = 4ull << count;
    *w = (uint32_t)((((uint64_t)width << level) + den - 1) / den) * 8u;
    *h = (uint32_t)((((uint64_t)height << level) + den - 1) / den) * 8u;
}

/* src/main.rs:402-404 with glam 0.21.3 (Cargo.lock:465-467):
 * from_rotation_z(a) = (0,0,sin a/2,cos a/2), from_rotation_x(a) = (sin a/2,0,0,cos a/2),
 * Hamilton product; to_array() = [x,y,z,w].  sin/cos are libm here: the quaternion is an INPUT
 * of the hot path, the kernels never recompute it. */
void ora_camera_quat(float yaw, float pitch, float out[4]) {
    float hz = -yaw * 0.5f, hx = pitch * 0.5f;
    float zs = sinf(hz), zc = cosf(hz), xs = sinf(hx), xc = cosf(hx);
    /* (0,0,zs,zc) * (xs,0,0,xc) */
    out[0] = zc * xs; /* x = w1*x2 + x1*w2 + y1*z2 - z1*y2 */
    out[1] = zs * xs; /* y = w1*y2 - x1*z2 + y1*w2 + z1*x2 */
    out[2] = zs * xc; /* z = w1*z2 + x1*y2 - y1*x2 + z1*w2 */
    out[3] = zc * xc; /* w = w1*w2 - x1*x2 - y1*y2 - z1*z2 */
}

void ora_rotate(const float q[4], const float v[3], float out[3]) {
    v3 r = rotate_q(q, v3_load(v));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ---- shaders/compute.glsl:34-68 traceCone --------------------------------------------- */
static float trace_cone3(const ora_scene* sc, const ora_config* cfg, v3 origin, v3 step, float threshold,
                         uint64_t* n_steps, uint64_t* n_sdf) {
    float distances[ORA_MAX_OBJECTS];
    const uint32_t n = sc->objCount;
    for (uint32_t i = 0; i < n; i++) distances[i] = sphere_sdf(origin, &sc->objs[i], cfg); /* :37-39 */
    *n_sdf += n;

    float len = 0.0f, last = 0.0f;
    uint32_t it = 0;
    while (len < cfg->render_dist) { /* :44 */
        if (cfg->max_steps && it++ >= cfg->max_steps) break;
        (*n_steps)++;
        v3 position = v3_fma(step, len, origin);      /* :45 */
        float dist = cfg->render_dist;                /* :49 */
        float radius = (len + 1.0f) * threshold;      /* :50 */
        for (uint32_t i = 0; i < n; i++) {            /* :51-57 */
            distances[i] -= last;
            if (distances[i] <= radius) { distances[i] = sphere_sdf(position, &sc->objs[i], cfg); (*n_sdf)++; }
            dist = fminf(dist, distances[i]);
        }
        last = fmaxf(dist, 0.0f); /* :59 */
        len += last;              /* :60 */
        if (dist <= radius) {     /* :62-65 */
            len -= radius;
            break;
        }
    }
    return len;
}

/* shaders/tracing_algorithms.txt:2-13 ("algorithm 1: checks real distance for each object every
 * update") placed where compute.glsl:46-65 has algorithm 3; the radius is taken AFTER the step, as
 * that listing does. */
static float trace_cone1(const ora_scene* sc, const ora_config* cfg, v3 origin, v3 step, float threshold,
                         uint64_t* n_steps, uint64_t* n_sdf) {
    float len = 0.0f;
    uint32_t it = 0;
    while (len < cfg->render_dist) {
        if (cfg->max_steps && it++ >= cfg->max_steps) break;
        (*n_steps)++;
        v3 position = v3_fma(step, len, origin);
        float dist = sphere_sdf(position, &sc->objs[0], cfg);                                                    /* :2 */
        for (uint32_t i = 1; i < sc->objCount; i++) dist = fminf(dist, sphere_sdf(position, &sc->objs[i], cfg)); /* :3-5 */
        *n_sdf += sc->objCount;
        len += dist;                                /* :7 */
        float radius = (len + 1.0f) * threshold;    /* :8 */
        if (dist <= radius) { len -= radius; break; } /* :9-12 */
    }
    return len;
}

/* shaders/tracing_algorithms.txt:16-37 ("algorithm 2: only checks real distance when necessary"):
 * one SDF evaluation per step, of the object whose cached bound is smallest.  `distances` starts as
 * compute.glsl:37-39 fills it; `position` is the loop-top position of compute.glsl:45, i.e. the
 * refreshed distance belongs to the point BEFORE this step's advance - the listing read literally
 * (the author notes its edges do not "look clean", compute.glsl:47-48). */
static float trace_cone2(const ora_scene* sc, const ora_config* cfg, v3 origin, v3 step, float threshold,
                         uint64_t* n_steps, uint64_t* n_sdf) {
    float distances[ORA_MAX_OBJECTS];
    const uint32_t n = sc->objCount;
    for (uint32_t i = 0; i < n; i++) distances[i] = sphere_sdf(origin, &sc->objs[i], cfg);
    *n_sdf += n;
    uint32_t closest = 0; /* :19 */
    float nearest = 0.0f; /* :20 */
    float len = 0.0f;
    uint32_t it = 0;
    while (len < cfg->render_dist) {
        if (cfg->max_steps && it++ >= cfg->max_steps) break;
        (*n_steps)++;
        v3 position = v3_fma(step, len, origin);
        for (uint32_t i = 0; i < n; i++) { /* :22-27 */
            distances[i] -= nearest;
            if (distances[i] < distances[closest]) closest = i;
        }
        nearest = distances[closest];                                        /* :29 */
        len += nearest;                                                      /* :30 */
        distances[closest] = sphere_sdf(position, &sc->objs[closest], cfg);  /* :31 */
        (*n_sdf)++;
        float radius = (len + 1.0f) * threshold; /* :33 */
        if (distances[closest] <= radius) {      /* :34-37 */
            len += distances[closest] - radius;
            break;
        }
    }
    return len;
}

static float trace_cone(const ora_scene* sc, const ora_config* cfg, v3 origin, v3 step, float threshold,
                        uint64_t* n_steps, uint64_t* n_sdf) {
    switch (cfg->march_algorithm) {
        case 1: return trace_cone1(sc, cfg, origin, step, threshold, n_steps, n_sdf);
        case 2: return trace_cone2(sc, cfg, origin, step, threshold, n_steps, n_sdf);
        default: return trace_cone3(sc, cfg, origin, step, threshold, n_steps, n_sdf);
    }
}

/* ---- shaders/fragment.glsl:89-121 shadowRay ------------------------------------------- */
static float shadow_ray(const ora_scene* sc, const ora_config* cfg, v3 origin, v3 step, float end,
                        uint64_t* n_steps, uint64_t* n_sdf) {
    float distances[ORA_MAX_OBJECTS];
    const uint32_t n = sc->objCount;
    for (uint32_t i = 0; i < n; i++) distances[i] = sphere_sdf(origin, &sc->objs[i], cfg); /* :92-94 */
    *n_sdf += n;

    float last = 0.0f, nearest = 1.0f; /* :96-97 */
    uint32_t it = 0;
    for (float len = 0.0f; len < end; len += last + cfg->ray_radius) { /* :99 */
        if (cfg->max_steps && it++ >= cfg->max_steps) break;
        (*n_steps)++;
        v3 position = v3_fma(step, len, origin); /* :100 */
        float dist = end;                        /* :104 */
        for (uint32_t i = 0; i < n; i++) {       /* :105-111 */
            distances[i] -= last;
            if (distances[i] <= nearest) { distances[i] = sphere_sdf(position, &sc->objs[i], cfg); (*n_sdf)++; }
            dist = fminf(dist, distances[i]);
        }
        if (dist <= cfg->ray_radius) return 0.0f; /* :113-115 */
        last = fmaxf(dist, 0.0f);                 /* :117 */
        nearest = fminf(nearest, dist);           /* :118 */
    }
    return nearest; /* :120 */
}

float ora_trace_cone(const ora_scene* scene, const ora_config* cfg, const float origin[3], const float dir[3], float threshold) {
    uint64_t a = 0, b = 0;
    return trace_cone(scene, cfg, v3_load(origin), v3_load(dir), threshold, &a, &b);
}

float ora_shadow_ray(const ora_scene* scene, const ora_config* cfg, const float origin[3], const float dir[3], float end) {
    uint64_t a = 0, b = 0;
    return shadow_ray(scene, cfg, v3_load(origin), v3_load(dir), end, &a, &b);
}

/* shaders/tracing_algorithms.txt:2-13 ("algorithm 1") as an independent cross-check of algorithm 3. */
float ora_trace_bruteforce(const ora_scene* sc, const ora_config* cfg, const float origin[3], const float dir[3], float threshold) {
    uint64_t a = 0, b = 0;
    return trace_cone1(sc, cfg, v3_load(origin), v3_load(dir), threshold, &a, &b);
}

/* ---- shaders/compute.glsl:70-87 main, one invocation ---------------------------------- */
static inline float cone_pixel(const ora_scene* sc, const ora_config* cfg, uint32_t gx, uint32_t gy, uint32_t iter,
                               const float image_size[2], const float ratio[2], const float rot[4], v3 pos,
                               const float jitter[2], const float* parent, uint32_t parent_w,
                               uint64_t* n_steps, uint64_t* n_sdf) {
    /* :71  (gid*2 + 1) * imageSize - 1   (+ build-side jitter, 0 for the reference) */
    float nx = fmaf((float)(gx * 2u + 1u), image_size[0], -1.0f) + jitter[0];
    float ny = fmaf((float)(gy * 2u + 1u), image_size[1], -1.0f) + jitter[1];
    nx *= ratio[0]; /* :72 */
    ny *= ratio[1];
    float threshold = (1.4142135f * 8.0f) * image_size[0]; /* :75, gl_WorkGroupSize.x = 8 (:5) */
    v3 step = v3_normalize(rotate_q(rot, v3_make(nx, 1.0f, ny))); /* :77 */
    float len = 1.0f; /* :79 */
    if (iter > 0) len = parent[(size_t)(gy >> 1) * parent_w + (gx >> 1)]; /* :80-82, ivec2(gid*0.5) */
    len += trace_cone(sc, cfg, v3_fma(step, len, pos), step, threshold, n_steps, n_sdf); /* :84 */
    return fmaxf(len, 0.0f); /* :86 */
}

/* ---- shaders/fragment.glsl:144-186: nearest sphere, material, light loop for surface point `position` seen from `eye`
 * along the unit direction `step` (for the camera ray eye = push_constants.pos; for a mirror bounce the previous hit) ---- */
typedef struct {
    v3 normal;
    float specular, diffuse;   /* mat.specular / mat.diffuse of the surface: per-material scales of the mirror / transmission chains */
    const ora_object* object;
} surface_t;

static inline void shade_point(const ora_scene* sc, const ora_config* cfg, v3 position, v3 eye, v3 step, float out[3], surface_t* surf,
                               ora_counters* ct) {
    /* :144-156 nearest sphere, strict '<', material index = object index */
    uint32_t best = 0;
    float dist = sphere_sdf(position, &sc->objs[0], cfg);
    for (uint32_t i = 1; i < sc->objCount; i++) {
        float nd = sphere_sdf(position, &sc->objs[i], cfg);
        if (nd < dist) { best = i; dist = nd; }
    }
    const ora_object* object = &sc->objs[best];
    const ora_material* mat = &sc->mats[best];

    float cam_dist = v3_length(v3_sub(position, eye));                              /* :162 */
    float cam_fall = fmaxf(cfg->cam_fall_off * fmaf(cam_dist, cam_dist, 1.0f), 1.0f); /* :163 */
    v3 normal = v3_normalize(v3_sub(domain(position, cfg), v3_load(object->pos)));  /* :166, :39-41 */
    v3 cam_dir = v3_neg(step);
    float normal_fall = fmaxf(v3_dot(normal, cam_dir), 0.0f);                       /* :167 */

    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t i = 0; i < sc->lightCount; i++) { /* :170-186 */
        const ora_light* light = &sc->lights[i];
        v3 to_light = v3_sub(v3_load(light->pos), position);
        v3 light_dir = v3_normalize(to_light);                    /* :173 */
        float light_dist = v3_length(v3_sub(position, v3_load(light->pos))); /* :174 */

        ct->shadow_rays++;
        float soft = fminf(shadow_ray(sc, cfg, v3_add(position, light_dir), light_dir, light_dist,
                                      &ct->shadow_steps, &ct->shadow_sdf), 1.0f); /* :176 */
        float light_fall = fmaxf((cfg->light_fall_off * light_dist) * light_dist, 1.0f); /* :178 */

        float diffuse = fmaxf(v3_dot(normal, light_dir), 0.0f); /* :180, :43-45 */
        /* :181, :47-50  reflect(I,N) = I - 2*dot(N,I)*N with I = -lightDir */
        v3 inc = v3_neg(light_dir);
        float k = 2.0f * v3_dot(normal, inc);
        v3 refl = v3_make(fmaf(-k, normal.x, inc.x), fmaf(-k, normal.y, inc.y), fmaf(-k, normal.z, inc.z));
        float base = v3_dot(refl, cam_dir);
        /* pow(negative, y) is undefined in GLSL (NaN on GPUs, then max(NaN,0) = 0): defined
         * here as 0 for base <= 0 */
        float spec = base > 0.0f ? fmaxf(diffuse * powf(base, mat->shine), 0.0f) : 0.0f;

        float s = fmaxf(diffuse + spec, 0.0f); /* :183 */
        float dr = ((s * light->color[0]) / light_fall) * soft;
        float dg = ((s * light->color[1]) / light_fall) * soft;
        float db = ((s * light->color[2]) / light_fall) * soft;
        /* :185  (ambient + direct) / camDistFallOff * normalFallOff * mat.color */
        r = fmaf(((mat->ambient + dr) / cam_fall) * normal_fall, mat->color[0], r);
        g = fmaf(((mat->ambient + dg) / cam_fall) * normal_fall, mat->color[1], g);
        b = fmaf(((mat->ambient + db) / cam_fall) * normal_fall, mat->color[2], b);
    }
    out[0] = r; out[1] = g; out[2] = b;
    surf->normal = normal;
    surf->specular = mat->specular;
    surf->diffuse = mat->diffuse;
    surf->object = object;
}

/* ---- shaders/fragment.glsl:127-187 main, one invocation -------------------------------- */
static inline void shade_pixel(const ora_scene* sc, const ora_config* cfg, uint32_t px, uint32_t py, const float view[2],
                               const float ratio[2], const float rot[4], v3 pos, const float jitter[2], float total_dist,
                               float out[3], ora_counters* ct) {
    /* :129  gl_FragCoord.xy * 2 / cs.view - 1.0, gl_FragCoord = pixel + 0.5 */
    float nx = (((float)px + 0.5f) * 2.0f) / view[0] - 1.0f + jitter[0];
    float ny = (((float)py + 0.5f) * 2.0f) / view[1] - 1.0f + jitter[1];
    nx *= ratio[0]; /* :131 */
    ny *= ratio[1];
    v3 step = v3_normalize(rotate_q(rot, v3_make(nx, 1.0f, ny))); /* :133 */

    out[0] = out[1] = out[2] = 0.0f;
    if (total_dist >= cfg->render_dist) return; /* :137-140 */
    ct->hit_pixels++;

    v3 position = v3_fma(step, total_dist, pos); /* :142 */
    surface_t first;
    shade_point(sc, cfg, position, pos, step, out, &first, ct);
    const v3 first_position = position, first_step = step;

    /* mirror reflections - build-defined (fragment.glsl:125 is a TODO), specified at ora_config.reflections in oracle.h */
    float weight = 1.0f;
    surface_t surf = first;
    for (uint32_t bounce = 0; bounce < cfg->reflections; bounce++) {
        weight *= cfg->reflectivity * surf.specular;
        const float k = 2.0f * v3_dot(surf.normal, step);
        const v3 r = v3_make(fmaf(-k, surf.normal.x, step.x), fmaf(-k, surf.normal.y, step.y), fmaf(-k, surf.normal.z, step.z)); /* reflect(step, normal) */
        uint64_t dummy_steps = 0, dummy_sdf = 0;
        ct->reflection_rays++;
        const float len = 1.0f + trace_cone3(sc, cfg, v3_add(position, r), r, cfg->ray_radius, &dummy_steps, &dummy_sdf);
        if (!(len < cfg->render_dist)) break;
        const v3 hit = v3_fma(r, fmaxf(len, 0.0f), position);
        float rgb[3];
        shade_point(sc, cfg, hit, position, r, rgb, &surf, ct);
        out[0] = fmaf(weight, rgb[0], out[0]);
        out[1] = fmaf(weight, rgb[1], out[1]);
        out[2] = fmaf(weight, rgb[2], out[2]);
        position = hit;
        step = r;
    }

    /* transmission - build-defined (fragment.glsl:124 "TODO: transparency", :126 "TODO: refraction"), specified at
     * ora_config.transmissions in oracle.h; starts again at the camera ray's hit */
    weight = 1.0f;
    surf = first;
    v3 P = first_position, I = first_step;
    const float idx = cfg->refraction_index;
    const int bend = idx != 1.0f;
    for (uint32_t pass = 0; pass < cfg->transmissions; pass++) {
        weight *= cfg->transparency * surf.diffuse;
        v3 T = I;
        if (bend) { /* T = normalize(refract(I, n, 1 / index)) */
            const float eta = 1.0f / idx;
            const float ci = v3_dot(surf.normal, I);
            const float k = fmaf(-(eta * eta), fmaf(-ci, ci, 1.0f), 1.0f);
            if (k < 0.0f) break;
            const float s = fmaf(eta, ci, sqrtf(k));
            T = v3_normalize(v3_make(fmaf(-s, surf.normal.x, eta * I.x), fmaf(-s, surf.normal.y, eta * I.y), fmaf(-s, surf.normal.z, eta * I.z)));
        }
        /* across the sphere to its far side, in closed form */
        const v3 oc = v3_sub(domain(P, cfg), v3_load(surf.object->pos));
        const float b = v3_dot(oc, T);
        const float cc = fmaf(-surf.object->size, surf.object->size, v3_dot(oc, oc));
        const float disc = fmaf(b, b, -cc);
        float t = disc > 0.0f ? sqrtf(disc) - b : 0.0f;
        if (!(t > 0.0f)) t = 0.0f;
        const v3 Q = v3_fma(T, t, P);
        v3 D = T;
        if (bend) { /* D = normalize(refract(T, -n2, index)), n2 = outward normal at the exit point */
            const v3 n2 = v3_normalize(v3_fma(T, t, oc));
            const float ce = v3_dot(n2, T);
            const float k2 = fmaf(-(idx * idx), fmaf(-ce, ce, 1.0f), 1.0f);
            if (k2 < 0.0f) break; /* total internal reflection */
            const float s2 = fmaf(-idx, ce, sqrtf(k2));
            D = v3_normalize(v3_make(fmaf(s2, n2.x, idx * T.x), fmaf(s2, n2.y, idx * T.y), fmaf(s2, n2.z, idx * T.z)));
        }
        uint64_t dummy_steps = 0, dummy_sdf = 0;
        ct->transmission_rays++;
        const float len = 1.0f + trace_cone3(sc, cfg, v3_add(Q, D), D, cfg->ray_radius, &dummy_steps, &dummy_sdf);
        if (!(len < cfg->render_dist)) break;
        const v3 hit = v3_fma(D, fmaxf(len, 0.0f), Q);
        float rgb[3];
        shade_point(sc, cfg, hit, Q, D, rgb, &surf, ct);
        out[0] = fmaf(weight, rgb[0], out[0]);
        out[1] = fmaf(weight, rgb[1], out[1]);
        out[2] = fmaf(weight, rgb[2], out[2]);
        P = hit;
        I = D;
    }
}

static int scene_valid(const ora_scene* s) {
    return s->objCount >= 1 && s->objCount <= ORA_MAX_OBJECTS && s->lightCount <= ORA_MAX_LIGHTS &&
           s->matCount <= ORA_MAX_MATERIALS;
}

int ora_render_a(const ora_scene* scene, const ora_config* cfg, uint32_t width, uint32_t height, const float ratio[2],
                 const float rot[4], const float pos[3], const float jitter[2], float* const* levels, float* rgb,
                 ora_counters* counters, int threads) {
    if (!scene || !cfg || !ratio || !rot || !pos || width == 0 || height == 0) return -1;
    if (!scene_valid(scene)) return -2;
    static const float zero2[2] = {0.0f, 0.0f};
    if (!jitter) jitter = zero2;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    const uint32_t count = ora_level_count(width);
    const float view[2] = {(float)width, (float)height};
    const v3 p = v3_load(pos);
    ora_counters total;
    memset(&total, 0, sizeof total);

    float* cur = NULL;
    float* prev = NULL;
    uint32_t prev_w = 0;
    /* src/main.rs:300-316: strictly ordered levels, level i reads level i-1 */
    for (uint32_t i = 0; i < count; i++) {
        uint32_t w, h;
        ora_level_dims(width, height, count, i, &w, &h);
        cur = (float*)malloc((size_t)w * h * sizeof(float));
        if (!cur) { free(prev); return -3; }
        /* :303-305  imageSize = 2^(count-1-i) / view */
        const float pw = (float)(1u << (count - 1u - i));
        const float image_size[2] = {pw / view[0], pw / view[1]};
        uint64_t steps = 0, sdf = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : steps, sdf)
        for (uint32_t gy = 0; gy < h; gy++)
            for (uint32_t gx = 0; gx < w; gx++)
                cur[(size_t)gy * w + gx] = cone_pixel(scene, cfg, gx, gy, i, image_size, ratio, rot, p, jitter, prev, prev_w, &steps, &sdf);
        total.cone_threads += (uint64_t)w * h;
        total.cone_steps += steps;
        total.cone_sdf += sdf;
        if (levels && levels[i]) memcpy(levels[i], cur, (size_t)w * h * sizeof(float));
        free(prev);
        prev = cur;
        prev_w = w;
    }
    /* src/main.rs:318-338: full-screen draw, fragment shader reads the last level */
    if (rgb) {
        uint64_t hit = 0, srays = 0, ssteps = 0, ssdf = 0, rrays = 0, trays = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : hit, srays, ssteps, ssdf, rrays, trays)
        for (uint32_t py = 0; py < height; py++) {
            ora_counters ct;
            memset(&ct, 0, sizeof ct);
            for (uint32_t px = 0; px < width; px++)
                shade_pixel(scene, cfg, px, py, view, ratio, rot, p, jitter, prev[(size_t)py * prev_w + px],
                            rgb + ((size_t)py * width + px) * 3, &ct);
            hit += ct.hit_pixels; srays += ct.shadow_rays; ssteps += ct.shadow_steps; ssdf += ct.shadow_sdf; rrays += ct.reflection_rays; trays += ct.transmission_rays;
        }
        total.hit_pixels = hit; total.shadow_rays = srays; total.shadow_steps = ssteps; total.shadow_sdf = ssdf; total.reflection_rays = rrays; total.transmission_rays = trays;
    }
    free(prev);
    if (counters) *counters = total;
    return 0;
}

/* UNORM8 store of a *_UNORM colour attachment (src/main.rs:471-486): linear, clamped;
 * alpha is never written by fragment.glsl (:138,:159) -> defined as 255 */
void ora_to_unorm8(const float* rgb, uint64_t n_pixels, uint8_t* rgba) {
    for (uint64_t i = 0; i < n_pixels; i++) {
        for (int c = 0; c < 3; c++) {
            float v = rgb[i * 3 + c];
            v = v > 0.0f ? v : 0.0f; /* also maps NaN to 0 */
            v = v < 1.0f ? v : 1.0f;
            rgba[i * 4 + c] = (uint8_t)rintf(v * 255.0f);
        }
        rgba[i * 4 + 3] = 255;
    }
}
