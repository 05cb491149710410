/*
 * oracle_b.c — Oracle B: CPU statement of the build-defined triangle/BVH path tracer.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  NO REFERENCE COUNTERPART: the reference
 * (IvoteSligte/raytracing_engine) has no triangles, BVH, RNG, spp or bounces, so
 * this oracle pins nothing against the reference — "parity unpinned by the reference".  It is the
 * executable form of the specification in DESIGN.md §6, which the HIP kernels in
 * raytracing_engine_amd/csrc/path_b.hip implement independently; the camera model is the
 * reference's (shaders/fragment.glsl:129-133, shaders/utilities.glsl:26-29).
 *
 * Results do not depend on the BVH (boxes are conservative, closest hit is the (t, index)
 * lexicographic minimum, occlusion is a boolean), so this file builds its OWN simple BVH
 * (median split) and offers a brute-force mode to prove that independence.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;
static inline v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 neg(v3 a) { return mk(-a.x, -a.y, -a.z); }
static inline v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
static inline v3 fma3(v3 a, float s, v3 b) { return mk(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross(v3 a, v3 b) {
    return mk(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 normalize(v3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }
static inline v3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

static inline v3 rotate_q(const float q[4], v3 v) { /* shaders/utilities.glsl:26-29 */
    v3 qv = mk(q[0], q[1], q[2]);
    v3 c = cross(qv, v);
    v3 t = mk(fmaf(q[3], v.x, c.x), fmaf(q[3], v.y, c.y), fmaf(q[3], v.z, c.z));
    v3 c2 = cross(qv, t);
    return mk(fmaf(2.0f, c2.x, v.x), fmaf(2.0f, c2.y, v.y), fmaf(2.0f, c2.z, v.z));
}

/* ---- spec §6.2: counter-based RNG ---------------------------------------------------------- */
static inline uint32_t hash32(uint32_t x) { /* "lowbias32" integer finaliser */
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline uint32_t path_key(uint32_t pixel, uint32_t sample, uint32_t seed) {
    return hash32(hash32(pixel + hash32(seed)) + sample);
}
static inline float rnd(uint32_t key, uint32_t depth, uint32_t dim) {
    uint32_t h = hash32(key + (depth * 8u + dim + 1u) * 0x9e3779b9U);
    return (float)(h >> 8) * 0x1p-24f; /* [0, 1) on a 2^-24 grid, exact */
}
float orb_rand(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t dim, uint32_t seed) {
    return rnd(path_key(pixel, sample, seed), depth, dim);
}

/* ---- spec §6.5: sin/cos of 2*pi*u from fma polynomials only (no libm, bit-reproducible) ---- */
static inline void sincos_2pi(float u, float* s_out, float* c_out) {
    float t = u * 4.0f;
    uint32_t q = (uint32_t)t;
    if (q > 3u) q = 3u;
    float th = ((t - (float)q) - 0.5f) * 1.57079632679f; /* [-pi/4, pi/4) */
    float th2 = th * th;
    float ps = fmaf(th2, fmaf(th2, fmaf(th2, fmaf(th2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
    float s = th * ps;
    float c = fmaf(th2, fmaf(th2, fmaf(th2, fmaf(th2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
    const float R = 0.70710678f;
    float cA = (q == 0u || q == 3u) ? R : -R; /* angle = q*pi/2 + pi/4 + th */
    float sA = (q < 2u) ? R : -R;
    *c_out = fmaf(cA, c, -(sA * s));
    *s_out = fmaf(sA, c, cA * s);
}
void orb_sincos_2pi(float u, float* s, float* c) { sincos_2pi(u, s, c); }

/* cosine-weighted direction around unit normal n (branchless orthonormal basis) */
static inline v3 cosine_dir(v3 n, float u1, float u2) {
    float r = sqrtf(u1), s, c;
    sincos_2pi(u2, &s, &c);
    float x = r * c, y = r * s, z = sqrtf(fmaxf(0.0f, 1.0f - u1));
    float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n.z);
    float b = (n.x * n.y) * a;
    v3 b1 = mk(fmaf(sign, (n.x * n.x) * a, 1.0f), sign * b, -sign * n.x);
    v3 b2 = mk(b, fmaf(n.y * n.y, a, sign), -n.y);
    return mk(fmaf(n.x, z, fmaf(b2.x, y, b1.x * x)), fmaf(n.y, z, fmaf(b2.y, y, b1.y * x)), fmaf(n.z, z, fmaf(b2.z, y, b1.z * x)));
}
void orb_cosine_dir(const float n[3], float u1, float u2, float out[3]) {
    v3 d = cosine_dir(ld3(n), u1, u2);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}

/* ---- scene --------------------------------------------------------------------------------- */
typedef struct { float lo[2][3], hi[2][3]; int32_t child[2]; int32_t count[2]; } bnode; /* child pair, 64 B */

struct orb_scene {
    uint32_t n;
    float* v0; float* e1; float* e2; /* n*3 each (original order) */
    float* albedo; float* emission;
    uint32_t n_lights; uint32_t* lights; /* emissive triangle indices, ascending */
    uint32_t* order;  /* BVH leaf order -> original triangle index */
    bnode* nodes; uint32_t n_nodes, cap_nodes;
    float pad;
};

/* spec §6.3: ray / triangle (Moeller-Trumbore, division only for accepted candidates) */
static inline int tri_test(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float* t_out) {
    v3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    if (det == 0.0f) return 0;
    v3 tvec = sub(o, v0);
    float u = dot(tvec, pvec);
    v3 qvec = cross(tvec, e1);
    float v = dot(d, qvec);
    if (det > 0.0f) { if (u < 0.0f || v < 0.0f || u + v > det) return 0; }
    else            { if (u > 0.0f || v > 0.0f || u + v < det) return 0; }
    *t_out = dot(e2, qvec) / det;
    return 1;
}

typedef struct { float t; int32_t tri; } hit_t;

static inline void closest_update(const orb_scene* s, uint32_t tri, v3 o, v3 d, hit_t* h) {
    float t;
    if (tri_test(o, d, ld3(s->v0 + 3 * tri), ld3(s->e1 + 3 * tri), ld3(s->e2 + 3 * tri), &t) && t > 0.0f &&
        (t < h->t || (t == h->t && (int32_t)tri < h->tri))) { h->t = t; h->tri = (int32_t)tri; }
}

#define SHADOW_TMAX 0.999f

/* conservative slab test against a padded box; returns entry distance, <0 miss flag via *hit */
static inline int box_test(const float lo[3], const float hi[3], v3 o, v3 inv, float tmax, float* tn_out) {
    float t0x = (lo[0] - o.x) * inv.x, t1x = (hi[0] - o.x) * inv.x;
    float t0y = (lo[1] - o.y) * inv.y, t1y = (hi[1] - o.y) * inv.y;
    float t0z = (lo[2] - o.z) * inv.z, t1z = (hi[2] - o.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    *tn_out = tn;
    return tn <= tf * 1.0000004f;
}
static inline v3 safe_inv(v3 d) {
    float x = fabsf(d.x) > 1e-20f ? d.x : copysignf(1e-20f, d.x);
    float y = fabsf(d.y) > 1e-20f ? d.y : copysignf(1e-20f, d.y);
    float z = fabsf(d.z) > 1e-20f ? d.z : copysignf(1e-20f, d.z);
    return mk(1.0f / x, 1.0f / y, 1.0f / z);
}

static hit_t closest_bvh(const orb_scene* s, v3 o, v3 d, uint64_t* n_nodes, uint64_t* n_tris) {
    hit_t h = {INFINITY, -1};
    v3 inv = safe_inv(d);
    int32_t stack[128];
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        const bnode* nd = &s->nodes[node];
        (*n_nodes)++;
        float tn[2];
        int hit[2];
        hit[0] = box_test(nd->lo[0], nd->hi[0], o, inv, h.t, &tn[0]);
        hit[1] = box_test(nd->lo[1], nd->hi[1], o, inv, h.t, &tn[1]);
        int32_t next[2]; int nn = 0;
        int first = (hit[0] && hit[1] && tn[1] < tn[0]) ? 1 : 0;
        for (int k = 0; k < 2; k++) {
            int c = k ^ first;
            if (!hit[c]) continue;
            if (nd->count[c] > 0) {
                for (int32_t i = 0; i < nd->count[c]; i++) { (*n_tris)++; closest_update(s, s->order[nd->child[c] + i], o, d, &h); }
            } else next[nn++] = nd->child[c];
        }
        if (nn == 2) { stack[sp++] = next[1]; node = next[0]; }
        else if (nn == 1) node = next[0];
        else if (sp > 0) node = stack[--sp];
        else break;
    }
    return h;
}

static int occluded_bvh(const orb_scene* s, v3 o, v3 d, uint64_t* n_nodes, uint64_t* n_tris) {
    v3 inv = safe_inv(d);
    int32_t stack[128];
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        const bnode* nd = &s->nodes[node];
        (*n_nodes)++;
        int32_t next[2]; int nn = 0;
        for (int c = 0; c < 2; c++) {
            float tn;
            if (!box_test(nd->lo[c], nd->hi[c], o, inv, SHADOW_TMAX, &tn)) continue;
            if (nd->count[c] > 0) {
                for (int32_t i = 0; i < nd->count[c]; i++) {
                    uint32_t tri = s->order[nd->child[c] + i];
                    float t;
                    (*n_tris)++;
                    if (tri_test(o, d, ld3(s->v0 + 3 * tri), ld3(s->e1 + 3 * tri), ld3(s->e2 + 3 * tri), &t) && t > 0.0f && t < SHADOW_TMAX) return 1;
                }
            } else next[nn++] = nd->child[c];
        }
        if (nn == 2) { stack[sp++] = next[1]; node = next[0]; }
        else if (nn == 1) node = next[0];
        else if (sp > 0) node = stack[--sp];
        else break;
    }
    return 0;
}

static hit_t closest_brute(const orb_scene* s, v3 o, v3 d, uint64_t* n_tris) {
    hit_t h = {INFINITY, -1};
    for (uint32_t i = 0; i < s->n; i++) closest_update(s, i, o, d, &h);
    *n_tris += s->n;
    return h;
}
static int occluded_brute(const orb_scene* s, v3 o, v3 d, uint64_t* n_tris) {
    for (uint32_t i = 0; i < s->n; i++) {
        float t;
        (*n_tris)++;
        if (tri_test(o, d, ld3(s->v0 + 3 * i), ld3(s->e1 + 3 * i), ld3(s->e2 + 3 * i), &t) && t > 0.0f && t < SHADOW_TMAX) return 1;
    }
    return 0;
}

/* ---- the oracle's own BVH: median split on the longest centroid axis, leaves of <= 4 ------- */
static void tri_bounds(const orb_scene* s, uint32_t tri, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; a++) {
        float p0 = s->v0[3 * tri + a], p1 = p0 + s->e1[3 * tri + a], p2 = p0 + s->e2[3 * tri + a];
        /* v1/v2 are re-derived from the edges; the pad absorbs the rounding */
        lo[a] = fminf(p0, fminf(p1, p2));
        hi[a] = fmaxf(p0, fmaxf(p1, p2));
    }
}
static float centroid(const orb_scene* s, uint32_t tri, int a) {
    return s->v0[3 * tri + a] + (s->e1[3 * tri + a] + s->e2[3 * tri + a]) * (1.0f / 3.0f);
}
static void range_bounds(const orb_scene* s, uint32_t first, uint32_t count, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (uint32_t i = 0; i < count; i++) {
        float l[3], h[3];
        tri_bounds(s, s->order[first + i], l, h);
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], l[a]); hi[a] = fmaxf(hi[a], h[a]); }
    }
    for (int a = 0; a < 3; a++) { lo[a] -= s->pad; hi[a] += s->pad; }
}
static void select_median(const orb_scene* s, uint32_t* idx, uint32_t n, uint32_t k, int axis) { /* quickselect */
    int64_t lo = 0, hi = (int64_t)n - 1;
    while (lo < hi) {
        float pivot = centroid(s, idx[(lo + hi) / 2], axis);
        int64_t i = lo, j = hi;
        do {
            while (centroid(s, idx[i], axis) < pivot) i++;
            while (centroid(s, idx[j], axis) > pivot) j--;
            if (i <= j) { uint32_t t = idx[i]; idx[i] = idx[j]; idx[j] = t; i++; j--; }
        } while (i <= j);
        if ((int64_t)k <= j) hi = j; else if ((int64_t)k >= i) lo = i; else break;
    }
}
/* fills child slot: either a leaf (count>0, child=first) or a new inner node */
static void build_child(orb_scene* s, uint32_t first, uint32_t count, int32_t* child, int32_t* cnt);
static int32_t build_node(orb_scene* s, uint32_t first, uint32_t count) {
    uint32_t me = s->n_nodes++;
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < count; i++)
        for (int a = 0; a < 3; a++) { float c = centroid(s, s->order[first + i], a); clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c); }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    uint32_t half = count / 2;
    select_median(s, s->order + first, count, half, axis);
    int32_t c[2], k[2];
    build_child(s, first, half, &c[0], &k[0]);
    build_child(s, first + half, count - half, &c[1], &k[1]);
    bnode* nd = &s->nodes[me];
    range_bounds(s, first, half, nd->lo[0], nd->hi[0]);
    range_bounds(s, first + half, count - half, nd->lo[1], nd->hi[1]);
    nd->child[0] = c[0]; nd->child[1] = c[1]; nd->count[0] = k[0]; nd->count[1] = k[1];
    return (int32_t)me;
}
static void build_child(orb_scene* s, uint32_t first, uint32_t count, int32_t* child, int32_t* cnt) {
    if (count <= 4) { *child = (int32_t)first; *cnt = (int32_t)count; }
    else { *child = build_node(s, first, count); *cnt = 0; }
}

orb_scene* orb_scene_create(const orb_mesh* m) {
    if (!m || m->n_tris == 0 || !m->verts || !m->albedo || !m->emission) return NULL;
    orb_scene* s = (orb_scene*)calloc(1, sizeof *s);
    uint32_t n = s->n = m->n_tris;
    s->v0 = malloc(12u * (size_t)n); s->e1 = malloc(12u * (size_t)n); s->e2 = malloc(12u * (size_t)n);
    s->albedo = malloc(12u * (size_t)n); s->emission = malloc(12u * (size_t)n);
    s->order = malloc(4u * (size_t)n); s->lights = malloc(4u * (size_t)n);
    float maxabs = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        const float* v = m->verts + 9 * (size_t)i;
        for (int a = 0; a < 3; a++) {
            s->v0[3 * i + a] = v[a];
            s->e1[3 * i + a] = v[3 + a] - v[a]; /* spec §6.1: edges are formed once, in fp32 */
            s->e2[3 * i + a] = v[6 + a] - v[a];
            maxabs = fmaxf(maxabs, fmaxf(fabsf(v[a]), fmaxf(fabsf(v[3 + a]), fabsf(v[6 + a]))));
        }
        s->order[i] = i;
        if (m->emission[3 * i] > 0.0f || m->emission[3 * i + 1] > 0.0f || m->emission[3 * i + 2] > 0.0f) s->lights[s->n_lights++] = i;
    }
    memcpy(s->albedo, m->albedo, 12u * (size_t)n);
    memcpy(s->emission, m->emission, 12u * (size_t)n);
    s->pad = 2e-5f * fmaxf(maxabs, 1.0f);
    s->cap_nodes = n + 1;
    s->nodes = calloc(s->cap_nodes, sizeof(bnode));
    if (n <= 4) { /* one node; both child slots name the same leaf (testing it twice is idempotent) */
        bnode* nd = &s->nodes[0];
        s->n_nodes = 1;
        for (int c = 0; c < 2; c++) {
            range_bounds(s, 0, n, nd->lo[c], nd->hi[c]);
            nd->child[c] = 0; nd->count[c] = (int32_t)n;
        }
    } else build_node(s, 0, n);
    return s;
}

void orb_scene_destroy(orb_scene* s) {
    if (!s) return;
    free(s->v0); free(s->e1); free(s->e2); free(s->albedo); free(s->emission); free(s->order); free(s->lights); free(s->nodes);
    free(s);
}

int32_t orb_closest_hit(const orb_scene* s, const float origin[3], const float dir[3], float* t_out, int use_bvh) {
    uint64_t a = 0, b = 0;
    hit_t h = use_bvh ? closest_bvh(s, ld3(origin), ld3(dir), &a, &b) : closest_brute(s, ld3(origin), ld3(dir), &b);
    if (t_out) *t_out = h.t;
    return h.tri;
}
int orb_occluded(const orb_scene* s, const float origin[3], const float dir[3], int use_bvh) {
    uint64_t a = 0, b = 0;
    return use_bvh ? occluded_bvh(s, ld3(origin), ld3(dir), &a, &b) : occluded_brute(s, ld3(origin), ld3(dir), &b);
}

/* ---- spec §6.4-6.6: one path -------------------------------------------------------------- */
static void trace_path(const orb_scene* s, const orb_params* p, uint32_t px, uint32_t py, uint32_t sample, int use_bvh,
                       float L[3], orb_counters* ct) {
    const uint32_t key = path_key(py * p->width + px, sample, p->seed);
    /* camera ray: fragment.glsl:129-133 with the pixel-centre 0.5 replaced by a random offset */
    float nx = ((((float)px + rnd(key, 0, 0)) * 2.0f) / (float)p->width - 1.0f) * p->ratio[0];
    float ny = ((((float)py + rnd(key, 0, 1)) * 2.0f) / (float)p->height - 1.0f) * p->ratio[1];
    v3 d = normalize(rotate_q(p->rot, mk(nx, 1.0f, ny)));
    v3 o = ld3(p->pos);
    float Tr = 1.0f, Tg = 1.0f, Tb = 1.0f;
    L[0] = L[1] = L[2] = 0.0f;
    for (uint32_t depth = 0;; depth++) {
        if (depth == 0) ct->camera_rays++; else ct->bounce_rays++;
        hit_t h = use_bvh ? closest_bvh(s, o, d, &ct->nodes_visited, &ct->tris_tested) : closest_brute(s, o, d, &ct->tris_tested);
        if (h.tri < 0) { /* left the scene */
            L[0] = fmaf(Tr, p->sky[0], L[0]); L[1] = fmaf(Tg, p->sky[1], L[1]); L[2] = fmaf(Tb, p->sky[2], L[2]);
            break;
        }
        const uint32_t tri = (uint32_t)h.tri;
        const float* em = s->emission + 3 * tri;
        if (em[0] > 0.0f || em[1] > 0.0f || em[2] > 0.0f) { /* lights are seen directly only by camera rays (NEE covers the rest) */
            if (depth == 0) { L[0] = fmaf(Tr, em[0], L[0]); L[1] = fmaf(Tg, em[1], L[1]); L[2] = fmaf(Tb, em[2], L[2]); }
            break;
        }
        const float* alb = s->albedo + 3 * tri;
        v3 n = normalize(cross(ld3(s->e1 + 3 * tri), ld3(s->e2 + 3 * tri)));
        if (dot(n, d) > 0.0f) n = neg(n);
        v3 pt = fma3(d, h.t, o);
        v3 po = fma3(n, p->ray_eps, pt);
        if (s->n_lights > 0) { /* next-event estimation: one uniformly chosen light triangle, uniform point on it */
            uint32_t k = (uint32_t)(rnd(key, depth, 2) * (float)s->n_lights);
            if (k > s->n_lights - 1) k = s->n_lights - 1;
            const uint32_t lt = s->lights[k];
            float su = sqrtf(rnd(key, depth, 3)), u2 = rnd(key, depth, 4);
            float b1 = su * (1.0f - u2), b2 = su * u2;
            v3 lv0 = ld3(s->v0 + 3 * lt), le1 = ld3(s->e1 + 3 * lt), le2 = ld3(s->e2 + 3 * lt);
            v3 q = mk(fmaf(le2.x, b2, fmaf(le1.x, b1, lv0.x)), fmaf(le2.y, b2, fmaf(le1.y, b1, lv0.y)), fmaf(le2.z, b2, fmaf(le1.z, b1, lv0.z)));
            v3 wi = sub(q, po);
            float d2 = dot(wi, wi);
            v3 nl = cross(le1, le2); /* |nl| = 2 * area */
            float cs = dot(n, wi), cl = fabsf(dot(nl, wi));
            if (cs > 0.0f && cl > 0.0f && d2 > 0.0f) {
                /* cos_s cos_l / d^2 * area * n_lights / pi, with the normalisations folded in */
                float w = ((cs * cl) * ((float)s->n_lights * 0.15915494f)) / (d2 * d2);
                const float* le = s->emission + 3 * lt;
                float cr = ((Tr * alb[0]) * le[0]) * w, cg = ((Tg * alb[1]) * le[1]) * w, cb = ((Tb * alb[2]) * le[2]) * w;
                ct->shadow_rays++;
                int occ = use_bvh ? occluded_bvh(s, po, wi, &ct->nodes_visited, &ct->tris_tested) : occluded_brute(s, po, wi, &ct->tris_tested);
                if (!occ) { L[0] += cr; L[1] += cg; L[2] += cb; }
            }
        }
        if (depth >= p->bounces) break;
        d = cosine_dir(n, rnd(key, depth, 5), rnd(key, depth, 6));
        o = po;
        Tr *= alb[0]; Tg *= alb[1]; Tb *= alb[2];
    }
}

int orb_render(const orb_scene* s, const orb_params* p, float* rgb, orb_counters* ct_out, int use_bvh, int threads) {
    return orb_render_rows(s, p, 0, p ? p->height : 0, rgb, ct_out, use_bvh, threads);
}

/* rows [row0, row1) of the frame only (rgb holds (row1-row0)*width*3 floats): lets tests compare a
 * band of a full-size frame, which works because the RNG is keyed by the global pixel index */
int orb_render_rows(const orb_scene* s, const orb_params* p, uint32_t row0, uint32_t row1, float* rgb, orb_counters* ct_out, int use_bvh,
                    int threads) {
    if (!s || !p || !rgb || p->width == 0 || p->height == 0 || p->spp == 0 || row0 >= row1 || row1 > p->height) return -1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    uint64_t cam = 0, bnc = 0, shd = 0, nv = 0, tt = 0;
#pragma omp parallel for schedule(dynamic, 2) num_threads(threads) reduction(+ : cam, bnc, shd, nv, tt)
    for (uint32_t py = row0; py < row1; py++) {
        orb_counters ct;
        memset(&ct, 0, sizeof ct);
        for (uint32_t px = 0; px < p->width; px++) {
            float acc[3] = {0.0f, 0.0f, 0.0f};
            for (uint32_t sidx = 0; sidx < p->spp; sidx++) { /* spec §6.6: samples are summed in index order */
                float L[3];
                trace_path(s, p, px, py, sidx, use_bvh, L, &ct);
                acc[0] += L[0]; acc[1] += L[1]; acc[2] += L[2];
            }
            float* o = rgb + ((size_t)(py - row0) * p->width + px) * 3;
            o[0] = acc[0] / (float)p->spp; o[1] = acc[1] / (float)p->spp; o[2] = acc[2] / (float)p->spp;
        }
        cam += ct.camera_rays; bnc += ct.bounce_rays; shd += ct.shadow_rays; nv += ct.nodes_visited; tt += ct.tris_tested;
    }
    if (ct_out) {
        ct_out->camera_rays = cam; ct_out->bounce_rays = bnc; ct_out->shadow_rays = shd;
        ct_out->nodes_visited = nv; ct_out->tris_tested = tt; ct_out->n_nodes = s->n_nodes;
    }
    return 0;
}
