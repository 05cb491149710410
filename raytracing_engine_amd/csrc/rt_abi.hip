// rt_abi.hip — C-ABI entry points of librt_amd.so (include/rt_abi.h) and the host-side launch
// schedule they drive.  The schedule restates get_command_buffer (src/main.rs:268-341 of the
// reference): for each pyramid level, in order, one cone-march launch whose push constants are
// {iter = i, imageSize = 2^(count-1-i)/view}; then one shading pass over the last level.  HIP
// stream order replaces the image barriers vulkano inserts between the dispatches.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "rt_internal.h"
#include "rt_roctx.h"

using rt::Ctx;

namespace {

thread_local std::string g_create_err;

// src/main.rs:639  (view.x / 8.0).log2() as usize + 1  — floor form (the resize path at :845
// uses ceil and overflows the 9-image descriptor array at 4K; the floor form is the one the
// reference starts with), capped at COMPUTE_IMAGE_COUNT = 9 (src/main.rs:359).
uint32_t level_count_for(uint32_t width) {
    uint32_t q = width / 8u, l = 0;
    while (q > 1u) {
        q >>= 1;
        l++;
    }
    const uint32_t c = l + 1u;
    return c > RT_MAX_LEVELS ? RT_MAX_LEVELS : c;
}

// src/main.rs:209-213  dims = ceil((1 << i) * res / (4 << count)) * 8 (exact in integers)
void level_dims_for(uint32_t width, uint32_t height, uint32_t count, uint32_t level, uint32_t* w, uint32_t* h) {
    const uint64_t den = 4ull << count;
    *w = (uint32_t)((((uint64_t)width << level) + den - 1) / den) * 8u;
    *h = (uint32_t)((((uint64_t)height << level) + den - 1) / den) * 8u;
}

void free_frame(Ctx* c) {
    for (auto& p : c->d_level) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    if (c->d_rgb) (void)hipFree(c->d_rgb);
    c->d_rgb = nullptr;
    if (c->d_rgba8) (void)hipFree(c->d_rgba8);
    c->d_rgba8 = nullptr;
    c->level_batch = 0;
    c->last_image = 0;
    c->frame_valid = false;
}

int bind(Ctx* c) {
    RT_HIP(c, hipSetDevice(c->device));
    return RT_OK;
}

void fill_sphere_set(const rt_mutable_data& s, rt::SphereSet* out) {
    for (uint32_t i = 0; i < RT_MAX_OBJECTS; i++)
        out->s[i] = make_float4(s.objs[i].pos[0], s.objs[i].pos[1], s.objs[i].pos[2], s.objs[i].size);
}

void fill_shade_set(const rt_mutable_data& s, rt::ShadeSet* out) {
    for (uint32_t i = 0; i < RT_MAX_OBJECTS; i++) {
        out->sphere[i] = make_float4(s.objs[i].pos[0], s.objs[i].pos[1], s.objs[i].pos[2], s.objs[i].size);
        // material index = object index (fragment.glsl:154); mat.diffuse / mat.specular are never read
        out->mat_color_ambient[i] = make_float4(s.mats[i].color[0], s.mats[i].color[1], s.mats[i].color[2], s.mats[i].ambient);
        out->mat_shine[i] = s.mats[i].shine;
        out->mat_specular[i] = s.mats[i].specular;
        out->mat_diffuse[i] = s.mats[i].diffuse;
    }
    for (uint32_t i = 0; i < RT_MAX_LIGHTS; i++) {
        out->light_pos[i] = make_float4(s.lights[i].pos[0], s.lights[i].pos[1], s.lights[i].pos[2], 0.0f);
        out->light_color[i] = make_float4(s.lights[i].color[0], s.lights[i].color[1], s.lights[i].color[2], 0.0f);
    }
    out->light_count = s.lightCount;
}

uint32_t owned_tiles(const rt::Partition& p) {
    const uint32_t total = p.tiles_x * p.tiles_y;
    return total > p.rank ? (total - p.rank + p.n_ranks - 1u) / p.n_ranks : 0u;
}

// spp must be n*n; returns n or 0
uint32_t strata_of(uint32_t spp) {
    for (uint32_t n = 1; n <= 64; n++)
        if (n * n == spp) return n;
    return 0;
}

// Samples of one pixel are independent full frames, so up to kSampleBatch of them share each launch
// (grid.y = sample for the pyramid levels, an in-thread loop in index order for shading): the coarse
// levels are latency-bound chains of a few hundred threads and cost the same for 1 or 16 samples.
constexpr uint32_t kSampleBatch = 16;

// Every level holds `batch` images (one per sample of a batch).
int ensure_levels(Ctx* c, uint32_t batch) {
    if (batch <= c->level_batch) return RT_OK;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < c->level_count; i++) {
        if (c->d_level[i]) (void)hipFree(c->d_level[i]);
        c->d_level[i] = nullptr;
    }
    c->level_batch = 0;
    c->frame_valid = false;
    for (uint32_t i = 0; i < c->level_count; i++) {
        const size_t bytes = (size_t)c->dims[i][0] * c->dims[i][1] * sizeof(float) * batch;
        // zero-filled ON THE CONTEXT'S STREAM: its streams are non-blocking, so a hipMemset on the null stream (asynchronous
        // to the host for device memory) would not be ordered before the level kernels that follow and could wipe their output
        if (hipMalloc((void**)&c->d_level[i], bytes) != hipSuccess || hipMemsetAsync(c->d_level[i], 0, bytes, c->stream) != hipSuccess)
            return c->fail(RT_ERR_OOM, "pyramid level %u x %u samples (%zu bytes)", i, batch, bytes);
    }
    c->level_batch = batch;
    return RT_OK;
}

// Enqueue samples s0 .. s0 + nb - 1 of spp: all pyramid levels, then shading (which adds the samples
// to the running sum in index order).
int enqueue_samples(Ctx* c, const float rot[4], const float pos[3], uint32_t s0, uint32_t nb, uint32_t n_strata, uint32_t spp, float* dst,
                    int tile_major, bool stage_events) {
    rt::Camera cam{};
    std::memcpy(cam.rot, rot, 16);
    std::memcpy(cam.pos, pos, 12);
    cam.ratio[0] = c->ratio[0];
    cam.ratio[1] = c->ratio[1];
    rt::sample_jitter(s0, n_strata, c->width, c->height, &cam.jitter[0], &cam.jitter[1]);  // used by the fused schedule (nb = 1)

    const uint32_t count = c->level_count;
    uint32_t ev = 0;
    rt::SphereSet spheres;
    fill_sphere_set(c->scene, &spheres);
    if (c->cfg.fuse_levels) {  // one launch for the whole pyramid (path_a.hip pyramid_tile_kernel), one sample at a time
        rt::PyramidParams fp{};
        fp.cam = cam;
        fp.width = c->width;
        fp.height = c->height;
        fp.render_dist = c->cfg.render_dist;
        fp.max_steps = c->cfg.max_steps;
        fp.part = c->part;
        fp.count = count;
        for (uint32_t i = 0; i < count; i++) {
            const float pw = (float)(1u << (count - 1u - i));  // src/main.rs:303-305
            fp.image_size[i][0] = pw / (float)c->width;
            fp.image_size[i][1] = pw / (float)c->height;
            fp.level_w[i] = c->dims[i][0];
            fp.level[i] = c->d_level[i];
        }
        rt::RoctxRange rr("rt.path_a.pyramid_fused");
        if (stage_events) RT_HIP(c, hipEventRecord(c->ev_stage[ev++], c->stream));
        if (int rc = rt::launch_pyramid_fused(c, spheres, c->scene.objCount, fp)) return rc;
    } else {
        const bool partitioned = c->part.n_ranks > 1;
        for (uint32_t i = 0; i < count; i++) {  // src/main.rs:300-316
            rt::ConeLevelParams p{};
            p.cam = cam;
            const float pw = (float)(1u << (count - 1u - i));  // :303-305
            p.image_size[0] = pw / (float)c->width;
            p.image_size[1] = pw / (float)c->height;
            p.level = i;
            p.w = c->dims[i][0];
            p.h = c->dims[i][1];
            p.parent_w = i ? c->dims[i - 1][0] : 0;
            p.shift = count - 1u - i;
            p.width = c->width;
            p.height = c->height;
            p.render_dist = c->cfg.render_dist;
            p.max_steps = c->cfg.max_steps;
            p.part = c->part;
            p.partitioned = partitioned ? 1u : 0u;
            p.sample0 = s0;
            p.n_strata = n_strata;
            p.level_stride = c->dims[i][0] * c->dims[i][1];
            p.parent_stride = i ? c->dims[i - 1][0] * c->dims[i - 1][1] : 0;
            p.alg = c->cfg.march_algorithm ? c->cfg.march_algorithm : 3u;
            std::memcpy(p.repeat, c->cfg.repeat, sizeof p.repeat);
            rt::RoctxRange rr("rt.path_a.cone_level", i);
            if (stage_events) RT_HIP(c, hipEventRecord(c->ev_stage[ev++], c->stream));
            int rc = rt::launch_cone_level(c, spheres, c->scene.objCount, p, i ? c->d_level[i - 1] : nullptr, c->d_level[i], nb);
            if (rc) return rc;
        }
    }
    if (stage_events) RT_HIP(c, hipEventRecord(c->ev_stage[ev++], c->stream));

    rt::ShadeSet set;
    fill_shade_set(c->scene, &set);
    rt::ShadeParams sp{};
    sp.cam = cam;
    sp.view[0] = (float)c->width;
    sp.view[1] = (float)c->height;
    sp.width = c->width;
    sp.height = c->height;
    sp.depth_w = c->dims[count - 1][0];
    sp.render_dist = c->cfg.render_dist;
    sp.cam_fall_off = c->cfg.cam_fall_off;
    sp.light_fall_off = c->cfg.light_fall_off;
    sp.ray_radius = c->cfg.ray_radius;
    sp.max_steps = c->cfg.max_steps;
    sp.part = c->part;
    sp.tile_major = tile_major ? 1u : 0u;
    sp.mode = (s0 > 0 ? 1u : 0u) | ((spp > 1 && s0 + nb == spp) ? 2u : 0u);
    sp.spp = (float)spp;
    sp.sample0 = s0;
    sp.n_batch = nb;
    sp.n_strata = n_strata;
    sp.depth_stride = c->dims[count - 1][0] * c->dims[count - 1][1];
    std::memcpy(sp.repeat, c->cfg.repeat, sizeof sp.repeat);
    sp.reflections = c->cfg.reflections;
    sp.reflectivity = c->cfg.reflectivity;
    sp.transmissions = c->cfg.transmissions;
    sp.transparency = c->cfg.transparency;
    sp.refraction_index = c->cfg.refraction_index;
    int rc;
    {
        rt::RoctxRange rr("rt.path_a.shade");
        rc = rt::launch_shade(c, set, c->scene.objCount, sp, c->d_level[count - 1], dst, c->d_counters);
    }
    if (rc) return rc;
    if (stage_events) RT_HIP(c, hipEventRecord(c->ev_stage[ev++], c->stream));
    c->last_image = nb - 1u;
    return RT_OK;
}

int render_common(Ctx* c, const float rot[4], const float pos[3], uint32_t spp, float* dst_dev, int tile_major, bool sync) {
    if (!c) return RT_ERR_INVALID;
    if (!rot || !pos) return c->fail(RT_ERR_INVALID, "rot/pos must not be NULL");
    if (!c->have_scene) return c->fail(RT_ERR_STATE, "rt_set_scene has not been called");
    if (!c->width) return c->fail(RT_ERR_STATE, "rt_resize has not been called");
    const uint32_t n_strata = strata_of(spp);
    if (!n_strata) return c->fail(RT_ERR_INVALID, "spp %u is not a square n*n (n <= 64)", spp);
    if (int rc = bind(c)) return rc;

    const bool stage_events = c->cfg.profile_stages != 0;
    RT_HIP(c, hipMemsetAsync(c->d_counters, 0, 4 * 1024 * sizeof(uint64_t), c->stream));
    RT_HIP(c, hipEventRecord(c->ev_begin, c->stream));
    const uint32_t batch = c->cfg.fuse_levels ? 1u : std::min(spp, kSampleBatch);
    if (int rc = ensure_levels(c, batch)) return rc;
    for (uint32_t s = 0; s < spp; s += batch) {
        // per-stage events only bracket the last batch (ev_stage is reused per batch)
        const uint32_t nb = std::min(batch, spp - s);
        int rc = enqueue_samples(c, rot, pos, s, nb, n_strata, spp, dst_dev, tile_major, stage_events && s + nb == spp);
        if (rc) return rc;
    }
    RT_HIP(c, hipEventRecord(c->ev_end, c->stream));
    c->frame_valid = true;
    c->stats.frames++;
    c->stats.spp = spp;

    uint64_t cone_threads = 0;
    for (uint32_t i = 0; i < c->level_count; i++) cone_threads += (uint64_t)c->dims[i][0] * c->dims[i][1];
    c->stats.cone_threads = cone_threads * spp;  // upper bound when partitioned
    if (sync) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        uint64_t slots[4 * 1024], counters[4] = {0, 0, 0, 0};  // hit pixels, secondary hits shaded, mirror rays, transmitted rays
        RT_HIP(c, hipMemcpy(slots, c->d_counters, sizeof slots, hipMemcpyDeviceToHost));
        for (int k = 0; k < 4 * 1024; k++) counters[k >> 10] += slots[k];
        uint64_t owned_px = 0;
        {  // pixels inside the frame that belong to this rank's tiles
            const rt::Partition& pt = c->part;
            for (uint32_t t = pt.rank; t < pt.tiles_x * pt.tiles_y; t += pt.n_ranks) {
                const uint32_t ty = t / pt.tiles_x, tx = t % pt.tiles_x;
                const uint32_t w = std::min<uint32_t>(RT_TILE, c->width - tx * RT_TILE), h = std::min<uint32_t>(RT_TILE, c->height - ty * RT_TILE);
                owned_px += (uint64_t)w * h;
            }
        }
        c->stats.primary_rays = owned_px * spp;
        c->stats.hit_pixels = counters[0];
        c->stats.shadow_rays = (counters[0] + counters[1]) * c->scene.lightCount;  // one shadowRay per light per shaded surface point
        c->stats.reflection_rays = counters[2];
        c->stats.transmission_rays = counters[3];
        RT_HIP(c, hipEventElapsedTime(&c->stats.ms_total, c->ev_begin, c->ev_end));
        c->stats.ms_cone = c->stats.ms_shade = 0.0f;
        std::memset(c->stats.ms_level, 0, sizeof c->stats.ms_level);
        c->stats.ms_fused = 0.0f;
        if (stage_events && c->cfg.fuse_levels) {
            RT_HIP(c, hipEventElapsedTime(&c->stats.ms_fused, c->ev_stage[0], c->ev_stage[1]));
            c->stats.ms_cone = c->stats.ms_fused;
            RT_HIP(c, hipEventElapsedTime(&c->stats.ms_shade, c->ev_stage[1], c->ev_stage[2]));
        } else if (stage_events) {
            for (uint32_t i = 0; i < c->level_count; i++) {
                RT_HIP(c, hipEventElapsedTime(&c->stats.ms_level[i], c->ev_stage[i], c->ev_stage[i + 1]));
                c->stats.ms_cone += c->stats.ms_level[i];
            }
            RT_HIP(c, hipEventElapsedTime(&c->stats.ms_shade, c->ev_stage[c->level_count], c->ev_stage[c->level_count + 1]));
        }
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(int* count) {
    if (!count) return RT_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        *count = 0;
        return RT_ERR_NO_DEVICE;
    }
    *count = n;
    return RT_OK;
}

int rt_default_config(rt_config* cfg) {
    if (!cfg) return RT_ERR_INVALID;
    cfg->render_dist = 1000.0f;   // src/main.rs:362
    cfg->cam_fall_off = 0.01f;    // shaders/fragment.glsl:35
    cfg->light_fall_off = 0.01f;  // shaders/fragment.glsl:36
    cfg->ray_radius = 0.01f;      // shaders/fragment.glsl:37
    cfg->max_steps = 1u << 20;
    cfg->profile_stages = 0;
    cfg->fuse_levels = 0;  // measured slower than one launch per level (path_a.hip)
    cfg->march_algorithm = 0;
    cfg->repeat[0] = cfg->repeat[1] = cfg->repeat[2] = 0.0f;
    cfg->reflections = 0;  // the reference has none (fragment.glsl:125 is a TODO)
    cfg->reflectivity = 0.5f;
    cfg->transmissions = 0;  // the reference has none (fragment.glsl:124,126 are TODOs)
    cfg->transparency = 0.5f;
    cfg->refraction_index = 1.0f;
    return RT_OK;
}

// src/main.rs:524-591
int rt_default_scene(rt_mutable_data* s) {
    if (!s) return RT_ERR_INVALID;
    std::memset(s, 0, sizeof *s);
    const float colors[4][3] = {{0.2f, 0.2f, 1.0f}, {0.1f, 1.0f, 0.1f}, {1.0f, 1.0f, 0.1f}, {1.0f, 0.1f, 0.1f}};
    const float shine[4] = {1.0f, 10.0f, 1.0f, 1.0f};
    const float spheres[4][4] = {{5.0f, 5.0f, -1.0f, 3.0f}, {5.0f, 4.0f, 10.0f, 6.0f}, {-3.0f, 3.0f, -3.0f, 1.0f}, {4.0f, -1.0f, 0.0f, 2.0f}};
    const float lpos[2][3] = {{-1.0f, 0.0f, -3.0f}, {8.0f, -5.0f, 10.0f}};
    const float lcol[2][3] = {{0.1f, 0.5f, 0.6f}, {1.2f, 0.2f, 0.3f}};
    s->matCount = 4;
    s->objCount = 4;
    s->lightCount = 2;
    for (int i = 0; i < 4; i++) {
        std::memcpy(s->mats[i].color, colors[i], 12);
        s->mats[i].diffuse = s->mats[i].specular = 1.0f;
        s->mats[i].shine = shine[i];
        s->mats[i].ambient = 0.05f;
        std::memcpy(s->objs[i].pos, spheres[i], 12);
        s->objs[i].size = spheres[i][3];
    }
    for (int i = 0; i < 2; i++) {
        std::memcpy(s->lights[i].pos, lpos[i], 12);
        std::memcpy(s->lights[i].color, lcol[i], 12);
    }
    return RT_OK;
}

int rt_create(rt_ctx** out, int device_ordinal) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_create_err = "no HIP device available (librt_amd has no CPU fallback)";
        return RT_ERR_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        g_create_err = "device ordinal " + std::to_string(device_ordinal) + " out of range [0," + std::to_string(n) + ")";
        return RT_ERR_NO_DEVICE;
    }
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return RT_ERR_OOM;
    c->device = device_ordinal;
    rt_default_config(&c->cfg);
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_begin);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_end);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_counters, 4 * 1024 * sizeof(uint64_t));
    for (uint32_t i = 0; e == hipSuccess && i < RT_MAX_LEVELS + 2; i++) {
        hipEvent_t ev;
        e = hipEventCreate(&ev);
        if (e == hipSuccess) c->ev_stage.push_back(ev);
    }
    if (e != hipSuccess) {
        g_create_err = std::string("HIP initialisation failed: ") + hipGetErrorString(e);
        rt_destroy(reinterpret_cast<rt_ctx*>(c));
        return RT_ERR_HIP;
    }
    c->stream = c->own_stream;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_ordinal) == hipSuccess && cus > 0) c->n_cus = cus;
    *out = reinterpret_cast<rt_ctx*>(c);
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    rt::frames_free(c);
    rt::comm_free(c);
    free_frame(c);
    rt::pt_free(c);
    if (c->d_counters) (void)hipFree(c->d_counters);
    for (auto ev : c->ev_stage) (void)hipEventDestroy(ev);
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_end) (void)hipEventDestroy(c->ev_end);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* rt_last_error(const rt_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? c->err.c_str() : g_create_err.c_str();
}

int rt_set_config(rt_ctx* ctx, const rt_config* cfg) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!cfg) return c->fail(RT_ERR_INVALID, "cfg is NULL");
    if (!(cfg->render_dist > 0.0f) || !(cfg->ray_radius > 0.0f)) return c->fail(RT_ERR_INVALID, "render_dist and ray_radius must be > 0");
    if (cfg->march_algorithm > 3) return c->fail(RT_ERR_INVALID, "march_algorithm %u (0..3)", cfg->march_algorithm);
    const bool variant = (cfg->march_algorithm != 0 && cfg->march_algorithm != 3) || cfg->repeat[0] > 0.0f || cfg->repeat[1] > 0.0f || cfg->repeat[2] > 0.0f;
    for (float r : cfg->repeat)
        if (!(r >= 0.0f) || !(r < 3.0e38f)) return c->fail(RT_ERR_INVALID, "repeat periods must be finite and >= 0");
    if (variant && cfg->fuse_levels) return c->fail(RT_ERR_INVALID, "march_algorithm / repeat need fuse_levels = 0");
    if (cfg->reflections > 8u || !(cfg->reflectivity >= 0.0f) || !(cfg->reflectivity <= 1.0f)) return c->fail(RT_ERR_INVALID, "reflections %u (0..8) / reflectivity %g (0..1)", cfg->reflections, (double)cfg->reflectivity);
    if (cfg->transmissions > 8u || !(cfg->transparency >= 0.0f) || !(cfg->transparency <= 1.0f) || !(cfg->refraction_index >= 1.0f) || !(cfg->refraction_index <= 4.0f))
        return c->fail(RT_ERR_INVALID, "transmissions %u (0..8) / transparency %g (0..1) / refraction_index %g (1..4)", cfg->transmissions, (double)cfg->transparency,
                       (double)cfg->refraction_index);
    c->cfg = *cfg;
    c->state_version++;
    return RT_OK;
}

int rt_set_scene(rt_ctx* ctx, const void* mutable_data, size_t bytes) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!mutable_data) return c->fail(RT_ERR_INVALID, "scene is NULL");
    if (bytes != sizeof(rt_mutable_data)) return c->fail(RT_ERR_INVALID, "scene is %zu bytes, MutableData is %zu", bytes, sizeof(rt_mutable_data));
    rt_mutable_data s;
    std::memcpy(&s, mutable_data, sizeof s);
    if (s.objCount < 1 || s.objCount > RT_MAX_OBJECTS) return c->fail(RT_ERR_INVALID, "objCount %u out of [1,%u]", s.objCount, RT_MAX_OBJECTS);
    if (s.lightCount > RT_MAX_LIGHTS) return c->fail(RT_ERR_INVALID, "lightCount %u > %u", s.lightCount, RT_MAX_LIGHTS);
    if (s.matCount > RT_MAX_MATERIALS) return c->fail(RT_ERR_INVALID, "matCount %u > %u", s.matCount, RT_MAX_MATERIALS);
    c->scene = s;
    c->have_scene = true;
    c->state_version++;
    return RT_OK;
}

int rt_resize(rt_ctx* ctx, uint32_t width, uint32_t height, const float ratio[2]) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (width == 0 || height == 0 || width > 16384 || height > 16384) return c->fail(RT_ERR_INVALID, "view %ux%u out of range", width, height);
    if (int rc = bind(c)) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    rt::frames_free(c);  // slots are sized for the old view
    free_frame(c);
    c->width = c->height = 0;
    const uint32_t count = level_count_for(width);
    for (uint32_t i = 0; i < count; i++) level_dims_for(width, height, count, i, &c->dims[i][0], &c->dims[i][1]);
    c->level_count = count;
    if (int rc = ensure_levels(c, 1)) {
        free_frame(c);
        c->level_count = 0;
        return rc;
    }
    if (hipMalloc((void**)&c->d_rgb, (size_t)width * height * 3 * sizeof(float)) != hipSuccess) {
        free_frame(c);
        return c->fail(RT_ERR_OOM, "frame buffer");
    }
    c->level_count = count;
    c->width = width;
    c->height = height;
    if (ratio) {
        c->ratio[0] = ratio[0];
        c->ratio[1] = ratio[1];
    } else {  // src/main.rs:364,610  ratio = [FOV, FOV * h / w], FOV = 1
        c->ratio[0] = 1.0f;
        c->ratio[1] = 1.0f * (float)height / (float)width;
    }
    c->part.tiles_x = (width + RT_TILE - 1) / RT_TILE;
    c->part.tiles_y = (height + RT_TILE - 1) / RT_TILE;
    c->stats = rt_stats{};
    c->stats.width = width;
    c->stats.height = height;
    c->stats.level_count = count;
    return RT_OK;
}

int rt_level_info(const rt_ctx* ctx, uint32_t* count, uint32_t dims[RT_MAX_LEVELS][2]) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    if (!c || !c->width) return RT_ERR_STATE;
    if (count) *count = c->level_count;
    if (dims) std::memcpy(dims, c->dims, sizeof c->dims);
    return RT_OK;
}

int rt_set_partition(rt_ctx* ctx, uint32_t rank, uint32_t n_ranks) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (n_ranks == 0 || rank >= n_ranks) return c->fail(RT_ERR_INVALID, "rank %u of %u", rank, n_ranks);
    c->part.rank = rank;
    c->part.n_ranks = n_ranks;
    return RT_OK;
}

int rt_tile_info(const rt_ctx* ctx, uint32_t* tiles_x, uint32_t* tiles_y, uint32_t* owned) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    if (!c || !c->width) return RT_ERR_STATE;
    if (tiles_x) *tiles_x = c->part.tiles_x;
    if (tiles_y) *tiles_y = c->part.tiles_y;
    if (owned) *owned = owned_tiles(c->part);
    return RT_OK;
}

int rt_set_stream(rt_ctx* ctx, void* hip_stream) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (int rc = bind(c)) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : c->own_stream;
    return RT_OK;
}

int rt_render_spp(rt_ctx* ctx, const float rot[4], const float pos[3], uint32_t spp, float* rgb_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (c->part.n_ranks > 1) {
        // a partitioned context only fills its own tiles; clear the rest so the frame is defined
        if (int rc = bind(c)) return rc;
        if (c->d_rgb) RT_HIP(c, hipMemsetAsync(c->d_rgb, 0, (size_t)c->width * c->height * 3 * sizeof(float), c->stream));
    }
    int rc = render_common(c, rot, pos, spp, c->d_rgb, 0, true);
    if (rc) return rc;
    if (rgb_out) RT_HIP(c, hipMemcpy(rgb_out, c->d_rgb, (size_t)c->width * c->height * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render(rt_ctx* ctx, const float rot[4], const float pos[3], float* rgb_out, float* depth_out) {
    int rc = rt_render_spp(ctx, rot, pos, 1, rgb_out);
    if (rc) return rc;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (depth_out) {
        const uint32_t l = c->level_count - 1;
        RT_HIP(c, hipMemcpy(depth_out, c->d_level[l] + (size_t)c->last_image * c->dims[l][0] * c->dims[l][1], (size_t)c->dims[l][0] * c->dims[l][1] * sizeof(float), hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_render_device(rt_ctx* ctx, const float rot[4], const float pos[3], uint32_t spp, void* rgb_dev, int tile_major) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!rgb_dev) return c->fail(RT_ERR_INVALID, "rgb_dev is NULL");
    return render_common(c, rot, pos, spp, static_cast<float*>(rgb_dev), tile_major, false);
}

int rt_detile_device(rt_ctx* ctx, const void* tiles_dev, uint32_t n_ranks, uint32_t tiles_per_rank, void* rgb_dev) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!c->width) return c->fail(RT_ERR_STATE, "rt_resize has not been called");
    if (!tiles_dev || !rgb_dev || n_ranks == 0) return c->fail(RT_ERR_INVALID, "NULL buffer or n_ranks = 0");
    const uint32_t total = c->part.tiles_x * c->part.tiles_y;
    if ((uint64_t)tiles_per_rank * n_ranks < total) return c->fail(RT_ERR_INVALID, "%u ranks x %u tiles < %u tiles", n_ranks, tiles_per_rank, total);
    if (int rc = bind(c)) return rc;
    return rt::launch_detile(c, static_cast<const float*>(tiles_dev), n_ranks, tiles_per_rank, static_cast<float*>(rgb_dev));
}

int rt_synchronize(rt_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (int rc = bind(c)) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->frame_valid && c->ev_end) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) c->stats.ms_total = ms;
    }
    return RT_OK;
}

int rt_read_level(rt_ctx* ctx, uint32_t level, float* out, uint32_t* w, uint32_t* h) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!c->frame_valid) return c->fail(RT_ERR_STATE, "no frame rendered yet");
    if (level >= c->level_count) return c->fail(RT_ERR_INVALID, "level %u >= %u", level, c->level_count);
    if (int rc = bind(c)) return rc;
    if (w) *w = c->dims[level][0];
    if (h) *h = c->dims[level][1];
    if (out) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        RT_HIP(c, hipMemcpy(out, c->d_level[level] + (size_t)c->last_image * c->dims[level][0] * c->dims[level][1], (size_t)c->dims[level][0] * c->dims[level][1] * sizeof(float),
                            hipMemcpyDeviceToHost));
    }
    return RT_OK;
}

int rt_read_rgba8(rt_ctx* ctx, uint8_t* rgba_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!rgba_out) return c->fail(RT_ERR_INVALID, "rgba_out is NULL");
    if (!c->frame_valid) return c->fail(RT_ERR_STATE, "no frame rendered yet");
    if (int rc = bind(c)) return rc;
    const uint64_t n = (uint64_t)c->width * c->height;
    // staging buffer of the view's size, kept until rt_resize / rt_destroy: this call sits in a per-frame loop
    if (!c->d_rgba8 && hipMalloc((void**)&c->d_rgba8, n * 4) != hipSuccess) {
        c->d_rgba8 = nullptr;
        return c->fail(RT_ERR_OOM, "rgba8 staging buffer");
    }
    if (int rc = rt::launch_to_rgba8(c, c->d_rgb, c->d_rgba8, n)) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(rgba_out, c->d_rgba8, n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_selftest_math(rt_ctx* ctx, uint64_t* mismatches) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!mismatches) return c->fail(RT_ERR_INVALID, "mismatches is NULL");
    if (int rc = bind(c)) return rc;
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, sizeof *d) != hipSuccess) return c->fail(RT_ERR_OOM, "self-test counter");
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, c->stream);
    int rc = e == hipSuccess ? rt::launch_selftest_sqrt(c, d) : RT_OK;
    unsigned long long h = 0;
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (!rc && e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) return c->fail(RT_ERR_HIP, "math self-test: %s", hipGetErrorString(e));
    *mismatches = h;
    return RT_OK;
}

int rt_get_stats(const rt_ctx* ctx, rt_stats* stats) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    if (!c || !stats) return RT_ERR_INVALID;
    *stats = c->stats;
    return RT_OK;
}

}  // extern "C"
