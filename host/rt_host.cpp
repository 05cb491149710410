// rt_host.cpp — native host harness over the C ABI (include/rt_abi.h).
//
// Stands in for the reference's Rust `main` (src/main.rs), which cannot be built here (no Rust
// toolchain): scene setup (:524-591), camera state -> push constants (:402-414, :770-773), the resize
// rule (:698-709), one frame per "loop iteration" and — promised by the north star, absent in the
// reference — image write-out (PPM from the UNORM8 view a *_UNORM swapchain would hold, PFM for f32).
// Everything GPU-side goes through librt_amd.so; this file contains no kernels and no fallbacks.
//
//   rt_host [--size WxH] [--yaw R] [--pitch R] [--pos x,y,z] [--move right,forward,up] [--spp N]
//           [--scene default|soup:N] [--two-level] [--bounces N] [--seed N] [--frames N] [--out file.ppm|file.pfm]
//           [--march 1|2|3] [--repeat x,y,z] [--mirror N[,reflectivity]] [--transmit N[,transparency[,index]]] [--inflight K]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_abi.h"

namespace {

struct Quat { float x, y, z, w; };

// glam 0.21.3 Quat::from_rotation_z(-yaw) * Quat::from_rotation_x(pitch)  (src/main.rs:402-404)
Quat camera_quat(float yaw, float pitch) {
    const float hz = -yaw * 0.5f, hx = pitch * 0.5f;
    const float zs = std::sin(hz), zc = std::cos(hz), xs = std::sin(hx), xc = std::cos(hx);
    return Quat{zc * xs, zs * xs, zs * xc, zc * xc};
}

// glam Quat::mul_vec3
void rotate(const Quat& q, const float v[3], float out[3]) {
    const float b[3] = {q.x, q.y, q.z};
    const float d = b[0] * v[0] + b[1] * v[1] + b[2] * v[2];
    const float bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    const float c[3] = {b[1] * v[2] - b[2] * v[1], b[2] * v[0] - b[0] * v[2], b[0] * v[1] - b[1] * v[0]};
    for (int i = 0; i < 3; i++) out[i] = 2.0f * d * b[i] + (q.w * q.w - bb) * v[i] + 2.0f * q.w * c[i];
}

uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// same generator as raytracing_engine_amd/scenes.py soup_scene (counter hash, BASELINE configs[2..4])
void soup_scene(uint32_t n, uint32_t seed, float edge, std::vector<float>& verts, std::vector<float>& albedo, std::vector<float>& emission) {
    verts.assign((size_t)n * 9, 0.0f);
    albedo.assign((size_t)n * 3, 0.0f);
    emission.assign((size_t)n * 3, 0.0f);
    auto u = [&](uint32_t stream, uint32_t i) {
        const uint32_t base = hash32(seed * 0x9E3779B9u + stream);
        return (float)(hash32(i + base) >> 8) * 0x1p-24f;
    };
    for (uint32_t i = 0; i < n; i++) {
        const float v0[3] = {u(0, i) * 20 - 10, u(1, i) * 20 + 5, u(2, i) * 20 - 10};
        for (int a = 0; a < 3; a++) {
            const float e1 = (u(3 + a, i) * 2 - 1) * edge, e2 = (u(6 + a, i) * 2 - 1) * edge;
            verts[(size_t)i * 9 + a] = v0[a];
            verts[(size_t)i * 9 + 3 + a] = v0[a] + e1;
            verts[(size_t)i * 9 + 6 + a] = v0[a] + e2;
            albedo[(size_t)i * 3 + a] = u(9 + a, i) * 0.7f + 0.2f;
        }
    }
    const float q[4][3] = {{-4, 11, 12}, {4, 11, 12}, {4, 19, 12}, {-4, 19, 12}};
    const int idx[2][3] = {{0, 1, 2}, {0, 2, 3}};
    for (int t = 0; t < 2; t++) {
        const size_t i = n - 2 + t;
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) verts[i * 9 + k * 3 + a] = q[idx[t][k]][a];
        for (int a = 0; a < 3; a++) {
            albedo[i * 3 + a] = 0.0f;
            emission[i * 3 + a] = 30.0f;
        }
    }
}

bool write_ppm(const char* path, const uint8_t* rgba, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (uint32_t y = 0; y < h; y++)  // row 0 of the frame is the bottom of the image (+Z is up)
        for (uint32_t x = 0; x < w; x++) std::fwrite(rgba + ((size_t)(h - 1 - y) * w + x) * 4, 1, 3, f);
    return std::fclose(f) == 0;
}

bool write_pfm(const char* path, const float* rgb, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", w, h);  // PFM stores rows bottom-up: frame row 0 first
    std::fwrite(rgb, sizeof(float), (size_t)w * h * 3, f);
    return std::fclose(f) == 0;
}

int fail(rt_ctx* ctx, const char* what, int rc) {
    std::fprintf(stderr, "rt_host: %s failed (%d): %s\n", what, rc, rt_last_error(ctx));
    if (ctx) rt_destroy(ctx);
    return 1;
}

}  // namespace

int main(int argc, char** argv) {
    uint32_t w = 1024, h = 768, spp = 1, bounces = 1, seed = 1, frames = 1;
    float yaw = 0.0f, pitch = 0.0f, pos[3] = {0, 0, 0}, move[3] = {0, 0, 0};
    std::string scene = "default", out = "frame.ppm";
    uint32_t march = 0, inflight = 0, mirror = 0;
    bool two_level = false;  // soup scenes: top-level BVH over 64 bottom-level chunks (rt_set_mesh_ex)
    float repeat[3] = {0, 0, 0}, reflectivity = 0.5f, transparency = 0.5f, refraction_index = 1.0f;
    unsigned transmit = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--size") std::sscanf(next(), "%ux%u", &w, &h);
        else if (a == "--yaw") yaw = (float)std::atof(next());
        else if (a == "--pitch") pitch = (float)std::atof(next());
        else if (a == "--pos") std::sscanf(next(), "%f,%f,%f", &pos[0], &pos[1], &pos[2]);
        else if (a == "--move") std::sscanf(next(), "%f,%f,%f", &move[0], &move[1], &move[2]);
        else if (a == "--spp") spp = (uint32_t)std::atoi(next());
        else if (a == "--bounces") bounces = (uint32_t)std::atoi(next());
        else if (a == "--seed") seed = (uint32_t)std::atoi(next());
        else if (a == "--frames") frames = (uint32_t)std::atoi(next());
        else if (a == "--scene") scene = next();
        else if (a == "--out") out = next();
        else if (a == "--march") march = (uint32_t)std::atoi(next());
        else if (a == "--repeat") std::sscanf(next(), "%f,%f,%f", &repeat[0], &repeat[1], &repeat[2]);
        else if (a == "--mirror") std::sscanf(next(), "%u,%f", &mirror, &reflectivity);
        else if (a == "--transmit") std::sscanf(next(), "%u,%f,%f", &transmit, &transparency, &refraction_index);
        else if (a == "--two-level") two_level = true;
        else if (a == "--inflight") inflight = (uint32_t)std::atoi(next());
        else {
            std::fprintf(stderr, "usage: rt_host [--size WxH] [--yaw R] [--pitch R] [--pos x,y,z] [--move r,f,u] [--spp N] "
                                 "[--scene default|soup:N] [--two-level] [--bounces N] [--seed N] [--frames N] [--out file.ppm|file.pfm] "
                                 "[--march 1|2|3] [--repeat x,y,z] [--mirror N[,reflectivity]] [--transmit N[,transparency[,index]]] [--inflight K]\n");
            return 2;
        }
    }
    // "the shaders are based on the assumption that width is less than height" (src/main.rs:702-706):
    // a window narrower than it is tall is squared up
    if (w < h) h = w;
    // pitch clamp (src/main.rs:770) and Data::position (:406-414): move along the rotated local axes
    const float half_pi = 1.57079632679f;
    pitch = std::fmin(std::fmax(pitch, -half_pi), half_pi);
    const Quat q = camera_quat(yaw, pitch);
    const float axes[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // RIGHT, FORWARD, UP (src/main.rs:350-357)
    for (int k = 0; k < 3; k++) {
        float r[3];
        rotate(q, axes[k], r);
        for (int a = 0; a < 3; a++) pos[a] += move[k] * r[a];
    }
    const float rot[4] = {q.x, q.y, q.z, q.w};

    rt_ctx* ctx = nullptr;
    int rc = rt_create(&ctx, 0);
    if (rc) return fail(nullptr, "rt_create", rc);
    if ((rc = rt_resize(ctx, w, h, nullptr))) return fail(ctx, "rt_resize", rc);

    std::vector<float> rgb((size_t)w * h * 3);
    const bool tri = scene.rfind("soup:", 0) == 0;
    rt_pt_params prm;
    rt_default_pt_params(&prm);
    if (tri) {
        const uint32_t n = (uint32_t)std::atoi(scene.c_str() + 5);
        std::vector<float> verts, albedo, emission;
        soup_scene(n < 3 ? 3 : n, 1, n >= 500000 ? 0.08f : 0.25f, verts, albedo, emission);
        const rt_mesh_options opt{two_level ? 2u : 1u, 0u};
        if ((rc = rt_set_mesh_ex(ctx, verts.data(), albedo.data(), emission.data(), n < 3 ? 3 : n, &opt))) return fail(ctx, "rt_set_mesh_ex", rc);
        prm.spp = spp;
        prm.bounces = bounces;
        prm.seed = seed;
        prm.sky[0] = prm.sky[1] = 0.2f;
        prm.sky[2] = 0.25f;
    } else {
        rt_mutable_data s;
        rt_default_scene(&s);  // src/main.rs:524-591
        if ((rc = rt_set_scene(ctx, &s, sizeof s))) return fail(ctx, "rt_set_scene", rc);
        if (march || mirror || transmit || repeat[0] > 0 || repeat[1] > 0 || repeat[2] > 0) {  // the march loops / repeat() / reflections the author sketched
            rt_config cfg;
            rt_default_config(&cfg);
            cfg.march_algorithm = march;
            for (int a = 0; a < 3; a++) cfg.repeat[a] = repeat[a];
            cfg.reflections = mirror;
            cfg.reflectivity = reflectivity;
            cfg.transmissions = transmit;  // fragment.glsl:124 / :126: transparency (index 1) or refraction (index > 1)
            cfg.transparency = transparency;
            cfg.refraction_index = refraction_index;
            if ((rc = rt_set_config(ctx, &cfg))) return fail(ctx, "rt_set_config", rc);
        }
    }
    if (inflight && !tri) {
        // the reference's frame loop with its fences (src/main.rs:664-667, 882-927): K swapchain-image slots,
        // frame f goes to slot f % K and is collected K-1 submissions later; host-visible frames per second
        if ((rc = rt_frames_configure(ctx, inflight, RT_FRAME_RGBA8))) return fail(ctx, "rt_frames_configure", rc);
        const void* px = nullptr;
        size_t nbytes = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t f = 0; f < frames; f++) {
            if ((rc = rt_frame_submit(ctx, f % inflight, rot, pos, spp))) return fail(ctx, "rt_frame_submit", rc);
            if (f + 1 >= inflight && (rc = rt_frame_wait(ctx, (f + 1 - inflight) % inflight, &px, &nbytes))) return fail(ctx, "rt_frame_wait", rc);
        }
        for (uint32_t f = frames > inflight - 1 ? frames - (inflight - 1) : 0; f < frames; f++)
            if ((rc = rt_frame_wait(ctx, f % inflight, &px, &nbytes))) return fail(ctx, "rt_frame_wait", rc);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%u frame(s) %ux%u through %u slots, pixels on the host every frame: %.1f fps\n", frames, w, h, inflight, frames / sec);
        const bool ok = px && write_ppm(out.c_str(), static_cast<const uint8_t*>(px), w, h);
        rt_destroy(ctx);
        if (!ok) {
            std::fprintf(stderr, "rt_host: cannot write %s\n", out.c_str());
            return 1;
        }
        std::printf("wrote %s\n", out.c_str());
        return 0;
    }
    double ms_sum = 0.0;
    uint64_t rays = 0;
    for (uint32_t f = 0; f < frames; f++) {  // the reference's frame loop (src/main.rs:721-928) minus the window
        if (tri) {
            if ((rc = rt_render_pt(ctx, rot, pos, &prm, rgb.data()))) return fail(ctx, "rt_render_pt", rc);
            rt_pt_stats st;
            rt_get_pt_stats(ctx, &st);
            ms_sum += st.ms_total;
            rays = st.camera_rays + st.bounce_rays + st.shadow_rays;
        } else {
            if ((rc = rt_render_spp(ctx, rot, pos, spp, rgb.data()))) return fail(ctx, "rt_render_spp", rc);
            rt_stats st;
            rt_get_stats(ctx, &st);
            ms_sum += st.ms_total;
            rays = st.primary_rays + st.shadow_rays;
        }
    }
    std::printf("%u frame(s) %ux%u, %.3f ms/frame on device, %.1f Mrays/s, %.1f fps\n", frames, w, h, ms_sum / frames,
                (double)rays / (ms_sum / frames) / 1e3, 1e3 / (ms_sum / frames));  // the reference prints FPS (src/main.rs:730)
    bool ok;
    if (out.size() > 4 && out.substr(out.size() - 4) == ".pfm") {
        ok = write_pfm(out.c_str(), rgb.data(), w, h);
    } else {
        std::vector<uint8_t> rgba((size_t)w * h * 4);
        if (tri) {  // rt_read_rgba8 converts the context's frame buffer, which rt_render_pt also fills
            if ((rc = rt_read_rgba8(ctx, rgba.data()))) return fail(ctx, "rt_read_rgba8", rc);
        } else if ((rc = rt_read_rgba8(ctx, rgba.data()))) return fail(ctx, "rt_read_rgba8", rc);
        ok = write_ppm(out.c_str(), rgba.data(), w, h);
    }
    rt_destroy(ctx);
    if (!ok) {
        std::fprintf(stderr, "rt_host: cannot write %s\n", out.c_str());
        return 1;
    }
    std::printf("wrote %s\n", out.c_str());
    return 0;
}
