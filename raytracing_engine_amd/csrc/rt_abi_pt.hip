// rt_abi_pt.hip — C-ABI entry points of path B (triangle mesh + BVH + wavefront path tracer) and
// the host-side stage schedule.  No reference counterpart (include/rt_abi.h, "Path B").
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <system_error>
#include <vector>

#include "bvh_build.h"
#include "rt_internal.h"
#include "rt_roctx.h"

using rt::Ctx;
using rt::PtData;

namespace {

constexpr uint64_t kMaxPathsInFlight = 1ull << 25;  // 33.5 M paths = 4.3 GB of wavefront state
constexpr uint32_t kMaxBounces = 15;
constexpr size_t kStatWords = 16;  // device-side traversal statistics (path_b.hip)
constexpr uint32_t kDefaultPacketMode = rt::PACKET_INTERVAL_ONLY;
constexpr uint32_t kDefaultTriMode = rt::TRI_MODE_INLINE;  // rt_pt_params.tune_tri_mode = 0
constexpr float kCameraReach = 32.0f;  // camera |coordinate| limit in units of the mesh's largest |coordinate| (render_pt_common)

int bind(Ctx* c) {
    RT_HIP(c, hipSetDevice(c->device));
    return RT_OK;
}

template <class T>
void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void free_wavefront(PtData& pt) {
    dfree(pt.st.ray_o);
    dfree(pt.st.ray_d);
    dfree(pt.st.thr);
    dfree(pt.st.rad);
    dfree(pt.st.hit);
    dfree(pt.st.sh_o);
    dfree(pt.st.sh_d);
    dfree(pt.st.sh_c);
    dfree(pt.d_queue[0]);
    dfree(pt.d_queue[1]);
    dfree(pt.d_acc);
    pt.cap_paths = 0;
    pt.cap_slots = 0;
}

void free_mesh(PtData& pt) {
    if (pt.borrowed_mesh) {  // another context owns the arrays
        pt.d_nodes = pt.d_tris = pt.d_albedo = pt.d_emission = nullptr;
        pt.d_lights = nullptr;
        pt.borrowed_mesh = false;
    }
    dfree(pt.d_nodes);
    dfree(pt.d_tris);
    dfree(pt.d_albedo);
    dfree(pt.d_emission);
    dfree(pt.d_lights);
    dfree(pt.d_spill);
    pt.spill_words = 0;
    pt.host.reset();
    pt.cap_nodes = 0;
    pt.n_tris = pt.n_nodes = pt.n_lights = 0;
    pt.stats = rt_pt_stats{};
}

template <class T>
bool dalloc(T*& p, size_t count) {
    return hipMalloc((void**)&p, count * sizeof(T)) == hipSuccess;
}

int ensure_wavefront(Ctx* c, uint64_t n_paths, uint64_t n_slots) {
    PtData& pt = c->pt;
    if (!pt.d_ctr) {
        if (!dalloc(pt.d_ctr, (size_t)rt::PT_CTR_STRIDE * (kMaxBounces + 3)) || !dalloc(pt.d_stats, kStatWords)) return c->fail(RT_ERR_OOM, "path-tracer counters");
    }
    if (n_paths > pt.cap_paths || n_slots > pt.cap_slots) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        free_wavefront(pt);
        const size_t n = (size_t)n_paths;
        const bool ok = dalloc(pt.st.ray_o, n) && dalloc(pt.st.ray_d, n) && dalloc(pt.st.thr, n) && dalloc(pt.st.rad, n) && dalloc(pt.st.hit, n) &&
                        dalloc(pt.st.sh_o, n) && dalloc(pt.st.sh_d, n) && dalloc(pt.st.sh_c, n) && dalloc(pt.d_queue[0], n) &&
                        dalloc(pt.d_queue[1], n) && dalloc(pt.d_acc, (size_t)n_slots * 3);
        if (!ok) {
            free_wavefront(pt);
            return c->fail(RT_ERR_OOM, "wavefront buffers for %llu paths", (unsigned long long)n_paths);
        }
        pt.cap_paths = n_paths;
        pt.cap_slots = n_slots;
    }
    return RT_OK;
}

// Traversal stack + persistent grid.  The first lds_cap entries of every lane's stack live in LDS
// (8-byte entries: 2 x lds_cap KiB per 256-thread workgroup, next to the kernel's 2 KiB octant table), which bounds residency at floor(160 KiB / that) workgroups
// per CU, 8 at most (32 waves per CU); the rest of the builder's worst case spills to global memory.
int stack_config(Ctx* c, uint32_t tune_lds, uint32_t tune_blocks, uint64_t n_paths, rt::StackCfg* sk, uint32_t* grid, uint32_t extra_lds_bytes = 0) {
    PtData& pt = c->pt;
    const uint32_t need = std::max<uint32_t>(pt.stack_need, 1u);
    // default: up to ten entries in LDS - the whole stack of the 1 M-triangle tree (depth 9) - at seven workgroups per CU
    // (measured 1 % ahead of eight entries at eight workgroups)
    const uint32_t lds_cap = std::min<uint32_t>(need, tune_lds ? std::min<uint32_t>(tune_lds, 78u) : 10u);
    const uint32_t fit = std::max<uint32_t>(1u, std::min<uint32_t>(8u, (160u * 1024u) / (2048u * lds_cap + 2048u + extra_lds_bytes)));  // 2 KiB per entry per workgroup + the 2 KiB octant table (+ the triangle pools)
    // Few paths (a rank's small share of a frame): fewer resident waves.  Every lane of the grid takes a ray
    // at once, so with ~2 rays per lane the rays in flight span half the frame instead of a compact window
    // and the short launches are all ramp and tail; about four rays per lane and more measured best
    // (1/8 of the headline frame: 2.87 -> 2.62 ms with 4 instead of 8 workgroups per CU).
    const uint32_t by_load = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(3, (n_paths + (uint64_t)c->n_cus * 1024 - 1) / ((uint64_t)c->n_cus * 1024)));
    const uint32_t blocks_per_cu = tune_blocks ? std::min<uint32_t>(tune_blocks, fit) : std::min(fit, by_load);
    *grid = (uint32_t)c->n_cus * blocks_per_cu;
    sk->lds_cap = (int)lds_cap;
    sk->spill_cap = (int)(need - lds_cap);
    sk->spill_stride = (size_t)c->n_cus * 8u * 256u;  // covers every grid this function can return
    const size_t half = std::max<size_t>(1, (size_t)sk->spill_cap) * sk->spill_stride;
    const size_t words = 2 * half;  // second half: the shadow kernel when it overlaps the next closest-hit kernel
    if (words > pt.spill_words) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        dfree(pt.d_spill);
        pt.spill_words = 0;
        if (!dalloc(pt.d_spill, words)) return c->fail(RT_ERR_OOM, "traversal spill stack (%zu words)", words);
        pt.spill_words = words;
    }
    sk->spill = pt.d_spill;
    pt.spill_half = half;
    return RT_OK;
}

rt::PtScene scene_view(const PtData& pt) {
    rt::PtScene s{};
    s.nodes = pt.d_nodes;
    s.tris = pt.d_tris;
    s.albedo = pt.d_albedo;
    s.emission = pt.d_emission;
    s.lights = pt.d_lights;
    s.n_lights = pt.n_lights;
    s.n_tris = pt.n_tris;
    return s;
}

uint32_t owned_tiles(const rt::Partition& p) {
    const uint32_t total = p.tiles_x * p.tiles_y;
    return total > p.rank ? (total - p.rank + p.n_ranks - 1u) / p.n_ranks : 0u;
}

struct StageTimer {  // HIP-event pairs around launches, summed per stage after the frame
    Ctx* c;
    bool on;
    std::vector<hipEvent_t>& pool;  // the context's own events: created after hipSetDevice(c->device), freed by pt_free
    std::vector<std::pair<int, size_t>> marks;  // stage, index of begin event
    size_t used = 0;
    hipError_t err = hipSuccess;  // first failure of an event call (reported by the frame)
    hipEvent_t next() {
        if (used == pool.size()) {
            hipEvent_t e;
            const hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) {
                if (err == hipSuccess) err = rc;
                return nullptr;
            }
            pool.push_back(e);
        }
        return pool[used++];
    }
    void record() {
        hipEvent_t e = next();
        if (!e) return;
        const hipError_t rc = hipEventRecord(e, c->stream);
        if (rc != hipSuccess && err == hipSuccess) err = rc;
    }
    void begin(int stage) {
        if (!on) return;
        marks.push_back({stage, used});
        record();
    }
    void end() {
        if (on) record();
    }
};

int render_pt_common(Ctx* c, const float rot[4], const float pos[3], const rt_pt_params* prm, float* dst_dev, int tile_major, bool sync) {
    if (!c) return RT_ERR_INVALID;
    if (!rot || !pos || !prm) return c->fail(RT_ERR_INVALID, "rot/pos/params must not be NULL");
    if (!c->pt.n_tris) return c->fail(RT_ERR_STATE, "rt_set_mesh has not been called");
    if (!c->width) return c->fail(RT_ERR_STATE, "rt_resize has not been called");
    if (prm->spp == 0 || prm->bounces > kMaxBounces) return c->fail(RT_ERR_INVALID, "spp %u / bounces %u out of range", prm->spp, prm->bounces);
    {
        // The slab test's rounding error is about 2^-22 * (|origin| + |plane|) in position space and the boxes are
        // padded by 2e-5 * M (M = largest |vertex coordinate|, at least 1): the invariant "results do not depend on
        // which boxes are visited"  holds for ray origins within about 40 M.  Bounce and shadow
        // rays start on the mesh; the camera is checked here.
        const float reach = kCameraReach * c->pt.bvh_maxabs;  // stored at build time (pad / 2e-5f does not round-trip in fp32)
        for (int a = 0; a < 3; a++)
            if (!(std::fabs(pos[a]) <= reach))
                return c->fail(RT_ERR_INVALID, "camera position %g is outside +-%g (= %g x the mesh's largest |coordinate|): beyond the range the BVH box padding covers",
                               (double)pos[a], (double)reach, (double)kCameraReach);
    }
    if (int rc = bind(c)) return rc;
    PtData& pt = c->pt;

    const uint32_t owned = owned_tiles(c->part);
    const uint64_t n_slots = (uint64_t)owned * RT_TILE * RT_TILE;
    if (n_slots == 0) return RT_OK;
    if (n_slots > kMaxPathsInFlight) return c->fail(RT_ERR_INVALID, "view too large for one rank");
    const uint64_t max_paths = prm->max_paths ? std::min<uint64_t>(prm->max_paths, kMaxPathsInFlight) : kMaxPathsInFlight;
    const uint32_t spp_batch = (uint32_t)std::min<uint64_t>(prm->spp, std::max<uint64_t>(1, max_paths / n_slots));
    if (int rc = ensure_wavefront(c, n_slots * spp_batch, n_slots)) return rc;

    const rt::PtScene sc = scene_view(pt);
    const bool count = prm->count_traversal != 0;
    rt::StackCfg stack_cap{};
    uint32_t grid_persistent = 0;
    // triangle-test schedule of the per-lane kernels: byte 0 = mode (0 = default), bytes 1-2 = the pool's flush parameters
    const uint32_t tri_mode = (prm->tune_tri_mode & 0xffu) ? (prm->tune_tri_mode & 0xffu) : kDefaultTriMode;
    if (tri_mode < rt::TRI_MODE_INLINE || tri_mode > rt::TRI_MODE_INLINE_PF) return c->fail(RT_ERR_INVALID, "tune_tri_mode %u (0 .. 4)", tri_mode);
    const uint32_t tri_cfg = prm->tune_tri_mode >> 8;
    // low byte: idle lanes that trigger a refill (24; 8 with the software-pipelined refill, whose refill is a read from LDS); next byte (tuning): inner steps per round
    const uint32_t refill_min = (prm->tune_refill_min & 0xffu ? std::min<uint32_t>(prm->tune_refill_min & 0xffu, 64u) : tri_mode == rt::TRI_MODE_INLINE_PF ? 8u : 24u) |
                                (prm->tune_refill_min & 0xff00u);
    if (int rc = stack_config(c, prm->tune_lds_stack, prm->tune_blocks_per_cu, n_slots * spp_batch, &stack_cap, &grid_persistent, rt::pt_pool_lds_bytes(tri_mode))) return rc;
    const uint32_t grid_stride = (uint32_t)c->n_cus * 2u;  // 1024-thread workgroups, grid-stride

    StageTimer tm{c, c->cfg.profile_stages != 0, pt.ev_pool, {}, 0};
    rt::RoctxRange frame_range("rt.path_b.frame");

    RT_HIP(c, hipMemsetAsync(pt.d_stats, 0, kStatWords * sizeof(unsigned long long), c->stream));
    RT_HIP(c, hipEventRecord(c->ev_begin, c->stream));
    uint64_t cam = 0, bnc = 0, shd = 0;
    uint32_t launches_closest = 0, launches_shadow = 0, launches_fused = 0;
    uint64_t shd_fused = 0;  // shadow rays traced inside fused launches
    // How shadow(d) and closest(d + 1) share the machine (they are independent: the shadow rays only add to the paths'
    // radiance, the closest-hit rays only read rays):  0 (default) one persistent launch pulls from both queues
    // (pt_trace_fused);  2 two launches on two streams (the auxiliary stream exists from the first frame that wants it:
    // streams are dealt onto a few hardware queues in creation order, and a context that only renders path A should
    // not occupy two; falls back to 1 under per-stage timing);  1 one launch after the other on one stream.
    const uint32_t overlap_mode = pt.n_lights == 0 ? 1u : (tm.on && prm->tune_no_overlap == 2u) ? 1u : prm->tune_no_overlap;  // per-stage timing needs one stream
    const bool fused = overlap_mode == 0u;
    if (overlap_mode == 2u && !c->aux_stream) RT_HIP(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    const bool overlap = overlap_mode == 2u && c->aux_stream != nullptr;
    bool shadow_pending = false;
    if (overlap && !pt.ev_shaded) {
        RT_HIP(c, hipEventCreateWithFlags(&pt.ev_shaded, hipEventDisableTiming));
        RT_HIP(c, hipEventCreateWithFlags(&pt.ev_shadowed, hipEventDisableTiming));
    }
    std::vector<uint32_t> h_ctr((size_t)rt::PT_CTR_STRIDE * (prm->bounces + 2));

    for (uint32_t s0 = 0; s0 < prm->spp; s0 += spp_batch) {
        const uint32_t nb = std::min(spp_batch, prm->spp - s0);
        rt::PtFrame f{};
        std::memcpy(f.cam.rot, rot, 16);
        std::memcpy(f.cam.pos, pos, 12);
        f.cam.ratio[0] = c->ratio[0];
        f.cam.ratio[1] = c->ratio[1];
        f.width = c->width;
        f.height = c->height;
        f.part = c->part;
        f.n_slots = (uint32_t)n_slots;
        f.spp_batch = nb;
        f.n_paths = (uint32_t)(n_slots * nb);
        f.sample0 = s0;
        f.spp_total = prm->spp;
        f.bounces = prm->bounces;
        f.seed = prm->seed;
        std::memcpy(f.sky, prm->sky, 12);
        f.ray_eps = prm->ray_eps;

        const size_t ctr_words = (size_t)rt::PT_CTR_STRIDE * (prm->bounces + 2);
        RT_HIP(c, hipMemsetAsync(pt.d_ctr, 0, ctr_words * sizeof(uint32_t), c->stream));
        // camera rays through the packet kernel, which makes its own rays: no generate stage, no queue 0.  Its wave-uniform stack is a
        // fixed LDS array: a tree that may need more (a deep two-level tree) takes the per-lane kernel, whose stack is sized from stack_need
        const bool packet = prm->tune_no_packet != rt::PACKET_OFF && pt.stack_need <= rt::kPacketStackEntries;
        if (!packet) {
            rt::RoctxRange rr("rt.path_b.generate");
            tm.begin(0);
            if (int rc = rt::launch_pt_generate(c, f, pt.st, pt.d_queue[0], pt.d_ctr)) return rc;
            tm.end();
        }
        bool shadow_deferred = false;  // fused mode: shadow(d - 1) waits for the launch of closest(d)
        for (uint32_t d = 0; d <= prm->bounces; d++) {
            uint32_t* ctr_d = pt.d_ctr + (size_t)rt::PT_CTR_STRIDE * d;
            uint32_t* ctr_n = pt.d_ctr + (size_t)rt::PT_CTR_STRIDE * (d + 1);
            const uint32_t* q = pt.d_queue[d & 1];
            uint32_t* qn = pt.d_queue[(d + 1) & 1];
            {
            rt::RoctxRange rr(d == 0 && packet ? "rt.path_b.trace_packet depth" : shadow_deferred ? "rt.path_b.trace_fused depth" : "rt.path_b.trace_closest depth", d);
            tm.begin(d == 0 && packet ? 5 : shadow_deferred ? 6 : 1);
            if (d == 0 && packet) {  // camera rays: one shared origin, coherent 4x4-pixel blocks per wave
                if (int rc = rt::launch_pt_trace_packet(c, sc, f, pt.st, pt.d_stats, count, prm->tune_no_packet == rt::PACKET_DEFAULT ? kDefaultPacketMode : prm->tune_no_packet)) return rc;
            } else if (shadow_deferred) {  // closest(d) + shadow(d - 1): ctr_d holds both the closest count of depth d and the shadow count of depth d - 1
                if (int rc = rt::launch_pt_trace_fused(c, sc, pt.st, q, ctr_d + rt::PT_CTR_COUNT, ctr_d + rt::PT_CTR_HEAD_CLOSEST, ctr_d + rt::PT_CTR_SHADOW_COUNT,
                                                       ctr_d + rt::PT_CTR_HEAD_SHADOW, pt.d_stats, count, grid_persistent, stack_cap, refill_min, tri_mode, tri_cfg))
                    return rc;
                shadow_deferred = false;
                launches_shadow++;
                launches_fused++;
            } else if (int rc = rt::launch_pt_trace(c, sc, pt.st, q, ctr_d + rt::PT_CTR_COUNT, ctr_d + rt::PT_CTR_HEAD_CLOSEST, pt.d_stats, false, count, grid_persistent, stack_cap, refill_min, tri_mode, tri_cfg)) {
                return rc;
            }
            tm.end();
            }
            launches_closest++;
            if (shadow_pending) {  // shade(d) adds sky/emission after shadow(d-1)'s contribution
                RT_HIP(c, hipStreamWaitEvent(c->stream, pt.ev_shadowed, 0));
                shadow_pending = false;
            }
            {
                rt::RoctxRange rr("rt.path_b.shade depth", d);
                tm.begin(2);
                if (int rc = rt::launch_pt_shade(c, sc, f, pt.st, d == 0 && packet ? nullptr : q, ctr_d + rt::PT_CTR_COUNT, d, qn, ctr_n, grid_stride, prm->tune_sort_rays)) return rc;
                tm.end();
            }
            if (pt.n_lights) {
                if (fused && d < prm->bounces) {  // goes into the same launch as closest(d + 1)
                    shadow_deferred = true;
                    continue;
                }
                // two-stream mode: the shadow kernel runs on the auxiliary stream beside the next closest-hit kernel;
                // shade(d+1) and resolve read the radiance and therefore wait for it (keeps the per-path sum order)
                rt::StackCfg sk2 = stack_cap;
                hipStream_t main_stream = c->stream;
                if (overlap) {
                    sk2.spill = stack_cap.spill + pt.spill_half;
                    RT_HIP(c, hipEventRecord(pt.ev_shaded, main_stream));
                    RT_HIP(c, hipStreamWaitEvent(c->aux_stream, pt.ev_shaded, 0));
                    c->stream = c->aux_stream;
                }
                int rc;
                {
                    rt::RoctxRange rr("rt.path_b.trace_shadow depth", d);
                    tm.begin(3);
                    rc = rt::launch_pt_trace(c, sc, pt.st, nullptr, ctr_n + rt::PT_CTR_SHADOW_COUNT, ctr_n + rt::PT_CTR_HEAD_SHADOW, pt.d_stats, true, count, grid_persistent, sk2, refill_min, tri_mode, tri_cfg);
                    tm.end();
                }
                if (overlap) {
                    hipError_t e = rc ? hipSuccess : hipEventRecord(pt.ev_shadowed, c->aux_stream);
                    c->stream = main_stream;
                    if (rc) return rc;
                    RT_HIP(c, e);
                    shadow_pending = true;
                } else if (rc) return rc;
                launches_shadow++;
            }
        }
        if (shadow_pending) {
            RT_HIP(c, hipStreamWaitEvent(c->stream, pt.ev_shadowed, 0));
            shadow_pending = false;
        }
        {
            rt::RoctxRange rr("rt.path_b.resolve");
            tm.begin(4);
            if (int rc = rt::launch_pt_resolve(c, f, pt.st, pt.d_acc, dst_dev, tile_major)) return rc;
            tm.end();
        }
        if (sync) {  // ray counters of this batch (the copy is ordered after the kernels on the stream)
            RT_HIP(c, hipMemcpyAsync(h_ctr.data(), pt.d_ctr, ctr_words * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            RT_HIP(c, hipStreamSynchronize(c->stream));
            if (packet) {  // no queue 0: camera rays = samples of the owned pixels inside the frame
                uint64_t px_owned = 0;
                for (uint32_t t = c->part.rank; t < c->part.tiles_x * c->part.tiles_y; t += c->part.n_ranks) {
                    const uint32_t ty = t / c->part.tiles_x, tx = t % c->part.tiles_x;
                    px_owned += (uint64_t)std::min<uint32_t>(RT_TILE, c->width - tx * RT_TILE) * std::min<uint32_t>(RT_TILE, c->height - ty * RT_TILE);
                }
                cam += px_owned * nb;
            } else {
                cam += h_ctr[rt::PT_CTR_COUNT];
            }
            for (uint32_t d = 1; d <= prm->bounces + 1; d++) {
                if (d <= prm->bounces) bnc += h_ctr[(size_t)rt::PT_CTR_STRIDE * d + rt::PT_CTR_COUNT];
                shd += h_ctr[(size_t)rt::PT_CTR_STRIDE * d + rt::PT_CTR_SHADOW_COUNT];
                if (fused && d <= prm->bounces) shd_fused += h_ctr[(size_t)rt::PT_CTR_STRIDE * d + rt::PT_CTR_SHADOW_COUNT];  // shadow(d - 1) rides with closest(d)
            }
        }
    }
    RT_HIP(c, hipEventRecord(c->ev_end, c->stream));
    if (tm.err != hipSuccess) return c->fail(RT_ERR_HIP, "stage-timing event: %s", hipGetErrorString(tm.err));
    c->frame_valid = true;
    pt.stats.launches_trace_closest = launches_closest;
    pt.stats.launches_trace_shadow = launches_shadow;
    if (sync) {
        RT_HIP(c, hipStreamSynchronize(c->stream));
        unsigned long long st[kStatWords] = {};
        RT_HIP(c, hipMemcpy(st, pt.d_stats, sizeof st, hipMemcpyDeviceToHost));
        pt.stats.camera_rays = cam;
        pt.stats.bounce_rays = bnc;
        pt.stats.shadow_rays = shd;
        pt.stats.nodes_visited = st[0];
        pt.stats.tris_tested = st[1];
        pt.stats.shadow_nodes_visited = st[4] + st[11];
        pt.stats.shadow_tris_tested = st[5] + st[12];
        pt.stats.fused_shadow_nodes = st[11];
        pt.stats.fused_shadow_tris = st[12];
        pt.stats.packet_nodes_fetched = st[9];
        pt.stats.packet_tris_fetched = st[10];
        pt.stats.fused_shadow_rays = shd_fused;
        pt.stats.launches_trace_fused = launches_fused;
        pt.stats.wave_rounds = st[6];
        pt.stats.alive_lane_rounds = st[7];
        pt.stats.packets = st[8];
        pt.stats.pool_flushes = st[13];
        pt.stats.wave_rounds_all = st[14];
        pt.stats.stack_overflow = (uint32_t)st[2];
        RT_HIP(c, hipEventElapsedTime(&pt.stats.ms_total, c->ev_begin, c->ev_end));
        float sums[7] = {};
        if (tm.on) {
            for (auto& m : tm.marks) {
                float ms = 0.0f;
                if (m.second + 1 < tm.used && hipEventElapsedTime(&ms, pt.ev_pool[m.second], pt.ev_pool[m.second + 1]) == hipSuccess) sums[m.first] += ms;
            }
        }
        pt.stats.ms_generate = sums[0];
        pt.stats.ms_trace_closest = sums[1];
        pt.stats.ms_shade = sums[2];
        pt.stats.ms_trace_shadow = sums[3];
        pt.stats.ms_resolve = sums[4];
        pt.stats.ms_trace_packet = sums[5];
        pt.stats.ms_trace_fused = sums[6];
    }
    return RT_OK;
}

}  // namespace

namespace rt {
void pt_free(Ctx* c) {
    free_wavefront(c->pt);
    free_mesh(c->pt);
    dfree(c->pt.d_ctr);
    dfree(c->pt.d_stats);
    if (c->pt.ev_shaded) (void)hipEventDestroy(c->pt.ev_shaded);
    if (c->pt.ev_shadowed) (void)hipEventDestroy(c->pt.ev_shadowed);
    c->pt.ev_shaded = c->pt.ev_shadowed = nullptr;
    for (hipEvent_t e : c->pt.ev_pool) (void)hipEventDestroy(e);
    c->pt.ev_pool.clear();
}

// `lane` renders with `owner`'s device mesh (read-only during rendering); the owner tells its lanes
// before it frees or replaces the mesh (frames_drop_mesh).
void pt_borrow_mesh(Ctx* lane, const Ctx* owner) {  // owner == nullptr: only forget what was borrowed
    PtData& d = lane->pt;
    free_mesh(d);
    if (!owner || !owner->pt.n_tris) return;
    const PtData& s = owner->pt;
    d.borrowed_mesh = true;
    d.n_tris = s.n_tris;
    d.n_nodes = s.n_nodes;
    d.n_lights = s.n_lights;
    d.bvh_depth = s.bvh_depth;
    d.bvh_build_ms = s.bvh_build_ms;
    d.bvh_pad = s.bvh_pad;
    d.bvh_maxabs = s.bvh_maxabs;
    d.d_nodes = s.d_nodes;
    d.d_tris = s.d_tris;
    d.d_albedo = s.d_albedo;
    d.d_emission = s.d_emission;
    d.d_lights = s.d_lights;
    d.stack_need = s.stack_need;
    d.stats = rt_pt_stats{};
    d.stats.n_tris = s.n_tris;
    d.stats.n_nodes = s.n_nodes;
    d.stats.bvh_depth = s.bvh_depth;
    d.stats.stack_need = s.stack_need;
    d.stats.n_lights = s.n_lights;
    d.stats.bvh_build_ms = s.bvh_build_ms;
}
}  // namespace rt

extern "C" {

int rt_default_pt_params(rt_pt_params* p) {
    if (!p) return RT_ERR_INVALID;
    p->spp = 4;
    p->bounces = 1;
    p->seed = 1;
    p->sky[0] = p->sky[1] = p->sky[2] = 0.0f;
    p->ray_eps = 1e-3f;
    p->count_traversal = 0;
    p->max_paths = 0;
    p->tune_refill_min = 0;
    p->tune_blocks_per_cu = 0;
    p->tune_lds_stack = 0;
    p->tune_no_overlap = 0;
    p->tune_no_packet = 0;
    p->tune_sort_rays = 0;
    p->tune_tri_mode = 0;
    return RT_OK;
}

}  // extern "C"

namespace {

// leaf-order records [li0, li1) of the device triangle / material arrays from the host copies
void pack_leaf_range(const rt::BvhResult& bvh, const float* v0, const float* e1, const float* e2, const float* albedo, const float* emission, size_t li0,
                     size_t li1, float* tris, float* alb, float* emi) {
    for (size_t li = li0; li < li1; li++) {
        const uint32_t t = bvh.order[li];
        float* r = &tris[12 * (li - li0)];
        r[0] = v0[3 * (size_t)t]; r[1] = v0[3 * (size_t)t + 1]; r[2] = v0[3 * (size_t)t + 2]; r[3] = e1[3 * (size_t)t];
        r[4] = e1[3 * (size_t)t + 1]; r[5] = e1[3 * (size_t)t + 2]; r[6] = e2[3 * (size_t)t]; r[7] = e2[3 * (size_t)t + 1];
        r[8] = e2[3 * (size_t)t + 2];
        std::memcpy(&r[9], &t, 4);
        // word 10: 1 = emissive.  pt_shade reads the record anyway (normal) and skips the 16-byte emission gather for the
        // triangles that are not lights - all but a handful
        const uint32_t is_light = (emission[3 * (size_t)t] > 0.0f || emission[3 * (size_t)t + 1] > 0.0f || emission[3 * (size_t)t + 2] > 0.0f) ? 1u : 0u;
        std::memcpy(&r[10], &is_light, 4);
        r[11] = 0.0f;
        for (int a = 0; a < 3; a++) {
            alb[4 * (li - li0) + a] = albedo[3 * (size_t)t + a];
            emi[4 * (li - li0) + a] = emission[3 * (size_t)t + a];
        }
        alb[4 * (li - li0) + 3] = emi[4 * (li - li0) + 3] = 0.0f;
    }
}

void publish_bvh_stats(PtData& pt, const rt::BvhResult& bvh) {
    pt.n_nodes = bvh.n_nodes;
    pt.bvh_depth = bvh.depth;
    pt.stack_need = bvh.stack_need;
    pt.bvh_pad = bvh.pad;
    pt.bvh_maxabs = bvh.maxabs;
    pt.stats.n_nodes = bvh.n_nodes;
    pt.stats.bvh_depth = bvh.depth;
    pt.stats.stack_need = bvh.stack_need;
    pt.stats.bvh_build_ms = pt.bvh_build_ms;
    pt.stats.bvh_levels = pt.host ? 2u : 1u;
    pt.stats.blas_chunks = pt.host ? (uint32_t)pt.host->tl.blas.size() : 0u;
    pt.stats.tlas_nodes = pt.host ? pt.host->tl.tlas_nodes : 0u;
}

int set_mesh_impl(Ctx* c, const float* verts, const float* albedo, const float* emission, uint32_t n_tris, const rt_mesh_options* opt) {
    if (!verts || !albedo || !emission) return c->fail(RT_ERR_INVALID, "mesh arrays must not be NULL");
    if (n_tris == 0 || n_tris >= (1u << 28)) return c->fail(RT_ERR_INVALID, "n_tris %u out of [1, 2^28)", n_tris);
    const uint32_t levels = opt ? opt->bvh_levels : 1u, chunks = opt && opt->blas_chunks ? opt->blas_chunks : 64u;
    if (levels != 1u && levels != 2u) return c->fail(RT_ERR_INVALID, "bvh_levels %u (1 or 2)", levels);
    if (chunks > 65536u) return c->fail(RT_ERR_INVALID, "blas_chunks %u > 65536", chunks);
    for (size_t i = 0; i < (size_t)n_tris * 9; i++)
        if (!std::isfinite(verts[i])) return c->fail(RT_ERR_INVALID, "vertex data is not finite at float %zu", i);
    if (int rc = bind(c)) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->aux_stream) RT_HIP(c, hipStreamSynchronize(c->aux_stream));
    rt::frames_drop_mesh(c);  // frame-slot lanes render with this mesh
    PtData& pt = c->pt;
    free_mesh(pt);
    c->state_version++;

    const size_t n = n_tris;
    // spec section 6.1: edges are formed once, in fp32
    std::vector<float> v0(3 * n), e1(3 * n), e2(3 * n);
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            v0[3 * i + a] = verts[9 * i + a];
            e1[3 * i + a] = verts[9 * i + 3 + a] - verts[9 * i + a];
            e2[3 * i + a] = verts[9 * i + 6 + a] - verts[9 * i + a];
        }
    const auto t0 = std::chrono::steady_clock::now();
    rt::BvhResult bvh;
    if (levels == 2u) {
        pt.host.reset(new rt::MeshHost());
        if (!rt::build_bvh_two_level(v0.data(), e1.data(), e2.data(), n_tris, chunks, rt::kBvhMaxDepth, &pt.host->tl, &bvh)) {
            pt.host.reset();
            return c->fail(RT_ERR_INVALID, "two-level BVH build failed");
        }
    } else if (!rt::build_bvh(v0.data(), e1.data(), e2.data(), n_tris, rt::kBvhMaxDepth, &bvh)) {
        return c->fail(RT_ERR_INVALID, "BVH build failed");
    }
    pt.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();

    // leaf-order triangle records + materials; lights in ascending original index
    std::vector<float> tris(12 * n), alb(4 * n), emi(4 * n);
    pack_leaf_range(bvh, v0.data(), e1.data(), e2.data(), albedo, emission, 0, n, tris.data(), alb.data(), emi.data());
    std::vector<uint32_t> leaf_pos(n);
    for (size_t li = 0; li < n; li++) leaf_pos[bvh.order[li]] = (uint32_t)li;
    std::vector<uint32_t> lights, light_ids;
    for (size_t t = 0; t < n; t++)
        if (emission[3 * t] > 0.0f || emission[3 * t + 1] > 0.0f || emission[3 * t + 2] > 0.0f) {
            lights.push_back(leaf_pos[t]);
            light_ids.push_back((uint32_t)t);
        }

    // a two-level mesh keeps room for the node count to move when a chunk is rebuilt
    pt.cap_nodes = levels == 2u ? (size_t)bvh.n_nodes + bvh.n_nodes / 8 + 1024 : bvh.n_nodes;
    const bool ok = dalloc(pt.d_nodes, pt.cap_nodes * 5) && dalloc(pt.d_tris, n * 3) && dalloc(pt.d_albedo, n) && dalloc(pt.d_emission, n) &&
                    dalloc(pt.d_lights, std::max<size_t>(lights.size(), 1));
    if (!ok) {
        free_mesh(pt);
        return c->fail(RT_ERR_OOM, "mesh of %u triangles", n_tris);
    }
    RT_HIP(c, hipMemcpy(pt.d_nodes, bvh.nodes.data(), (size_t)bvh.n_nodes * 80, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(pt.d_tris, tris.data(), n * 48, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(pt.d_albedo, alb.data(), n * 16, hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(pt.d_emission, emi.data(), n * 16, hipMemcpyHostToDevice));
    if (!lights.empty()) RT_HIP(c, hipMemcpy(pt.d_lights, lights.data(), lights.size() * 4, hipMemcpyHostToDevice));
    RT_HIP(c, hipDeviceSynchronize());  // the uploads ran on the null stream; the context's streams are non-blocking and do not wait for it
    pt.n_tris = n_tris;
    pt.n_lights = (uint32_t)lights.size();
    pt.stats = rt_pt_stats{};
    pt.stats.n_tris = n_tris;
    pt.stats.n_lights = pt.n_lights;
    if (pt.host) {  // what a chunk rebuild needs: the mesh in original order and where the lights are
        pt.host->v0.swap(v0);
        pt.host->e1.swap(e1);
        pt.host->e2.swap(e2);
        pt.host->albedo.assign(albedo, albedo + 3 * n);
        pt.host->emission.assign(emission, emission + 3 * n);
        pt.host->light_ids.swap(light_ids);
        pt.stats.ms_build_blas = (float)pt.host->tl.ms_blas;
        pt.stats.ms_build_tlas = (float)pt.host->tl.ms_tlas;
        pt.stats.ms_build_flatten = (float)pt.host->tl.ms_flatten;
    }
    publish_bvh_stats(pt, bvh);
    return RT_OK;
}

int update_chunk_impl(Ctx* c, uint32_t chunk, const float* verts, uint32_t n_tris) {
    PtData& pt = c->pt;
    if (!pt.host || pt.borrowed_mesh) return c->fail(RT_ERR_STATE, "rt_update_mesh_chunk needs a two-level mesh (rt_set_mesh_ex with bvh_levels = 2) owned by this context");
    rt::MeshHost& h = *pt.host;
    if (chunk >= h.tl.blas.size()) return c->fail(RT_ERR_INVALID, "chunk %u of %zu", chunk, h.tl.blas.size());
    if (!verts) return c->fail(RT_ERR_INVALID, "verts is NULL");
    const uint32_t first = h.tl.first[chunk], count = h.tl.first[chunk + 1] - first;
    if (n_tris != count) return c->fail(RT_ERR_INVALID, "chunk %u holds %u triangles, verts holds %u (rt_mesh_chunk_info)", chunk, count, n_tris);
    for (size_t i = 0; i < (size_t)count * 9; i++)
        if (!std::isfinite(verts[i])) return c->fail(RT_ERR_INVALID, "vertex data is not finite at float %zu", i);
    if (int rc = bind(c)) return rc;
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->aux_stream) RT_HIP(c, hipStreamSynchronize(c->aux_stream));
    rt::frames_drop_mesh(c);  // lanes re-borrow the mesh on their next submit
    c->state_version++;
    const auto t0 = std::chrono::steady_clock::now();
    // Transactional: the host copy and the chunk's bottom-level structure change first and are put back if anything up to
    // the device allocation fails; the device arrays are written only after everything host-side (and the node array's
    // regrow) has succeeded.  An upload that fails half-way leaves the device arrays undefined: the mesh is dropped then.
    std::vector<float> old((size_t)count * 9);
    auto swap_in = [&](const float* src, bool edges_formed) {
        for (uint32_t i = 0; i < count; i++) {
            const size_t t = h.tl.sorted[first + i];
            for (int a = 0; a < 3; a++) {
                h.v0[3 * t + a] = src[9 * (size_t)i + a];
                h.e1[3 * t + a] = edges_formed ? src[9 * (size_t)i + 3 + a] : src[9 * (size_t)i + 3 + a] - src[9 * (size_t)i + a];
                h.e2[3 * t + a] = edges_formed ? src[9 * (size_t)i + 6 + a] : src[9 * (size_t)i + 6 + a] - src[9 * (size_t)i + a];
            }
        }
    };
    for (uint32_t i = 0; i < count; i++) {
        const size_t t = h.tl.sorted[first + i];
        for (int a = 0; a < 3; a++) {
            old[9 * (size_t)i + a] = h.v0[3 * t + a];
            old[9 * (size_t)i + 3 + a] = h.e1[3 * t + a];
            old[9 * (size_t)i + 6 + a] = h.e2[3 * t + a];
        }
    }
    swap_in(verts, false);
    rt::BvhResult bvh, displaced;
    const uint32_t n = pt.n_tris;
    bool built = false;
    try {
        built = rt::rebuild_chunk(h.v0.data(), h.e1.data(), h.e2.data(), n, chunk, rt::kBvhMaxDepth, &h.tl, &bvh, &displaced);
    } catch (...) {
        swap_in(old.data(), true);
        throw;  // guarded() turns it into a status; the mesh is as it was
    }
    if (!built) {
        swap_in(old.data(), true);
        return c->fail(RT_ERR_INVALID, "chunk rebuild refused: the moved vertices leave the coordinate range the mesh's box padding was chosen for (call rt_set_mesh_ex again)");
    }
    auto roll_back = [&]() {
        std::swap(h.tl.blas[chunk], displaced);
        swap_in(old.data(), true);
    };
    pt.bvh_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    float4* new_nodes = nullptr;
    size_t new_cap = pt.cap_nodes;
    if (bvh.n_nodes > pt.cap_nodes) {
        new_cap = (size_t)bvh.n_nodes + bvh.n_nodes / 8 + 1024;
        if (!dalloc(new_nodes, new_cap * 5)) {
            roll_back();
            return c->fail(RT_ERR_OOM, "node array of %u nodes (the mesh is unchanged)", bvh.n_nodes);
        }
    }
    // the chunk's triangles keep their range of the leaf order (chunks are laid out in chunk order); inside it the order is new
    size_t li0 = 0;
    for (uint32_t b = 0; b < chunk; b++) li0 += h.tl.blas[b].order.size();
    const size_t li1 = li0 + count;
    std::vector<float> tris, alb, emi;
    std::vector<uint32_t> leaf_of;
    try {
        tris.resize(12 * (size_t)count);
        alb.resize(4 * (size_t)count);
        emi.resize(4 * (size_t)count);
        pack_leaf_range(bvh, h.v0.data(), h.e1.data(), h.e2.data(), h.albedo.data(), h.emission.data(), li0, li1, tris.data(), alb.data(), emi.data());
        if (!h.light_ids.empty()) {  // lights are listed by leaf position, in ascending original index
            bool moved = false;
            for (size_t li = li0; li < li1 && !moved; li++) moved = std::binary_search(h.light_ids.begin(), h.light_ids.end(), bvh.order[li]);
            if (moved) {
                std::vector<uint32_t> leaf_pos(n);
                for (size_t li = 0; li < n; li++) leaf_pos[bvh.order[li]] = (uint32_t)li;
                leaf_of.resize(h.light_ids.size());
                for (size_t k = 0; k < h.light_ids.size(); k++) leaf_of[k] = leaf_pos[h.light_ids[k]];
            }
        }
    } catch (...) {
        dfree(new_nodes);
        roll_back();
        throw;
    }
    // commit to the device
    if (new_nodes) {
        dfree(pt.d_nodes);
        pt.d_nodes = new_nodes;
        pt.cap_nodes = new_cap;
    }
    hipError_t e = hipMemcpy(pt.d_nodes, bvh.nodes.data(), (size_t)bvh.n_nodes * 80, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pt.d_tris + li0 * 3, tris.data(), (size_t)count * 48, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pt.d_albedo + li0, alb.data(), (size_t)count * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pt.d_emission + li0, emi.data(), (size_t)count * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess && !leaf_of.empty()) e = hipMemcpy(pt.d_lights, leaf_of.data(), leaf_of.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipDeviceSynchronize();  // null-stream uploads before anything on the context's non-blocking streams
    if (e != hipSuccess) {
        free_mesh(pt);  // host tree and device arrays may disagree: no frame may be traced against them
        return c->fail(RT_ERR_STATE, "chunk upload failed (%s): the mesh has been dropped, call rt_set_mesh_ex again", hipGetErrorString(e));
    }
    pt.stats.ms_build_blas = (float)h.tl.ms_blas;
    pt.stats.ms_build_tlas = (float)h.tl.ms_tlas;
    pt.stats.ms_build_flatten = (float)h.tl.ms_flatten;
    publish_bvh_stats(pt, bvh);
    return RT_OK;
}

template <class F>
int guarded(Ctx* c, const char* what, F&& f, bool drop_mesh = true) {  // drop_mesh = false: the callee has put the mesh back before it threw
    // the builder allocates host vectors sized by n_tris and starts std::threads: nothing may leave an entry point
    // as a C++ exception (include/rt_abi.h: never throws or aborts across the boundary)
    try {
        return f();
    } catch (const std::bad_alloc&) {
        if (drop_mesh) free_mesh(c->pt);
        return c->fail(RT_ERR_OOM, "%s: out of host memory", what);
    } catch (const std::system_error& e) {
        if (drop_mesh) free_mesh(c->pt);
        return c->fail(RT_ERR_STATE, "%s: %s", what, e.what());
    } catch (const std::exception& e) {
        if (drop_mesh) free_mesh(c->pt);
        return c->fail(RT_ERR_INVALID, "%s: %s", what, e.what());
    } catch (...) {
        if (drop_mesh) free_mesh(c->pt);
        return c->fail(RT_ERR_INVALID, "%s failed", what);
    }
}
}  // namespace

extern "C" {

int rt_set_mesh(rt_ctx* ctx, const float* verts, const float* albedo, const float* emission, uint32_t n_tris) {
    return rt_set_mesh_ex(ctx, verts, albedo, emission, n_tris, nullptr);
}

int rt_set_mesh_ex(rt_ctx* ctx, const float* verts, const float* albedo, const float* emission, uint32_t n_tris, const rt_mesh_options* options) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    return guarded(c, "mesh build", [&] { return set_mesh_impl(c, verts, albedo, emission, n_tris, options); });
}

int rt_mesh_chunk_info(rt_ctx* ctx, uint32_t chunk, uint32_t* count, uint32_t* tri_ids, uint32_t capacity) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!c->pt.host) return c->fail(RT_ERR_STATE, "not a two-level mesh");
    const rt::TwoLevelBvh& tl = c->pt.host->tl;
    if (chunk >= tl.blas.size()) return c->fail(RT_ERR_INVALID, "chunk %u of %zu", chunk, tl.blas.size());
    const uint32_t first = tl.first[chunk], n = tl.first[chunk + 1] - first;
    if (count) *count = n;
    if (tri_ids) {
        if (capacity < n) return c->fail(RT_ERR_INVALID, "tri_ids holds %u of %u triangles", capacity, n);
        std::memcpy(tri_ids, &tl.sorted[first], (size_t)n * 4);
    }
    return RT_OK;
}

int rt_update_mesh_chunk(rt_ctx* ctx, uint32_t chunk, const float* verts, uint32_t n_tris) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    return guarded(c, "chunk rebuild", [&] { return update_chunk_impl(c, chunk, verts, n_tris); }, false);
}

int rt_render_pt(rt_ctx* ctx, const float rot[4], const float pos[3], const rt_pt_params* params, float* rgb_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!c->width) return c->fail(RT_ERR_STATE, "rt_resize has not been called");
    if (int rc = bind(c)) return rc;
    RT_HIP(c, hipMemsetAsync(c->d_rgb, 0, (size_t)c->width * c->height * 3 * sizeof(float), c->stream));
    if (int rc = render_pt_common(c, rot, pos, params, c->d_rgb, 0, true)) return rc;
    if (rgb_out) RT_HIP(c, hipMemcpy(rgb_out, c->d_rgb, (size_t)c->width * c->height * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_render_pt_device(rt_ctx* ctx, const float rot[4], const float pos[3], const rt_pt_params* params, void* rgb_dev, int tile_major) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!rgb_dev) return c->fail(RT_ERR_INVALID, "rgb_dev is NULL");
    return render_pt_common(c, rot, pos, params, static_cast<float*>(rgb_dev), tile_major, false);
}

int rt_get_pt_stats(rt_ctx* ctx, rt_pt_stats* stats) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !stats) return RT_ERR_INVALID;
    *stats = c->pt.stats;
    return RT_OK;
}

int rt_trace_rays_counted(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out, int32_t* tri_out,
                          uint32_t* counts_out) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!origins || !dirs || !t_out || !tri_out) return c->fail(RT_ERR_INVALID, "NULL ray buffer");
    if (!c->pt.n_tris) return c->fail(RT_ERR_STATE, "rt_set_mesh has not been called");
    if (n == 0) return RT_OK;
    if (int rc = bind(c)) return rc;
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int* d_i = nullptr;
    uint32_t* d_c = nullptr;
    const size_t nb = (size_t)n;
    int rc = RT_OK;
    if (!dalloc(d_o, nb * 3) || !dalloc(d_d, nb * 3) || !dalloc(d_t, nb) || !dalloc(d_i, nb) || (counts_out && !dalloc(d_c, nb * 2))) rc = c->fail(RT_ERR_OOM, "ray buffers");
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpy(d_o, origins, nb * 12, hipMemcpyHostToDevice);
    if (!rc && e == hipSuccess) e = hipMemcpy(d_d, dirs, nb * 12, hipMemcpyHostToDevice);
    rt::StackCfg sk{};
    uint32_t grid = 0;
    if (!rc) rc = stack_config(c, 0, 0, (uint64_t)n, &sk, &grid);
    if (!rc && e == hipSuccess) rc = rt::launch_pt_trace_rays(c, scene_view(c->pt), d_o, d_d, n, any_hit, d_t, d_i, d_c, sk, std::min<uint32_t>(grid, (n + 255u) / 256u));
    if (!rc && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (!rc && e == hipSuccess) e = hipMemcpy(t_out, d_t, nb * 4, hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess) e = hipMemcpy(tri_out, d_i, nb * 4, hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && counts_out) e = hipMemcpy(counts_out, d_c, nb * 8, hipMemcpyDeviceToHost);
    dfree(d_o);
    dfree(d_d);
    dfree(d_t);
    dfree(d_i);
    dfree(d_c);
    if (rc) return rc;
    if (e != hipSuccess) return c->fail(RT_ERR_HIP, "rt_trace_rays: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_trace_rays(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out, int32_t* tri_out) {
    return rt_trace_rays_counted(ctx, origins, dirs, n, any_hit, t_out, tri_out, nullptr);
}

}  // extern "C"
