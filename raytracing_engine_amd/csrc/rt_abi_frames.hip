// rt_abi_frames.hip — frames in flight behind the C ABI.
//
// Reference: src/main.rs:664-667 (one command buffer and one fence per swapchain image) and :882-927
// (wait the acquired image's fence, record, submit behind the previous frame's future, present,
// keep the new fence).  Here a "swapchain image" is a slot = device frame + pinned host frame + two
// events, "present" is the pixels arriving in host memory, and the GPU-side ordering is two HIP
// streams.  Every slot is also a render LANE: a child context of its own (own stream, own pyramid /
// wavefront buffers; configuration and scene copied from the parent, the triangle mesh borrowed from
// it), so the frames in different slots overlap on the GPU as well - the coarse pyramid levels of
// path A and the tails of path B's traversal kernels are latency-bound and leave most of the chip
// idle for one frame alone.  Each read-back runs on the copy stream behind the render it belongs to.
// (The reference chains its frames on one queue, src/main.rs:909-915; overlapping them is this
// build's addition and does not change any frame.)
#include <vector>

#include "rt_internal.h"

using rt::Ctx;

namespace {

struct Slot {
    rt_ctx* lane = nullptr;     // child context that renders this slot's frames
    uint64_t lane_version = 0;  // parent state_version the lane was last synchronised with
    float* d_rgb = nullptr;     // render target (f32 x 3)
    uint8_t* d_rgba = nullptr;  // RT_FRAME_RGBA8 only
    void* h_pixels = nullptr;   // pinned
    hipEvent_t ev_rendered = nullptr, ev_ready = nullptr;
    bool pending = false;    // submitted and not yet known to be complete
    bool submitted = false;  // holds (or will hold) a frame
};

struct Frames {
    std::vector<Slot> slots;
    uint32_t format = RT_FRAME_F32;
    uint32_t width = 0, height = 0;
    size_t bytes = 0;  // per host frame
    hipStream_t copy_stream = nullptr;
};

void release(Frames* f) {
    if (!f) return;
    if (f->copy_stream) (void)hipStreamSynchronize(f->copy_stream);
    for (Slot& s : f->slots) {
        if (s.lane) rt_destroy(s.lane);  // synchronises the lane's streams first
        if (s.d_rgb) (void)hipFree(s.d_rgb);
        if (s.d_rgba) (void)hipFree(s.d_rgba);
        if (s.h_pixels) (void)hipHostFree(s.h_pixels);
        if (s.ev_rendered) (void)hipEventDestroy(s.ev_rendered);
        if (s.ev_ready) (void)hipEventDestroy(s.ev_ready);
    }
    if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
    delete f;
}

// Bring a slot's lane up to date with the parent's configuration, scene and mesh.
int sync_lane(Ctx* c, Slot& s) {
    if (s.lane_version == c->state_version) return RT_OK;
    Ctx* l = reinterpret_cast<Ctx*>(s.lane);
    RT_HIP(c, hipStreamSynchronize(l->stream));
    if (l->aux_stream) RT_HIP(c, hipStreamSynchronize(l->aux_stream));
    l->cfg = c->cfg;
    l->scene = c->scene;
    l->have_scene = c->have_scene;
    rt::pt_borrow_mesh(l, c);
    s.lane_version = c->state_version;
    return RT_OK;
}

// Common part of the two submit entry points: checks, the slot's fence, then `render` enqueues the
// frame into the slot's device buffer on the slot's lane.
template <typename Render>
int submit(Ctx* c, uint32_t slot, Render render) {
    Frames* f = static_cast<Frames*>(c->frames);
    if (!f) return c->fail(RT_ERR_STATE, "rt_frames_configure has not been called (or the view was resized since)");
    if (slot >= f->slots.size()) return c->fail(RT_ERR_INVALID, "slot %u of %zu", slot, f->slots.size());
    if (c->part.n_ranks > 1) return c->fail(RT_ERR_STATE, "frames in flight need an unpartitioned context");
    RT_HIP(c, hipSetDevice(c->device));
    Slot& s = f->slots[slot];
    if (s.pending) {  // the image fence: the slot's previous frame must have left the device (src/main.rs:882-884)
        RT_HIP(c, hipEventSynchronize(s.ev_ready));
        s.pending = false;
    }
    if (int rc = sync_lane(c, s)) return rc;
    Ctx* l = reinterpret_cast<Ctx*>(s.lane);
    if (int rc = render(s.lane, s.d_rgb)) return c->fail(rc, "%s", rt_last_error(s.lane));
    const void* src = s.d_rgb;
    if (f->format == RT_FRAME_RGBA8) {
        if (int rc = rt::launch_to_rgba8(l, s.d_rgb, s.d_rgba, (uint64_t)f->width * f->height)) return c->fail(rc, "%s", rt_last_error(s.lane));
        src = s.d_rgba;
    }
    RT_HIP(c, hipEventRecord(s.ev_rendered, l->stream));
    RT_HIP(c, hipStreamWaitEvent(f->copy_stream, s.ev_rendered, 0));
    RT_HIP(c, hipMemcpyAsync(s.h_pixels, src, f->bytes, hipMemcpyDeviceToHost, f->copy_stream));
    RT_HIP(c, hipEventRecord(s.ev_ready, f->copy_stream));
    s.pending = true;
    s.submitted = true;
    return RT_OK;
}

}  // namespace

namespace rt {
void frames_free(Ctx* c) {
    release(static_cast<Frames*>(c->frames));
    c->frames = nullptr;
}

void frames_drop_mesh(Ctx* c) {
    Frames* f = static_cast<Frames*>(c->frames);
    if (!f) return;
    for (Slot& s : f->slots) {
        Ctx* l = reinterpret_cast<Ctx*>(s.lane);
        if (!l) continue;
        (void)hipStreamSynchronize(l->stream);
        if (l->aux_stream) (void)hipStreamSynchronize(l->aux_stream);
        pt_borrow_mesh(l, nullptr);
        s.lane_version = 0;
    }
}
}  // namespace rt

extern "C" {

int rt_frames_configure(rt_ctx* ctx, uint32_t n_slots, uint32_t format) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (n_slots < 1 || n_slots > RT_MAX_FRAME_SLOTS) return c->fail(RT_ERR_INVALID, "n_slots %u out of [1,%u]", n_slots, RT_MAX_FRAME_SLOTS);
    if (format != RT_FRAME_F32 && format != RT_FRAME_RGBA8) return c->fail(RT_ERR_INVALID, "unknown frame format %u", format);
    if (!c->width) return c->fail(RT_ERR_STATE, "rt_resize has not been called");
    RT_HIP(c, hipSetDevice(c->device));
    if (c->stream) RT_HIP(c, hipStreamSynchronize(c->stream));
    rt::frames_free(c);
    Frames* f = new (std::nothrow) Frames;
    if (!f) return c->fail(RT_ERR_OOM, "frame slots");
    f->format = format;
    f->width = c->width;
    f->height = c->height;
    const size_t px = (size_t)c->width * c->height;
    f->bytes = format == RT_FRAME_RGBA8 ? px * 4 : px * 3 * sizeof(float);
    f->slots.resize(n_slots);
    // the copy stream gets the highest priority: priority levels have hardware queues of their own, so the
    // read-backs never queue up behind a lane's kernels (streams of one priority share a handful of queues)
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    hipError_t e = hipStreamCreateWithPriority(&f->copy_stream, hipStreamNonBlocking, prio_high);
    for (Slot& s : f->slots) {
        if (e == hipSuccess && (rt_create(&s.lane, c->device) != RT_OK || rt_resize(s.lane, c->width, c->height, c->ratio) != RT_OK)) e = hipErrorOutOfMemory;
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_rgb, px * 3 * sizeof(float));
        if (e == hipSuccess && format == RT_FRAME_RGBA8) e = hipMalloc((void**)&s.d_rgba, px * 4);
        if (e == hipSuccess) e = hipHostMalloc(&s.h_pixels, f->bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_rendered, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_ready, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        release(f);
        return c->fail(e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, "frame slots: %s", hipGetErrorString(e));
    }
    c->frames = f;
    return RT_OK;
}

int rt_frame_submit(rt_ctx* ctx, uint32_t slot, const float rot[4], const float pos[3], uint32_t spp) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    return submit(c, slot, [&](rt_ctx* lane, float* dst) { return rt_render_device(lane, rot, pos, spp, dst, 0); });
}

int rt_frame_submit_pt(rt_ctx* ctx, uint32_t slot, const float rot[4], const float pos[3], const rt_pt_params* params) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    return submit(c, slot, [&](rt_ctx* lane, float* dst) { return rt_render_pt_device(lane, rot, pos, params, dst, 0); });
}

int rt_frame_wait(rt_ctx* ctx, uint32_t slot, const void** pixels, size_t* bytes) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    Frames* f = static_cast<Frames*>(c->frames);
    if (!f) return c->fail(RT_ERR_STATE, "rt_frames_configure has not been called (or the view was resized since)");
    if (slot >= f->slots.size() || !pixels) return c->fail(RT_ERR_INVALID, "slot %u of %zu / NULL output", slot, f->slots.size());
    Slot& s = f->slots[slot];
    if (!s.submitted) return c->fail(RT_ERR_STATE, "slot %u holds no frame", slot);
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipEventSynchronize(s.ev_ready));
    s.pending = false;
    *pixels = s.h_pixels;
    if (bytes) *bytes = f->bytes;
    return RT_OK;
}

int rt_frame_poll(rt_ctx* ctx, uint32_t slot, int* ready) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    Frames* f = static_cast<Frames*>(c->frames);
    if (!f) return c->fail(RT_ERR_STATE, "rt_frames_configure has not been called (or the view was resized since)");
    if (slot >= f->slots.size() || !ready) return c->fail(RT_ERR_INVALID, "slot %u of %zu / NULL output", slot, f->slots.size());
    Slot& s = f->slots[slot];
    if (!s.submitted) return c->fail(RT_ERR_STATE, "slot %u holds no frame", slot);
    RT_HIP(c, hipSetDevice(c->device));
    const hipError_t e = hipEventQuery(s.ev_ready);
    if (e != hipSuccess && e != hipErrorNotReady) return c->fail(RT_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(e));
    *ready = e == hipSuccess ? 1 : 0;
    if (e == hipSuccess) s.pending = false;
    return RT_OK;
}

}  // extern "C"
