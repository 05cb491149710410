// rt_abi_frames.hip — frames in flight behind the C ABI.
//
// Reference: src/main.rs:664-667 (one command buffer and one fence per swapchain image) and :882-927
// (wait the acquired image's fence, record, submit behind the previous frame's future, present,
// keep the new fence).  Here a "swapchain image" is a slot = device frame + pinned host frame + two
// events, "present" is the pixels arriving in host memory, and the GPU-side ordering is two HIP
// streams: renders run back to back on the context's stream (they share the pyramid / wavefront
// buffers, like the reference's single queue), each read-back runs on a copy stream behind the
// render it belongs to, so the copy of frame k overlaps the render of frame k+1.
#include <vector>

#include "rt_internal.h"

using rt::Ctx;

namespace {

struct Slot {
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
        if (s.d_rgb) (void)hipFree(s.d_rgb);
        if (s.d_rgba) (void)hipFree(s.d_rgba);
        if (s.h_pixels) (void)hipHostFree(s.h_pixels);
        if (s.ev_rendered) (void)hipEventDestroy(s.ev_rendered);
        if (s.ev_ready) (void)hipEventDestroy(s.ev_ready);
    }
    if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
    delete f;
}

// Common part of the two submit entry points: checks, the slot's fence, then `render` enqueues the
// frame into the slot's device buffer on the context's stream.
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
    if (int rc = render(s.d_rgb)) return rc;
    const void* src = s.d_rgb;
    if (f->format == RT_FRAME_RGBA8) {
        if (int rc = rt::launch_to_rgba8(c, s.d_rgb, s.d_rgba, (uint64_t)f->width * f->height)) return rc;
        src = s.d_rgba;
    }
    RT_HIP(c, hipEventRecord(s.ev_rendered, c->stream));
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
    hipError_t e = hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking);
    for (Slot& s : f->slots) {
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
    return submit(c, slot, [&](float* dst) { return rt_render_device(ctx, rot, pos, spp, dst, 0); });
}

int rt_frame_submit_pt(rt_ctx* ctx, uint32_t slot, const float rot[4], const float pos[3], const rt_pt_params* params) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    return submit(c, slot, [&](float* dst) { return rt_render_pt_device(ctx, rot, pos, params, dst, 0); });
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
