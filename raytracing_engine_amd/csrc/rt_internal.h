// rt_internal.h — context object and launch-parameter blocks shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "bvh_build.h"

static_assert(sizeof(rt_material) == 32, "std140 Material stride");
static_assert(sizeof(rt_object) == 16, "std140 Object stride");
static_assert(sizeof(rt_light) == 32, "std140 Light stride");
static_assert(sizeof(rt_mutable_data) == 656, "std140 MutableData size");
static_assert(offsetof(rt_mutable_data, mats) == 16 && offsetof(rt_mutable_data, objs) == 272 &&
                  offsetof(rt_mutable_data, lights) == 400,
              "std140 MutableData offsets");

namespace rt {

// Spheres of the scene, passed by value in the kernarg segment: the compiler keeps them in
// SGPRs (s_load from kernarg), so the 8 objects cost no VGPRs and no vector memory traffic.
struct SphereSet {
    float4 s[RT_MAX_OBJECTS];  // xyz = centre, w = radius
};

struct ShadeSet {  // everything fragment.glsl reads from MutableData
    float4 sphere[RT_MAX_OBJECTS];
    float4 mat_color_ambient[RT_MAX_MATERIALS];  // rgb, ambient
    float mat_shine[RT_MAX_MATERIALS];
    float mat_specular[RT_MAX_MATERIALS];  // read only by the mirror-reflection variant (the reference never reads mat.specular)
    float mat_diffuse[RT_MAX_MATERIALS];   // read only by the transmission variant (the reference never reads mat.diffuse)
    float4 light_pos[RT_MAX_LIGHTS];
    float4 light_color[RT_MAX_LIGHTS];
    uint32_t light_count;
};

// Framebuffer partition (multi-GPU): RT_TILE^2 tiles, tile t belongs to rank t % n_ranks.
struct Partition {
    uint32_t rank, n_ranks, tiles_x, tiles_y;
};

struct Camera {
    float rot[4];
    float pos[3];
    float ratio[2];
    float jitter[2];
};

// compute.glsl push constants + ConstantBuffer for one pyramid level
struct ConeLevelParams {
    Camera cam;
    float image_size[2];  // 2^(count-1-level) / view   (src/main.rs:303-305)
    uint32_t level;       // pc.iter
    uint32_t w, h;        // level image dims (multiples of 8)
    uint32_t parent_w;
    uint32_t shift;       // count-1-level: level pixel -> full-res pixel
    uint32_t width, height;  // full-res view
    float render_dist;
    uint32_t max_steps;
    Partition part;
    uint32_t partitioned;  // 1: skip level tiles this rank does not own / that are off-screen
    // sample batch: blockIdx.y = b traces sample sample0 + b of an n_strata x n_strata stratified pixel
    // into image b of the level (images level_stride floats apart; parents parent_stride apart)
    uint32_t sample0, n_strata;
    uint32_t level_stride, parent_stride;
    uint32_t alg;     // march loop body: 3 = compute.glsl:46-65, 1 / 2 = tracing_algorithms.txt:2-13 / :16-37
    float repeat[3];  // > 0: domain repetition period on that axis (utilities.glsl:31-34)
};

struct ShadeParams {
    Camera cam;
    float view[2];
    uint32_t width, height;
    uint32_t depth_w;  // row pitch of the last pyramid level
    float render_dist, cam_fall_off, light_fall_off, ray_radius;
    uint32_t max_steps;
    Partition part;
    uint32_t tile_major;  // 0: dst is a full frame, 1: dst holds owned tiles packed tile-major
    uint32_t mode;        // bit0: accumulate onto dst, bit1: divide by spp after adding
    float spp;
    // sample batch: the kernel shades samples sample0 .. sample0 + n_batch - 1 of the pixel in index
    // order (depth image b is depth_stride floats after image 0) and adds them in that order
    uint32_t sample0, n_batch, n_strata;
    uint32_t depth_stride;
    float repeat[3];  // > 0: domain repetition period on that axis (utilities.glsl:31-34)
    uint32_t reflections;  // mirror bounces (0 = the reference as shipped)
    float reflectivity;
    uint32_t transmissions;  // spheres a transmitted ray may cross (0 = the reference as shipped)
    float transparency, refraction_index;
};

// Sub-pixel offset of sample s of an n x n stratified pixel in NDC: the stratum centre (i + 0.5)/n inside
// the pixel is ((2i + 1)/n - 1)/view; n = 1 gives exactly 0 = the reference's pixel-centre sample.  The
// same IEEE expression on host and device (correctly rounded divisions on both).
__host__ __device__ inline void sample_jitter(uint32_t s, uint32_t n, uint32_t width, uint32_t height, float* jx, float* jy) {
    const uint32_t si = s % n, sj = s / n;
    *jx = ((float)(2u * si + 1u) / (float)n - 1.0f) / (float)width;
    *jy = ((float)(2u * sj + 1u) / (float)n - 1.0f) / (float)height;
}

// One-launch pyramid: every level for a 32x32 pixel block per workgroup (path_a.hip)
struct PyramidParams {
    Camera cam;
    uint32_t width, height;
    float render_dist;
    uint32_t max_steps;
    Partition part;
    uint32_t count;                        // pyramid levels
    float image_size[RT_MAX_LEVELS][2];    // per level: 2^(count-1-level) / view
    uint32_t level_w[RT_MAX_LEVELS];       // row pitch of each level image
    float* level[RT_MAX_LEVELS];           // level images in HBM
};

// ---- path B (triangles + BVH + path tracing) -------------------------------------
// Per-depth counter block.  The ray queues are consumed through PT_HEADS interleaved streams: stream k
// owns the 64-entry blocks k, k + PT_HEADS, k + 2 PT_HEADS ... of the queue and has its own head word
// on its own 128-byte line (one hot word answers only ~90 atomics/us chip-wide; sixteen words let the
// traversal waves refill a few lanes at a time while all streams together still advance as one
// compact window over the queue).
enum {
    PT_HEADS = 16,
    PT_HEAD_STRIDE = 32,  // words between head words
    PT_CTR_COUNT = 0,
    PT_CTR_SHADOW_COUNT = 1,
    PT_CTR_HEAD_CLOSEST = PT_HEAD_STRIDE,
    PT_CTR_HEAD_SHADOW = PT_HEAD_STRIDE * (1 + PT_HEADS),
    PT_CTR_STRIDE = PT_HEAD_STRIDE * (1 + 2 * PT_HEADS)
};

struct PtScene {
    const float4* nodes;     // 5 x float4 (80 B) per compressed 8-wide BVH node (bvh_build.h layout)
    const float4* tris;      // 3 x float4 per triangle, leaf order: v0.xyz e1.x | e1.yz e2.xy | e2.z id emissive-flag -
    const float4* albedo;    // leaf order
    const float4* emission;  // leaf order
    const uint32_t* lights;  // leaf-order indices of emissive triangles, ascending original id
    uint32_t n_lights;
    uint32_t n_tris;
};

constexpr uint32_t kPacketStackEntries = 40;  // pt_trace_packet's LDS stack of node groups (path_b.hip)
// rt_pt_params.tune_no_packet: 0 = default, 1 = no packet kernel (camera rays through the per-lane kernel), then the packet kernel's node test:
// per-ray slab tests of all eight children; interval test for the pass, per-ray tests of the children that pass; interval test only; the second without the best-hit cap
enum { PACKET_DEFAULT = 0, PACKET_OFF = 1, PACKET_EXACT = 2, PACKET_INTERVAL = 3, PACKET_INTERVAL_ONLY = 4, PACKET_INTERVAL_NOCAP = 5 };
enum { TRI_MODE_INLINE = 1, TRI_MODE_POOL = 2, TRI_MODE_DEFER = 3, TRI_MODE_INLINE_PF = 4 };  // rt_pt_params.tune_tri_mode, byte 0 (path_b.hip: TRI_INLINE, TRI_POOL)

struct StackCfg {  // per-lane traversal stack of 8-byte entries: lds_cap in LDS, then spill_cap in global memory
    unsigned long long* spill;  // spill_cap x spill_stride entries, entry-major
    size_t spill_stride;  // = threads of the persistent grid
    int lds_cap, spill_cap;
};

struct PtState {  // SoA over path ids; one float4 per lane per array = 16-byte coalesced accesses
    float4* ray_o;
    float4* ray_d;
    float4* thr;   // path throughput
    float4* rad;   // accumulated radiance of the path
    float2* hit;   // t, leaf-order triangle index (int bits, -1 = miss)
    float4* sh_o;  // shadow queue: origin.xyz, path id bits
    float4* sh_d;  // shadow queue: unnormalised direction to the light sample
    float4* sh_c;  // shadow queue: contribution if unoccluded
};

struct PtFrame {
    Camera cam;
    uint32_t width, height;
    Partition part;
    uint32_t n_slots;    // owned tiles * 4096
    uint32_t n_paths;    // n_slots * spp_batch
    uint32_t spp_batch;  // samples in flight per pixel in this pass
    uint32_t sample0;    // first sample index of this pass
    uint32_t spp_total;
    uint32_t bounces;
    uint32_t seed;
    float sky[3];
    float ray_eps;
};

struct MeshHost {  // host side of a two-level mesh: what rt_update_mesh_chunk needs to rebuild one chunk
    TwoLevelBvh tl;
    std::vector<float> v0, e1, e2;          // original triangle order
    std::vector<float> albedo, emission;    // original triangle order, 3 floats each
    std::vector<uint32_t> light_ids;        // emissive triangles, ascending
};

struct PtData {  // device residency of one mesh + the wavefront buffers
    std::unique_ptr<MeshHost> host;  // two-level meshes only
    size_t cap_nodes = 0;            // nodes d_nodes has room for
    bool borrowed_mesh = false;  // the mesh arrays belong to another context (frame-slot lanes share their parent's mesh)
    uint32_t n_tris = 0, n_nodes = 0, n_lights = 0, bvh_depth = 0;
    float bvh_build_ms = 0.0f, bvh_pad = 0.0f, bvh_maxabs = 1.0f;  // bvh_maxabs = max(1, largest |vertex coordinate|): what the padding covers
    float4* d_nodes = nullptr;
    float4* d_tris = nullptr;
    float4* d_albedo = nullptr;
    float4* d_emission = nullptr;
    uint32_t* d_lights = nullptr;
    uint32_t stack_need = 0;  // worst-case traversal stack occupancy reported by the builder
    unsigned long long* d_spill = nullptr;
    size_t spill_words = 0, spill_half = 0;
    hipEvent_t ev_shaded = nullptr, ev_shadowed = nullptr;  // ordering between the main and the auxiliary stream
    std::vector<hipEvent_t> ev_pool;  // profile_stages: timing events, created on this context's device, freed by pt_free
    // wavefront buffers, sized for cap_paths
    uint64_t cap_paths = 0;
    uint32_t cap_depth = 0;
    PtState st{};
    uint32_t* d_queue[2] = {nullptr, nullptr};
    uint32_t* d_ctr = nullptr;
    unsigned long long* d_stats = nullptr;
    float* d_acc = nullptr;  // per-slot running sums across sample batches
    uint64_t cap_slots = 0;
    rt_pt_stats stats{};
};

struct Ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;  // path B: shadow kernel beside the next closest-hit kernel
    std::string err;

    rt_config cfg{};
    bool have_scene = false;
    rt_mutable_data scene{};

    uint32_t width = 0, height = 0;
    float ratio[2] = {1.0f, 1.0f};
    uint32_t level_count = 0;
    uint32_t dims[RT_MAX_LEVELS][2] = {};
    float* d_level[RT_MAX_LEVELS] = {};  // level i: level_batch images of dims[i], one per sample of a batch
    uint32_t level_batch = 0;            // images allocated per level
    uint32_t last_image = 0;             // image of the batch that holds the last sample rendered
    float* d_rgb = nullptr;         // full frame, f32 x 3
    uint8_t* d_rgba8 = nullptr;     // rt_read_rgba8 staging, width*height*4, allocated on first use, freed with the frame
    uint64_t* d_counters = nullptr;  // 4 x 1024 slots: hit pixels, secondary hits shaded, mirror rays, transmitted rays; summed on the host
    Partition part{0, 1, 0, 0};

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_stage;  // profile_stages
    rt_stats stats{};
    bool frame_valid = false;
    PtData pt;
    int n_cus = 256;
    void* frames = nullptr;  // frames-in-flight slots (rt_abi_frames.hip)
    uint64_t state_version = 1;  // bumped by rt_set_config / rt_set_scene / rt_set_mesh: frame-slot lanes re-sync on submit
    void* comm = nullptr;  // ncclComm_t once rt_comm_init ran (rt_abi_comm.hip)
    uint32_t comm_rank = 0, comm_ranks = 1;

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define RT_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return (ctx)->fail(RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

void frames_free(Ctx* c);          // rt_abi_frames.hip: waits for frames in flight, releases every slot
void frames_drop_mesh(Ctx* c);     // rt_abi_frames.hip: the parent's mesh is about to be freed: idle the lanes, forget the borrowed arrays
void pt_borrow_mesh(Ctx* lane, const Ctx* owner);  // rt_abi_pt.hip: lane renders with owner's device mesh

// path_a.hip
int launch_cone_level(Ctx* c, const SphereSet& spheres, uint32_t n_obj, const ConeLevelParams& p, const float* parent,
                      float* out, uint32_t batch);
int launch_shade(Ctx* c, const ShadeSet& set, uint32_t n_obj, const ShadeParams& p, const float* depth, float* dst,
                 uint64_t* counters);
int launch_pyramid_fused(Ctx* c, const SphereSet& spheres, uint32_t n_obj, const PyramidParams& fp);
int launch_selftest_sqrt(Ctx* c, unsigned long long* mismatches_dev);
int launch_detile(Ctx* c, const float* tiles, uint32_t n_ranks, uint32_t tiles_per_rank, float* rgb);
int launch_to_rgba8(Ctx* c, const float* rgb, uint8_t* rgba, uint64_t n_pixels);

// path_b.hip
int launch_pt_generate(Ctx* c, const PtFrame& f, const PtState& st, uint32_t* queue, uint32_t* ctr);
int launch_pt_trace(Ctx* c, const PtScene& sc, const PtState& st, const uint32_t* queue, const uint32_t* count_ptr, uint32_t* head,
                    unsigned long long* stats, bool any_hit, bool count, uint32_t grid, const StackCfg& stack_cap, uint32_t refill_min, uint32_t tri_mode,
                    uint32_t tri_cfg);
int launch_pt_trace_fused(Ctx* c, const PtScene& sc, const PtState& st, const uint32_t* queue, const uint32_t* closest_count, uint32_t* closest_head,
                          const uint32_t* shadow_count, uint32_t* shadow_head, unsigned long long* stats, bool count, uint32_t grid,
                          const StackCfg& stack_cap, uint32_t refill_min, uint32_t tri_mode, uint32_t tri_cfg);
uint32_t pt_pool_lds_bytes(uint32_t tri_mode);  // static LDS a 256-thread workgroup of the per-lane kernels needs beyond the stacks and the octant table
int launch_pt_trace_packet(Ctx* c, const PtScene& sc, const PtFrame& f, const PtState& st, unsigned long long* stats, bool count, uint32_t mode);
int launch_pt_shade(Ctx* c, const PtScene& sc, const PtFrame& f, const PtState& st, const uint32_t* queue, const uint32_t* count_ptr,
                    uint32_t depth, uint32_t* next_queue, uint32_t* next_ctr, uint32_t grid, uint32_t sort_rays);
int launch_pt_resolve(Ctx* c, const PtFrame& f, const PtState& st, float* acc, float* dst, int tile_major);
int launch_pt_trace_rays(Ctx* c, const PtScene& sc, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out,
                         int* tri_out, uint32_t* counts, const StackCfg& sk, uint32_t grid);
void pt_free(Ctx* c);
void comm_free(Ctx* c);  // rt_abi_comm.hip

}  // namespace rt
