/*
 * rt_abi.h — C ABI of librt_amd.so: the MI355X (gfx950) replacement for the GPU hot path of
 * IvoteSligte/raytracing_engine (per-pixel ray/scene intersection + shading).
 *
 * The reference has no FFI/plugin API for this path: its host (src/main.rs) reaches the GLSL
 * kernels through vulkano descriptor sets and push constants.  The drop-in boundary is therefore
 * that binding contract, restated as plain C:
 *
 *   reference interface (file:line in the reference repo)          replaced by
 *   -------------------------------------------------------------  ---------------------------
 *   Vulkan instance/device/queue selection  src/main.rs:418-460    rt_create / rt_destroy
 *   spec-const RENDER_DIST  src/main.rs:521,636; compute.glsl:32;
 *     GLSL consts fragment.glsl:35-37                              rt_set_config
 *   set0/binding1 MutableData UBO upload  src/main.rs:593-605      rt_set_scene
 *   ConstantBuffer{view,ratio} + image pyramid allocation
 *     src/main.rs:203-266, 608-615, 639-648, resize :813-861       rt_resize / rt_level_info
 *   get_command_buffer: per-level push constants + dispatch,
 *     then the full-screen draw  src/main.rs:268-341;
 *     submit + fence  src/main.rs:909-927                          rt_render / rt_render_spp /
 *                                                                  rt_render_device
 *   swapchain *_UNORM colour attachment  src/main.rs:471-486       rt_read_rgba8
 *   (none: the reference cannot read a pyramid level back)         rt_read_level  [test hook]
 *   FPS println  src/main.rs:719,730                               rt_get_stats
 *   swapchain images + one fence per image: wait the image's fence,
 *     record, submit after the previous frame, present
 *     src/main.rs:664-667, 882-927                                 rt_frames_configure /
 *                                                                  rt_frame_submit(_pt) /
 *                                                                  rt_frame_wait / rt_frame_poll
 *
 * Conventions: every function returns RT_OK (0) or a negative rt_status and never throws or
 * aborts across the boundary; the caller owns every host/device pointer it passes; the context
 * owns all device memory it allocates; one context is bound to one GPU and is single-threaded
 * (like the reference's single queue, src/main.rs:460); distinct contexts are independent (one
 * per GPU / per process).  There is NO CPU fallback: rt_create fails with RT_ERR_NO_DEVICE when
 * no gfx950 device is usable.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 4

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,   /* bad argument (NULL, size mismatch, out-of-range count) */
    RT_ERR_NO_DEVICE = -2, /* no usable HIP device / ordinal out of range */
    RT_ERR_HIP = -3,       /* a HIP runtime call failed; see rt_last_error */
    RT_ERR_STATE = -4,     /* call order violated (e.g. render before set_scene/resize) */
    RT_ERR_OOM = -5        /* host or device allocation failed */
} rt_status;

#define RT_MAX_MATERIALS 8u /* shaders/utilities.glsl:2 */
#define RT_MAX_OBJECTS 8u   /* shaders/utilities.glsl:3 */
#define RT_MAX_LIGHTS 8u    /* shaders/utilities.glsl:4 */
#define RT_MAX_LEVELS 9u    /* shaders/compute.glsl:14-15; src/main.rs:359 */
#define RT_TILE 64u         /* multi-GPU framebuffer tile edge in full-resolution pixels */

/* std140 images of the reference's shader structs, byte-identical to what
 * vulkano_shaders::shader! generates (src/main.rs:47-66; fields at :524-591), so the
 * reference's own `shaders::ty::MutableData` can be passed by pointer. */
typedef struct rt_material { float color[3]; float diffuse; float specular; float shine; float ambient; uint32_t _pad; } rt_material; /* 32 B, utilities.glsl:8-14 */
typedef struct rt_object { float pos[3]; float size; } rt_object;                                                            /* 16 B, utilities.glsl:16-19 */
typedef struct rt_light { float pos[3]; uint32_t _pad0; float color[3]; uint32_t _pad1; } rt_light;                          /* 32 B, utilities.glsl:21-24 */
typedef struct rt_mutable_data { /* compute.glsl:17-24 == fragment.glsl:17-24; 656 B */
    uint32_t matCount, objCount, lightCount, _pad;
    rt_material mats[RT_MAX_MATERIALS];
    rt_object objs[RT_MAX_OBJECTS];
    rt_light lights[RT_MAX_LIGHTS];
} rt_mutable_data;

/* constants the reference bakes in at pipeline-creation / shader-compile time */
typedef struct rt_config {
    float render_dist;    /* RENDER_DIST = 1000   src/main.rs:362 */
    float cam_fall_off;   /* CAM_FALL_OFF = 0.01  fragment.glsl:35 */
    float light_fall_off; /* LIGHT_FALL_OFF = 0.01 fragment.glsl:36 */
    float ray_radius;     /* RAY_RADIUS = 0.01    fragment.glsl:37 */
    uint32_t max_steps;   /* cap on either march loop so every wave terminates (default 1<<20; 0 = none) */
    uint32_t profile_stages; /* 1: bracket every kernel with HIP events (rt_get_stats.stage_ms) */
    uint32_t fuse_levels;    /* 0 (default): one launch per level, the reference's schedule (src/main.rs:300-316);
                                1: the whole depth pyramid in one launch (parent levels kept in LDS; measured slower).
                                Same results; in fused mode level texels without a descendant inside the frame are left 0 */
    /* SDF feature growth the author sketched; 0 / {0,0,0} = the reference as shipped.  Both need fuse_levels = 0. */
    uint32_t march_algorithm; /* cone-march loop body: 0 or 3 = compute.glsl:46-65 ("algorithm 3");
                                 1, 2 = shaders/tracing_algorithms.txt:2-13 / :16-37 in the same loop */
    float repeat[3];          /* > 0: domain repetition period on that axis, repeat() of utilities.glsl:31-34,
                                 applied to the position of every SDF evaluation and of the surface normal */
    uint32_t reflections;     /* mirror bounces, 0..8 (shaders/fragment.glsl:125 "TODO: reflection"; build-defined): after
                                 shading a hit P seen along `step`, r = reflect(step, normal) starts one unit off the surface
                                 (the shadowRay idiom, fragment.glsl:176) and is marched by compute.glsl's loop (:44-66) with
                                 cone threshold RAY_RADIUS: len = 1 + traceCone(P + r, r, RAY_RADIUS); a hit at
                                 P + r * max(len, 0) is shaded by fragment.glsl:144-186 with P as the eye and added with
                                 weight prod(reflectivity * mat.specular) over the surfaces passed */
    float reflectivity;       /* 0..1, default 0.5 (mat.specular, which the reference never reads, scales it per material) */
    uint32_t transmissions;   /* spheres a transmitted ray may cross, 0..8 (shaders/fragment.glsl:124 "TODO: transparency", :126 "TODO:
                                 refraction"; build-defined): after shading a hit P on sphere S seen along I (outward normal n) the ray
                                 enters S - straight on for refraction_index == 1, else T = normalize(refract(I, n, 1 / index)) -
                                 crosses it to the far side in closed form (oc = domain(P) - S.pos, b = oc.T,
                                 disc = b*b - (oc.oc - S.size^2), t = max(disc > 0 ? sqrt(disc) - b : 0, 0), Q = P + T t), leaves along
                                 D = T or normalize(refract(T, -n2, index)) with n2 = normalize(oc + T t) (total internal reflection
                                 ends the chain), and is marched like a mirror ray: len = 1 + traceCone(Q + D, D, RAY_RADIUS); a hit at
                                 Q + D * max(len, 0) is shaded by fragment.glsl:144-186 with Q as the eye and added with weight
                                 prod(transparency * mat.diffuse) over the surfaces passed.  Starts at the camera ray's hit,
                                 independent of the mirror chain.  Needs fuse_levels = 0 only as far as march_algorithm / repeat do. */
    float transparency;       /* 0..1, default 0.5 (mat.diffuse, which the reference never reads, scales it per material: 1 in its scene) */
    float refraction_index;   /* 1..4, default 1 = straight through (transparency); > 1 bends the ray at both surfaces (refraction) */
} rt_config;

typedef struct rt_stats {
    uint32_t width, height, level_count, spp;
    uint64_t frames;           /* rt_render* calls since rt_resize */
    uint64_t primary_rays;     /* last call: width*height*spp (owned tiles only) */
    uint64_t shadow_rays;      /* last call: lightCount per shaded surface point (hit pixel samples + reflection / transmission hits) */
    uint64_t hit_pixels;       /* last call */
    uint64_t cone_threads;     /* last call: compute-stage invocations over all levels */
    float    ms_total;         /* last call: HIP-event time around the whole stage loop */
    float    ms_cone;          /* last call: sum over level kernels (profile_stages=1 only) */
    float    ms_shade;         /* last call: shade kernel(s) (profile_stages=1 only) */
    float    ms_level[RT_MAX_LEVELS]; /* last call, last sample (profile_stages=1, fuse_levels=0 only) */
    float    ms_fused;         /* last call, last sample: the one-launch pyramid kernel (profile_stages=1, fuse_levels=1) */
    uint64_t reflection_rays;  /* last call: mirror rays marched (rt_config.reflections > 0) */
    uint64_t transmission_rays; /* last call: rays marched behind a crossed sphere (rt_config.transmissions > 0) */
} rt_stats;

typedef struct rt_ctx rt_ctx;

int rt_abi_version(void);
int rt_device_count(int* count);

/* device_ordinal >= 0: HIP device index.  No CPU backend exists. */
int rt_create(rt_ctx** out, int device_ordinal);
void rt_destroy(rt_ctx* ctx);
/* message of the last failure on ctx (ctx == NULL: last rt_create failure of this thread) */
const char* rt_last_error(const rt_ctx* ctx);

int rt_default_config(rt_config* cfg);
int rt_set_config(rt_ctx* ctx, const rt_config* cfg);
/* default scene of src/main.rs:524-591 */
int rt_default_scene(rt_mutable_data* scene);
/* bytes must equal sizeof(rt_mutable_data) == 656 */
int rt_set_scene(rt_ctx* ctx, const void* mutable_data, size_t bytes);

/* (re)allocate the depth pyramid for a width x height view; ratio == NULL selects
 * {FOV, FOV*height/width} with FOV = 1 (src/main.rs:364,610) */
int rt_resize(rt_ctx* ctx, uint32_t width, uint32_t height, const float ratio[2]);
/* level count (src/main.rs:639, floor form, capped at 9) and dims[level] = {w,h} (src/main.rs:209-213) */
int rt_level_info(const rt_ctx* ctx, uint32_t* count, uint32_t dims[RT_MAX_LEVELS][2]);

/* Framebuffer partition for multi-GPU rendering: RT_TILE x RT_TILE tiles, tile t (row-major)
 * belongs to rank t % n_ranks.  Default rank 0 of 1 = whole frame. */
int rt_set_partition(rt_ctx* ctx, uint32_t rank, uint32_t n_ranks);
/* tiles_x, tiles_y of the current view and the number of tiles owned by this rank */
int rt_tile_info(const rt_ctx* ctx, uint32_t* tiles_x, uint32_t* tiles_y, uint32_t* owned);

/* Use an external HIP stream (hipStream_t as void*) for all launches; NULL = context's own. */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);

/* One frame at the reference's single pixel-centre sample.  rot = quaternion [x,y,z,w]
 * (push_constants.rot, src/main.rs:772), pos = camera position (:773).
 * rgb_out: host, width*height*3 f32, row-major, row 0 = gl_FragCoord.y 0.5, may be NULL;
 * depth_out: host, the last pyramid level (level dims, see rt_level_info), may be NULL.
 * Synchronous: results are complete on return. */
int rt_render(rt_ctx* ctx, const float rot[4], const float pos[3], float* rgb_out, float* depth_out);
/* spp = n*n stratified sub-pixel centres (n = 1,2,3,..), each sample one full frame, averaged
 * in sample order; spp = 1 is rt_render. */
int rt_render_spp(rt_ctx* ctx, const float rot[4], const float pos[3], uint32_t spp, float* rgb_out);
/* Asynchronous device-side variant: rgb_dev is a device pointer receiving either the full frame
 * (tile_major = 0: width*height*3 f32) or only this rank's tiles packed tile-major
 * (tile_major = 1: owned * RT_TILE*RT_TILE*3 f32, tile k = global tile rank + k*n_ranks).
 * Work is enqueued on the context's stream; no host synchronisation. */
int rt_render_device(rt_ctx* ctx, const float rot[4], const float pos[3], uint32_t spp, void* rgb_dev, int tile_major);
/* Scatter gathered tile-major buffers (n_ranks * tiles_per_rank tiles, rank-major) into a full
 * frame on the device: the de-tile step after an RCCL gather. */
int rt_detile_device(rt_ctx* ctx, const void* tiles_dev, uint32_t n_ranks, uint32_t tiles_per_rank, void* rgb_dev);
int rt_synchronize(rt_ctx* ctx);

/* The exchange step for hosts without a communication library of their own (no reference
 * counterpart: the reference drives one GPU, src/main.rs:448-460).  One process per GPU:
 * rank 0 calls rt_comm_unique_id and hands the 128 bytes to every rank (file, socket, MPI...);
 * every rank calls rt_comm_init (collective; it also sets the framebuffer partition to rank/n_ranks);
 * after rendering tile-major, every rank calls rt_gather_tiles: the tiles go to rank 0 over RCCL
 * (grouped ncclSend/ncclRecv, one direct xGMI link per peer) on the context's stream, into
 * gathered_dev[rank][tiles_per_rank][64][64][3] on rank 0 (ignored elsewhere); rt_detile_device then
 * scatters them into the frame.  tiles_dev holds the `owned` tiles of this rank (rt_tile_info), exactly
 * what rt_render*_device(tile_major = 1) wrote; every rank sends its own count and the root receives each
 * peer's own count, so uneven splits (510 tiles on 8 ranks: 64,64,64,64,64,64,63,63) read and write
 * nothing beyond the owned tiles; tiles_per_rank (the same on every rank) must be >= ceil(tiles / n_ranks).
 * RCCL is loaded on first use. */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]);
int rt_comm_init(rt_ctx* ctx, const uint8_t id[RT_COMM_ID_BYTES], uint32_t rank, uint32_t n_ranks);
int rt_gather_tiles(rt_ctx* ctx, const void* tiles_dev, void* gathered_dev, uint32_t tiles_per_rank);
int rt_comm_destroy(rt_ctx* ctx);

/* Test hook: copy pyramid level `level` of the last frame to host. */
int rt_read_level(rt_ctx* ctx, uint32_t level, float* out, uint32_t* w, uint32_t* h);
/* Last frame as a *_UNORM swapchain would hold it (src/main.rs:471-486): linear, clamped,
 * round-to-nearest, alpha = 255 (never written by fragment.glsl:138,159). width*height*4 bytes. */
int rt_read_rgba8(rt_ctx* ctx, uint8_t* rgba_out);

int rt_get_stats(const rt_ctx* ctx, rt_stats* stats);
/* Test hook: the kernels take square roots with a shortened correctly rounded sequence; this runs it against the
 * compiler's IEEE sqrt on the device for every one of the 2^32 fp32 inputs and returns how many differ (0). */
int rt_selftest_math(rt_ctx* ctx, uint64_t* mismatches);

/* Frames in flight - the reference's swapchain loop (src/main.rs:664-667, 882-927: one fence per
 * swapchain image; a frame waits for its image's fence, is recorded, submitted behind the previous
 * frame and presented) with "present" = the pixels arriving in host memory.  A slot is one swapchain
 * image: a device frame, a pinned host frame and two events.  rt_frame_submit waits until the slot's
 * previous frame has reached the host (the image fence, :882-884), enqueues the render on the
 * slot's own render lane (a child context: own stream and buffers, configuration / scene / mesh of
 * this context as of the submit) and the read-back on a copy stream behind it, and returns without
 * waiting: frames in different slots overlap on the GPU, and every read-back overlaps later renders.
 * Frames of one slot complete in submit order; different slots are independent.  rt_frame_wait blocks until the slot's
 * pixels are in host memory and returns a pointer that stays valid until the slot is submitted
 * again.  format RT_FRAME_F32 = width*height*3 f32 (linear RGB), RT_FRAME_RGBA8 = width*height*4
 * bytes as rt_read_rgba8 defines them.  rt_resize releases the slots: configure again afterwards.
 * Single-rank contexts only (a partitioned context gathers tiles instead). */
#define RT_FRAME_F32 0u
#define RT_FRAME_RGBA8 1u
#define RT_MAX_FRAME_SLOTS 8u
int rt_frames_configure(rt_ctx* ctx, uint32_t n_slots, uint32_t format);
int rt_frame_submit(rt_ctx* ctx, uint32_t slot, const float rot[4], const float pos[3], uint32_t spp);
int rt_frame_wait(rt_ctx* ctx, uint32_t slot, const void** pixels, size_t* bytes);
int rt_frame_poll(rt_ctx* ctx, uint32_t slot, int* ready);

/* ------------------------------------------------------------------------------------------------
 * Path B — build-defined extension (BASELINE.json configs[2..4]): triangle meshes, a BVH and a
 * wavefront path tracer.  NO REFERENCE COUNTERPART: the reference has only sphere SDFs, one
 * deterministic sample and direct lighting (shaders/utilities.glsl:16-19, fragment.glsl:123-126
 * list reflection etc. as TODO).  The camera model (rot/pos/ratio, rt_resize) and the framebuffer
 * partition are shared with path A.  Specification: DESIGN.md §6.
 * ------------------------------------------------------------------------------------------------ */
typedef struct rt_pt_params {
    uint32_t spp;      /* samples per pixel (any value >= 1) */
    uint32_t bounces;  /* indirect bounces after the camera ray (0..15) */
    uint32_t seed;
    float sky[3];      /* radiance of rays that leave the scene */
    float ray_eps;     /* origin offset along the shading normal (default 1e-3) */
    uint32_t count_traversal; /* 1: count BVH nodes fetched / triangles tested (rt_pt_stats) */
    uint32_t max_paths;       /* cap on paths in flight per pass (0 = default 2^25); spp is split into passes */
    uint32_t tune_refill_min;    /* tuning: idle lanes per wave that trigger a refill (0 = default 24; byte 1: triangle tests per round, 0 = 1) */
    uint32_t tune_blocks_per_cu; /* tuning: persistent workgroups per CU (0 = as many as the LDS stacks allow, fewer for few paths) */
    uint32_t tune_lds_stack;     /* tuning: traversal-stack entries kept in LDS per lane (0 = default 8), rest spills */
    uint32_t tune_no_overlap;    /* tuning: how the shadow rays of depth d and the closest-hit rays of depth d + 1 (independent work) share the
                                    GPU: 0 (default) one persistent launch pulls from both queues; 1 one launch after the other;
                                    2 two launches on two streams */
    uint32_t tune_no_packet;     /* tuning: how the camera rays are traced.  0 = default (4); 1 = through the per-lane traversal kernel
                                    like every other ray; 2..5 = wave-uniform packet traversal (one tree walk per 64 camera rays)
                                    with the node test as: 2 = every ray's slab test of all eight children; 3 = one interval test
                                    per child for the whole wave, then the rays' slab tests of the children that pass; 4 = the
                                    interval test alone; 5 = 3 without the cap at the wave's farthest best hit.
                                    Frames do not depend on it. */
    uint32_t tune_sort_rays;     /* tuning: 1 = bounce and shadow rays are sorted inside each 1024-ray workgroup of the shade stage
                                    (LDS counting sort on direction octant + origin cell) before they enter the queues;
                                    2 = the same sort on keys that predict work (shadow rays: segment length in quarter-octaves;
                                    bounce rays: dominant axis and steepness of the direction) */
    uint32_t tune_tri_mode;      /* tuning: how the per-lane traversal kernels schedule their ray/triangle tests.  Byte 0: 0 = default,
                                    1 = inline (every round ends with a triangle phase for the lanes that hold a leaf hit),
                                    2 = wave-pooled (leaf hits go to a per-wave ring in LDS; the whole wave tests 64 of them at once,
                                    results merged with a 64-bit LDS minimum on (t, triangle id)).  Pooled mode: byte 1 = groups in the
                                    ring that trigger a flush (0 = default), byte 2 = rounds a group may wait (0 = default).
                                    3 = postponed tests (bytes 1-2: holding / stuck lanes that trigger the phase), 4 = inline with a
                                    software-pipelined refill (a ring of ready rays per wave in LDS).  Inline mode, byte 1 bit 0: the
                                    fused launch as two loops one after the other instead of one loop with a per-lane ray kind.
                                    Frames do not depend on any of it. */
} rt_pt_params;

typedef struct rt_pt_stats {
    uint32_t n_tris, n_nodes, bvh_depth, n_lights; /* n_nodes / bvh_depth of the compressed 8-wide BVH */
    uint32_t stack_need;       /* worst-case traversal stack entries for this BVH */
    float bvh_build_ms;
    uint32_t stack_overflow;   /* must be 0: traversal stack never exceeded */
    uint64_t camera_rays, bounce_rays, shadow_rays; /* last render: rays handed to BVH traversal */
    uint64_t nodes_visited, tris_tested;            /* per-lane closest-hit traversal (pt_trace<closest>, alone or inside a fused launch): BVH node
                                                       records fetched / triangle records fetched and tested; last render, count_traversal = 1 only */
    uint64_t shadow_nodes_visited, shadow_tris_tested; /* any-hit (shadow) launches, same condition */
    uint64_t wave_rounds, alive_lane_rounds;           /* closest-hit launches: traversal rounds per wave summed, lanes holding a live ray summed */
    float ms_total;            /* last render: HIP-event time around the stage loop */
    float ms_generate, ms_trace_closest, ms_shade, ms_trace_shadow, ms_resolve; /* profile_stages = 1 only */
    uint32_t launches_trace_closest, launches_trace_shadow;
    uint64_t packets;          /* camera-ray waves walked by the packet kernel; last render, count_traversal = 1 only */
    uint64_t packet_nodes_fetched, packet_tris_fetched; /* packet kernel: records fetched, once per wave of 64 camera rays (count_traversal = 1) */
    uint64_t fused_shadow_nodes, fused_shadow_tris, fused_shadow_rays; /* the part of the shadow_* counts / of shadow_rays that ran inside fused launches */
    float ms_trace_packet, ms_trace_fused; /* profile_stages = 1: the packet kernel; the fused shadow(d) + closest(d + 1) launches */
    uint32_t launches_trace_fused;
    uint32_t bvh_levels, blas_chunks, tlas_nodes; /* 1 / 0 / 0 for a single-level mesh */
    float ms_build_blas, ms_build_tlas, ms_build_flatten; /* two-level meshes: phases of the last build or chunk rebuild (bvh_build_ms = all of it) */
    uint64_t pool_flushes;     /* count_traversal = 1, wave-pooled triangle tests: passes in which a wave tested up to 64 pooled triangles */
    uint64_t wave_rounds_all;  /* count_traversal = 1: traversal rounds per wave summed over closest-hit AND any-hit launches */
} rt_pt_stats;

/* How rt_set_mesh_ex builds the acceleration structure.  bvh_levels = 1: one BVH8 over all triangles (rt_set_mesh).
 * bvh_levels = 2 (BASELINE.json configs[2] "2-level BVH"): the triangles are cut into blas_chunks chunks (0 = 64), the
 * leaves of a top-down binned-SAH split that always divides the fullest leaf, every chunk gets a bottom-level BVH8 of its
 * own, a top-level BVH8 is built over the chunk boxes, and both levels are flattened into the one node array the kernels traverse.  Frames are identical either
 * way (results do not depend on the tree); the two-level mesh can have one chunk's vertices replaced and only that
 * chunk rebuilt (rt_update_mesh_chunk). */
typedef struct rt_mesh_options {
    uint32_t bvh_levels;  /* 1 or 2 */
    uint32_t blas_chunks; /* bvh_levels = 2: number of bottom-level chunks, 0 = 64 (on average at least four triangles per chunk) */
} rt_mesh_options;

int rt_default_pt_params(rt_pt_params* p);
/* verts: n_tris*9 (v0,v1,v2), albedo: n_tris*3, emission: n_tris*3 (any component > 0 = light).
 * Uploads the mesh and builds the BVH on the host (binned SAH -> compressed 8-wide nodes, threaded; the result does
 * not depend on the thread count).  Host pointers, copied. */
int rt_set_mesh(rt_ctx* ctx, const float* verts, const float* albedo, const float* emission, uint32_t n_tris);
/* rt_set_mesh with build options (options == NULL: rt_set_mesh). */
int rt_set_mesh_ex(rt_ctx* ctx, const float* verts, const float* albedo, const float* emission, uint32_t n_tris, const rt_mesh_options* options);
/* Two-level meshes: the triangles of bottom-level chunk `chunk` (count, and their original indices into tri_ids[capacity]
 * when tri_ids != NULL), in the order rt_update_mesh_chunk expects their vertices. */
int rt_mesh_chunk_info(rt_ctx* ctx, uint32_t chunk, uint32_t* count, uint32_t* tri_ids, uint32_t capacity);
/* Two-level meshes: replace the vertices of one chunk's triangles (n_tris * 9 floats, rt_mesh_chunk_info order; materials
 * and chunk membership stay) and rebuild that chunk's BVH, the top level and the flattened node array only.  n_tris must
 * equal the chunk's triangle count (RT_ERR_INVALID otherwise: nothing is read).  The new vertices must stay within the
 * coordinate range of the mesh as first set (RT_ERR_INVALID otherwise: set the mesh again).  Transactional: on any error
 * other than RT_ERR_STATE the mesh is exactly as it was; RT_ERR_STATE means an upload failed and the mesh was dropped. */
int rt_update_mesh_chunk(rt_ctx* ctx, uint32_t chunk, const float* verts, uint32_t n_tris);
/* Synchronous path-traced frame of the current view (rt_resize) into host memory.
 * Every |pos| component must be <= 32 x max(1, largest |vertex coordinate| of the mesh): that is the range
 * over which the BVH's conservative box padding covers the fp32 rounding of the ray/box test (beyond it the
 * frame could depend on the tree); RT_ERR_INVALID otherwise. */
int rt_render_pt(rt_ctx* ctx, const float rot[4], const float pos[3], const rt_pt_params* params, float* rgb_out);
/* Asynchronous device-side variant, same output layouts as rt_render_device. */
int rt_render_pt_device(rt_ctx* ctx, const float rot[4], const float pos[3], const rt_pt_params* params, void* rgb_dev, int tile_major);
/* Path B frame into a frames-in-flight slot (see rt_frame_submit). */
int rt_frame_submit_pt(rt_ctx* ctx, uint32_t slot, const float rot[4], const float pos[3], const rt_pt_params* params);
int rt_get_pt_stats(rt_ctx* ctx, rt_pt_stats* stats);
/* Test hook: trace n caller-supplied rays (host arrays, n*3 each).  any_hit = 0: closest hit,
 * t_out[i] = distance (inf on miss), tri_out[i] = original triangle index or -1;
 * any_hit = 1: tri_out[i] = 1 if the open segment (o, o + 0.999*d) is occluded. */
int rt_trace_rays(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out, int32_t* tri_out);
/* Test hook: as rt_trace_rays, and counts_out[2i], counts_out[2i+1] (may be NULL) receive the number of BVH
 * nodes fetched and triangles tested for ray i by the traversal step functions the render kernels run:
 * tests/native/bvh8_walk.cpp walks the same tree on the host and must reproduce them exactly, which is what
 * entitles bench.py to quote rt_pt_stats.nodes_visited / tris_tested. */
int rt_trace_rays_counted(rt_ctx* ctx, const float* origins, const float* dirs, uint32_t n, int any_hit, float* t_out, int32_t* tri_out,
                          uint32_t* counts_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
