// rt_abi_comm.hip — the one exchange step of a multi-GPU frame behind the C ABI: every rank's
// tile-major buffer goes to rank 0 over RCCL (xGMI inside a node).  No reference counterpart: the
// reference drives one GPU (src/main.rs:448-460).  A host without PyTorch (host/rt_host.cpp, the
// Rust main) uses these entry points; bench.py does the same exchange with torch.distributed.
//
// RCCL is opened lazily with dlopen so that single-GPU users and processes that already carry
// PyTorch's own RCCL copy never load a second one.  The pattern is a grouped ncclSend / ncclRecv to
// the root - every peer has its own direct xGMI link to it, so the 7 transfers run in parallel; a ring
// collective would serialise them over per-link bandwidth.  Payload per rank at 1920x1080 f32 RGB:
// 64 tiles x 48 KiB = 3 MiB.
#include <dlfcn.h>

#include <cstring>

#include "rt_internal.h"
#include "rt_roctx.h"

using rt::Ctx;

namespace {

// the slice of rccl.h this file needs (the library is resolved at run time)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7 };

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (r.handle) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
            r.Send = (decltype(r.Send))dlsym(r.handle, "ncclSend");
            r.Recv = (decltype(r.Recv))dlsym(r.handle, "ncclRecv");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) r.handle = nullptr;
        }
    }
    return r.handle ? &r : nullptr;
}

int nccl_fail(Ctx* c, const char* what, ncclResult_t e) {
    Rccl* r = rccl();
    return c->fail(RT_ERR_HIP, "%s: %s", what, r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
}

}  // namespace

namespace rt {
void comm_free(Ctx* c) {
    if (c->comm) {
        if (Rccl* r = rccl()) (void)r->CommDestroy(static_cast<ncclComm_t>(c->comm));
        c->comm = nullptr;
    }
}
}  // namespace rt

extern "C" {

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
    if (!id) return RT_ERR_INVALID;
    Rccl* r = rccl();
    if (!r) return RT_ERR_NO_DEVICE;
    ncclUniqueId u;
    if (r->GetUniqueId(&u) != ncclSuccess) return RT_ERR_HIP;
    static_assert(sizeof u == RT_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id, &u, sizeof u);
    return RT_OK;
}

int rt_comm_init(rt_ctx* ctx, const uint8_t id[RT_COMM_ID_BYTES], uint32_t rank, uint32_t n_ranks) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!id || n_ranks == 0 || rank >= n_ranks) return c->fail(RT_ERR_INVALID, "rank %u of %u", rank, n_ranks);
    Rccl* r = rccl();
    if (!r) return c->fail(RT_ERR_NO_DEVICE, "librccl.so could not be loaded");
    RT_HIP(c, hipSetDevice(c->device));
    rt::comm_free(c);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r->CommInitRank(&comm, (int)n_ranks, u, (int)rank);
    if (e != ncclSuccess) return nccl_fail(c, "ncclCommInitRank", e);
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_ranks = n_ranks;
    c->part.rank = rank;  // the communicator's ranks are the framebuffer partition's ranks
    c->part.n_ranks = n_ranks;
    return RT_OK;
}

int rt_gather_tiles(rt_ctx* ctx, const void* tiles_dev, void* gathered_dev, uint32_t tiles_per_rank) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (!c->comm) return c->fail(RT_ERR_STATE, "rt_comm_init has not been called");
    if (!tiles_dev || tiles_per_rank == 0) return c->fail(RT_ERR_INVALID, "NULL tile buffer or tiles_per_rank = 0");
    if (c->comm_rank == 0 && !gathered_dev) return c->fail(RT_ERR_INVALID, "rank 0 needs the gather buffer");
    if (!c->width) return c->fail(RT_ERR_STATE, "rt_resize has not been called");
    rt::RoctxRange rr("rt.gather_tiles");
    // Every rank sends exactly the tiles it owns (tiles_dev holds `owned` tiles, as rt_render*_device(tile_major)
    // documents: with 510 tiles on 8 ranks, ranks 6 and 7 own 63, not ceil(510/8) = 64) and the root receives
    // each peer's own count into that peer's block of tiles_per_rank tiles; pad tiles are never touched.
    const uint32_t total = c->part.tiles_x * c->part.tiles_y, n = c->comm_ranks;
    auto owned_by = [&](uint32_t rank) { return total > rank ? (total - rank + n - 1u) / n : 0u; };
    if (tiles_per_rank < owned_by(0)) return c->fail(RT_ERR_INVALID, "tiles_per_rank %u < %u tiles owned by rank 0", tiles_per_rank, owned_by(0));
    Rccl* r = rccl();
    RT_HIP(c, hipSetDevice(c->device));
    const size_t tile_floats = (size_t)RT_TILE * RT_TILE * 3;
    const size_t block = (size_t)tiles_per_rank * tile_floats;  // floats per rank in gathered_dev
    ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return nccl_fail(c, "ncclGroupStart", e);
    if (c->comm_rank == 0) {
        float* dst = static_cast<float*>(gathered_dev);
        for (uint32_t peer = 1; peer < n && e == ncclSuccess; peer++)
            if (owned_by(peer)) e = r->Recv(dst + (size_t)peer * block, owned_by(peer) * tile_floats, ncclFloat32, (int)peer, comm, c->stream);
    } else if (owned_by(c->comm_rank)) {
        e = r->Send(tiles_dev, owned_by(c->comm_rank) * tile_floats, ncclFloat32, 0, comm, c->stream);
    }
    const ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess) return nccl_fail(c, "ncclSend/ncclRecv", e);
    if (e2 != ncclSuccess) return nccl_fail(c, "ncclGroupEnd", e2);
    if (c->comm_rank == 0 && gathered_dev != tiles_dev && owned_by(0))  // the root's own tiles: a device-to-device copy on the same stream
        RT_HIP(c, hipMemcpyAsync(gathered_dev, tiles_dev, owned_by(0) * tile_floats * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

int rt_comm_destroy(rt_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return RT_ERR_INVALID;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rt::comm_free(c);
    return RT_OK;
}

}  // extern "C"
