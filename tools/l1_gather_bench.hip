// l1_gather_bench.hip — what does the vector L1 (TCP) charge for gathers on gfx950?
// Each lane issues 16-byte loads; variants differ only in how the 64 addresses of a wave are grouped:
//   0: every lane its own random 16-byte chunk            (fully divergent)
//   1: each quad reads the four chunks of one random 64-byte line (lane k -> chunk k)
//   2: every 8 lanes read one random 128-byte line
//   3: fully coalesced (lane k -> base + 16k)
// Table small enough to stay in L2 (4 MiB) / and a large one (256 MiB) to see the miss path.
//   hipcc -O3 --offload-arch=gfx950 tools/l1_gather_bench.hip -o /tmp/l1bench && /tmp/l1bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ table, uint32_t n_chunks, int iters, float4* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    float4 acc = make_float4(0, 0, 0, 0);
    uint32_t s = hash32(gid + 1u);
    for (int it = 0; it < iters; it++) {
        s = hash32(s + it);
        uint32_t idx;
        if (MODE == 0) idx = s % n_chunks;
        else if (MODE == 1) { uint32_t q = __shfl(s, lane & ~3u); idx = ((q % (n_chunks / 4)) * 4u) + (lane & 3u); }
        else if (MODE == 2) { uint32_t q = __shfl(s, lane & ~7u); idx = ((q % (n_chunks / 8)) * 8u) + (lane & 7u); }
        else { uint32_t q = __shfl(s, 0); idx = ((q % (n_chunks / 64)) * 64u) + lane; }
        const float4 v = table[idx];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[gid] = acc;
}

// Record patterns: 80-byte records (the BVH node format), one random record per lane and iteration.
//   REC 0: each lane issues five 16-byte loads of its own record (the traversal kernel's pattern)
//   REC 1: quad-cooperative: in instruction j (0..3) the four lanes of a quad read the first 64 bytes of
//          the record owned by lane j of the quad; the fifth chunk is loaded by the owner itself
// 64-byte records, 64-byte aligned: four 16-byte loads per lane (what a 6-bit-quantised node would cost)
__global__ __launch_bounds__(256) void gather_records64(const float4* __restrict__ table, uint32_t n_rec, int iters, float4* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    uint32_t s = hash32(gid + 1u);
    for (int it = 0; it < iters; it++) {
        s = hash32(s + it);
        const float4* p = table + (size_t)(s % n_rec) * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x += a.x + b.x + c.x + d.x;
    }
    out[gid] = acc;
}

template <int REC>
__global__ __launch_bounds__(256) void gather_records(const float4* __restrict__ table, uint32_t n_rec, int iters, float4* out) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    float4 acc = make_float4(0, 0, 0, 0);
    uint32_t s = hash32(gid + 1u);
    for (int it = 0; it < iters; it++) {
        s = hash32(s + it);
        const uint32_t rec = s % n_rec;
        if (REC == 0) {
            const float4* p = table + (size_t)rec * 5;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc.x += a.x + b.x + c.x + d.x + e.x;
        } else {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t owner_rec = __shfl(rec, (lane & ~3u) + j);
                v[j] = table[(size_t)owner_rec * 5 + (lane & 3u)];
            }
            const float4 e = table[(size_t)rec * 5 + 4];
            acc.x += v[0].x + v[1].x + v[2].x + v[3].x + e.x;
        }
    }
    out[gid] = acc;
}

int main() {
    const size_t sizes[4] = {2u << 20, 16u << 20, 20u << 20, 512u << 20};
    for (size_t bytes : sizes) {
        const uint32_t n_chunks = (uint32_t)(bytes / 16);
        float4* table; float4* out;
        hipMalloc(&table, bytes); hipMemset(table, 0, bytes);
        const int blocks = 256 * 8, iters = 2000;
        hipMalloc(&out, (size_t)blocks * 256 * 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 4; mode++) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, n_chunks, iters, out);
                if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_chunks, iters, out);
                if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, table, n_chunks, iters, out);
                if (mode == 3) hipLaunchKernelGGL(gather<3>, dim3(blocks), dim3(256), 0, 0, table, n_chunks, iters, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double lanes = (double)blocks * 256 * iters;
            printf("table %4zu MiB mode %d: %.3f ms  %.2f lane-loads/clk/CU (2.3 GHz)  %.1f GB/s/CU  %.2f TB/s\n", bytes >> 20, mode, best,
                   lanes / (best * 1e-3) / 256 / 2.3e9, lanes * 16 / (best * 1e-3) / 256 / 1e9, lanes * 16 / (best * 1e-3) / 1e12);
        }
        for (int rec = 0; rec < 2; rec++) {
            const uint32_t n_rec = (uint32_t)(bytes / 80);
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                if (rec == 0) hipLaunchKernelGGL(gather_records<0>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters / 4, out);
                else hipLaunchKernelGGL(gather_records<1>, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters / 4, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double recs = (double)blocks * 256 * (iters / 4);
            printf("table %4zu MiB 80-byte records, pattern %d: %.3f ms  %.3f records/clk/CU  %.2f TB/s useful\n", bytes >> 20, rec, best,
                   recs / (best * 1e-3) / 256 / 2.3e9, recs * 80 / (best * 1e-3) / 1e12);
        }
        {
            const uint32_t n_rec = (uint32_t)(bytes / 64);
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(gather_records64, dim3(blocks), dim3(256), 0, 0, table, n_rec, iters / 4, out);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            const double recs = (double)blocks * 256 * (iters / 4);
            printf("table %4zu MiB 64-byte aligned records: %.3f ms  %.3f records/clk/CU  %.2f TB/s useful\n", bytes >> 20, best,
                   recs / (best * 1e-3) / 256 / 2.3e9, recs * 64 / (best * 1e-3) / 1e12);
        }
        hipFree(table); hipFree(out);
    }
    return 0;
}
