// Microbenchmarks for a bucket-blocked probe on MI355X:
//  (1) random 8-byte loads confined, per XCD (blockIdx % 8), to a 1 MB window of a 3.6 GB table that moves every `per`
//      loads -- does the window stay in the XCD's 4 MB L2?
//  (2) scattering 8-byte elements into 3456 contiguous streams by a per-stream atomic cursor (direct global scatter)
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/blocked_probe.hip -o tools/exp/blocked_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

__global__ __launch_bounds__(256) void k_windowed(const uint64_t *__restrict__ table, uint64_t n_entries, uint64_t win_entries,
                                                  int64_t loads_per_block_per_window, int n_windows, uint64_t *__restrict__ sink) {
    const int xcd = blockIdx.x & 7;
    const int64_t blk = blockIdx.x >> 3;
    uint64_t acc = 0;
    for (int w = 0; w < n_windows; w++) {
        const uint64_t base = (mix(w * 8 + xcd) % (n_entries / win_entries)) * win_entries;     // this XCD's window
        for (int64_t i = threadIdx.x; i < loads_per_block_per_window; i += 256 * 4) {
            uint64_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = table[base + __umul64hi(mix(((uint64_t)w << 40) ^ ((uint64_t)blk << 24) ^ (uint64_t)(i + u * 256)), win_entries)];
#pragma unroll
            for (int u = 0; u < 4; u++) acc ^= v[u];
        }
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

__global__ __launch_bounds__(256) void k_scatter(uint64_t *__restrict__ out, unsigned long long *__restrict__ cursor, int n_streams,
                                                 int64_t stream_cap, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t h = mix(i);
        const int s = (int)__umul64hi(h, (uint64_t)n_streams);
        const unsigned long long p = atomicAdd(&cursor[s], 1ull);
        if ((int64_t)p < stream_cap) out[(int64_t)s * stream_cap + p] = h;
    }
}

// block-aggregated: ranks via LDS counters per tile, one global atomic per (tile, stream) with elements
template <int NS>
__global__ __launch_bounds__(1024) void k_scatter_tile(uint64_t *__restrict__ out, unsigned long long *__restrict__ cursor,
                                                       int64_t stream_cap, int64_t n, int per_thread) {
    __shared__ uint32_t cnt[NS];
    __shared__ unsigned long long basep[NS];
    const int64_t tile = (int64_t)per_thread * 1024;
    for (int64_t t0 = (int64_t)blockIdx.x * tile; t0 < n; t0 += (int64_t)gridDim.x * tile) {
        for (int s = threadIdx.x; s < NS; s += 1024) cnt[s] = 0;
        __syncthreads();
        uint64_t h[16]; uint32_t rk[16]; int st[16];
        for (int u = 0; u < per_thread; u++) {
            const int64_t i = t0 + u * 1024 + threadIdx.x;
            h[u] = mix(i); st[u] = (int)__umul64hi(h[u], (uint64_t)NS);
            rk[u] = i < n ? atomicAdd(&cnt[st[u]], 1u) : 0u;
        }
        __syncthreads();
        for (int s = threadIdx.x; s < NS; s += 1024) basep[s] = cnt[s] ? atomicAdd(&cursor[s], (unsigned long long)cnt[s]) : 0ull;
        __syncthreads();
        for (int u = 0; u < per_thread; u++) {
            const int64_t i = t0 + u * 1024 + threadIdx.x;
            const unsigned long long p = basep[st[u]] + rk[u];
            if (i < n && (int64_t)p < stream_cap) out[(int64_t)st[u] * stream_cap + p] = h[u];
        }
        __syncthreads();
    }
}

int main() {
    uint64_t *sink; CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t big = 3623443816ull;
    uint64_t *table; CK(hipMalloc(&table, big)); CK(hipMemset(table, 1, big));
    const double wins_mb[] = {0.25, 0.5, 1, 2, 4, 8};
    for (double mb : wins_mb) {
        const uint64_t win_entries = (uint64_t)(mb * 1048576 / 8);
        const int n_windows = 64; const int64_t per = 1 << 14;      // per block per window; 256 blocks per XCD
        const double loads = 2048.0 * n_windows * per;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_windowed, dim3(2048), dim3(256), 0, 0, table, big / 8, win_entries, per, n_windows, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("window %5.2f MB per XCD, %d loads per entry-window sweep: %.1f G loads/s\n", mb, (int)(256 * per), loads / ms / 1e6);
        }
    }
    // scatter
    const int NS = 3456; const int64_t n = 2400000000ll; const int64_t cap = n / NS + n / NS / 8;
    uint64_t *out; CK(hipMalloc(&out, (size_t)NS * cap * 8));
    unsigned long long *cursor; CK(hipMalloc(&cursor, NS * 8));
    for (int variant = 0; variant < 3; variant++) for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(cursor, 0, NS * 8));
        CK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(k_scatter, dim3(2048), dim3(256), 0, 0, out, cursor, NS, cap, n);
        else if (variant == 1) hipLaunchKernelGGL((k_scatter_tile<3456>), dim3(1024), dim3(1024), 0, 0, out, cursor, cap, n, 8);
        else hipLaunchKernelGGL((k_scatter_tile<3456>), dim3(1024), dim3(1024), 0, 0, out, cursor, cap, n, 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("scatter variant %d: %.2f ms for %.1f GB -> %.0f GB/s\n", variant, ms, n * 8 / 1e9, n * 8.0 / ms / 1e6);
    }
    return 0;
}
