// Microbenchmark: random 8-byte loads per second from a table of S bytes on MI355X (what bounds the index probe).
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/rand_read.hip -o tools/exp/rand_read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int U>
__global__ __launch_bounds__(256) void k_rand(const uint64_t *__restrict__ table, uint64_t n_entries, int64_t n_loads,
                                              uint64_t *__restrict__ sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n_loads; i0 += stride * U) {
        uint64_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = table[__umul64hi(mix(i0 + u * stride), n_entries)];
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

// two-level: every load first tests one bit of a small bitmap (S1 bytes); a fraction `pass`/256 goes on to the big table
template <int U>
__global__ __launch_bounds__(256) void k_two(const uint64_t *__restrict__ filter, uint64_t n_filter_words,
                                             const uint64_t *__restrict__ table, uint64_t n_entries, int64_t n_loads,
                                             unsigned pass, uint64_t *__restrict__ sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n_loads; i0 += stride * U) {
        uint64_t f[U], h[U];
#pragma unroll
        for (int u = 0; u < U; u++) { h[u] = mix(i0 + u * stride); f[u] = filter[__umul64hi(h[u], n_filter_words)]; }
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc ^= f[u];
            if (((h[u] >> 13) & 255u) < pass) acc ^= table[__umul64hi(h[u] * 0x9E3779B97F4A7C15ull, n_entries)];
        }
    }
    if (acc == 0x1234567ull) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int64_t n_loads = 1ll << 32;
    uint64_t *sink; CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t big = 3623443816ull;      // the 3.6 GB directory
    uint64_t *table; CK(hipMalloc(&table, big)); CK(hipMemset(table, 1, big));
    const double sizes_mb[] = {2, 16, 32, 56, 128, 200, 256, 512, 1024, 3455};
    for (double mb : sizes_mb) {
        uint64_t n_entries = (uint64_t)(mb * 1048576.0 / 8);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_rand<4>, dim3(2048), dim3(256), 0, 0, table, n_entries, n_loads, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("table %7.0f MB: %.2f G loads/s\n", mb, n_loads / ms / 1e6);
        }
    }
    uint64_t *filter; CK(hipMalloc(&filter, 256ull << 20)); CK(hipMemset(filter, 1, 256ull << 20));
    const double fmb[] = {56, 128, 200};
    const unsigned passes[] = {128, 50, 13};
    for (double mb : fmb) for (unsigned pass : passes) {
        uint64_t nfw = (uint64_t)(mb * 1048576.0 / 8);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_two<4>, dim3(2048), dim3(256), 0, 0, filter, nfw, table, big / 8, n_loads, pass, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("filter %4.0f MB, %3u/256 continue to the 3.6 GB table: %.2f G queries/s\n", mb, pass, n_loads / ms / 1e6);
        }
    }
    return 0;
}
