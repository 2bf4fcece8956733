// Microbenchmark: what HBM write rate does the FlatKmers store pattern reach on MI355X?
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/store_bw.hip -o tools/exp/store_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// variant 0: 4 columns, one record per lane per iteration (8+4+8+4 B per lane)
__global__ __launch_bounds__(256) void k_cols(uint64_t *h, uint32_t *n, uint64_t *r, float *a, int64_t N) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        h[i] = i * 0x9E3779B97F4A7C15ull; n[i] = (uint32_t)i; r[i] = i; a[i] = 1.0f;
    }
}
// variant 1: same but each wave owns a contiguous chunk of 8*64 records (like k_emit_interior)
__global__ __launch_bounds__(256) void k_cols_chunk(uint64_t *h, uint32_t *n, uint64_t *r, float *a, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t b = wave * 512; b < N; b += n_waves * 512) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            int64_t i = b + u * 64 + lane;
            if (i < N) { h[i] = i * 0x9E3779B97F4A7C15ull; n[i] = (uint32_t)i; r[i] = i; a[i] = 1.0f; }
        }
    }
}
// variant 6/7: like the interior kernel: each wave streams CH consecutive 64-record groups, records shifted by `off`
template <int CH>
__global__ __launch_bounds__(256) void k_cols_stream(uint64_t *h, uint32_t *n, uint64_t *r, float *a, int64_t N, int off) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t b = wave * (64 * CH); b < N; b += n_waves * (64 * CH)) {
        for (int u = 0; u < CH; u++) {
            int64_t i = b + u * 64 + lane + off;
            if (i < N && (lane & 15) != 15 - off % 2 * 20) { h[i] = i * 0x9E3779B97F4A7C15ull; n[i] = (uint32_t)i; r[i] = i; a[i] = 1.0f; }
        }
    }
}
// variant 2: only the 8-byte hash column
__global__ __launch_bounds__(256) void k_one(uint64_t *h, int64_t N) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) h[i] = i * 0x9E3779B97F4A7C15ull;
}
// variant 3: two records per lane: 16-B hash/ref stores, 8-B node/af stores
__global__ __launch_bounds__(256) void k_cols2(ulonglong2 *h, uint2 *n, ulonglong2 *r, float2 *a, int64_t N2) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N2; i += stride) {
        h[i] = make_ulonglong2(i * 0x9E3779B97F4A7C15ull, i); n[i] = make_uint2((uint32_t)i, 1u);
        r[i] = make_ulonglong2(i, i + 1); a[i] = make_float2(1.f, 2.f);
    }
}
// variant 4: float4 copy (read + write)
__global__ __launch_bounds__(256) void k_copy(const float4 *in, float4 *out, int64_t N4) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N4; i += stride) out[i] = in[i];
}
// variant 5: 16 B per lane pure store
__global__ __launch_bounds__(256) void k_fill16(float4 *out, int64_t N4) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N4; i += stride) out[i] = make_float4(1, 2, 3, (float)i);
}

int main(int argc, char **argv) {
    int64_t N = argc > 1 ? atoll(argv[1]) : 1000000000ll;
    uint64_t *h, *r; uint32_t *n; float *a;
    CK(hipMalloc(&h, N * 8)); CK(hipMalloc(&r, N * 8)); CK(hipMalloc(&n, N * 4)); CK(hipMalloc(&a, N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {1792, 2048, 8192}) {
        for (int v = 0; v < 9; v++) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                switch (v) {
                case 0: hipLaunchKernelGGL(k_cols, dim3(grid), dim3(256), 0, 0, h, n, r, a, N); break;
                case 1: hipLaunchKernelGGL(k_cols_chunk, dim3(grid), dim3(256), 0, 0, h, n, r, a, N); break;
                case 2: hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, 0, h, N); break;
                case 3: hipLaunchKernelGGL(k_cols2, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)h, (uint2 *)n, (ulonglong2 *)r, (float2 *)a, N / 2); break;
                case 4: hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, (const float4 *)h, (float4 *)r, N * 8 / 16); break;
                case 5: hipLaunchKernelGGL(k_fill16, dim3(grid), dim3(256), 0, 0, (float4 *)h, N * 8 / 16); break;
                case 6: hipLaunchKernelGGL(k_cols_stream<64>, dim3(grid), dim3(256), 0, 0, h, n, r, a, N - 8, 0); break;
                case 7: hipLaunchKernelGGL(k_cols_stream<64>, dim3(grid), dim3(256), 0, 0, h, n, r, a, N - 8, 3); break;
                case 8: hipLaunchKernelGGL(k_cols_stream<8>, dim3(grid), dim3(256), 0, 0, h, n, r, a, N - 8, 3); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            double bytes = v == 2 || v == 5 ? N * 8.0 : v == 4 ? N * 16.0 : N * 24.0;
            const char *names[] = {"4col", "4col-chunk", "hash-only", "4col-2rec", "copy16", "fill16", "stream64", "stream64+3", "stream8+3"};
            printf("grid %5d %-11s %8.3f ms  %7.1f GB/s\n", grid, names[v], best, bytes / best / 1e6);
        }
    }
    return 0;
}
