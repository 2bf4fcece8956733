// Measurement helpers of libgki_hip.so: the box's own store ceiling for the FlatKmers column pattern (the roofline of
// the headline kernel is the write stream, and boxes of one pool differ by a quarter), and a device-side read simulator
// so that the read-mapping benchmark can run at BASELINE configs[4]'s 1e8 reads (their host simulation takes minutes).
#include "gki_common.h"

namespace {

// Four FlatKmers columns (8 + 4 + 8 + 4 bytes per record), every wave streaming 64 consecutive 64-record groups:
// the store pattern of k_emit_interior_runs with nothing else in the kernel (tools/exp/store_bw.hip, variant stream64).
__global__ __launch_bounds__(256) void k_store_columns(uint64_t *__restrict__ h, uint32_t *__restrict__ n, uint64_t *__restrict__ r,
                                                       float *__restrict__ a, int64_t N) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t b = wave * 4096; b < N; b += n_waves * 4096) {
        for (int u = 0; u < 64; u++) {
            const int64_t i = b + u * 64 + lane;
            if (i < N) { h[i] = (uint64_t)i * 0x9E3779B97F4A7C15ull; n[i] = (uint32_t)i; r[i] = (uint64_t)i; a[i] = 1.0f; }
        }
    }
}

__device__ __host__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// One wave per read.  Everything about a read is a pure function of (seed, read index, base index):
//   h0 = mix(2r + seed * K), h1 = mix(2r + 1 + seed * K)      start = h0 mod (hap_len - L + 1)
//   random read iff (h1 & 0xFFFF) < p_random_q16;  reverse strand iff bit 16 of h1
//   base i: hb = mix(mix(seed) + r * L + i); random read: code = hb & 3; else code = hap[start + i], substituted by
//           (code + 1 + ((hb >> 24) mod 3)) & 3 iff ((hb >> 8) & 0xFFFF) < p_sub_q16
//   reverse strand: letter i of the read = complement of code L-1-i
// tests/test_gpu_reads_sim.py restates this in NumPy.
__global__ __launch_bounds__(256) void k_simulate_reads(const uint8_t *__restrict__ hap, int64_t hap_len, int64_t n_reads, int L,
                                                        uint64_t seed, uint32_t p_sub_q16, uint32_t p_random_q16,
                                                        int64_t first_read, uint8_t *__restrict__ letters) {
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t seed2 = splitmix64(seed), sk = seed * 0xD1342543DE82EF95ull;
    for (int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; q < n_reads; q += n_waves) {
        const uint64_t r = (uint64_t)(first_read + q);
        const uint64_t h0 = splitmix64(2 * r + sk), h1 = splitmix64(2 * r + 1 + sk);
        const int64_t start = (int64_t)(h0 % (uint64_t)(hap_len - L + 1));
        const bool is_random = (uint32_t)(h1 & 0xFFFF) < p_random_q16, rc = (h1 >> 16) & 1;
        for (int i = lane; i < L; i += 64) {
            const uint64_t hb = splitmix64(seed2 + r * (uint64_t)L + (uint64_t)i);
            uint32_t code;
            if (is_random) code = (uint32_t)(hb & 3);
            else {
                code = hap[start + i] & 3u;
                if ((uint32_t)((hb >> 8) & 0xFFFF) < p_sub_q16) code = (code + 1u + (uint32_t)((hb >> 24) % 3)) & 3u;
            }
            const int o = rc ? L - 1 - i : i;
            if (rc) code = 3u - code;
            letters[q * L + o] = (uint8_t)("ACGT"[code]);
        }
    }
}

}  // namespace

// gki_wave_incl_sum (DPP) against the shuffle form, lane by lane, on pseudo-random and on extreme inputs
__global__ __launch_bounds__(256) void k_selftest_wave_scan(uint64_t seed, int *__restrict__ n_bad) {
    const int lane = threadIdx.x & 63;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const int kind = blockIdx.x & 3;
    const int x = kind == 0 ? (int)(z & 31) : kind == 1 ? (int)(z & 0xFFFFF) : kind == 2 ? ((z >> 7) & 1 ? 1 : 0) : (lane == (int)(seed & 63) ? 1000 : 0);
    int ref = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(ref, d, 64); if (lane >= d) ref += t; }
    const int got = gki_wave_incl_sum(x);
    const int tot = gki_lane_value(got, 63);
    if (got != ref || tot != __shfl(ref, 63, 64)) atomicAdd(n_bad, 1);
}

extern "C" {

int gki_measure_store_bw(void *d_hashes, void *d_nodes, void *d_ref_offsets, void *d_af32, int64_t n, double *bytes_per_s) {
    *bytes_per_s = 0.0;
    if (n < (1 << 20)) return gki_set_error(GKI_ERR_BAD_ARG, "measure_store_bw: at least 2^20 records");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float best = 0.f;
    int rc = GKI_OK;
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    HIP_G(hipEventCreate(&e0));
    HIP_G(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {              // best of the last two launches
        float ms = 0.f;
        HIP_G(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_store_columns, dim3(2048), dim3(256), 0, 0, (uint64_t *)d_hashes, (uint32_t *)d_nodes,
                           (uint64_t *)d_ref_offsets, (float *)d_af32, n);
        HIP_G(hipGetLastError());
        HIP_G(hipEventRecord(e1, 0));
        HIP_G(hipEventSynchronize(e1));
        HIP_G(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && (best == 0.f || ms < best)) best = ms;
    }
    if (best > 0.f) *bytes_per_s = 24.0 * (double)n / ((double)best * 1e-3);
done:
#undef HIP_G
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

int gki_selftest_wave_scan(int64_t *n_bad) {
    int *d_bad = nullptr;
    int h_bad = -1;
    HIP_TRY(hipMalloc(&d_bad, sizeof(int)));
    hipError_t e = hipMemset(d_bad, 0, sizeof(int));
    for (int rep = 0; rep < 4 && e == hipSuccess; rep++) {
        hipLaunchKernelGGL(k_selftest_wave_scan, dim3(1024), dim3(256), 0, 0, (uint64_t)(0x1234567ull * (rep + 1) + rep * 17), d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return gki_set_error(GKI_ERR_HIP, "selftest_wave_scan: %s", hipGetErrorString(e));
    *n_bad = h_bad;
    return GKI_OK;
}

int gki_simulate_reads(const void *d_haplotype, int64_t hap_len, int64_t n_reads, int read_len, uint64_t seed,
                       double p_substitution, double p_random_read, int64_t first_read, void *d_letters) {
    if (read_len < 1 || hap_len < read_len) return gki_set_error(GKI_ERR_BAD_ARG, "simulate_reads: haplotype shorter than a read");
    if (p_substitution < 0 || p_substitution > 1 || p_random_read < 0 || p_random_read > 1)
        return gki_set_error(GKI_ERR_BAD_ARG, "simulate_reads: probabilities must lie in [0, 1]");
    if (n_reads <= 0) return GKI_OK;
    const uint32_t ps = (uint32_t)(p_substitution * 65536.0 + 0.5), pr = (uint32_t)(p_random_read * 65536.0 + 0.5);
    hipLaunchKernelGGL(k_simulate_reads, dim3(stream_grid(n_reads * 64, 256)), dim3(256), 0, 0, (const uint8_t *)d_haplotype, hap_len,
                       n_reads, read_len, seed, ps, pr, first_read, (uint8_t *)d_letters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return GKI_OK;
}

}  // extern "C"
