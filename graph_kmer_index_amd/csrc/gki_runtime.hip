// Runtime plumbing of libgki_hip.so: errors, memory, and the exclusive-scan primitive.
#include "gki_common.h"
#include <stdarg.h>
#include <time.h>

thread_local char gki_err_buf[512] = "";

int gki_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(gki_err_buf, sizeof(gki_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

// ------------------------------------------------------------------------------------------ device memory pool
// hipMalloc / hipFree of tens of GB are not cheap on this stack: a fresh 76 GB allocation costs 0.4 s, and after two
// alloc/free rounds of that size the third hipMalloc was measured at 4.3 s (the frees are deferred and paid then).  The
// build and probe paths allocate their temporaries per call, so every buffer of the library goes through a cache of
// freed blocks instead: gki_dev_free parks the block, gki_dev_malloc reuses a parked block of at least the requested
// size and at most 1/8 more.  At most half of the device's memory is parked; when the device runs out, the cache is
// released and the allocation retried.
// gki_trim() releases it on demand; GKI_POOL=0 disables the cache.
#include <map>
#include <mutex>
#include <unordered_map>
namespace {
std::mutex g_pool_mu;
std::multimap<size_t, void *> g_pool_free;          // parked blocks by size
std::unordered_map<void *, size_t> g_pool_live;     // size of every block handed out
size_t g_pool_cached = 0;                           // bytes parked
// what the device allocator itself cost this process (gki_pool_stats): calls that reached hipMalloc / hipFree, their time
int64_t g_n_malloc = 0, g_n_free = 0;
double g_ms_malloc = 0.0, g_ms_free = 0.0;
double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec; }
hipError_t timed_malloc(void **p, size_t bytes) {
    const double t0 = now_ms();
    hipError_t e = hipMalloc(p, bytes);
    g_ms_malloc += now_ms() - t0; g_n_malloc++;
    return e;
}
hipError_t timed_free(void *p) {
    const double t0 = now_ms();
    hipError_t e = hipFree(p);
    g_ms_free += now_ms() - t0; g_n_free++;
    return e;
}
size_t pool_cap() {                                 // park at most half of the device's memory (other libraries allocate too)
    static size_t cap = 0;
    if (!cap) { size_t f = 0, t = 0; cap = hipMemGetInfo(&f, &t) == hipSuccess ? t / 2 : (size_t)64 << 30; }
    return cap;
}
bool pool_enabled() { static const bool on = !(getenv("GKI_POOL") && atoi(getenv("GKI_POOL")) == 0); return on; }
void pool_trim() {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    for (auto &kv : g_pool_free) (void)timed_free(kv.second);
    g_pool_free.clear();
    g_pool_cached = 0;
}
}  // namespace

hipError_t gki_dev_malloc(void **ptr, size_t bytes) {
    *ptr = nullptr;
    if (bytes == 0) bytes = 16;
    // size classes: 256 B steps below 1 MB, above it 1/16 of the size's power of two (at least 2 MB), so that buffers
    // sized by record counts that differ by a fraction of a percent (shards, slices) land in the same class
    size_t gran = 256;
    if (bytes >= (1u << 20)) {
        size_t p2 = (size_t)1 << 20;
        while ((p2 << 1) <= bytes) p2 <<= 1;
        gran = p2 / 16 > ((size_t)2 << 20) ? p2 / 16 : (size_t)2 << 20;
    }
    bytes = (bytes + gran - 1) / gran * gran;
    if (pool_enabled()) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        auto it = g_pool_free.lower_bound(bytes);
        if (it != g_pool_free.end() && it->first <= bytes + bytes / 8) {
            *ptr = it->second;
            g_pool_live[*ptr] = it->first;
            g_pool_cached -= it->first;
            g_pool_free.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = timed_malloc(ptr, bytes);
    if (e != hipSuccess) {                               // out of memory: give the cache back and retry once
        (void)hipGetLastError();
        pool_trim();
        e = timed_malloc(ptr, bytes);
    }
    if (e == hipSuccess && pool_enabled()) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        g_pool_live[*ptr] = bytes;
    }
    return e;
}

hipError_t gki_dev_free(void *ptr) {
    if (!ptr) return hipSuccess;
    if (pool_enabled()) {
        // like hipFree, do not return before work that may still use the block has finished
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(g_pool_mu);
        auto it = g_pool_live.find(ptr);
        if (it != g_pool_live.end()) {
            const size_t bytes = it->second;
            g_pool_live.erase(it);
            if (g_pool_cached + bytes > pool_cap()) return timed_free(ptr);
            g_pool_free.emplace(bytes, ptr);
            g_pool_cached += bytes;
            return hipSuccess;
        }
    }
    return timed_free(ptr);
}

namespace {
// order-independent checksums of a column: sum mod 2^64 and xor of the zero-extended elements
template <typename T>
__global__ __launch_bounds__(256) void k_checksum(const T *__restrict__ v, int64_t n, unsigned long long *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint64_t s = 0, x = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const uint64_t e = v[i]; s += e; x ^= e; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_down(s, d, 64); x ^= __shfl_down(x, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], (unsigned long long)s); atomicXor(&out[1], (unsigned long long)x); }
}
}  // namespace

namespace {
// ---- stable compaction of FlatKmers columns by a byte flag per record
__global__ __launch_bounds__(256) void k_widen_flags(const uint8_t *__restrict__ f, int64_t n, uint32_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = f[i] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_compact_flat(const uint8_t *__restrict__ f, const int64_t *__restrict__ pos, int64_t n,
                                                      int64_t base, const uint64_t *__restrict__ h, const uint32_t *__restrict__ nd,
                                                      const uint64_t *__restrict__ r, const float *__restrict__ af,
                                                      uint64_t *__restrict__ oh, uint32_t *__restrict__ ond,
                                                      uint64_t *__restrict__ orf, float *__restrict__ oaf) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!f[i]) continue;
        const int64_t o = base + pos[i];
        oh[o] = h[i]; ond[o] = nd[i]; orf[o] = r[i]; oaf[o] = af[i];
    }
}
}  // namespace

extern "C" {

const char *gki_last_error(void) { return gki_err_buf; }

int gki_device_count(int *count) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; return gki_set_error(GKI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = c;
    return GKI_OK;
}

int gki_set_device(int device) { HIP_TRY(hipSetDevice(device)); return GKI_OK; }

int gki_malloc(void **d_ptr, int64_t bytes) {
    *d_ptr = nullptr;
    if (bytes < 0) return gki_set_error(GKI_ERR_BAD_ARG, "gki_malloc: negative size");
    HIP_TRY(gki_dev_malloc(d_ptr, (size_t)bytes));
    return GKI_OK;
}

int gki_free(void *d_ptr) { if (d_ptr) HIP_TRY(gki_dev_free(d_ptr)); return GKI_OK; }

int gki_trim(void) { pool_trim(); return GKI_OK; }

int gki_pool_stats(int64_t *n_device_mallocs, int64_t *n_device_frees, double *ms_in_malloc, double *ms_in_free, int64_t *bytes_parked) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    *n_device_mallocs = g_n_malloc; *n_device_frees = g_n_free; *ms_in_malloc = g_ms_malloc; *ms_in_free = g_ms_free;
    *bytes_parked = (int64_t)g_pool_cached;
    return GKI_OK;
}

int gki_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes) {
    if (bytes > 0) HIP_TRY(hipMemcpy(d_dst, h_src, (size_t)bytes, hipMemcpyHostToDevice));
    return GKI_OK;
}

int gki_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes) {
    if (bytes > 0) HIP_TRY(hipMemcpy(h_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost));
    return GKI_OK;
}
int gki_memcpy_d2d(void *d_dst, const void *d_src, int64_t bytes) {
    if (bytes > 0) HIP_TRY(hipMemcpy(d_dst, d_src, (size_t)bytes, hipMemcpyDeviceToDevice));
    return GKI_OK;
}

int gki_memset(void *d_dst, int value, int64_t bytes) {
    if (bytes > 0) HIP_TRY(hipMemset(d_dst, value, (size_t)bytes));
    return GKI_OK;
}

int gki_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return GKI_OK; }

int gki_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f; *total_bytes = (int64_t)t;
    return GKI_OK;
}

int gki_compact_flat(const void *d_flags, int64_t n, const void *d_hashes, const void *d_nodes, const void *d_ref_offsets,
                     const void *d_af32, void *d_out_hashes, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                     int64_t out_capacity, int64_t *n_out) {
    *n_out = 0;
    if (n <= 0) return GKI_OK;
    const int64_t CH = 1ll << 27;                     // records per scan: bounds the temporaries at 1.6 GB
    const int64_t m = n < CH ? n : CH;
    uint32_t *wide = nullptr; int64_t *pos = nullptr; void *tmp = nullptr;
    const int64_t tmp_bytes = gki_scan_tmp_bytes(m);
    hipError_t e = gki_dev_malloc((void **)&wide, (size_t)m * 4);
    if (e == hipSuccess) e = gki_dev_malloc((void **)&pos, (size_t)(m + 1) * 8);
    if (e == hipSuccess) e = gki_dev_malloc(&tmp, (size_t)tmp_bytes);
    int rc = e == hipSuccess ? GKI_OK : gki_set_error(GKI_ERR_HIP, "gki_compact_flat: %s", hipGetErrorString(e));
    int64_t base = 0;
    for (int64_t a = 0; a < n && rc == GKI_OK; a += CH) {
        const int64_t c = n - a < CH ? n - a : CH;
        const uint8_t *f = (const uint8_t *)d_flags + a;
        hipLaunchKernelGGL(k_widen_flags, dim3(stream_grid(c, 256)), dim3(256), 0, 0, f, c, wide);
        rc = gki_scan_u32_to_i64(wide, c, pos, tmp, tmp_bytes, 0);
        int64_t kept = 0;
        if (rc == GKI_OK && hipMemcpy(&kept, pos + c, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = gki_set_error(GKI_ERR_HIP, "copy of the chunk total failed");
        if (rc == GKI_OK && base + kept > out_capacity)
            rc = gki_set_error(GKI_ERR_BAD_ARG, "gki_compact_flat: output needs more than %lld records", (long long)out_capacity);
        if (rc == GKI_OK && kept > 0) {
            hipLaunchKernelGGL(k_compact_flat, dim3(stream_grid(c, 256)), dim3(256), 0, 0, f, pos, c, base,
                               (const uint64_t *)d_hashes + a, (const uint32_t *)d_nodes + a, (const uint64_t *)d_ref_offsets + a,
                               (const float *)d_af32 + a, (uint64_t *)d_out_hashes, (uint32_t *)d_out_nodes,
                               (uint64_t *)d_out_ref_offsets, (float *)d_out_af32);
            if (hipGetLastError() != hipSuccess) rc = gki_set_error(GKI_ERR_HIP, "k_compact_flat launch failed");
        }
        base += kept;
    }
    if (hipDeviceSynchronize() != hipSuccess && rc == GKI_OK) rc = gki_set_error(GKI_ERR_HIP, "gki_compact_flat failed");
    (void)gki_dev_free(wide); (void)gki_dev_free(pos); (void)gki_dev_free(tmp);
    if (rc == GKI_OK) *n_out = base;
    return rc;
}

int gki_column_checksum(const void *d_column, int64_t n, int elem_bytes, uint64_t *sum, uint64_t *xor_fold) {
    *sum = 0; *xor_fold = 0;
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8)
        return gki_set_error(GKI_ERR_BAD_ARG, "elem_bytes must be 1, 2, 4 or 8");
    if (n <= 0) return GKI_OK;
    unsigned long long *d = nullptr;
    HIP_TRY(gki_dev_malloc((void **)&d, 16));
    hipError_t e = hipMemsetAsync(d, 0, 16, 0);
    if (e == hipSuccess) {
        const dim3 grid(stream_grid(n, 256)), block(256);
        if (elem_bytes == 8) hipLaunchKernelGGL(k_checksum<uint64_t>, grid, block, 0, 0, (const uint64_t *)d_column, n, d);
        else if (elem_bytes == 4) hipLaunchKernelGGL(k_checksum<uint32_t>, grid, block, 0, 0, (const uint32_t *)d_column, n, d);
        else if (elem_bytes == 2) hipLaunchKernelGGL(k_checksum<uint16_t>, grid, block, 0, 0, (const uint16_t *)d_column, n, d);
        else hipLaunchKernelGGL(k_checksum<uint8_t>, grid, block, 0, 0, (const uint8_t *)d_column, n, d);
        e = hipGetLastError();
    }
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    (void)gki_dev_free(d);
    HIP_TRY(e);
    *sum = h[0]; *xor_fold = h[1];
    return GKI_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ scan
namespace {
constexpr int SB = 256;        // threads per block
constexpr int SI = 8;          // items per thread
constexpr int STILE = SB * SI; // 2048 items per block

template <typename TAcc>
__device__ __forceinline__ TAcc block_exclusive(TAcc v, TAcc *total, TAcc *lds /* [SB/64 + 1] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TAcc inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        TAcc o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    TAcc wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SB / 64; w++) {
        TAcc s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return wave_off + inc - v;
}

// A thread's SI = 8 consecutive items: two 16-byte loads when they are 4-byte items of a full, aligned tile (the
// element-wise form costs the address unit eight passes over the same lines).
template <typename TIn, typename TAcc>
__device__ __forceinline__ void load_items(const TIn *__restrict__ in, int64_t base, int64_t n, TAcc v[SI]) {
    static_assert(SI == 8, "two uint4 per thread");
    if (sizeof(TIn) == 4 && base + SI <= n && (reinterpret_cast<uintptr_t>(in + base) & 15) == 0) {
        const uint4 a = reinterpret_cast<const uint4 *>(in + base)[0], b = reinterpret_cast<const uint4 *>(in + base)[1];
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < SI; i++) { TIn t; __builtin_memcpy(&t, &w[i], 4); v[i] = (TAcc)t; }
        return;
    }
#pragma unroll
    for (int i = 0; i < SI; i++) v[i] = (base + i < n) ? (TAcc)in[base + i] : (TAcc)0;
}

template <typename TIn, typename TAcc>
__global__ __launch_bounds__(SB) void k_block_sums(const TIn *__restrict__ in, int64_t n, TAcc *__restrict__ sums) {
    __shared__ TAcc lds[SB / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * STILE + (int64_t)threadIdx.x * SI;
    TAcc v[SI];
    load_items<TIn, TAcc>(in, base, n, v);
    TAcc s = 0;
#pragma unroll
    for (int i = 0; i < SI; i++) s += v[i];
    TAcc tot;
    block_exclusive<TAcc>(s, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// exclusive scan with per-block offset; writes out[i] for i < n and out[n] = grand total.
template <typename TIn, typename TAcc, typename TOut>
__global__ __launch_bounds__(SB) void k_block_scan(const TIn *__restrict__ in, int64_t n,
                                                   const TAcc *__restrict__ block_off, TOut *__restrict__ out) {
    __shared__ TAcc lds[SB / 64 + 1];
    int64_t base = (int64_t)blockIdx.x * STILE + (int64_t)threadIdx.x * SI;
    TAcc v[SI];
    load_items<TIn, TAcc>(in, base, n, v);
    TAcc s = 0;
#pragma unroll
    for (int i = 0; i < SI; i++) s += v[i];
    TAcc tot;
    TAcc ex = block_exclusive<TAcc>(s, &tot, lds);
    ex += block_off ? block_off[blockIdx.x] : (TAcc)0;
    if (sizeof(TOut) == 8 && base + SI < n && (reinterpret_cast<uintptr_t>(out + base) & 15) == 0) {
        // a full tile that does not hold the last item: four 16-byte stores
        ulonglong2 *o2 = reinterpret_cast<ulonglong2 *>(out + base);
#pragma unroll
        for (int i = 0; i < SI; i += 2) {
            ulonglong2 p;
            p.x = (unsigned long long)(TOut)ex; ex += v[i];
            p.y = (unsigned long long)(TOut)ex; ex += v[i + 1];
            o2[i / 2] = p;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < SI; i++) {
        if (base + i < n) out[base + i] = (TOut)ex;
        ex += v[i];
        if (base + i == n - 1) out[n] = (TOut)ex;
    }
}

template <typename TIn, typename TAcc, typename TOut>
int scan_impl(const TIn *d_in, int64_t n, TOut *d_out, void *d_tmp, int64_t tmp_bytes, hipStream_t s) {
    if (n <= 0) {
        if (n == 0) HIP_TRY(hipMemsetAsync(d_out, 0, sizeof(TOut), s));
        return GKI_OK;
    }
    int64_t nb = ceil_div(n, STILE);
    if (nb == 1) {
        hipLaunchKernelGGL((k_block_scan<TIn, TAcc, TOut>), dim3(1), dim3(SB), 0, s, d_in, n, (const TAcc *)nullptr, d_out);
        HIP_TRY(hipGetLastError());
        return GKI_OK;
    }
    // tmp: sums[nb] then scanned[nb+1], then the next level's tmp
    int64_t need = (int64_t)sizeof(TAcc) * (2 * nb + 1);
    if (tmp_bytes < need) return gki_set_error(GKI_ERR_BAD_ARG, "scan: tmp too small (%lld < %lld)", (long long)tmp_bytes, (long long)need);
    TAcc *sums = (TAcc *)d_tmp;
    TAcc *scanned = sums + nb;
    hipLaunchKernelGGL((k_block_sums<TIn, TAcc>), dim3((unsigned)nb), dim3(SB), 0, s, d_in, n, sums);
    HIP_TRY(hipGetLastError());
    GKI_TRY((scan_impl<TAcc, TAcc, TAcc>(sums, nb, scanned, (char *)d_tmp + need, tmp_bytes - need, s)));
    hipLaunchKernelGGL((k_block_scan<TIn, TAcc, TOut>), dim3((unsigned)nb), dim3(SB), 0, s, d_in, n, (const TAcc *)scanned, d_out);
    HIP_TRY(hipGetLastError());
    return GKI_OK;
}
}  // namespace

int64_t gki_scan_tmp_bytes(int64_t n) {
    int64_t total = 0;
    while (n > STILE) {
        int64_t nb = ceil_div(n, STILE);
        total += 8 * (2 * nb + 1);
        n = nb;
    }
    return total + 64;
}

int gki_scan_u32_to_i64(const uint32_t *d_in, int64_t n, int64_t *d_out, void *d_tmp, int64_t tmp_bytes, hipStream_t s) {
    return scan_impl<uint32_t, int64_t, int64_t>(d_in, n, d_out, d_tmp, tmp_bytes, s);
}
int gki_scan_i32_to_i64(const int32_t *d_in, int64_t n, int64_t *d_out, void *d_tmp, int64_t tmp_bytes, hipStream_t s) {
    return scan_impl<int32_t, int64_t, int64_t>(d_in, n, d_out, d_tmp, tmp_bytes, s);
}
int gki_scan_u32_to_u32(const uint32_t *d_in, int64_t n, uint32_t *d_out, void *d_tmp, int64_t tmp_bytes, hipStream_t s) {
    return scan_impl<uint32_t, int64_t, uint32_t>(d_in, n, d_out, d_tmp, tmp_bytes, s);
}
