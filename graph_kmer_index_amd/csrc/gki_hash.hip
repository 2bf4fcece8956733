// Hash kernels outside the graph walk: k-windows of a plain sequence (A1/A4), read k-mers (A10),
// reverse complement / complement of hashes (A8).
#include "gki_common.h"

namespace {

__global__ __launch_bounds__(256) void k_hash_windows(const uint64_t *__restrict__ seq2, int64_t n_out, int k,
                                                      uint64_t *__restrict__ out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride)
        out[i] = gki_extract(seq2, i, k);
}

// rc(x) = digit-reverse(~x) >> (64 - 2k)   (kmer_hashing.py:24-28 restated on bits; SURVEY.md 8a')
__device__ __forceinline__ uint64_t revcomp_bits(uint64_t x, int k) {
    uint64_t y = __brevll(~x);
    y = ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
    return y >> (64 - 2 * k);
}

template <bool REVERSE>
__global__ __launch_bounds__(256) void k_complement(const uint64_t *__restrict__ in, int64_t n, int k,
                                                    uint64_t *__restrict__ out) {
    const uint64_t mask = (1ull << (2 * k)) - 1ull;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t x = in[i];
        out[i] = REVERSE ? revcomp_bits(x & mask, k) : (~x & mask);
    }
}

__global__ __launch_bounds__(256) void k_read_counts(const int64_t *__restrict__ read_start, int64_t n_reads, int k,
                                                     uint32_t *__restrict__ cnt) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += stride) {
        int64_t len = read_start[r + 1] - read_start[r];
        cnt[r] = len >= k ? (uint32_t)(len - k + 1) : 0u;
    }
}

// 31 low bits -> even bit positions (Morton spread)
__device__ __forceinline__ uint64_t spread31(uint64_t x) {
    x &= 0x7FFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// One wave per read.  64 letters at a time are turned into two wave-uniform bit planes with
// __ballot (low / high bit of the 2-bit code); a k-window is then two 31-bit fields of the planes,
// interleaved.  strand 1 walks the read backwards and complements ACGT (other letters stay 0),
// i.e. hashes str(Seq(read).reverse_complement()) (read_kmers.py:23-26).
template <int STRAND>
__global__ __launch_bounds__(256) void k_hash_reads(const uint8_t *__restrict__ reads,
                                                    const int64_t *__restrict__ read_start, int64_t n_reads, int k,
                                                    const int64_t *__restrict__ out_start,
                                                    uint64_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t kmask = (1ull << k) - 1ull;
    for (int64_t r = wave; r < n_reads; r += n_waves) {
        const int64_t s = read_start[r];
        const int64_t len = read_start[r + 1] - s;
        if (len < k) continue;
        const int64_t n_out = len - k + 1;
        uint64_t *o = out + out_start[r];
        uint64_t lo_cur = 0, hi_cur = 0;
        // chunk c holds bases [64c, 64c+64) of the (possibly reverse-complemented) read
        for (int64_t c = 0; c * 64 < len + 64; c++) {
            const int64_t i = c * 64 + lane;
            unsigned code = 0;
            if (i < len) {
                unsigned ch = reads[STRAND ? s + len - 1 - i : s + i] | 0x20u;
                unsigned fwd = ch == 'c' ? 1u : ch == 'g' ? 2u : ch == 't' ? 3u : 0u;
                bool acgt = ch == 'a' || fwd != 0u;
                code = STRAND ? (acgt ? 3u - fwd : 0u) : fwd;
            }
            const uint64_t lo_next = __ballot(code & 1u);
            const uint64_t hi_next = __ballot(code & 2u);
            if (c > 0) {
                // outputs j = 64(c-1) + lane need bases j .. j+k-1: bits lane.. of (cur, next)
                const int64_t j = (c - 1) * 64 + lane;
                if (j < n_out) {
                    uint64_t l = lane ? (lo_cur >> lane) | (lo_next << (64 - lane)) : lo_cur;
                    uint64_t h = lane ? (hi_cur >> lane) | (hi_next << (64 - lane)) : hi_cur;
                    o[j] = spread31(l & kmask) | (spread31(h & kmask) << 1);
                }
            }
            lo_cur = lo_next; hi_cur = hi_next;
        }
    }
}

}  // namespace

extern "C" {

int gki_hash_sequence(const void *d_codes, int64_t n, int k, void *d_out) {
    if (k < 1 || k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31");
    if (n < k) return GKI_OK;
    void *seq2 = nullptr;
    int64_t n_u64 = ceil_div(n, 32) + 2;
    HIP_TRY(gki_dev_malloc(&seq2, (size_t)n_u64 * 8));
    HIP_TRY(hipMemsetAsync(seq2, 0, (size_t)n_u64 * 8, 0));
    int rc = gki_launch_pack((const uint8_t *)d_codes, n, (uint32_t *)seq2, 0);
    if (rc == GKI_OK) {
        int64_t n_out = n - k + 1;
        hipLaunchKernelGGL(k_hash_windows, dim3(stream_grid(n_out, 256)), dim3(256), 0, 0, (const uint64_t *)seq2, n_out, k,
                           (uint64_t *)d_out);
        if (hipGetLastError() != hipSuccess) rc = gki_set_error(GKI_ERR_HIP, "k_hash_windows launch failed");
    }
    hipError_t e = hipStreamSynchronize(0);
    (void)gki_dev_free(seq2);
    if (rc != GKI_OK) return rc;
    HIP_TRY(e);
    return GKI_OK;
}

int gki_hash_reads(const void *d_reads, const void *d_read_start, int64_t n_reads, int k, int strand,
                   void *d_out_start, void *d_out, int64_t out_capacity, int64_t *n_out) {
    *n_out = 0;
    if (k < 1 || k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31");
    if (strand != 0 && strand != 1) return gki_set_error(GKI_ERR_BAD_ARG, "strand must be 0 or 1");
    if (n_reads <= 0) { HIP_TRY(hipMemset(d_out_start, 0, 8)); return GKI_OK; }
    void *cnt = nullptr, *tmp = nullptr;
    int64_t tmp_bytes = gki_scan_tmp_bytes(n_reads);
    HIP_TRY(gki_dev_malloc(&cnt, (size_t)n_reads * 4));
    HIP_TRY(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
    hipLaunchKernelGGL(k_read_counts, dim3(stream_grid(n_reads, 256)), dim3(256), 0, 0, (const int64_t *)d_read_start,
                       n_reads, k, (uint32_t *)cnt);
    int rc = gki_scan_u32_to_i64((const uint32_t *)cnt, n_reads, (int64_t *)d_out_start, tmp, tmp_bytes, 0);
    int64_t total = 0;
    hipError_t e = hipMemcpy(&total, (const int64_t *)d_out_start + n_reads, 8, hipMemcpyDeviceToHost);
    (void)gki_dev_free(cnt);
    (void)gki_dev_free(tmp);
    if (rc != GKI_OK) return rc;
    HIP_TRY(e);
    *n_out = total;
    if (d_out == nullptr) return GKI_OK;               // count only
    if (total > out_capacity) return gki_set_error(GKI_ERR_BAD_ARG, "hash_reads: output needs %lld entries, capacity %lld",
                                                   (long long)total, (long long)out_capacity);
    int64_t blocks = ceil_div(n_reads, 4);
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (strand == 0)
        hipLaunchKernelGGL(k_hash_reads<0>, dim3((unsigned)blocks), dim3(256), 0, 0, (const uint8_t *)d_reads,
                           (const int64_t *)d_read_start, n_reads, k, (const int64_t *)d_out_start, (uint64_t *)d_out);
    else
        hipLaunchKernelGGL(k_hash_reads<1>, dim3((unsigned)blocks), dim3(256), 0, 0, (const uint8_t *)d_reads,
                           (const int64_t *)d_read_start, n_reads, k, (const int64_t *)d_out_start, (uint64_t *)d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return GKI_OK;
}

int gki_reverse_complement(const void *d_in, int64_t n, int k, void *d_out) {
    if (k < 1 || k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31");   // kmer_hashing.py:25
    if (n <= 0) return GKI_OK;
    hipLaunchKernelGGL(k_complement<true>, dim3(stream_grid(n, 256)), dim3(256), 0, 0, (const uint64_t *)d_in, n, k,
                       (uint64_t *)d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return GKI_OK;
}

int gki_complement(const void *d_in, int64_t n, int k, void *d_out) {
    if (k < 1 || k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31");
    if (n <= 0) return GKI_OK;
    hipLaunchKernelGGL(k_complement<false>, dim3(stream_grid(n, 256)), dim3(256), 0, 0, (const uint64_t *)d_in, n, k,
                       (uint64_t *)d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return GKI_OK;
}

}  // extern "C"
