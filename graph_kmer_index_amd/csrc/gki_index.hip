// CollisionFreeKmerIndex on MI355X: bucket = kmer % modulo, records stably sorted by bucket,
// bucket directory (first position + length), per-kmer frequencies, batched probe.
//
// Build = 4 kernels families:
//   k_bucket_keys      bucket of every record (u64 % modulo) + identity permutation
//   radix passes       stable LSD radix sort of (bucket, index) pairs, 8 bits per pass, only the
//                      bits modulo-1 occupies; per pass: tile histograms -> scan -> ranked scatter with an
//                      in-LDS reorder so every digit's run leaves the CU as one contiguous store
//   k_pack_rows / k_gather_rows   the payload packed into 32-byte rows, permuted once by the sorted index
//   k_directory / k_frequencies_*   bucket heads, lengths, distinct-ref_offset counts
// Inside a bucket records keep their input order (stable sort), which is also what the oracle's
// stable restatement produces, so the build is comparable element by element.
#include "gki_common.h"
#include <cstring>
#include <cstdlib>
#include <mutex>

int gki_index_build_rows(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                         uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                         int group_bits, const int64_t *h_group_start, const void *d_rows_in, const void *d_keys_in,
                         void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers, void *d_out_nodes,
                         void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies, void *d_out_permutation, int *done,
                         int by_node);
int gki_partition_columns_by_part(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                                  uint64_t modulo, int n_parts, int sub_bits, int64_t max_rows_per_pass, void *d_out_kmers,
                                  void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32, void *d_out_rows, void *d_out_keys,
                                  int64_t *h_part_start);
int gki_frequencies_for_rows(const int64_t *d_row_begin, const int64_t *d_row_end, int n_ranges, uint64_t modulo,
                             uint64_t bucket_begin, const void *d_hashes_to_index, const void *d_n_kmers,
                             const void *d_kmers, const void *d_refs, void *d_freq, int64_t n, hipStream_t s);

namespace {

constexpr int RB = 256;             // threads per block in the radix kernels
constexpr int RI = 16;              // items per thread
constexpr int RTILE = RB * RI;      // 4096 items per tile
constexpr int RBINS = 256;

__global__ __launch_bounds__(256) void k_bucket_keys(const uint64_t *__restrict__ kmers, int64_t n, uint64_t modulo,
                                                     uint64_t bucket_begin, uint64_t n_buckets,
                                                     uint32_t *__restrict__ keys, uint32_t *__restrict__ idx,
                                                     int *__restrict__ out_of_range) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t b = kmers[i] % modulo - bucket_begin;       // collision_free_kmer_index.py:433
        if (b >= n_buckets) *out_of_range = 1;
        keys[i] = b < n_buckets ? (uint32_t)b : 0u;
        idx[i] = (uint32_t)i;
    }
}

// histogram of one 8-bit digit per tile, stored bin-major: hist[bin * n_tiles + tile]
__global__ __launch_bounds__(RB) void k_radix_hist(const uint32_t *__restrict__ keys, int64_t n, int shift,
                                                   uint32_t *__restrict__ hist, int64_t n_tiles) {
    __shared__ uint32_t h[RBINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RTILE;
#pragma unroll
    for (int r = 0; r < RI; r++) {
        int64_t i = base + r * RB + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one tile.  Element order inside a tile is (wave, round, lane): wave w owns the
// contiguous slice [w*1024, (w+1)*1024) of the tile, round r its r-th group of 64.
__global__ __launch_bounds__(RB) void k_radix_scatter(const uint32_t *__restrict__ keys_in,
                                                      const uint32_t *__restrict__ vals_in, int64_t n, int shift,
                                                      const uint32_t *__restrict__ offs /* scanned hist */,
                                                      int64_t n_tiles, uint32_t *__restrict__ keys_out,
                                                      uint32_t *__restrict__ vals_out) {
    __shared__ uint32_t wave_cnt[4][RBINS];     // per wave, per digit: running count, then exclusive offset
    __shared__ uint32_t digit_start[RBINS];     // start of each digit's run in the locally sorted tile
    __shared__ uint32_t s_keys[RTILE];
    __shared__ uint32_t s_vals[RTILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * RBINS; i += RB) (&wave_cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t tile_base = (int64_t)blockIdx.x * RTILE;
    const int64_t wave_base = tile_base + (int64_t)wave * (RTILE / 4);
    uint32_t key[RI], val[RI], rank[RI];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int64_t i = wave_base + r * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[r] = valid ? vals_in[i] : 0u;
        const uint32_t d = valid ? ((key[r] >> shift) & 0xFF) : 0x100u;     // 0x100: matches no real digit
        // lanes of this wave with the same digit (8 ballots) -> rank among them, group size
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t bit = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bit : ~bit;
        }
        const uint32_t before = (uint32_t)__popcll(same & lt_mask);
        uint32_t prev = 0;
        if (valid) {
            prev = wave_cnt[wave][d];                     // all lanes of the group read the same value ...
        }
        rank[r] = prev + before;
        // ... then the group's first lane publishes the new count (rounds are sequential per wave)
        if (valid && before == 0) wave_cnt[wave][d] = prev + (uint32_t)__popcll(same);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // per digit: exclusive offsets over waves, and the digit's start in the locally sorted tile
    {
        const int d = threadIdx.x;       // RB == RBINS
        uint32_t c0 = wave_cnt[0][d], c1 = wave_cnt[1][d], c2 = wave_cnt[2][d], c3 = wave_cnt[3][d];
        wave_cnt[0][d] = 0; wave_cnt[1][d] = c0; wave_cnt[2][d] = c0 + c1; wave_cnt[3][d] = c0 + c1 + c2;
        const uint32_t tot = c0 + c1 + c2 + c3;
        // exclusive scan of tot over the 256 digits (4 waves of 64)
        const uint32_t inc = gki_wave_incl_sum(tot);
        __shared__ uint32_t wsum[4];
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        digit_start[d] = woff + inc - tot;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int64_t i = wave_base + r * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[r] >> shift) & 0xFF;
            const uint32_t pos = digit_start[d] + wave_cnt[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
    }
    __syncthreads();
    const int64_t n_here = (n - tile_base) < RTILE ? (n - tile_base) : RTILE;
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int p = r * RB + threadIdx.x;
        if (p < n_here) {
            const uint32_t k = s_keys[p];
            const uint32_t d = (k >> shift) & 0xFF;
            const int64_t dst = (int64_t)offs[(int64_t)d * n_tiles + blockIdx.x] + (p - digit_start[d]);
            keys_out[dst] = k;
            vals_out[dst] = s_vals[p];
        }
    }
}

// Payload permutation.  Gathering four columns by a random index costs four random 64-byte sectors per record
// (measured: 30 ms of a 50 ms build at 3.2e8 records).  The columns are first packed into 32-byte rows (one
// streaming pass), so the random access is one sector per record; the permuted rows are unpacked into the output
// columns with coalesced stores.
struct Row { uint64_t kmer, ref; uint32_t node; float af; uint64_t pad; };     // 32 B

__global__ __launch_bounds__(256) void k_pack_rows(const uint64_t *__restrict__ kmers, const uint32_t *__restrict__ nodes,
                                                   const uint64_t *__restrict__ refs, const float *__restrict__ af, int64_t n,
                                                   uint4 *__restrict__ rows) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t k = kmers[i], r = refs[i];
        rows[2 * i] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)r, (uint32_t)(r >> 32));
        rows[2 * i + 1] = make_uint4(nodes[i], __float_as_uint(af[i]), 0u, 0u);
    }
}

__global__ __launch_bounds__(256) void k_gather_rows(const uint32_t *__restrict__ idx, int64_t n, const uint4 *__restrict__ rows,
                                                     uint64_t *__restrict__ o_kmers, uint32_t *__restrict__ o_nodes,
                                                     uint64_t *__restrict__ o_refs, float *__restrict__ o_af) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t j = idx[i];
        const uint4 a = rows[2 * j], b = rows[2 * j + 1];
        o_kmers[i] = ((uint64_t)a.y << 32) | a.x;
        o_refs[i] = ((uint64_t)a.w << 32) | a.z;
        o_nodes[i] = b.x;
        o_af[i] = __uint_as_float(b.y);
    }
}

// end of the run of equal keys starting at i (keys sorted): a short walk, then a binary search -- a k-mer repeated
// millions of times must not cost one lane millions of dependent loads
__device__ __forceinline__ int64_t run_end(const uint32_t *__restrict__ keys, int64_t n, int64_t i, uint32_t b) {
    int64_t e = i + 1;
    const int64_t walk_to = i + 16 < n ? i + 16 : n;
    while (e < walk_to && keys[e] == b) e++;
    if (e < walk_to || e == n) return e;
    int64_t lo = e, hi = n;                       // keys[lo-1] == b; first position with a larger key
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] == b) lo = mid + 1; else hi = mid; }
    return lo;
}

// bucket heads -> directory (collision_free_kmer_index.py:444-457).  A head lane finds the end of its run.
__global__ __launch_bounds__(256) void k_directory(const uint32_t *__restrict__ keys, int64_t n,
                                                   int32_t *__restrict__ hashes_to_index, uint32_t *__restrict__ n_kmers) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t b = keys[i];
        if (i > 0 && keys[i - 1] == b) continue;
        const int64_t e = run_end(keys, n, i, b);
        hashes_to_index[b] = (int32_t)i;
        n_kmers[b] = (uint32_t)(e - i);
    }
}

constexpr int SMALL_BUCKET = 24;

// set_frequencies (collision_free_kmer_index.py:267-293): for every record, the number of distinct
// ref_offsets among the records of its bucket that carry the same k-mer.  Buckets of up to
// SMALL_BUCKET records: one lane per record.  Larger buckets are queued for k_frequencies_large.
__global__ __launch_bounds__(256) void k_frequencies_small(const uint32_t *__restrict__ keys, int64_t n,
                                                           const int32_t *__restrict__ hashes_to_index,
                                                           const uint32_t *__restrict__ n_kmers,
                                                           const uint64_t *__restrict__ kmers, const uint64_t *__restrict__ refs,
                                                           uint16_t *__restrict__ freq, uint32_t *__restrict__ large_list,
                                                           unsigned int *__restrict__ n_large) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t b = keys[i];
        const int64_t s = hashes_to_index[b];
        const int64_t m = n_kmers[b];
        if (m > SMALL_BUCKET) {
            if (i == s) large_list[atomicAdd(n_large, 1u)] = b;
            continue;
        }
        const uint64_t km = kmers[i];
        int count = 0;
        for (int64_t j = s; j < s + m; j++) {
            if (kmers[j] != km) continue;
            const uint64_t r = refs[j];
            bool dup = false;
            for (int64_t c = s; c < j; c++) dup |= (kmers[c] == km && refs[c] == r);
            count += dup ? 0 : 1;
        }
        freq[i] = (uint16_t)count;            // uint16 as in :270
    }
}

struct Pair { uint64_t kmer, ref; };
__device__ __forceinline__ bool pair_less(const Pair &a, const Pair &b) {
    return a.kmer < b.kmer || (a.kmer == b.kmer && a.ref < b.ref);
}

// One block per large bucket: bitonic sort of its (kmer, ref_offset) pairs in a global scratch
// slice (padded to a power of two with max sentinels), then distinct counting per k-mer.
__global__ __launch_bounds__(256) void k_frequencies_large(const uint32_t *__restrict__ large_list, unsigned int n_large,
                                                           const int32_t *__restrict__ hashes_to_index,
                                                           const uint32_t *__restrict__ n_kmers,
                                                           const uint64_t *__restrict__ kmers, const uint64_t *__restrict__ refs,
                                                           const int64_t *__restrict__ scratch_start, Pair *__restrict__ scratch,
                                                           uint16_t *__restrict__ freq) {
    for (unsigned int li = blockIdx.x; li < n_large; li += gridDim.x) {
        const uint32_t b = large_list[li];
        const int64_t s = hashes_to_index[b];
        const int64_t m = n_kmers[b];
        Pair *p = scratch + scratch_start[li];
        int64_t cap = 1;
        while (cap < m) cap <<= 1;
        for (int64_t i = threadIdx.x; i < cap; i += blockDim.x) {
            Pair v;
            if (i < m) { v.kmer = kmers[s + i]; v.ref = refs[s + i]; } else { v.kmer = ~0ull; v.ref = ~0ull; }
            p[i] = v;
        }
        __syncthreads();
        for (int64_t size = 2; size <= cap; size <<= 1) {
            for (int64_t str = size >> 1; str > 0; str >>= 1) {
                for (int64_t t = threadIdx.x; t < (cap >> 1); t += blockDim.x) {
                    const int64_t lo = ((t / str) * (str << 1)) + (t % str);
                    const int64_t hi = lo + str;
                    const bool up = ((lo & size) == 0);
                    Pair a = p[lo], c = p[hi];
                    if (pair_less(c, a) == up) { p[lo] = c; p[hi] = a; }
                }
                __threadfence_block();
                __syncthreads();
            }
        }
        // every record: distinct refs among pairs with its kmer = (first position of kmer .. last), counting ref changes
        for (int64_t i = threadIdx.x; i < m; i += blockDim.x) {
            const uint64_t km = kmers[s + i];
            int64_t lo = 0, hi = m;                       // first pair with kmer >= km
            while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (p[mid].kmer < km) lo = mid + 1; else hi = mid; }
            int64_t count = 0;
            uint64_t prev = 0;
            for (int64_t j = lo; j < m && p[j].kmer == km; j++) {
                if (j == lo || p[j].ref != prev) count++;
                prev = p[j].ref;
            }
            freq[s + i] = (uint16_t)count;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_large_sizes(const uint32_t *__restrict__ large_list, unsigned int n_large,
                                                     const uint32_t *__restrict__ n_kmers, uint32_t *__restrict__ sizes) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_large; i += gridDim.x * blockDim.x) {
        uint32_t m = n_kmers[large_list[i]], cap = 1;
        while (cap < m) cap <<= 1;
        sizes[i] = cap;
    }
}

// ------------------------------------------------------------------------------------ probe
struct IndexDev {
    const int32_t *h2i; const uint32_t *nk; const uint64_t *kmers; const uint32_t *nodes; const uint64_t *refs;
    const uint16_t *freq; const float *af; uint64_t modulo, bucket_begin, n_buckets;
};

// CollisionFreeKmerIndex.get (collision_free_kmer_index.py:303-315) for one query per lane.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_lookup(IndexDev ix, const uint64_t *__restrict__ queries, int64_t q, int64_t max_hits,
                                                uint32_t *__restrict__ cnt, const int64_t *__restrict__ hit_start,
                                                uint32_t *__restrict__ o_nodes, uint64_t *__restrict__ o_refs,
                                                int64_t *__restrict__ o_query, uint16_t *__restrict__ o_freq,
                                                float *__restrict__ o_af, int64_t *__restrict__ o_pos) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += stride) {
        const uint64_t km = queries[i];
        const uint64_t b = km % ix.modulo - ix.bucket_begin;          // :304 (slice-relative)
        const bool mine = b < ix.n_buckets;
        const int64_t s = mine ? ix.h2i[b] : 0;                       // :305
        const int64_t m = mine ? ix.nk[b] : 0;                        // :306
        if (!EMIT) {
            uint32_t c = 0;
            bool first = true, too_frequent = false;
            for (int64_t j = s; j < s + m; j++) {
                if (ix.kmers[j] != km) continue;                      // :309
                if (first) { too_frequent = ix.freq && (int64_t)ix.freq[j] > max_hits; first = false; }   // :312
                c++;
            }
            cnt[i] = too_frequent ? 0u : c;
        } else {
            int64_t o = hit_start[i];
            if (hit_start[i + 1] == o) continue;
            for (int64_t j = s; j < s + m; j++) {
                if (ix.kmers[j] != km) continue;
                if (o_nodes) o_nodes[o] = ix.nodes[j];                // :315
                if (o_refs) o_refs[o] = ix.refs[j];
                if (o_query) o_query[o] = i;                          // read_offsets = query index (:365)
                if (o_freq) o_freq[o] = ix.freq ? ix.freq[j] : (uint16_t)0;
                if (o_af) o_af[o] = ix.af[j];
                if (o_pos) o_pos[o] = j;
                o++;
            }
        }
    }
}

// kmer_mapper-style node counting fused with the probe (collision_free_kmer_index.py:210-212 map_kmers,
// CounterKmerIndex.get_node_counts :39-40): counts[node] += 1 for every hit of every query.
__global__ __launch_bounds__(256) void k_count_nodes(IndexDev ix, const uint64_t *__restrict__ queries, int64_t q, int64_t max_hits,
                                                     unsigned int *__restrict__ counts, int64_t n_counts) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += stride) {
        const uint64_t km = queries[i];
        const uint64_t b = km % ix.modulo - ix.bucket_begin;
        if (b >= ix.n_buckets) continue;
        const int64_t s = ix.h2i[b];
        const int64_t m = ix.nk[b];
        bool first = true;
        for (int64_t j = s; j < s + m; j++) {
            if (ix.kmers[j] != km) continue;
            if (first) { if (ix.freq && (int64_t)ix.freq[j] > max_hits) break; first = false; }
            const uint32_t node = ix.nodes[j];
            if ((int64_t)node < n_counts) atomicAdd(&counts[node], 1u);
        }
    }
}


static int key_bits(uint64_t max_key) {
    int bits = 0;
    while (bits < 32 && (max_key >> bits) != 0) bits++;
    return bits;
}

// Stable LSD sort of (key, value) pairs on the low `bits` bits of the key; ping-pongs between the two
// buffers of each pair, *cur_out = index of the buffers holding the result.
static int radix_sort_pairs(uint32_t *keys[2], uint32_t *vals[2], int64_t n, int bits, uint32_t *hist, uint32_t *offs,
                            void *tmp, int64_t tmp_bytes, hipStream_t s, int *cur_out) {
    const int64_t n_tiles = ceil_div(n, RTILE);
    const int64_t hist_n = (int64_t)RBINS * n_tiles;
    int cur = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3((unsigned)n_tiles), dim3(RB), 0, s, keys[cur], n, shift, hist, n_tiles);
        HIP_TRY(hipGetLastError());
        GKI_TRY(gki_scan_u32_to_u32(hist, hist_n, offs, tmp, tmp_bytes, s));
        hipLaunchKernelGGL(k_radix_scatter, dim3((unsigned)n_tiles), dim3(RB), 0, s, keys[cur], vals[cur], n, shift, offs,
                           n_tiles, keys[1 - cur], vals[1 - cur]);
        HIP_TRY(hipGetLastError());
        cur = 1 - cur;
    }
    *cur_out = cur;
    return GKI_OK;
}

// FlatKmers.get_new_without_singletons (flat_kmers.py:98-125): a record is kept iff an EARLIER record carries the same
// hash.  After the stable sort by bucket the records of a bucket are in input order, so "earlier" = an earlier position
// of the bucket's run; the flag goes back to the record's original position.
__global__ __launch_bounds__(256) void k_flag_repeats(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ perm,
                                                      const uint64_t *__restrict__ kmers, int64_t n,
                                                      const int32_t *__restrict__ first_of_bucket, uint8_t *__restrict__ flags) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const uint32_t me = perm[p];
        const uint64_t km = kmers[me];
        bool seen = false;
        for (int64_t j = first_of_bucket[keys[p]]; j < p && !seen; j++) seen = kmers[perm[j]] == km;
        flags[me] = seen ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------ reverse index
// ReverseKmerIndex.from_flat_kmers (reverse_kmer_index.py:47-60): records stably sorted by node,
// nodes_to_index_positions[node] = first record (uint32), nodes_to_n_hashes[node] = run length (uint16, wraps
// like the NumPy assignment at :56).
// n_hashes of the reverse index is uint16 (the NumPy assignment at reverse_kmer_index.py:56 stores the count modulo 2^16)
__global__ __launch_bounds__(256) void k_narrow_counts(const uint32_t *__restrict__ in, int64_t n, uint16_t *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint16_t)in[i];
}

__global__ __launch_bounds__(256) void k_node_keys(const uint32_t *__restrict__ nodes, int64_t n, uint64_t n_nodes, uint32_t *__restrict__ keys,
                                                   uint32_t *__restrict__ idx, int *__restrict__ bad) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t v = nodes[i];
        if ((uint64_t)v >= n_nodes) { *bad = 1; v = (uint32_t)(n_nodes - 1); }      // (the directory is never indexed out of range)
        keys[i] = v; idx[i] = (uint32_t)i;
    }
}

// Node directory of the reverse index, two streaming passes and no search: PASS 0 -- the head of a run records where the
// node's records start; PASS 1 (launched after it) -- the tail of a run records how many there are.  A node has tens of
// records, so "the head lane finds the end of its run" (run_end: 16 steps, then a binary search over the rest of the
// array, ~28 dependent loads across 1.2 GB) took 12.5 ms of the 27 ms build on 3.1e8 records.
template <int PASS>
__global__ __launch_bounds__(256) void k_node_directory(const uint32_t *__restrict__ keys, int64_t n,
                                                        uint32_t *__restrict__ first, uint16_t *__restrict__ count) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t b = keys[i];
        if (PASS == 0) {
            if (i == 0 || keys[i - 1] != b) first[b] = (uint32_t)i;
        } else {
            if (i == n - 1 || keys[i + 1] != b) count[b] = (uint16_t)(i + 1 - (int64_t)first[b]);   // mod 2^16, reverse_kmer_index.py:56
        }
    }
}

__global__ __launch_bounds__(256) void k_pack_pairs(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, int64_t n,
                                                    uint4 *__restrict__ rows) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t x = a[i], y = b[i];
        rows[i] = make_uint4((uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32));
    }
}

__global__ __launch_bounds__(256) void k_gather_pairs(const uint32_t *__restrict__ idx, int64_t n, const uint4 *__restrict__ rows,
                                                      uint64_t *__restrict__ o_a, uint64_t *__restrict__ o_b) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint4 r = rows[idx[i]];
        o_a[i] = ((uint64_t)r.y << 32) | r.x;
        o_b[i] = ((uint64_t)r.w << 32) | r.z;
    }
}

}  // namespace

// Frequencies of the buckets in large_list (more than SMALL_BUCKET records each): sizes -> scratch offsets -> one block
// per bucket.  `sizes` is scratch for n_large uint32.
static int frequencies_large_pass(const uint32_t *large_list, unsigned int n_large, uint32_t *sizes, const void *d_hashes_to_index,
                                  const void *d_n_kmers, const void *d_kmers, const void *d_refs, void *d_freq, hipStream_t s) {
    int64_t *starts = nullptr;
    Pair *scratch = nullptr;
    void *tmp2 = nullptr;
    hipLaunchKernelGGL(k_large_sizes, dim3(stream_grid(n_large, 256)), dim3(256), 0, s, large_list, n_large,
                       (const uint32_t *)d_n_kmers, sizes);
    HIP_TRY(hipGetLastError());
    const int64_t tmp2_bytes = gki_scan_tmp_bytes(n_large);
    int r = GKI_OK;
    if (gki_dev_malloc((void **)&starts, ((size_t)n_large + 1) * 8) != hipSuccess) r = GKI_ERR_HIP;
    if (r == GKI_OK && gki_dev_malloc(&tmp2, (size_t)tmp2_bytes) != hipSuccess) r = GKI_ERR_HIP;
    if (r == GKI_OK) r = gki_scan_u32_to_i64(sizes, n_large, starts, tmp2, tmp2_bytes, s);
    int64_t total = 0;
    if (r == GKI_OK && hipMemcpy(&total, starts + n_large, 8, hipMemcpyDeviceToHost) != hipSuccess) r = GKI_ERR_HIP;
    if (r == GKI_OK && gki_dev_malloc((void **)&scratch, (size_t)total * sizeof(Pair)) != hipSuccess) r = GKI_ERR_HIP;
    if (r == GKI_OK) {
        const unsigned grid = n_large < 2048 ? n_large : 2048;
        hipLaunchKernelGGL(k_frequencies_large, dim3(grid), dim3(256), 0, s, large_list, n_large, (const int32_t *)d_hashes_to_index,
                           (const uint32_t *)d_n_kmers, (const uint64_t *)d_kmers, (const uint64_t *)d_refs, starts, scratch,
                           (uint16_t *)d_freq);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) r = GKI_ERR_HIP;
    }
    (void)gki_dev_free(starts); (void)gki_dev_free(tmp2); (void)gki_dev_free(scratch);
    if (r != GKI_OK) return gki_set_error(r, "large-bucket frequency pass failed");
    return GKI_OK;
}

// The frequency pass of the row-carrying form (gki_index_rows.hip) for the rows it left open: ranges [begin, end) of the
// finished columns, each a whole number of buckets.  Same rule as k_frequencies_small, the bucket recomputed from the
// k-mer.
__global__ __launch_bounds__(256) void k_frequencies_ranges(const int64_t *__restrict__ rb, const int64_t *__restrict__ re, int n_ranges,
                                                            uint64_t modulo, uint64_t bucket_begin,
                                                            const int32_t *__restrict__ hashes_to_index, const uint32_t *__restrict__ n_kmers,
                                                            const uint64_t *__restrict__ kmers, const uint64_t *__restrict__ refs,
                                                            uint16_t *__restrict__ freq, uint32_t *__restrict__ large_list,
                                                            unsigned int *__restrict__ n_large) {
    for (int q = blockIdx.x; q < n_ranges; q += gridDim.x) {
        for (int64_t i = rb[q] + threadIdx.x; i < re[q]; i += blockDim.x) {
            const uint64_t km = kmers[i];
            const uint64_t b = km % modulo - bucket_begin;
            const int64_t s = hashes_to_index[b];
            const int64_t m = n_kmers[b];
            if (m > SMALL_BUCKET) {
                if (i == s) large_list[atomicAdd(n_large, 1u)] = (uint32_t)b;
                continue;
            }
            int count = 0;
            for (int64_t j = s; j < s + m; j++) {
                if (kmers[j] != km) continue;
                const uint64_t r = refs[j];
                bool dup = false;
                for (int64_t c = s; c < j; c++) dup |= (kmers[c] == km && refs[c] == r);
                count += dup ? 0 : 1;
            }
            freq[i] = (uint16_t)count;
        }
    }
}

int gki_frequencies_for_rows(const int64_t *d_row_begin, const int64_t *d_row_end, int n_ranges, uint64_t modulo,
                             uint64_t bucket_begin, const void *d_hashes_to_index, const void *d_n_kmers, const void *d_kmers,
                             const void *d_refs, void *d_freq, int64_t n, hipStream_t s) {
    if (n_ranges <= 0) return GKI_OK;
    uint32_t *large_list = nullptr, *sizes = nullptr;
    unsigned int *n_large_d = nullptr;
    const size_t cap = (size_t)(n / SMALL_BUCKET + 1);
    int rc = GKI_OK;
    if (gki_dev_malloc((void **)&large_list, cap * 4) != hipSuccess || gki_dev_malloc((void **)&sizes, cap * 4) != hipSuccess ||
        gki_dev_malloc((void **)&n_large_d, 16) != hipSuccess || hipMemsetAsync(n_large_d, 0, 4, s) != hipSuccess)
        rc = gki_set_error(GKI_ERR_HIP, "frequency pass: allocation failed");
    unsigned int n_large = 0;
    if (rc == GKI_OK) {
        hipLaunchKernelGGL(k_frequencies_ranges, dim3(n_ranges < 4096 ? n_ranges : 4096), dim3(256), 0, s, d_row_begin, d_row_end,
                           n_ranges, modulo, bucket_begin, (const int32_t *)d_hashes_to_index, (const uint32_t *)d_n_kmers,
                           (const uint64_t *)d_kmers, (const uint64_t *)d_refs, (uint16_t *)d_freq, large_list, n_large_d);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&n_large, n_large_d, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            rc = gki_set_error(GKI_ERR_HIP, "frequency pass over open rows failed");
    }
    if (rc == GKI_OK && n_large > 0)
        rc = frequencies_large_pass(large_list, n_large, sizes, d_hashes_to_index, d_n_kmers, d_kmers, d_refs, d_freq, s);
    (void)gki_dev_free(large_list); (void)gki_dev_free(sizes); (void)gki_dev_free(n_large_d);
    return rc;
}

extern "C" {

int gki_index_build_pairs(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                          uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                          void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers, void *d_out_nodes,
                          void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies, void *d_out_permutation) {
    if (modulo == 0 || modulo > 0xFFFFFFFFull) return gki_set_error(GKI_ERR_BAD_ARG, "modulo must be in 1..2^32-1");
    if (n_buckets == 0 || bucket_begin + n_buckets > modulo)
        return gki_set_error(GKI_ERR_BAD_ARG, "bucket range [%llu, +%llu) outside [0, modulo)", (unsigned long long)bucket_begin,
                             (unsigned long long)n_buckets);
    if (n >= (1ll << 31))
        return gki_set_error(GKI_ERR_OVERFLOW, "%lld records: the reference's directory is int32 "
                             "(collision_free_kmer_index.py:453); shard the build", (long long)n);
    hipStream_t s = 0;
    HIP_TRY(hipMemsetAsync(d_hashes_to_index, 0, (size_t)n_buckets * 4, s));       // :453
    HIP_TRY(hipMemsetAsync(d_n_kmers, 0, (size_t)n_buckets * 4, s));               // :456
    if (n <= 0) { HIP_TRY(hipStreamSynchronize(s)); return GKI_OK; }
    HIP_TRY(hipMemsetAsync(d_out_frequencies, 0, (size_t)n * 2, s));               // :270
    const int64_t n_tiles = ceil_div(n, RTILE);
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *hist = nullptr, *offs = nullptr;
    uint4 *rows = nullptr;
    void *tmp = nullptr;
    const int64_t hist_n = (int64_t)RBINS * n_tiles;
    const int64_t tmp_bytes = gki_scan_tmp_bytes(hist_n);
    int rc = GKI_OK;
#define CLEANUP_RETURN(code) do { rc = (code); goto done; } while (0)
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        for (int i = 0; i < 2; i++) {
            HIP_G(gki_dev_malloc((void **)&keys[i], (size_t)n * 4));
            HIP_G(gki_dev_malloc((void **)&vals[i], (size_t)n * 4));
        }
        HIP_G(gki_dev_malloc((void **)&hist, (size_t)hist_n * 4));
        HIP_G(gki_dev_malloc((void **)&offs, (size_t)(hist_n + 1) * 4));
        HIP_G(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
        int *bad = (int *)hist;                       // hist is not in use yet
        HIP_G(hipMemsetAsync(bad, 0, 4, s));
        hipLaunchKernelGGL(k_bucket_keys, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const uint64_t *)d_kmers, n, modulo,
                           bucket_begin, n_buckets, keys[0], vals[0], bad);
        HIP_G(hipGetLastError());
        if (n_buckets != modulo) {
            int h_bad = 0;
            HIP_G(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            if (h_bad) CLEANUP_RETURN(gki_set_error(GKI_ERR_BAD_ARG, "a record's bucket lies outside [%llu, +%llu)",
                                                    (unsigned long long)bucket_begin, (unsigned long long)n_buckets));
        }
        int cur = 0;
        {
            int r = radix_sort_pairs(keys, vals, n, key_bits(n_buckets - 1), hist, offs, tmp, tmp_bytes, s, &cur);
            if (r != GKI_OK) CLEANUP_RETURN(r);
        }
        HIP_G(gki_dev_malloc((void **)&rows, (size_t)n * 32));
        hipLaunchKernelGGL(k_pack_rows, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const uint64_t *)d_kmers,
                           (const uint32_t *)d_nodes, (const uint64_t *)d_ref_offsets, (const float *)d_af32, n, rows);
        HIP_G(hipGetLastError());
        hipLaunchKernelGGL(k_gather_rows, dim3(stream_grid(n, 256)), dim3(256), 0, s, vals[cur], n, (const uint4 *)rows,
                           (uint64_t *)d_out_kmers, (uint32_t *)d_out_nodes, (uint64_t *)d_out_ref_offsets, (float *)d_out_af32);
        HIP_G(hipGetLastError());
        if (d_out_permutation) HIP_G(hipMemcpyAsync(d_out_permutation, vals[cur], (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_directory, dim3(stream_grid(n, 256)), dim3(256), 0, s, keys[cur], n, (int32_t *)d_hashes_to_index,
                           (uint32_t *)d_n_kmers);
        HIP_G(hipGetLastError());
        if (!skip_frequencies) {
            // large-bucket work list: at most n / SMALL_BUCKET entries; reuse the spare key/val buffers
            uint32_t *large_list = keys[1 - cur];
            unsigned int *n_large_d = (unsigned int *)hist;
            HIP_G(hipMemsetAsync(n_large_d, 0, 4, s));
            hipLaunchKernelGGL(k_frequencies_small, dim3(stream_grid(n, 256)), dim3(256), 0, s, keys[cur], n,
                               (const int32_t *)d_hashes_to_index, (const uint32_t *)d_n_kmers, (const uint64_t *)d_out_kmers,
                               (const uint64_t *)d_out_ref_offsets, (uint16_t *)d_out_frequencies, large_list, n_large_d);
            HIP_G(hipGetLastError());
            unsigned int n_large = 0;
            HIP_G(hipMemcpyAsync(&n_large, n_large_d, 4, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            if (n_large > 0) {
                int r = frequencies_large_pass(large_list, n_large, vals[1 - cur], d_hashes_to_index, d_n_kmers, d_out_kmers,
                                               d_out_ref_offsets, d_out_frequencies, s);
                if (r != GKI_OK) CLEANUP_RETURN(r);
            }
        }
        HIP_G(hipStreamSynchronize(s));
    }
done:
    for (int i = 0; i < 2; i++) { (void)gki_dev_free(keys[i]); (void)gki_dev_free(vals[i]); }
    (void)gki_dev_free(hist); (void)gki_dev_free(offs); (void)gki_dev_free(tmp); (void)gki_dev_free(rows);
#undef HIP_G
#undef CLEANUP_RETURN
    return rc;
}

int gki_index_build_range_grouped(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                                  uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                                  int group_bits, const int64_t *h_group_start,
                                  void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers, void *d_out_nodes,
                                  void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies, void *d_out_permutation) {
    if (modulo == 0 || modulo > 0xFFFFFFFFull) return gki_set_error(GKI_ERR_BAD_ARG, "modulo must be in 1..2^32-1");
    if (n_buckets == 0 || bucket_begin + n_buckets > modulo)
        return gki_set_error(GKI_ERR_BAD_ARG, "bucket range [%llu, +%llu) outside [0, modulo)", (unsigned long long)bucket_begin,
                             (unsigned long long)n_buckets);
    if (group_bits < 0 || group_bits > 10 || (group_bits > 0 && !h_group_start))
        return gki_set_error(GKI_ERR_BAD_ARG, "group_bits must be in 0..10, with the group bounds when > 0");
    if (n >= (1ll << 31))
        return gki_set_error(GKI_ERR_OVERFLOW, "%lld records: the reference's directory is int32 "
                             "(collision_free_kmer_index.py:453); shard the build", (long long)n);
    if (n > 0) {
        // the row-carrying form (gki_index_rows.hip); the pair-sorting form below takes over for inputs outside its domain
        // (it sorts on the whole key: a grouping of the input is of no use to it and does no harm)
        int done = 0;
        GKI_TRY(gki_index_build_rows(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, bucket_begin, n_buckets, skip_frequencies,
                                     group_bits, h_group_start, nullptr, nullptr, d_hashes_to_index, d_n_kmers, d_out_kmers, d_out_nodes,
                                     d_out_ref_offsets, d_out_af32, d_out_frequencies, d_out_permutation, &done, 0));
        if (done) return GKI_OK;
    }
    return gki_index_build_pairs(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, bucket_begin, n_buckets, skip_frequencies,
                                 d_hashes_to_index, d_n_kmers, d_out_kmers, d_out_nodes, d_out_ref_offsets, d_out_af32,
                                 d_out_frequencies, d_out_permutation);
}

int gki_index_build_range(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                          uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                          void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers, void *d_out_nodes,
                          void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies, void *d_out_permutation) {
    return gki_index_build_range_grouped(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, bucket_begin, n_buckets, skip_frequencies,
                                         0, nullptr, d_hashes_to_index, d_n_kmers, d_out_kmers, d_out_nodes, d_out_ref_offsets,
                                         d_out_af32, d_out_frequencies, d_out_permutation);
}

int gki_index_build(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                    uint64_t modulo, int skip_frequencies, void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers,
                    void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies,
                    void *d_out_permutation) {
    return gki_index_build_range(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, 0, modulo, skip_frequencies,
                                 d_hashes_to_index, d_n_kmers, d_out_kmers, d_out_nodes, d_out_ref_offsets, d_out_af32,
                                 d_out_frequencies, d_out_permutation);
}

int gki_partition_by_bucket_range_grouped(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                                          int64_t n, uint64_t modulo, int n_parts, int group_bits, int64_t max_rows_per_pass,
                                          void *d_out_kmers, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                                          int64_t *h_start) {
    if (modulo == 0 || modulo > 0xFFFFFFFFull) return gki_set_error(GKI_ERR_BAD_ARG, "modulo must be in 1..2^32-1");
    if (n_parts < 1 || n_parts > 256) return gki_set_error(GKI_ERR_BAD_ARG, "n_parts must be in 1..256");
    if (group_bits < 0 || (n_parts << group_bits) > 1024) return gki_set_error(GKI_ERR_BAD_ARG, "n_parts << group_bits must not exceed 1024");
    for (int p = 0; p <= (n_parts << group_bits); p++) h_start[p] = 0;
    if (n <= 0) return GKI_OK;
    // stable passes of the row-carrying build's partition kernel, columns in, columns out (gki_index_rows.hip)
    return gki_partition_columns_by_part(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, n_parts, group_bits, max_rows_per_pass,
                                         d_out_kmers, d_out_nodes, d_out_ref_offsets, d_out_af32, nullptr, nullptr, h_start);
}

int gki_partition_rows_by_bucket_range(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                                       int64_t n, uint64_t modulo, int n_parts, int group_bits, int64_t max_rows_per_pass,
                                       void *d_rows, void *d_keys, int64_t *h_start) {
    if (modulo == 0 || modulo > 0xFFFFFFFFull) return gki_set_error(GKI_ERR_BAD_ARG, "modulo must be in 1..2^32-1");
    if (n_parts < 1 || n_parts > 256) return gki_set_error(GKI_ERR_BAD_ARG, "n_parts must be in 1..256");
    if (group_bits < 0 || (n_parts << group_bits) > 1024) return gki_set_error(GKI_ERR_BAD_ARG, "n_parts << group_bits must not exceed 1024");
    if (!d_rows || !d_keys) return gki_set_error(GKI_ERR_BAD_ARG, "rows and keys buffers are needed");
    for (int p = 0; p <= (n_parts << group_bits); p++) h_start[p] = 0;
    if (n <= 0) return GKI_OK;
    return gki_partition_columns_by_part(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, n_parts, group_bits, max_rows_per_pass,
                                         nullptr, nullptr, nullptr, nullptr, d_rows, d_keys, h_start);
}

int gki_index_build_range_from_rows(const void *d_rows, const void *d_keys, int64_t n, uint64_t modulo, uint64_t bucket_begin,
                                    uint64_t n_buckets, int skip_frequencies, int group_bits, const int64_t *h_group_start,
                                    void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers, void *d_out_nodes,
                                    void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies) {
    if (modulo == 0 || modulo > 0xFFFFFFFFull) return gki_set_error(GKI_ERR_BAD_ARG, "modulo must be in 1..2^32-1");
    if (n_buckets == 0 || bucket_begin + n_buckets > modulo)
        return gki_set_error(GKI_ERR_BAD_ARG, "bucket range [%llu, +%llu) outside [0, modulo)", (unsigned long long)bucket_begin,
                             (unsigned long long)n_buckets);
    if (group_bits < 0 || group_bits > 10 || (group_bits > 0 && !h_group_start))
        return gki_set_error(GKI_ERR_BAD_ARG, "group_bits must be in 0..10, with the group bounds when > 0");
    if (n >= (1ll << 31))
        return gki_set_error(GKI_ERR_OVERFLOW, "%lld records: the reference's directory is int32 "
                             "(collision_free_kmer_index.py:453); shard the build", (long long)n);
    hipStream_t s = 0;
    if (n <= 0) {
        HIP_TRY(hipMemsetAsync(d_hashes_to_index, 0, (size_t)n_buckets * 4, s));       // :453
        HIP_TRY(hipMemsetAsync(d_n_kmers, 0, (size_t)n_buckets * 4, s));               // :456
        HIP_TRY(hipStreamSynchronize(s));
        return GKI_OK;
    }
    if (!d_rows || !d_keys) return gki_set_error(GKI_ERR_BAD_ARG, "rows and keys are needed");
    int done = 0;
    GKI_TRY(gki_index_build_rows(nullptr, nullptr, nullptr, nullptr, n, modulo, bucket_begin, n_buckets, skip_frequencies, group_bits,
                                 h_group_start, d_rows, d_keys, d_hashes_to_index, d_n_kmers, d_out_kmers, d_out_nodes,
                                 d_out_ref_offsets, d_out_af32, d_out_frequencies, nullptr, &done, 0));
    if (!done)
        return gki_set_error(GKI_ERR_OUT_OF_DOMAIN, "the records are outside the row-carrying build's domain (a group of neighbouring buckets with "
                             "more than 2^22 records): build this slice from its columns (gki_index_build_range)");
    return GKI_OK;
}

int gki_partition_by_bucket_range_chunked(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                                          int64_t n, uint64_t modulo, int n_parts, int64_t max_rows_per_pass, void *d_out_kmers,
                                          void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32, int64_t *h_part_start) {
    return gki_partition_by_bucket_range_grouped(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, n_parts, 0, max_rows_per_pass,
                                                 d_out_kmers, d_out_nodes, d_out_ref_offsets, d_out_af32, h_part_start);
}

int gki_partition_by_bucket_range(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                                  int64_t n, uint64_t modulo, int n_parts, void *d_out_kmers, void *d_out_nodes,
                                  void *d_out_ref_offsets, void *d_out_af32, int64_t *h_part_start) {
    return gki_partition_by_bucket_range_grouped(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, n_parts, 0, 0, d_out_kmers,
                                                 d_out_nodes, d_out_ref_offsets, d_out_af32, h_part_start);
}

int gki_flag_repeated_kmers(const void *d_kmers, int64_t n, void *d_flags) {
    if (n <= 0) return GKI_OK;
    if (n >= (1ll << 31)) return gki_set_error(GKI_ERR_OVERFLOW, "%lld records: at most 2^31-1 at a time", (long long)n);
    // a bucket table about twice as large as the input keeps the runs short; its size only has to be odd-ish
    uint64_t modulo = (uint64_t)n * 2 + 1;
    if (modulo > 0xFFFFFFFBull) modulo = 0xFFFFFFFBull;
    hipStream_t s = 0;
    const int64_t hist_n = (int64_t)RBINS * ceil_div(n, RTILE);
    const int64_t tmp_bytes = gki_scan_tmp_bytes(hist_n);
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *hist = nullptr, *offs = nullptr, *cnt = nullptr;
    int32_t *first = nullptr;
    void *tmp = nullptr;
    int rc = GKI_OK;
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        for (int i = 0; i < 2; i++) {
            HIP_G(gki_dev_malloc((void **)&keys[i], (size_t)n * 4));
            HIP_G(gki_dev_malloc((void **)&vals[i], (size_t)n * 4));
        }
        HIP_G(gki_dev_malloc((void **)&hist, (size_t)hist_n * 4));
        HIP_G(gki_dev_malloc((void **)&offs, (size_t)(hist_n + 1) * 4));
        HIP_G(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
        HIP_G(gki_dev_malloc((void **)&first, (size_t)modulo * 4));
        HIP_G(gki_dev_malloc((void **)&cnt, (size_t)modulo * 4));
        int *bad = (int *)hist;
        hipLaunchKernelGGL(k_bucket_keys, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const uint64_t *)d_kmers, n, modulo,
                           (uint64_t)0, modulo, keys[0], vals[0], bad);
        HIP_G(hipGetLastError());
        int cur = 0;
        rc = radix_sort_pairs(keys, vals, n, key_bits(modulo - 1), hist, offs, tmp, tmp_bytes, s, &cur);
        if (rc != GKI_OK) goto done;
        hipLaunchKernelGGL(k_directory, dim3(stream_grid(n, 256)), dim3(256), 0, s, keys[cur], n, first, cnt);
        HIP_G(hipGetLastError());
        hipLaunchKernelGGL(k_flag_repeats, dim3(stream_grid(n, 256)), dim3(256), 0, s, keys[cur], vals[cur],
                           (const uint64_t *)d_kmers, n, first, (uint8_t *)d_flags);
        HIP_G(hipGetLastError());
        HIP_G(hipStreamSynchronize(s));
    }
done:
    for (int i = 0; i < 2; i++) { (void)gki_dev_free(keys[i]); (void)gki_dev_free(vals[i]); }
    (void)gki_dev_free(hist); (void)gki_dev_free(offs); (void)gki_dev_free(tmp); (void)gki_dev_free(first); (void)gki_dev_free(cnt);
#undef HIP_G
    return rc;
}

int gki_reverse_index_build(const void *d_nodes, const void *d_kmers, const void *d_ref_offsets, int64_t n, int64_t n_nodes,
                            void *d_index_positions, void *d_n_hashes, void *d_out_kmers, void *d_out_ref_offsets) {
    if (n_nodes <= 0 || n_nodes > (1ll << 32)) return gki_set_error(GKI_ERR_BAD_ARG, "n_nodes must be in 1..2^32");
    if (n >= (1ll << 32)) return gki_set_error(GKI_ERR_OVERFLOW, "%lld records do not fit the uint32 directory", (long long)n);
    hipStream_t s = 0;
    // GKI_REVERSE_FORM=pairs (read per call): the pair-sorting form, for the parity tests of the path behind the row form
    const char *form = getenv("GKI_REVERSE_FORM");
    const bool pairs_only = form && strcmp(form, "pairs") == 0;
    if (!pairs_only && n > 0 && n < (1ll << 31) && n_nodes < (1ll << 32)) {
        // the row-carrying form (gki_index_rows.hip) with the node id as the key: the payload travels with its key through the
        // staged partition passes and the in-LDS finish leaves the node directory as it goes -- no gather, no separate
        // directory passes (round 4; the pair-sorting form below stays for what lies outside its domain)
        uint32_t *nk32 = nullptr;
        HIP_TRY(gki_dev_malloc((void **)&nk32, (size_t)n_nodes * 4));
        int done = 0;
        int rc = gki_index_build_rows(d_kmers, d_nodes, d_ref_offsets, nullptr, n, 1, 0, (uint64_t)n_nodes, 1, 0, nullptr, nullptr, nullptr,
                                      d_index_positions, nk32, d_out_kmers, nullptr, d_out_ref_offsets, nullptr, nullptr, nullptr, &done, 1);
        if (rc == GKI_OK && done) {
            hipLaunchKernelGGL(k_narrow_counts, dim3(stream_grid(n_nodes, 256)), dim3(256), 0, s, nk32, n_nodes, (uint16_t *)d_n_hashes);
            hipError_t e = hipGetLastError();
            hipError_t e2 = hipStreamSynchronize(s);
            (void)gki_dev_free(nk32);
            HIP_TRY(e); HIP_TRY(e2);
            return GKI_OK;
        }
        (void)gki_dev_free(nk32);
        if (rc != GKI_OK) return rc;
    }
    HIP_TRY(hipMemsetAsync(d_index_positions, 0, (size_t)n_nodes * 4, s));      // reverse_kmer_index.py:53
    HIP_TRY(hipMemsetAsync(d_n_hashes, 0, (size_t)n_nodes * 2, s));             // :54
    if (n <= 0) { HIP_TRY(hipStreamSynchronize(s)); return GKI_OK; }
    const int64_t hist_n = (int64_t)RBINS * ceil_div(n, RTILE);
    const int64_t tmp_bytes = gki_scan_tmp_bytes(hist_n);
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *hist = nullptr, *offs = nullptr;
    uint4 *rows = nullptr;
    void *tmp = nullptr;
    int *bad = nullptr;
    int rc = GKI_OK;
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        for (int i = 0; i < 2; i++) {
            HIP_G(gki_dev_malloc((void **)&keys[i], (size_t)n * 4));
            HIP_G(gki_dev_malloc((void **)&vals[i], (size_t)n * 4));
        }
        HIP_G(gki_dev_malloc((void **)&hist, (size_t)hist_n * 4));
        HIP_G(gki_dev_malloc((void **)&offs, (size_t)(hist_n + 1) * 4));
        HIP_G(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
        HIP_G(gki_dev_malloc((void **)&rows, (size_t)n * 16));
        HIP_G(gki_dev_malloc((void **)&bad, 4));
        HIP_G(hipMemsetAsync(bad, 0, 4, s));
        hipLaunchKernelGGL(k_node_keys, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const uint32_t *)d_nodes, n, (uint64_t)n_nodes, keys[0], vals[0], bad);
        HIP_G(hipGetLastError());
        int cur = 0;
        rc = radix_sort_pairs(keys, vals, n, key_bits((uint64_t)n_nodes - 1), hist, offs, tmp, tmp_bytes, s, &cur);
        if (rc != GKI_OK) goto done;
        hipLaunchKernelGGL(k_pack_pairs, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const uint64_t *)d_kmers,
                           (const uint64_t *)d_ref_offsets, n, rows);
        HIP_G(hipGetLastError());
        hipLaunchKernelGGL(k_gather_pairs, dim3(stream_grid(n, 256)), dim3(256), 0, s, vals[cur], n, (const uint4 *)rows,
                           (uint64_t *)d_out_kmers, (uint64_t *)d_out_ref_offsets);
        HIP_G(hipGetLastError());
        hipLaunchKernelGGL(k_node_directory<0>, dim3(stream_grid(n, 256)), dim3(256), 0, s, keys[cur], n,
                           (uint32_t *)d_index_positions, (uint16_t *)d_n_hashes);
        HIP_G(hipGetLastError());
        hipLaunchKernelGGL(k_node_directory<1>, dim3(stream_grid(n, 256)), dim3(256), 0, s, keys[cur], n,
                           (uint32_t *)d_index_positions, (uint16_t *)d_n_hashes);
        HIP_G(hipGetLastError());
        int h_bad = 0;
        HIP_G(hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s));
        HIP_G(hipStreamSynchronize(s));
        if (h_bad) rc = gki_set_error(GKI_ERR_BAD_ARG, "a record's node id is not below n_nodes = %lld", (long long)n_nodes);
    }
done:
    for (int i = 0; i < 2; i++) { (void)gki_dev_free(keys[i]); (void)gki_dev_free(vals[i]); }
    (void)gki_dev_free(hist); (void)gki_dev_free(offs); (void)gki_dev_free(tmp); (void)gki_dev_free(rows); (void)gki_dev_free(bad);
#undef HIP_G
    return rc;
}

static IndexDev view_of(const gki_index_view *ix) {
    IndexDev d;
    d.h2i = (const int32_t *)ix->d_hashes_to_index; d.nk = (const uint32_t *)ix->d_n_kmers;
    d.kmers = (const uint64_t *)ix->d_kmers; d.nodes = (const uint32_t *)ix->d_nodes;
    d.refs = (const uint64_t *)ix->d_ref_offsets; d.freq = (const uint16_t *)ix->d_frequencies;
    d.af = (const float *)ix->d_af32; d.modulo = ix->modulo;
    d.bucket_begin = ix->n_buckets ? ix->bucket_begin : 0; d.n_buckets = ix->n_buckets ? ix->n_buckets : ix->modulo;
    return d;
}

int gki_index_count_nodes(const gki_index_view *ix, const void *d_queries, int64_t q, int64_t max_hits, void *d_counts,
                          int64_t n_counts) {
    if (ix->modulo == 0) return gki_set_error(GKI_ERR_BAD_ARG, "modulo is 0");
    if (q <= 0) return GKI_OK;
    hipLaunchKernelGGL(k_count_nodes, dim3(stream_grid(q, 256)), dim3(256), 0, 0, view_of(ix), (const uint64_t *)d_queries, q,
                       max_hits, (unsigned int *)d_counts, n_counts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return GKI_OK;
}

int gki_index_lookup_count(const gki_index_view *ix, const void *d_queries, int64_t q, int64_t max_hits, void *d_hit_start,
                           int64_t *n_hits) {
    *n_hits = 0;
    if (ix->modulo == 0) return gki_set_error(GKI_ERR_BAD_ARG, "modulo is 0");
    if (q <= 0) { HIP_TRY(hipMemset(d_hit_start, 0, 8)); return GKI_OK; }
    uint32_t *cnt = nullptr;
    void *tmp = nullptr;
    int64_t tmp_bytes = gki_scan_tmp_bytes(q);
    HIP_TRY(gki_dev_malloc((void **)&cnt, (size_t)q * 4));
    HIP_TRY(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
    hipLaunchKernelGGL(k_lookup<false>, dim3(stream_grid(q, 256)), dim3(256), 0, 0, view_of(ix), (const uint64_t *)d_queries, q,
                       max_hits, cnt, (const int64_t *)nullptr, (uint32_t *)nullptr, (uint64_t *)nullptr, (int64_t *)nullptr,
                       (uint16_t *)nullptr, (float *)nullptr, (int64_t *)nullptr);
    int rc = hipGetLastError() == hipSuccess ? GKI_OK : gki_set_error(GKI_ERR_HIP, "k_lookup launch failed");
    if (rc == GKI_OK) rc = gki_scan_u32_to_i64(cnt, q, (int64_t *)d_hit_start, tmp, tmp_bytes, 0);
    int64_t total = 0;
    hipError_t e = hipMemcpy(&total, (const int64_t *)d_hit_start + q, 8, hipMemcpyDeviceToHost);
    (void)gki_dev_free(cnt); (void)gki_dev_free(tmp);
    if (rc != GKI_OK) return rc;
    HIP_TRY(e);
    *n_hits = total;
    return GKI_OK;
}

int gki_index_lookup_emit(const gki_index_view *ix, const void *d_queries, int64_t q, int64_t max_hits, const void *d_hit_start,
                          void *d_hit_nodes, void *d_hit_ref_offsets, void *d_hit_query, void *d_hit_frequencies,
                          void *d_hit_af32, void *d_hit_position) {
    if (q <= 0) return GKI_OK;
    hipLaunchKernelGGL(k_lookup<true>, dim3(stream_grid(q, 256)), dim3(256), 0, 0, view_of(ix), (const uint64_t *)d_queries, q,
                       max_hits, (uint32_t *)nullptr, (const int64_t *)d_hit_start, (uint32_t *)d_hit_nodes,
                       (uint64_t *)d_hit_ref_offsets, (int64_t *)d_hit_query, (uint16_t *)d_hit_frequencies, (float *)d_hit_af32,
                       (int64_t *)d_hit_position);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    return GKI_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------ scalar get
// CollisionFreeKmerIndex.get (collision_free_kmer_index.py:303-315) for a handful of k-mers: one launch, one wave per
// query, queries and results in pinned host memory the kernel reads and writes directly -- no staging copies, no
// second pass.  A query's hits (positions in the payload arrays, bucket order) land in its own `cap`-sized segment.
namespace {
constexpr int SMALL_Q = 64;                 // queries per call
constexpr int SMALL_CAP = 1024;             // hits kept per query (more: the count says so and the caller goes batched)
struct SmallStage {                         // pinned, device-mapped
    uint64_t query[SMALL_Q];
    int64_t n_hits[SMALL_Q];
    int64_t pos[SMALL_Q * SMALL_CAP];
};
__global__ __launch_bounds__(64) void k_get_small(IndexDev ix, SmallStage *st, int64_t max_hits) {
    const int qi = blockIdx.x, lane = threadIdx.x;
    const uint64_t km = st->query[qi];
    const uint64_t b = km % ix.modulo - ix.bucket_begin;                  // :304
    int64_t total = 0;
    if (b < ix.n_buckets) {
        const int64_t s = ix.h2i[b], m = ix.nk[b];                        // :305-306
        bool first = true;
        for (int64_t j0 = 0; j0 < m; j0 += 64) {
            const int64_t j = s + j0 + lane;
            const bool hit = j0 + lane < m && ix.kmers[j] == km;          // :309
            const uint64_t mask = __ballot(hit);
            if (mask == 0) continue;
            if (first) {                                                  // :312 frequency of the FIRST hit
                const int64_t jf = s + j0 + (__ffsll((unsigned long long)mask) - 1);
                if (ix.freq && (int64_t)ix.freq[jf] > max_hits) { total = 0; break; }
                first = false;
            }
            if (hit) {
                const int64_t o = total + __popcll(mask & ((1ull << lane) - 1ull));
                if (o < SMALL_CAP) st->pos[(int64_t)qi * SMALL_CAP + o] = j;
            }
            total += __popcll(mask);
        }
    }
    if (lane == 0) st->n_hits[qi] = total;
}
std::mutex g_small_mu;
SmallStage *g_small_host = nullptr;         // pinned + portable: every device maps it; its device address is asked for per call.
                                            // 0.5 MB for the life of the process: freeing it from a static destructor would call
                                            // into a HIP runtime that may already be gone
}  // namespace

extern "C" int gki_index_get_small(const gki_index_view *ix, const uint64_t *h_queries, int q, int64_t max_hits,
                                   int64_t *h_n_hits, int64_t *h_positions, int64_t capacity_per_query) {
    if (q < 0 || q > SMALL_Q) return gki_set_error(GKI_ERR_BAD_ARG, "get_small: 0..%d queries per call", SMALL_Q);
    if (capacity_per_query < 0) return gki_set_error(GKI_ERR_BAD_ARG, "get_small: negative capacity");
    if (q == 0) return GKI_OK;
    std::lock_guard<std::mutex> lock(g_small_mu);
    if (!g_small_host) HIP_TRY(hipHostMalloc((void **)&g_small_host, sizeof(SmallStage), hipHostMallocMapped | hipHostMallocPortable));
    SmallStage *g_small_dev = nullptr;              // the mapping of the device that is current NOW (a process may switch devices)
    HIP_TRY(hipHostGetDevicePointer((void **)&g_small_dev, g_small_host, 0));
    for (int i = 0; i < q; i++) g_small_host->query[i] = h_queries[i];
    hipLaunchKernelGGL(k_get_small, dim3((unsigned)q), dim3(64), 0, 0, view_of(ix), g_small_dev, max_hits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(0));
    for (int i = 0; i < q; i++) {
        const int64_t n = g_small_host->n_hits[i];
        h_n_hits[i] = n;                              // may exceed what was kept: min(n, SMALL_CAP, capacity) positions are valid
        int64_t keep = n < SMALL_CAP ? n : SMALL_CAP;
        if (keep > capacity_per_query) keep = capacity_per_query;
        for (int64_t j = 0; j < keep; j++) h_positions[(int64_t)i * capacity_per_query + j] = g_small_host->pos[(int64_t)i * SMALL_CAP + j];
    }
    return GKI_OK;
}
