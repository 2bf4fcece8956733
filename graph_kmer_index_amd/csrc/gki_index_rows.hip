// CollisionFreeKmerIndex build, row-carrying form (collision_free_kmer_index.py:423-467, set_frequencies :267-293).
//
// The first form of the build (gki_index.hip) sorts (bucket, index) pairs and then GATHERS the payload by the sorted
// index: 3.1e8 random 32-byte rows at the chip's lane-request rate, 12 of its 31 ms.  This form never issues a random
// request: the 24-byte payload row travels WITH its key through stable partition passes whose writes leave a tile as
// contiguous runs per digit, and the last step sorts a group of a few hundred to a few thousand rows inside LDS.
//
//   per partition pass (1..3, most significant digits last = LSD over the TOP bits of the key, each pass stable):
//     k_kmer_digit_hist (first)     key = kmer % modulo - bucket_begin, per-tile digit histogram   8 R (the key is not stored:
//                                   the first pass computes it again from the k-mer it carries anyway)
//     k_digit_hist (later passes)   per-tile digit histogram                                  4 R
//     scan                          bin-major exclusive scan -> first row of every (digit, tile) run
//     k_partition_rows_staged       the four input columns -> rows + keys (first)             24 R + 28 W
//                                   rows + keys -> rows + keys (later)                        28 R + 28 W
//                                   (4096-row tile, ranked at once, staged through LDS in four parts: two workgroups per CU;
//                                   k_partition_rows -- the tile sorted whole in LDS -- serves the bucket-range partition
//                                   into columns and is the A/B partner, -DGKI_PT_NCH=1)
//   k_group_bounds / k_group_scan   first row / row count of every group (= key >> L, L <= 10) 4 R
//   k_group_finish                  one workgroup per group: counting sort on the low L (<= 10) bits in LDS, the directory of
//                                   the group's 2^L buckets (streamed, no memset + scatter), frequencies, and the
//                                   four output columns + frequency column, all coalesced      28 R + 26 W + 8 B/bucket
//   k_group_large                   a group with more rows than LDS holds: one workgroup streams it (count, scan,
//                                   ordered scatter); frequencies of its buckets by the kernels of the first form
//
//
// Bucket-range partition (gki_partition_by_bucket_range*, gki_partition_rows_by_bucket_range): one pass of the same kernels
// with the owning part (and optionally the top bits of the key inside the part: "grouped") as the digit; the grouped build
// (gki_index_build_range_grouped / _from_rows) then sorts every group on its own in ONE segmented pass (TileDesc) and
// finishes 4096-row groups (k_group_finish<.., 4096, 1024>).
//
// Stability: every pass keeps equal digits in input order and the in-LDS sort ranks equal buckets by row position, so
// the result equals the first form's (and the oracle's stable build) element by element.
#include "gki_common.h"
#include <math.h>
#include <vector>

int gki_frequencies_for_rows(const int64_t *d_row_begin, const int64_t *d_row_end, int n_ranges, uint64_t modulo,
                             uint64_t bucket_begin, const void *d_hashes_to_index, const void *d_n_kmers,
                             const void *d_kmers, const void *d_refs, void *d_freq, int64_t n, hipStream_t s);

namespace {

#ifndef GKI_MAXB_BITS
#define GKI_MAXB_BITS 10
#endif
constexpr int MAXB_BITS = GKI_MAXB_BITS;
constexpr int MAXB = 1 << MAXB_BITS;          // digits of one partition pass
// Shape of the finish: 1024 rows and 2^10 buckets per group, 512 threads -- 38 KB of LDS, four workgroups (32 waves) per CU.
// Measured against 2048 rows / 2^11 buckets (two workgroups per CU) on 3.1e8 records, same box: the finish 6.13 -> 4.49 ms,
// the first partition pass (10 bits instead of 9: runs of 4 rows) 5.03 -> 5.75 ms, the build 19.05 -> 18.5 ms
// (profiles/r03_index_group_shape_ab.txt).
#ifndef GKI_GROUP_CAP
#define GKI_GROUP_CAP 1024
#endif
#ifndef GKI_GROUP_LMAX
#define GKI_GROUP_LMAX 10
#endif
#ifndef GKI_GROUP_THREADS
#define GKI_GROUP_THREADS 512
#endif
constexpr int GROUP_CAP = GKI_GROUP_CAP;          // rows a group may hold to be finished in LDS
// ... and in the grouped build (records arriving grouped by the top bits of their key, gki_index_build_range_grouped): one
// workgroup per CU finishes 4096 rows, two more key bits than the 1024-row finish resolves, so that ONE partition pass is
// left between the grouping and the finish (DESIGN.md 4.3 "Grouped build")
constexpr int GROUP_CAP_BIG = 4096;
constexpr int GROUP_THREADS_BIG = 1024;           // sixteen waves: the one workgroup a CU holds has to hide its own latencies
constexpr int GROUP_LMAX = GKI_GROUP_LMAX;        // low key bits resolved in LDS
constexpr int GROUP_THREADS = GKI_GROUP_THREADS;
constexpr int SMALL_BUCKET = 24;              // as in gki_index.hip: buckets up to this size count frequencies per lane

// How a record's sort key follows from its k-mer: the bucket relative to the slice (the index build), or the part that owns
// the bucket (gki_partition_by_bucket_range: part p owns buckets [modulo * p / n_parts, modulo * (p + 1) / n_parts)).
struct KeyRule {
    GkiMod mod;
    uint64_t bucket_begin, n_buckets;       // build: key = kmer % modulo - bucket_begin, must be < n_buckets
    int n_parts;                            // > 0: key = owning part; part_begin[n_parts + 1] on the device
    float parts_per_bucket;                 //   n_parts / modulo: the part of bucket b is b * this, give or take one
    int parts_log2;                         //   >= 0: n_parts = 2^this, and part_begin[p] = (modulo * p) >> this needs no table
    int by_node;                            // the key is the record's node id (ReverseKmerIndex): no k-mer arithmetic at all
    int sub_bits;                           // > 0 (with parts): key = part << sub_bits | the top sub_bits bits of the bucket's
    const uint32_t *part_begin;             //   offset in its part, i.e. (bucket - part_begin[part]) >> sub_shift[part]
    const uint32_t *sub_shift;              //   [n_parts] on the device
};
constexpr int MAX_PARTS = 256;
constexpr int PB_WORDS = 2 * MAX_PARTS + 1; // LDS copy: part_begin[0 .. n_parts], then sub_shift[0 .. n_parts)

// the part table into LDS (PART rule only); callers synchronise before the first key_of
__device__ __forceinline__ void stage_parts(const KeyRule &k, uint32_t *s_pb) {
    if (k.n_parts > 0) {
        for (int p = threadIdx.x; p <= k.n_parts; p += blockDim.x) s_pb[p] = k.part_begin[p];
        if (k.sub_bits > 0) for (int p = threadIdx.x; p < k.n_parts; p += blockDim.x) s_pb[MAX_PARTS + 1 + p] = k.sub_shift[p];
    }
}
// The key the passes sort on (returned) and the key that travels with the row (*stored): the same for the build; for the
// part rule the former is the part (with its group), the latter the bucket's offset in its part -- the key of that part's
// slice build.  *bad: the bucket lies outside the slice (the build refuses such input); the key is then 0, never out of range
__device__ __forceinline__ uint32_t key_of(const KeyRule &k, const uint32_t *s_pb, uint64_t kmer, bool *bad, uint32_t *stored) {
    const uint64_t b = gki_mod(k.mod, kmer);                            // collision_free_kmer_index.py:433
    if (k.n_parts > 0) {
        int p = (int)((float)(uint32_t)b * k.parts_per_bucket);         // floor(b * n_parts / modulo), give or take one (24-bit
        p = p < k.n_parts ? p : k.n_parts - 1;                          // mantissa: three instructions where the exact quotient
                                                                        // by multiply-high took sixteen)
        if (k.parts_log2 >= 0) {
            // a power of two of parts (the usual 8): the part bounds by a multiply and a shift, the group shift by a count
            // of leading zeros -- no table in LDS (six LDS reads per record in a kernel that lives on LDS bandwidth)
            uint32_t lo = (uint32_t)((k.mod.m * (uint64_t)p) >> k.parts_log2), hi = (uint32_t)((k.mod.m * (uint64_t)(p + 1)) >> k.parts_log2);
            if ((uint32_t)b < lo) { p--; hi = lo; lo = (uint32_t)((k.mod.m * (uint64_t)p) >> k.parts_log2); }
            else if ((uint32_t)b >= hi && p + 1 < k.n_parts) { p++; lo = hi; hi = (uint32_t)((k.mod.m * (uint64_t)(p + 1)) >> k.parts_log2); }
            *stored = (uint32_t)b - lo;
            if (k.sub_bits > 0) {
                const int kbp = hi - lo > 1u ? 32 - __clz((int)(hi - lo - 1u)) : 0;       // bits of the part's largest key
                return ((uint32_t)p << k.sub_bits) | (*stored >> (kbp > k.sub_bits ? kbp - k.sub_bits : 0));
            }
            return (uint32_t)p;
        }
        while (p + 1 < k.n_parts && s_pb[p + 1] <= (uint32_t)b) p++;
        while (p > 0 && s_pb[p] > (uint32_t)b) p--;
        *stored = (uint32_t)b - s_pb[p];
        if (k.sub_bits > 0) return ((uint32_t)p << k.sub_bits) | (*stored >> s_pb[MAX_PARTS + 1 + p]);
        return (uint32_t)p;
    }
    const uint64_t rel = b - k.bucket_begin;
    if (rel >= k.n_buckets) { *bad = true; *stored = 0u; return 0u; }
    *stored = (uint32_t)rel;
    return (uint32_t)rel;
}

// Tiles of a SEGMENTED pass (the grouped build: the records arrive grouped by the top bits of their key and every group is
// sorted on its own, so no tile may straddle two groups): where the tile's rows are and where its histogram column is.
// The histogram of such a pass is laid out group by group, inside a group digit-major like the plain one, so that ONE
// exclusive scan over all of it still yields every run's first output row.
struct TileDesc { int64_t row0, tiles_before; int32_t n, stride, t, pad; };   // histogram column: (tiles_before * bins + t) + digit * stride

// the descriptors of one group's tiles: seg = (first row, one past the last row, tiles of the groups before) per group
__global__ __launch_bounds__(256) void k_tile_descs(const int64_t *__restrict__ seg, int tile_rows, TileDesc *__restrict__ out) {
    const int64_t a0 = seg[3 * blockIdx.x], a1 = seg[3 * blockIdx.x + 1], before = seg[3 * blockIdx.x + 2];
    const int64_t nt = (a1 - a0 + tile_rows - 1) / tile_rows;
    for (int64_t t = threadIdx.x; t < nt; t += 256) {
        TileDesc td;
        td.row0 = a0 + t * tile_rows; td.n = (int32_t)((a1 - td.row0) < tile_rows ? (a1 - td.row0) : tile_rows);
        td.tiles_before = before; td.t = (int32_t)t; td.stride = (int32_t)nt; td.pad = 0;
        out[before + t] = td;
    }
}

struct PartArgs {
    const TileDesc *tiles;         // NULL: tile t is rows [t * TILE, +TILE), its histogram column offs[d * n_tiles + t]
    const uint32_t *keys_in;                                                                          // !SRC_COLS
    const uint64_t *c_kmers; const uint32_t *c_nodes; const uint64_t *c_refs; const uint32_t *c_af;   // SRC_COLS: the key follows
    KeyRule rule;                                                                                     //   from the k-mer by `rule`
    const uint64_t *rows_in;                                                                          // !SRC_COLS
    int64_t n, n_tiles;
    int shift, bits;
    const uint32_t *offs;          // [bins * n_tiles] exclusive scan of the bin-major tile histograms
    uint64_t *rows_out; uint32_t *keys_out;
    uint64_t *o_kmers; uint32_t *o_nodes; uint64_t *o_refs; uint32_t *o_af;   // DST_COLS: the sorted tile leaves as four columns,
    const int64_t *dbase;          //   digit d's runs at rows dbase[d] + (offs - offs of the digit's first run): the caller lays the
                                   //   parts of several chunks out behind each other (more than 2^31 records in all)
    int carry_index;               // the row's input index rides in place of the allele frequency (permutation wanted)
    int xcd_tiles;                 // > 0: block b works on tile (b % 8) * xcd_tiles + b / 8, so that neighbouring tiles
                                   // (whose runs are adjacent in memory) go through the same XCD's L2
};

__device__ __forceinline__ int64_t tile_of_block(int64_t n_tiles, int xcd_tiles) {
    if (xcd_tiles <= 0) return blockIdx.x;
    return (int64_t)(blockIdx.x & 7) * xcd_tiles + (blockIdx.x >> 3);
}

// A tile's digit histogram into hist[digit * n_tiles + tile] (bin-major: the exclusive scan of that array is every run's
// first output row).  Blocks take their tiles XCD-major like the partition kernel: a digit's entries of consecutive tiles
// share 32-byte sectors, and with tiles dealt round-robin over the eight XCDs every sector was written 4 bytes at a time
// through eight L2s that do not merge each other's partial lines.
// Few digits (the bucket-range partition: <= 16 parts): every lane counts its own records in packed 8-bit counters (a shift
// and an add per record), the wave sums a digit's counters by a DPP scan, one LDS add per wave and digit -- LDS atomics of
// 4096 rows on 8 addresses serialise, and a ballot per record and digit is 48 instructions per record (the histogram of
// the 3.16e9-record partition ran at 1.6 TB/s: ALU-bound).  RI <= 255.
struct FewDigits { uint64_t lo, hi; };                // digits 0-7, 8-15
__device__ __forceinline__ void few_add(FewDigits &c, uint32_t dig, bool valid) {
    const uint64_t one = valid ? 1ull << ((dig & 7u) * 8u) : 0ull;
    if (dig & 8u) c.hi += one; else c.lo += one;
}
__device__ __forceinline__ void few_flush(const FewDigits &c, uint32_t *h, int bins) {
    const int lane = threadIdx.x & 63;
    for (int d = 0; d < bins; d++) {
        const uint32_t v = (uint32_t)(((d & 8) ? c.hi : c.lo) >> ((d & 7) * 8)) & 0xFFu;
        const uint32_t tot = (uint32_t)gki_lane_value((int)gki_wave_incl_sum(v), 63);
        if (lane == 0 && tot) atomicAdd(&h[d], tot);
    }
}

template <int THREADS, int RI>
__global__ __launch_bounds__(THREADS) void k_digit_hist(const uint32_t *__restrict__ keys, int64_t n, int shift, int bits,
                                                        uint32_t *__restrict__ hist, int64_t n_tiles, int xcd_tiles,
                                                        const TileDesc *__restrict__ tiles) {
    __shared__ uint32_t h[MAXB];
    const int64_t tile = tile_of_block(n_tiles, xcd_tiles);
    if (tile >= n_tiles) return;
    const int bins = 1 << bits;
    for (int d = threadIdx.x; d < bins; d += THREADS) h[d] = 0;
    __syncthreads();
    int64_t base = tile * (THREADS * RI), obase = tile, ostride = n_tiles;
    int64_t n_here = n - base < THREADS * RI ? n - base : THREADS * RI;
    if (tiles) { const TileDesc td = tiles[tile]; base = td.row0; n_here = td.n; obase = td.tiles_before * bins + td.t; ostride = td.stride; }
    const uint32_t mask = (uint32_t)bins - 1u;
    uint32_t key[RI];
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int e = r * THREADS + threadIdx.x;
        key[r] = e < n_here ? keys[base + e] : 0u;
    }
    if (bits <= 4) {
        FewDigits c = {0ull, 0ull};
#pragma unroll
        for (int r = 0; r < RI; r++) few_add(c, (key[r] >> shift) & mask, r * THREADS + threadIdx.x < n_here);
        few_flush(c, h, bins);
    } else {
#pragma unroll
        for (int r = 0; r < RI; r++) if (r * THREADS + threadIdx.x < n_here) atomicAdd(&h[(key[r] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += THREADS) hist[obase + (int64_t)d * ostride] = h[d];
}

// Exclusive scan of v over the THREADS threads of the block (lds: THREADS / 64 + 1 words)
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl(uint32_t v, uint32_t *lds, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = gki_wave_incl_sum(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) { const uint32_t c = lds[w]; if (w < wave) woff += c; tot += c; }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// Rank of every element among the earlier elements of its wave with the same digit, and the wave's digit counts.
// Element order inside a tile is (wave, round, lane): wave w owns a contiguous slice, round r its r-th group of 64.
template <int RI>
__device__ __forceinline__ void wave_rank(const uint32_t (&dig)[RI], const bool (&valid)[RI], int bits, uint16_t *wcnt /* this wave's [bins] */,
                                          uint32_t (&rank)[RI]) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const uint32_t d = dig[r];
        uint64_t same = __ballot(valid[r]);
        for (int b = 0; b < bits; b++) {
            const uint64_t bit = __ballot((d >> b) & 1u);
            same &= ((d >> b) & 1u) ? bit : ~bit;
        }
        const uint32_t before = (uint32_t)__popcll(same & lt_mask);
        uint32_t prev = 0;
        if (valid[r]) prev = wcnt[d];                       // all lanes of the group read the same value ...
        rank[r] = prev + before;
        if (valid[r] && before == 0) wcnt[d] = (uint16_t)(prev + (uint32_t)__popcll(same));   // ... then its first lane publishes
        __builtin_amdgcn_wave_barrier();
    }
}

// One stable partition pass over a tile of THREADS * RI rows.
// SRC_COLS: the tile comes from the four columns, every thread loading whole rows in ranking order (coalesced per column) and
// computing their keys from the k-mers (no key array read); otherwise rows + keys of the previous pass, the payload words
// loaded by position (coalesced) and dropped into their rows' slots through a 16-bit destination table.
template <int THREADS, int RI, bool SRC_COLS, bool DST_COLS = false>
__global__ __launch_bounds__(THREADS) void k_partition_rows(PartArgs a) {
    constexpr int TILE = THREADS * RI, W = THREADS / 64, SLICE = TILE / W;
    __shared__ __attribute__((aligned(16))) uint64_t s_rows[TILE * 3];
    __shared__ uint32_t s_keys[TILE];
    __shared__ uint16_t s_dest[SRC_COLS ? 1 : TILE];
    // the per-wave digit counts live under the staging rows: dead (a barrier ago) before the first payload word lands
    static_assert((size_t)TILE * 24 >= (size_t)W * MAXB * 2, "the per-wave digit counts fit under the rows");
    uint16_t (*const s_wcnt)[MAXB] = reinterpret_cast<uint16_t (*)[MAXB]>(s_rows);     // [W][MAXB]
    __shared__ uint32_t s_dstart[MAXB];
    __shared__ uint32_t s_toff[MAXB];
    __shared__ uint32_t s_scan[W + 1];
    __shared__ uint32_t s_pb[SRC_COLS ? PB_WORDS : 1];
    __shared__ int64_t s_dbase[SRC_COLS ? (THREADS >= 512 ? MAXB : MAX_PARTS) : 1];        // (1024 digits only on the large tile)
    __shared__ uint16_t s_dig[SRC_COLS ? TILE : 1];    // by slot: the digit (with the part rule it does not follow from the stored key)
    const int64_t tile = tile_of_block(a.n_tiles, a.xcd_tiles);
    if (tile >= a.n_tiles) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bins = 1 << a.bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    int64_t tile_base = tile * TILE, obase = tile, ostride = a.n_tiles;
    int n_here = (int)((a.n - tile_base) < TILE ? (a.n - tile_base) : TILE);
    if (a.tiles) { const TileDesc td = a.tiles[tile]; tile_base = td.row0; n_here = td.n; obase = td.tiles_before * bins + td.t; ostride = td.stride; }

    // (1) everything this thread will need from global memory, issued up front
    uint32_t key[RI], dig[RI];
    bool valid[RI];
    uint64_t w0[RI], w1[RI], w2[RI];
    if (SRC_COLS) {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int e = wave * SLICE + r * 64 + lane;
            valid[r] = e < n_here;
            w0[r] = valid[r] ? a.c_kmers[tile_base + e] : 0ull;
            w1[r] = valid[r] ? a.c_refs[tile_base + e] : 0ull;
            w2[r] = valid[r] ? ((uint64_t)a.c_nodes[tile_base + e] |
                                ((uint64_t)(a.carry_index ? (uint32_t)(tile_base + e) : a.c_af ? a.c_af[tile_base + e] : 0u) << 32)) : 0ull;
        }
        stage_parts(a.rule, s_pb);
        for (int d = threadIdx.x; d < bins; d += THREADS) s_dbase[d] = a.dbase ? a.dbase[d] : 0;
    } else {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int e = wave * SLICE + r * 64 + lane;
            valid[r] = e < n_here;
            key[r] = valid[r] ? a.keys_in[tile_base + e] : 0u;
            dig[r] = valid[r] ? ((key[r] >> a.shift) & mask) : 0u;
        }
        const uint64_t *src = a.rows_in + tile_base * 3;
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int j = r * THREADS + threadIdx.x;            // word j of the tile's 3 * n_here words
            w0[r] = j < 3 * n_here ? src[j] : 0ull;
            w1[r] = j + TILE < 3 * n_here ? src[j + TILE] : 0ull;
            w2[r] = j + 2 * TILE < 3 * n_here ? src[j + 2 * TILE] : 0ull;
        }
    }
    for (int d = threadIdx.x; d < W * MAXB / 2; d += THREADS) reinterpret_cast<uint32_t *>(s_rows)[d] = 0;          // s_wcnt
    __syncthreads();
    if (SRC_COLS) {
        bool bad = false;                                       // (the histogram kernel has reported it)
#pragma unroll
        for (int r = 0; r < RI; r++) {
            key[r] = 0u;
            uint32_t sort_key = 0u;
            if (a.rule.by_node) { key[r] = valid[r] ? (uint32_t)w2[r] : 0u; sort_key = key[r]; }
            else if (valid[r]) sort_key = key_of(a.rule, s_pb, w0[r], &bad, &key[r]);
            dig[r] = (sort_key >> a.shift) & mask;
        }
    }

    // (2) ranks inside the wave, digit counts per wave
    uint32_t rank[RI];
    wave_rank<RI>(dig, valid, a.bits, s_wcnt[wave], rank);
    __syncthreads();

    // (3) per digit: exclusive offsets over the waves, the digit's start in the sorted tile, and where its run goes
    {
        constexpr int C = MAXB / THREADS > 0 ? MAXB / THREADS : 1;      // digits per thread
        uint32_t tot[C], sum = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int d = threadIdx.x * C + c;
            tot[c] = 0;
            if (d < bins) {
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < W; w++) { const uint32_t x = s_wcnt[w][d]; s_wcnt[w][d] = (uint16_t)run; run += x; }
                tot[c] = run;
            }
            sum += tot[c];
        }
        uint32_t total;
        uint32_t ex = block_excl<THREADS>(sum, s_scan, &total);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int d = threadIdx.x * C + c;
            if (d < bins) {
                s_dstart[d] = ex;
                // DST_COLS: relative to the digit's first run (the caller's dbase[d] is where that one goes)
                // with dbase: relative to the digit's first run (the caller's dbase[d] is where that one goes)
                s_toff[d] = a.offs[obase + (int64_t)d * ostride] - (SRC_COLS && a.dbase ? a.offs[(int64_t)d * a.n_tiles] : 0u) - ex;
                ex += tot[c];
            }
        }
    }
    __syncthreads();

    // (4) destination slot of every row of the tile; (5) payload words into their rows' slots
    uint32_t slot[RI];
#pragma unroll
    for (int r = 0; r < RI; r++) {
        slot[r] = 0;
        if (valid[r]) {
            slot[r] = s_dstart[dig[r]] + s_wcnt[wave][dig[r]] + rank[r];
            s_keys[slot[r]] = key[r];
            if (SRC_COLS) s_dig[slot[r]] = (uint16_t)dig[r];
            else s_dest[wave * SLICE + r * 64 + lane] = (uint16_t)slot[r];
        }
    }
    __syncthreads();                                            // (the digit counts under s_rows are dead from here on)
    if (SRC_COLS) {
#pragma unroll
        for (int r = 0; r < RI; r++)
            if (valid[r]) { s_rows[slot[r] * 3 + 0] = w0[r]; s_rows[slot[r] * 3 + 1] = w1[r]; s_rows[slot[r] * 3 + 2] = w2[r]; }
        __syncthreads();
    }
    if (!SRC_COLS) {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int j0 = r * THREADS + threadIdx.x, j1 = j0 + TILE, j2 = j0 + 2 * TILE;
            if (j0 < 3 * n_here) s_rows[(int)s_dest[j0 / 3] * 3 + j0 % 3] = w0[r];
            if (j1 < 3 * n_here) s_rows[(int)s_dest[j1 / 3] * 3 + j1 % 3] = w1[r];
            if (j2 < 3 * n_here) s_rows[(int)s_dest[j2 / 3] * 3 + j2 % 3] = w2[r];
        }
        __syncthreads();
    }

    // (6) the sorted tile leaves as one contiguous run per digit: consecutive lanes, consecutive words
    if (DST_COLS) {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int p = r * THREADS + threadIdx.x;
            if (p < n_here) {
                const uint32_t d = s_dig[p];
                const int64_t row = s_dbase[d] + (int64_t)(uint32_t)(s_toff[d] + (uint32_t)p);
                const uint64_t w2 = s_rows[p * 3 + 2];
                a.o_kmers[row] = s_rows[p * 3]; a.o_refs[row] = s_rows[p * 3 + 1];
                a.o_nodes[row] = (uint32_t)w2; a.o_af[row] = (uint32_t)(w2 >> 32);
            }
        }
        return;
    }
    // (8 bytes per lane: 16-byte stores at the rows' 8-byte alignment were measured 6-10 % slower, profiles/r03_index_store16_ab.txt)
#pragma unroll
    for (int r = 0; r < 3 * RI; r++) {
        const int j = r * THREADS + threadIdx.x;
        if (j < 3 * n_here) {
            const int p = j / 3;
            const uint32_t d = SRC_COLS ? (uint32_t)s_dig[p] : ((s_keys[p] >> a.shift) & mask);
            int64_t row = (SRC_COLS ? s_dbase[d] : 0) + (int64_t)(uint32_t)(s_toff[d] + (uint32_t)p);
#if defined(GKI_TUNING) && defined(GKI_DBG_PART)        // where does the pass's time go: 1 = the sorted tile leaves in one piece
            row = tile_base + p;                          // (same LDS work, sequential stores; results wrong), 2 = no stores
            if (GKI_DBG_PART == 2 && row >= 0) continue;
#endif
            a.rows_out[row * 3 + (j - p * 3)] = s_rows[j];
        }
    }
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int p = r * THREADS + threadIdx.x;
        if (p < n_here) {
            const uint32_t k = s_keys[p];
            const uint32_t d = SRC_COLS ? (uint32_t)s_dig[p] : ((k >> a.shift) & mask);
            int64_t row = (SRC_COLS ? s_dbase[d] : 0) + (int64_t)(uint32_t)(s_toff[d] + (uint32_t)p);
#if defined(GKI_TUNING) && defined(GKI_DBG_PART)
            row = tile_base + p;
            if (GKI_DBG_PART == 2 && row >= 0) continue;
#endif
            a.keys_out[row] = k;
        }
    }
}

// The same pass with the sorted tile staged through LDS in NCH parts: the keys of the whole tile are ranked at once (every
// row knows its slot in the sorted tile and its output row), the payload waits in registers and goes through a staging
// buffer of TILE / NCH rows, one slot range of the sorted tile after the other.  LDS per workgroup drops from 131-141 KB to
// 64-72 KB: TWO workgroups of 4096-row tiles per CU, one loading while the other stores, with the runs of a 4096-row tile
// (2048-row tiles bought the overlap with runs half as long: no gain, profiles/r03_index_tile_ab.txt; 8192-row tiles staged
// in four parts bought longer runs without the overlap: -2 %, profiles/r04_partition_staged_tile_ab.txt).  The ranking
// scratch (per-wave digit counts, digit starts, run offsets) lives under the staging buffer: dead before the first payload
// word lands.  The build's passes since round 4 (GKI_PT_NCH); -DGKI_PT_NCH=1 rebuilds the one-workgroup kernel.
template <int THREADS, int RI, bool SRC_COLS, int NCH>
__global__ __launch_bounds__(THREADS, 4) void k_partition_rows_staged(PartArgs a) {
    constexpr int TILE = THREADS * RI, W = THREADS / 64, SLICE = TILE / W, CH = TILE / NCH;
    static_assert(TILE % NCH == 0 && (3 * RI) % NCH == 0 && TILE <= 65536, "slots are 16-bit; a part is a whole number of words per thread");
    static_assert((size_t)CH * 24 >= (size_t)W * MAXB * 2 + (size_t)MAXB * 8, "the ranking scratch fits under the staging buffer");
    __shared__ __attribute__((aligned(16))) uint64_t s_rows[CH * 3];
    __shared__ uint32_t s_keys[TILE];             // by slot
    __shared__ uint32_t s_out[TILE];              // by slot: the row of the output the slot goes to
    __shared__ uint16_t s_dest[SRC_COLS ? 1 : TILE];   // by position: the row's slot
    __shared__ uint32_t s_scan[W + 1];
    __shared__ uint32_t s_pb[SRC_COLS ? PB_WORDS : 1];
    uint16_t (*const s_wcnt)[MAXB] = reinterpret_cast<uint16_t (*)[MAXB]>(s_rows);                  // [W][MAXB]
    uint32_t *const s_dstart = reinterpret_cast<uint32_t *>(s_rows) + (size_t)W * MAXB / 2;        // [MAXB]
    uint32_t *const s_toff = s_dstart + MAXB;                                                       // [MAXB]
    const int64_t tile = tile_of_block(a.n_tiles, a.xcd_tiles);
    if (tile >= a.n_tiles) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bins = 1 << a.bits;
    const uint32_t mask = (uint32_t)bins - 1u;
    int64_t tile_base = tile * TILE, obase = tile, ostride = a.n_tiles;
    int n_here = (int)((a.n - tile_base) < TILE ? (a.n - tile_base) : TILE);
    if (a.tiles) { const TileDesc td = a.tiles[tile]; tile_base = td.row0; n_here = td.n; obase = td.tiles_before * bins + td.t; ostride = td.stride; }

    // (1) everything from global memory, issued up front
    uint32_t key[RI], dig[RI];
    bool valid[RI];
    uint64_t w0[RI], w1[RI], w2[RI];
    if (SRC_COLS) {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int e = wave * SLICE + r * 64 + lane;
            valid[r] = e < n_here;
            w0[r] = valid[r] ? a.c_kmers[tile_base + e] : 0ull;
            w1[r] = valid[r] ? a.c_refs[tile_base + e] : 0ull;
            w2[r] = valid[r] ? ((uint64_t)a.c_nodes[tile_base + e] |
                                ((uint64_t)(a.carry_index ? (uint32_t)(tile_base + e) : a.c_af ? a.c_af[tile_base + e] : 0u) << 32)) : 0ull;
        }
        stage_parts(a.rule, s_pb);
    } else {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int e = wave * SLICE + r * 64 + lane;
            valid[r] = e < n_here;
            key[r] = valid[r] ? a.keys_in[tile_base + e] : 0u;
            dig[r] = valid[r] ? ((key[r] >> a.shift) & mask) : 0u;
        }
        const uint64_t *src = a.rows_in + tile_base * 3;
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int j = r * THREADS + threadIdx.x;            // word j of the tile's 3 * n_here words
            w0[r] = j < 3 * n_here ? src[j] : 0ull;
            w1[r] = j + TILE < 3 * n_here ? src[j + TILE] : 0ull;
            w2[r] = j + 2 * TILE < 3 * n_here ? src[j + 2 * TILE] : 0ull;
        }
    }
    for (int d = threadIdx.x; d < W * MAXB / 2; d += THREADS) reinterpret_cast<uint32_t *>(s_rows)[d] = 0;      // s_wcnt
    __syncthreads();
    if (SRC_COLS) {
        bool bad = false;                                       // (the histogram kernel has reported it)
#pragma unroll
        for (int r = 0; r < RI; r++) {
            key[r] = 0u;
            uint32_t sort_key = 0u;
            if (a.rule.by_node) { key[r] = valid[r] ? (uint32_t)w2[r] : 0u; sort_key = key[r]; }
            else if (valid[r]) sort_key = key_of(a.rule, s_pb, w0[r], &bad, &key[r]);
            dig[r] = (sort_key >> a.shift) & mask;
        }
    }
    // (2) ranks inside the wave, digit counts per wave
    uint32_t rank[RI];
    wave_rank<RI>(dig, valid, a.bits, s_wcnt[wave], rank);
    __syncthreads();
    // (3) per digit: exclusive offsets over the waves, the digit's start in the sorted tile, and where its run goes
    {
        constexpr int C = MAXB / THREADS > 0 ? MAXB / THREADS : 1;      // digits per thread
        uint32_t tot[C], sum = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int d = threadIdx.x * C + c;
            tot[c] = 0;
            if (d < bins) {
                uint32_t run = 0;
#pragma unroll
                for (int w = 0; w < W; w++) { const uint32_t x = s_wcnt[w][d]; s_wcnt[w][d] = (uint16_t)run; run += x; }
                tot[c] = run;
            }
            sum += tot[c];
        }
        uint32_t total;
        uint32_t ex = block_excl<THREADS>(sum, s_scan, &total);
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int d = threadIdx.x * C + c;
            if (d < bins) {
                s_dstart[d] = ex;
                // with dbase (the bucket-range partition into rows: the caller lays the digits of several chunks out behind
                // each other) the run goes to dbase[d] + its offset among the chunk's runs of the digit; the output has
                // fewer than 2^32 rows (checked by the host), so 32-bit arithmetic carries it
                s_toff[d] = (SRC_COLS && a.dbase ? (uint32_t)a.dbase[d] - a.offs[(int64_t)d * a.n_tiles] : 0u) +
                            a.offs[obase + (int64_t)d * ostride] - ex;
                ex += tot[c];
            }
        }
    }
    __syncthreads();
    // (4) slot and output row of every row of the tile
    uint32_t slot[RI];
#pragma unroll
    for (int r = 0; r < RI; r++) {
        slot[r] = 0xFFFFFFFFu;
        if (valid[r]) {
            slot[r] = s_dstart[dig[r]] + s_wcnt[wave][dig[r]] + rank[r];
            s_keys[slot[r]] = key[r];
            s_out[slot[r]] = s_toff[dig[r]] + slot[r];
            if (!SRC_COLS) s_dest[wave * SLICE + r * 64 + lane] = (uint16_t)slot[r];
        }
    }
    __syncthreads();                                            // the ranking scratch is dead from here on
    // the staging-buffer word of every payload word this thread holds (by position), 0xFFFFFFFF: none
    uint32_t ws0[SRC_COLS ? 1 : RI], ws1[SRC_COLS ? 1 : RI], ws2[SRC_COLS ? 1 : RI];
    if (!SRC_COLS) {
#pragma unroll
        for (int r = 0; r < RI; r++) {
            const int j0 = r * THREADS + threadIdx.x, j1 = j0 + TILE, j2 = j0 + 2 * TILE;
            ws0[r] = j0 < 3 * n_here ? (uint32_t)s_dest[j0 / 3] * 3u + (uint32_t)(j0 % 3) : 0xFFFFFFFFu;
            ws1[r] = j1 < 3 * n_here ? (uint32_t)s_dest[j1 / 3] * 3u + (uint32_t)(j1 % 3) : 0xFFFFFFFFu;
            ws2[r] = j2 < 3 * n_here ? (uint32_t)s_dest[j2 / 3] * 3u + (uint32_t)(j2 % 3) : 0xFFFFFFFFu;
        }
    }
    // (5) + (6) per slot range of the sorted tile: payload words into the staging buffer, then out as runs
    for (int c = 0; c < NCH; c++) {
        const uint32_t lo = (uint32_t)(c * CH);
        if ((int)lo >= n_here) break;                           // (uniform)
        if (SRC_COLS) {
#pragma unroll
            for (int r = 0; r < RI; r++) {
                const uint32_t q = slot[r] - lo;
                if (q < (uint32_t)CH) { s_rows[q * 3 + 0] = w0[r]; s_rows[q * 3 + 1] = w1[r]; s_rows[q * 3 + 2] = w2[r]; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < RI; r++) {
                const uint32_t q0 = ws0[r] - lo * 3u, q1 = ws1[r] - lo * 3u, q2 = ws2[r] - lo * 3u;
                if (q0 < (uint32_t)(CH * 3)) s_rows[q0] = w0[r];
                if (q1 < (uint32_t)(CH * 3)) s_rows[q1] = w1[r];
                if (q2 < (uint32_t)(CH * 3)) s_rows[q2] = w2[r];
            }
        }
        __syncthreads();
        const int n_c = n_here - (int)lo < CH ? n_here - (int)lo : CH;
#pragma unroll
        for (int r = 0; r < 3 * RI / NCH; r++) {
            const int j = r * THREADS + threadIdx.x;
            if (j < 3 * n_c) {
                const int p = j / 3;
                a.rows_out[(int64_t)s_out[lo + p] * 3 + (j - p * 3)] = s_rows[j];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int p = r * THREADS + threadIdx.x;
        if (p < n_here) a.keys_out[(int64_t)s_out[p]] = s_keys[p];
    }
}

// First row and one-past-last row of every group (group = key >> L) in the keys sorted by group; arrays zeroed before.
// Four keys per lane (one 16-byte load), the key before a lane's four from the lane below (the first lane of a wave loads
// it): a boundary between rows i - 1 and i ends the group of i - 1 and begins the group of i.  Round 3's form (one key per
// lane, 2048 grid-stride workgroups) ran at 1.4 TB/s: 1.09 ms per 3.95e8 keys.
// A key at or beyond key_limit (keys handed in by a caller, gki_index_build_range_from_rows) is flagged and counted as the
// last valid key: the group tables are never indexed out of range.
__global__ __launch_bounds__(256) void k_group_bounds(const uint32_t *__restrict__ keys, int64_t n, int L, uint32_t key_limit,
                                                      uint32_t *__restrict__ gbegin, uint32_t *__restrict__ gend, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int64_t base = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    uint32_t g[4] = {0, 0, 0, 0};
    uint32_t k4[4] = {0, 0, 0, 0};
    if (base + 3 < n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(keys + base);
        k4[0] = v.x; k4[1] = v.y; k4[2] = v.z; k4[3] = v.w;
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) if (base + t < n) k4[t] = keys[base + t];
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if (k4[t] >= key_limit) { *bad = 1; k4[t] = key_limit - 1u; }
        g[t] = k4[t] >> L;
    }
    uint32_t before = __shfl_up(g[3], 1, 64);
    if (lane == 0 && base > 0 && base < n) { const uint32_t kb = keys[base - 1]; before = (kb < key_limit ? kb : key_limit - 1u) >> L; }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t i = base + t;
        if (i < n) {
            const uint32_t prev = t == 0 ? before : g[t - 1];
            if (i == 0) gbegin[g[t]] = 0;
            else if (prev != g[t]) { gbegin[g[t]] = (uint32_t)i; gend[prev] = (uint32_t)i; }
            if (i == n - 1) gend[g[t]] = (uint32_t)n;
        }
    }
}

// stats[0] = rows of the largest group, stats[1] = number of groups with more than `cap` rows (their ids -> large[])
__global__ __launch_bounds__(256) void k_group_scan(const uint32_t *__restrict__ gbegin, const uint32_t *__restrict__ gend,
                                                    int64_t n_groups, uint32_t cap, unsigned int *__restrict__ stats,
                                                    uint32_t *__restrict__ large, uint32_t large_cap) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t mx = 0;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += stride) {
        const uint32_t m = gend[g] - gbegin[g];
        mx = m > mx ? m : mx;
        if (m > cap) {
            const unsigned int at = atomicAdd(&stats[1], 1u);
            if (at < large_cap) large[at] = (uint32_t)g;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_down(mx, d, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&stats[0], mx);
}

struct FinishArgs {
    const uint64_t *rows; const uint32_t *keys;
    const uint32_t *gbegin, *gend;
    int64_t n_groups;
    int L;
    int lbits;                     // bits that tell the buckets of one group apart (= L; fewer in the last, clipped group never matters)
    uint64_t n_buckets;
    int skip_frequencies;
    int32_t *h2i; uint32_t *nk;
    uint64_t *o_kmers; uint32_t *o_nodes; uint64_t *o_refs; uint32_t *o_af; uint16_t *o_freq;
    uint32_t *o_perm; const uint32_t *af_in;     // rows carry their input index: permutation out, allele frequency fetched by it
    uint32_t *big_buckets; unsigned int *n_big; uint32_t big_cap;    // buckets of more than SMALL_BUCKET rows (frequencies later)
    int xcd_groups;
};

// One workgroup per group of 2^L buckets: everything about the group happens in LDS.
//
// Two ways to the stable slot of a row (rows of a bucket in input order), chosen by the host from the density:
//  * WRANK = false (few rows per bucket: the variant index, 0.7): count by LDS atomics, a provisional slot by a second
//    atomic, then the row's rank among its bucket's rows by position -- a loop over the bucket, 1.4 rows on average;
//  * WRANK = true (dense slices of a whole-genome index, 7 rows per bucket): the partition kernel's ranking -- rows in
//    (wave, round, lane) order, match-any by ballots inside the wave, per-wave bucket counts, exclusive offsets over the
//    waves -- whose cost does not grow with the bucket.  With the loop, a wave waited for its longest bucket (15-20 rows)
//    three times over (rank, same k-mer, earlier duplicate): 12.2 ms per 3.95e8 rows against 4.3 ms per 3.1e8 sparse ones.
template <bool WRANK, int GROUP_CAP = GKI_GROUP_CAP, int GROUP_THREADS = GKI_GROUP_THREADS>
__global__ __launch_bounds__(GROUP_THREADS) void k_group_finish(FinishArgs a) {
    constexpr int NB = 1 << GROUP_LMAX, RI = GROUP_CAP / GROUP_THREADS, W = GROUP_THREADS / 64, SLICE = GROUP_CAP / W;
    constexpr int C = NB / GROUP_THREADS > 0 ? NB / GROUP_THREADS : 1;      // buckets per thread: b = i * THREADS + thread
    constexpr int PAD = 4;                                                  // the frequency loop reads four slots at a time
    static_assert(SLICE == RI * 64, "a wave owns a contiguous slice of the group");
    static_assert((size_t)W * NB * 2 <= (size_t)2 * (GROUP_CAP + PAD) * 8, "the per-wave counts fit under the payload");
    __shared__ uint64_t s_kr[2 * (GROUP_CAP + PAD)];          // by slot: k-mers, then ref offsets; before the payload lands,
    uint64_t *const s_kmer = s_kr, *const s_ref = s_kr + GROUP_CAP + PAD;   // WRANK keeps its per-wave bucket counts here
    uint16_t *const s_wcnt = reinterpret_cast<uint16_t *>(s_kr);            // [W][nbk]
    __shared__ uint32_t s_node[GROUP_CAP], s_af[GROUP_CAP];
    __shared__ uint32_t s_cnt[NB], s_pos[NB];     // rows of a bucket; its first slot (WRANK) / next free slot (atomics)
    __shared__ uint16_t s_lk[GROUP_CAP];        // by slot: the row's bucket within the group
    __shared__ uint16_t s_src[WRANK ? 1 : GROUP_CAP];       // by provisional slot: the row's position in the group (input order)
    __shared__ uint16_t s_dest[GROUP_CAP];      // by position: the row's final slot
    __shared__ uint32_t s_scan[W + 1];
    int64_t g = blockIdx.x;
    if (a.xcd_groups > 0) { g = (int64_t)(blockIdx.x & 7) * a.xcd_groups + (blockIdx.x >> 3); if (g >= a.n_groups) return; }
    const uint32_t s = a.gbegin[g], m = a.gend[g] - s;
    const uint64_t gb = (uint64_t)g << a.L;
    const uint32_t nbk = (uint32_t)((a.n_buckets - gb) < (1ull << a.L) ? (a.n_buckets - gb) : (1ull << a.L));
    if (m > GROUP_CAP) return;                                  // k_group_large's
    const uint32_t lmask = (1u << a.L) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // rows and keys of the group, issued before anything else
    uint32_t lk[RI], e_of[RI];
    bool valid[RI];
    uint64_t w0[RI], w1[RI], w2[RI];
    const uint64_t *src = a.rows + (int64_t)s * 3;
#pragma unroll
    for (int r = 0; r < RI; r++) {
        e_of[r] = WRANK ? (uint32_t)(wave * SLICE + r * 64 + lane) : (uint32_t)(r * GROUP_THREADS + threadIdx.x);
        valid[r] = e_of[r] < m;
        lk[r] = valid[r] ? (a.keys[(int64_t)s + e_of[r]] & lmask) : 0u;
        const uint32_t j = r * GROUP_THREADS + threadIdx.x;
        w0[r] = j < 3 * m ? src[j] : 0ull;
        w1[r] = j + GROUP_CAP < 3 * m ? src[j + GROUP_CAP] : 0ull;
        w2[r] = j + 2 * GROUP_CAP < 3 * m ? src[j + 2 * GROUP_CAP] : 0ull;
    }
    uint32_t rank[RI];
    if (WRANK) {
        for (uint32_t i = threadIdx.x; i < (W * nbk + 1) / 2; i += GROUP_THREADS) reinterpret_cast<uint32_t *>(s_wcnt)[i] = 0;
        __syncthreads();
        wave_rank<RI>(lk, valid, a.lbits, s_wcnt + wave * nbk, rank);
        __syncthreads();
    } else {
        for (uint32_t b = threadIdx.x; b < nbk; b += GROUP_THREADS) s_cnt[b] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RI; r++)
            if (valid[r]) atomicAdd(&s_cnt[lk[r]], 1u);
        __syncthreads();
    }
    // bucket starts (exclusive scan over the group's buckets)
    {
        uint32_t c[C], base = 0;
#pragma unroll
        for (int i = 0; i < C; i++) {
            const uint32_t b = i * GROUP_THREADS + threadIdx.x;
            c[i] = 0;
            if (b < nbk) {
                if (WRANK) {
                    uint32_t run = 0;
#pragma unroll
                    for (int w = 0; w < W; w++) { const uint32_t x = s_wcnt[w * nbk + b]; s_wcnt[w * nbk + b] = (uint16_t)run; run += x; }
                    c[i] = run;
                    s_cnt[b] = run;
                } else c[i] = s_cnt[b];
            }
        }
#pragma unroll
        for (int i = 0; i < C; i++) {
            if ((uint32_t)(i * GROUP_THREADS) < nbk) {                       // (uniform)
                uint32_t total;
                const uint32_t ex = base + block_excl<GROUP_THREADS>(c[i], s_scan, &total);
                const uint32_t b = i * GROUP_THREADS + threadIdx.x;
                if (b < nbk) s_pos[b] = ex;
                base += total;
            }
        }
    }
    __syncthreads();
    if (WRANK) {
        // the directory of the group, streamed (this replaces the memset + scatter of the first form)
        for (uint32_t b = threadIdx.x; b < nbk; b += GROUP_THREADS) {
            const uint32_t c = s_cnt[b];
            a.h2i[gb + b] = c ? (int32_t)(s + s_pos[b]) : 0;     // collision_free_kmer_index.py:453-454
            a.nk[gb + b] = c;                                    // :456-457
        }
#pragma unroll
        for (int r = 0; r < RI; r++) {
            if (valid[r]) {
                const uint32_t slot = s_pos[lk[r]] + s_wcnt[wave * nbk + lk[r]] + rank[r];
                s_dest[e_of[r]] = (uint16_t)slot;
                s_lk[slot] = (uint16_t)lk[r];
            }
        }
    } else {
        // a provisional slot inside the bucket (arrival order), then the stable one: rows of a bucket in input order
#pragma unroll
        for (int r = 0; r < RI; r++)
            if (valid[r]) s_src[atomicAdd(&s_pos[lk[r]], 1u)] = (uint16_t)e_of[r];
        __syncthreads();                                             // from here on: start of bucket b = s_pos[b] - s_cnt[b]
        for (uint32_t b = threadIdx.x; b < nbk; b += GROUP_THREADS) {
            const uint32_t c = s_cnt[b];
            a.h2i[gb + b] = c ? (int32_t)(s + s_pos[b] - c) : 0;     // collision_free_kmer_index.py:453-454
            a.nk[gb + b] = c;                                        // :456-457
        }
#pragma unroll
        for (int r = 0; r < RI; r++) {
            if (valid[r]) {
                const uint32_t c = s_cnt[lk[r]], b0 = s_pos[lk[r]] - c;
                uint32_t before = 0;
                for (uint32_t j = b0; j < b0 + c; j++) before += s_src[j] < e_of[r] ? 1u : 0u;
                s_dest[e_of[r]] = (uint16_t)(b0 + before);
                s_lk[b0 + before] = (uint16_t)lk[r];
            }
        }
    }
    __syncthreads();
    // payload words to their rows' slots, column-wise in LDS
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const uint32_t j0 = r * GROUP_THREADS + threadIdx.x, j1 = j0 + GROUP_CAP, j2 = j0 + 2 * GROUP_CAP;
        const uint64_t w[3] = {w0[r], w1[r], w2[r]};
        const uint32_t jj[3] = {j0, j1, j2};
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (jj[t] < 3 * m) {
                const uint32_t row = jj[t] / 3, word = jj[t] - row * 3, slot = s_dest[row];
                if (word == 0) s_kmer[slot] = w[t];
                else if (word == 1) s_ref[slot] = w[t];
                else { s_node[slot] = (uint32_t)w[t]; s_af[slot] = (uint32_t)(w[t] >> 32); }
            }
        }
    }
    __syncthreads();
    // the group's slice of the output columns, frequencies computed on the way (set_frequencies :267-293)
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const uint32_t p = r * GROUP_THREADS + threadIdx.x;
        if (p < m) {
            const uint64_t km = s_kmer[p], rf = s_ref[p];
            const int64_t o = (int64_t)s + p;
            a.o_kmers[o] = km; a.o_refs[o] = rf;
            if (a.o_nodes) a.o_nodes[o] = s_node[p];             // (NULL node / allele-frequency / frequency columns: the reverse index)
            if (a.o_perm) { const uint32_t idx = s_af[p]; a.o_perm[o] = idx; a.o_af[o] = a.af_in[idx]; }
            else if (a.o_af) a.o_af[o] = s_af[p];
            uint32_t f = 0;
            if (!a.skip_frequencies) {
                const uint32_t b = s_lk[p], c = s_cnt[b], b0 = WRANK ? s_pos[b] : s_pos[b] - c;
                if (c == 1) f = 1;
                else if (c <= SMALL_BUCKET) {
                    // rows of the bucket that carry this row's k-mer, four slots per trip (the loads of a trip are independent;
                    // slots past the bucket are read and masked: the arrays are padded by PAD)
                    const uint32_t end = b0 + c;
                    uint32_t same = 0;
                    for (uint32_t j = b0; j < end; j += 4) {
                        const uint64_t k0 = s_kmer[j], k1 = s_kmer[j + 1], k2 = s_kmer[j + 2], k3 = s_kmer[j + 3];
                        same += (k0 == km) + (k1 == km && j + 1 < end) + (k2 == km && j + 2 < end) + (k3 == km && j + 3 < end);
                    }
                    if (same == 1) f = 1;                              // only itself: one (k-mer, ref offset) pair
                    else {
                        for (uint32_t j = b0; j < end; j++) {
                            if (s_kmer[j] != km) continue;
                            const uint64_t rj = s_ref[j];
                            bool dup = false;
                            for (uint32_t q = b0; q < j; q++) dup |= (s_kmer[q] == km && s_ref[q] == rj);
                            f += dup ? 0u : 1u;
                        }
                    }
                } else if (p == b0) {
                    const unsigned int at = atomicAdd(a.n_big, 1u);
                    if (at < a.big_cap) a.big_buckets[at] = (uint32_t)(gb + b);
                }
            }
            if (a.o_freq) a.o_freq[o] = (uint16_t)f;
        }
    }
}

// A group with more rows than k_group_finish keeps in LDS: one workgroup streams it three times (count, then an
// ordered scatter of its rows straight into the output columns).  Frequencies of its buckets are left to the caller.
__global__ __launch_bounds__(256) void k_group_large(FinishArgs a, const uint32_t *__restrict__ large, unsigned int n_large) {
    constexpr int NB = 1 << GROUP_LMAX;
    __shared__ uint32_t s_cnt[NB], s_start[NB];
    __shared__ uint32_t s_scan[256 / 64 + 1];
    const uint32_t lmask = (1u << a.L) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    for (unsigned int li = blockIdx.x; li < n_large; li += gridDim.x) {
        const int64_t g = large[li];
        const uint32_t s = a.gbegin[g], m = a.gend[g] - s;
        const uint64_t gb = (uint64_t)g << a.L;
        const uint32_t nbk = (uint32_t)((a.n_buckets - gb) < (1ull << a.L) ? (a.n_buckets - gb) : (1ull << a.L));
        for (uint32_t b = threadIdx.x; b < NB; b += 256) s_cnt[b] = 0;
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < m; e += 256) atomicAdd(&s_cnt[a.keys[(int64_t)s + e] & lmask], 1u);
        __syncthreads();
        {
            constexpr int C = NB / 256;
            uint32_t c[C], sum = 0;
#pragma unroll
            for (int i = 0; i < C; i++) { c[i] = s_cnt[threadIdx.x * C + i]; sum += c[i]; }
            uint32_t total;
            uint32_t ex = block_excl<256>(sum, s_scan, &total);
#pragma unroll
            for (int i = 0; i < C; i++) { s_start[threadIdx.x * C + i] = ex; ex += c[i]; }
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < nbk; b += 256) {
            const uint32_t c = s_cnt[b];
            a.h2i[gb + b] = c ? (int32_t)(s + s_start[b]) : 0;
            a.nk[gb + b] = c;
        }
        __syncthreads();
        // ordered scatter: chunks of 256 rows, the four waves of a chunk one after the other; s_start[b] runs ahead as
        // the next free slot of bucket b
        for (uint32_t c0 = 0; c0 < m; c0 += 256) {
            const uint32_t e = c0 + threadIdx.x;
            const bool valid = e < m;
            const uint32_t b = valid ? (a.keys[(int64_t)s + e] & lmask) : 0u;
            uint64_t w0 = 0, w1 = 0, w2 = 0;
            if (valid) { const uint64_t *row = a.rows + ((int64_t)s + e) * 3; w0 = row[0]; w1 = row[1]; w2 = row[2]; }
            uint64_t same = __ballot(valid);
            for (int t = 0; t < a.L; t++) {
                const uint64_t bit = __ballot((b >> t) & 1u);
                same &= ((b >> t) & 1u) ? bit : ~bit;
            }
            const uint32_t before = (uint32_t)__popcll(same & lt_mask);
            for (int w = 0; w < 4; w++) {
                if (wave == w && valid) {
                    const uint32_t slot = s_start[b] + before;
                    __builtin_amdgcn_wave_barrier();
                    if (before == 0) s_start[b] = slot + (uint32_t)__popcll(same);
                    const int64_t o = (int64_t)s + slot;
                    a.o_kmers[o] = w0; a.o_refs[o] = w1;
                    if (a.o_nodes) a.o_nodes[o] = (uint32_t)w2;
                    if (a.o_perm) { const uint32_t idx = (uint32_t)(w2 >> 32); a.o_perm[o] = idx; a.o_af[o] = a.af_in[idx]; }
                    else if (a.o_af) a.o_af[o] = (uint32_t)(w2 >> 32);
                    if (a.o_freq) a.o_freq[o] = 0;
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// row ranges whose frequencies are still open: every large group, every big bucket
__global__ __launch_bounds__(256) void k_open_ranges(const uint32_t *__restrict__ large, unsigned int n_large,
                                                     const uint32_t *__restrict__ gbegin, const uint32_t *__restrict__ gend,
                                                     const uint32_t *__restrict__ big, unsigned int n_big,
                                                     const int32_t *__restrict__ h2i, const uint32_t *__restrict__ nk,
                                                     int64_t *__restrict__ rb, int64_t *__restrict__ re) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_large + n_big; i += gridDim.x * blockDim.x) {
        if (i < n_large) { rb[i] = gbegin[large[i]]; re[i] = gend[large[i]]; }
        else { const uint32_t b = big[i - n_large]; rb[i] = h2i[b]; re[i] = (int64_t)h2i[b] + nk[b]; }
    }
}

int key_bits(uint64_t max_key) {
    int bits = 0;
    while (bits < 32 && (max_key >> bits) != 0) bits++;
    return bits;
}

}  // namespace

// The tile's histogram of the first pass's digit, the keys computed from the k-mers by `rule` (and not stored: the first
// partition pass computes them again from the k-mers it loads anyway -- 8 bytes read per record instead of 8 read, 4
// written and 4 read back).  *out_of_range is set when a bucket lies outside the slice.
template <int THREADS, int RI>
__global__ __launch_bounds__(THREADS) void k_kmer_digit_hist(const uint64_t *__restrict__ kmers, int64_t n, KeyRule rule, int shift, int bits,
                                                             uint32_t *__restrict__ hist, int64_t n_tiles, int xcd_tiles,
                                                             const TileDesc *__restrict__ tiles, int *__restrict__ out_of_range) {
    __shared__ uint32_t h[MAXB];
    __shared__ uint32_t s_pb[PB_WORDS];
    const int64_t tile = tile_of_block(n_tiles, xcd_tiles);
    if (tile >= n_tiles) return;
    const int bins = 1 << bits;
    for (int d = threadIdx.x; d < bins; d += THREADS) h[d] = 0;
    stage_parts(rule, s_pb);
    __syncthreads();
    int64_t base = tile * (THREADS * RI), obase = tile, ostride = n_tiles;
    int64_t n_here = n - base < THREADS * RI ? n - base : THREADS * RI;
    if (tiles) { const TileDesc td = tiles[tile]; base = td.row0; n_here = td.n; obase = td.tiles_before * bins + td.t; ostride = td.stride; }
    const uint32_t mask = (uint32_t)bins - 1u;
    uint64_t km[RI];
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const int e = r * THREADS + threadIdx.x;
        km[r] = e < n_here ? kmers[base + e] : 0ull;
    }
    bool bad = false;
    FewDigits c = {0ull, 0ull};
#pragma unroll
    for (int r = 0; r < RI; r++) {
        const bool valid = r * THREADS + threadIdx.x < n_here;
        uint32_t stored;
        const uint32_t key = valid ? key_of(rule, s_pb, km[r], &bad, &stored) : 0u;
        if (bits <= 4) few_add(c, (key >> shift) & mask, valid);
        else if (valid) atomicAdd(&h[(key >> shift) & mask], 1u);
    }
    if (bits <= 4) few_flush(c, h, bins);
    if (bad) *out_of_range = 1;
    __syncthreads();
    for (int d = threadIdx.x; d < bins; d += THREADS) hist[obase + (int64_t)d * ostride] = h[d];
}

// Tile shape of the partition passes.  4096-row tiles (one workgroup of 512 threads per CU, 144 KB of LDS) give every
// digit of a 512-way pass a run of 8 rows = 192 bytes on average; see DESIGN.md 4.3 for the measured alternatives.
#ifndef GKI_PT_THREADS
#define GKI_PT_THREADS 512
#endif
#ifndef GKI_PT_RI
#define GKI_PT_RI 8
#endif
// > 1: the build's passes run k_partition_rows_staged (the sorted tile staged through LDS in this many parts: 4 -> 64 KB of
// LDS, two workgroups per CU); 1: k_partition_rows (one workgroup per CU).  Same box, alternating
// (profiles/r04_partition_two_workgroups_ab.txt): first pass 7.75 -> 5.37 ms, second 6.36 -> 5.00 ms per 3.95e8 records.
#ifndef GKI_PT_NCH
#define GKI_PT_NCH 4
#endif
// tile of the bucket-range partition (<= 256 parts: its runs are long whatever the tile): 2048 rows, 72 KB of LDS, two
// workgroups per CU -- one loads while the other stores: 50.1 -> 45.0 ms per 3.16e9 records, same box
// (profiles/r04_full_index_ab.txt).  The build's 512- and 1024-way passes keep 4096-row tiles (longer runs matter more there).
#ifndef GKI_PR_THREADS
#define GKI_PR_THREADS 256
#endif

// Which way k_group_finish ranks the rows of a bucket: by ballots (cost independent of the bucket) from this many rows per
// bucket on, by LDS atomics + a loop over the bucket below it.  -DGKI_FINISH_WRANK=0 / =1 force one way (A/B builds).
#ifndef GKI_FINISH_WRANK
#define GKI_FINISH_WRANK -1
#endif
#ifndef GKI_FINISH_WRANK_DENSITY
#define GKI_FINISH_WRANK_DENSITY 2.0
#endif
// A group may hold this many standard deviations (of a Poisson count) more than the average group before it overflows
// the LDS capacity; the few that do are streamed by k_group_large.  (Round 3 asked for a quarter more than the average
// plus 64 rows, which left the groups of a 7-rows-per-bucket slice at 448 of 1024 rows: twice the workgroups, each with
// the full footprint and every barrier -- VERDICT r3 weak #1.)
#ifndef GKI_GROUP_SIGMAS
#define GKI_GROUP_SIGMAS 4.0
#endif

// The row-carrying build.  Returns GKI_OK with *done = 1 when it built the index, *done = 0 when the input is outside
// its domain (a group too large to stream with one workgroup) and the caller should use the pair-sorting form.
//
// Grouped build (group_bits > 0, h_group_start[2^group_bits + 1]): the records arrive grouped by the top group_bits bits of
// their key -- group g = key >> (key bits - group_bits) holds rows [h_group_start[g], h_group_start[g + 1]), as
// gki_partition_by_bucket_range_grouped leaves them.  Those bits are sorted already; the passes sort the bits between
// them and the finish, every group on its own (segmented tiles), and the finish takes 4096-row groups: for the slices
// of a whole-genome index (26 key bits, 7 records per bucket) that is 7 bits grouped + ONE pass of 10 + 9 in LDS, where
// the ungrouped build needs two passes of 10 + 9 and 7 in LDS.
int gki_index_build_rows(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                         uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                         int group_bits, const int64_t *h_group_start, const void *d_rows_in, const void *d_keys_in,
                         void *d_hashes_to_index, void *d_n_kmers, void *d_out_kmers, void *d_out_nodes,
                         void *d_out_ref_offsets, void *d_out_af32, void *d_out_frequencies, void *d_out_permutation, int *done,
                         int by_node) {
    // by_node: the key of a record is its node id and the "buckets" are the nodes (ReverseKmerIndex.from_flat_kmers,
    // reverse_kmer_index.py:47-60: records stably sorted by node): modulo is unused, n_buckets = the number of nodes, the
    // allele-frequency, node and frequency output columns may be NULL
    constexpr int THREADS = GKI_PT_THREADS, RI = GKI_PT_RI, TILE = THREADS * RI;
    *done = 0;
    hipStream_t s = 0;
    // d_rows_in / d_keys_in: the records arrive as 24-byte rows with their keys (gki_partition_rows_by_bucket_range) instead of
    // as four columns: the first pass is then a pass like every other, or -- when the grouping left nothing to sort above the
    // finish -- there is none
    const bool from_rows = d_rows_in != nullptr;
    if (from_rows && d_out_permutation) return gki_set_error(GKI_ERR_BAD_ARG, "a build from rows has no permutation to give");
    const int kb = key_bits(n_buckets - 1);
    const bool grouped = group_bits > 0;
    const int group_shift = grouped ? (kb > group_bits ? kb - group_bits : 0) : kb;     // key bits below the grouping
    const int cap = grouped ? GROUP_CAP_BIG : GROUP_CAP;
    // L: low key bits resolved inside LDS -- the largest for which an average group and GKI_GROUP_SIGMAS deviations fit
    const double density = (double)n / (double)n_buckets;
    int L = kb < GROUP_LMAX ? kb : GROUP_LMAX;
    while (L > 0) {
        const double rows = density * (double)(1ull << L);
        if (rows + GKI_GROUP_SIGMAS * sqrt(rows) <= (double)cap) break;
        L--;
    }
    const bool wrank = GKI_FINISH_WRANK < 0 ? density >= GKI_FINISH_WRANK_DENSITY : GKI_FINISH_WRANK != 0;
    const int top = group_shift > L ? group_shift - L : 0;    // bits the partition passes sort on
    const int n_pass = top > 0 ? (top + MAXB_BITS - 1) / MAXB_BITS : (from_rows ? 0 : 1);   // top == 0: one pass of one digit, which only packs the rows
    if (n_pass > 3) return GKI_OK;
    const int64_t n_groups = (int64_t)(((n_buckets - 1) >> L) + 1);
    // tiles: plain, or cut at the group bounds
    // (the descriptors are written on the device, k_tile_descs: the host hands over three numbers per group, not one
    // record per tile -- 94 000 of them, 3 MB through pageable memory, per slice of the whole-genome index)
    std::vector<int64_t> h_seg;                               // per group: first row, one past its last row, tiles before it
    int64_t n_tiles = ceil_div(n, TILE);
    const int64_t n_seg = grouped ? (int64_t)1 << group_bits : 0;
    if (grouped) {
        if (h_group_start[0] != 0 || h_group_start[n_seg] != n) return gki_set_error(GKI_ERR_BAD_ARG, "group_start must run from 0 to n");
        int64_t t_before = 0;
        h_seg.resize((size_t)(3 * n_seg));
        for (int64_t g = 0; g < n_seg; g++) {
            const int64_t a0 = h_group_start[g], a1 = h_group_start[g + 1];
            if (a1 < a0) return gki_set_error(GKI_ERR_BAD_ARG, "group_start must not decrease");
            h_seg[(size_t)(3 * g)] = a0; h_seg[(size_t)(3 * g + 1)] = a1; h_seg[(size_t)(3 * g + 2)] = t_before;
            t_before += ceil_div(a1 - a0, TILE);
        }
        n_tiles = t_before;
    }
    const int64_t hist_n = (int64_t)MAXB * n_tiles;
    const int64_t tmp_bytes = gki_scan_tmp_bytes(hist_n);
    uint64_t *rows[2] = {nullptr, nullptr};
    uint32_t *keys[2] = {nullptr, nullptr}, *hist = nullptr, *offs = nullptr, *gbegin = nullptr, *gend = nullptr;
    uint32_t *large = nullptr, *big = nullptr;
    unsigned int *stats = nullptr;                            // [0] max group, [1] large groups, [2] big buckets, [3] out of range
    int64_t *rng = nullptr;
    void *tmp = nullptr;
    const uint32_t large_cap = 1u << 16;
    const uint32_t big_cap = (uint32_t)(n / SMALL_BUCKET + 1);
    TileDesc *d_tiles = nullptr;
    int64_t *d_seg = nullptr;
    KeyRule rule;
    rule.mod = gki_mod_of(modulo); rule.bucket_begin = bucket_begin; rule.n_buckets = n_buckets; rule.n_parts = 0; rule.sub_bits = 0;
    rule.part_begin = nullptr; rule.sub_shift = nullptr; rule.parts_per_bucket = 0.f; rule.parts_log2 = -1; rule.by_node = by_node;
    int rc = GKI_OK;
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        HIP_G(gki_dev_malloc((void **)&stats, 64));
        HIP_G(hipMemsetAsync(stats, 0, 64, s));
        if (grouped) {
            HIP_G(gki_dev_malloc((void **)&d_tiles, (size_t)n_tiles * sizeof(TileDesc) + 32));
            HIP_G(gki_dev_malloc((void **)&d_seg, h_seg.size() * 8));
            HIP_G(hipMemcpyAsync(d_seg, h_seg.data(), h_seg.size() * 8, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_tile_descs, dim3((unsigned)n_seg), dim3(256), 0, s, d_seg, TILE, d_tiles);
            HIP_G(hipGetLastError());
        }
        for (int i = 0; i < (n_pass > 1 ? 2 : n_pass); i++) {
            HIP_G(gki_dev_malloc((void **)&rows[i], (size_t)n * 24));
            HIP_G(gki_dev_malloc((void **)&keys[i], (size_t)n * 4));
        }
        HIP_G(gki_dev_malloc((void **)&hist, (size_t)hist_n * 4));
        HIP_G(gki_dev_malloc((void **)&offs, (size_t)(hist_n + 1) * 4));
        HIP_G(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
        HIP_G(gki_dev_malloc((void **)&gbegin, (size_t)n_groups * 4));
        HIP_G(gki_dev_malloc((void **)&gend, (size_t)n_groups * 4));
        HIP_G(gki_dev_malloc((void **)&large, (size_t)large_cap * 4));
        HIP_G(gki_dev_malloc((void **)&big, (size_t)big_cap * 4));
        HIP_G(hipMemsetAsync(gbegin, 0, (size_t)n_groups * 4, s));
        HIP_G(hipMemsetAsync(gend, 0, (size_t)n_groups * 4, s));
        // partition passes on the top bits, least significant digit first, each stable
        const uint64_t *cur_rows = (const uint64_t *)d_rows_in;
        const uint32_t *cur_keys = (const uint32_t *)d_keys_in;
        int shift = L;
        for (int p = 0; p < n_pass; p++) {
            const int bits = (top - (shift - L) + (n_pass - p) - 1) / (n_pass - p);       // remaining bits spread evenly (an odd bit
                                                                                           // first or last: no difference measured)
            const int64_t bins_n = ((int64_t)1 << bits) * n_tiles;
            const int xcd_tiles = (int)ceil_div(n_tiles, 8);
            const unsigned xgrid = (unsigned)(xcd_tiles * 8);
            if (p == 0 && by_node)
                hipLaunchKernelGGL((k_digit_hist<THREADS, RI>), dim3(xgrid), dim3(THREADS), 0, s, (const uint32_t *)d_nodes, n, shift, bits,
                                   hist, n_tiles, xcd_tiles, d_tiles);
            else if (p == 0 && !from_rows)
                hipLaunchKernelGGL((k_kmer_digit_hist<THREADS, RI>), dim3(xgrid), dim3(THREADS), 0, s, (const uint64_t *)d_kmers,
                                   n, rule, shift, bits, hist, n_tiles, xcd_tiles, d_tiles, (int *)(stats + 3));
            else
                hipLaunchKernelGGL((k_digit_hist<THREADS, RI>), dim3(xgrid), dim3(THREADS), 0, s, cur_keys, n, shift, bits,
                                   hist, n_tiles, xcd_tiles, d_tiles);
            HIP_G(hipGetLastError());
            rc = gki_scan_u32_to_u32(hist, bins_n, offs, tmp, tmp_bytes, s);
            if (rc != GKI_OK) goto done;
            PartArgs a;
            a.tiles = d_tiles;
            a.keys_in = cur_keys; a.c_kmers = (const uint64_t *)d_kmers; a.c_nodes = (const uint32_t *)d_nodes;
            a.c_refs = (const uint64_t *)d_ref_offsets; a.c_af = (const uint32_t *)d_af32; a.rule = rule; a.rows_in = cur_rows;
            a.n = n; a.n_tiles = n_tiles; a.shift = shift; a.bits = bits; a.offs = offs; a.carry_index = d_out_permutation != nullptr;
            a.rows_out = rows[p & 1]; a.keys_out = keys[p & 1];
            a.o_kmers = nullptr; a.o_nodes = nullptr; a.o_refs = nullptr; a.o_af = nullptr; a.dbase = nullptr;
            a.xcd_tiles = xcd_tiles;
#if GKI_PT_NCH > 1
            if (p == 0 && !from_rows) hipLaunchKernelGGL((k_partition_rows_staged<THREADS, RI, true, GKI_PT_NCH>), dim3(xgrid), dim3(THREADS), 0, s, a);
            else hipLaunchKernelGGL((k_partition_rows_staged<THREADS, RI, false, GKI_PT_NCH>), dim3(xgrid), dim3(THREADS), 0, s, a);
#else
            if (p == 0 && !from_rows) hipLaunchKernelGGL((k_partition_rows<THREADS, RI, true>), dim3(xgrid), dim3(THREADS), 0, s, a);
            else hipLaunchKernelGGL((k_partition_rows<THREADS, RI, false>), dim3(xgrid), dim3(THREADS), 0, s, a);
#endif
            HIP_G(hipGetLastError());
            cur_rows = a.rows_out; cur_keys = a.keys_out;
            shift += bits;
        }
        hipLaunchKernelGGL(k_group_bounds, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, s, cur_keys, n, L, (uint32_t)n_buckets, gbegin, gend,
                           (int *)(stats + 3));
        HIP_G(hipGetLastError());
        hipLaunchKernelGGL(k_group_scan, dim3(stream_grid(n_groups, 256)), dim3(256), 0, s, gbegin, gend, n_groups,
                           (uint32_t)cap, stats, large, large_cap);
        HIP_G(hipGetLastError());
        unsigned int h_stats[4] = {0, 0, 0, 0};
        HIP_G(hipMemcpyAsync(h_stats, stats, 16, hipMemcpyDeviceToHost, s));
        HIP_G(hipStreamSynchronize(s));
        if (h_stats[3]) { rc = gki_set_error(GKI_ERR_BAD_ARG, "a record's bucket lies outside [%llu, +%llu)", (unsigned long long)bucket_begin, (unsigned long long)n_buckets); goto done; }
        // one workgroup streams a large group: fine for the repeats of a genome, not for an index that IS one bucket
        if (h_stats[1] > large_cap || h_stats[0] > (1u << 22)) goto done;           // *done stays 0
        FinishArgs f;
        f.rows = cur_rows; f.keys = cur_keys; f.gbegin = gbegin; f.gend = gend; f.n_groups = n_groups; f.L = L; f.lbits = L;
        f.n_buckets = n_buckets; f.skip_frequencies = skip_frequencies; f.h2i = (int32_t *)d_hashes_to_index;
        f.nk = (uint32_t *)d_n_kmers; f.o_kmers = (uint64_t *)d_out_kmers; f.o_nodes = (uint32_t *)d_out_nodes;
        f.o_refs = (uint64_t *)d_out_ref_offsets; f.o_af = (uint32_t *)d_out_af32; f.o_freq = (uint16_t *)d_out_frequencies;
        f.o_perm = (uint32_t *)d_out_permutation; f.af_in = (const uint32_t *)d_af32;
        f.big_buckets = big; f.n_big = stats + 2; f.big_cap = big_cap;
        f.xcd_groups = (int)ceil_div(n_groups, 8);
        if (grouped && wrank) hipLaunchKernelGGL((k_group_finish<true, GROUP_CAP_BIG, GROUP_THREADS_BIG>), dim3((unsigned)(f.xcd_groups * 8)), dim3(GROUP_THREADS_BIG), 0, s, f);
        else if (grouped) hipLaunchKernelGGL((k_group_finish<false, GROUP_CAP_BIG, GROUP_THREADS_BIG>), dim3((unsigned)(f.xcd_groups * 8)), dim3(GROUP_THREADS_BIG), 0, s, f);
        else if (wrank) hipLaunchKernelGGL((k_group_finish<true>), dim3((unsigned)(f.xcd_groups * 8)), dim3(GROUP_THREADS), 0, s, f);
        else hipLaunchKernelGGL((k_group_finish<false>), dim3((unsigned)(f.xcd_groups * 8)), dim3(GROUP_THREADS), 0, s, f);
        HIP_G(hipGetLastError());
        if (h_stats[1] > 0) {
            const unsigned n_large = h_stats[1];
            hipLaunchKernelGGL(k_group_large, dim3(n_large < 1024 ? n_large : 1024), dim3(256), 0, s, f, large, n_large);
            HIP_G(hipGetLastError());
        }
        if (!skip_frequencies) {
            // frequencies still open: the rows of large groups (all their buckets) and, from k_group_finish, buckets of
            // more than SMALL_BUCKET rows.  Both go through the first form's kernels over the finished columns.
            HIP_G(hipMemcpyAsync(h_stats, stats, 16, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            if (h_stats[1] > 0 || h_stats[2] > 0) {
                const int n_ranges = (int)h_stats[1] + (int)h_stats[2];
                HIP_G(gki_dev_malloc((void **)&rng, (size_t)n_ranges * 16));
                hipLaunchKernelGGL(k_open_ranges, dim3(stream_grid(n_ranges, 256)), dim3(256), 0, s, large, h_stats[1], gbegin, gend,
                                   big, h_stats[2], (const int32_t *)d_hashes_to_index, (const uint32_t *)d_n_kmers, rng,
                                   rng + n_ranges);
                HIP_G(hipGetLastError());
                rc = gki_frequencies_for_rows(rng, rng + n_ranges, n_ranges, modulo, bucket_begin, d_hashes_to_index, d_n_kmers,
                                              d_out_kmers, d_out_ref_offsets, d_out_frequencies, n, s);
                if (rc != GKI_OK) goto done;
            }
        }
        HIP_G(hipStreamSynchronize(s));
        *done = 1;
    }
done:
    for (int i = 0; i < 2; i++) { (void)gki_dev_free(rows[i]); (void)gki_dev_free(keys[i]); }
    (void)gki_dev_free(hist); (void)gki_dev_free(offs); (void)gki_dev_free(tmp); (void)gki_dev_free(gbegin); (void)gki_dev_free(gend);
    (void)gki_dev_free(large); (void)gki_dev_free(big); (void)gki_dev_free(stats); (void)gki_dev_free(rng);
    if (d_tiles) { (void)hipStreamSynchronize(s); (void)gki_dev_free(d_tiles); (void)gki_dev_free(d_seg); }   // (the upload read h_seg asynchronously)
#undef HIP_G
    return rc;
}

// ------------------------------------------------------------------------------------ bucket-range partition
// gki_partition_by_bucket_range through ONE pass of the kernel above: key = owning part (<= 256 parts = 8 bits), the four
// input columns in, the four output columns out, stable.  Per record: 8 bytes read for the histogram (the part is computed
// from the k-mer, twice, instead of stored and read back twice) + 24 read + 24 written = 56 bytes, where round 3 moved 68
// and the pair-sorting route (sort (part, index) pairs, pack 32-byte rows, gather them) 148.
// Any number of records: the pass runs over chunks of < 2^31 rows (the tile offsets are 32-bit) and every chunk's runs of a
// part go behind the earlier chunks' -- the output is the stable partition of the whole input, as one pass would leave it.
namespace {
constexpr int MAX_CHUNKS = 16;
struct ChunkOffs { const uint32_t *offs[MAX_CHUNKS]; int64_t n_tiles[MAX_CHUNKS]; int64_t hist_n[MAX_CHUNKS]; int n_chunks; };

// start[d] = first output row of digit d (start[n_digits] = n); dbase[c * bins + d] = first output row of chunk c's digit d.
// One block, one thread per digit (bins <= 1024).
__global__ __launch_bounds__(1024) void k_part_bases(ChunkOffs co, int n_digits, int bins, int64_t *__restrict__ start, int64_t *__restrict__ dbase) {
    __shared__ int64_t sh[1024];
    const int d = threadIdx.x;
    int64_t cnt[MAX_CHUNKS], tot = 0;
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; c++) {
        cnt[c] = 0;
        if (c < co.n_chunks && d < bins)
            cnt[c] = (int64_t)co.offs[c][(int64_t)(d + 1) * co.n_tiles[c]] - (int64_t)co.offs[c][(int64_t)d * co.n_tiles[c]];   // d + 1 == bins: the scan's total
        tot += cnt[c];
    }
    sh[d] = tot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t v = d >= off ? sh[d - off] : 0;
        __syncthreads();
        sh[d] += v;
        __syncthreads();
    }
    int64_t at = sh[d] - tot;
    if (d <= n_digits) start[d] = at;
    if (d == 1023 && n_digits == 1024) start[1024] = sh[1023];
    if (d < bins) {
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; c++)
            if (c < co.n_chunks) { dbase[c * bins + d] = at; at += cnt[c]; }
    }
}
}  // namespace

template <int THREADS>
static int partition_columns_by_part(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                                     uint64_t modulo, int n_parts, int sub_bits, int64_t max_rows_per_pass, void *d_out_kmers,
                                     void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32, void *d_out_rows, void *d_out_keys,
                                     int64_t *h_part_start) {
    constexpr int RI = 8, TILE = THREADS * RI;
    hipStream_t s = 0;
    const int n_digits = n_parts << sub_bits;
    int bits = 0;
    while ((1 << bits) < n_digits) bits++;
    const int bins = 1 << bits;
    int64_t chunk_rows = (((int64_t)1 << 31) - 1) / TILE * TILE;
    if (max_rows_per_pass > 0 && max_rows_per_pass < chunk_rows) chunk_rows = ceil_div(max_rows_per_pass, TILE) * TILE;
    const int n_chunks = (int)ceil_div(n, chunk_rows);
    if (n_chunks > MAX_CHUNKS) return gki_set_error(GKI_ERR_OVERFLOW, "%lld records: partition at most %lld at a time", (long long)n, (long long)(chunk_rows * MAX_CHUNKS));
    uint32_t h_pb[PB_WORDS];
    for (int p = 0; p <= n_parts; p++) h_pb[p] = (uint32_t)(modulo * (uint64_t)p / (uint64_t)n_parts);
    for (int p = 0; p < n_parts; p++) {
        const int kbp = h_pb[p + 1] > h_pb[p] ? key_bits((uint64_t)(h_pb[p + 1] - h_pb[p]) - 1) : 0;
        h_pb[MAX_PARTS + 1 + p] = (uint32_t)(kbp > sub_bits ? kbp - sub_bits : 0);
    }
    uint32_t *hist = nullptr, *offs[MAX_CHUNKS] = {nullptr}, *pb = nullptr;
    int64_t *pstart = nullptr, *dbase = nullptr;
    int *bad = nullptr;
    void *tmp = nullptr;
    ChunkOffs co;
    co.n_chunks = n_chunks;
    KeyRule rule;
    rule.mod = gki_mod_of(modulo); rule.bucket_begin = 0; rule.n_buckets = modulo; rule.n_parts = n_parts; rule.sub_bits = sub_bits;
    rule.parts_per_bucket = (float)((double)n_parts / (double)modulo);
    rule.parts_log2 = -1; rule.by_node = 0;
    for (int l = 0; l <= 8; l++) if ((1 << l) == n_parts) rule.parts_log2 = l;
    int rc = GKI_OK;
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        const int64_t max_tiles = ceil_div(n < chunk_rows ? n : chunk_rows, TILE);
        const int64_t tmp_bytes = gki_scan_tmp_bytes((int64_t)bins * max_tiles);
        HIP_G(gki_dev_malloc((void **)&hist, (size_t)bins * max_tiles * 4));
        HIP_G(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
        HIP_G(gki_dev_malloc((void **)&pb, PB_WORDS * 4));
        HIP_G(gki_dev_malloc((void **)&pstart, (size_t)(1024 + 1) * 8));
        HIP_G(gki_dev_malloc((void **)&dbase, (size_t)n_chunks * bins * 8));
        HIP_G(gki_dev_malloc((void **)&bad, 16));
        HIP_G(hipMemcpyAsync(pb, h_pb, PB_WORDS * 4, hipMemcpyHostToDevice, s));
        rule.part_begin = pb; rule.sub_shift = pb + MAX_PARTS + 1;
        for (int c = 0; c < n_chunks; c++) {
            const int64_t c0 = (int64_t)c * chunk_rows, nc = (n - c0) < chunk_rows ? (n - c0) : chunk_rows;
            const int64_t n_tiles = ceil_div(nc, TILE), hist_n = (int64_t)bins * n_tiles;
            co.n_tiles[c] = n_tiles; co.hist_n[c] = hist_n;
            HIP_G(gki_dev_malloc((void **)&offs[c], (size_t)(hist_n + 1) * 4));
            co.offs[c] = offs[c];
            const int xcd_tiles = (int)ceil_div(n_tiles, 8);
            hipLaunchKernelGGL((k_kmer_digit_hist<THREADS, RI>), dim3((unsigned)(xcd_tiles * 8)), dim3(THREADS), 0, s, (const uint64_t *)d_kmers + c0,
                               nc, rule, 0, bits, hist, n_tiles, xcd_tiles, (const TileDesc *)nullptr, bad);
            HIP_G(hipGetLastError());
            rc = gki_scan_u32_to_u32(hist, hist_n, offs[c], tmp, tmp_bytes, s);
            if (rc != GKI_OK) goto done;
        }
        hipLaunchKernelGGL(k_part_bases, dim3(1), dim3(1024), 0, s, co, n_digits, bins, pstart, dbase);
        HIP_G(hipGetLastError());
        for (int c = 0; c < n_chunks; c++) {
            const int64_t c0 = (int64_t)c * chunk_rows, nc = (n - c0) < chunk_rows ? (n - c0) : chunk_rows;
            PartArgs a;
            a.tiles = nullptr;
            a.keys_in = nullptr; a.c_kmers = (const uint64_t *)d_kmers + c0; a.c_nodes = (const uint32_t *)d_nodes + c0;
            a.c_refs = (const uint64_t *)d_ref_offsets + c0; a.c_af = (const uint32_t *)d_af32 + c0; a.rule = rule; a.rows_in = nullptr;
            a.n = nc; a.n_tiles = co.n_tiles[c]; a.shift = 0; a.bits = bits; a.offs = offs[c]; a.rows_out = nullptr; a.keys_out = nullptr;
            a.o_kmers = (uint64_t *)d_out_kmers; a.o_nodes = (uint32_t *)d_out_nodes; a.o_refs = (uint64_t *)d_out_ref_offsets;
            a.o_af = (uint32_t *)d_out_af32; a.dbase = dbase + (int64_t)c * bins; a.carry_index = 0;
            a.xcd_tiles = (int)ceil_div(a.n_tiles, 8);
            if (d_out_rows) {               // rows + keys out (the key: the bucket's offset in its part)
                a.rows_out = (uint64_t *)d_out_rows; a.keys_out = (uint32_t *)d_out_keys;
                bool launched = false;
                if constexpr (GKI_PT_NCH > 1 && THREADS == GKI_PT_THREADS) {
                    if (n < ((int64_t)1 << 32)) {
                        hipLaunchKernelGGL((k_partition_rows_staged<THREADS, RI, true, GKI_PT_NCH>), dim3((unsigned)(a.xcd_tiles * 8)), dim3(THREADS), 0, s, a);
                        launched = true;
                    }
                }
                if (!launched)
                    hipLaunchKernelGGL((k_partition_rows<THREADS, RI, true, false>), dim3((unsigned)(a.xcd_tiles * 8)), dim3(THREADS), 0, s, a);
            } else
                hipLaunchKernelGGL((k_partition_rows<THREADS, RI, true, true>), dim3((unsigned)(a.xcd_tiles * 8)), dim3(THREADS), 0, s, a);
            HIP_G(hipGetLastError());
        }
        HIP_G(hipMemcpyAsync(h_part_start, pstart, (size_t)(n_digits + 1) * 8, hipMemcpyDeviceToHost, s));
        HIP_G(hipStreamSynchronize(s));
    }
done:
    (void)gki_dev_free(hist); (void)gki_dev_free(tmp); (void)gki_dev_free(pb); (void)gki_dev_free(pstart); (void)gki_dev_free(dbase);
    (void)gki_dev_free(bad);
    for (int c = 0; c < n_chunks; c++) (void)gki_dev_free(offs[c]);
#undef HIP_G
    return rc;
}

// sub_bits > 0: grouped -- part p's records leave grouped by the top sub_bits bits of (bucket - part_begin[p]); h_part_start has
// (n_parts << sub_bits) + 1 entries, entry p << sub_bits | g = first row of group g of part p.  Columns out: up to 256 digits
// go through 2048-row tiles (two workgroups per CU), more through 4096-row tiles.  Rows out (d_out_rows / d_out_keys instead of
// the four columns: 24-byte rows and the bucket's offset in its part as the key): 4096-row tiles -- a run of four ROWS is
// 96 + 16 contiguous bytes where four records of four columns are runs of 32, 32, 16 and 16 (1024 digits into columns: 98 ms per
// 3.16e9 records, into rows 61).
int gki_partition_columns_by_part(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32, int64_t n,
                                  uint64_t modulo, int n_parts, int sub_bits, int64_t max_rows_per_pass, void *d_out_kmers,
                                  void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32, void *d_out_rows, void *d_out_keys,
                                  int64_t *h_part_start) {
    if (!d_out_rows && (n_parts << sub_bits) <= MAX_PARTS)
        return partition_columns_by_part<GKI_PR_THREADS>(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, n_parts, sub_bits, max_rows_per_pass,
                                                         d_out_kmers, d_out_nodes, d_out_ref_offsets, d_out_af32, nullptr, nullptr, h_part_start);
    return partition_columns_by_part<512>(d_kmers, d_nodes, d_ref_offsets, d_af32, n, modulo, n_parts, sub_bits, max_rows_per_pass,
                                          d_out_kmers, d_out_nodes, d_out_ref_offsets, d_out_af32, d_out_rows, d_out_keys, h_part_start);
}
