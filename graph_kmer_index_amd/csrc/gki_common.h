// Shared device/host helpers of libgki_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>
#include <stdio.h>
#include <string.h>
#include "../../include/gki.h"

#define GKI_WAVE 64

extern thread_local char gki_err_buf[512];
int gki_set_error(int code, const char *fmt, ...);

#define HIP_TRY(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,     \
                                 hipGetErrorString(e_));                                       \
    } while (0)

#define GKI_TRY(call)                  \
    do {                               \
        int r_ = (call);               \
        if (r_ != GKI_OK) return r_;   \
    } while (0)

// pooled device memory (gki_runtime.hip): every buffer of the library is allocated and freed through these
hipError_t gki_dev_malloc(void **ptr, size_t bytes);
hipError_t gki_dev_free(void *ptr);
template <typename T> static inline hipError_t gki_dev_malloc(T **ptr, size_t bytes) { return gki_dev_malloc((void **)ptr, bytes); }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid for a grid-stride streaming kernel: enough blocks to fill 256 CUs x 8 blocks, no more.
static inline int stream_grid(int64_t work_items, int block) {
    int64_t g = ceil_div(work_items, block);
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

// ---------------------------------------------------------------------------------- 2-bit sequence
// seq2: bases packed 32 per uint64, base i of the graph at bits [2*(i%32), 2*(i%32)+2) of word i/32.
// With the reference's hash (first base least significant, kmer_hashing.py:4-9) the hash of the
// t bases starting at global position P is simply a 2t-bit field of that bit stream.
__device__ __forceinline__ uint64_t gki_extract(const uint64_t *__restrict__ seq2, int64_t P, int t) {
    int64_t w = P >> 5;
    int sh = (int)(P & 31) * 2;
    uint64_t lo = seq2[w];
    uint64_t hi = seq2[w + 1];
    uint64_t v = (lo >> sh) | ((hi << 1) << (63 - sh));       // branch-free funnel shift (sh may be 0)
    return v & ((1ull << (2 * t)) - 1ull);   // t <= 31
}

// ---------------------------------------------------------------------------------- x % modulo
// bucket = kmer % modulo (collision_free_kmer_index.py:304, :433) without the 64-bit division routine (~100 instructions
// for a run-time divisor): q = mulhi(x, floor((2^64 - 1) / modulo)) is the quotient or one or two short.
struct GkiMod { uint64_t m, inv; };
static inline GkiMod gki_mod_of(uint64_t modulo) { GkiMod d; d.m = modulo; d.inv = ~0ull / modulo; return d; }
__device__ __forceinline__ uint64_t gki_mod(const GkiMod &d, uint64_t x) {
    uint64_t r = x - __umul64hi(x, d.inv) * d.m;
    while (r >= d.m) r -= d.m;
    return r;
}

// ---------------------------------------------------------------------------------- device graph view
struct alignas(32) NodeWalk { // everything the boundary walk needs about a node, one aligned 32-B record
    int64_t seq_start;
    uint64_t tail;           // the node's last min(size, 31) bases, 2 bits each, the first of them least significant:
                             // the context a successor's window takes from this node, without touching the sequence
    int32_t rev_begin;       // first predecessor in rev_edges -- or, when rev_cnt == 1, the predecessor itself
    int32_t size;
    float af;                // allele frequency as float32 (flat_kmers.py:90); the float64 value is DevGraph::allele_freq
    uint16_t rev_cnt;        // number of predecessors (0xFFFF: 65535 or more, see rev_start)
    uint8_t is_ref, pad;
};
static_assert(sizeof(NodeWalk) == 32, "NodeWalk is one 32-byte record");
static_assert(offsetof(NodeWalk, seq_start) == 0 && offsetof(NodeWalk, tail) == 8, "csrc/gki_forward.hip reads (seq_start, tail) as the record's first 16 bytes");

// The early-stop search's view of a node (csrc/gki_forward.hip): one aligned 32-byte record holds everything a forward step
// onto the node needs -- its first bases, its allele frequency, and its successors when there are at most two -- so a
// descent is ONE record instead of the walk record, two sequence words, two edge offsets and an edge.
struct alignas(32) NodeFwd {
    uint64_t head;           // the node's first min(size, 31) bases, 2 bits each, the first of them least significant
    double af;               // allele frequency (float64, as the finder reports it)
    int32_t size;
    int32_t e0;              // cnt == 1, 2: the first successor itself; cnt > 2: the index of the first successor in edges
    int32_t e1;              // cnt == 2: the second successor
    uint16_t cnt;            // number of successors (0xFFFF: 65535 or more, see edge_start)
    uint8_t is_ref, pad;
};
static_assert(sizeof(NodeFwd) == 32, "NodeFwd is one 32-byte record");

struct DevGraph {
    int64_t n_nodes, n_bases, n_words64;     // n_words64 = ceil(n_bases / 64): one bitmap word per 64 bases
    const int32_t *node_size;
    const int64_t *seq_start;                // [n_nodes+1]
    const uint8_t *seq;                      // uint8 bases (kept for re-prepare)
    const uint64_t *seq2;                    // 2-bit packed, padded
    const int64_t *rev_start;
    const int32_t *rev_edges;
    const int64_t *edge_start;
    const int32_t *edges;
    const uint8_t *is_ref;
    const double *allele_freq;
    const int64_t *pos_base;                 // position id of (node, 0)
    const uint64_t *start_mask;              // bit p%64 of word p/64 set iff a non-empty node starts at base p
    const uint32_t *start_rank;              // number of node starts in words before this one
    const NodeWalk *walk;                    // [n_nodes] packed per-node records for the predecessor walk
    const int32_t *nonempty;                 // ids of non-empty nodes, ascending (= sequence order)
    const int32_t *node_rank;                // [n_nodes] rank of a non-empty node in `nonempty` (unused for empty ones)
    int64_t n_nonempty;
};

// Predecessor iteration: (cur, end) index rev_edges; cur < 0 stands for "the single predecessor ~cur, not visited yet".
__device__ __forceinline__ void preds_begin(const DevGraph &g, const NodeWalk &w, int64_t n, int32_t *cur, int32_t *end) {
    if (w.rev_cnt == 1) { *cur = ~w.rev_begin; *end = 0; }
    else { *cur = w.rev_begin; *end = w.rev_cnt == 0xFFFF ? (int32_t)g.rev_start[n + 1] : w.rev_begin + (int32_t)w.rev_cnt; }
}
__device__ __forceinline__ int32_t preds_next(const DevGraph &g, int32_t *cur) {
    if (*cur < 0) { const int32_t q = ~*cur; *cur = 0; return q; }
    return g.rev_edges[(*cur)++];
}
// the last t (<= min(size, 31)) bases of the node, first of them least significant
__device__ __forceinline__ uint64_t node_tail(const NodeWalk &w, int t) {
    const int t31 = w.size < 31 ? w.size : 31;
    return (w.tail >> (2 * (t31 - t))) & ((1ull << (2 * t)) - 1ull);
}

// arena behind the slow path of a graph walk (see "stacks of the graph walks" below): cap levels for each of `lanes` lanes
struct DeepArena { char *base; int64_t lanes; int32_t cap; int32_t pad; };

// early-stop search: the finished k-mers the count pass wrote down for the emit pass (csrc/gki_forward.hip, "script"), and
// the call they belong to -- the emit call uses them only when it is given the same arguments
struct FwdScript {
    void *entries;                           // FW_SLOTS entries of up to three 16-byte pieces per start position, piece-major
    uint8_t *ncomp;                          // [n_pos] entries in use, 0xFF: the start position did not fit, walk it again
    int64_t n_pos, overflow;
    int64_t *over_list; int64_t over_cap;    // the start positions that did not fit (the first over_cap of them), for the emit pass
    const void *nodes, *offsets, *follow, *rec_start;
    int k, M, one_node, valid;
};

struct gki_graph {
    DevGraph d;
    int64_t *h_seq_start;                    // host copy of d.seq_start [n_nodes+1] (chunk bounds without a device read)
    hipStream_t stream;
    int device;
    void *owned[24];
    int n_owned;
    bool owns_seq;
    hipEvent_t ev_prep0, ev_prep1;
    DeepArena fwd_deep;                      // early-stop search: arena of its slow path, cap > 0 after a count call that needed it
    int64_t fwd_deep_bytes;
    FwdScript fwd_script;                    // early-stop search: what the count call left for the emit call (csrc/gki_forward.hip)
    NodeFwd *fwd_nodes;                      // early-stop search: its per-node records, built by the first search after a prepare
};

// GKI_ERR_BAD_ARG unless the device that is current is the one the graph was uploaded to (gki_finder.hip)
int gki_check_graph_device(const gki_graph *g, const char *who);

// ---------------------------------------------------------------------------------- device error word
// Kernels report the two conditions a run can end on through ONE word, as bits, so that the outcome does not depend on
// which lane stored last (plain stores of different codes raced): bit 0 = the reference's assertion
// (kmer_finder.py:402, GKI_ERR_NOT_ONE_REF_SUCC), bit 1 = a window or history deeper than the kernels' stacks
// (GKI_ERR_WINDOW_TOO_DEEP).  Too-deep wins on the host: a run that could not look at every window cannot vouch for
// the assertion either way.
__device__ __forceinline__ void gki_raise(int *err, int code) {
    atomicOr((unsigned int *)err, code == GKI_ERR_WINDOW_TOO_DEEP ? 2u : 1u);
}
// bit 2 = too deep for a reason no deeper stack cures (a history enumeration out of its step budget): the finder's
// slow path (deep kernel variants, gki_finder_count) is not tried for it
__device__ __forceinline__ void gki_raise_budget(int *err) { atomicOr((unsigned int *)err, 4u); }
static inline int gki_error_of_word(int64_t word) {
    return (word & 6) ? GKI_ERR_WINDOW_TOO_DEEP : (word & 1) ? GKI_ERR_NOT_ONE_REF_SUCC : GKI_OK;
}

// ---------------------------------------------------------------------------------- stacks of the graph walks
// Where the levels below the top live.  The product kernels keep MAXN of them per lane in scratch (LocalStack: a plain
// array); a window over more than MAXN - 2 nodes -- sixteen or more EMPTY nodes inside one 31-base window -- makes them
// raise GKI_ERR_WINDOW_TOO_DEEP, and gki_finder_count then runs the pass again with the DEEP instantiation of the same
// kernels, whose stacks lie in a global-memory arena sized for the run (ArenaStack: level-major, lane-interleaved, so a
// wave's accesses to one level coalesce; every lane reads only what it wrote).  Same walk, same arithmetic: the slow path
// differs in where a level is stored, in counters wide enough for its depth, and in all-nodes mode in who writes a
// many-node window (its lane alone).
template <class T, int N> struct LocalStack {
    T v[N];
    __device__ __forceinline__ T &operator[](int i) { return v[i]; }
    __device__ __forceinline__ const T &operator[](int i) const { return v[i]; }
};
template <class T> struct ArenaStack {
    T *base; int64_t stride;
    __device__ __forceinline__ T &operator[](int i) const { return base[(int64_t)i * stride]; }
};
template <class T> __device__ __forceinline__ ArenaStack<T> arena_stack(const DeepArena &da, int offset, int64_t lane_global) {
    ArenaStack<T> st;
    st.base = reinterpret_cast<T *>(da.base + (int64_t)da.cap * da.lanes * offset) + lane_global;
    st.stride = da.lanes;
    return st;
}
template <class T, int N, bool DEEP> struct StackOf { typedef LocalStack<T, N> type; };
template <class T, int N> struct StackOf<T, N, true> { typedef ArenaStack<T> type; };
template <class T, int N> __device__ __forceinline__ void bind(LocalStack<T, N> &, const DeepArena &, int, int64_t) {}
template <class T> __device__ __forceinline__ void bind(ArenaStack<T> &st, const DeepArena &da, int offset, int64_t lane_global) {
    st = arena_stack<T>(da, offset, lane_global);
}
template <class T, int N> __device__ __forceinline__ T *raw(LocalStack<T, N> &st) { return st.v; }
template <class T> __device__ __forceinline__ ArenaStack<T> raw(const ArenaStack<T> &st) { return st; }
template <bool DEEP> struct CountOf { typedef uint8_t T; };      // variant-node counters of a suspended level
template <> struct CountOf<true> { typedef uint16_t T; };


// ---------------------------------------------------------------------------------- wave prefix sums (DPP)
// Inclusive prefix sum over the 64 lanes of a wave on the data-parallel-primitive path: row shifts by 1, 2, 4, 8 inside
// the four rows of 16 lanes (out-of-row sources read as zero), then lane 15 of a row broadcast into the next row and
// lane 31 into the upper half -- six VALU additions.  The shuffle form (__shfl_up in a loop) compiles to six dependent
// ds_bpermute_b32, i.e. six LDS round trips of ~100 cycles; the finder's expansion runs five such scans per queue and
// one per trip in all-nodes mode -- 40 % of that kernel's LDS instructions.  Every lane of the wave must be active.
// gki_selftest_wave_scan (gki_measure.hip) checks it against the shuffle form on the device.
__device__ __forceinline__ int gki_wave_incl_sum(int x) {
#ifdef GKI_SCAN_SHUFFLE                                                 // A/B builds: the shuffle form
    const int lane_ = (int)(threadIdx.x & 63);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(x, d, 64); if (lane_ >= d) x += t; }
    return x;
#endif
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);    // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);    // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t gki_wave_incl_sum(uint32_t x) { return (uint32_t)gki_wave_incl_sum((int)x); }
// the value lane `l` (a constant) holds, in a scalar register
__device__ __forceinline__ int gki_lane_value(int x, int l) { return __builtin_amdgcn_readlane(x, l); }

// ---------------------------------------------------------------------------------- exclusive scan
// out[0..n] (n+1 entries), out[n] = total.  Three launches: block sums, scan of block sums, rescan.
int gki_scan_u32_to_i64(const uint32_t *d_in, int64_t n, int64_t *d_out, void *d_tmp, int64_t tmp_bytes,
                        hipStream_t s);
int gki_scan_i32_to_i64(const int32_t *d_in, int64_t n, int64_t *d_out, void *d_tmp, int64_t tmp_bytes,
                        hipStream_t s);
int gki_scan_u32_to_u32(const uint32_t *d_in, int64_t n, uint32_t *d_out, void *d_tmp, int64_t tmp_bytes,
                        hipStream_t s);
int64_t gki_scan_tmp_bytes(int64_t n);

// uint8 bases -> 2-bit stream (out sized ceil(n/16) uint32), asynchronous on `s` (gki_graph.hip).
int gki_launch_pack(const uint8_t *d_seq, int64_t n_bases, uint32_t *d_out, hipStream_t s);
