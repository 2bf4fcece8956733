// DenseKmerFinder on MI355X: every (end position, backward k-window) of the graph, data-parallel.
//
// Formulation (SURVEY.md 8a'): the reference's DFS (kmer_finder.py:254-417)
// emits, for every base position e = (node, offset) and every backward path P of exactly k real
// bases ending at e whose window holds at most `max_variant_nodes` non-linear-ref nodes, one record
// per distinct node of P.  End positions are independent, so
//   * positions whose window lies inside their own node ("interior": offset >= bnd_len[node],
//     normally k-1) have exactly one window and one node: a pure streaming kernel
//     (k_emit_interior) -- this is ~90-100% of all records;
//   * the first bnd_len bases of every node ("boundary") walk the predecessor lists backwards with
//     a small per-lane stack (k_count_boundary / k_emit_boundary).
// Output slots come from a count pass + exclusive scans, never from atomics, so the record order is
// deterministic: by (node id, offset), windows in predecessor-list order, nodes ascending.
#include "gki_common.h"
#include <limits.h>
#include <math.h>

namespace {

constexpr int MAXN = GKI_MAX_WINDOW_NODES;

struct NodeEmit {       // per-node constants of the interior kernel, 32 B, one 2x16-B gather
    int64_t seq_start;  // global base index of (node, 0)
    int64_t D;          // record index of interior position p is p + D
    int64_t E;          // position id of base p is p + E
    int32_t lo;         // first interior offset of this node in this run
    float af;           // allele frequency as float32 (flat_kmers.py:90)
};

struct FindArgs {
    int32_t k, M, one_node, has_lossy;
    int64_t node_begin, off_begin, node_end, off_end;
};

struct OutFlat { uint64_t *hash; uint32_t *node; uint64_t *ref_offset; float *af; };
struct OutV2 { int64_t *hash; int32_t *start_node; int16_t *start_offset; int32_t *node; double *af; };

__device__ __forceinline__ bool in_range(const FindArgs &a, int64_t n, int64_t o) {
    if (n < a.node_begin || n > a.node_end) return false;
    if (n == a.node_begin && o < a.off_begin) return false;
    if (n == a.node_end && o >= a.off_end) return false;
    return true;
}

// SURVEY.md 8a' E1: a restart at a critical point (N, c) with 0 < c < k-1 is not rewound
// (kmer_finder.py:231-232), so no window contains both (N, c-1) and (N, c).
__device__ __forceinline__ bool lossy_hit(const uint16_t *__restrict__ lossy, int32_t node, int a, int b) {
    int c = lossy[node];
    return c != 0xFFFF && a <= c - 1 && c <= b;
}

// ------------------------------------------------------------------------------------ per-node setup
__global__ __launch_bounds__(256) void k_node_setup(DevGraph g, FindArgs a, const uint16_t *__restrict__ lossy,
                                                    int32_t *__restrict__ bnd_len, int32_t *__restrict__ icnt) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        int32_t size = g.node_size[n];
        int32_t bl = 0, ic = 0;
        if (n >= a.node_begin && n <= a.node_end && size > 0) {
            int32_t reach = a.k - 1;                                       // offsets < k-1 look into predecessors
            if (a.has_lossy && lossy[n] != 0xFFFF) reach = lossy[n] + a.k - 1;   // E1 windows end up to c+k-2
            if (!g.is_ref[n] && a.M < 1) reach = size;                     // variant node, limit 0: nothing admissible
            bl = size < reach ? size : reach;
            int64_t lo = bl, hi = size;
            if (n == a.node_begin && a.off_begin > lo) lo = a.off_begin;
            if (n == a.node_end && a.off_end < hi) hi = a.off_end;
            ic = hi > lo ? (int32_t)(hi - lo) : 0;
        }
        bnd_len[n] = bl;
        icnt[n] = ic;
    }
}

__global__ __launch_bounds__(256) void k_node_emit(DevGraph g, FindArgs a, const int32_t *__restrict__ bnd_len,
                                                   const int64_t *__restrict__ bnd_start,
                                                   const int64_t *__restrict__ bscan,
                                                   const int64_t *__restrict__ iscan, NodeEmit *__restrict__ ne) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        NodeEmit e;
        int64_t lo = bnd_len[n];
        if (n == a.node_begin && a.off_begin > lo) lo = a.off_begin;
        e.seq_start = g.seq_start[n];
        e.lo = (int32_t)lo;
        e.D = bscan[bnd_start[n + 1]] + iscan[n] - lo - e.seq_start;
        e.E = g.pos_base[n] - e.seq_start;
        e.af = (float)g.allele_freq[n];
        ne[n] = e;
    }
}

// ------------------------------------------------------------------------------------ backward windows
// Depth-first walk over predecessor lists from end position (n, o).  on_window(L) is called with
// nd[0..L) = nodes of the window (end node first) and na[j] = bases still missing after level j.
template <bool HAS_LOSSY, typename F>
__device__ __forceinline__ uint32_t walk_windows(const DevGraph &g, const uint16_t *__restrict__ lossy, int k, int M,
                                                 int32_t n, int32_t o, int32_t *nd, int32_t *cur, uint8_t *na,
                                                 uint8_t *vc, int *err, F &&on_window) {
    int t = o + 1 < k ? o + 1 : k;
    if (HAS_LOSSY && lossy_hit(lossy, n, o + 1 - t, o)) return 0;
    int v = g.is_ref[n] ? 0 : 1;
    if (v > M) return 0;
    nd[0] = n; na[0] = (uint8_t)(k - t); vc[0] = (uint8_t)v;
    if (k - t == 0) return on_window(1);
    cur[0] = (int32_t)g.rev_start[n];
    uint32_t total = 0;
    int L = 1;
    while (L > 0) {
        const int j = L - 1;
        const int32_t e = cur[j];
        if ((int64_t)e >= g.rev_start[nd[j] + 1]) { L--; continue; }
        cur[j] = e + 1;
        const int32_t q = g.rev_edges[e];
        const int vq = vc[j] + (g.is_ref[q] ? 0 : 1);
        if (vq > M) continue;                       // kmer_finder.py:391-403 in order-free form
        if (L >= MAXN) { *err = GKI_ERR_WINDOW_TOO_DEEP; continue; }
        const int s = g.node_size[q];
        const int need = na[j];
        nd[L] = q; vc[L] = (uint8_t)vq;
        if (s == 0) {                               // empty node: in the node set, adds no base (:261-265)
            na[L] = (uint8_t)need; cur[L] = (int32_t)g.rev_start[q]; L++;
            continue;
        }
        const int tq = s < need ? s : need;
        if (HAS_LOSSY && lossy_hit(lossy, q, s - tq, s - 1)) continue;
        na[L] = (uint8_t)(need - tq);
        if (need == tq) { total += on_window(L + 1); continue; }
        cur[L] = (int32_t)g.rev_start[q]; L++;
    }
    return total;
}

__device__ __forceinline__ int64_t node_of_boundary_index(const int64_t *__restrict__ bnd_start, int64_t n_nodes, int64_t i) {
    int64_t lo = 0, hi = n_nodes;           // largest n with bnd_start[n] <= i
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (bnd_start[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

template <bool HAS_LOSSY>
__global__ __launch_bounds__(256) void k_count_boundary(DevGraph g, FindArgs a, const uint16_t *__restrict__ lossy,
                                                        const int64_t *__restrict__ bnd_start, int64_t B,
                                                        uint32_t *__restrict__ cnt, int *__restrict__ err) {
    int32_t nd[MAXN], cur[MAXN];
    uint8_t na[MAXN], vc[MAXN];
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
        int64_t n = node_of_boundary_index(bnd_start, g.n_nodes, i);
        int32_t o = (int32_t)(i - bnd_start[n]);
        uint32_t c = 0;
        if (in_range(a, n, o)) {
            const bool one = a.one_node;
            c = walk_windows<HAS_LOSSY>(g, lossy, a.k, a.M, (int32_t)n, o, nd, cur, na, vc, err,
                                        [&](int L) -> uint32_t { return one ? 1u : (uint32_t)L; });
        }
        cnt[i] = c;
    }
}

template <int FMT> struct OutSel;
template <> struct OutSel<0> { typedef OutFlat T; };
template <> struct OutSel<1> { typedef OutV2 T; };

__device__ __forceinline__ void put(const OutFlat &o, int64_t idx, uint64_t h, int32_t node, int32_t end_node,
                                    int32_t end_off, int64_t pos_id, double af) {
    (void)end_node; (void)end_off;
    if (o.hash) o.hash[idx] = h;
    if (o.node) o.node[idx] = (uint32_t)node;
    if (o.ref_offset) o.ref_offset[idx] = (uint64_t)pos_id;
    if (o.af) o.af[idx] = (float)af;
}
__device__ __forceinline__ void put(const OutV2 &o, int64_t idx, uint64_t h, int32_t node, int32_t end_node,
                                    int32_t end_off, int64_t pos_id, double af) {
    (void)pos_id;
    if (o.hash) o.hash[idx] = (int64_t)h;
    if (o.start_node) o.start_node[idx] = end_node;
    if (o.start_offset) o.start_offset[idx] = (int16_t)end_off;
    if (o.node) o.node[idx] = node;
    if (o.af) o.af[idx] = af;
}

template <bool HAS_LOSSY, int FMT>
__global__ __launch_bounds__(256) void k_emit_boundary(DevGraph g, FindArgs a, const uint16_t *__restrict__ lossy,
                                                       const int64_t *__restrict__ bnd_start, int64_t B,
                                                       const int64_t *__restrict__ bscan,
                                                       const int64_t *__restrict__ iscan,
                                                       typename OutSel<FMT>::T out, int *__restrict__ err) {
    int32_t nd[MAXN], cur[MAXN];
    uint8_t na[MAXN], vc[MAXN];
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
        if (bscan[i + 1] == bscan[i]) continue;           // nothing to write for this position
        int64_t n = node_of_boundary_index(bnd_start, g.n_nodes, i);
        int32_t o = (int32_t)(i - bnd_start[n]);
        int64_t idx = bscan[i] + iscan[n];
        const int64_t pos_id = g.pos_base[n] + o;
        const int k = a.k;
        const bool one = a.one_node;
        walk_windows<HAS_LOSSY>(g, lossy, k, a.M, (int32_t)n, o, nd, cur, na, vc, err, [&](int L) -> uint32_t {
            uint64_t h = 0;
            int32_t mn = INT_MAX;
            double maf = INFINITY;
            int need_before = k;
            for (int j = 0; j < L; j++) {
                const int32_t q = nd[j];
                mn = q < mn ? q : mn;
                maf = fmin(maf, g.allele_freq[q]);               // np.min, kmer_finder.py:143
                const int t = need_before - na[j];
                if (t > 0) {
                    const int avail = j == 0 ? o + 1 : g.node_size[q];
                    h |= gki_extract(g.seq2, g.seq_start[q] + avail - t, t) << (2 * na[j]);
                }
                need_before = na[j];
            }
            if (one) {                                            // :145-146 nodes[0] of np.unique
                put(out, idx, h, mn, (int32_t)n, o, pos_id, maf);
                idx++;
                return 1u;
            }
            // one record per distinct node, ascending (np.unique, :134): selection by repeated minimum
            int32_t last = INT_MIN;
            for (int r = 0; r < L; r++) {
                int32_t best = INT_MAX;
                for (int j = 0; j < L; j++) { int32_t q = nd[j]; if (q > last && q < best) best = q; }
                put(out, idx, h, best, (int32_t)n, o, pos_id, maf);
                idx++;
                last = best;
            }
            return (uint32_t)L;
        });
    }
}

// ------------------------------------------------------------------------------------ interior stream
// One wave per 64-base word of the sequence, WPW consecutive words per wave trip.  Lane l owns base
// p = 64*w + l: node = rank + popcount(mask bits <= l) - 1 (wave-uniform mask/rank), per-node
// constants from one 32-B gather, hash = 2k-bit field of the 2-bit stream, four coalesced stores.
constexpr int WPW = 8;

template <int FMT>
__global__ __launch_bounds__(256) void k_emit_interior(DevGraph g, FindArgs a, const NodeEmit *__restrict__ ne,
                                                       typename OutSel<FMT>::T out, int64_t word_begin,
                                                       int64_t word_end) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int k = a.k;
    const uint64_t lane_mask = ~0ull >> (63 - lane);       // bits 0..lane
    for (int64_t w0 = word_begin + wave * WPW; w0 < word_end; w0 += n_waves * WPW) {
#pragma unroll
        for (int u = 0; u < WPW; u++) {
            const int64_t w = w0 + u;
            if (w >= word_end) break;
            const uint64_t mask = g.start_mask[w];
            const uint32_t rank = g.start_rank[w];
            const int64_t p = w * 64 + lane;
            if (p >= g.n_bases) continue;
            const uint32_t j = rank + (uint32_t)__popcll(mask & lane_mask) - 1u;
            const int32_t n = g.nonempty[j];
            const NodeEmit e = ne[n];
            const int64_t o = p - e.seq_start;
            if (o < e.lo) continue;
            if (n < a.node_begin || n > a.node_end) continue;
            if (n == a.node_end && o >= a.off_end) continue;
            const uint64_t h = gki_extract(g.seq2, p - (k - 1), k);
            const double af = FMT == 1 ? g.allele_freq[n] : (double)e.af;   // v2 keeps float64 (kmer_finder.py:58)
            put(out, p + e.D, h, n, n, (int32_t)o, p + e.E, af);
        }
    }
}

__global__ void k_totals(const int64_t *bscan, int64_t B, const int64_t *iscan, int64_t n_nodes, const int *err,
                         int64_t *out3) {
    out3[0] = bscan[B];
    out3[1] = iscan[n_nodes];
    out3[2] = *err;
}

}  // namespace

struct gki_finder {
    gki_graph *g;
    hipStream_t stream;
    int32_t *bnd_len, *icnt;
    int64_t *bnd_start, *iscan;
    uint32_t *cnt; int64_t cnt_cap;
    int64_t *bscan;
    NodeEmit *ne;
    uint16_t *lossy;
    void *scan_tmp; int64_t scan_tmp_bytes;
    int *d_err; int64_t *d_totals;
    FindArgs args;
    int64_t B, n_boundary_records, n_interior_records;
    int64_t word_begin, word_end;
    bool counted;
    hipEvent_t ev[10];   // pairs: 0/1 count-boundary, 2/3 emit-interior, 4/5 emit-boundary, 6/7 scans(count phase)
    bool ev_valid[5];
};

static int ensure_cnt(gki_finder *f, int64_t B) {
    if (B <= f->cnt_cap) return GKI_OK;
    if (f->cnt) HIP_TRY(hipFree(f->cnt));
    if (f->bscan) HIP_TRY(hipFree(f->bscan));
    f->cnt = nullptr; f->bscan = nullptr; f->cnt_cap = 0;
    int64_t cap = B + B / 8 + 1024;
    HIP_TRY(hipMalloc((void **)&f->cnt, (size_t)cap * 4));
    HIP_TRY(hipMalloc((void **)&f->bscan, (size_t)(cap + 1) * 8));
    f->cnt_cap = cap;
    int64_t need = gki_scan_tmp_bytes(cap > f->g->d.n_nodes ? cap : f->g->d.n_nodes);
    if (need > f->scan_tmp_bytes) {
        if (f->scan_tmp) HIP_TRY(hipFree(f->scan_tmp));
        HIP_TRY(hipMalloc(&f->scan_tmp, (size_t)need));
        f->scan_tmp_bytes = need;
    }
    return GKI_OK;
}

template <int FMT>
static int emit_impl(gki_finder *f, typename OutSel<FMT>::T out) {
    if (!f->counted) return gki_set_error(GKI_ERR_STATE, "gki_finder_emit_* called before gki_finder_count");
    const DevGraph &d = f->g->d;
    hipStream_t s = f->stream;
    const FindArgs a = f->args;
    HIP_TRY(hipEventRecord(f->ev[2], s));
    if (f->n_interior_records > 0 && f->word_end > f->word_begin) {
        int64_t n_words = f->word_end - f->word_begin;
        int64_t waves = ceil_div(n_words, WPW);
        int64_t blocks = ceil_div(waves, 4);
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_emit_interior<FMT>, dim3((unsigned)blocks), dim3(256), 0, s, d, a, f->ne, out, f->word_begin,
                           f->word_end);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(f->ev[3], s));
    HIP_TRY(hipEventRecord(f->ev[4], s));
    if (f->n_boundary_records > 0) {
        if (a.has_lossy)
            hipLaunchKernelGGL((k_emit_boundary<true, FMT>), dim3(stream_grid(f->B, 256)), dim3(256), 0, s, d, a, f->lossy,
                               f->bnd_start, f->B, f->bscan, f->iscan, out, f->d_err);
        else
            hipLaunchKernelGGL((k_emit_boundary<false, FMT>), dim3(stream_grid(f->B, 256)), dim3(256), 0, s, d, a, f->lossy,
                               f->bnd_start, f->B, f->bscan, f->iscan, out, f->d_err);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(f->ev[5], s));
    f->ev_valid[1] = true; f->ev_valid[2] = true;
    return GKI_OK;
}

extern "C" {

int gki_finder_create(gki_graph *g, gki_finder **out) {
    *out = nullptr;
    if (!g) return gki_set_error(GKI_ERR_BAD_ARG, "finder_create: graph is NULL");
    gki_finder *f = new gki_finder();
    memset(f, 0, sizeof(*f));
    f->g = g;
    const int64_t n = g->d.n_nodes;
    HIP_TRY(hipStreamCreate(&f->stream));
    for (int i = 0; i < 10; i++) HIP_TRY(hipEventCreate(&f->ev[i]));
    HIP_TRY(hipMalloc((void **)&f->bnd_len, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&f->icnt, (size_t)n * 4));
    HIP_TRY(hipMalloc((void **)&f->bnd_start, (size_t)(n + 1) * 8));
    HIP_TRY(hipMalloc((void **)&f->iscan, (size_t)(n + 1) * 8));
    HIP_TRY(hipMalloc((void **)&f->ne, (size_t)n * sizeof(NodeEmit)));
    HIP_TRY(hipMalloc((void **)&f->lossy, (size_t)n * 2));
    HIP_TRY(hipMalloc((void **)&f->d_err, 4));
    HIP_TRY(hipMalloc((void **)&f->d_totals, 3 * 8));
    f->scan_tmp_bytes = gki_scan_tmp_bytes(n);
    HIP_TRY(hipMalloc(&f->scan_tmp, (size_t)f->scan_tmp_bytes));
    *out = f;
    return GKI_OK;
}

int gki_finder_destroy(gki_finder *f) {
    if (!f) return GKI_OK;
    (void)hipStreamSynchronize(f->stream);
    void *ptrs[] = {f->bnd_len, f->icnt, f->bnd_start, f->iscan, f->cnt, f->bscan, f->ne, f->lossy, f->scan_tmp,
                    f->d_err, f->d_totals};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (int i = 0; i < 10; i++) (void)hipEventDestroy(f->ev[i]);
    (void)hipStreamDestroy(f->stream);
    delete f;
    return GKI_OK;
}

int gki_finder_count(gki_finder *f, const gki_find_params *p, int64_t *n_records) {
    *n_records = 0;
    f->counted = false;
    const DevGraph &d = f->g->d;
    if (p->k < 1 || p->k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31 (got %d)", p->k);
    if (p->max_variant_nodes < 0) return gki_set_error(GKI_ERR_BAD_ARG, "max_variant_nodes < 0");
    if (p->node_begin < 0 || p->node_end > d.n_nodes || p->node_begin > p->node_end)
        return gki_set_error(GKI_ERR_BAD_ARG, "bad node range [%lld, %lld]", (long long)p->node_begin, (long long)p->node_end);
    hipStream_t s = f->stream;
    FindArgs a;
    a.k = p->k; a.M = p->max_variant_nodes > 255 ? 255 : p->max_variant_nodes;
    a.one_node = p->one_node_per_kmer ? 1 : 0;
    a.has_lossy = p->h_lossy_crit ? 1 : 0;
    a.node_begin = p->node_begin; a.off_begin = p->off_begin; a.node_end = p->node_end; a.off_end = p->off_end;
    f->args = a;
    if (a.has_lossy) HIP_TRY(hipMemcpyAsync(f->lossy, p->h_lossy_crit, (size_t)d.n_nodes * 2, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(f->d_err, 0, 4, s));

    HIP_TRY(hipEventRecord(f->ev[6], s));
    hipLaunchKernelGGL(k_node_setup, dim3(stream_grid(d.n_nodes, 256)), dim3(256), 0, s, d, a, f->lossy, f->bnd_len, f->icnt);
    HIP_TRY(hipGetLastError());
    GKI_TRY(gki_scan_i32_to_i64(f->bnd_len, d.n_nodes, f->bnd_start, f->scan_tmp, f->scan_tmp_bytes, s));
    GKI_TRY(gki_scan_i32_to_i64(f->icnt, d.n_nodes, f->iscan, f->scan_tmp, f->scan_tmp_bytes, s));
    HIP_TRY(hipEventRecord(f->ev[7], s));
    int64_t B = 0;
    HIP_TRY(hipMemcpyAsync(&B, f->bnd_start + d.n_nodes, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    f->B = B;
    GKI_TRY(ensure_cnt(f, B > 0 ? B : 1));

    HIP_TRY(hipEventRecord(f->ev[0], s));
    if (B > 0) {
        if (a.has_lossy)
            hipLaunchKernelGGL(k_count_boundary<true>, dim3(stream_grid(B, 256)), dim3(256), 0, s, d, a, f->lossy,
                               f->bnd_start, B, f->cnt, f->d_err);
        else
            hipLaunchKernelGGL(k_count_boundary<false>, dim3(stream_grid(B, 256)), dim3(256), 0, s, d, a, f->lossy,
                               f->bnd_start, B, f->cnt, f->d_err);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(f->ev[1], s));
    GKI_TRY(gki_scan_u32_to_i64(f->cnt, B, f->bscan, f->scan_tmp, f->scan_tmp_bytes, s));
    hipLaunchKernelGGL(k_node_emit, dim3(stream_grid(d.n_nodes, 256)), dim3(256), 0, s, d, a, f->bnd_len, f->bnd_start,
                       f->bscan, f->iscan, f->ne);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, s, f->bscan, B, f->iscan, d.n_nodes, f->d_err, f->d_totals);
    HIP_TRY(hipGetLastError());
    int64_t tot[3];
    HIP_TRY(hipMemcpyAsync(tot, f->d_totals, 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (tot[2] != 0)
        return gki_set_error((int)tot[2], "a k-window crosses more than %d nodes (too many empty nodes in a row)", MAXN);
    f->n_boundary_records = tot[0];
    f->n_interior_records = tot[1];
    // words of the sequence covered by the node range
    int64_t p0 = 0, p1 = d.n_bases;
    if (a.node_begin > 0 || a.node_end < d.n_nodes) {
        int64_t v[2];
        int64_t nb = a.node_begin, ne_ = a.node_end < d.n_nodes ? a.node_end + 1 : d.n_nodes;
        HIP_TRY(hipMemcpy(&v[0], d.seq_start + nb, 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&v[1], d.seq_start + ne_, 8, hipMemcpyDeviceToHost));
        p0 = v[0]; p1 = v[1];
    }
    f->word_begin = p0 >> 6;
    f->word_end = ceil_div(p1, 64);
    f->ev_valid[0] = true; f->ev_valid[3] = true;
    f->counted = true;
    *n_records = tot[0] + tot[1];
    return GKI_OK;
}

int gki_finder_emit_flat(gki_finder *f, void *d_hashes, void *d_nodes, void *d_ref_offsets, void *d_af32) {
    OutFlat o{(uint64_t *)d_hashes, (uint32_t *)d_nodes, (uint64_t *)d_ref_offsets, (float *)d_af32};
    return emit_impl<0>(f, o);
}

int gki_finder_emit_v2(gki_finder *f, void *d_hashes, void *d_start_nodes, void *d_start_offsets, void *d_nodes,
                       void *d_af64) {
    OutV2 o{(int64_t *)d_hashes, (int32_t *)d_start_nodes, (int16_t *)d_start_offsets, (int32_t *)d_nodes, (double *)d_af64};
    return emit_impl<1>(f, o);
}

int gki_finder_synchronize(gki_finder *f) { HIP_TRY(hipStreamSynchronize(f->stream)); return GKI_OK; }

int gki_finder_kernel_ms(gki_finder *f, int which, float *ms) {
    *ms = 0.f;
    if (which == 4) { HIP_TRY(hipEventElapsedTime(ms, f->g->ev_prep0, f->g->ev_prep1)); return GKI_OK; }
    if (which < 0 || which > 3) return gki_set_error(GKI_ERR_BAD_ARG, "kernel id %d", which);
    if (!f->ev_valid[which]) return gki_set_error(GKI_ERR_STATE, "kernel %d has not run", which);
    HIP_TRY(hipEventElapsedTime(ms, f->ev[2 * which], f->ev[2 * which + 1]));
    return GKI_OK;
}

int64_t gki_finder_interior_records(const gki_finder *f) { return f->n_interior_records; }

}  // extern "C"
