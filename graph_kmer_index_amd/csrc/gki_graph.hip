// Device graph: obgraph's node-sequence / edge arrays resident in HBM, 2-bit packed, plus the
// node-start bitmap + rank that turns "which node owns base p" into a popcount.
#include "gki_common.h"
#include <vector>
#include <stdlib.h>
#include <limits.h>

namespace {

// 16 bases (uint8 each, values 0..3) -> 32 bits, first base in the least significant 2 bits.
__device__ __forceinline__ uint32_t pack4(uint32_t x) {
    x = (x | (x >> 6)) & 0x000F000Fu;
    return (x | (x >> 12)) & 0xFFu;
}

__global__ __launch_bounds__(256) void k_pack_2bit(const uint8_t *__restrict__ seq, int64_t n_bases,
                                                   uint32_t *__restrict__ seq2_u32, int64_t n_u32) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_u32; i += stride) {
        int64_t b = i * 16;
        uint32_t r;
        if (b + 16 <= n_bases) {
            uint4 v = *reinterpret_cast<const uint4 *>(seq + b);     // 16 B per lane, coalesced
            r = pack4(v.x & 0x03030303u) | (pack4(v.y & 0x03030303u) << 8) |
                (pack4(v.z & 0x03030303u) << 16) | (pack4(v.w & 0x03030303u) << 24);
        } else {
            r = 0;
            for (int j = 0; j < 16; j++)
                if (b + j < n_bases) r |= (uint32_t)(seq[b + j] & 3) << (2 * j);
        }
        seq2_u32[i] = r;
    }
}

__global__ __launch_bounds__(256) void k_start_mask(const int32_t *__restrict__ nonempty, int64_t n_nonempty,
                                                    const int64_t *__restrict__ seq_start,
                                                    unsigned long long *__restrict__ mask) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_nonempty; j += stride) {
        int64_t p = seq_start[nonempty[j]];
        atomicOr(&mask[p >> 6], 1ull << (p & 63));
    }
}

__global__ __launch_bounds__(256) void k_build_walk(DevGraph g, NodeWalk *__restrict__ out) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        NodeWalk w;
        w.seq_start = g.seq_start[n];
        w.size = g.node_size[n];
        const int t31 = w.size < 31 ? w.size : 31;
        w.tail = t31 > 0 ? gki_extract(g.seq2, w.seq_start + w.size - t31, t31) : 0ull;      // the 2-bit pack ran before
        const int64_t r0 = g.rev_start[n], cnt = g.rev_start[n + 1] - r0;
        w.rev_begin = cnt == 1 ? g.rev_edges[r0] : (int32_t)r0;
        w.rev_cnt = (uint16_t)(cnt < 0xFFFF ? cnt : 0xFFFF);
        w.is_ref = g.is_ref[n] ? 1 : 0;
        w.pad = 0;
        w.af = (float)g.allele_freq[n];
        out[n] = w;
    }
}

__global__ __launch_bounds__(256) void k_popcount(const uint64_t *__restrict__ mask, int64_t n, uint32_t *__restrict__ cnt) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) cnt[i] = (uint32_t)__popcll(mask[i]);
}

}  // namespace

// uint8 bases -> 2-bit stream (out sized ceil(n/16) uint32), asynchronous on `s`.
int gki_launch_pack(const uint8_t *d_seq, int64_t n_bases, uint32_t *d_out, hipStream_t s) {
    int64_t n_u32 = ceil_div(n_bases, 16);
    if (n_u32 <= 0) return GKI_OK;
    hipLaunchKernelGGL(k_pack_2bit, dim3(stream_grid(n_u32, 256)), dim3(256), 0, s, d_seq, n_bases, d_out, n_u32);
    HIP_TRY(hipGetLastError());
    return GKI_OK;
}

namespace {
template <typename T>
int upload(gki_graph *g, const T *h, int64_t n, const T **d_out) {
    void *d = nullptr;
    int64_t bytes = (int64_t)sizeof(T) * (n > 0 ? n : 1);
    HIP_TRY(hipMalloc(&d, (size_t)bytes));
    g->owned[g->n_owned++] = d;
    if (n > 0) HIP_TRY(hipMemcpy(d, h, (size_t)(sizeof(T) * n), hipMemcpyHostToDevice));
    *d_out = (const T *)d;
    return GKI_OK;
}

int graph_create_common(gki_graph **out, int64_t n_nodes, const int32_t *h_node_size, const uint8_t *h_seq,
                        const void *d_seq, bool seq_on_device, int64_t n_bases, const int64_t *h_edge_start, const int32_t *h_edges,
                        const int64_t *h_rev_start, const int32_t *h_rev_edges, int64_t n_edges,
                        const uint8_t *h_is_ref, const double *h_allele_freq, const int64_t *h_position_base) {
    *out = nullptr;
    if (n_nodes <= 0 || n_bases < 0 || n_edges < 0) return gki_set_error(GKI_ERR_BAD_ARG, "graph_create: bad sizes");
    if (n_edges >= INT32_MAX) return gki_set_error(GKI_ERR_BAD_ARG, "graph_create: more than 2^31-1 edges");
    gki_graph *g = new gki_graph();
    memset(g, 0, sizeof(*g));
    HIP_TRY(hipGetDevice(&g->device));
    HIP_TRY(hipStreamCreate(&g->stream));
    HIP_TRY(hipEventCreate(&g->ev_prep0));
    HIP_TRY(hipEventCreate(&g->ev_prep1));
    DevGraph &d = g->d;
    d.n_nodes = n_nodes; d.n_bases = n_bases; d.n_words64 = ceil_div(n_bases, 64);

    std::vector<int64_t> seq_start((size_t)n_nodes + 1);
    std::vector<int32_t> nonempty, node_rank((size_t)n_nodes, 0);
    nonempty.reserve((size_t)n_nodes);
    seq_start[0] = 0;
    for (int64_t n = 0; n < n_nodes; n++) {
        if (h_node_size[n] < 0) { delete g; return gki_set_error(GKI_ERR_BAD_ARG, "negative node size"); }
        seq_start[n + 1] = seq_start[n] + h_node_size[n];
        if (h_node_size[n] > 0) { node_rank[(size_t)n] = (int32_t)nonempty.size(); nonempty.push_back((int32_t)n); }
    }
    if (seq_start[n_nodes] != n_bases) { delete g; return gki_set_error(GKI_ERR_BAD_ARG, "sum(node_size) != n_bases"); }
    d.n_nonempty = (int64_t)nonempty.size();

    g->h_seq_start = (int64_t *)malloc((size_t)(n_nodes + 1) * 8);
    if (!g->h_seq_start) { delete g; return gki_set_error(GKI_ERR_HIP, "graph_create: out of host memory"); }
    memcpy(g->h_seq_start, seq_start.data(), (size_t)(n_nodes + 1) * 8);
    GKI_TRY(upload(g, h_node_size, n_nodes, &d.node_size));
    GKI_TRY(upload(g, seq_start.data(), n_nodes + 1, &d.seq_start));
    GKI_TRY(upload(g, h_edge_start, n_nodes + 1, &d.edge_start));
    GKI_TRY(upload(g, h_edges, n_edges, &d.edges));
    GKI_TRY(upload(g, h_rev_start, n_nodes + 1, &d.rev_start));
    GKI_TRY(upload(g, h_rev_edges, n_edges, &d.rev_edges));
    GKI_TRY(upload(g, h_is_ref, n_nodes, &d.is_ref));
    GKI_TRY(upload(g, h_allele_freq, n_nodes, &d.allele_freq));
    GKI_TRY(upload(g, h_position_base ? h_position_base : seq_start.data(), n_nodes, &d.pos_base));
    GKI_TRY(upload(g, nonempty.data(), d.n_nonempty, &d.nonempty));
    GKI_TRY(upload(g, node_rank.data(), n_nodes, &d.node_rank));
    if (seq_on_device) {
        d.seq = (const uint8_t *)d_seq;
        g->owns_seq = false;
    } else {
        GKI_TRY(upload(g, h_seq, n_bases, &d.seq));
        g->owns_seq = true;
    }
    // 2-bit sequence: ceil(n/32) words + 2 words of zero padding (gki_extract reads word w+1)
    int64_t n_u64 = ceil_div(n_bases, 32) + 2;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, (size_t)n_u64 * 8)); g->owned[g->n_owned++] = p; d.seq2 = (const uint64_t *)p;
    HIP_TRY(hipMalloc(&p, (size_t)(d.n_words64 + 1) * 8)); g->owned[g->n_owned++] = p; d.start_mask = (const uint64_t *)p;
    HIP_TRY(hipMalloc(&p, (size_t)(d.n_words64 + 2) * 4)); g->owned[g->n_owned++] = p; d.start_rank = (const uint32_t *)p;
    HIP_TRY(hipMalloc(&p, (size_t)n_nodes * sizeof(NodeWalk))); g->owned[g->n_owned++] = p; d.walk = (const NodeWalk *)p;
    *out = g;
    return gki_graph_prepare(g);
}
}  // namespace

extern "C" {

int gki_graph_prepare(gki_graph *g) {
    DevGraph &d = g->d;
    hipStream_t s = g->stream;
    int64_t n_u64 = ceil_div(d.n_bases, 32) + 2;
    if (g->fwd_nodes) { (void)hipFree(g->fwd_nodes); g->fwd_nodes = nullptr; }     // of the sequence before: the next search builds them
    HIP_TRY(hipEventRecord(g->ev_prep0, s));
    HIP_TRY(hipMemsetAsync((void *)d.seq2, 0, (size_t)n_u64 * 8, s));
    GKI_TRY(gki_launch_pack(d.seq, d.n_bases, (uint32_t *)d.seq2, s));
    hipLaunchKernelGGL(k_build_walk, dim3(stream_grid(d.n_nodes, 256)), dim3(256), 0, s, d, (NodeWalk *)d.walk);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync((void *)d.start_mask, 0, (size_t)(d.n_words64 + 1) * 8, s));
    if (d.n_nonempty > 0) {
        hipLaunchKernelGGL(k_start_mask, dim3(stream_grid(d.n_nonempty, 256)), dim3(256), 0, s, d.nonempty,
                           d.n_nonempty, d.seq_start, (unsigned long long *)d.start_mask);
        HIP_TRY(hipGetLastError());
    }
    if (d.n_words64 > 0) {
        // rank = exclusive scan of per-word popcounts (counts staged in a scratch buffer)
        void *cnt = nullptr, *tmp = nullptr;
        int64_t tmp_bytes = gki_scan_tmp_bytes(d.n_words64);
        HIP_TRY(hipMalloc(&cnt, (size_t)d.n_words64 * 4));
        HIP_TRY(hipMalloc(&tmp, (size_t)tmp_bytes));
        hipLaunchKernelGGL(k_popcount, dim3(stream_grid(d.n_words64, 256)), dim3(256), 0, s, d.start_mask,
                           d.n_words64, (uint32_t *)cnt);
        HIP_TRY(hipGetLastError());
        int r = gki_scan_u32_to_u32((const uint32_t *)cnt, d.n_words64, (uint32_t *)d.start_rank, tmp, tmp_bytes, s);
        HIP_TRY(hipEventRecord(g->ev_prep1, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipFree(cnt));
        HIP_TRY(hipFree(tmp));
        if (r != GKI_OK) return r;
    } else {
        HIP_TRY(hipEventRecord(g->ev_prep1, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return GKI_OK;
}

int gki_graph_create(gki_graph **out, int64_t n_nodes, const int32_t *h_node_size, const uint8_t *h_seq,
                     int64_t n_bases, const int64_t *h_edge_start, const int32_t *h_edges,
                     const int64_t *h_rev_start, const int32_t *h_rev_edges, int64_t n_edges,
                     const uint8_t *h_is_ref, const double *h_allele_freq, const int64_t *h_position_base) {
    return graph_create_common(out, n_nodes, h_node_size, h_seq, nullptr, false, n_bases, h_edge_start, h_edges, h_rev_start,
                               h_rev_edges, n_edges, h_is_ref, h_allele_freq, h_position_base);
}

int gki_graph_create_dseq(gki_graph **out, int64_t n_nodes, const int32_t *h_node_size, const void *d_seq,
                          int64_t n_bases, const int64_t *h_edge_start, const int32_t *h_edges,
                          const int64_t *h_rev_start, const int32_t *h_rev_edges, int64_t n_edges,
                          const uint8_t *h_is_ref, const double *h_allele_freq, const int64_t *h_position_base) {
    if (!d_seq && n_bases > 0) return gki_set_error(GKI_ERR_BAD_ARG, "d_seq is NULL");
    return graph_create_common(out, n_nodes, h_node_size, nullptr, d_seq, true, n_bases,
                               h_edge_start, h_edges, h_rev_start, h_rev_edges, n_edges, h_is_ref, h_allele_freq,
                               h_position_base);
}

int gki_graph_destroy(gki_graph *g) {
    if (!g) return GKI_OK;
    for (int i = 0; i < g->n_owned; i++) (void)hipFree(g->owned[i]);
    if (g->fwd_deep.base) (void)gki_dev_free(g->fwd_deep.base);
    if (g->fwd_script.entries) (void)gki_dev_free(g->fwd_script.entries);
    if (g->fwd_script.ncomp) (void)gki_dev_free(g->fwd_script.ncomp);
    if (g->fwd_script.over_list) (void)gki_dev_free(g->fwd_script.over_list);
    if (g->fwd_nodes) (void)hipFree(g->fwd_nodes);
    (void)hipEventDestroy(g->ev_prep0);
    (void)hipEventDestroy(g->ev_prep1);
    (void)hipStreamDestroy(g->stream);
    free(g->h_seq_start);
    delete g;
    return GKI_OK;
}

int64_t gki_graph_n_bases(const gki_graph *g) { return g->d.n_bases; }

// critical_graph_paths.py:42-104, host walk along the linear reference.
int gki_topological_rank(int64_t n_nodes, const int64_t *edge_start, const int32_t *edges, int32_t *out_rank) {
    std::vector<int32_t> indeg((size_t)n_nodes, 0), queue;
    for (int64_t e = 0; e < edge_start[n_nodes]; e++) indeg[(size_t)edges[e]]++;
    queue.reserve((size_t)n_nodes);
    for (int64_t n = 0; n < n_nodes; n++) if (indeg[(size_t)n] == 0) queue.push_back((int32_t)n);
    size_t head = 0;
    int32_t next = 0;
    while (head < queue.size()) {
        const int32_t n = queue[head++];
        out_rank[n] = next++;
        for (int64_t e = edge_start[n]; e < edge_start[n + 1]; e++)
            if (--indeg[(size_t)edges[e]] == 0) queue.push_back(edges[e]);
    }
    if ((int64_t)next != n_nodes) return gki_set_error(GKI_ERR_BAD_ARG, "the graph has a cycle");
    return GKI_OK;
}

// Node classes for the order-free form of the variant limit on arbitrary DAGs (include/gki.h GKI_NODE_*;
// kmer_finder.py:383-417).  One pass in topological order.
namespace {
// Is the non-free node n entered by the search at all: is there a backward path over alive edges through reachable
// (already classified) nodes on which every step into a non-free node saw fewer than M variant nodes in the k bases
// before it?  Same enumeration as history_ok() of gki_finder.hip for a window that is just the node.
bool host_node_has_history(const int32_t *node_size, const int64_t *rev_start, const int32_t *rev_edges, const uint8_t *flags,
                           int k, int M, int32_t n, int *too_deep) {
    if (M < 1) return false;
    constexpr int HMAXH = 4096;
    std::vector<int32_t> hn, hd, hsz;
    std::vector<int64_t> hcur, hend;
    std::vector<uint8_t> hf;
    hcur.push_back(rev_start[n]); hend.push_back(rev_start[n + 1]);
    size_t h = 0;                                      // nodes of the current history = h (slot h is being filled)
    for (int64_t budget = (int64_t)1 << 24;; budget--) {
        if (budget == 0) { *too_deep = 1; return false; }        // refuse rather than enumerate for minutes
        if (hcur[h] >= hend[h]) {
            if (h == 0) return false;
            hcur.pop_back(); hend.pop_back(); hn.pop_back(); hd.pop_back(); hsz.pop_back(); hf.pop_back();
            h--;
            continue;
        }
        const int32_t p = rev_edges[hcur[h]++];
        const uint8_t fp = flags[p];
        if (fp & GKI_NODE_DEAD) continue;
        const int32_t child = h == 0 ? n : hn[h - 1];
        if ((fp & GKI_NODE_HFS) && !(flags[child] & GKI_NODE_FORCED)) continue;
        const int32_t d = h == 0 ? 0 : hd[h - 1] + hsz[h - 1];
        const int32_t sz = node_size[p] > (1 << 20) ? (1 << 20) : node_size[p];
        // counts with p appended at slot h
        auto nonref = [&](size_t x) { return x == h ? !(fp & GKI_NODE_REF) : !(hf[x] & GKI_NODE_REF); };
        auto dist = [&](size_t x) { return x == h ? d : hd[x]; };
        bool ok = true;
        if (!(fp & GKI_NODE_REF)) {
            int cnt = 0;                               // the node itself: reach k, budget M
            for (size_t x = 0; x <= h; x++) if (nonref(x) && dist(x) < k) cnt++;
            if (cnt >= M) ok = false;
            for (size_t l = 0; l < h && ok; l++) {
                if (hf[l] & (GKI_NODE_REF | GKI_NODE_FORCED)) continue;
                int c2 = 0;
                for (size_t x = l + 1; x <= h; x++) if (nonref(x) && dist(x) - dist(l + 1) < k) c2++;
                if (c2 >= M) ok = false;
            }
        }
        if (!ok) continue;
        if (fp & (GKI_NODE_T | GKI_NODE_SIMPLE)) return true;
        const int32_t end = d + sz;                    // every open constraint closes inside p?
        bool closed = end >= k;
        for (size_t l = 0; l < h && closed; l++)
            if (!(hf[l] & (GKI_NODE_REF | GKI_NODE_FORCED)) && end - dist(l + 1) < k) closed = false;
        if (closed) return true;
        if (!(fp & GKI_NODE_NESTED)) continue;
        if (h + 1 >= (size_t)HMAXH) { *too_deep = 1; return false; }
        hn.push_back(p); hd.push_back(d); hsz.push_back(sz); hf.push_back(fp);
        hcur.push_back(rev_start[p]); hend.push_back(rev_start[p + 1]);
        h++;
    }
}
}  // namespace

int gki_classify_nodes(int64_t n_nodes, const int32_t *node_size, const int64_t *edge_start, const int32_t *edges,
                       const int64_t *rev_start, const int32_t *rev_edges, const uint8_t *is_ref, const uint8_t *follow,
                       const int32_t *roots, int n_roots, int k, int max_variant_nodes, uint16_t *out16, int32_t *general) {
    *general = 0;
    if (n_nodes <= 0) return GKI_OK;
    std::vector<uint8_t> flag_bytes((size_t)n_nodes, 0), bound((size_t)n_nodes, 0);
    uint8_t *out = flag_bytes.data();
    bool ids_topological = true;
    for (int64_t n = 0; n < n_nodes && ids_topological; n++)
        for (int64_t e = edge_start[n]; e < edge_start[n + 1]; e++)
            if (edges[e] <= n) { ids_topological = false; break; }
    std::vector<int32_t> order;
    if (!ids_topological) {
        std::vector<int32_t> rank((size_t)n_nodes);
        GKI_TRY(gki_topological_rank(n_nodes, edge_start, edges, rank.data()));
        order.resize((size_t)n_nodes);
        for (int64_t n = 0; n < n_nodes; n++) order[(size_t)rank[(size_t)n]] = (int32_t)n;
    }
    std::vector<uint8_t> is_root((size_t)n_nodes, 0);
    for (int i = 0; i < n_roots; i++) if (roots[i] >= 0 && roots[i] < n_nodes) is_root[(size_t)roots[i]] = 1;
    const int64_t INF = (int64_t)1 << 40;
    std::vector<int64_t> clean((size_t)n_nodes, 0);   // linear-ref bases of the best history right before the node
    for (int64_t n = 0; n < n_nodes; n++) {
        uint8_t f = is_ref[n] ? GKI_NODE_REF : 0;
        if (follow && follow[n]) f |= GKI_NODE_FORCED;
        const int64_t e0 = edge_start[n], e1 = edge_start[n + 1];
        bool hfs = false;
        int n_ref = 0;
        for (int64_t e = e0; e < e1; e++) {
            if (follow && follow[edges[e]]) hfs = true;
            if (is_ref[edges[e]]) n_ref++;
        }
        if (hfs) f |= GKI_NODE_HFS;
        if (e1 > e0 && !hfs && n_ref != 1) f |= GKI_NODE_CHECK;
        out[n] = f;
    }
    bool gen = false;
    int too_deep = 0;
    for (int64_t i = 0; i < n_nodes; i++) {
        const int64_t n = ids_topological ? i : order[(size_t)i];
        uint8_t f = out[n];
        const int64_t r0 = rev_start[n], r1 = rev_start[n + 1];
        // upper bound on the variant nodes of the k bases before n, over every backward path (saturating)
        {
            int ub = 0;
            for (int64_t r = rev_start[n]; r < rev_start[n + 1]; r++) {
                const int32_t p = rev_edges[r];
                if (out[p] & GKI_NODE_DEAD) continue;
                int u = ((out[p] & GKI_NODE_REF) ? 0 : 1) + (node_size[p] >= k ? 0 : (int)bound[(size_t)p]);
                if (u > ub) ub = u;
            }
            bound[(size_t)n] = (uint8_t)(ub > 255 ? 255 : ub);
        }
        if (is_root[(size_t)n]) {             // a search starts here with no history (chromosome start, critical point)
            clean[(size_t)n] = INF;
            out[n] = f | GKI_NODE_T;
            if (f & (GKI_NODE_CHECK | GKI_NODE_HFS | GKI_NODE_FORCED)) gen = true;
            continue;
        }
        bool any_pred = false, any_t = false;
        int64_t best = 0;
        for (int64_t r = r0; r < r1; r++) {
            const int32_t p = rev_edges[r];
            const uint8_t fp = out[p];
            if (fp & GKI_NODE_DEAD) continue;
            if ((fp & GKI_NODE_HFS) && !(f & GKI_NODE_FORCED)) continue;       // edge removed by a forced sibling
            any_pred = true;
            if (fp & GKI_NODE_T) any_t = true;
            if (fp & GKI_NODE_REF) {
                int64_t c = clean[(size_t)p] + node_size[p];
                if (c > INF) c = INF;
                if (c > best) best = c;
            }
        }
        const bool has_edges = r1 > r0 || edge_start[n + 1] > edge_start[n];
        if (!any_pred) {
            out[n] = f | GKI_NODE_DEAD;
            if (has_edges) gen = true;
            continue;
        }
        if (f & GKI_NODE_REF) {
            clean[(size_t)n] = best;
            if (best >= k) f |= GKI_NODE_T;
        }
        if (!(f & GKI_NODE_T)) f |= any_t ? GKI_NODE_SIMPLE : GKI_NODE_NESTED;
        out[n] = f;
        if (!(f & (GKI_NODE_REF | GKI_NODE_FORCED))) {       // not free to enter: does an admissible history exist?
            if (max_variant_nodes < 1 ||
                (!any_t && !host_node_has_history(node_size, rev_start, rev_edges, out, k, max_variant_nodes, (int32_t)n, &too_deep))) {
                f = (uint8_t)((f & ~(GKI_NODE_SIMPLE | GKI_NODE_NESTED)) | GKI_NODE_DEAD);
                out[n] = f;
                // with limit 0 on a graph of the simple class the fast kernels skip variant nodes by themselves
                if ((f & GKI_NODE_CHECK) || !any_t) gen = true;
                continue;
            }
        }
        if (f & (GKI_NODE_NESTED | GKI_NODE_CHECK | GKI_NODE_HFS | GKI_NODE_FORCED)) gen = true;
    }
    if (too_deep) return gki_set_error(GKI_ERR_WINDOW_TOO_DEEP, "classify: the histories behind one node cross more than 4096 "
                                       "nodes or take more than 2^24 steps to enumerate");
    for (int64_t n = 0; n < n_nodes; n++) out16[n] = (uint16_t)(out[n] | ((uint16_t)bound[(size_t)n] << 8));
    *general = gen ? 1 : 0;
    return GKI_OK;
}

int gki_critical_paths(int64_t n_nodes, const int32_t *node_size, const int64_t *edge_start, const int32_t *edges,
                       const int64_t *rev_start, const uint8_t *is_ref, const int32_t *chrom_start, int n_chrom,
                       int k, uint32_t *out_nodes, uint16_t *out_offsets, int64_t *n_out) {
    int64_t found = 0;
    *n_out = 0;
    for (int c = 0; c < n_chrom; c++) {
        int64_t cur = chrom_start[c];
        int64_t depth = 0, since_join = 0, steps = 0;
        while (true) {
            if (cur < 0 || cur >= n_nodes || ++steps > n_nodes + 1)
                return gki_set_error(GKI_ERR_BAD_ARG, "critical paths: walk left the graph or found a cycle");
            const int64_t indeg = rev_start[cur + 1] - rev_start[cur];
            const bool was_open = depth > 1;
            depth -= indeg;
            if (was_open && depth == 0) since_join = 0;
            const int64_t size = node_size[cur];
            if (depth == 0 && size != 0 && since_join <= k && since_join + size >= k) {
                const int64_t off = (int64_t)k - since_join - 1;
                if (off < 0)
                    return gki_set_error(GKI_ERR_BAD_ARG, "critical paths: node %lld is reached after exactly k bases "
                                         "of single-edge chain; the reference raises here (uint16 offset -1)", (long long)cur);
                out_nodes[found] = (uint32_t)cur;
                out_offsets[found] = (uint16_t)off;
                found++;
            }
            const int64_t e0 = edge_start[cur], e1 = edge_start[cur + 1];
            depth += e1 - e0;
            if (e1 == e0) break;
            if (e1 - e0 == 1) {
                since_join += size;
                cur = edges[e0];
            } else {
                int64_t next = -1, n_ref = 0;
                for (int64_t e = e0; e < e1; e++) if (is_ref[edges[e]]) { next = edges[e]; n_ref++; }
                if (n_ref != 1)
                    return gki_set_error(GKI_ERR_BAD_ARG, "critical paths: node %lld has %lld linear-ref successors "
                                         "(the reference requires exactly one)", (long long)cur, (long long)n_ref);
                cur = next;
            }
        }
    }
    *n_out = found;
    return GKI_OK;
}

}  // extern "C"
