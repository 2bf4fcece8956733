// gki_classify_nodes on the device (the node classes of include/gki.h GKI_NODE_*, for the order-free form of the
// variant limit kmer_finder.py:383-417 on any DAG).
//
// The host version (gki_graph.hip) is one pass in topological order, 0.1 s at 1.5e7 nodes in front of a 15 ms step.
// What it computes is a least fixed point -- "entered" spreads from the search roots along usable edges, the
// linear-ref bases right before a node and the variant-node bound only grow as more predecessors are entered -- so it
// can be reached by relaxation instead: every node recomputes itself from its predecessors until nothing changes.
// Search roots are dense (every critical node is one), so a handful of sweeps suffice on graphs built along a genome;
// the sweep count is bounded and the host pass takes over beyond it.
//
// One decision is not monotone: whether a nested non-free node (no T predecessor) has an admissible history at all -- an
// enumeration of the histories behind it (host_node_has_history of gki_graph.hip; k_cls_history here, one lane per such
// node over the flag words of the fixed point).  A node without one is DEAD, which takes histories away from the nodes
// after it, so the two steps alternate: relax with the nodes found dead so far held dead, classify, enumerate again for
// every candidate -- starting from "every candidate is entered" this only ever removes nodes, and on a DAG it ends at
// what the host pass computes in topological order.  A history deeper than the kernel's stack, more steps than its
// budget, or more rounds than MAX_ROUNDS: `needs_host`, and the caller runs the host pass.  Graphs of SNP / indel
// bubbles have no such node.
#include "gki_common.h"

namespace {

constexpr int MAX_SWEEPS = 512, MAX_ROUNDS = 48, HIST_DEPTH = 64, HIST_BUDGET = 1 << 20;

struct alignas(8) NodeState { uint8_t alive, is_t, bound, any_t; int32_t clean; };     // clean saturates at k; one 8-byte access
static_assert(sizeof(NodeState) == 8, "one 8-byte word per node");

__global__ __launch_bounds__(256) void k_cls_local(DevGraph g, const uint8_t *__restrict__ follow, uint8_t *__restrict__ local) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        uint8_t f = g.is_ref[n] ? GKI_NODE_REF : 0;
        if (follow && follow[n]) f |= GKI_NODE_FORCED;
        const int64_t e0 = g.edge_start[n], e1 = g.edge_start[n + 1];
        bool hfs = false;
        int n_ref = 0;
        for (int64_t e = e0; e < e1; e++) {
            const int32_t c = g.edges[e];
            if (follow && follow[c]) hfs = true;
            if (g.is_ref[c]) n_ref++;
        }
        if (hfs) f |= GKI_NODE_HFS;
        if (e1 > e0 && !hfs && n_ref != 1) f |= GKI_NODE_CHECK;
        local[n] = f;
    }
}

__global__ __launch_bounds__(256) void k_cls_roots(const int32_t *__restrict__ roots, int n_roots, int64_t n_nodes, int k,
                                                   uint8_t *__restrict__ is_root, NodeState *__restrict__ st) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_roots; i += gridDim.x * blockDim.x) {
        const int32_t r = roots[i];
        if (r < 0 || r >= n_nodes) continue;
        is_root[r] = 1;
        NodeState s; s.alive = 1; s.is_t = 1; s.bound = 0; s.any_t = 0; s.clean = k;     // no history: clean = infinity
        st[r] = s;
    }
}

// One sweep.  Reads of predecessors may see this sweep's or the previous sweep's values: every quantity only grows, so
// the fixed point is the same either way.
__global__ __launch_bounds__(256) void k_cls_relax(DevGraph g, const uint8_t *__restrict__ local, const uint8_t *__restrict__ is_root,
                                                   int k, int M, NodeState *__restrict__ st, unsigned int *__restrict__ changed,
                                                   const uint8_t *__restrict__ dead_hist) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool any = false;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        const uint8_t f = local[n];
        const NodeState old = st[n];
        bool any_pred = false, any_t = false;
        int best = 0, ub = 0;
        for (int64_t r = g.rev_start[n]; r < g.rev_start[n + 1]; r++) {
            const int32_t p = g.rev_edges[r];
            const NodeState sp = st[p];
            if (!sp.alive) continue;
            const uint8_t fp = local[p];
            const int size_p = g.node_size[p];
            const int u = ((fp & GKI_NODE_REF) ? 0 : 1) + (size_p >= k ? 0 : (int)sp.bound);
            ub = u > ub ? u : ub;
            if ((fp & GKI_NODE_HFS) && !(f & GKI_NODE_FORCED)) continue;        // edge removed by a forced sibling
            any_pred = true;
            if (sp.is_t) any_t = true;
            if (fp & GKI_NODE_REF) {
                int c = sp.clean + size_p;
                c = c > k ? k : c;
                best = c > best ? c : best;
            }
        }
        NodeState s = old;
        s.bound = (uint8_t)(ub > 255 ? 255 : ub);
        if (!is_root[n]) {
            const bool nonfree = !(f & (GKI_NODE_REF | GKI_NODE_FORCED));
            s.alive = (any_pred && !(nonfree && M < 1) && !dead_hist[n]) ? 1 : 0;
            s.any_t = any_t ? 1 : 0;
            s.clean = (f & GKI_NODE_REF) ? best : 0;
            s.is_t = ((f & GKI_NODE_REF) && any_pred && best >= k) ? 1 : 0;
        }
        if (s.alive != old.alive || s.is_t != old.is_t || s.bound != old.bound || s.any_t != old.any_t || s.clean != old.clean) {
            st[n] = s;
            any = true;
        }
    }
    if (any) *changed = 1u;
}

__global__ __launch_bounds__(256) void k_cls_final(DevGraph g, const uint8_t *__restrict__ local, const uint8_t *__restrict__ is_root,
                                                   const NodeState *__restrict__ st, int M, uint16_t *__restrict__ out16,
                                                   unsigned int *__restrict__ verdict /* [0] general, [1] needs the host pass */) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool gen = false, host = false;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        uint8_t f = local[n];
        const NodeState s = st[n];
        if (is_root[n]) {
            if (f & (GKI_NODE_CHECK | GKI_NODE_HFS | GKI_NODE_FORCED)) gen = true;
            f |= GKI_NODE_T;
        } else {
            // any usable predecessor entered?  (alive is that, except for a non-free node under limit 0)
            const bool nonfree = !(f & (GKI_NODE_REF | GKI_NODE_FORCED));
            bool any_pred = s.alive != 0;
            if (!any_pred && nonfree && M < 1) {
                for (int64_t r = g.rev_start[n]; r < g.rev_start[n + 1] && !any_pred; r++) {
                    const int32_t p = g.rev_edges[r];
                    if (st[p].alive && !((local[p] & GKI_NODE_HFS) && !(f & GKI_NODE_FORCED))) any_pred = true;
                }
            }
            const bool has_edges = g.rev_start[n + 1] > g.rev_start[n] || g.edge_start[n + 1] > g.edge_start[n];
            if (!any_pred) {
                f |= GKI_NODE_DEAD;
                if (has_edges) gen = true;
            } else {
                if (s.is_t) f |= GKI_NODE_T;
                else f |= s.any_t ? GKI_NODE_SIMPLE : GKI_NODE_NESTED;
                if (nonfree && M < 1) {
                    f = (uint8_t)((f & ~(GKI_NODE_SIMPLE | GKI_NODE_NESTED)) | GKI_NODE_DEAD);
                    if ((f & GKI_NODE_CHECK) || !s.any_t) gen = true;
                } else {
                    if (nonfree && !s.any_t) host = true;              // does an admissible history exist?  host enumeration
                    if (f & (GKI_NODE_NESTED | GKI_NODE_CHECK | GKI_NODE_HFS | GKI_NODE_FORCED)) gen = true;
                }
            }
        }
        if (out16) out16[n] = (uint16_t)(f | ((uint16_t)s.bound << 8));
    }
    if (gen) verdict[0] = 1u;
    if (host) verdict[1] = 1u;
}

}  // namespace

// Does the nested non-free node n have an admissible history: a backward path over alive edges on which every step into
// a non-free node saw fewer than M variant nodes in the k bases before it?  host_node_has_history (gki_graph.hip) with
// fixed stacks; flags8 = the flag words of the current fixed point.  Returns 1 yes, 0 no, -1 too deep / out of budget.
__device__ int node_has_history(const DevGraph &g, const uint16_t *__restrict__ flags16, int k, int M, int32_t n) {
    int32_t hn[HIST_DEPTH], hd[HIST_DEPTH], hsz[HIST_DEPTH], hcur[HIST_DEPTH + 1], hend[HIST_DEPTH + 1];
    uint8_t hf[HIST_DEPTH];
    hcur[0] = (int32_t)g.rev_start[n]; hend[0] = (int32_t)g.rev_start[n + 1];
    int h = 0;                                            // nodes of the current history (slot h is being filled)
    for (int budget = HIST_BUDGET;; budget--) {
        if (budget == 0) return -1;
        if (hcur[h] >= hend[h]) {
            if (h == 0) return 0;
            h--;
            continue;
        }
        const int32_t p = g.rev_edges[hcur[h]++];
        const uint8_t fp = (uint8_t)flags16[p];
        if (fp & GKI_NODE_DEAD) continue;
        const int32_t child = h == 0 ? n : hn[h - 1];
        if ((fp & GKI_NODE_HFS) && !((uint8_t)flags16[child] & GKI_NODE_FORCED)) continue;
        const int32_t d = h == 0 ? 0 : hd[h - 1] + hsz[h - 1];
        const int32_t raw = g.node_size[p], sz = raw > (1 << 20) ? (1 << 20) : raw;
        bool ok = true;
        if (!(fp & GKI_NODE_REF)) {                       // counts with p appended at slot h
            int cnt = d < k ? 1 : 0;                      // the node itself: reach k, budget M
            for (int x = 0; x < h; x++) if (!(hf[x] & GKI_NODE_REF) && hd[x] < k) cnt++;
            if (cnt >= M) ok = false;
            for (int l = 0; l < h && ok; l++) {
                if (hf[l] & (GKI_NODE_REF | GKI_NODE_FORCED)) continue;
                const int32_t from = l + 1 == h ? d : hd[l + 1];
                int c2 = d - from < k ? 1 : 0;
                for (int x = l + 1; x < h; x++) if (!(hf[x] & GKI_NODE_REF) && hd[x] - from < k) c2++;
                if (c2 >= M) ok = false;
            }
        }
        if (!ok) continue;
        if (fp & (GKI_NODE_T | GKI_NODE_SIMPLE)) return 1;
        const int32_t end = d + sz;                       // every open constraint closes inside p?
        bool closed = end >= k;
        for (int l = 0; l < h && closed; l++)
            if (!(hf[l] & (GKI_NODE_REF | GKI_NODE_FORCED)) && end - (l + 1 == h ? d : hd[l + 1]) < k) closed = false;
        if (closed) return 1;
        if (!(fp & GKI_NODE_NESTED)) continue;
        if (h + 1 >= HIST_DEPTH) return -1;
        hn[h] = p; hd[h] = d; hsz[h] = sz; hf[h] = fp;
        h++;
        hcur[h] = (int32_t)g.rev_start[p]; hend[h] = (int32_t)g.rev_start[p + 1];
    }
}

// One lane per candidate: a non-root, non-free node that the fixed point holds alive with no T predecessor.
__global__ __launch_bounds__(64) void k_cls_history(DevGraph g, const uint8_t *__restrict__ local, const uint8_t *__restrict__ is_root,
                                                    const NodeState *__restrict__ st, const uint16_t *__restrict__ flags16, int k, int M,
                                                    uint8_t *__restrict__ dead_hist, unsigned int *__restrict__ small /* [0] changed, [2] gave up */) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        const uint8_t f = local[n];
        const NodeState s = st[n];
        if (is_root[n] || (f & (GKI_NODE_REF | GKI_NODE_FORCED)) || !s.alive || s.any_t || M < 1) continue;
        const int r = node_has_history(g, flags16, k, M, (int32_t)n);
        if (r < 0) small[2] = 1u;
        else if (r == 0) { dead_hist[n] = 1; small[0] = 1u; }
    }
}

extern "C" int gki_graph_classify_nodes(gki_graph *gr, const uint8_t *h_follow, const int32_t *h_roots, int n_roots, int k,
                                        int max_variant_nodes, uint16_t *h_out_flags, int always_copy_flags, int32_t *general,
                                        int32_t *needs_host) {
    *general = 0; *needs_host = 0;
    const DevGraph &g = gr->d;
    const int64_t n = g.n_nodes;
    if (n <= 0) return GKI_OK;
    HIP_TRY(hipSetDevice(gr->device));
    hipStream_t s = gr->stream;
    char *arena = nullptr;
    int rc = GKI_OK;
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        size_t off = 0;
        auto carve = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
        const size_t o_state = carve((size_t)n * sizeof(NodeState)), o_local = carve((size_t)n), o_root = carve((size_t)n),
                     o_follow = carve(h_follow ? (size_t)n : 0), o_roots = carve((size_t)(n_roots > 0 ? n_roots : 1) * 4),
                     o_out = carve((size_t)n * 2), o_small = carve(256), o_dead = carve((size_t)n);
        HIP_G(gki_dev_malloc((void **)&arena, off));
        NodeState *st = (NodeState *)(arena + o_state);
        uint8_t *local = (uint8_t *)(arena + o_local), *is_root = (uint8_t *)(arena + o_root);
        uint8_t *follow = h_follow ? (uint8_t *)(arena + o_follow) : nullptr;
        int32_t *roots = (int32_t *)(arena + o_roots);
        uint16_t *out16 = (uint16_t *)(arena + o_out);
        unsigned int *small = (unsigned int *)(arena + o_small);          // [0] changed, [1] general, [2] candidates / gave up
        uint8_t *dead_hist = (uint8_t *)(arena + o_dead);                // nodes found without an admissible history so far
        HIP_G(hipMemsetAsync(is_root, 0, (size_t)n, s));
        HIP_G(hipMemsetAsync(dead_hist, 0, (size_t)n, s));
        HIP_G(hipMemsetAsync(small, 0, 16, s));
        if (h_follow) HIP_G(hipMemcpyAsync(follow, h_follow, (size_t)n, hipMemcpyHostToDevice, s));
        if (n_roots > 0) HIP_G(hipMemcpyAsync(roots, h_roots, (size_t)n_roots * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_cls_local, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, (const uint8_t *)follow, local);
        HIP_G(hipGetLastError());
        for (int round = 0;; round++) {
            if (round >= MAX_ROUNDS) { *needs_host = 1; goto done; }
            // ---- least fixed point with the nodes found dead so far held dead
            HIP_G(hipMemsetAsync(st, 0, (size_t)n * sizeof(NodeState), s));
            if (n_roots > 0) {
                hipLaunchKernelGGL(k_cls_roots, dim3(stream_grid(n_roots, 256)), dim3(256), 0, s, (const int32_t *)roots, n_roots, n, k, is_root, st);
                HIP_G(hipGetLastError());
            }
            bool converged = false;
            for (int sweep = 0; sweep < MAX_SWEEPS && !converged; sweep += 4) {
                HIP_G(hipMemsetAsync(small, 0, 4, s));
                for (int i = 0; i < 4; i++) {                           // four sweeps per look at the flag
                    if (i == 3) HIP_G(hipMemsetAsync(small, 0, 4, s));  // the last sweep of a batch alone decides
                    hipLaunchKernelGGL(k_cls_relax, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, (const uint8_t *)local,
                                       (const uint8_t *)is_root, k, max_variant_nodes, st, small, (const uint8_t *)dead_hist);
                    HIP_G(hipGetLastError());
                }
                unsigned int changed = 0;
                HIP_G(hipMemcpyAsync(&changed, small, 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipStreamSynchronize(s));
                converged = changed == 0;
            }
            if (!converged) { *needs_host = 1; goto done; }             // a dependency chain longer than the sweep budget
            HIP_G(hipMemsetAsync(small, 0, 16, s));
            hipLaunchKernelGGL(k_cls_final, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, (const uint8_t *)local, (const uint8_t *)is_root,
                               (const NodeState *)st, max_variant_nodes, out16, small + 1);
            HIP_G(hipGetLastError());
            unsigned int verdict[2] = {0, 0};
            HIP_G(hipMemcpyAsync(verdict, small + 1, 8, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            *general = verdict[0] ? 1 : 0;
            if (!verdict[1]) break;                                     // no nested non-free node: the fixed point is the answer
            // ---- which of the candidates have no admissible history?  (every candidate, every round: a node found dead takes
            // histories away from the nodes after it)
            HIP_G(hipMemsetAsync(small, 0, 16, s));
            hipLaunchKernelGGL(k_cls_history, dim3(stream_grid(n, 64)), dim3(64), 0, s, g, (const uint8_t *)local, (const uint8_t *)is_root,
                               (const NodeState *)st, (const uint16_t *)out16, k, max_variant_nodes, dead_hist, small);
            HIP_G(hipGetLastError());
            unsigned int res[3] = {0, 0, 0};
            HIP_G(hipMemcpyAsync(res, small, 12, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            if (res[2]) { *needs_host = 1; goto done; }                 // a history deeper than the kernel's stack or budget
            if (!res[0]) break;                                         // every candidate is entered: the flags stand
        }
        if (h_out_flags && (*general || always_copy_flags))             // the flags are read only by the general kernels
            HIP_G(hipMemcpy(h_out_flags, out16, (size_t)n * 2, hipMemcpyDeviceToHost));
    }
done:
    (void)gki_dev_free(arena);
#undef HIP_G
    return rc;
}
