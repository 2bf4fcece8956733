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
// One decision is NOT made here: whether a nested non-free node (no T predecessor) has an admissible history at all.
// That is an enumeration of histories with a deep stack (host_node_has_history); if the fixed point holds such a node
// the call reports `needs_host` and the caller runs the host pass.  Graphs of SNP / indel bubbles never do.
#include "gki_common.h"

namespace {

constexpr int MAX_SWEEPS = 512;

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
                                                   int k, int M, NodeState *__restrict__ st, unsigned int *__restrict__ changed) {
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
            s.alive = (any_pred && !(nonfree && M < 1)) ? 1 : 0;
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
                     o_out = carve((size_t)n * 2), o_small = carve(256);
        HIP_G(gki_dev_malloc((void **)&arena, off));
        NodeState *st = (NodeState *)(arena + o_state);
        uint8_t *local = (uint8_t *)(arena + o_local), *is_root = (uint8_t *)(arena + o_root);
        uint8_t *follow = h_follow ? (uint8_t *)(arena + o_follow) : nullptr;
        int32_t *roots = (int32_t *)(arena + o_roots);
        uint16_t *out16 = (uint16_t *)(arena + o_out);
        unsigned int *small = (unsigned int *)(arena + o_small);          // [0] changed, [1] general, [2] needs host
        HIP_G(hipMemsetAsync(st, 0, (size_t)n * sizeof(NodeState), s));
        HIP_G(hipMemsetAsync(is_root, 0, (size_t)n, s));
        HIP_G(hipMemsetAsync(small, 0, 16, s));
        if (h_follow) HIP_G(hipMemcpyAsync(follow, h_follow, (size_t)n, hipMemcpyHostToDevice, s));
        if (n_roots > 0) HIP_G(hipMemcpyAsync(roots, h_roots, (size_t)n_roots * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_cls_local, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, (const uint8_t *)follow, local);
        HIP_G(hipGetLastError());
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
                                   (const uint8_t *)is_root, k, max_variant_nodes, st, small);
                HIP_G(hipGetLastError());
            }
            unsigned int changed = 0;
            HIP_G(hipMemcpyAsync(&changed, small, 4, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            converged = changed == 0;
        }
        if (!converged) { *needs_host = 1; goto done; }             // a dependency chain longer than the sweep budget
        hipLaunchKernelGGL(k_cls_final, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, (const uint8_t *)local, (const uint8_t *)is_root,
                           (const NodeState *)st, max_variant_nodes, out16, small + 1);
        HIP_G(hipGetLastError());
        unsigned int verdict[2] = {0, 0};
        HIP_G(hipMemcpyAsync(verdict, small + 1, 8, hipMemcpyDeviceToHost, s));
        HIP_G(hipStreamSynchronize(s));
        *general = verdict[0] ? 1 : 0;
        *needs_host = verdict[1] ? 1 : 0;
        if (!verdict[1] && h_out_flags && (verdict[0] || always_copy_flags))       // the flags are read only by the general kernels
            HIP_G(hipMemcpy(h_out_flags, out16, (size_t)n * 2, hipMemcpyDeviceToHost));
    }
done:
    (void)gki_dev_free(arena);
#undef HIP_G
    return rc;
}
