// CriticalGraphPaths.from_graph (critical_graph_paths.py:42-104) on the device.
//
// The reference walks the linear reference of every chromosome node by node, carrying two numbers: `depth` (open
// branches: minus the in-degree on entry, plus the out-degree on exit) and `bp_since_last_join`.  On 1.5e7 nodes that
// walk is 36 ms of sequential host work in front of a 15 ms GPU step.  Here it is a handful of data-parallel passes:
//
//   next[v]      the node the walk steps to from v -- a function of v alone: the single successor, or the one
//                linear-ref successor of a branching node (:84-100); a node without successors ends the walk
//   jump tables  J_i[v] = next^(2^i)(v) by pointer doubling (ceil(log2 n) + 1 rounds), with the step counts, so the
//                length of every chromosome's walk is known and a cycle shows as a walk that never ends
//   the path     path[j + 2^i] = J_i[path[j]], top bit first: every position is written exactly once
//   the walk's state as scans over the path: depth on entry = exclusive sum of (out-degree - in-degree);
//                bp_since_last_join = sum of the sizes of single-exit nodes since the last reset (:68-70 resets it when
//                the depth returns to 0 from above 1) = a prefix sum minus its value at the segment's head
//   the test     depth 0, non-empty, b <= k, b + size >= k -> critical at offset k - b - 1 (:76-82), compacted in order
//
// On a graph built along a genome the path needs no jump tables: it is the linear-ref(-dummy) nodes in id order, which
// is guessed by a compaction and verified against next[] in parallel; only a graph where that guess fails pays for the
// pointer doubling.  Errors are the reference's: a branching node of the walk without exactly one linear-ref successor
// (:96-100, raised inside the walk) and an offset of -1 (b == k, surfacing at :104 after all walks).
#include "gki_common.h"

namespace {

constexpr uint32_t ERR_NONE = 0xFFFFFFFFu;

__global__ __launch_bounds__(256) void k_walk_next(DevGraph g, int32_t *__restrict__ jump, uint32_t *__restrict__ cnt,
                                                   uint8_t *__restrict__ bad) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < g.n_nodes; v += stride) {
        const int64_t e0 = g.edge_start[v], e1 = g.edge_start[v + 1];
        int32_t nx = (int32_t)v;
        uint8_t b = 0;
        if (e1 - e0 == 1) nx = g.edges[e0];
        else if (e1 - e0 > 1) {
            int n_ref = 0;
            for (int64_t e = e0; e < e1; e++) if (g.is_ref[g.edges[e]]) { nx = g.edges[e]; n_ref++; }
            if (n_ref != 1) { nx = (int32_t)v; b = 1; }             // the reference raises when the walk stands here
        }
        jump[v] = nx;
        cnt[v] = nx != (int32_t)v ? 1u : 0u;
        bad[v] = b;
    }
}

__global__ __launch_bounds__(256) void k_walk_double(const int32_t *__restrict__ j_in, const uint32_t *__restrict__ c_in, int64_t n,
                                                     int32_t *__restrict__ j_out, uint32_t *__restrict__ c_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) {
        const int32_t m = j_in[v];
        j_out[v] = j_in[m];
        const uint64_t c = (uint64_t)c_in[v] + c_in[m];
        c_out[v] = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;     // saturates on a cycle
    }
}

struct Chroms { int n; int64_t begin[65]; int32_t start[64]; };          // path slice of every chromosome

__device__ __forceinline__ int chrom_of(const Chroms &c, int64_t pos) {
    int lo = 0;
    for (int i = 1; i < c.n; i++) if (c.begin[i] <= pos) lo = i;
    return lo;
}

// one level of the path: positions j = 0 (mod 2^(level+1)) hand their node's 2^level-th successor to j + 2^level
__global__ __launch_bounds__(256) void k_walk_fill(Chroms c, const int32_t *__restrict__ jump_level, int level, int64_t total,
                                                   int32_t *__restrict__ path) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, half = (int64_t)1 << level;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t * 2 * half < total + 2 * half * c.n; t += stride) {
        // t enumerates (chromosome, multiple of 2^(level+1)) pairs chromosome by chromosome
        int64_t rest = t;
        for (int q = 0; q < c.n; q++) {
            const int64_t len = c.begin[q + 1] - c.begin[q];
            const int64_t slots = (len + 2 * half - 1) / (2 * half);
            if (rest < slots) {
                const int64_t j = rest * 2 * half;
                if (j + half < len) path[c.begin[q] + j + half] = jump_level[path[c.begin[q] + j]];
                break;
            }
            rest -= slots;
        }
    }
}

__global__ __launch_bounds__(256) void k_walk_deltas(DevGraph g, const int32_t *__restrict__ path, int64_t total,
                                                     int32_t *__restrict__ delta, uint32_t *__restrict__ weight) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int32_t v = path[p];
        const int32_t in = (int32_t)(g.rev_start[v + 1] - g.rev_start[v]), out = (int32_t)(g.edge_start[v + 1] - g.edge_start[v]);
        delta[p] = out - in;
        weight[p] = out == 1 ? (uint32_t)g.node_size[v] : 0u;       // :90-92 only a single edge adds to bp_since_last_join
    }
}

// segment heads: the first node of a chromosome and every node at which the depth returns to 0 from above 1
__global__ __launch_bounds__(256) void k_walk_heads(DevGraph g, Chroms c, const int32_t *__restrict__ path, int64_t total,
                                                    const int64_t *__restrict__ depth_sum, uint32_t *__restrict__ head) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int q = chrom_of(c, p);
        const int32_t v = path[p];
        const int64_t enter = depth_sum[p] - depth_sum[c.begin[q]];              // depth before `depth -= in-degree`
        const int64_t in = g.rev_start[v + 1] - g.rev_start[v];
        head[p] = (p == c.begin[q] || (enter > 1 && enter - in == 0)) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void k_walk_head_values(const uint32_t *__restrict__ head, const uint32_t *__restrict__ seg,
                                                          const int64_t *__restrict__ bp_sum, int64_t total,
                                                          int64_t *__restrict__ head_bp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride)
        if (head[p]) head_bp[seg[p]] = bp_sum[p];              // seg = exclusive count of heads: this head's own number
}

__global__ __launch_bounds__(256) void k_walk_test(DevGraph g, Chroms c, const int32_t *__restrict__ path, int64_t total,
                                                   const int64_t *__restrict__ depth_sum, const int64_t *__restrict__ bp_sum,
                                                   const uint32_t *__restrict__ head, const uint32_t *__restrict__ seg,
                                                   const int64_t *__restrict__ head_bp, const uint8_t *__restrict__ bad, int k,
                                                   uint32_t *__restrict__ crit, uint16_t *__restrict__ off_out,
                                                   unsigned int *__restrict__ first_err /* [0] not-one-ref position, [1] offset -1 position */) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int q = chrom_of(c, p);
        const int32_t v = path[p];
        const int64_t in = g.rev_start[v + 1] - g.rev_start[v];
        const int64_t depth = depth_sum[p] - depth_sum[c.begin[q]] - in;          // after entering the node (:67)
        const int64_t size = g.node_size[v];
        const uint32_t s = seg[p] + (head[p] ? 1u : 0u) - 1u;                     // number of the head at or before p
        const int64_t b = bp_sum[p] - head_bp[s];                                 // bp_since_last_join at this node
        uint32_t is_crit = 0;
        if (depth == 0 && size != 0 && b <= k && b + size >= k) {                 // :76-82
            const int64_t off = (int64_t)k - b - 1;
            if (off < 0) atomicMin(&first_err[1], (unsigned int)p);               // :104 uint16(-1)
            else { is_crit = 1; off_out[p] = (uint16_t)off; }
        }
        crit[p] = is_crit;
        if (bad[v]) atomicMin(&first_err[0], (unsigned int)p);                    // :96-100
    }
}

__global__ __launch_bounds__(256) void k_walk_emit(const int32_t *__restrict__ path, const uint32_t *__restrict__ crit,
                                                   const uint32_t *__restrict__ pos, const uint16_t *__restrict__ off, int64_t total,
                                                   uint32_t *__restrict__ out_nodes, uint16_t *__restrict__ out_offsets) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride)
        if (crit[p]) { out_nodes[pos[p]] = (uint32_t)path[p]; out_offsets[pos[p]] = off[p]; }
}

// ---- the common case without jump tables: on a graph built along a genome the walk is exactly the linear-ref(-dummy)
// nodes in id order.  Guess that list (a compaction), check it against next[] in parallel -- every listed node's next
// must be the following one, the last one must end the walk -- and use it; any mismatch falls back to pointer doubling.
__global__ __launch_bounds__(256) void k_walk_ref_flags(DevGraph g, uint32_t *__restrict__ flag) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < g.n_nodes; v += stride) flag[v] = g.is_ref[v] ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_walk_ref_list(const uint32_t *__restrict__ flag, const uint32_t *__restrict__ at, int64_t n,
                                                       int32_t *__restrict__ list) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += stride) if (flag[v]) list[at[v]] = (int32_t)v;
}
struct RefSlices { int n; int64_t from[64], to[64], begin[65]; };      // chromosome q = list[from[q], to[q]) -> path[begin[q] ..)
__global__ __launch_bounds__(256) void k_walk_ref_path(RefSlices r, const int32_t *__restrict__ list, const int32_t *__restrict__ next,
                                                       int64_t total, int32_t *__restrict__ path, unsigned int *__restrict__ mismatch) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        int q = 0;
        for (int i = 1; i < r.n; i++) if (r.begin[i] <= p) q = i;
        const int64_t j = r.from[q] + (p - r.begin[q]);
        const int32_t v = list[j];
        path[p] = v;
        const int32_t want = j + 1 < r.to[q] ? list[j + 1] : v;              // the slice's last node must end the walk
        if (next[v] != want) *mismatch = 1u;
    }
}

}  // namespace

extern "C" int gki_graph_critical_paths(gki_graph *gr, const int32_t *h_chrom_start, int n_chrom, int k, uint32_t *h_out_nodes,
                                        uint16_t *h_out_offsets, int64_t *n_out) {
    *n_out = 0;
    const DevGraph &g = gr->d;
    const int64_t n = g.n_nodes;
    if (n_chrom < 1 || n_chrom > 64) return gki_set_error(GKI_ERR_BAD_ARG, "critical paths: 1..64 chromosomes on the device path");
    for (int c = 0; c < n_chrom; c++)
        if (h_chrom_start[c] < 0 || h_chrom_start[c] >= n)
            return gki_set_error(GKI_ERR_BAD_ARG, "critical paths: walk left the graph or found a cycle");
    HIP_TRY(hipSetDevice(gr->device));
    hipStream_t s = gr->stream;
    int rounds = 1;
    while (((int64_t)1 << rounds) < n + 1) rounds++;
    rounds++;                                               // 2^rounds > n + 1 steps: only a cycle is still walking then
    int32_t **jump = (int32_t **)calloc((size_t)rounds + 1, sizeof(int32_t *));
    uint32_t *cnt[2] = {nullptr, nullptr};
    char *arena = nullptr, *arena2 = nullptr;
    uint32_t *d_out_nodes = nullptr;
    uint16_t *d_out_off = nullptr;
    int rc = GKI_OK;
    if (!jump) return gki_set_error(GKI_ERR_BAD_ARG, "out of host memory");
#define HIP_G(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = gki_set_error(GKI_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); goto done; } } while (0)
    {
        // one allocation for everything sized by the node count (the allocator, not the kernels, was the cost of a cold call)
        size_t off = 0;
        auto carve = [&off](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
        const size_t o_next = carve((size_t)n * 4), o_bad = carve((size_t)n), o_flag = carve((size_t)n * 4),
                     o_at = carve((size_t)(n + 1) * 4), o_list = carve((size_t)n * 4), o_tmp = carve((size_t)gki_scan_tmp_bytes(n + 1)),
                     o_small = carve(256);
        HIP_G(gki_dev_malloc((void **)&arena, off));
        int32_t *next = (int32_t *)(arena + o_next), *list = (int32_t *)(arena + o_list);
        uint8_t *bad = (uint8_t *)(arena + o_bad);
        uint32_t *flag = (uint32_t *)(arena + o_flag), *at = (uint32_t *)(arena + o_at);
        void *tmp = arena + o_tmp;
        const int64_t tmp_bytes = gki_scan_tmp_bytes(n + 1);
        unsigned int *small = (unsigned int *)(arena + o_small);     // [0] not-one-ref position, [1] offset -1 position, [2] mismatch
        HIP_G(hipMemsetAsync(small, 0xFF, 8, s));
        HIP_G(hipMemsetAsync(small + 2, 0, 4, s));
        HIP_G(gki_dev_malloc((void **)&cnt[0], (size_t)n * 4));
        hipLaunchKernelGGL(k_walk_next, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, next, cnt[0], bad);
        HIP_G(hipGetLastError());
        // ---- the guess: linear-ref(-dummy) nodes in id order
        hipLaunchKernelGGL(k_walk_ref_flags, dim3(stream_grid(n, 256)), dim3(256), 0, s, g, flag);
        HIP_G(hipGetLastError());
        rc = gki_scan_u32_to_u32(flag, n, at, tmp, tmp_bytes, s);
        if (rc != GKI_OK) goto done;
        hipLaunchKernelGGL(k_walk_ref_list, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const uint32_t *)flag, (const uint32_t *)at, n, list);
        HIP_G(hipGetLastError());
        Chroms c;
        c.n = n_chrom;
        c.begin[0] = 0;
        bool guessed = true;
        {
            // slice of every chromosome in the list: from its start node to the next chromosome start (by id) or the end
            uint32_t h_at[64], h_flag[64], n_ref = 0;
            for (int q = 0; q < n_chrom; q++) {
                HIP_G(hipMemcpyAsync(&h_at[q], at + h_chrom_start[q], 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipMemcpyAsync(&h_flag[q], flag + h_chrom_start[q], 4, hipMemcpyDeviceToHost, s));
            }
            HIP_G(hipMemcpyAsync(&n_ref, at + n, 4, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            RefSlices r;
            r.n = n_chrom;
            r.begin[0] = 0;
            for (int q = 0; q < n_chrom && guessed; q++) {
                if (!h_flag[q]) { guessed = false; break; }
                int64_t to = n_ref;
                for (int o = 0; o < n_chrom; o++)
                    if (h_chrom_start[o] > h_chrom_start[q] && (int64_t)h_at[o] < to) to = h_at[o];
                r.from[q] = h_at[q]; r.to[q] = to;
                r.begin[q + 1] = r.begin[q] + (to - (int64_t)h_at[q]);
                c.start[q] = h_chrom_start[q];
                c.begin[q + 1] = r.begin[q + 1];
            }
            if (guessed) {
                const int64_t total = r.begin[n_chrom];
                HIP_G(gki_dev_malloc((void **)&arena2, (size_t)total * 4 + 256));
                hipLaunchKernelGGL(k_walk_ref_path, dim3(stream_grid(total, 256)), dim3(256), 0, s, r, (const int32_t *)list,
                                   (const int32_t *)next, total, (int32_t *)arena2, small + 2);
                HIP_G(hipGetLastError());
                unsigned int mismatch = 0;
                HIP_G(hipMemcpyAsync(&mismatch, small + 2, 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipStreamSynchronize(s));
                if (mismatch) { guessed = false; (void)gki_dev_free(arena2); arena2 = nullptr; }
            }
        }
        int32_t *path = (int32_t *)arena2;
        if (!guessed) {
            // ---- any DAG: jump tables by pointer doubling, then the path level by level
            jump[0] = nullptr;
            for (int i = 1; i <= rounds; i++) HIP_G(gki_dev_malloc((void **)&jump[i], (size_t)n * 4));
            HIP_G(gki_dev_malloc((void **)&cnt[1], (size_t)n * 4));
            for (int i = 0; i < rounds; i++) {
                hipLaunchKernelGGL(k_walk_double, dim3(stream_grid(n, 256)), dim3(256), 0, s, (const int32_t *)(i ? jump[i] : next),
                                   (const uint32_t *)cnt[i & 1], n, jump[i + 1], cnt[(i + 1) & 1]);
                HIP_G(hipGetLastError());
            }
            for (int q = 0; q < n_chrom; q++) {
                uint32_t steps = 0;
                int32_t last = 0, after = 0;
                HIP_G(hipMemcpyAsync(&steps, cnt[rounds & 1] + h_chrom_start[q], 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipMemcpyAsync(&last, jump[rounds] + h_chrom_start[q], 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipStreamSynchronize(s));
                HIP_G(hipMemcpyAsync(&after, next + last, 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipStreamSynchronize(s));
                if (after != last || (int64_t)steps > n) { rc = gki_set_error(GKI_ERR_BAD_ARG, "critical paths: walk left the graph or found a cycle"); goto done; }
                c.start[q] = h_chrom_start[q];
                c.begin[q + 1] = c.begin[q] + (int64_t)steps + 1;
            }
            const int64_t total = c.begin[n_chrom];
            HIP_G(gki_dev_malloc((void **)&arena2, (size_t)total * 4 + 256));
            path = (int32_t *)arena2;
            for (int q = 0; q < n_chrom; q++)
                HIP_G(hipMemcpyAsync(path + c.begin[q], &c.start[q], 4, hipMemcpyHostToDevice, s));
            HIP_G(hipStreamSynchronize(s));                      // c.start lives on this stack frame
            int top = 0;
            int64_t longest = 1;
            for (int q = 0; q < n_chrom; q++) if (c.begin[q + 1] - c.begin[q] > longest) longest = c.begin[q + 1] - c.begin[q];
            while (((int64_t)1 << (top + 1)) < longest) top++;
            for (int level = top; level >= 0; level--) {
                const int64_t work = total / ((int64_t)2 << level) + n_chrom;
                hipLaunchKernelGGL(k_walk_fill, dim3(stream_grid(work, 256)), dim3(256), 0, s, c, (const int32_t *)(level ? jump[level] : next),
                                   level, total, path);
                HIP_G(hipGetLastError());
            }
        }
        // ---- the walk's state over the path, as scans
        const int64_t total = c.begin[n_chrom];
        {
            size_t off2 = 0;
            auto carve2 = [&off2](size_t bytes) { const size_t at2 = off2; off2 += (bytes + 255) / 256 * 256; return at2; };
            const size_t p_delta = carve2((size_t)total * 4), p_weight = carve2((size_t)total * 4), p_head = carve2((size_t)total * 4),
                         p_seg = carve2((size_t)(total + 1) * 4), p_crit = carve2((size_t)total * 4), p_pos = carve2((size_t)(total + 1) * 4),
                         p_off = carve2((size_t)total * 2), p_depth = carve2((size_t)(total + 1) * 8), p_bp = carve2((size_t)(total + 1) * 8),
                         p_hbp = carve2((size_t)(total + 1) * 8), p_tmp = carve2((size_t)gki_scan_tmp_bytes(total + 1)), p_err = carve2(256);
            char *work = nullptr;
            HIP_G(gki_dev_malloc((void **)&work, off2));
            (void)gki_dev_free(cnt[0]); cnt[0] = (uint32_t *)work;          // freed with the rest below
            int32_t *delta = (int32_t *)(work + p_delta);
            uint32_t *weight = (uint32_t *)(work + p_weight), *head = (uint32_t *)(work + p_head), *seg = (uint32_t *)(work + p_seg),
                     *crit = (uint32_t *)(work + p_crit), *pos = (uint32_t *)(work + p_pos);
            uint16_t *offp = (uint16_t *)(work + p_off);
            int64_t *depth_sum = (int64_t *)(work + p_depth), *bp_sum = (int64_t *)(work + p_bp), *head_bp = (int64_t *)(work + p_hbp);
            void *tmp2 = work + p_tmp;
            const int64_t tmp2_bytes = gki_scan_tmp_bytes(total + 1);
            unsigned int *first_err = (unsigned int *)(work + p_err);
            HIP_G(hipMemsetAsync(first_err, 0xFF, 16, s));
            hipLaunchKernelGGL(k_walk_deltas, dim3(stream_grid(total, 256)), dim3(256), 0, s, g, (const int32_t *)path, total, delta, weight);
            HIP_G(hipGetLastError());
            rc = gki_scan_i32_to_i64(delta, total, depth_sum, tmp2, tmp2_bytes, s);
            if (rc == GKI_OK) rc = gki_scan_u32_to_i64(weight, total, bp_sum, tmp2, tmp2_bytes, s);
            if (rc != GKI_OK) goto done;
            hipLaunchKernelGGL(k_walk_heads, dim3(stream_grid(total, 256)), dim3(256), 0, s, g, c, (const int32_t *)path, total,
                               (const int64_t *)depth_sum, head);
            HIP_G(hipGetLastError());
            rc = gki_scan_u32_to_u32(head, total, seg, tmp2, tmp2_bytes, s);
            if (rc != GKI_OK) goto done;
            hipLaunchKernelGGL(k_walk_head_values, dim3(stream_grid(total, 256)), dim3(256), 0, s, (const uint32_t *)head,
                               (const uint32_t *)seg, (const int64_t *)bp_sum, total, head_bp);
            HIP_G(hipGetLastError());
            hipLaunchKernelGGL(k_walk_test, dim3(stream_grid(total, 256)), dim3(256), 0, s, g, c, (const int32_t *)path, total,
                               (const int64_t *)depth_sum, (const int64_t *)bp_sum, (const uint32_t *)head, (const uint32_t *)seg,
                               (const int64_t *)head_bp, (const uint8_t *)bad, k, crit, offp, first_err);
            HIP_G(hipGetLastError());
            rc = gki_scan_u32_to_u32(crit, total, pos, tmp2, tmp2_bytes, s);
            if (rc != GKI_OK) goto done;
            unsigned int h_err[2] = {ERR_NONE, ERR_NONE};
            uint32_t found = 0;
            HIP_G(hipMemcpyAsync(h_err, first_err, 8, hipMemcpyDeviceToHost, s));
            HIP_G(hipMemcpyAsync(&found, pos + total, 4, hipMemcpyDeviceToHost, s));
            HIP_G(hipStreamSynchronize(s));
            if (h_err[0] != ERR_NONE || h_err[1] != ERR_NONE) {
                // the reference raises inside the walk for a branching node without exactly one linear-ref successor (:96-100);
                // the offset -1 only surfaces when the offsets become uint16 after all walks (:104) -- so the former wins
                int32_t node = 0;
                const bool off_err = h_err[0] == ERR_NONE;
                HIP_G(hipMemcpy(&node, path + (off_err ? h_err[1] : h_err[0]), 4, hipMemcpyDeviceToHost));
                rc = off_err ? gki_set_error(GKI_ERR_BAD_ARG, "critical paths: node %d is reached after exactly k bases of single-edge "
                                             "chain; the reference raises here (uint16 offset -1)", node)
                             : gki_set_error(GKI_ERR_BAD_ARG, "critical paths: node %d does not have exactly one linear-ref successor "
                                             "(the reference requires exactly one)", node);
                goto done;
            }
            if (found > 0) {
                HIP_G(gki_dev_malloc((void **)&d_out_nodes, (size_t)found * 4));
                HIP_G(gki_dev_malloc((void **)&d_out_off, (size_t)found * 2));
                hipLaunchKernelGGL(k_walk_emit, dim3(stream_grid(total, 256)), dim3(256), 0, s, (const int32_t *)path, (const uint32_t *)crit,
                                   (const uint32_t *)pos, (const uint16_t *)offp, total, d_out_nodes, d_out_off);
                HIP_G(hipGetLastError());
                HIP_G(hipMemcpyAsync(h_out_nodes, d_out_nodes, (size_t)found * 4, hipMemcpyDeviceToHost, s));
                HIP_G(hipMemcpyAsync(h_out_offsets, d_out_off, (size_t)found * 2, hipMemcpyDeviceToHost, s));
                HIP_G(hipStreamSynchronize(s));
            }
            *n_out = found;
        }
    }
done:
    for (int i = 1; i <= rounds; i++) (void)gki_dev_free(jump[i]);
    free(jump);
    (void)gki_dev_free(cnt[0]); (void)gki_dev_free(cnt[1]); (void)gki_dev_free(arena); (void)gki_dev_free(arena2);
    (void)gki_dev_free(d_out_nodes); (void)gki_dev_free(d_out_off);
#undef HIP_G
    return rc;
}
