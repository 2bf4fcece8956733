// Forward windows from given start positions: DenseKmerFinder.find_only_kmers_starting_at_position
// (kmer_finder.py:170-177, early-stop mode of search_from :254-347): every forward path of exactly k real bases
// that starts at (node, offset) yields one window; records are reported with the END position as
// start_node/start_offset like every other record of the finder (:324).  A branch onto a non-linear-ref
// successor is taken only while the path holds fewer than max_variant_nodes variant nodes (:391-403; the window
// of an early-stop search is the whole path so far).  One lane per start position, depth-first in successor-list
// order, so the records of one start position come out in the reference's order.
#include "gki_common.h"
#include <limits.h>
#include <math.h>

namespace {
constexpr int FMAX = GKI_MAX_WINDOW_NODES;

struct FwdOut { int64_t *hash; int32_t *start_node; int16_t *start_offset; int32_t *node; double *af; };

// only_follow_nodes (kmer_finder.py:386-388): when a node has successors in the follow set, only those are taken
// and the variant limit is waived for that step.
__device__ __forceinline__ bool any_followed(const DevGraph &g, const uint8_t *__restrict__ follow, int32_t node) {
    if (!follow) return false;
    for (int64_t e = g.edge_start[node]; e < g.edge_start[node + 1]; e++) if (follow[g.edges[e]]) return true;
    return false;
}

// kmer_finder.py:397-402: at the limit only the linear-ref successor is followed, and the reference asserts that there is
// exactly one (no assertion when the node has no successors :390 or when the step is forced :397).
__device__ __forceinline__ void check_one_ref_successor(const DevGraph &g, int32_t node, int *err) {
    const int64_t e0 = g.edge_start[node], e1 = g.edge_start[node + 1];
    if (e1 == e0) return;
    int n_ref = 0;
    for (int64_t e = e0; e < e1; e++) n_ref += g.is_ref[g.edges[e]] ? 1 : 0;
    if (n_ref != 1) gki_raise(err, GKI_ERR_NOT_ONE_REF_SUCC);
}

template <bool EMIT>
__device__ void forward_walk(const DevGraph &g, int k, int M, bool one_node, const uint8_t *__restrict__ follow,
                             int32_t n0, int32_t o0, int64_t idx, FwdOut out, uint32_t *count_out, int *err) {
    int32_t nd[FMAX], cur[FMAX], end[FMAX], last[FMAX];
    uint8_t have[FMAX], vc[FMAX], forced[FMAX];
    uint64_t hs[FMAX];
    uint32_t count = 0;
    const NodeWalk w0 = g.walk[n0];
    if (o0 < 0 || o0 > w0.size) { *count_out = 0; return; }
    // level 0: the start node from offset o0 (an empty start node contributes no base)
    int L = 0;
    {
        const int avail = w0.size - o0;
        const int t = avail < k ? avail : k;
        nd[0] = n0; vc[0] = (uint8_t)(w0.is_ref ? 0 : 1);
        hs[0] = t > 0 ? gki_extract(g.seq2, w0.seq_start + o0, t) : 0ull;
        have[0] = (uint8_t)t;
        cur[0] = (int32_t)g.edge_start[n0]; end[0] = (int32_t)g.edge_start[n0 + 1];
        forced[0] = any_followed(g, follow, n0) ? 1 : 0;
        last[0] = INT_MIN;
        L = 1;
        if (t == k) { cur[0] = end[0]; }       // window complete inside the start node: handled below as a completion
        else if (!EMIT && !forced[0] && vc[0] >= M) check_one_ref_successor(g, n0, err);
    }
    // completion inside the start node
    if (have[0] == k) {
        if (EMIT) {
            out.hash[idx] = (int64_t)hs[0]; out.start_node[idx] = n0; out.start_offset[idx] = (int16_t)(o0 + k - 1);
            out.node[idx] = n0; out.af[idx] = g.allele_freq[n0];
        }
        *count_out = 1;
        return;
    }
    while (L > 0) {
        const int j = L - 1;
        if (cur[j] >= end[j]) { L--; continue; }
        int32_t q;
        if (forced[j]) {
            // :386-388 forced traversal: only the successors in the follow set.  The reference iterates a Python set
            // there, and CPython orders a set of small ints by hash & table mask, not by value (list({7, 8}) is
            // [8, 7]): among SEVERAL forced successors of one node the reference's order is an accident of the
            // interpreter and is not reproduced.  They are taken in ascending id here (deterministic); parity for
            // such nodes is on the multiset of records.  last[j] is the latest forced successor taken.
            q = INT_MAX;
            for (int32_t e = (int32_t)g.edge_start[nd[j]]; e < end[j]; e++) {
                const int32_t c = g.edges[e];
                if (follow[c] && c > last[j] && c < q) q = c;
            }
            if (q == INT_MAX) { cur[j] = end[j]; continue; }
            last[j] = q;
        } else {
            q = g.edges[cur[j]++];
        }
        const NodeWalk wq = g.walk[q];
        if (forced[j]) {
        } else if (vc[j] >= M && !wq.is_ref) {
            continue;                                                   // :397-403 only the linear-ref successor
        }
        if (L >= FMAX - 1) { gki_raise(err, GKI_ERR_WINDOW_TOO_DEEP); continue; }
        const int hv = have[j];
        const int t = wq.size < k - hv ? wq.size : k - hv;
        nd[L] = q; vc[L] = (uint8_t)(vc[j] + (wq.is_ref ? 0 : 1));
        hs[L] = hs[j] | (t > 0 ? gki_extract(g.seq2, wq.seq_start, t) << (2 * hv) : 0ull);
        have[L] = (uint8_t)(hv + t);
        if (hv + t == k) {                          // first k-mer of this path: emit and stop (early stop, :326-330)
            const int Lw = L + 1;
            if (EMIT) {
                int32_t mn = INT_MAX; double maf = INFINITY;
                for (int i = 0; i < Lw; i++) { mn = nd[i] < mn ? nd[i] : mn; maf = fmin(maf, g.allele_freq[nd[i]]); }
                if (one_node) {
                    out.hash[idx] = (int64_t)hs[L]; out.start_node[idx] = q; out.start_offset[idx] = (int16_t)(t - 1);
                    out.node[idx] = mn; out.af[idx] = maf; idx++;
                } else {
                    // one record per distinct node, ascending (np.unique, kmer_finder.py:134).  Node ids usually grow along
                    // a forward path: then the path is the order (one pass instead of a selection per record)
                    bool asc = true;
                    for (int i = 1; i < Lw; i++) asc = asc && nd[i] > nd[i - 1];
                    if (asc) {
                        for (int r = 0; r < Lw; r++) {
                            out.hash[idx] = (int64_t)hs[L]; out.start_node[idx] = q; out.start_offset[idx] = (int16_t)(t - 1);
                            out.node[idx] = nd[r]; out.af[idx] = maf; idx++;
                        }
                    } else {
                        int32_t last = INT_MIN;
                        for (int r = 0; r < Lw; r++) {
                            int32_t best = INT_MAX;
                            for (int i = 0; i < Lw; i++) if (nd[i] > last && nd[i] < best) best = nd[i];
                            out.hash[idx] = (int64_t)hs[L]; out.start_node[idx] = q; out.start_offset[idx] = (int16_t)(t - 1);
                            out.node[idx] = best; out.af[idx] = maf; idx++;
                            last = best;
                        }
                    }
                }
            }
            count += one_node ? 1u : (uint32_t)Lw;
            continue;
        }
        cur[L] = (int32_t)g.edge_start[q]; end[L] = (int32_t)g.edge_start[q + 1];
        forced[L] = any_followed(g, follow, q) ? 1 : 0;
        last[L] = INT_MIN;
        if (!EMIT && !forced[L] && vc[L] >= M) check_one_ref_successor(g, q, err);
        L++;
    }
    *count_out = count;
}

// One lane per start position, every read from global memory: latency-bound, so resident waves are what counts.  Without
// the occupancy request the compiler expands the per-level stacks into register select chains (178 VGPRs = 2 waves per
// SIMD, 2 200 instructions); with it 25-32 VGPRs, 700 instructions, 8 waves.  tools/bench_forward.py, 1.14e7 starts on the
// 1 Gbp graph, same box: 8.29 -> 5.15 ms per batch (all nodes), 6.81 -> 3.0 ms (one node per k-mer).
// Then, all-nodes mode: an ascending path is written straight through instead of a selection per record: 5.0 -> 4.05 ms.
// Tried and dropped: carrying the path's smallest node and minimum allele frequency down the walk instead of looping
// over the path at every finished k-mer -- two more scratch stores per step cost more than the loops (5.0 -> 5.2 ms).
template <bool EMIT>
__global__ __launch_bounds__(64, 8) void k_forward(DevGraph g, int k, int M, int one_node, const uint8_t *__restrict__ follow,
                                                const int32_t *__restrict__ nodes,
                                                const int32_t *__restrict__ offsets, int64_t n_pos,
                                                uint32_t *__restrict__ cnt, const int64_t *__restrict__ rec_start, FwdOut out,
                                                int *__restrict__ err) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos) return;
    uint32_t c = 0;
    const int32_t n0 = nodes[i];
    if (n0 < 0 || n0 >= g.n_nodes) { if (!EMIT) cnt[i] = 0; return; }
    forward_walk<EMIT>(g, k, M, one_node != 0, follow, n0, offsets[i], EMIT ? rec_start[i] : 0, out, &c, err);
    if (!EMIT) cnt[i] = c;
}
}  // namespace

extern "C" {

int gki_forward_count(gki_graph *gr, int k, int max_variant_nodes, int one_node, const void *d_follow, const void *d_nodes,
                      const void *d_offsets, int64_t n_pos, void *d_rec_start, int64_t *n_records) {
    *n_records = 0;
    if (k < 1 || k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31");
    if (n_pos <= 0) { HIP_TRY(hipMemset(d_rec_start, 0, 8)); return GKI_OK; }
    uint32_t *cnt = nullptr; void *tmp = nullptr; int *d_err = nullptr;
    int64_t tmp_bytes = gki_scan_tmp_bytes(n_pos);
    HIP_TRY(gki_dev_malloc((void **)&cnt, (size_t)n_pos * 4));
    HIP_TRY(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
    HIP_TRY(gki_dev_malloc((void **)&d_err, 4));
    HIP_TRY(hipMemset(d_err, 0, 4));
    FwdOut none{nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_forward<false>, dim3((unsigned)ceil_div(n_pos, 64)), dim3(64), 0, 0, gr->d, k, max_variant_nodes > 250 ? 250 : max_variant_nodes,
                       one_node, (const uint8_t *)d_follow, (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, cnt,
                       (const int64_t *)nullptr, none, d_err);
    int rc = hipGetLastError() == hipSuccess ? GKI_OK : gki_set_error(GKI_ERR_HIP, "k_forward launch failed");
    if (rc == GKI_OK) rc = gki_scan_u32_to_i64(cnt, n_pos, (int64_t *)d_rec_start, tmp, tmp_bytes, 0);
    int64_t total = 0; int herr = 0;
    hipError_t e1 = hipMemcpy(&total, (const int64_t *)d_rec_start + n_pos, 8, hipMemcpyDeviceToHost);
    hipError_t e2 = hipMemcpy(&herr, d_err, 4, hipMemcpyDeviceToHost);
    (void)gki_dev_free(cnt); (void)gki_dev_free(tmp); (void)gki_dev_free(d_err);
    if (rc != GKI_OK) return rc;
    HIP_TRY(e1); HIP_TRY(e2);
    herr = gki_error_of_word(herr);
    if (herr == GKI_ERR_NOT_ONE_REF_SUCC)
        return gki_set_error(herr, "a path at the variant limit ends a node that does not have exactly one linear-ref "
                             "successor: the reference asserts here (kmer_finder.py:402)");
    if (herr) return gki_set_error(herr, "a forward k-window crosses more than %d nodes", FMAX - 2);
    *n_records = total;
    return GKI_OK;
}

int gki_forward_emit(gki_graph *gr, int k, int max_variant_nodes, int one_node, const void *d_follow, const void *d_nodes,
                     const void *d_offsets, int64_t n_pos, const void *d_rec_start, void *d_hashes, void *d_start_nodes, void *d_start_offsets,
                     void *d_nodes_out, void *d_af64) {
    if (n_pos <= 0) return GKI_OK;
    int *d_err = nullptr;
    HIP_TRY(gki_dev_malloc((void **)&d_err, 4));
    HIP_TRY(hipMemset(d_err, 0, 4));
    FwdOut out{(int64_t *)d_hashes, (int32_t *)d_start_nodes, (int16_t *)d_start_offsets, (int32_t *)d_nodes_out, (double *)d_af64};
    hipLaunchKernelGGL(k_forward<true>, dim3((unsigned)ceil_div(n_pos, 64)), dim3(64), 0, 0, gr->d, k, max_variant_nodes > 250 ? 250 : max_variant_nodes,
                       one_node, (const uint8_t *)d_follow, (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos,
                       (uint32_t *)nullptr, (const int64_t *)d_rec_start, out, d_err);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    (void)gki_dev_free(d_err);
    HIP_TRY(e); HIP_TRY(e2);
    return GKI_OK;
}

}  // extern "C"
