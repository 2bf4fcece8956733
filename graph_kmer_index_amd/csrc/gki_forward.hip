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
#ifndef GKI_FWD_LEVELS
#define GKI_FWD_LEVELS GKI_MAX_WINDOW_NODES
#endif
constexpr int FMAX = GKI_FWD_LEVELS;

struct FwdOut { int64_t *hash; int32_t *start_node; int16_t *start_offset; int32_t *node; double *af; };

// The output columns are written once and never read by this kernel.  With one record per k-mer (one_node) they leave
// with the non-temporal hint, as the boundary kernels' records do (csrc/gki_finder.hip): the lines of the graph the walk
// reads stay in L2 -- 2.98 -> 2.79 ms per batch of 1.14e7 start positions, same box, alternating.  In all-nodes mode a
// lane writes a k-mer's records (one per path node) into consecutive slots and the plain stores merge into whole lines
// in L2; the hint sends every partial line to memory on its own: 3.96 -> 7.3 ms, so that mode keeps plain stores
// (profiles/r03_forward_nt_stores_ab.txt; -DGKI_FWD_NT=0 / =2 rebuild the two partners: hint nowhere / everywhere).
#ifndef GKI_FWD_NT
#define GKI_FWD_NT 1
#endif
template <bool NT, class T> __device__ __forceinline__ void fst(T *p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool ONE>
__device__ __forceinline__ void put_record(const FwdOut &out, int64_t idx, uint64_t h, int32_t end_node, int end_off, int32_t node, double af) {
    constexpr bool NT = GKI_FWD_NT == 2 || (GKI_FWD_NT == 1 && ONE);
    fst<NT>(&out.hash[idx], (int64_t)h); fst<NT>(&out.start_node[idx], end_node); fst<NT>(&out.start_offset[idx], (int16_t)end_off);
    fst<NT>(&out.node[idx], node); fst<NT>(&out.af[idx], af);
}

// "Script" of a search (GKI_FWD_SCRIPT, round 3): the emit pass used to repeat the count pass's whole walk.  Instead the
// count pass writes every finished k-mer down -- hash, end position, minimum allele frequency, the path's nodes, the
// number of the k-mer's first record among its start position's -- in one of FW_SLOTS 48-byte entries per start position,
// and the emit pass is a streaming expansion with one thread per entry.  A start position with more finished k-mers, a
// path over more than FW_SN nodes or a path whose node ids do not ascend is marked 0xFF and walked by the emit kernel as
// before (with SNP bubbles ~600 bases apart and FW_SLOTS = 4 that is one start in several hundred).  The script lives in
// gki_graph::fwd_script between gki_forward_count and the gki_forward_emit call with the same arguments; any other emit
// call, the slow path for deep windows, or no memory for the script (192 B per start position) mean the emit pass walks.
// Measured, 1.14e7 start positions on the 1 Gbp graph, alternating on one box (profiles/r03_forward_script_ab.txt):
// all-nodes mode 4.16 -> 3.73 ms, and 3.75 -> 3.21 ms with short_path_facts below; one node per k-mer 2.98 -> 3.06 ms
// with round 3's layout of the script.  With round 4's (below) both modes take it: 3 Gbp graph, 3.43e7 start positions,
// all nodes 8.1 -> 6.2 ms, one node per k-mer 6.7 -> 5.1 ms (profiles/r04_forward_script_layout_ab.txt).
#ifndef GKI_FWD_SCRIPT
#define GKI_FWD_SCRIPT 1
#endif
// Layout (GKI_FWD_SCRIPT_SOA, round 4): the three 16-byte pieces of slot c of start position i lie at
// [(c * 3 + piece) * n_pos + i] -- the lanes of a wave (neighbouring start positions, mostly in step) write a piece of
// their c-th k-mer into consecutive 16-byte cells, whole lines per store instruction, and slots nobody uses are lines nobody
// touches.  (Rounds 3-4 had the four 48-byte entries of a start position side by side, 192 B apart from lane to lane: 64
// partly written lines per store instruction, and the script's 100 written bytes per start position cost the count pass
// 1.9 ms on top of a 2.9 ms walk.)  =0 rebuilds that layout.
#ifndef GKI_FWD_SCRIPT_ONE
#define GKI_FWD_SCRIPT_ONE 1
#endif
#ifndef GKI_FWD_SCRIPT_SOA
#define GKI_FWD_SCRIPT_SOA 1
#endif
constexpr int FW_SLOTS = 4, FW_SN = 5, FW_ENTRY_U4 = 3;
// cell of (start position i, slot c, piece p) in units of uint4
__device__ __forceinline__ size_t script_cell(int64_t i, int c, int p, int64_t n_pos) {
    return GKI_FWD_SCRIPT_SOA ? (size_t)(c * FW_ENTRY_U4 + p) * (size_t)n_pos + (size_t)i
                              : ((size_t)i * FW_SLOTS + (size_t)c) * FW_ENTRY_U4 + (size_t)p;
}
// An entry: piece 0 = (hash, minimum allele frequency); piece 1 = (end node q, end offset | records << 16 | flags << 24, first two
// nodes of the list); piece 2 = (the list's third and fourth node), written and read only when the list is that long.  The
// list: in all-nodes mode the path's nodes but the last, which is q (a path over three nodes -- start node, allele, the node
// that completes the k-mer -- is two pieces); in one-node mode (flag bit 0) the one node the record reports.  The number of
// the entry's first record among its start position's is not stored: it is the sum of the records of the slots before it.
// The script's cells leave with the non-temporal hint: they are read once, by another kernel, and the lines of the graph the
// walk reads stay in L2 -- all nodes 4.27 -> 3.97 ms, same box, alternating (profiles/r04_forward_script_layout_ab.txt;
// -DGKI_FWD_SCRIPT_NT=0 rebuilds the plain stores).
#ifndef GKI_FWD_SCRIPT_NT
#define GKI_FWD_SCRIPT_NT 1
#endif
__device__ __forceinline__ void script_store(uint4 *p, const uint4 &v) {
    if (GKI_FWD_SCRIPT_NT) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p));
    } else *p = v;
}
__device__ __forceinline__ void script_write(uint4 *script, int64_t i, int c, int64_t n_pos, uint64_t h, double maf, int32_t q, int off, int lw,
                                             bool one_node, const int32_t *nodes) {
    const uint64_t mb = (uint64_t)__double_as_longlong(maf);
    const int listed = one_node ? 1 : lw - 1;
    const uint4 p0 = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)mb, (uint32_t)(mb >> 32));
    const uint4 p1 = make_uint4((uint32_t)q, ((uint32_t)off & 0xFFFFu) | ((uint32_t)lw << 16) | (one_node ? 1u << 24 : 0u),
                                (uint32_t)nodes[0], (uint32_t)nodes[1]);
    script_store(&script[script_cell(i, c, 0, n_pos)], p0);
    script_store(&script[script_cell(i, c, 1, n_pos)], p1);
    if (listed > 2) script_store(&script[script_cell(i, c, 2, n_pos)], make_uint4((uint32_t)nodes[2], (uint32_t)nodes[3], 0u, 0u));
}

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

// byte offset of every per-level array of the walk inside one (level, lane) cell of the slow path's arena (DeepArena)
enum { FW_NM = 0, FW_CE = 8, FW_HS = 16, FW_MF = 24, FW_LAST = 32, FW_CELL = 36 };
// A path that has not reached its first k-mer after this many descents below level FW_BUDGET_FROM belongs to an
// exponential family (a run of insertion sites with no variant limit to cut it): the search from that position is wound
// up and the call refused, as in the finder (csrc/gki_finder.hip, STEP_BUDGET).
constexpr int FW_BUDGET = 1 << 22, FW_BUDGET_FROM = 8;     // (every step of this walk reads global memory: ~1 us each)

// The walk's levels: the first GKI_FWD_REG_LEVELS of them in registers (a read is a select over R values, a write R
// predicated moves), the rest in the stack behind (scratch; the slow path's arena).  A forward path of the usual search --
// the start node, an allele, the node that completes the k-mer, which is never stored -- lives in two levels, so the
// product kernel touches no scratch at all for it (rounds 2-4 kept every level in scratch: three stores per descent and
// two loads per turn of the loop, each a trip to L2 -- a wave's levels do not fit L1 beside its neighbours').
#ifndef GKI_FWD_BLOCK
#define GKI_FWD_BLOCK 64                  // threads per workgroup of the product kernels (one start position per lane)
#endif
#ifndef GKI_FWD_WAVES
#define GKI_FWD_WAVES 8                   // waves per SIMD the product kernels are held to
#endif
#ifndef GKI_FWD_REG_LEVELS
#define GKI_FWD_REG_LEVELS 2
#endif
// (the register levels are a variable of their own, apart from the stack behind them: one aggregate holding both is one
// stack object to the compiler, and the dynamically indexed part keeps the whole of it in scratch)
template <class T, int R> struct RegLevels { T r[R > 0 ? R : 1]; };
template <class T, int R, class ST> __device__ __forceinline__ T lv_get(const RegLevels<T, R> &rg, ST &st, int j) {
    if (R > 0 && j < R) {
        T v = rg.r[0];
#pragma unroll
        for (int i = 1; i < R; i++) if (j == i) v = rg.r[i];
        return v;
    }
    return st[j];
}
template <class T, int R, class ST> __device__ __forceinline__ void lv_set(RegLevels<T, R> &rg, ST &st, int j, T v) {
    if (R > 0 && j < R) {
#pragma unroll
        for (int i = 0; i < R; i++) if (j == i) rg.r[i] = v;
    } else st[j] = v;
}

// A node's record in two 16-byte loads (field by field the compiler asks for it in four)
__device__ __forceinline__ NodeFwd fwd_node(const NodeFwd *__restrict__ fw, int32_t n) {
    const uint4 *p = reinterpret_cast<const uint4 *>(fw + n);
    const uint4 a = p[0], b = p[1];
    NodeFwd w;
    w.head = (uint64_t)a.x | ((uint64_t)a.y << 32);
    w.af = __longlong_as_double((long long)((uint64_t)a.z | ((uint64_t)a.w << 32)));
    w.size = (int32_t)b.x; w.e0 = (int32_t)b.y; w.e1 = (int32_t)b.z;
    w.cnt = (uint16_t)(b.w & 0xFFFFu); w.is_ref = (uint8_t)((b.w >> 16) & 0xFFu); w.pad = 0;
    return w;
}

// Successors of a node as the level's (cur, end) pair.  end >= 0: cur .. end index g.edges.  end < 0: the successors are in
// the pair itself (NodeFwd: at most two) -- cur is the next one, or -1 when there is none left, and ~end the one after it
// (INT_MIN: none).
#define FW_CE(cur_, end_) ((uint64_t)(uint32_t)(cur_) | ((uint64_t)(uint32_t)(end_) << 32))
__device__ __forceinline__ uint64_t succ_begin(const DevGraph &g, const NodeFwd &w, int32_t n) {
    if (w.cnt == 0) return FW_CE(-1, INT_MIN);
    if (w.cnt == 1) return FW_CE(w.e0, INT_MIN);
    if (w.cnt == 2) return FW_CE(w.e0, ~w.e1);
    return FW_CE(w.e0, w.cnt == 0xFFFF ? (int32_t)g.edge_start[n + 1] : w.e0 + (int32_t)w.cnt);
}

// DEEP = false: the product kernel, its levels in registers and, beyond those, FMAX levels per lane in scratch.  DEEP = true:
// the slow path for forward windows over more nodes than that (sixteen or more empty nodes before the first k-mer is
// complete), the same walk with its levels in a global-memory arena of da.cap levels per lane (gki_forward_count grows it
// until the walk fits).
template <bool EMIT, bool DEEP, bool SCRIPT>
__device__ void forward_walk(const DevGraph &g, const NodeFwd *__restrict__ fw, int k, int M, bool one_node, const uint8_t *__restrict__ follow,
                             int32_t n0, int32_t o0, int64_t idx, FwdOut out, uint32_t *count_out, int *err,
                             const DeepArena &da, int64_t lane_global, uint4 *script, int64_t pos, int64_t n_pos, uint32_t *used_out) {
    static_assert(!SCRIPT || (!EMIT && !DEEP), "the script is written by the product count kernel");
    constexpr int R = DEEP ? 0 : GKI_FWD_REG_LEVELS;
    uint32_t used = 0;                        // SCRIPT: entries written, 0xFF = this start position does not fit
    // Per level four 64-bit words: (node, meta) with meta = bases collected (8 bits) | "forced traversal" (8) | variant
    // nodes on the path (16); the successors still to take (cur, end); the hash so far; the smallest allele frequency on the
    // path so far (it comes with the node's record, so a finished k-mer asks for nothing).  The level that completes a k-mer
    // (every path's last) is never stored.  `last` (forced traversal only) is touched only when a follow set is given.
    typename StackOf<int32_t, FMAX, DEEP>::type last;
    typename StackOf<uint64_t, FMAX, DEEP>::type nm_st, ce_st, hs_st;
    typename StackOf<double, FMAX, DEEP>::type mf_st;
    RegLevels<uint64_t, R> nm_r, ce_r, hs_r;
    RegLevels<double, R> mf_r;
    bind(nm_st, da, FW_NM, lane_global); bind(ce_st, da, FW_CE, lane_global); bind(hs_st, da, FW_HS, lane_global);
    bind(mf_st, da, FW_MF, lane_global); bind(last, da, FW_LAST, lane_global);
#define FW_NODE_OF(x_) ((int32_t)(uint32_t)(x_))
#define FW_META_OF(x_) ((uint32_t)((x_) >> 32))
#define FW_MK(have_, forced_, vc_) ((uint32_t)(have_) | ((uint32_t)(forced_) << 8) | ((uint32_t)(vc_) << 16))
#define FW_HAVE_OF(m_) ((int)((m_) & 0xFFu))
#define FW_FORCED_OF(m_) ((int)(((m_) >> 8) & 0xFFu))
#define FW_VC_OF(m_) ((int)((m_) >> 16))
    const int cap = DEEP ? da.cap : FMAX;
    int steps_left = FW_BUDGET;
    uint32_t count = 0;
    const NodeFwd w0 = fwd_node(fw, n0);
    const uint4 st = reinterpret_cast<const uint4 *>(g.walk + n0)[0];        // NodeWalk: (seq_start, tail)
    if (o0 < 0 || o0 > w0.size) { *count_out = 0; if (SCRIPT) *used_out = 0; return; }
    // level 0: the start node from offset o0 (an empty start node contributes no base)
    int L = 0, have0 = 0;
    uint64_t h0 = 0;
    const double maf0 = fmin((double)INFINITY, w0.af);
    {
        const int avail = w0.size - o0;
        const int t = avail < k ? avail : k;
        // The start node's bases from o0.  A window that goes on into a successor takes the node's LAST t bases: they are in
        // the boundary walk's record of the node (NodeWalk::tail, its first 16 bytes asked for together with the search's own
        // record) -- no trip to the sequence, which was the third dependent round trip of every start position.  A window
        // complete inside the start node: inside the node's first 31 bases from the record's head, else from the sequence.
        if (t > 0 && t < k) {
            const int t31 = w0.size < 31 ? w0.size : 31;
            h0 = (((uint64_t)st.z | ((uint64_t)st.w << 32)) >> (2 * (t31 - t))) & ((1ull << (2 * t)) - 1ull);
        } else if (t > 0) {
            h0 = o0 + t <= 31 ? (w0.head >> (2 * o0)) & ((1ull << (2 * t)) - 1ull)
                              : gki_extract(g.seq2, (int64_t)((uint64_t)st.x | ((uint64_t)st.y << 32)) + o0, t);
        }
        have0 = t;
        if (t < k) {                           // (t == k: window complete inside the start node, handled below)
            const int vc0 = w0.is_ref ? 0 : 1, forced0 = any_followed(g, follow, n0) ? 1 : 0;
            lv_set(nm_r, nm_st, 0, (uint64_t)(uint32_t)n0 | ((uint64_t)FW_MK(t, forced0, vc0) << 32));
            lv_set(hs_r, hs_st, 0, h0);
            lv_set(mf_r, mf_st, 0, maf0);
            lv_set(ce_r, ce_st, 0, forced0 ? FW_CE((int32_t)g.edge_start[n0], (int32_t)g.edge_start[n0 + 1]) : succ_begin(g, w0, n0));
            if (follow) last[0] = INT_MIN;
            L = 1;
            if (!EMIT && !forced0 && vc0 >= M) check_one_ref_successor(g, n0, err);
        }
    }
    // completion inside the start node
    if (have0 == k) {
        if (EMIT) {
            put_record<false>(out, idx, h0, n0, o0 + k - 1, n0, w0.af);
        }
        if (SCRIPT) {
            const int32_t one[FW_SN] = {n0, 0, 0, 0, 0};
            script_write(script, pos, 0, n_pos, h0, w0.af, n0, o0 + k - 1, 1, true, one);
            *used_out = 1;
        }
        *count_out = 1;
        return;
    }
    while (L > 0) {
        const int j = L - 1;
        const uint64_t cej = lv_get(ce_r, ce_st, j);
        const int32_t curj = (int32_t)(uint32_t)cej, endj = (int32_t)(uint32_t)(cej >> 32);
        if (endj >= 0 ? curj >= endj : curj < 0) { L--; continue; }
        const uint64_t nmj = lv_get(nm_r, nm_st, j);
        const uint32_t mj = FW_META_OF(nmj);
        int32_t q;
        if (FW_FORCED_OF(mj)) {
            // :386-388 forced traversal: only the successors in the follow set.  The reference iterates a Python set
            // there, and CPython orders a set of small ints by hash & table mask, not by value (list({7, 8}) is
            // [8, 7]): among SEVERAL forced successors of one node the reference's order is an accident of the
            // interpreter and is not reproduced.  They are taken in ascending id here (deterministic); parity for
            // such nodes is on the multiset of records.  last[j] is the latest forced successor taken.
            // (a forced level's (cur, end) always index g.edges)
            q = INT_MAX;
            for (int32_t e = (int32_t)g.edge_start[FW_NODE_OF(nmj)]; e < endj; e++) {
                const int32_t c = g.edges[e];
                if (follow[c] && c > last[j] && c < q) q = c;
            }
            if (q == INT_MAX) { lv_set(ce_r, ce_st, j, FW_CE(endj, endj)); continue; }
            last[j] = q;
        } else if (endj < 0) {
            q = curj;
            lv_set(ce_r, ce_st, j, endj == INT_MIN ? FW_CE(-1, INT_MIN) : FW_CE(~endj, INT_MIN));
        } else {
            q = g.edges[curj];
            lv_set(ce_r, ce_st, j, FW_CE(curj + 1, endj));
        }
        const NodeFwd wq = fwd_node(fw, q);
        if (FW_FORCED_OF(mj)) {
        } else if (FW_VC_OF(mj) >= M && !wq.is_ref) {
            continue;                                                   // :397-403 only the linear-ref successor
        }
        if (L >= cap - 1) { gki_raise(err, GKI_ERR_WINDOW_TOO_DEEP); continue; }
        const int hv = FW_HAVE_OF(mj);
        const int t = wq.size < k - hv ? wq.size : k - hv;              // (t <= 31 - hv: the record's head holds these bases)
        const int vcL = FW_VC_OF(mj) + (wq.is_ref ? 0 : 1);
        const uint64_t hL = lv_get(hs_r, hs_st, j) | (t > 0 ? (wq.head & ((1ull << (2 * t)) - 1ull)) << (2 * hv) : 0ull);
        const double mafL = fmin(lv_get(mf_r, mf_st, j), wq.af);
        auto node_at = [&](int i) -> int32_t { return i == L ? q : FW_NODE_OF(lv_get(nm_r, nm_st, i)); };      // (level L itself is not stored)
        if (hv + t == k) {                          // first k-mer of this path: emit and stop (early stop, :326-330)
            const int Lw = L + 1;
            // the path's nodes when there are at most FW_SN of them (slots beyond the path: 0), whether their ids ascend, and
            // the smallest of them
            int32_t v[FW_SN] = {0, 0, 0, 0, 0}; int32_t mn = q;
            bool asc = true;
            const bool short_path = Lw <= FW_SN;
            if (short_path) {
#pragma unroll
                for (int r = 0; r < FW_SN; r++) v[r] = r < L ? FW_NODE_OF(lv_get(nm_r, nm_st, r)) : r == L ? q : 0;
#pragma unroll
                for (int r = 0; r < FW_SN; r++) if (r < L) mn = v[r] < mn ? v[r] : mn;
#pragma unroll
                for (int r = 1; r < FW_SN; r++) if (r < Lw) asc = asc && v[r] > v[r - 1];
            } else {
                for (int i = 0; i < L; i++) { const int32_t ni = node_at(i); mn = ni < mn ? ni : mn; }
                if (!one_node) for (int i = 1; i < Lw; i++) asc = asc && node_at(i) > node_at(i - 1);
            }
            if (EMIT) {
                if (one_node) {
                    put_record<true>(out, idx, hL, q, t - 1, mn, mafL); idx++;
                } else if (asc) {
                    // one record per distinct node, ascending (np.unique, kmer_finder.py:134).  Node ids usually grow along
                    // a forward path: then the path is the order (one pass instead of a selection per record)
                    if (short_path) {
#pragma unroll
                        for (int r = 0; r < FW_SN; r++)
                            if (r < Lw) { put_record<false>(out, idx, hL, q, t - 1, v[r], mafL); idx++; }
                    } else {
                        for (int r = 0; r < Lw; r++) {
                            put_record<false>(out, idx, hL, q, t - 1, node_at(r), mafL); idx++;
                        }
                    }
                } else {
                    int32_t last = INT_MIN;
                    for (int r = 0; r < Lw; r++) {
                        int32_t best = INT_MAX;
                        for (int i = 0; i < Lw; i++) { const int32_t ni = node_at(i); if (ni > last && ni < best) best = ni; }
                        put_record<false>(out, idx, hL, q, t - 1, best, mafL); idx++;
                        last = best;
                    }
                }
            }
            if (SCRIPT && used != 0xFFu) {
                const bool fits = used < (uint32_t)FW_SLOTS && (one_node || (short_path && asc));
                if (fits) {
                    if (one_node) v[0] = mn;
                    script_write(script, pos, (int)used, n_pos, hL, mafL, q, t - 1, one_node ? 1 : Lw, one_node, v);
                    used++;
                } else used = 0xFFu;
            }
            count += one_node ? 1u : (uint32_t)Lw;
            continue;
        }
        const int forcedL = any_followed(g, follow, q) ? 1 : 0;
        lv_set(ce_r, ce_st, L, forcedL ? FW_CE((int32_t)g.edge_start[q], (int32_t)g.edge_start[q + 1]) : succ_begin(g, wq, q));
        lv_set(nm_r, nm_st, L, (uint64_t)(uint32_t)q | ((uint64_t)FW_MK(hv + t, forcedL, vcL) << 32));
        lv_set(hs_r, hs_st, L, hL);
        lv_set(mf_r, mf_st, L, mafL);
        if (follow) last[L] = INT_MIN;
        if (!EMIT && !forcedL && vcL >= M) check_one_ref_successor(g, q, err);
        L++;
        // (out of budget: the walk ends through its ordinary exit, see STEP_BUDGET in csrc/gki_finder.hip)
        if (L > FW_BUDGET_FROM && --steps_left < 0) { gki_raise_budget(err); L = 0; }
    }
    *count_out = count;
    if (SCRIPT) *used_out = used;
}

// One lane per start position, every read from global memory.  What the counters show (profiles/r04_forward_node_records_ab.txt):
// the wave slots are full and the waves wait nine cycles in ten; a wave lasts as long as its slowest lane and pays a memory
// round trip per memory instruction of every turn of the loop, so the kernel's time is (turns of the slowest lane) x (memory
// instructions per turn) x latency / (resident waves).  Hence, in the order they were found:
//  - resident waves: without the occupancy request the compiler expands the per-level stacks into register select chains
//    (178 VGPRs = 2 waves per SIMD); with it 8 waves.  1.14e7 starts on the 1 Gbp graph: 8.29 -> 5.15 ms (round 2);
//  - an ascending path is written straight through instead of by a selection per record: 5.0 -> 4.05 ms (round 2);
//  - the records of one-node mode leave with the non-temporal hint (2.98 -> 2.79 ms); all-nodes mode must not (round 3);
//  - the script (the emit pass stops walking; round 3) and its piece-major layout (round 4: 8.1 -> 6.2 ms, 3.43e7 starts);
//  - memory instructions per turn: one NodeFwd record per descent, the first levels in registers, the minimum allele
//    frequency carried down (round 4: 6.4 -> 4.7 ms; round 2 had tried carrying it in scratch: slower);
//  - 64-thread workgroups: with 256 a slot waits for the slowest of four waves (4.6 -> 4.8 ms).
//  - bytes of script: a three-node path's entry in two pieces instead of three (4.75 -> 4.3 ms).
// What did NOT move it: a dependent round trip fewer per descent (round 4's early-edges switch) or per start position
// (the start node's bases from NodeWalk::tail instead of the sequence: 1 %), fewer scratch levels, a third register level,
// 6 instead of 8 waves per SIMD.  So the bound is the memory instructions the waves issue -- each a request per distinct
// line among 64 lanes that share little -- not the length of a lane's dependent chain.
// Still there: the slowest lane -- a start position whose window crosses a second variant takes twice the turns of its
// neighbours, and they wait for it (a refilling state-machine form of the walk was tried against that and was slower:
// profiles/r04_forward_node_records_ab.txt).
template <bool EMIT, bool DEEP = false, bool SCRIPT = false>
__global__ __launch_bounds__(DEEP ? 64 : GKI_FWD_BLOCK, DEEP ? 1 : GKI_FWD_WAVES) void k_forward(DevGraph g, const NodeFwd *__restrict__ fw, int k, int M, int one_node, const uint8_t *__restrict__ follow,
                                                const int32_t *__restrict__ nodes,
                                                const int32_t *__restrict__ offsets, int64_t n_pos,
                                                uint32_t *__restrict__ cnt, const int64_t *__restrict__ rec_start, FwdOut out,
                                                int *__restrict__ err, DeepArena da, uint4 *__restrict__ script, uint8_t *__restrict__ used,
                                                int64_t *__restrict__ list, int64_t list_n) {
    // list: SCRIPT -- the start positions that do not fit the script are appended to it (up to list_n of them); EMIT -- when
    // given, only these list_n start positions are walked (what the script's expansion left out)
    const int64_t lane_global = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_items = EMIT && list ? list_n : n_pos;
    // product kernel: one position per lane; slow path: the arena's lanes walk the positions grid-stride
    for (int64_t t = lane_global; t < n_items; t += DEEP ? da.lanes : n_items) {
        const int64_t i = EMIT && list ? list[t] : t;
        uint32_t c = 0, u = 0;
        if (EMIT && used && !list && used[i] != 0xFF) continue;          // written by the expansion of the script
        const int32_t n0 = nodes[i];
        if (n0 < 0 || n0 >= g.n_nodes) { if (!EMIT) cnt[i] = 0; if (SCRIPT) used[i] = 0; continue; }
        forward_walk<EMIT, DEEP, SCRIPT>(g, fw, k, M, one_node != 0, follow, n0, offsets[i], EMIT ? rec_start[i] : 0, out, &c, err, da, lane_global,
                                         script, i, n_pos, &u);
        if (!EMIT) cnt[i] = c;
        if (SCRIPT) {
            used[i] = (uint8_t)u;
            if (u == 0xFFu) { const int slot = atomicAdd(err + 1, 1); if (slot < list_n) list[slot] = i; }
        }
    }
}

// Emit pass over the script.  One thread per entry decodes it; the RECORDS of a wave's entries -- neighbours in the output:
// the records of one start position follow each other, and so do neighbouring start positions' -- are then numbered across
// the wave (a prefix sum of the entries' record counts) and written one lane per record, 64 consecutive records per store
// instruction.  (Round 3 had every entry's lane write its own 1-5 records in a loop: neighbouring lanes stored three
// records apart, five partially filled store instructions per column: 2.9 ms of the 9.0 ms step on the 3 Gbp graph.)
// A record's entry is found by a binary search over the prefix sums (six LDS reads); its facts come from the entry's
// image in LDS.  Plain stores: start positions that did not fit the script leave gaps that the walking kernel fills.
__global__ __launch_bounds__(256) void k_forward_expand(const uint4 *__restrict__ script, const uint8_t *__restrict__ used,
                                                         const int64_t *__restrict__ rec_start, int64_t n_pos, FwdOut out) {
    __shared__ uint4 s_e[4][64][FW_ENTRY_U4];
    __shared__ int64_t s_idx[4][64];
    __shared__ uint16_t s_ex[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = tid / FW_SLOTS;
    const int c = (int)(tid % FW_SLOTS);
    int lw = 0;
    int64_t base = 0;
    if (i < n_pos) {
        const uint32_t n = used[i];
        if (n != 0xFFu && (uint32_t)c < n) {
            const uint4 a = script[script_cell(i, c, 0, n_pos)], b = script[script_cell(i, c, 1, n_pos)];
            lw = (int)((b.y >> 16) & 0xFFu);
            s_e[wave][lane][0] = a; s_e[wave][lane][1] = b;
            if (!(b.y >> 24 & 1u) && lw > 3) s_e[wave][lane][2] = script[script_cell(i, c, 2, n_pos)];       // (lists of three and four nodes)
            base = rec_start[i];
        }
    }
    const int incl = gki_wave_incl_sum(lw);
    const int R = gki_lane_value(incl, 63);
    const int excl = incl - lw;
    // an entry's first record among its start position's = the records of the slots before it: the four slots of a start
    // position are four neighbouring lanes, and the prefix sum at the first of them is what lies before the start position
    const int excl_slot0 = __shfl(excl, lane & ~3, 64);
    s_idx[wave][lane] = base + (int64_t)(excl - excl_slot0);
    s_ex[wave][lane] = (uint16_t)excl;
    __builtin_amdgcn_wave_barrier();
    for (int rr0 = 0; rr0 < R; rr0 += 64) {
        const int rr = rr0 + lane;
        if (rr < R) {
            int o = 0;                                          // the last lane whose first record is <= rr
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (o + step < 64 && (int)s_ex[wave][o + step] <= rr) o += step;
            const int t = rr - (int)s_ex[wave][o];
            const uint4 a = s_e[wave][o][0], b = s_e[wave][o][1];
            const uint32_t *listed = reinterpret_cast<const uint32_t *>(&s_e[wave][o][1]) + 2;      // b.z, b.w, d.x, d.y
            const int lw_o = (int)((b.y >> 16) & 0xFFu);
            const bool from_list = (b.y >> 24 & 1u) || t < lw_o - 1;                                // (else: the path's last node, q)
            const uint64_t h = (uint64_t)a.x | ((uint64_t)a.y << 32);
            const double maf = __longlong_as_double((long long)((uint64_t)a.z | ((uint64_t)a.w << 32)));
            put_record<false>(out, s_idx[wave][o] + t, h, (int32_t)b.x, (int)(int16_t)(b.y & 0xFFFFu), from_list ? (int32_t)listed[t] : (int32_t)b.x, maf);
        }
    }
}
// NodeFwd of every node (gki_common.h); the 2-bit sequence is in place (gki_graph_prepare ran)
__global__ __launch_bounds__(256) void k_build_fwd(DevGraph g, NodeFwd *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < g.n_nodes; n += stride) {
        NodeFwd w;
        w.size = g.node_size[n];
        const int t31 = w.size < 31 ? w.size : 31;
        w.head = t31 > 0 ? gki_extract(g.seq2, g.seq_start[n], t31) : 0ull;
        w.af = g.allele_freq[n];
        const int64_t e0 = g.edge_start[n], cnt = g.edge_start[n + 1] - e0;
        w.e0 = cnt == 1 || cnt == 2 ? g.edges[e0] : (int32_t)e0;
        w.e1 = cnt == 2 ? g.edges[e0 + 1] : 0;
        w.cnt = (uint16_t)(cnt < 0xFFFF ? cnt : 0xFFFF);
        w.is_ref = g.is_ref[n] ? 1 : 0;
        w.pad = 0;
        out[n] = w;
    }
}
}  // namespace

namespace {
// the search's per-node records: built by the first search on a graph (and again after gki_graph_prepare), 32 B per node
int fwd_nodes_ready(gki_graph *gr) {
    if (gr->fwd_nodes) return GKI_OK;
    void *p = nullptr;
    if (hipMalloc(&p, (size_t)gr->d.n_nodes * sizeof(NodeFwd)) != hipSuccess) {
        (void)hipGetLastError();
        return gki_set_error(GKI_ERR_HIP, "forward search: no memory for %lld node records", (long long)gr->d.n_nodes);
    }
    hipLaunchKernelGGL(k_build_fwd, dim3(stream_grid(gr->d.n_nodes, 256)), dim3(256), 0, 0, gr->d, (NodeFwd *)p);
    const hipError_t e = hipGetLastError(), e2 = hipStreamSynchronize(0);
    if (e != hipSuccess || e2 != hipSuccess) { (void)hipFree(p); HIP_TRY(e); HIP_TRY(e2); }
    gr->fwd_nodes = (NodeFwd *)p;
    return GKI_OK;
}

// the slow path's arena: grown by the call that needs it, returned to the pool when the search is over
void deep_release(gki_graph *gr) {
    if (gr->fwd_deep.base) (void)gki_dev_free(gr->fwd_deep.base);
    gr->fwd_deep = DeepArena{nullptr, 0, 0, 0};
    gr->fwd_deep_bytes = 0;
}
int deep_grow(gki_graph *gr, int next_cap) {
    const int64_t lanes = 64 * 256, bytes = lanes * (int64_t)next_cap * FW_CELL;
    if (bytes > gr->fwd_deep_bytes) {
        if (gr->fwd_deep.base) (void)gki_dev_free(gr->fwd_deep.base);
        gr->fwd_deep.base = nullptr; gr->fwd_deep_bytes = 0;
        if (gki_dev_malloc((void **)&gr->fwd_deep.base, (size_t)bytes) != hipSuccess)
            return gki_set_error(GKI_ERR_HIP, "forward search: no memory for %lld bytes of deep stacks", (long long)bytes);
        gr->fwd_deep_bytes = bytes;
    }
    gr->fwd_deep.lanes = lanes; gr->fwd_deep.cap = next_cap; gr->fwd_deep.pad = 0;
    return GKI_OK;
}

void script_drop(gki_graph *gr) {
    FwdScript &sc = gr->fwd_script;
    if (sc.entries) (void)gki_dev_free(sc.entries);
    if (sc.ncomp) (void)gki_dev_free(sc.ncomp);
    if (sc.over_list) (void)gki_dev_free(sc.over_list);
    sc = FwdScript{};
}
}  // namespace

extern "C" {

int gki_forward_count(gki_graph *gr, int k, int max_variant_nodes, int one_node, const void *d_follow, const void *d_nodes,
                      const void *d_offsets, int64_t n_pos, void *d_rec_start, int64_t *n_records) {
    *n_records = 0;
    if (k < 1 || k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31");
    GKI_TRY(gki_check_graph_device(gr, "gki_forward_count"));
    if (n_pos <= 0) { HIP_TRY(hipMemset(d_rec_start, 0, 8)); return GKI_OK; }
    GKI_TRY(fwd_nodes_ready(gr));
    deep_release(gr);                     // a slow-path arena of an earlier search goes back to the pool (up to 5.6 GB)
    uint32_t *cnt = nullptr; void *tmp = nullptr; int *d_err = nullptr;
    int64_t tmp_bytes = gki_scan_tmp_bytes(n_pos);
    HIP_TRY(gki_dev_malloc((void **)&cnt, (size_t)n_pos * 4));
    HIP_TRY(gki_dev_malloc(&tmp, (size_t)tmp_bytes));
    HIP_TRY(gki_dev_malloc((void **)&d_err, 8));        // [0] the error word, [1] start positions that did not fit the script
    HIP_TRY(hipMemset(d_err, 0, 8));
    FwdOut none{nullptr, nullptr, nullptr, nullptr, nullptr};
    const int M = max_variant_nodes > 250 ? 250 : max_variant_nodes;
    gr->fwd_deep.cap = 0;                 // the product kernel first; the emit call that follows uses what this call settles on
    // the script for the emit call (see FW_SLOTS above); without memory for it the emit call walks as it always did
    script_drop(gr);
    FwdScript &sc = gr->fwd_script;
    if (GKI_FWD_SCRIPT && (!one_node || GKI_FWD_SCRIPT_ONE)) {
        if (gki_dev_malloc(&sc.entries, (size_t)n_pos * FW_SLOTS * FW_ENTRY_U4 * 16) != hipSuccess ||
            gki_dev_malloc((void **)&sc.ncomp, (size_t)n_pos) != hipSuccess) { (void)hipGetLastError(); script_drop(gr); }
        else {
            sc.over_cap = n_pos < (1 << 20) ? n_pos : (1 << 20);
            if (gki_dev_malloc((void **)&sc.over_list, (size_t)sc.over_cap * 8) != hipSuccess) { (void)hipGetLastError(); sc.over_list = nullptr; sc.over_cap = 0; }
        }
    }
    int64_t total = 0; int word[2] = {0, 0};
    int rc = GKI_OK;
    hipError_t e1 = hipSuccess, e2 = hipSuccess, e3 = hipSuccess;
    for (;;) {
        const DeepArena da = gr->fwd_deep;
        e3 = hipMemset(d_err, 0, 8);
        if (da.cap > 0) {
            script_drop(gr);                // the slow path writes no script
            hipLaunchKernelGGL((k_forward<false, true>), dim3((unsigned)(da.lanes / 64)), dim3(64), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                               (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, cnt, (const int64_t *)nullptr, none, d_err, da,
                               (uint4 *)nullptr, (uint8_t *)nullptr, (int64_t *)nullptr, (int64_t)0);
        } else if (sc.entries)
            hipLaunchKernelGGL((k_forward<false, false, true>), dim3((unsigned)ceil_div(n_pos, GKI_FWD_BLOCK)), dim3(GKI_FWD_BLOCK), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                               (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, cnt, (const int64_t *)nullptr, none, d_err, da,
                               (uint4 *)sc.entries, sc.ncomp, sc.over_list, sc.over_cap);
        else
            hipLaunchKernelGGL((k_forward<false, false>), dim3((unsigned)ceil_div(n_pos, GKI_FWD_BLOCK)), dim3(GKI_FWD_BLOCK), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                               (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, cnt, (const int64_t *)nullptr, none, d_err, da,
                               (uint4 *)nullptr, (uint8_t *)nullptr, (int64_t *)nullptr, (int64_t)0);
        rc = hipGetLastError() == hipSuccess ? GKI_OK : gki_set_error(GKI_ERR_HIP, "k_forward launch failed");
        if (rc == GKI_OK) rc = gki_scan_u32_to_i64(cnt, n_pos, (int64_t *)d_rec_start, tmp, tmp_bytes, 0);
        e1 = hipMemcpy(&total, (const int64_t *)d_rec_start + n_pos, 8, hipMemcpyDeviceToHost);
        e2 = hipMemcpy(word, d_err, 8, hipMemcpyDeviceToHost);
        if (rc != GKI_OK || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) break;
        // bit 1: a stack of the walk was too short -- again with the deep variant, twice the levels each time round
        const int next_cap = da.cap == 0 ? 4 * FMAX : 2 * da.cap;
        if (!(word[0] & 2) || (word[0] & 4) || next_cap > GKI_MAX_DEEP_WINDOW_NODES) break;
        rc = deep_grow(gr, next_cap);
        if (rc != GKI_OK) break;
    }
    (void)gki_dev_free(cnt); (void)gki_dev_free(tmp); (void)gki_dev_free(d_err);
    const int herr = rc == GKI_OK && e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess ? gki_error_of_word(word[0]) : -1;
    if (herr != GKI_OK || !sc.entries) script_drop(gr);
    else {                                  // the emit call with these very arguments may expand the script
        sc.n_pos = n_pos; sc.overflow = word[1]; sc.nodes = d_nodes; sc.offsets = d_offsets; sc.follow = d_follow; sc.rec_start = d_rec_start;
        sc.k = k; sc.M = M; sc.one_node = one_node ? 1 : 0; sc.valid = 1;
    }
    if (rc != GKI_OK) return rc;
    HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3);
    if (herr == GKI_ERR_NOT_ONE_REF_SUCC)
        return gki_set_error(herr, "a path at the variant limit ends a node that does not have exactly one linear-ref "
                             "successor: the reference asserts here (kmer_finder.py:402)");
    if (herr) return gki_set_error(herr, (word[0] & 4) ? "the paths from one start position take more than %d descents to enumerate: too many paths"
                                   : "a forward k-window crosses more than %d nodes", (word[0] & 4) ? FW_BUDGET : GKI_MAX_DEEP_WINDOW_NODES - 2);
    *n_records = total;
    return GKI_OK;
}

int gki_forward_emit(gki_graph *gr, int k, int max_variant_nodes, int one_node, const void *d_follow, const void *d_nodes,
                     const void *d_offsets, int64_t n_pos, const void *d_rec_start, void *d_hashes, void *d_start_nodes, void *d_start_offsets,
                     void *d_nodes_out, void *d_af64) {
    if (n_pos <= 0) return GKI_OK;
    GKI_TRY(gki_check_graph_device(gr, "gki_forward_emit"));
    GKI_TRY(fwd_nodes_ready(gr));
    int *d_err = nullptr;
    HIP_TRY(gki_dev_malloc((void **)&d_err, 4));
    FwdOut out{(int64_t *)d_hashes, (int32_t *)d_start_nodes, (int16_t *)d_start_offsets, (int32_t *)d_nodes_out, (double *)d_af64};
    const int M = max_variant_nodes > 250 ? 250 : max_variant_nodes;
    FwdScript &sc = gr->fwd_script;
    int word = 0, rc = GKI_OK;
    hipError_t e = hipSuccess, e2 = hipSuccess, e3 = hipSuccess;
    // The arena state on the graph handle is whatever the LAST count call left (cap > 0: it needed the slow path), which
    // need not be this search's count call (count(A), count(B), emit(A); two finders on one graph).  The emit pass
    // therefore settles the depth itself: it reads its own error word and goes round again with the deep variant like the
    // count pass does (the word was never read and a too-short stack lost records silently).  Records already
    // written are written again with the same values.
    for (;;) {
        const DeepArena da = gr->fwd_deep;
        e3 = hipMemset(d_err, 0, 4);
        const bool scripted = sc.valid && da.cap == 0 && sc.n_pos == n_pos && sc.nodes == d_nodes && sc.offsets == d_offsets && sc.follow == d_follow &&
                              sc.rec_start == d_rec_start && sc.k == k && sc.M == M && sc.one_node == (one_node ? 1 : 0);
        if (da.cap > 0)
            hipLaunchKernelGGL((k_forward<true, true>), dim3((unsigned)(da.lanes / 64)), dim3(64), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                               (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, (uint32_t *)nullptr, (const int64_t *)d_rec_start, out, d_err, da,
                               (uint4 *)nullptr, (uint8_t *)nullptr, (int64_t *)nullptr, (int64_t)0);
        else if (scripted) {
            hipLaunchKernelGGL(k_forward_expand, dim3((unsigned)ceil_div(n_pos * FW_SLOTS, 256)), dim3(256), 0, 0, (const uint4 *)sc.entries, sc.ncomp,
                               (const int64_t *)d_rec_start, n_pos, out);
            // the start positions the script could not hold are walked as before: those on the count pass's list, or -- when
            // there were more than the list holds -- whichever the script marks, one lane per start position of the call
            if (sc.overflow > 0 && sc.overflow <= sc.over_cap)
                hipLaunchKernelGGL((k_forward<true, false>), dim3((unsigned)ceil_div(sc.overflow, GKI_FWD_BLOCK)), dim3(GKI_FWD_BLOCK), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                                   (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, (uint32_t *)nullptr, (const int64_t *)d_rec_start, out, d_err, da,
                                   (uint4 *)nullptr, sc.ncomp, sc.over_list, sc.overflow);
            else if (sc.overflow > 0)
                hipLaunchKernelGGL((k_forward<true, false>), dim3((unsigned)ceil_div(n_pos, GKI_FWD_BLOCK)), dim3(GKI_FWD_BLOCK), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                                   (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, (uint32_t *)nullptr, (const int64_t *)d_rec_start, out, d_err, da,
                                   (uint4 *)nullptr, sc.ncomp, (int64_t *)nullptr, (int64_t)0);
        } else
            hipLaunchKernelGGL((k_forward<true, false>), dim3((unsigned)ceil_div(n_pos, GKI_FWD_BLOCK)), dim3(GKI_FWD_BLOCK), 0, 0, gr->d, gr->fwd_nodes, k, M, one_node, (const uint8_t *)d_follow,
                               (const int32_t *)d_nodes, (const int32_t *)d_offsets, n_pos, (uint32_t *)nullptr, (const int64_t *)d_rec_start, out, d_err, da,
                               (uint4 *)nullptr, (uint8_t *)nullptr, (int64_t *)nullptr, (int64_t)0);
        e = hipGetLastError();
        e2 = hipMemcpy(&word, d_err, 4, hipMemcpyDeviceToHost);          // (synchronises)
        if (e != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) break;
        const int next_cap = da.cap == 0 ? 4 * FMAX : 2 * da.cap;
        if (!(word & 2) || (word & 4) || next_cap > GKI_MAX_DEEP_WINDOW_NODES) break;
        rc = deep_grow(gr, next_cap);
        if (rc != GKI_OK) break;
    }
    (void)gki_dev_free(d_err);
    script_drop(gr);                        // one emit per count: a second emit call walks
    deep_release(gr);                       // and the arena goes back to the pool
    if (rc != GKI_OK) return rc;
    HIP_TRY(e); HIP_TRY(e2); HIP_TRY(e3);
    const int herr = gki_error_of_word(word);
    if (herr) return gki_set_error(herr, "the emit pass of the early-stop search left records unwritten (error word %d)", word);
    return GKI_OK;
}

}  // extern "C"
