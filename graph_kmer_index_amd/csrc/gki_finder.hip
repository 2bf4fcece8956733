// DenseKmerFinder on MI355X: every (end position, backward k-window) of the graph, data-parallel.
//
// Formulation (SURVEY.md 8a'): the reference's DFS (kmer_finder.py:254-417)
// emits, for every base position e = (node, offset) and every backward path P of exactly k real
// bases ending at e whose window holds at most `max_variant_nodes` non-linear-ref nodes, one record
// per distinct node of P.  End positions are independent, so
//   * positions whose window lies inside their own node ("interior": offset >= bnd_len(node),
//     normally k-1) have exactly one window and one node: a pure streaming kernel
//     (k_emit_interior) -- ~90-100% of all records;
//   * the first bnd_len bases of a node ("boundary") share one predecessor tree: ONE lane walks it
//     once per node and, each time it steps onto a predecessor q with c context bases already
//     collected, completes the windows of every offset o with c < k-1-o <= c+size(q) from a single
//     62-bit context register (k_count_boundary / k_emit_boundary_one).
// Output slots come from a count pass + one exclusive scan over nodes, never from atomics, so the
// record order is deterministic: by end node; inside a node first the boundary windows in walk
// order (offset ascending inside one step, nodes of a window ascending), then the interior offsets.
#include "gki_common.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int MAXN = GKI_MAX_WINDOW_NODES;

struct NodeEmit {       // per-node constants of the interior kernels, 48 B (12 dwords)
    int64_t glo;        // global base index of the first interior position of this node in this run
    int64_t D;          // record index of interior position p is p + D
    int64_t E;          // position id of base p is p + E
    int32_t node;       // node id
    float af;           // allele frequency as float32 (flat_kmers.py:90)
    int32_t cnt;        // interior positions of this node in this run: [glo, glo + cnt)
    int32_t pad0;
    int64_t pad1;
};

struct FindArgs {
    int32_t k, M, one_node, has_lossy;
    int64_t node_begin, off_begin, node_end, off_end;
    int64_t n0, n1;          // nodes [n0, n1) are the only ones this run touches (per-shard cost, not per-graph)
    const int32_t *rank;     // NULL, or a topological rank per node: the run is rank_begin <= rank[n] <= rank_end
    int32_t rank_begin, rank_end;
    int32_t split, pad;      // output layout: 0 = by end node (boundary block, then interior run, per node);
                             // 1 = all interior records (by position) first, then all boundary records (by node)
    const uint16_t *nflags;  // NULL, or per node: GKI_NODE_* byte | history bound << 8 (general graphs, include/gki.h)
    const uint8_t *store;    // NULL, or only_store_nodes membership per node (kmer_finder.py:153); needs nflags
};

// does node n belong to the run (critical-path chunk / shard)?
__device__ __forceinline__ bool in_run(const FindArgs &a, int64_t n) {
    if (a.rank) { const int32_t r = a.rank[n]; return r >= a.rank_begin && r <= a.rank_end; }
    return n >= a.node_begin && n <= a.node_end;
}

struct OutFlat { uint64_t *hash; uint32_t *node; uint64_t *ref_offset; float *af; };
struct OutV2 { int64_t *hash; int32_t *start_node; int16_t *start_offset; int32_t *node; double *af; };
struct OutFlatAll { uint64_t *hash; uint32_t *node; uint64_t *ref_offset; float *af; };   // all four columns present
template <int FMT> struct OutSel;
template <> struct OutSel<0> { typedef OutFlat T; };
template <> struct OutSel<1> { typedef OutV2 T; };
template <> struct OutSel<2> { typedef OutFlatAll T; };

// The boundary kernels' records leave with the non-temporal hint (global_store ... nt): they are written once and read by
// a later kernel, while the walk between two expansions re-reads what L2 holds for it -- graph records, predecessor
// lists, the lanes' scratch.  With plain stores 7-22 GB of records stream through the 4 MB L2 of each XCD per launch and
// push those lines out: measured with the stores folded onto a few L2-resident lines the kernels' compute takes 2.1 ms
// (one-node) / 5.5 ms (all-nodes), and the real stores ADD their whole transfer time at the store ceiling instead of
// hiding under it.  Same-box A/B (profiles/r03_boundary_nt_stores_ab.txt): one-node 2.69 -> 2.43 ms, all-nodes
// 7.28 -> 6.61 ms, and the count pass that follows 0.587 -> 0.566 ms.  (The interior kernels, which re-read nothing,
// were 7 % slower with the hint: DESIGN.md 4.1.)  put<false> = plain stores;
// GKI_BND_PLAIN_STORES builds the plain-store form of everything for the A/B.
#ifdef GKI_BND_PLAIN_STORES
template <bool NT, class T> __device__ __forceinline__ void st(T *p, T v) { *p = v; }
#else
template <bool NT, class T> __device__ __forceinline__ void st(T *p, T v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
#endif
template <bool NT = true>
__device__ __forceinline__ void put(const OutFlat &o, int64_t idx, uint64_t h, int32_t node, int32_t end_node,
                                    int32_t end_off, int64_t pos_id, double af) {
    (void)end_node; (void)end_off;
    if (o.hash) st<NT>(&o.hash[idx], h);
    if (o.node) st<NT>(&o.node[idx], (uint32_t)node);
    if (o.ref_offset) st<NT>(&o.ref_offset[idx], (uint64_t)pos_id);
    if (o.af) st<NT>(&o.af[idx], (float)af);
}
// No null checks: on gfx950 a branch between two stores makes the compiler drain the first one (vmcnt counts
// stores, in order), which turns four back-to-back column stores into four serialized round trips.
template <bool NT = true>
__device__ __forceinline__ void put(const OutFlatAll &o, int64_t idx, uint64_t h, int32_t node, int32_t end_node,
                                    int32_t end_off, int64_t pos_id, double af) {
    (void)end_node; (void)end_off;
    st<NT>(&o.hash[idx], h);
    st<NT>(&o.node[idx], (uint32_t)node);
    st<NT>(&o.ref_offset[idx], (uint64_t)pos_id);
    st<NT>(&o.af[idx], (float)af);
}
template <bool NT = true>
__device__ __forceinline__ void put(const OutV2 &o, int64_t idx, uint64_t h, int32_t node, int32_t end_node,
                                    int32_t end_off, int64_t pos_id, double af) {
    (void)pos_id;
    if (o.hash) st<NT>(&o.hash[idx], (int64_t)h);
    if (o.start_node) st<NT>(&o.start_node[idx], end_node);
    if (o.start_offset) st<NT>(&o.start_offset[idx], (int16_t)end_off);
    if (o.node) st<NT>(&o.node[idx], node);
    if (o.af) st<NT>(&o.af[idx], af);
}

// SURVEY.md 8a' E1: a restart at a critical point (N, c) with 0 < c < k-1 is not rewound
// (kmer_finder.py:231-232), so no window contains both (N, c-1) and (N, c).
__device__ __forceinline__ int lossy_of(const uint16_t *__restrict__ lossy, int32_t node) {
    int c = lossy[node];
    return c == 0xFFFF ? -1 : c;
}

// Number of leading offsets of node n that are handled by the boundary walk.
__device__ __forceinline__ int32_t bnd_len_of(const DevGraph &g, const FindArgs &a, const uint16_t *__restrict__ lossy,
                                              int64_t n, int32_t size) {
    if (size <= 0 || !in_run(a, n)) return 0;
    int32_t reach = a.k - 1;                                          // offsets < k-1 look into predecessors
    if (a.has_lossy && lossy[n] != 0xFFFF) reach = lossy[n] + a.k - 1;      // E1 windows end up to c+k-2
    if (a.nflags) { if (a.nflags[n] & GKI_NODE_DEAD) reach = size; }  // general graphs: the search never enters the node
    else if (a.M < 1 && !g.is_ref[n]) reach = size;                   // variant node, limit 0: nothing admissible
    return size < reach ? size : reach;
}

// ------------------------------------------------------------------------------------ walk cache (LDS)
// A wave walks the predecessor trees of 64 consecutive nodes, and in a graph built along a genome the predecessors of a
// node have ids just below it.  One lane per node gathering 32-byte records from global memory costs the address unit
// a pass per lane and instruction (measured: the count pass took as long with every record an L1 hit).  So each wave
// first copies the records of nodes [base - WC_HALO, base + 64) and the predecessor lists of its own nodes into LDS with
// coalesced loads, and the walk reads them there; a node or list entry outside falls back to global memory.
constexpr int WC_HALO = 32;
constexpr int WC_REV = 192;
template <bool FLG>
struct WalkCacheT {
    uint4 rec[2 * (WC_HALO + 64)];
    int32_t rev[WC_REV];
    uint32_t flg[FLG ? WC_HALO + 64 : 1];   // per staged node: flag word (general graphs) | lossy-restart offset << 16 (runs with such points)
};
typedef WalkCacheT<true> WalkCache;         // WalkCacheT<false>: a kernel variant that reads neither (all-nodes mode, 384 B per wave)
struct WalkView {                // wave-uniform bounds of what the cache holds
    int64_t lo, hi;              // nodes
    int64_t r0, r1;              // rev_edges entries
};
template <class WC>
__device__ __forceinline__ WalkView stage_walk(const DevGraph &g, WC &wc, int64_t base, int lane,
                                               const uint16_t *__restrict__ nflags = nullptr,
                                               const uint16_t *__restrict__ lossy = nullptr) {
    WalkView v;
    v.lo = base - WC_HALO < 0 ? 0 : base - WC_HALO;
    v.hi = base + 64 > g.n_nodes ? g.n_nodes : base + 64;
    const uint4 *src = reinterpret_cast<const uint4 *>(g.walk) + 2 * v.lo;
    const int n16 = (int)(2 * (v.hi - v.lo));
    v.r0 = g.rev_start[base];
    const int64_t r_end = g.rev_start[v.hi];
    v.r1 = r_end - v.r0 > WC_REV ? v.r0 + WC_REV : r_end;
    const int n_rev = (int)(v.r1 - v.r0);
    // all loads first, then the LDS writes: one round trip to memory instead of one per 64 entries
    constexpr int RC = 2 * (WC_HALO + 64) / 64, VC = WC_REV / 64;
    uint4 t[RC];
    int32_t r[VC];
#pragma unroll
    for (int u = 0; u < RC; u++) { const int j = u * 64 + lane; t[u] = j < n16 ? src[j] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < VC; u++) { const int j = u * 64 + lane; r[u] = j < n_rev ? g.rev_edges[v.r0 + j] : 0; }
    // flags and lossy-restart offsets (0xFFFF = none) of the staged nodes share one LDS word: the walk asks for both at
    // every predecessor step, and lossy[q] from global memory was a dependent load per step (+0.15 ms count, +0.38 ms emit
    // on the 3 Gbp graph); a separate LDS array for it pushed the general emit variant into spilling (+0.6 ms)
    uint32_t fl[2] = {0xFFFF0000u, 0xFFFF0000u};
    if (nflags || lossy) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = u * 64 + lane;
            if (j < (int)(v.hi - v.lo)) fl[u] = (nflags ? (uint32_t)nflags[v.lo + j] : 0u) | ((lossy ? (uint32_t)lossy[v.lo + j] : 0xFFFFu) << 16);
        }
    }
#pragma unroll
    for (int u = 0; u < RC; u++) wc.rec[u * 64 + lane] = t[u];
#pragma unroll
    for (int u = 0; u < VC; u++) wc.rev[u * 64 + lane] = r[u];
    if (sizeof(wc.flg) > 4 && (nflags || lossy)) {
        wc.flg[lane] = fl[0];
        if (lane < WC_HALO) wc.flg[64 + lane] = fl[1];
    }
    __builtin_amdgcn_wave_barrier();
    return v;
}
// keep a value loaded from LDS a VALUE: without this the compiler merges "LDS load or global load" into one FLAT load
// of a selected address, which takes the global address path either way (seen in the ISA: flat_load_dwordx4, no ds_read)
__device__ __forceinline__ void opaque(uint4 &x) { asm volatile("" : "+v"(x.x), "+v"(x.y), "+v"(x.z), "+v"(x.w)); }
__device__ __forceinline__ void opaque(int32_t &x) { asm volatile("" : "+v"(x)); }

template <class WC>
__device__ __forceinline__ NodeWalk cached_walk(const DevGraph &g, const WC &wc, const WalkView &v, int64_t q) {
    const bool in = q >= v.lo && q < v.hi;
    const int64_t slot = in ? q - v.lo : 0;
    uint4 lo16 = wc.rec[2 * slot], hi16 = wc.rec[2 * slot + 1];
    opaque(lo16); opaque(hi16);
    if (!in) {
        const uint4 *src = reinterpret_cast<const uint4 *>(g.walk) + 2 * q;
        lo16 = src[0]; hi16 = src[1];
    }
    NodeWalk w;
    w.seq_start = (int64_t)(((uint64_t)lo16.y << 32) | lo16.x);
    w.tail = ((uint64_t)lo16.w << 32) | lo16.z;
    w.rev_begin = (int32_t)hi16.x;
    w.size = (int32_t)hi16.y;
    w.af = __uint_as_float(hi16.z);
    w.rev_cnt = (uint16_t)(hi16.w & 0xFFFFu);
    w.is_ref = (uint8_t)((hi16.w >> 16) & 0xFFu);
    w.pad = 0;
    return w;
}
// flag word (low 16 bits, 0 without nflags) and lossy-restart offset (high 16 bits, 0xFFFF = none) of node q
template <class WC>
__device__ __forceinline__ uint32_t cached_flag(const uint16_t *__restrict__ nflags, const uint16_t *__restrict__ lossy,
                                                const WC &wc, const WalkView &v, int64_t q) {
    const bool in = q >= v.lo && q < v.hi;
    int32_t f = (int32_t)wc.flg[in ? q - v.lo : 0];
    opaque(f);
    if (!in) f = (int32_t)((nflags ? (uint32_t)nflags[q] : 0u) | ((lossy ? (uint32_t)lossy[q] : 0xFFFFu) << 16));
    return (uint32_t)f;
}
__device__ __forceinline__ int lossy_in(uint32_t fw) { const int c = (int)(fw >> 16); return c == 0xFFFF ? -1 : c; }
template <class WC>
__device__ __forceinline__ int32_t cached_preds_next(const DevGraph &g, const WC &wc, const WalkView &v, int32_t *cur) {
    if (*cur < 0) { const int32_t q = ~*cur; *cur = 0; return q; }
    const int32_t i = (*cur)++;
    const bool in = i >= v.r0 && i < v.r1;
    int32_t q = wc.rev[in ? i - v.r0 : 0];
    opaque(q);
    if (!in) q = g.rev_edges[i];
    return q;
}

// Interior offsets [lo, hi) of node n in this run (bl = bnd_len_of).  only_store_nodes (kmer_finder.py:153) drops the
// records of a node outside the set -- except those of the bulk path, which ignores the filter (:370-374): offsets
// k+2 .. size-2 of a node longer than 2k+3 (:272).
__device__ __forceinline__ void interior_range(const FindArgs &a, int64_t n, int32_t size, int32_t bl, int64_t *lo_out, int64_t *hi_out) {
    int64_t lo = bl, hi = size;
    if (n == a.node_begin && a.off_begin > lo) lo = a.off_begin;
    if (n == a.node_end && a.off_end < hi) hi = a.off_end;
    if (a.store && !a.store[n]) {
        if (size > 2 * a.k + 3) { if (lo < a.k + 2) lo = a.k + 2; if (hi > size - 1) hi = size - 1; }
        else hi = lo;
    }
    *lo_out = lo; *hi_out = hi;
}

// ------------------------------------------------------------------------------------ boundary walk
// One lane per node: depth-first over predecessor lists with an explicit stack.  Stepping onto predecessor q
// (size s) with c context bases collected before it completes the windows of offsets o with c < k-1-o <= c+s.

// The walk used by the count pass and by the emit pass keeps the TOP of the stack in registers and only
// the levels below it in (scratch) arrays: PMC showed the first version, with the whole stack in scratch, writing
// 1.4 GB of spills in the count pass and ~5 GB in the emit pass of the 3 Gbp graph, and every step began with a
// dependent scratch load.  Node facts come from one aligned 32-byte NodeWalk record per visited node.
// (LocalStack / ArenaStack / DeepArena, the storage of the levels below the top: csrc/gki_common.h)
// byte offset of every per-level array inside one (level, lane) cell of the arena
enum { DA_BELOW = 0, DA_PATH = 40, DA_LVLA = 44, DA_CT = 46, DA_CM = 47, DA_HN = 49, DA_HCUR = 53, DA_HEND = 57, DA_HD = 61,
       DA_HSZ = 65, DA_HF = 69, DA_CELL = 70 };
template <bool DEEP>
struct LevelLoT {                // what a suspended level needs to resume
    int32_t cur, end;
    uint8_t cum;
    typename CountOf<DEEP>::T vc;
    typename CountOf<DEEP>::T a; // general graphs: variant count at the first non-free node of the path (0: none yet)
};

// ------------------------------------------------------------------------------------ general graphs
// The search (kmer_finder.py:383-417) reaches a window iff SOME path to it took every step into a node that is
// not free to enter (free: linear-ref(-dummy), or forced by only_follow_nodes :386-388) while the k bases before
// that node held fewer than max_variant_nodes variant nodes.  For the nodes of the window itself that is a count
// over the window: with v(i) = variant nodes from the end node back to level i, the step into the non-free node at
// level i saw v(first node) - v(i) variant nodes, largest for the non-free node nearest the end -- `a` below is v
// at that node, and a path is dropped as soon as v - a >= M.  What the history BEFORE the window's first node q adds
// depends on q (include/gki.h): nothing if q is T or SIMPLE (an all-linear-ref history exists and dominates);
// if q is NESTED, history_ok() enumerates the histories backwards until one is found that ends in a T / SIMPLE node,
// or in a node inside which every open constraint closes -- what is left then is "that node is entered at all", which
// gki_classify_nodes settled on the host (GKI_NODE_DEAD is exact).
constexpr int HMAX = 40;         // nodes of one enumerated history (k-1 one-base nodes + slack)
// The number of paths into one end node has no bound: a run of s insertion sites inside one window has 2^s of them when
// no variant limit cuts them (max_variant_nodes large) -- the reference does not return from such a graph either.  A lane
// of the count kernels that has taken this many steps for ONE end node gives up and the run is refused
// (GKI_ERR_WINDOW_TOO_DEEP, "too many paths"), so that no kernel runs for hours; the emit pass walks what the count pass
// walked.  Only descents below level STEP_BUDGET_FROM are counted: every path of an exponential family goes that deep, the
// walks of ordinary graphs (two to four levels) never do, and a decrement at every step cost the count pass of the 3 Gbp
// graph 10 % (0.57 -> 0.63 ms).
constexpr int STEP_BUDGET = 1 << 24, STEP_BUDGET_FROM = 8;

// The graph arrays the enumeration reads, BY VALUE: with `const DevGraph &` the kernel's DevGraph escaped to memory for
// this out-of-line call and every load through its pointers in the kernel itself became a FLAT load (address space
// lost) -- which waits on the LDS counter as well as on the memory counter, in kernels full of LDS traffic.
struct WalkSrc { const NodeWalk *walk; const int32_t *rev_edges; const int64_t *rev_start; };
template <class P, class A8, class A16, class A32, class AU8>
__device__ __forceinline__ bool history_ok_impl(const WalkSrc gs, const WalkCache &wc, const WalkView &wv,
                                                const uint16_t *__restrict__ nf, int k, int M, P path, int L, int *err,
                                                A8 ct, A16 cm, A32 hn, A32 hcur, A32 hend, A32 hd, A32 hsz, AU8 hf, int hmax) {
    DevGraph g = {};                       // only these three members are read below (helpers are inlined)
    g.walk = gs.walk; g.rev_edges = gs.rev_edges; g.rev_start = gs.rev_start;
    // open constraints of the window's nodes path[0..L] (end node .. q): the step into a non-free node y still sees
    // the history nodes within t bases of q's entry and tolerates fewer than m variant nodes among them
    int nc = 0;
    {
        int between = 0, c = 0;                  // bases / variant nodes of path[i+1..L]
        for (int i = L; i >= 0; i--) {
            const int32_t y = path[i];
            const uint16_t fy = (uint16_t)cached_flag(nf, nullptr, wc, wv, y);
            if (!(fy & (GKI_NODE_REF | GKI_NODE_FORCED))) {
                const int m = M - c;
                if (m <= 0) return false;
                const int t = k - between;
                if (t > 0) { ct[nc] = (int8_t)t; cm[nc] = (int16_t)(m > 30000 ? 30000 : m); nc++; }
            }
            const int sy = cached_walk(g, wc, wv, y).size;
            between = between + sy > k ? k : between + sy;
            c += (fy & GKI_NODE_REF) ? 0 : 1;
        }
    }
    // (hn, hcur, hend, hd, hsz, hf: history nodes, nearest first; hd = bases nearer than the node)
    {
        const NodeWalk wq = cached_walk(g, wc, wv, path[L]);
        int32_t c0, e0;
        preds_begin(g, wq, path[L], &c0, &e0);
        hcur[0] = c0; hend[0] = e0;
    }
    int h = 0;
    for (int budget = 1 << 16;; budget--) {
        // every lane's enumeration ends: a graph with more than 65 536 history steps behind one window is refused
        // (GKI_ERR_WINDOW_TOO_DEEP) rather than walked for minutes
        if (budget == 0) { gki_raise_budget(err); return false; }
        if (hcur[h] >= hend[h]) { if (h == 0) return false; h--; continue; }
        int32_t cur_h = hcur[h];
        const int32_t p = cached_preds_next(g, wc, wv, &cur_h);
        hcur[h] = cur_h;
        const uint8_t fp = (uint8_t)cached_flag(nf, nullptr, wc, wv, p);
        if (fp & GKI_NODE_DEAD) continue;
        const int32_t child = h == 0 ? path[L] : hn[h - 1];
        if ((fp & GKI_NODE_HFS) && !(nf[child] & GKI_NODE_FORCED)) continue;      // edge removed by a forced sibling
        const NodeWalk wp = cached_walk(g, wc, wv, p);
        hn[h] = p; hf[h] = fp;
        hd[h] = h == 0 ? 0 : hd[h - 1] + hsz[h - 1];
        hsz[h] = wp.size > (1 << 20) ? (1 << 20) : wp.size;
        bool ok = true;
        if (!(fp & GKI_NODE_REF)) {              // only a variant node changes a count
            for (int j = 0; j < nc && ok; j++) {
                int cnt = 0;
                for (int x = 0; x <= h; x++) if (!(hf[x] & GKI_NODE_REF) && hd[x] < ct[j]) cnt++;
                if (cnt >= cm[j]) ok = false;
            }
            for (int l = 0; l < h && ok; l++) {  // steps into the non-free nodes of the history itself
                if (hf[l] & (GKI_NODE_REF | GKI_NODE_FORCED)) continue;
                int cnt = 0;
                for (int x = l + 1; x <= h; x++) if (!(hf[x] & GKI_NODE_REF) && hd[x] - hd[l + 1] < k) cnt++;
                if (cnt >= M) ok = false;
            }
        }
        if (!(fp & (GKI_NODE_REF | GKI_NODE_FORCED)) && M < 1) ok = false;
        if (!ok) continue;
        if (fp & (GKI_NODE_T | GKI_NODE_SIMPLE)) return true;     // beyond: an all-linear-ref history, no further variant node
        {
            const int end = hd[h] + hsz[h];                       // does every open constraint close inside p?
            bool closed = true;
            for (int j = 0; j < nc && closed; j++) if (ct[j] > end) closed = false;
            for (int l = 0; l < h && closed; l++)
                if (!(hf[l] & (GKI_NODE_REF | GKI_NODE_FORCED)) && end - hd[l + 1] < k) closed = false;
            if (closed) return true;                              // p is not DEAD: some admissible history enters it
        }
        if (!(fp & GKI_NODE_NESTED)) continue;
        if (h + 1 >= hmax) { gki_raise(err, GKI_ERR_WINDOW_TOO_DEEP); return false; }
        h++;
        {
            int32_t c0, e0;
            preds_begin(g, wp, p, &c0, &e0);
            hcur[h] = c0; hend[h] = e0;
        }
    }
}
// (kernel variants without the flag words never run the general branch that calls history_ok; this overload only
// lets that branch compile)
template <class P>
__device__ __forceinline__ bool history_ok(const WalkSrc, const WalkCacheT<false> &, const WalkView &, const uint16_t *, int, int,
                                           P, int, int *, const DeepArena &, int64_t) { return false; }
__device__ __noinline__ bool history_ok(const WalkSrc gs, const WalkCache &wc, const WalkView &wv,
                                        const uint16_t *__restrict__ nf, int k, int M, const int32_t *path, int L, int *err,
                                        const DeepArena &, int64_t) {
    int8_t ct[MAXN];
    int16_t cm[MAXN];
    int32_t hn[HMAX], hcur[HMAX], hend[HMAX], hd[HMAX], hsz[HMAX];
    uint8_t hf[HMAX];
    return history_ok_impl(gs, wc, wv, nf, k, M, path, L, err, ct, cm, hn, hcur, hend, hd, hsz, hf, HMAX);
}
__device__ __noinline__ bool history_ok(const WalkSrc gs, const WalkCache &wc, const WalkView &wv,
                                        const uint16_t *__restrict__ nf, int k, int M, ArenaStack<int32_t> path, int L, int *err,
                                        const DeepArena &da, int64_t lane_global) {
    return history_ok_impl(gs, wc, wv, nf, k, M, path, L, err, arena_stack<int8_t>(da, DA_CT, lane_global),
                           arena_stack<int16_t>(da, DA_CM, lane_global), arena_stack<int32_t>(da, DA_HN, lane_global),
                           arena_stack<int32_t>(da, DA_HCUR, lane_global), arena_stack<int32_t>(da, DA_HEND, lane_global),
                           arena_stack<int32_t>(da, DA_HD, lane_global), arena_stack<int32_t>(da, DA_HSZ, lane_global),
                           arena_stack<uint8_t>(da, DA_HF, lane_global), da.cap);
}

// only_store_nodes (kmer_finder.py:145-154): records a window with nodes path[0..n_path) yields -- its smallest node
// if that is in the set (only_save_one_node_per_kmer), else one per node in the set.
template <class P>
__device__ __forceinline__ int stored_nodes(const uint8_t *__restrict__ store, P path, int n_path, bool one_node) {
    if (one_node) {
        int32_t mn = path[0];
        for (int i = 1; i < n_path; i++) mn = path[i] < mn ? path[i] : mn;
        return store[mn] ? 1 : 0;
    }
    int c = 0;
    for (int i = 0; i < n_path; i++) c += store[path[i]] ? 1 : 0;
    return c;
}

// The walk is a chain of dependent LDS / memory reads: its speed is the number of resident waves.  The general variant
// needs 100 VGPRs unconstrained (4 waves per SIMD; the others 53-55 = 8) -- held to 64 it spills ~30 values around the
// history_ok call and is still faster at every step: 1.27 / 1.11 / 0.99 / 0.95 / 0.94 ms at 4 / 5 / 6 / 7 / 8 waves on the
// 3 Gbp SNP graph, 2.19 / 1.88 / 1.70 / 1.66 / 1.64 ms with 20 % nested sites (tools/exp/ab_libs.sh, same box).
#ifndef GKI_CNT_WPB
#define GKI_CNT_WPB 1
#endif
constexpr int CNT_WPB = GKI_CNT_WPB;      // waves per workgroup of the count kernel (independent waves: see BND_WPB)
template <bool HAS_LOSSY, bool GEN, bool DEEP = false>
__global__ __launch_bounds__(64 * CNT_WPB, (GEN && !DEEP) ? 8 : 1) void k_count_boundary(DevGraph g, FindArgs a, const uint16_t *__restrict__ lossy,
                                                        uint32_t *__restrict__ bcount, uint32_t *__restrict__ total,
                                                        int *__restrict__ err, DeepArena da) {
    __shared__ WalkCache s_wc[CNT_WPB];
    typedef LevelLoT<DEEP> LevelLo;
    typedef typename CountOf<DEEP>::T cnt_t;
    const int64_t lane_global = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (deep variant: the lane's column of the arena)
    const int cap = DEEP ? da.cap : MAXN;
    typename StackOf<LevelLo, MAXN, DEEP>::type below;
    typename StackOf<int32_t, GEN ? MAXN : 1, DEEP>::type path;   // general graphs: the node of every level (level 0 = the end node)
    bind(below, da, DA_BELOW, lane_global);
    bind(path, da, DA_PATH, lane_global);
    LevelLo below0 = {0, 0, 0, 0, 0};  // the first suspended level stays in registers (SNP/indel graphs never go deeper)
    // levels 1 and 2 of `path` live in registers and reach scratch only right before the rare calls that read the array
    // (history_ok, stored_nodes): a scratch store at every step of the walk sits in front of the next load (in-order vmcnt)
    int32_t pr1 = 0, pr2 = 0;
    const int k = a.k;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WalkCache &wc = s_wc[wib];
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = a.n0 + (int64_t)blockIdx.x * blockDim.x + wib * 64; base < a.n1; base += stride) {
        const WalkView wv = stage_walk(g, wc, base, lane, GEN ? a.nflags : nullptr, HAS_LOSSY ? lossy : nullptr);
        const int64_t n = base + lane;
        if (n >= a.n1) continue;
        const NodeWalk wn = cached_walk(g, wc, wv, n);
        const int32_t size = wn.size;
        const uint16_t fn = GEN ? (uint16_t)cached_flag(a.nflags, HAS_LOSSY ? lossy : nullptr, wc, wv, n) : (uint16_t)0;
        const bool reach_n = !(GEN && (fn & GKI_NODE_DEAD));
        const int32_t bl = bnd_len_of(g, a, lossy, n, size);
        uint32_t count = 0;
        const int o_lo = (n == a.node_begin) ? (int)(a.off_begin < bl ? a.off_begin : bl) : 0;
        const int o_hi = (n == a.node_end) ? (int)(a.off_end < bl ? a.off_end : bl) : bl;
        const int v0 = GEN ? ((fn & GKI_NODE_REF) ? 0 : 1) : (wn.is_ref ? 0 : 1);
        const bool nonfree0 = GEN ? !(fn & (GKI_NODE_REF | GKI_NODE_FORCED)) : v0 != 0;
        // kmer_finder.py:402: at the end of a node without exactly one linear-ref successor no reachable window may be
        // at the limit.  The windows in question end at the node's last base (oc), for an empty node "before offset 0".
        bool chk = GEN && (fn & GKI_NODE_CHECK) && reach_n && in_run(a, n);
        const int oc = size - 1;
        if (chk && size >= k) { if (v0 >= a.M) gki_raise(err, GKI_ERR_NOT_ONE_REF_SUCC); chk = false; }
        const bool chk_empty = chk && size == 0;
        if (reach_n && ((bl > 0 && o_lo < o_hi) || chk_empty) && !(nonfree0 && a.M < 1)) {
            const int cn = HAS_LOSSY ? lossy_of(lossy, (int32_t)n) : -1;
            for (int o = o_lo > k - 1 ? o_lo : k - 1; o < o_hi; o++) {
                if (HAS_LOSSY && cn >= 0 && o + 1 - k <= cn - 1 && cn <= o) continue;
                if (GEN && a.store && !a.store[n]) continue;
                count += 1;
            }
            int hi = o_hi < k - 1 ? o_hi : k - 1;
            if (HAS_LOSSY && cn >= 0 && cn < hi) hi = cn;
            const int w_lo = chk_empty ? -1 : o_lo;       // offset -1: the k bases before the node
            if (chk_empty) hi = 0;
            if (GEN && chk && wn.rev_cnt == 0 && v0 >= a.M) gki_raise(err, GKI_ERR_NOT_ONE_REF_SUCC);   // a root shorter than k
            if (w_lo < hi) {
                int32_t t_cur, t_end;
                preds_begin(g, wn, n, &t_cur, &t_end);
                int t_cum = 0, t_vc = v0, t_a = nonfree0 ? v0 : 0;
                int L = 1;
                if (GEN) path[0] = (int32_t)n;
                int steps_left = STEP_BUDGET;
                while (L > 0) {
                    if (t_cur >= t_end) {
                        L--;
                        if (L > 0) { LevelLo b = below0; if (L != 1) b = below[L - 1]; t_cur = b.cur; t_end = b.end; t_cum = b.cum; t_vc = b.vc; t_a = b.a; }
                        continue;
                    }
                    const int32_t q = cached_preds_next(g, wc, wv, &t_cur);
                    const NodeWalk wq = cached_walk(g, wc, wv, q);
                    const uint32_t fwq = (GEN || HAS_LOSSY) ? cached_flag(GEN ? a.nflags : nullptr, HAS_LOSSY ? lossy : nullptr, wc, wv, q) : 0u;
                    const uint16_t fq = (uint16_t)fwq;
                    if (GEN) {
                        if (fq & GKI_NODE_DEAD) continue;
                        if ((fq & GKI_NODE_HFS) && !(a.nflags[L == 1 ? (int32_t)n : L == 2 ? pr1 : L == 3 ? pr2 : path[L - 1]] & GKI_NODE_FORCED)) continue;
                    }
                    const int vq = t_vc + (GEN ? ((fq & GKI_NODE_REF) ? 0 : 1) : (wq.is_ref ? 0 : 1));
                    int aq = 0;
                    if (!GEN) {
                        if (vq > a.M) continue;                      // kmer_finder.py:391-403 in order-free form
                    } else {
                        aq = t_a ? t_a : ((fq & (GKI_NODE_REF | GKI_NODE_FORCED)) ? 0 : vq);
                        if (aq && vq - aq >= a.M) continue;
                    }
                    if (L >= cap - 1) { gki_raise(err, GKI_ERR_WINDOW_TOO_DEEP); continue; }
                    if (GEN) { if (L == 1) pr1 = q; else if (L == 2) pr2 = q; else path[L] = q; }
                    const int s = wq.size, c = t_cum;
                    bool deeper;
                    int new_cum;
                    if (s == 0) {                                    // empty node: in the node set, adds no base
                        deeper = true; new_cum = c;
                    } else {
                        int from = k - 1 - c - s; if (from < w_lo) from = w_lo;
                        int to = k - 1 - c; if (to > hi) to = hi;
                        const int cq = HAS_LOSSY ? lossy_in(fwq) : -1;
                        if (HAS_LOSSY && cq >= 0) {
                            const int min_ok = k - 1 - c - s + cq; if (from < min_ok) from = min_ok;
                            // the search restarted at (q, cq) with no history: at an end position whose window would
                            // reach before that point it holds the shorter window, which still decides :402
                            if (GEN && chk && oc >= w_lo && oc < min_ok && vq >= a.M) gki_raise(err, GKI_ERR_NOT_ONE_REF_SUCC);
                        }
                        if (GEN && from < to) {
                            bool ok = true;                          // a history before the window's first node?
                            if (!(fq & (GKI_NODE_T | GKI_NODE_SIMPLE)))
                                ok = (fq & GKI_NODE_NESTED) ? (vq + (int)(fq >> 8) < a.M || (path[1] = pr1, path[2] = pr2, history_ok(WalkSrc{g.walk, g.rev_edges, g.rev_start}, wc, wv, a.nflags, k, a.M, raw(path), L, err, da, lane_global))) : false;
                            if (ok && chk && from <= oc && oc < to && vq >= a.M) gki_raise(err, GKI_ERR_NOT_ONE_REF_SUCC);
                            if (!ok) to = from;
                        }
                        if (GEN && from < 0 && from < to) from = 0;      // offset -1 has no record
                        if (from < to) {
                            uint32_t per_window = a.one_node ? 1u : (uint32_t)(L + 1);
                            if (GEN && a.store) { path[1] = pr1; path[2] = pr2; per_window = (uint32_t)stored_nodes(a.store, raw(path), L + 1, a.one_node != 0); }
                            count += (uint32_t)(to - from) * per_window;
                        }
                        // the graph ends before the window of oc is complete (graph start): the search saw what there is
                        if (GEN && chk && wq.rev_cnt == 0 && c + s < k - 1 - oc && vq >= a.M) gki_raise(err, GKI_ERR_NOT_ONE_REF_SUCC);
                        deeper = (k - 1 - c - s > w_lo) && !(HAS_LOSSY && cq >= 0);
                        new_cum = c + s;
                    }
                    if (deeper) {
                        LevelLo b; b.cur = t_cur; b.end = t_end; b.cum = (uint8_t)t_cum; b.vc = (cnt_t)t_vc; b.a = (cnt_t)t_a;
                        if (L == 1) below0 = b; else below[L - 1] = b;
                        preds_begin(g, wq, q, &t_cur, &t_end);
                        t_cum = new_cum; t_vc = vq; t_a = aq;
                        L++;
                        // (out of budget: the walk is wound up through its ordinary exit -- a second way out of the loop
                        // cost the product kernel 13 %)
                        if (L > STEP_BUDGET_FROM && --steps_left < 0) { gki_raise_budget(err); L = 1; t_cur = t_end = 0; }
                    }
                }
            }
        }
        uint32_t ic = 0;                              // interior offsets of this node in this run
        if (size > 0 && in_run(a, n)) {
            int64_t lo, hi2;
            interior_range(a, n, size, bl, &lo, &hi2);
            ic = hi2 > lo ? (uint32_t)(hi2 - lo) : 0u;
        }
        bcount[n] = count;
        total[n] = a.split ? ic : count + ic;
    }
}

// Emit pass (ALL = false: only_save_one_node_per_kmer, the CLI `index` configuration; ALL = true: one record per
// distinct window node, the constructor's default): walk and write are separated inside the wave.
// Phase A -- the lanes walk their nodes (loads only) and park every finished step in the wave's LDS queue: per step
// {context register, first record slot, allele-frequency minimum, smallest node, offsets, owning lane [, node list]},
// per lane {own bases, position id, node}.  A step whose window spans more than NLQ nodes does not fit the queue and is
// written by the whole wave at once (all-nodes mode).
// Phase B (expand_queue) -- when the queue fills up, a node group ends or the walk is over, the wave writes the queue
// in output order with one lane per WINDOW: positions from a wave scan, the step of window r from a bitmap + popcount,
// in all-nodes mode the step's nodes sorted by a fixed network first; the column stores stream with no load in between.
// Keeping the stores out of the walk matters on gfx950: loads and stores retire in order through one counter, so a
// store inside the walk stalls the next dependent load.
// EVQ: 128 steps per queue; 96 would save 4.5 KB of LDS per workgroup but is ~7 % slower at equal occupancy (fewer steps
// per flush), which is why the LDS diet for a fifth workgroup per CU was dropped.
#ifndef GKI_EVQ
#define GKI_EVQ 128
#endif
constexpr int EVQ = GKI_EVQ;        // step descriptors per wave queue (a walk round adds at most 64)
constexpr int MW = EVQ * 32 / 64;   // words of the window bitmap (a step holds at most 31 windows)
static_assert(EVQ > 64 && EVQ <= 128 && EVQ % 2 == 0 && MW <= 64, "two steps per lane, one bitmap word per lane");

constexpr int NLQ = 5;              // all-nodes mode: node lists of up to NLQ nodes travel through the queue (two SNPs inside
                                    // one window: segment, allele, segment, allele, segment); longer lists are written by the wave

template <int FMT> struct MafOf { typedef float T; };        // float32 rounding is monotonic: min, then round == round, then min
template <> struct MafOf<1> { typedef double T; };           // get_flat_kmers(v="2") keeps float64 (kmer_finder.py:58)

template <bool ALL> struct IdxOf { typedef int64_t T; };
template <> struct IdxOf<true> { typedef uint32_t T; };
#ifdef GKI_EMIT_SLIM
constexpr bool EMIT_SLIM = true;     // one-node mode too: per-lane facts shuffled, 32-bit slots, no flag words -> 32 KB, 5 workgroups per CU
#else
constexpr bool EMIT_SLIM = false;
#endif

// One-node mode packs what a record's lane reads about its step and about the step's node into three aligned structs --
// one ds_read_b128 for {context, first slot}, one ds_read_b64 for {allele-frequency minimum, smallest node}, one
// ds_read_b128 for the node's {own bases, position id} -- and every mode packs the four byte fields of a step into one
// word: a record's lane issues 8 LDS reads where it issued 13.
template <int FMT> struct StepA { uint64_t ctx; int64_t idx; };
template <int FMT> struct StepB { typename MafOf<FMT>::T maf; int32_t mn; };
struct LaneC { uint64_t own; int64_t pos0; };
__device__ __forceinline__ uint32_t step_word(int from, int cnt, int ln, int seq) {
    return (uint32_t)from | ((uint32_t)cnt << 8) | ((uint32_t)ln << 16) | ((uint32_t)seq << 24);
}
__device__ __forceinline__ int sw_from(uint32_t w) { return (int)(w & 0xFFu); }
__device__ __forceinline__ int sw_cnt(uint32_t w) { return (int)((w >> 8) & 0xFFu); }
__device__ __forceinline__ int sw_ln(uint32_t w) { return (int)((w >> 16) & 0xFFu); }
__device__ __forceinline__ int sw_seq(uint32_t w) { return (int)(w >> 24); }

// one-node mode: per node of the group in progress (lane l walks node base + l) its first k-1 bases and the position id
// of (node, 0); per queued step the context register and first record slot, the allele-frequency minimum and the
// smallest node.  (All-nodes mode leaves the per-node facts in the walking lane's registers and shuffles them at
// expansion time; with 32-bit record slots, node lists of five and no `mn`: 40.9 KB per workgroup, a fourth per CU.)
template <int FMT, bool PACKED> struct EvPacked {};
template <int FMT> struct EvPacked<FMT, true> {
    alignas(16) LaneC c[64];
    alignas(16) StepA<FMT> a[EVQ];
    alignas(8) StepB<FMT> b[EVQ];
};

template <int FMT, bool ALL>
struct EvQueue : EvPacked<FMT, !ALL && !EMIT_SLIM> {
    static constexpr bool PACKED = !ALL && !EMIT_SLIM;
    // the other modes: per queued step
    uint64_t ctx[PACKED ? 1 : EVQ];
    typename IdxOf<true>::T idx[PACKED ? 1 : EVQ];  // first record slot of the step, relative to the node group's first
    typename MafOf<FMT>::T maf[PACKED ? 1 : EVQ];
    int32_t mn[(ALL || PACKED) ? 1 : EVQ];
    uint32_t small[EVQ];         // step_word: first offset, number of windows, the lane (node) that queued the step and its
                                 // number among that lane's queued steps
    uint8_t order[EVQ];          // slot of the step at position p of the output order
    alignas(8) uint8_t lbase[64];   // output position of a lane's first queued step (all-nodes mode: reused by the expansion as
                                    // eight 64-bit words of record-start marks once the output order is known)
    uint8_t nl[ALL ? EVQ : 1];                 // all-nodes mode: number of distinct window nodes of the step ...
    int32_t nodes[ALL ? EVQ : 1][NLQ];         // ... and the nodes, ascending (np.unique, kmer_finder.py:134)
    // lane-per-window expansion: the queue's windows (one per step and offset) numbered 0 .. T-1 in output order
    uint16_t pre[EVQ];               // first window number of the step at sorted position s
    uint64_t marks[EVQ * 32 / 64];   // bit r set: a step starts at window r (T <= EVQ * 32)
    uint16_t wrank[EVQ * 32 / 64];   // steps that start before word w of marks
};

#ifdef GKI_TUNING
__device__ int g_dbg_skip_expand = 0;     // tools/exp builds only (make tuning): 1 = phase A alone (the walk, no expansion),
                                          // 3 = phase A alone and no node lists built (all-nodes mode; nothing reads them)
                                          // 4 = phase A alone and windows over more than NLQ nodes dropped (output incomplete)
                                          // 5 = walk and expansion, the expansion's stores folded onto 16 384 record slots: the same
                                          //     instructions, but the lines stay in L2 and nothing drains to memory (output wrong)
#define GKI_DBG_SKIP_EXPAND_IS(v) (g_dbg_skip_expand == (v))
#define GKI_DBG_SLOT(i) (g_dbg_skip_expand == 5 ? ((i) & 0x3FFF) : (i))
#else
#define GKI_DBG_SKIP_EXPAND_IS(v) false
#define GKI_DBG_SLOT(i) (i)
#endif

// Expansion writes the queued steps in OUTPUT order.  Steps arrive in walk order, i.e. interleaved across the 64
// nodes of the wave; written that way every cache line of the wave's output block is touched several microseconds
// apart, and at ~5 TB/s of writes a 4-MB L2 turns over in ~6 us, so lines left half-written were flushed twice
// (PMC: 11.4 GB written for 7.4 GB of records).  Output order is (node, order of queueing): record slots grow with
// the node id and, inside a node, with every step queued.  The queue never holds steps of two node groups (it is
// flushed when a group is done), so the position of a step is "steps queued by lower lanes" + its number within its
// lane: one wave scan and a scatter -- the first version sorted packed keys with a 28-stage bitonic network.
template <int FMT, bool ALL>
__device__ __forceinline__ void expand_queue(EvQueue<FMT, ALL> &q, int n_ev, int my_cnt, typename OutSel<FMT>::T out, int k,
                                             uint64_t kmask, int lane, uint64_t own_reg = 0, int64_t pos0_reg = 0,
                                             int32_t n_reg = 0, int64_t idx_base = 0) {
    if (GKI_DBG_SKIP_EXPAND_IS(1) || GKI_DBG_SKIP_EXPAND_IS(3) || GKI_DBG_SKIP_EXPAND_IS(4)) return;
    {
        const int ps = gki_wave_incl_sum(my_cnt);
        q.lbase[lane] = (uint8_t)(ps - my_cnt);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int e = b * 64 + lane;
            if (e < n_ev) { const uint32_t sm = q.small[e]; q.order[(int)q.lbase[sw_ln(sm)] + sw_seq(sm)] = (uint8_t)e; }
            if (ALL && e < n_ev) {
                // the step's nodes ascending (records of a window are written per distinct node, ascending): a fixed
                // 9-comparator network on registers, every lane the same instructions; slots >= nl count as +inf
                static_assert(NLQ == 5, "sorting network for five");
                const int nl = (int)q.nl[e];
                int32_t v[NLQ];
#pragma unroll
                for (int t = 0; t < NLQ; t++) v[t] = t < nl ? q.nodes[e][t] : INT_MAX;
#define GKI_CSWAP(i, j) { const int32_t lo_ = v[i] < v[j] ? v[i] : v[j], hi_ = v[i] < v[j] ? v[j] : v[i]; v[i] = lo_; v[j] = hi_; }
                GKI_CSWAP(0, 1) GKI_CSWAP(3, 4)
                GKI_CSWAP(2, 4)
                GKI_CSWAP(2, 3) GKI_CSWAP(1, 4)
                GKI_CSWAP(0, 3)
                GKI_CSWAP(0, 2) GKI_CSWAP(1, 3)
                GKI_CSWAP(1, 2)
#undef GKI_CSWAP
#pragma unroll
                for (int t = 0; t < NLQ; t++) q.nodes[e][t] = v[t];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    int64_t mn_idx;                                      // smallest record slot of the queue
    if constexpr (EvQueue<FMT, ALL>::PACKED) mn_idx = q.a[q.order[0]].idx; else mn_idx = (int64_t)q.idx[q.order[0]];
    // One lane per WINDOW (= per record in one-node mode).  Half of the steps of a SNP graph hold a single window, so
    // "one step per half-wave" left two thirds of the lanes idle and the address unit paid for 6.4e7 store
    // instructions on the 3 Gbp graph (SQ_INSTS_VMEM_WR).  Here the T windows of the queue are numbered in output
    // order: an exclusive prefix of the sorted steps' sizes, a bitmap with a bit at every step's first window, and the
    // step of window r is a popcount over that bitmap -- no search.  Trips are aligned to 16 records of the output so
    // that a store instruction covers whole cache lines of every column wherever the steps are contiguous.  In
    // all-nodes mode a window has one record per distinct node: the lane writes them in turn (neighbouring lanes then
    // write `nl` records apart and the following turns fill the gaps while the lines are still in L2).
    const int c0 = lane < n_ev ? sw_cnt(q.small[q.order[lane]]) : 0;
    const int c1 = lane + 64 < n_ev ? sw_cnt(q.small[q.order[lane + 64]]) : 0;
    const int s0 = gki_wave_incl_sum(c0), s1 = gki_wave_incl_sum(c1);
    // (the totals: read into a scalar register in all-nodes mode, broadcast by shuffle otherwise -- measured, same box:
    // the one-node kernel takes 2.51 ms with the scalar form and 2.22 ms with the shuffle, the all-nodes kernel 6.65 ms
    // against 6.84 ms; profiles/r03_wave_scan_ab.txt)
    const int tot0 = ALL ? gki_lane_value(s0, 63) : __shfl(s0, 63, 64);
    const int T = tot0 + (ALL ? gki_lane_value(s1, 63) : __shfl(s1, 63, 64));
    const int ex0 = s0 - c0, ex1 = tot0 + s1 - c1;
    q.pre[lane] = (uint16_t)ex0;
    if (lane + 64 < EVQ) q.pre[lane + 64] = (uint16_t)ex1;
    if (lane < MW) q.marks[lane] = 0ull;
    __builtin_amdgcn_wave_barrier();
    unsigned int *marks32 = reinterpret_cast<unsigned int *>(q.marks);
    if (c0) atomicOr(&marks32[ex0 >> 5], 1u << (ex0 & 31));
    if (c1) atomicOr(&marks32[ex1 >> 5], 1u << (ex1 & 31));
    __builtin_amdgcn_wave_barrier();
    const int pc = lane < MW ? __popcll(q.marks[lane]) : 0;
    const int ps = gki_wave_incl_sum(pc);
    if (lane < MW) q.wrank[lane] = (uint16_t)(ps - pc);
    __builtin_amdgcn_wave_barrier();
    const int shift = ALL ? 0 : (int)((mn_idx + idx_base) & 15);
    if constexpr (!ALL && EMIT_SLIM) {
        for (int r0 = -shift; r0 < T; r0 += 64) {
            const int r = r0 + lane;
            const bool valid = r >= 0 && r < T;
            int e = 0, j = 0, o = 0, ln = 0;
            if (valid) {
                const int w = r >> 6;
                const int sp = (int)q.wrank[w] + __popcll(q.marks[w] & ((2ull << (r & 63)) - 1ull)) - 1;
                e = (int)q.order[sp];
                j = r - (int)q.pre[sp];
                const uint32_t sm = q.small[e];
                o = sw_from(sm) + j; ln = sw_ln(sm);
            }
            const uint64_t own_v = (uint64_t)__shfl((unsigned long long)own_reg, ln, 64);
            const int64_t pos0_v = (int64_t)__shfl((long long)pos0_reg, ln, 64);
            const int32_t n_v = __shfl(n_reg, ln, 64);
            if (valid) {
                const uint64_t h = ((q.ctx[e] >> (2 * o)) | (own_v << (2 * (k - 1 - o)))) & kmask;
                put(out, idx_base + (int64_t)q.idx[e] + j, h, q.mn[e], n_v, o, pos0_v + o, (double)q.maf[e]);
            }
        }
    } else if constexpr (!ALL) {
        for (int r0 = -shift; r0 < T; r0 += 64) {
            const int r = r0 + lane;
            if (r >= 0 && r < T) {
                const int w = r >> 6;
                const int sp = (int)q.wrank[w] + __popcll(q.marks[w] & ((2ull << (r & 63)) - 1ull)) - 1;
                const int e = (int)q.order[sp];
                const int j = r - (int)q.pre[sp];
                const uint32_t sm = q.small[e];
                const int o = sw_from(sm) + j, ln = sw_ln(sm);
                const StepA<FMT> sa = q.a[e];
                const StepB<FMT> sb = q.b[e];
                const LaneC lc = q.c[ln];
                const uint64_t h = ((sa.ctx >> (2 * o)) | (lc.own << (2 * (k - 1 - o)))) & kmask;
                put(out, GKI_DBG_SLOT(sa.idx + j), h, sb.mn, n_reg - lane + ln, o, lc.pos0 + o, (double)sb.maf);
            }
        }
    } else {
        // all-nodes mode: one record per distinct node of the window, nodes ascending.  What belongs to the walking lane
        // (its node's own bases, position id, node id) stays in that lane's registers and is fetched by shuffle -- with
        // every lane taking part, a lane outside the trip asks for lane 0.
        for (int r0 = 0; r0 < T; r0 += 64) {
            const int r = r0 + lane;
            const bool valid = r < T;
            int e = 0, j = 0, o = 0, ln = 0;
            if (valid) {
                const int w = r >> 6;
                const int sp = (int)q.wrank[w] + __popcll(q.marks[w] & ((2ull << (r & 63)) - 1ull)) - 1;
                e = (int)q.order[sp];
                j = r - (int)q.pre[sp];
                const uint32_t sm = q.small[e];
                o = sw_from(sm) + j; ln = sw_ln(sm);
            }
            const uint64_t own_v = (uint64_t)__shfl((unsigned long long)own_reg, ln, 64);
            const int64_t pos0_v = (int64_t)__shfl((long long)pos0_reg, ln, 64);
            const int32_t n_v = __shfl(n_reg, ln, 64);
#ifndef GKI_ALL_LANE_PER_WINDOW
            // One lane per RECORD: the trip's windows hold nl records each (1 .. NLQ); written by their window's lane in
            // turn, neighbouring lanes store nl records apart and every line is touched by up to five store instructions
            // (the expansion ran at 3.75 TB/s where one-node mode's, coalesced, reaches 5.7).  Here the trip's records are
            // renumbered 0 .. R-1 (a wave scan of nl), a small bitmap marks where each window's records start, and in
            // sub-trips of 64 every lane takes one record: its window by popcount, the window's facts by shuffle from the
            // lane that decoded it, its node from the queued list -- consecutive lanes, consecutive records.
            uint64_t h = 0;
            int nl = 0;
            uint32_t first = 0;                                                 // slot of the window's first record, from idx_base
            float maf32 = 0.f;
            double maf64 = 0.0;
            if (valid) {
                h = ((q.ctx[e] >> (2 * o)) | (own_v << (2 * (k - 1 - o)))) & kmask;
                nl = (int)q.nl[e];
                first = q.idx[e] + (uint32_t)(j * nl);
                if (FMT == 1) maf64 = (double)q.maf[e]; else maf32 = (float)q.maf[e];
            }
            const int incl = gki_wave_incl_sum(nl);
            const int excl = incl - nl, R = gki_lane_value(incl, 63);
            const int packed = e | (o << 8) | (excl << 16);                     // what a record's lane wants of its window, one shuffle
            static_assert(EVQ <= 256 && NLQ * 64 < 32768, "step slot, offset and record number share a word");
            uint64_t *rmarks = reinterpret_cast<uint64_t *>(q.lbase);           // 8 words >= (64 * NLQ) / 64 = 5
            static_assert(NLQ * 64 <= 8 * 64, "record marks of one trip fit the reused lbase words");
            if (lane < 8) rmarks[lane] = 0ull;
            __builtin_amdgcn_wave_barrier();
            if (nl > 0) atomicOr(reinterpret_cast<unsigned int *>(rmarks) + (excl >> 5), 1u << (excl & 31));
            __builtin_amdgcn_wave_barrier();
            int before = 0;                                                     // windows that start in earlier sub-trips
            for (int rr0 = 0; rr0 < R; rr0 += 64) {
                const int rr = rr0 + lane;
                const bool live = rr < R;
                const uint64_t mw = rmarks[rr0 >> 6];                           // one word per sub-trip, the same for all lanes
                const int owner = live ? before + __popcll(mw & ((2ull << lane) - 1ull)) - 1 : 0;
                before += __popcll(mw);
                const int packed_o = __shfl(packed, owner, 64);
                const int e_o = packed_o & 0xFF, o_o = (packed_o >> 8) & 0xFF, t = rr - (packed_o >> 16);
                const uint64_t h_o = (uint64_t)__shfl((unsigned long long)h, owner, 64);
                const int64_t first_o = idx_base + (int64_t)(uint32_t)__shfl((int)first, owner, 64);
                const int64_t pos0_o = (int64_t)__shfl((long long)pos0_v, owner, 64);
                const int32_t n_o = __shfl(n_v, owner, 64);
                const double maf_o = FMT == 1 ? __shfl(maf64, owner, 64) : (double)__shfl(maf32, owner, 64);
                if (live) put(out, GKI_DBG_SLOT(first_o + t), h_o, q.nodes[e_o][t], n_o, o_o, pos0_o + o_o, maf_o);
            }
            __builtin_amdgcn_wave_barrier();
#else
            if (valid) {
                const uint64_t h = ((q.ctx[e] >> (2 * o)) | (own_v << (2 * (k - 1 - o)))) & kmask;
                const int nl = (int)q.nl[e];
                const int64_t first = idx_base + (int64_t)q.idx[e] + (int64_t)j * nl;
                for (int t = 0; t < nl; t++)
                    put(out, first + t, h, q.nodes[e][t], n_v, o, pos0_v + o, (double)q.maf[e]);
            }
#endif
        }
    }
}

template <bool DEEP>
struct LevelEmitT {              // a suspended level of the emit walk
    uint64_t ctx;
    double maf;
    int32_t cur, end, mn;
    uint8_t cum, evf, evt;
    typename CountOf<DEEP>::T vc;
};
static_assert(sizeof(LevelEmitT<false>) == 32 && sizeof(LevelEmitT<true>) <= DA_PATH - DA_BELOW && sizeof(LevelLoT<true>) <= DA_PATH - DA_BELOW,
              "a level of the product kernels is 32 bytes; the deep variants' levels fit their arena cell");

// Workgroups of ONE wave: the waves of this kernel never talk to each other (each has its own queue and walk cache), so
// the workgroup is only the unit the hardware schedules -- and the smaller that unit, the better the uneven node groups
// balance over the CUs.  Same LDS per wave, same 16 waves per CU.  One-node 2.23 -> 2.13 ms same-box, one rank's shard
// of eight 0.330 -> 0.311 ms, all-nodes unchanged within its noise (profiles/r03_boundary_grid_ab.txt); two waves per
// workgroup: half of that.  (-DGKI_BND_WPB=4: the workgroups of rounds 1-3.)
#ifndef GKI_BND_WPB
#define GKI_BND_WPB 1
#endif
constexpr int BND_WPB = GKI_BND_WPB;      // waves (groups of 64 nodes) per workgroup of the emit kernel
template <bool HAS_LOSSY, int FMT, bool ALL, bool GEN, bool DEEP = false>
// (general one-node variants in the flat layouts: LDS allows 4 workgroups per CU; without the request they take 129 VGPRs)
__global__ __launch_bounds__(64 * BND_WPB, (EMIT_SLIM && !GEN && !ALL && !HAS_LOSSY && FMT != 1) ? 5 : (GEN && !ALL && FMT != 1 && !DEEP) ? 4 : 1) void k_emit_boundary_one(DevGraph g, FindArgs a, const uint16_t *__restrict__ lossy,
                                                           const uint32_t *__restrict__ bcount,
                                                           const int64_t *__restrict__ rec_base,
                                                           const int64_t *__restrict__ bnd_shift,
                                                           typename OutSel<FMT>::T out, int *__restrict__ err, DeepArena da) {
    typedef WalkCacheT<(!ALL && !EMIT_SLIM) || GEN || HAS_LOSSY> WCache;      // the flag words only where they are read (or LDS is not the limit)
    __shared__ EvQueue<FMT, ALL> s_q[BND_WPB];
    __shared__ WCache s_wc[BND_WPB];
    typedef LevelEmitT<DEEP> LevelEmit;
    typedef typename CountOf<DEEP>::T cnt_t;
    const int64_t lane_global = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (deep variant: the lane's column of the arena)
    const int cap = DEEP ? da.cap : MAXN;
    typename StackOf<LevelEmit, MAXN, DEEP>::type below;
    bind(below, da, DA_BELOW, lane_global);
    LevelEmit below0;                  // the first suspended level stays in registers
    below0.ctx = 0; below0.maf = 0.0; below0.cur = below0.end = below0.mn = 0; below0.cum = below0.vc = below0.evf = below0.evt = 0;
    typename StackOf<int32_t, (ALL || GEN) ? MAXN : 1, DEEP>::type path;   // all-nodes mode / general graphs: the node of every level of the walk (level 0 = the end node)
    typename StackOf<cnt_t, GEN ? MAXN : 1, DEEP>::type lvl_a;             // general graphs: `a` of the suspended levels (see history_ok)
    bind(path, da, DA_PATH, lane_global);
    bind(lvl_a, da, DA_LVLA, lane_global);
    constexpr bool LAZY = GEN && !ALL;     // path levels 1, 2 and lvl_a[1] in registers, see k_count_boundary
    int32_t pr1 = 0, pr2 = 0;
    int a_l1 = 0;
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    EvQueue<FMT, ALL> &q = s_q[wib];
    WCache &wc = s_wc[wib];
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int64_t n_threads = (int64_t)gridDim.x * blockDim.x;
    int n_ev = 0, my_cnt = 0;              // steps in the queue; of them, queued by this lane
    for (int64_t base = a.n0 + (int64_t)blockIdx.x * blockDim.x + wib * 64; base < a.n1; base += n_threads) {
        int64_t idx_base = 0;              // all-nodes mode: queued record slots are 32-bit offsets from the group's first
        if (ALL || EMIT_SLIM) {
            const int64_t last = base + 64 < a.n1 ? base + 64 : a.n1;
            idx_base = rec_base[base] + *bnd_shift;
            if (rec_base[last] - rec_base[base] > 0xFFFFFFFFll) { gki_raise(err, GKI_ERR_WINDOW_TOO_DEEP); continue; }
        }
        const WalkView wv = stage_walk(g, wc, base, lane, GEN ? a.nflags : nullptr, HAS_LOSSY ? lossy : nullptr);
        const int64_t n = base + lane;
        int L = 0, o_lo = 0, hi = 0;
        int64_t idx = 0, pos0 = 0;
        uint64_t own = 0;
        // top of the stack, in registers
        int32_t t_cur = 0, t_end = 0, t_mn = 0;
        int t_cum = 0, t_vc = 0, t_evf = 0, t_evt = 0;
        uint64_t t_ctx = 0;
        double t_maf = 0.0;
        int t_a = 0;
        if (n < a.n1 && bcount[n] > 0) {
            const NodeWalk wn = cached_walk(g, wc, wv, n);
            const uint16_t fn = GEN ? (uint16_t)cached_flag(a.nflags, HAS_LOSSY ? lossy : nullptr, wc, wv, n) : (uint16_t)0;
            const int32_t bl = bnd_len_of(g, a, lossy, n, wn.size);
            idx = rec_base[n] + *bnd_shift;
            pos0 = g.pos_base[n];
            o_lo = (n == a.node_begin) ? (int)(a.off_begin < bl ? a.off_begin : bl) : 0;
            const int o_hi = (n == a.node_end) ? (int)(a.off_end < bl ? a.off_end : bl) : bl;
            const int v0 = GEN ? ((fn & GKI_NODE_REF) ? 0 : 1) : (wn.is_ref ? 0 : 1);
            const bool nonfree0 = GEN ? !(fn & (GKI_NODE_REF | GKI_NODE_FORCED)) : v0 != 0;
            const int cn = HAS_LOSSY ? lossy_of(lossy, (int32_t)n) : -1;
            if (o_lo < o_hi && !(nonfree0 && a.M < 1)) {
                // windows inside the node itself (lossy-restart nodes only; rare, written by the lane alone)
                for (int o = o_lo > k - 1 ? o_lo : k - 1; o < o_hi; o++) {
                    if (HAS_LOSSY && cn >= 0 && o + 1 - k <= cn - 1 && cn <= o) continue;
                    if (GEN && a.store && !a.store[n]) continue;
                    put(out, idx++, gki_extract(g.seq2, wn.seq_start + o + 1 - k, k), (int32_t)n, (int32_t)n, o, pos0 + o,
                        FMT == 1 ? g.allele_freq[n] : (double)wn.af);
                }
                hi = o_hi < k - 1 ? o_hi : k - 1;
                if (HAS_LOSSY && cn >= 0 && cn < hi) hi = cn;
                if (o_lo < hi) {
                    preds_begin(g, wn, n, &t_cur, &t_end);
                    t_cum = 0; t_vc = v0; t_evf = t_evt = 0; t_ctx = 0;
                    t_mn = (int32_t)n; t_maf = FMT == 1 ? g.allele_freq[n] : (double)wn.af;   // float32 rounding is monotonic:
                    // the minimum of the rounded values is the rounded minimum (flat layout), v2 keeps float64
                    own = gki_extract(g.seq2, wn.seq_start, hi);
                    if constexpr (EvQueue<FMT, ALL>::PACKED) { LaneC lc; lc.own = own; lc.pos0 = pos0; q.c[lane] = lc; }
                    if (ALL || GEN) path[0] = (int32_t)n;
                    t_a = nonfree0 ? v0 : 0;
                    L = 1;
                }
            }
        }
        while (__any(L > 0)) {
            bool ev = false;
            int e_from = 0, e_to = 0, e_nl = 0;          // e_nl: nodes of the window = levels 0 .. e_nl-1 of `path`
            int32_t e_mn = 0;
            uint64_t e_ctx = 0;
            double e_maf = 0.0;
            if (L > 0) {
                if (t_cur >= t_end) {                       // leave the level: its own step comes after its subtree
                    if (t_evf < t_evt) { ev = true; e_from = t_evf; e_to = t_evt; e_ctx = t_ctx; e_mn = t_mn; e_maf = t_maf; e_nl = L; }
                    L--;
                    if (L > 0) {
                        // (by value: `L == 1 ? below0 : below[L - 1]` is a choice between two OBJECTS, which keeps below0
                        // addressable; general emit 2.98 -> 2.58 ms, all-nodes 5.91 -> 5.82 ms, count pass 0.57 -> 0.54 ms)
                        LevelEmit b = below0;
                        if (L != 1) b = below[L - 1];
                        t_cur = b.cur; t_end = b.end; t_cum = b.cum; t_vc = b.vc; t_evf = b.evf; t_evt = b.evt;
                        t_ctx = b.ctx; t_mn = b.mn; t_maf = b.maf;
                        if (GEN) t_a = (LAZY && L == 1) ? a_l1 : lvl_a[L];
                    }
                } else {
                    const int32_t qn = cached_preds_next(g, wc, wv, &t_cur);
                    const NodeWalk wq = cached_walk(g, wc, wv, qn);
                    const uint32_t fwq = (GEN || HAS_LOSSY) ? cached_flag(GEN ? a.nflags : nullptr, HAS_LOSSY ? lossy : nullptr, wc, wv, qn) : 0u;
                    const uint16_t fq = (uint16_t)fwq;
                    const int vq = t_vc + (GEN ? ((fq & GKI_NODE_REF) ? 0 : 1) : (wq.is_ref ? 0 : 1));
                    bool take;
                    int aq = 0;
                    if (!GEN) {
                        take = vq <= a.M;
                    } else {
                        aq = t_a ? t_a : ((fq & (GKI_NODE_REF | GKI_NODE_FORCED)) ? 0 : vq);
                        take = !(fq & GKI_NODE_DEAD) && !(aq && vq - aq >= a.M) &&
                               !((fq & GKI_NODE_HFS) && !(a.nflags[!LAZY ? path[L - 1] : L == 1 ? (int32_t)n : L == 2 ? pr1 : L == 3 ? pr2 : path[L - 1]] & GKI_NODE_FORCED));
                    }
                    if (take) {
                        if (L >= cap - 1) {
                            gki_raise(err, GKI_ERR_WINDOW_TOO_DEEP);
                        } else {
                            const int s = wq.size, c = t_cum;
                            if (LAZY) { if (L == 1) pr1 = qn; else if (L == 2) pr2 = qn; else path[L] = qn; }
                            else if (ALL || GEN) path[L] = qn;
                            const int32_t mn = qn < t_mn ? qn : t_mn;
                            const double maf = fmin(t_maf, FMT == 1 ? g.allele_freq[qn] : (double)wq.af);   // np.min, kmer_finder.py:143
                            bool deeper;
                            int new_cum, from = 0, to = 0;
                            uint64_t cx = t_ctx;
                            if (s == 0) {
                                deeper = true; new_cum = c;
                            } else {
                                from = k - 1 - c - s; if (from < o_lo) from = o_lo;
                                to = k - 1 - c; if (to > hi) to = hi;
                                const int cq = HAS_LOSSY ? lossy_in(fwq) : -1;
                                if (HAS_LOSSY && cq >= 0) { const int min_ok = k - 1 - c - s + cq; if (from < min_ok) from = min_ok; }
                                if (GEN && from < to && !(fq & (GKI_NODE_T | GKI_NODE_SIMPLE))) {      // a history before q?
                                    // (fq >> 8: no history holds more variant nodes in the k bases before q -- if even
                                    // that many fit under the limit, any history that enters q will do, and q is entered)
                                    const bool ok = (fq & GKI_NODE_NESTED) ? (vq + (int)(fq >> 8) < a.M || (LAZY ? (void)(path[1] = pr1, path[2] = pr2) : (void)0, history_ok(WalkSrc{g.walk, g.rev_edges, g.rev_start}, wc, wv, a.nflags, k, a.M, raw(path), L, err, da, lane_global))) : false;
                                    if (!ok) to = from;
                                }
                                const int tq = s < k - 1 - c ? s : k - 1 - c;
                                cx = t_ctx | (node_tail(wq, tq) << (2 * (k - 1 - c - tq)));
                                deeper = (k - 1 - c - s > o_lo) && !(HAS_LOSSY && cq >= 0);
                                new_cum = c + s;
                            }
                            if (deeper) {
                                LevelEmit b;
                                b.cur = t_cur; b.end = t_end; b.cum = (uint8_t)t_cum; b.vc = (cnt_t)t_vc;
                                b.evf = (uint8_t)t_evf; b.evt = (uint8_t)t_evt; b.ctx = t_ctx; b.mn = t_mn; b.maf = t_maf;
                                if (L == 1) below0 = b; else below[L - 1] = b;
                                if (GEN) { if (LAZY && L == 1) a_l1 = t_a; else lvl_a[L] = (cnt_t)t_a; t_a = aq; }
                                preds_begin(g, wq, qn, &t_cur, &t_end);
                                t_cum = new_cum; t_vc = vq;
                                t_evf = from < to ? from : 0; t_evt = from < to ? to : 0;
                                t_ctx = cx; t_mn = mn; t_maf = maf;
                                L++;
                            } else if (from < to) {
                                ev = true; e_from = from; e_to = to; e_ctx = cx; e_mn = mn; e_maf = maf; e_nl = L + 1;
                            }
                        }
                    }
                }
            }
            int e_nls = e_nl;                        // nodes of the window that get a record (only_store_nodes)
            if (GEN && a.store && ev) {
                if (LAZY) { path[1] = pr1; path[2] = pr2; }
                e_nls = stored_nodes(a.store, raw(path), e_nl, !ALL);
                if (e_nls == 0) ev = false;
            }
            if (ALL && ev && e_nl > NLQ && GKI_DBG_SKIP_EXPAND_IS(4)) ev = false;
            if (ALL && DEEP) {
                // deep variant: a window over more nodes than a queued step carries is written by its lane alone, the
                // nodes that get a record found in ascending order by repeated selection over the lane's path (the
                // nodes of a path are distinct) -- quadratic in the window's nodes, and only the slow path pays it
                if (ev && e_nl > NLQ) {
                    const bool filt = GEN && a.store;
                    const int cntw = e_to - e_from;
                    int32_t prev = INT_MIN;
                    for (int r = 0; r < e_nls; r++) {
                        int32_t best = INT_MAX;
                        for (int a2 = 0; a2 < e_nl; a2++) {
                            const int32_t v = path[a2];
                            if (v > prev && v < best && !(filt && !a.store[v])) best = v;
                        }
                        for (int w = 0; w < cntw; w++) {
                            const int o = e_from + w;
                            const uint64_t h = ((e_ctx >> (2 * o)) | (own << (2 * (k - 1 - o)))) & kmask;
                            put(out, idx + (int64_t)w * e_nls + r, h, best, (int32_t)n, o, pos0 + o, e_maf);
                        }
                        prev = best;
                    }
                    idx += (int64_t)cntw * e_nls;
                    ev = false;
                }
            }
            if (ALL && !DEEP) {
                // A window over more nodes than a queued step carries (rows of empty or 1-bp nodes, SNPs a few bases
                // apart): written by the WAVE, one such step at a time.  Written by its lane alone -- per record a
                // selection over the scratch-resident path -- a single step held its wave for thousands of dependent
                // scratch loads; on the 3 Gbp SNP graph those stragglers were 3.4 of the kernel's 12 ms.
                uint64_t ovf = __ballot(ev && e_nl > NLQ);
                if (ovf) {
                    int32_t *stage = reinterpret_cast<int32_t *>(q.marks);     // 2 * MAXN words, idle outside expand_queue
                    static_assert(MAXN <= 64 && sizeof(q.marks) >= 2 * MAXN * sizeof(int32_t), "staging for one node list");
                    const bool filt = GEN && a.store;
                    while (ovf) {
                        const int owner = __ffsll((unsigned long long)ovf) - 1;
                        ovf &= ovf - 1;
                        if (lane == owner) {                                    // its nodes that get a record, path order
                            int j2 = 0;
                            for (int a2 = 0; a2 < e_nl; a2++) {
                                const int32_t v = path[a2];
                                if (filt && !a.store[v]) continue;
                                stage[j2++] = v;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int nls = __shfl(e_nls, owner, 64), from = __shfl(e_from, owner, 64);
                        const int n_rec = (__shfl(e_to, owner, 64) - from) * nls;
                        const uint64_t cx = __shfl((unsigned long long)e_ctx, owner, 64), ow = __shfl((unsigned long long)own, owner, 64);
                        const int64_t ix = __shfl((long long)idx, owner, 64), p0 = __shfl((long long)pos0, owner, 64);
                        const double maf = __shfl(e_maf, owner, 64);
                        const int32_t nn = (int32_t)(base + owner);
                        if (lane < nls) {                                       // ascending: the nodes of a path are distinct
                            const int32_t v = stage[lane];
                            int rank = 0;
                            for (int b2 = 0; b2 < nls; b2++) rank += stage[b2] < v ? 1 : 0;
                            stage[MAXN + rank] = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                        for (int rec = lane; rec < n_rec; rec += 64) {          // per offset the distinct nodes ascending
                            const int w = rec / nls, o = from + w;
                            const uint64_t h = ((cx >> (2 * o)) | (ow << (2 * (k - 1 - o)))) & kmask;
                            put(out, ix + rec, h, stage[MAXN + rec - w * nls], nn, o, p0 + o, maf);
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (lane == owner) { idx += n_rec; ev = false; }
                    }
                }
            }
            const uint64_t pending = __ballot(ev);
            if (pending) {
                const int n_new = __popcll(pending);
                if (n_ev + n_new > EVQ) {                    // wave-uniform: make room first
                    expand_queue<FMT, ALL>(q, n_ev, my_cnt, out, k, kmask, lane, own, pos0, (int32_t)n, idx_base);
                    n_ev = 0; my_cnt = 0;
                }
                if (ev) {
                    const int slot = n_ev + __popcll(pending & lt_mask);
                    q.small[slot] = step_word(e_from, e_to - e_from, lane, my_cnt++);
                    if constexpr (EvQueue<FMT, ALL>::PACKED) {
                        StepA<FMT> sa; sa.ctx = e_ctx; sa.idx = idx;
                        StepB<FMT> sb; sb.maf = (typename MafOf<FMT>::T)e_maf; sb.mn = e_mn;
                        q.a[slot] = sa; q.b[slot] = sb;
                    } else {
                        q.ctx[slot] = e_ctx; q.idx[slot] = (typename IdxOf<true>::T)(idx - idx_base);
                        q.maf[slot] = (typename MafOf<FMT>::T)e_maf;
                        if (!ALL) q.mn[slot] = e_mn;
                    }
                    if (ALL) {
                        q.nl[slot] = (uint8_t)e_nls;
                        const bool filt = GEN && a.store;
                        if (!GKI_DBG_SKIP_EXPAND_IS(3)) {
                            // in path order; expand_queue sorts them, one lane per step (a sort here is a divergent double
                            // loop over scratch that the whole wave executes: 2/3 of this mode's extra walk time)
                            int j2 = 0;
                            for (int a2 = 0; a2 < e_nl; a2++) {
                                const int32_t v = path[a2];
                                if (filt && !a.store[v]) continue;
                                q.nodes[slot][j2++] = v;
                            }
                        }
                        idx += (int64_t)(e_to - e_from) * e_nls;
                    } else
                    idx += e_to - e_from;
                }
                n_ev += n_new;
            }
        }
        if (n_ev > 0) {                                      // the queue never mixes node groups (see expand_queue)
            expand_queue<FMT, ALL>(q, n_ev, my_cnt, out, k, kmask, lane, own, pos0, (int32_t)n, idx_base);
            n_ev = 0; my_cnt = 0;
        }
    }
}

// ------------------------------------------------------------------------------------ per-node constants
__global__ __launch_bounds__(256) void k_node_emit(DevGraph g, FindArgs a, const uint16_t *__restrict__ lossy,
                                                   const uint32_t *__restrict__ bcount,
                                                   const int64_t *__restrict__ rec_base, NodeEmit *__restrict__ ne) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // Only the run's nodes are refreshed; records of other nodes may be stale from an earlier run, which is why the
    // interior kernels also bound p by the run's base range [p_begin, p_end).
    for (int64_t n = a.n0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < a.n1; n += stride) {
        const int32_t size = g.node_size[n];
        if (size <= 0) continue;
        const bool inside = in_run(a, n);
        int64_t lo = 0, hi = size;
        if (inside) interior_range(a, n, size, bnd_len_of(g, a, lossy, n, size), &lo, &hi);
        const int64_t ss = g.seq_start[n];
        NodeEmit e;
        e.glo = inside ? ss + lo : (int64_t)0x7FFFFFFFFFFFFFFFll;        // outside the run: never interior
        e.D = inside ? rec_base[n] + (a.split ? 0 : (int64_t)bcount[n]) - lo - ss : 0;
        e.E = g.pos_base[n] - ss;
        e.node = (int32_t)n;
        e.af = (float)g.allele_freq[n];
        e.cnt = (inside && hi > lo) ? (int32_t)(hi - lo) : 0;
        e.pad0 = 0; e.pad1 = 0;
        ne[g.node_rank[n]] = e;         // indexed by rank among non-empty nodes: what the bitmap popcount yields
    }
}

// ------------------------------------------------------------------------------------ interior stream
// Offsets >= bnd_len of a node have exactly one window, inside the node: record index of interior position p of
// node n is p + D[n], its hash a 2k-bit field of the 2-bit stream.  The owner of base p is
// rank[p/64] + popcount(mask[p/64] & bits <= p%64) - 1 in the list of non-empty nodes.
constexpr int SW = 64;                  // 64-base words per wave trip of the interior kernels (4096 bases)
constexpr int INTERIOR_MAX_BLOCKS = 256 * 7;

// FlatKmers variant (all four columns), LDS-staged and run-aligned.  History, all measured on MI355X:
//  v1  per 64-base word: mask/rank -> per-lane gather of the node record -> four stores: 2.6-2.9 TB/s.  On gfx950
//      vmcnt retires loads AND stores in order and under a saturated write stream a dependent global load takes
//      microseconds; dropping only that gather gave 4.5 TB/s, dropping all loads 5.3 TB/s.
//  v3  bitmap words, ranks, the 2-bit window and the node-record slice staged in LDS per 4096-base window, no
//      global load in the inner loop: 3.9-4.5 TB/s.  But a store starting at record p + D touches 9 cache lines
//      instead of 8 once bubbles have shifted D (the store microbenchmark loses 13 % to exactly that).
//  v4  (this kernel) lanes are mapped to OUTPUT records instead: the wave walks the node runs of its 4096-base
//      window (from the per-node table in LDS, all values wave-uniform) and covers each run
//      [glo + D, glo + cnt + D) with 64-record groups that start at a multiple of 16 records, i.e. on a cache-line
//      boundary of all four columns: -6 %.  With the split output layout (no holes between runs) 6.1 TB/s.
template <int SWT, int CAP>
__global__ __launch_bounds__(256) void k_emit_interior_runs(DevGraph g, FindArgs a, const NodeEmit *__restrict__ ne,
                                                            OutFlat out, int64_t word_begin, int64_t word_end,
                                                            int64_t p_begin, int64_t p_end) {
    __shared__ uint32_t s_ne[4][CAP * 12];
    __shared__ uint64_t s_seq[4][2 * SWT + 8];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wave = (int64_t)blockIdx.x * 4 + wib;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1ull;
    const int64_t n_seq_words = ((g.n_bases + 31) >> 5) + 2;
    const uint32_t *ne32 = reinterpret_cast<const uint32_t *>(ne);
    uint32_t *my_ne = s_ne[wib];
    uint64_t *my_seq = s_seq[wib];
    for (int64_t sw0 = word_begin + wave * SWT; sw0 < word_end; sw0 += n_waves * SWT) {
        const int nw = (int)((word_end - sw0) < SWT ? (word_end - sw0) : SWT);
        int64_t win_lo = sw0 * 64, win_hi = (sw0 + nw) * 64;
        if (win_lo < p_begin) win_lo = p_begin;
        if (win_hi > p_end) win_hi = p_end;
        // ---- stage 1: node ranks at both ends of the window (uniform), 2-bit window -> LDS
        const uint64_t m0 = g.start_mask[sw0];
        const int64_t jbase = (int64_t)g.start_rank[sw0] + (int64_t)(m0 & 1ull) - 1;
        const int64_t jlast = (int64_t)g.start_rank[sw0 + nw] - 1;       // last node starting before the window's end
        const int64_t sb = 2 * sw0 - 2 > 0 ? 2 * sw0 - 2 : 0;            // first staged 2-bit word
#pragma unroll
        for (int t = 0; t < (2 * SWT + 8 + 63) / 64; t++) {
            const int i = t * 64 + lane;
            if (i < 2 * SWT + 8) {
                int64_t gw = sb + i;
                my_seq[i] = g.seq2[gw < n_seq_words ? gw : n_seq_words - 1];
            }
        }
        for (int64_t jc = jbase; jc <= jlast; jc += CAP) {
            // ---- stage 2: this chunk of the per-node table -> LDS (12 dwords per node, coalesced)
            const int n_rec = (int)((jlast - jc + 1) < CAP ? (jlast - jc + 1) : CAP);
            for (int t = lane; t < 12 * n_rec; t += 64) my_ne[t] = ne32[12 * jc + t];
            // ---- stage 3: node runs; everything about the run is wave-uniform
            for (int r = 0; r < n_rec; r++) {
                const uint32_t *e = my_ne + 12 * r;
                const int cnt = __builtin_amdgcn_readfirstlane((int)e[8]);
                if (cnt <= 0) continue;
                const int64_t glo = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)e[1]) << 32) |
                                              (uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]));
                int64_t lo_p = glo > win_lo ? glo : win_lo;
                int64_t hi_p = glo + cnt < win_hi ? glo + cnt : win_hi;
                if (lo_p >= hi_p) continue;
                const int64_t D = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)e[3]) << 32) |
                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)e[2]));
                const int64_t E = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)e[5]) << 32) |
                                            (uint32_t)__builtin_amdgcn_readfirstlane((int)e[4]));
                const uint32_t node = (uint32_t)__builtin_amdgcn_readfirstlane((int)e[6]);
                const float af = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)e[7]));
                const int64_t a_idx = lo_p + D, b_idx = hi_p + D;
                for (int64_t g0 = a_idx & ~15ll; g0 < b_idx; g0 += 64) {
                    const int64_t idx = g0 + lane;
                    if (idx >= a_idx && idx < b_idx) {
                        const int64_t p = idx - D;
                        const int64_t P = p - (k - 1);                   // >= 0 for an interior position
                        const int si = (int)((P >> 5) - sb);
                        const int sh = (int)(P & 31) * 2;
                        const uint64_t lo = my_seq[si], hi = my_seq[si + 1];
                        const uint64_t h = ((lo >> sh) | ((hi << 1) << (63 - sh))) & kmask;
                        out.hash[idx] = h;
                        out.node[idx] = node;
                        out.ref_offset[idx] = (uint64_t)(p + E);
                        out.af[idx] = af;
                    }
                }
            }
        }
    }
}

// General variant (any subset of columns, v2 layout): same mapping, branchy.
template <int FMT>
__global__ __launch_bounds__(256) void k_emit_interior(DevGraph g, FindArgs a, const NodeEmit *__restrict__ ne,
                                                       typename OutSel<FMT>::T out, int64_t word_begin,
                                                       int64_t word_end, int64_t p_begin, int64_t p_end) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int k = a.k;
    const uint64_t lane_mask = ~0ull >> (63 - lane);
    for (int64_t w = word_begin + wave; w < word_end; w += n_waves) {
        const uint64_t mask = g.start_mask[w];
        const uint32_t rank = g.start_rank[w];
        const int64_t p = w * 64 + lane;
        const uint32_t j = rank + (uint32_t)__popcll(mask & lane_mask) - 1u;
        const NodeEmit e = ne[j];
        if (p < e.glo || p >= e.glo + e.cnt || p < p_begin || p >= p_end) continue;      // (glo = INT64_MAX, cnt = 0: not in the run)
        const uint64_t h = gki_extract(g.seq2, p - (k - 1), k);
        if (FMT == 0) {
            put<false>(out, p + e.D, h, e.node, e.node, 0, p + e.E, (double)e.af);
        } else {
            const int64_t o = p - g.seq_start[e.node];
            put<false>(out, p + e.D, h, e.node, e.node, (int32_t)o, 0, g.allele_freq[e.node]);   // v2 keeps float64 (:58)
        }
    }
}

// out4 = {total records, error flag, boundary records (split layout only), 0}; out4[3] doubles as the zero shift
__global__ void k_totals(const int64_t *rec_base, const int64_t *bnd_base, int64_t n1, int split, const int *err, int64_t *out4) {
    const int64_t a = rec_base[n1];
    const int64_t b = split ? bnd_base[n1] : 0;
    out4[0] = a + b;
    out4[1] = *err;
    out4[2] = b;
    out4[3] = 0;
}

__global__ __launch_bounds__(256) void k_sum_u32(const uint32_t *__restrict__ x, int64_t n, unsigned long long *__restrict__ out) {
    unsigned long long s = 0;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += x[i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

}  // namespace

struct gki_finder {
    gki_graph *g;
    hipStream_t stream, stream2;
    uint32_t *bcount, *total;
    int64_t *rec_base, *bnd_base;     // by-node layout: rec_base only; split layout: interior bases / boundary bases
    NodeEmit *ne;
    uint16_t *lossy;
    const uint16_t *lossy_cur;        // the table of the run in progress: `lossy` (uploaded by the count) or the caller's device table
    uint16_t *nflags; uint8_t *store; // general graphs / only_store_nodes (gki_find_params), allocated on first use
    void *scan_tmp; int64_t scan_tmp_bytes;
    int *d_err; int64_t *d_totals; unsigned long long *d_bsum;
    int64_t *h_totals;                // pinned: {total, error, boundary records, 0, boundary sum} read back by every count;
                                      // [5]: the error word after the emit kernels (read by gki_finder_synchronize)
    bool emit_pending;                // an emit call's error word has not been looked at yet
    int32_t *d_rank;                  // topological ranks of the run in progress (non-topological node ids only)
    DeepArena deep;                   // cap > 0: the run in progress needed the deep kernel variants (their stacks live here)
    int64_t deep_bytes;
    FindArgs args;
    int64_t n_records, n_boundary_records, n_interior_records;
    int64_t word_begin, word_end, p_begin, p_end;
    bool counted;
    hipEvent_t ev[8];    // pairs: 0/1 count-boundary, 2/3 emit-interior, 4/5 emit-boundary, 6/7 scan + per-node constants
    hipEvent_t ev_ready, ev_join;
    bool ev_valid[4];
};

// Launch parameters are fixed in the product; `make tuning` (-DGKI_TUNING, used by tools/exp) reads them from the
// environment for A/B runs.
#ifdef GKI_TUNING
static int tuning_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
#define GKI_KNOB(name, dflt) tuning_int(name, dflt)
#else
#define GKI_KNOB(name, dflt) (dflt)
#endif

template <int FMT, bool ALL>
static int launch_boundary_mode(gki_finder *f, const DevGraph &d, const FindArgs &a, typename OutSel<FMT>::T out, hipStream_t s2,
                                const dim3 grid, const dim3 block, const int64_t *base, const int64_t *shift) {
    const int emit_pad = GKI_KNOB("GKI_EMIT_LDS_PAD", 0);  // tuning builds: unused dynamic LDS, to lower the occupancy
    // the lossy-restart logic costs 0.4-0.5 ms (emit) + 0.1-0.2 ms (count) per step on the 3 Gbp graphs even when the run has
    // no such point (a lossy[q] load per predecessor step): every variant exists without it
    const DeepArena da = f->deep;
    if (da.cap > 0) {
        // the count pass met a window deeper than the product kernels' stacks: the same kernels with their stacks in the
        // arena, on the grid the arena was sized for
        const dim3 dgrid((unsigned)(da.lanes / (64 * BND_WPB)));
        if (a.nflags && a.has_lossy)
            hipLaunchKernelGGL((k_emit_boundary_one<true, FMT, ALL, true, true>), dgrid, block, 0, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
        else if (a.nflags)
            hipLaunchKernelGGL((k_emit_boundary_one<false, FMT, ALL, true, true>), dgrid, block, 0, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
        else if (a.has_lossy)
            hipLaunchKernelGGL((k_emit_boundary_one<true, FMT, ALL, false, true>), dgrid, block, 0, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
        else
            hipLaunchKernelGGL((k_emit_boundary_one<false, FMT, ALL, false, true>), dgrid, block, 0, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
        HIP_TRY(hipGetLastError());
        return GKI_OK;
    }
    if (a.nflags && a.has_lossy)
        hipLaunchKernelGGL((k_emit_boundary_one<true, FMT, ALL, true>), grid, block, emit_pad, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
    else if (a.nflags)
        hipLaunchKernelGGL((k_emit_boundary_one<false, FMT, ALL, true>), grid, block, emit_pad, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
    else if (a.has_lossy)
        hipLaunchKernelGGL((k_emit_boundary_one<true, FMT, ALL, false>), grid, block, emit_pad, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
    else
        hipLaunchKernelGGL((k_emit_boundary_one<false, FMT, ALL, false>), grid, block, emit_pad, s2, d, a, f->lossy_cur, f->bcount, base, shift, out, f->d_err, da);
    HIP_TRY(hipGetLastError());
    return GKI_OK;
}

template <int FMT>
static int launch_boundary_fmt(gki_finder *f, const DevGraph &d, const FindArgs &a, typename OutSel<FMT>::T out, hipStream_t s2) {
#ifdef GKI_TUNING
    static bool dbg_set = false;
    if (!dbg_set) {
        int v = GKI_KNOB("GKI_DBG_SKIP_EXPAND", 0);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_skip_expand), &v, sizeof(int)));
        dbg_set = true;
    }
#endif
    const int bnd_blocks = GKI_KNOB("GKI_BND_BLOCKS", 0);
    // One workgroup per group of nodes (64 * BND_WPB; two groups in all-nodes mode), as many workgroups as that takes: the hardware hands
    // a finished CU its next workgroup, so the walk's uneven node groups balance themselves.  With 2048 workgroups walking
    // the groups grid-stride (rounds 1-2) the kernel waited for its slowest workgroups: same box, whole graph 2.52 -> 2.31 ms,
    // all-nodes 6.95 -> 5.89 ms, one rank's shard of eight 0.39 -> 0.35 ms (tools/exp/bnd_grid.sh,
    // profiles/r03_boundary_grid_ab.txt).  Tuning builds: GKI_BND_BLOCKS = g > 0 caps the grid, -g = g groups per workgroup.
    int gb = (int)ceil_div(a.n1 - a.n0, 64 * BND_WPB * (int64_t)(a.one_node ? 1 : 2));
    if (bnd_blocks > 0) { gb = stream_grid(a.n1 - a.n0, 256); if (gb > bnd_blocks) gb = bnd_blocks; }
    if (bnd_blocks < 0) gb = (int)ceil_div(a.n1 - a.n0, 256 * (int64_t)(-bnd_blocks));
    if (gb < 1) gb = 1;
    const dim3 grid(gb), block(64 * BND_WPB);
    // split layout: boundary block of node n starts at (number of interior records) + bnd_base[n]
    const int64_t *base = a.split ? f->bnd_base : f->rec_base;
    const int64_t *shift = a.split ? f->rec_base + a.n1 : f->d_totals + 3;
    if (a.one_node) return launch_boundary_mode<FMT, false>(f, d, a, out, s2, grid, block, base, shift);
    return launch_boundary_mode<FMT, true>(f, d, a, out, s2, grid, block, base, shift);
}
static int launch_boundary(gki_finder *f, const DevGraph &d, const FindArgs &a, OutFlat out, hipStream_t s2) {
    if (out.hash && out.node && out.ref_offset && out.af) {
        OutFlatAll all{out.hash, out.node, out.ref_offset, out.af};
        return launch_boundary_fmt<2>(f, d, a, all, s2);
    }
    return launch_boundary_fmt<0>(f, d, a, out, s2);
}
static int launch_boundary(gki_finder *f, const DevGraph &d, const FindArgs &a, OutV2 out, hipStream_t s2) {
    return launch_boundary_fmt<1>(f, d, a, out, s2);
}

static int launch_interior(gki_finder *f, const DevGraph &d, const FindArgs &a, OutFlat out, unsigned blocks) {
    hipStream_t s = f->stream;
    if (out.hash && out.node && out.ref_offset && out.af) {
        const int swt = GKI_KNOB("GKI_SW", 64);                 // bases per wave trip / 64
        const int rb = GKI_KNOB("GKI_RUN_BLOCKS", 256 * 5);
        const int cap = GKI_KNOB("GKI_NE_CAP", 128);
        const int64_t n_words = f->word_end - f->word_begin;
        unsigned gb = (unsigned)ceil_div(ceil_div(n_words, swt), 4);
        if (gb > (unsigned)rb) gb = (unsigned)rb;
#ifdef GKI_TUNING
        if (swt == 128)
            hipLaunchKernelGGL((k_emit_interior_runs<128, 128>), dim3(gb), dim3(256), 0, s, d, a, f->ne, out, f->word_begin, f->word_end, f->p_begin, f->p_end);
        else if (cap == 64)      // 16.7 KB of LDS per block: 8 blocks per CU
            hipLaunchKernelGGL((k_emit_interior_runs<64, 64>), dim3(gb), dim3(256), 0, s, d, a, f->ne, out, f->word_begin, f->word_end, f->p_begin, f->p_end);
        else if (cap == 32)
            hipLaunchKernelGGL((k_emit_interior_runs<64, 32>), dim3(gb), dim3(256), 0, s, d, a, f->ne, out, f->word_begin, f->word_end, f->p_begin, f->p_end);
        else
#else
        (void)cap;
#endif
        // 29 KB of LDS per block: 5 blocks per CU are resident
        hipLaunchKernelGGL((k_emit_interior_runs<64, 128>), dim3(gb), dim3(256), 0, s, d, a, f->ne, out, f->word_begin, f->word_end, f->p_begin, f->p_end);
    }
    else
        hipLaunchKernelGGL(k_emit_interior<0>, dim3(blocks), dim3(256), 0, s, d, a, f->ne, out, f->word_begin, f->word_end,
                           f->p_begin, f->p_end);
    HIP_TRY(hipGetLastError());
    return GKI_OK;
}
static int launch_interior(gki_finder *f, const DevGraph &d, const FindArgs &a, OutV2 out, unsigned blocks) {
    hipLaunchKernelGGL(k_emit_interior<1>, dim3(blocks), dim3(256), 0, f->stream, d, a, f->ne, out, f->word_begin, f->word_end,
                       f->p_begin, f->p_end);
    HIP_TRY(hipGetLastError());
    return GKI_OK;
}

template <int FMT>
static int emit_impl(gki_finder *f, typename OutSel<FMT>::T out) {
    if (!f->counted) return gki_set_error(GKI_ERR_STATE, "gki_finder_emit_* called before gki_finder_count");
    if (f->deep.cap > 0 && !f->deep.base) {                       // a second emit after one count: the arena again
        const int64_t bytes = f->deep.lanes * (int64_t)f->deep.cap * DA_CELL;
        HIP_TRY(gki_dev_malloc((void **)&f->deep.base, (size_t)bytes));
        f->deep_bytes = bytes;
    }
    const DevGraph &d = f->g->d;
    hipStream_t s = f->stream, s2 = f->stream2;
    const FindArgs a = f->args;
    // Measured on MI355X (3 Gbp graph): running the boundary and interior kernels on two streams is ~4% slower
    // than back to back (both press on the same write path), so one stream is the default.
    const bool overlap = GKI_KNOB("GKI_OVERLAP_EMIT", 0) != 0;
    if (!overlap) s2 = s;
    const bool boundary_first = GKI_KNOB("GKI_BOUNDARY_FIRST", 0) != 0;
    HIP_TRY(hipEventRecord(f->ev_ready, s));
    HIP_TRY(hipStreamWaitEvent(s2, f->ev_ready, 0));
    auto run_boundary = [&]() -> int {
        HIP_TRY(hipEventRecord(f->ev[4], s2));
        if (f->n_boundary_records > 0) GKI_TRY(launch_boundary(f, d, a, out, s2));
        HIP_TRY(hipEventRecord(f->ev[5], s2));
        return GKI_OK;
    };
    auto run_interior = [&]() -> int {
        HIP_TRY(hipEventRecord(f->ev[2], s));
        if (f->n_interior_records > 0 && f->word_end > f->word_begin) {
            int64_t n_words = f->word_end - f->word_begin;
            int64_t blocks = ceil_div(ceil_div(n_words, SW), 4);
            const int int_blocks = GKI_KNOB("GKI_INT_BLOCKS", INTERIOR_MAX_BLOCKS);
            if (blocks > int_blocks) blocks = int_blocks;
            GKI_TRY(launch_interior(f, d, a, out, (unsigned)blocks));
        }
        HIP_TRY(hipEventRecord(f->ev[3], s));
        return GKI_OK;
    };
    if (boundary_first) { GKI_TRY(run_boundary()); GKI_TRY(run_interior()); }
    else { GKI_TRY(run_interior()); GKI_TRY(run_boundary()); }
    // an emit kernel that cannot write a group's records (all-nodes mode: 64 consecutive nodes with more than 2^32
    // records between them; a window deeper than the count pass met) raises the error word and leaves them unwritten:
    // gki_finder_synchronize returns it (nothing read the word after an emit launch)
    f->h_totals[5] = 0;
    HIP_TRY(hipMemcpyAsync(&f->h_totals[5], f->d_err, 4, hipMemcpyDeviceToHost, s2));
    f->emit_pending = true;
    HIP_TRY(hipEventRecord(f->ev_join, s2));
    HIP_TRY(hipStreamWaitEvent(s, f->ev_join, 0));
    f->ev_valid[1] = true; f->ev_valid[2] = true;
    return GKI_OK;
}

// The graph's arrays live on ONE device: a finder (its streams and buffers) or an early-stop search made while another
// device is current would launch kernels there that dereference this device's memory (`index -t N` ranks
// selected their GPU after the graph upload).  Refuse instead of faulting.
int gki_check_graph_device(const gki_graph *g, const char *who) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != g->device)
        return gki_set_error(GKI_ERR_BAD_ARG, "%s: the graph was uploaded to device %d, the current device is %d "
                             "(call gki_set_device before gki_graph_create)", who, g->device, dev);
    return GKI_OK;
}

extern "C" {

static int finder_init(gki_finder *f, gki_graph *g) {
    const int64_t n = g->d.n_nodes;
    HIP_TRY(hipStreamCreate(&f->stream));
    HIP_TRY(hipStreamCreate(&f->stream2));
    for (int i = 0; i < 8; i++) HIP_TRY(hipEventCreate(&f->ev[i]));
    HIP_TRY(hipEventCreateWithFlags(&f->ev_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming));
    HIP_TRY(gki_dev_malloc((void **)&f->bcount, (size_t)n * 4));
    HIP_TRY(gki_dev_malloc((void **)&f->total, (size_t)n * 4));
    HIP_TRY(gki_dev_malloc((void **)&f->rec_base, (size_t)(n + 1) * 8));
    HIP_TRY(gki_dev_malloc((void **)&f->bnd_base, (size_t)(n + 1) * 8));
    HIP_TRY(gki_dev_malloc((void **)&f->ne, (size_t)(g->d.n_nonempty + 1) * sizeof(NodeEmit)));
    HIP_TRY(hipMemset(f->ne, 0, (size_t)(g->d.n_nonempty + 1) * sizeof(NodeEmit)));   // cnt = 0: a record never refreshed emits nothing
    HIP_TRY(gki_dev_malloc((void **)&f->lossy, (size_t)n * 2));
    HIP_TRY(gki_dev_malloc((void **)&f->d_err, 4));
    HIP_TRY(gki_dev_malloc((void **)&f->d_totals, 4 * 8));
    HIP_TRY(hipMemset(f->d_totals, 0, 4 * 8));
    HIP_TRY(gki_dev_malloc((void **)&f->d_bsum, 8));
    HIP_TRY(hipHostMalloc((void **)&f->h_totals, 6 * 8, hipHostMallocDefault));
    f->scan_tmp_bytes = gki_scan_tmp_bytes(n);
    HIP_TRY(gki_dev_malloc(&f->scan_tmp, (size_t)f->scan_tmp_bytes));
    return GKI_OK;
}

int gki_finder_create(gki_graph *g, gki_finder **out) {
    *out = nullptr;
    if (!g) return gki_set_error(GKI_ERR_BAD_ARG, "finder_create: graph is NULL");
    GKI_TRY(gki_check_graph_device(g, "gki_finder_create"));
    gki_finder *f = new gki_finder();
    memset(f, 0, sizeof(*f));
    f->g = g;
    const int rc = finder_init(f, g);
    if (rc != GKI_OK) { (void)gki_finder_destroy(f); return rc; }      // nothing of a half-built finder stays behind
    *out = f;
    return GKI_OK;
}

int gki_finder_destroy(gki_finder *f) {
    if (!f) return GKI_OK;
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->stream2) (void)hipStreamSynchronize(f->stream2);
    void *ptrs[] = {f->bcount, f->total, f->rec_base, f->bnd_base, f->ne, f->lossy, f->scan_tmp, f->d_err, f->d_totals, f->d_bsum,
                    f->d_rank, f->nflags, f->store, f->deep.base};
    for (void *p : ptrs) if (p) (void)gki_dev_free(p);
    if (f->h_totals) (void)hipHostFree(f->h_totals);
    for (int i = 0; i < 8; i++) if (f->ev[i]) (void)hipEventDestroy(f->ev[i]);
    if (f->ev_ready) (void)hipEventDestroy(f->ev_ready);
    if (f->ev_join) (void)hipEventDestroy(f->ev_join);
    if (f->stream) (void)hipStreamDestroy(f->stream);
    if (f->stream2) (void)hipStreamDestroy(f->stream2);
    delete f;
    return GKI_OK;
}

int gki_finder_count(gki_finder *f, const gki_find_params *p, int64_t *n_records) {
    *n_records = 0;
    f->counted = false;
    const DevGraph &d = f->g->d;
    if (p->struct_size != sizeof(gki_find_params))
        return gki_set_error(GKI_ERR_BAD_ARG, "gki_find_params.struct_size is %u, this library expects %u: the binding was "
                             "written against another version of include/gki.h", p->struct_size, (unsigned)sizeof(gki_find_params));
    if (p->k < 1 || p->k > GKI_MAX_K) return gki_set_error(GKI_ERR_BAD_ARG, "k must be in 1..31 (got %d)", p->k);
    if (p->max_variant_nodes < 0) return gki_set_error(GKI_ERR_BAD_ARG, "max_variant_nodes < 0");
    if (p->node_begin < 0 || p->node_end > d.n_nodes || (!p->h_node_rank && !p->d_node_rank && p->node_begin > p->node_end))
        return gki_set_error(GKI_ERR_BAD_ARG, "bad node range [%lld, %lld]", (long long)p->node_begin, (long long)p->node_end);
    hipStream_t s = f->stream;
    FindArgs a;
    a.k = p->k; a.M = p->max_variant_nodes > 250 ? 250 : p->max_variant_nodes;
    a.one_node = p->one_node_per_kmer ? 1 : 0;
    a.has_lossy = (p->h_lossy_crit || p->d_lossy_crit) ? 1 : 0;
    // tuning builds: run the lossy-restart kernel variants on a graph without such points (what does the variant cost?)
    const bool force_lossy = !a.has_lossy && GKI_KNOB("GKI_FORCE_LOSSY", 0) != 0;
    if (force_lossy) { HIP_TRY(hipMemsetAsync(f->lossy, 0xFF, (size_t)d.n_nodes * 2, s)); a.has_lossy = 1; }
    a.node_begin = p->node_begin; a.off_begin = p->off_begin; a.node_end = p->node_end; a.off_end = p->off_end;
    a.n0 = p->node_begin < d.n_nodes ? p->node_begin : d.n_nodes;
    a.n1 = p->node_end < d.n_nodes ? p->node_end + 1 : d.n_nodes;
    a.split = p->layout == GKI_LAYOUT_SPLIT ? 1 : 0; a.pad = 0;
    a.rank = nullptr; a.rank_begin = 0; a.rank_end = 0;
    a.nflags = nullptr; a.store = nullptr;
    // the four per-run tables: already on the device (d_ forms: nothing to upload), or host arrays uploaded per count
    if (p->d_store_nodes || p->h_store_nodes) {
        if (!p->h_node_flags && !p->d_node_flags) return gki_set_error(GKI_ERR_BAD_ARG, "store_nodes needs node_flags (gki_classify_nodes)");
        if (p->d_store_nodes) a.store = p->d_store_nodes;
        else {
            if (!f->store) HIP_TRY(gki_dev_malloc((void **)&f->store, (size_t)d.n_nodes));
            HIP_TRY(hipMemcpyAsync(f->store, p->h_store_nodes, (size_t)d.n_nodes, hipMemcpyHostToDevice, s));
            a.store = f->store;
        }
    }
    if (p->d_node_flags) a.nflags = p->d_node_flags;
    else if (p->h_node_flags) {
        if (!f->nflags) HIP_TRY(gki_dev_malloc((void **)&f->nflags, (size_t)d.n_nodes * 2));
        HIP_TRY(hipMemcpyAsync(f->nflags, p->h_node_flags, (size_t)d.n_nodes * 2, hipMemcpyHostToDevice, s));
        a.nflags = f->nflags;
    }
    if (p->d_node_rank) {
        a.rank = p->d_node_rank; a.rank_begin = p->rank_begin; a.rank_end = p->rank_end;
        a.n0 = 0; a.n1 = d.n_nodes;
    } else if (p->h_node_rank) {
        // node ids are not topological: membership by rank, every node is looked at
        if (!f->d_rank) HIP_TRY(gki_dev_malloc((void **)&f->d_rank, (size_t)d.n_nodes * 4));
        HIP_TRY(hipMemcpyAsync(f->d_rank, p->h_node_rank, (size_t)d.n_nodes * 4, hipMemcpyHostToDevice, s));
        a.rank = f->d_rank;
        a.rank_begin = p->node_begin < d.n_nodes ? p->h_node_rank[p->node_begin] : INT_MAX;
        a.rank_end = p->node_end < d.n_nodes ? p->h_node_rank[p->node_end] : INT_MAX;
        a.n0 = 0; a.n1 = d.n_nodes;
    }
    f->args = a;
    const int64_t n_run = a.n1 - a.n0;
    f->lossy_cur = f->lossy;
    if (a.has_lossy && !force_lossy) {
        if (p->d_lossy_crit) f->lossy_cur = p->d_lossy_crit;
        else HIP_TRY(hipMemcpyAsync(f->lossy, p->h_lossy_crit, (size_t)d.n_nodes * 2, hipMemcpyHostToDevice, s));
    }
    // the product kernels first; a window deeper than their stacks sends the pass round again.  The slow path's arena of an
    // earlier run (up to 14 GB) went back to the pool when that run's emit completed (gki_finder_synchronize).
    f->deep.cap = 0;
    int64_t *tot = f->h_totals;       // pinned host memory: the two small copies are true async DMAs
    unsigned long long bsum = 0;
    for (;;) {
        HIP_TRY(hipMemsetAsync(f->d_err, 0, 4, s));
        HIP_TRY(hipMemsetAsync(f->d_bsum, 0, 8, s));

        HIP_TRY(hipEventRecord(f->ev[0], s));
        if (n_run > 0) {
            // one group of 256 nodes per workgroup, no grid-stride loop: with 2048 workgroups of 29 groups each (7 fit a CU,
            // an eighth waits) the pass took 0.70 ms, with one group each 0.59 (GKI_CNT_BLOCKS sweeps it in tuning builds)
            const int64_t want = ceil_div(n_run, 64 * CNT_WPB), cap = GKI_KNOB("GKI_CNT_BLOCKS", 1 << 30);
            const int count_grid = (int)(want < cap ? want : cap);
            const int cnt_pad = GKI_KNOB("GKI_CNT_LDS_PAD", 0);   // tuning builds: unused dynamic LDS, to lower the occupancy
            const DeepArena da = f->deep;
            if (da.cap > 0) {
                const dim3 dgrid((unsigned)(da.lanes / (64 * CNT_WPB)));
                if (a.nflags && a.has_lossy)
                    hipLaunchKernelGGL((k_count_boundary<true, true, true>), dgrid, dim3(64 * CNT_WPB), 0, s, d, a, f->lossy_cur, f->bcount, f->total, f->d_err, da);
                else if (a.nflags)
                    hipLaunchKernelGGL((k_count_boundary<false, true, true>), dgrid, dim3(64 * CNT_WPB), 0, s, d, a, f->lossy_cur, f->bcount, f->total, f->d_err, da);
                else if (a.has_lossy)
                    hipLaunchKernelGGL((k_count_boundary<true, false, true>), dgrid, dim3(64 * CNT_WPB), 0, s, d, a, f->lossy_cur, f->bcount, f->total, f->d_err, da);
                else
                    hipLaunchKernelGGL((k_count_boundary<false, false, true>), dgrid, dim3(64 * CNT_WPB), 0, s, d, a, f->lossy_cur, f->bcount, f->total, f->d_err, da);
            }
            else if (a.nflags && a.has_lossy)
                hipLaunchKernelGGL((k_count_boundary<true, true>), dim3(count_grid), dim3(64 * CNT_WPB), cnt_pad, s, d, a, f->lossy_cur,
                                   f->bcount, f->total, f->d_err, da);
            else if (a.nflags)
                hipLaunchKernelGGL((k_count_boundary<false, true>), dim3(count_grid), dim3(64 * CNT_WPB), cnt_pad, s, d, a, f->lossy_cur,
                                   f->bcount, f->total, f->d_err, da);
            else if (a.has_lossy)
                hipLaunchKernelGGL((k_count_boundary<true, false>), dim3(count_grid), dim3(64 * CNT_WPB), cnt_pad, s, d, a, f->lossy_cur,
                                   f->bcount, f->total, f->d_err, da);
            else
                hipLaunchKernelGGL((k_count_boundary<false, false>), dim3(count_grid), dim3(64 * CNT_WPB), cnt_pad, s, d, a, f->lossy_cur,
                                   f->bcount, f->total, f->d_err, da);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(f->ev[1], s));
        HIP_TRY(hipEventRecord(f->ev[6], s));
        // records of nodes [n0, n1) only: rec_base[n0 + i] = exclusive prefix inside the run, rec_base[n1] = total
        GKI_TRY(gki_scan_u32_to_i64(f->total + a.n0, n_run, f->rec_base + a.n0, f->scan_tmp, f->scan_tmp_bytes, s));
        if (a.split)
            GKI_TRY(gki_scan_u32_to_i64(f->bcount + a.n0, n_run, f->bnd_base + a.n0, f->scan_tmp, f->scan_tmp_bytes, s));
        if (n_run > 0) {
            // one node per thread, no grid-stride loop: a thread's loads and its 48-byte store form one dependent chain, so the
            // kernel lives on the number of chains in flight
            hipLaunchKernelGGL(k_node_emit, dim3((unsigned)ceil_div(n_run, 256)), dim3(256), 0, s, d, a, f->lossy_cur, f->bcount,
                               f->rec_base, f->ne);
            HIP_TRY(hipGetLastError());
        }
        if (n_run > 0 && !a.split) {
            hipLaunchKernelGGL(k_sum_u32, dim3(stream_grid(n_run, 256)), dim3(256), 0, s, f->bcount + a.n0, n_run, f->d_bsum);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_totals, dim3(1), dim3(1), 0, s, f->rec_base, f->bnd_base, a.n1, a.split, f->d_err, f->d_totals);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(f->ev[7], s));
        HIP_TRY(hipMemcpyAsync(tot, f->d_totals, 32, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(tot + 4, f->d_bsum, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        bsum = (unsigned long long)tot[4];
        const int run_err = gki_error_of_word(tot[1]);
        if (run_err == GKI_ERR_NOT_ONE_REF_SUCC && !(tot[1] & 2))
            return gki_set_error(GKI_ERR_NOT_ONE_REF_SUCC, "a window at the variant limit ends a node that does not have exactly one "
                                 "linear-ref successor: the reference asserts here (kmer_finder.py:402); raise max_variant_nodes");
        if (run_err == GKI_OK) break;
        // GKI_ERR_WINDOW_TOO_DEEP.  Bit 1 of the word: a stack of the walk (or of a history) was too short -- the slow path:
        // the pass again with the deep kernel variants, their stacks in an arena of twice the levels each time round.
        // Bit 2: a history enumeration ran out of its step budget, which no depth cures.
        const int next_cap = f->deep.cap == 0 ? 4 * MAXN : 2 * f->deep.cap;
        if ((tot[1] & 4) || !(tot[1] & 2) || next_cap > GKI_MAX_DEEP_WINDOW_NODES)
            return gki_set_error(GKI_ERR_WINDOW_TOO_DEEP, (tot[1] & 4) ? "the k-windows ending in one node (or the histories that decide one of them) take more "
                                 "than %d descents to enumerate: too many paths" : "a k-window (or the history that decides it) crosses more than %d nodes",
                                 (tot[1] & 4) ? STEP_BUDGET : GKI_MAX_DEEP_WINDOW_NODES - 2);
        {
            // 64 workgroups walk the run (grid-stride): 16 384 lanes x cap levels x 70 bytes = 0.3 GB at the first cap
            const int64_t lanes = 64 * 256, bytes = lanes * (int64_t)next_cap * DA_CELL;
            if (bytes > f->deep_bytes) {
                if (f->deep.base) HIP_TRY(gki_dev_free(f->deep.base));
                f->deep.base = nullptr; f->deep_bytes = 0;
                HIP_TRY(gki_dev_malloc((void **)&f->deep.base, (size_t)bytes));
                f->deep_bytes = bytes;
            }
            f->deep.lanes = lanes; f->deep.cap = next_cap; f->deep.pad = 0;
        }
    }
    if (a.split) bsum = (unsigned long long)tot[2];
    f->n_records = tot[0];
    f->n_boundary_records = (int64_t)bsum;
    f->n_interior_records = tot[0] - (int64_t)bsum;
    // words of the sequence covered by the node range
    int64_t p0 = 0, p1 = d.n_bases;
    if (!a.rank && (a.node_begin > 0 || a.node_end < d.n_nodes)) {
        int64_t nb = a.node_begin < d.n_nodes ? a.node_begin : d.n_nodes;
        p0 = f->g->h_seq_start[nb];
        if (a.node_end < d.n_nodes) {
            p1 = f->g->h_seq_start[a.node_end];
            p1 += a.off_end;                        // (node_end, off_end) is exclusive
            if (p1 > d.n_bases) p1 = d.n_bases;
        }
    }
    f->word_begin = p0 >> 6;
    f->word_end = ceil_div(p1, 64);
    f->p_begin = p0;
    f->p_end = p1;
    f->ev_valid[0] = true; f->ev_valid[3] = true;
    f->counted = true;
    *n_records = tot[0];
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

int gki_finder_synchronize(gki_finder *f) {
    HIP_TRY(hipStreamSynchronize(f->stream));
    HIP_TRY(hipStreamSynchronize(f->stream2));
    if (f->deep.base) {                                           // the emit kernels are done with the slow path's stacks
        (void)gki_dev_free(f->deep.base);
        f->deep.base = nullptr; f->deep_bytes = 0;                // (cap stays: a second emit of this run allocates them again)
    }
    if (f->emit_pending) {
        f->emit_pending = false;
        const int64_t word = f->h_totals[5] & 0xFFFFFFFFll;
        if (word)
            return gki_set_error(gki_error_of_word(word), "the emit pass left records unwritten (error word %lld): in all-nodes mode 64 "
                                 "consecutive nodes may hold at most 2^32 records between them -- run the graph in chunks "
                                 "(gki_find_params.node_begin / node_end)", (long long)word);
    }
    return GKI_OK;
}

int gki_finder_kernel_ms(gki_finder *f, int which, float *ms) {
    *ms = 0.f;
    if (which == 4) { HIP_TRY(hipEventElapsedTime(ms, f->g->ev_prep0, f->g->ev_prep1)); return GKI_OK; }
    if (which < 0 || which > 3) return gki_set_error(GKI_ERR_BAD_ARG, "kernel id %d", which);
    if (!f->ev_valid[which]) return gki_set_error(GKI_ERR_STATE, "kernel %d has not run", which);
    HIP_TRY(hipEventElapsedTime(ms, f->ev[2 * which], f->ev[2 * which + 1]));
    return GKI_OK;
}

int64_t gki_finder_interior_records(const gki_finder *f) { return f->n_interior_records; }

int64_t gki_find_params_size(void) { return (int64_t)sizeof(gki_find_params); }

}  // extern "C"
