/*
 * gki.h -- C ABI of libgki_hip.so, the MI355X (gfx950) implementation of graph_kmer_index's
 * k-mer enumeration + hashing + index hot path.
 *
 * The reference (ivargr/graph_kmer_index v0.0.29) has no FFI seam for this path: it is plain
 * Python/NumPy.  The entry points below are therefore what a ctypes
 * binding inside the reference would call; each one names the reference function it replaces
 * (paths relative to /root/reference/graph_kmer_index/).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / numpy types.
 *   - every function returns 0 on success or a GKI_ERR_* code; gki_last_error() returns a
 *     thread-local description of the most recent failure.
 *   - "d_" pointers are device (HBM) addresses obtained from gki_malloc(); "h_" pointers are host
 *     memory owned by the caller.  Variable-size results use count -> allocate -> emit.
 *   - one HIP stream per handle; calls on distinct handles may run concurrently, calls on one
 *     handle are serialised by the caller.
 */
#ifndef GKI_H
#define GKI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GKI_OK 0
#define GKI_ERR_HIP 1            /* a HIP runtime call failed (message has the HIP error string) */
#define GKI_ERR_BAD_ARG 2
#define GKI_ERR_NO_DEVICE 3
#define GKI_ERR_WINDOW_TOO_DEEP 4 /* a k-window crosses more nodes than the kernels' stacks hold: GKI_MAX_DEEP_WINDOW_NODES, for
                                     gki_finder_count / emit and for the early-stop search (gki_forward_*) alike -- both
                                     fall back to their slow path above GKI_MAX_WINDOW_NODES.  Also: an emit pass that
                                     had to leave records unwritten (gki_finder_synchronize, gki_forward_emit) */
#define GKI_ERR_STATE 5          /* call order violated (e.g. emit before count) */
#define GKI_ERR_OVERFLOW 6       /* a count does not fit the reference's dtype (e.g. int32 directory) */
#define GKI_ERR_NOT_ONE_REF_SUCC 7 /* the reference's AssertionError kmer_finder.py:402: a reachable window at the
                                     variant limit ends a node that does not have exactly one linear-ref successor */
#define GKI_ERR_OUT_OF_DOMAIN 8  /* gki_index_build_range_from_rows: the records lie outside the row-carrying build's domain (a
                                     group of neighbouring buckets with more than 2^22 records); build that slice from its
                                     columns (gki_index_build_range hands it to the pair-sorting form) */

#define GKI_MAX_WINDOW_NODES 48        /* stacks of the product kernels (scratch) */
#define GKI_MAX_DEEP_WINDOW_NODES 12288 /* stacks of the slow path (a global-memory arena, grown 192, 384, ... levels): beyond
                                          the ~9 990 nodes at which the reference's own recursion ends in a RecursionError */
#define GKI_MAX_K 31             /* kmer_hashing.py:25 `assert k <= 31` */

/* ---------------------------------------------------------------- runtime */
const char *gki_last_error(void);
int gki_device_count(int *count);
int gki_set_device(int device);
int gki_malloc(void **d_ptr, int64_t bytes);
int gki_free(void *d_ptr);
/* gki_malloc / gki_free and the library's own temporaries go through a cache of freed blocks (a fresh hipMalloc of
 * tens of GB costs up to seconds on this stack); gki_trim returns the cache to the device.  GKI_POOL=0 disables it. */
int gki_trim(void);
/* What the device allocator itself cost this process so far: the calls that reached hipMalloc / hipFree (the pool
 * answers the others), the wall time spent inside them, and the bytes parked now.  hipMalloc / hipFree of tens of GB take
 * from tenths of a second to seconds on this stack; a benchmark that allocates inside its timed phases reports the
 * difference of two calls beside them. */
int gki_pool_stats(int64_t *n_device_mallocs, int64_t *n_device_frees, double *ms_in_malloc, double *ms_in_free,
                   int64_t *bytes_parked);
int gki_memcpy_h2d(void *d_dst, const void *h_src, int64_t bytes);
int gki_memcpy_d2h(void *h_dst, const void *d_src, int64_t bytes);
int gki_memcpy_d2d(void *d_dst, const void *d_src, int64_t bytes);
int gki_memset(void *d_dst, int value, int64_t bytes);
int gki_device_synchronize(void);
/* bytes free / total on the current device */
int gki_mem_info(int64_t *free_bytes, int64_t *total_bytes);
/* Order-independent checksums of a device column of n elements of 1, 2, 4 or 8 bytes (zero-extended): sum mod 2^64 and
 * xor.  Lets a caller check that two layouts / a set of shards hold the same multiset without copying it back. */
int gki_column_checksum(const void *d_column, int64_t n, int elem_bytes, uint64_t *sum, uint64_t *xor_fold);
/* FlatKmers.get_new_without_singletons (flat_kmers.py:98-125): d_flags uint8[n] = 1 for every record whose hash also
 * occurs at an earlier position (the first occurrence of each hash gets 0); follow with gki_compact_flat. */
int gki_flag_repeated_kmers(const void *d_kmers, int64_t n, void *d_flags);
/* Stable compaction of FlatKmers columns: keeps record i iff d_flags[i] (uint8) != 0.  Output columns sized by the
 * caller (out_capacity records; the number of set flags is gki_column_checksum's sum for flags of 0/1). */
int gki_compact_flat(const void *d_flags, int64_t n, const void *d_hashes, const void *d_nodes, const void *d_ref_offsets,
                     const void *d_af32, void *d_out_hashes, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                     int64_t out_capacity, int64_t *n_out);

/* ---------------------------------------------------------------- hashing (A1, A8, A10)
 * hash = sum_i base[i] * 4^i, first base least significant, a/n/m=0 c=1 g=2 t=3
 * (flat_kmers.py:134-145, kmer_hashing.py:4-9, snp_kmer_finder.py:19-26). */

/* Every k-window of a numeric sequence: replaces np.convolve(seq, power_array(k), 'valid')
 * (read_kmers.py:67-70, kmer_finder.py:350-352).  d_codes: uint8[n]; d_out: uint64[n-k+1]. */
int gki_hash_sequence(const void *d_codes, int64_t n, int k, void *d_out);

/* Reads laid out back to back (read r = bytes [read_start[r], read_start[r+1]) of ASCII
 * letters); hashes of read r go to d_out[out_start[r] ...), out_start[r] = sum of
 * max(0, len-k+1) of earlier reads (computed by the call, also returned in d_out_start).
 * strand 0: forward (read_kmers.py:21-22); strand 1: reverse complement of the read
 * (read_kmers.py:23-26, Bio.Seq.reverse_complement then the same hash).
 * d_reads uint8[], d_read_start int64[n_reads+1], d_out_start int64[n_reads+1], d_out uint64[]. */
int gki_hash_reads(const void *d_reads, const void *d_read_start, int64_t n_reads, int k, int strand,
                   void *d_out_start, void *d_out, int64_t out_capacity, int64_t *n_out);

/* kmer_hashing.py:24-28 kmer_hashes_to_reverse_complement_hash / :31-36 complement.
 * d_in, d_out: uint64[n] (may alias). */
int gki_reverse_complement(const void *d_in, int64_t n, int k, void *d_out);
int gki_complement(const void *d_in, int64_t n, int k, void *d_out);

/* ---------------------------------------------------------------- graph (the obgraph arrays in HBM)
 * Replaces the per-node obgraph accessor calls of kmer_finder.py:50,62,259,279,350,384,138,143,374.
 * Host arrays in (copied to HBM, 2-bit packed on device):
 *   node_size int32[n_nodes]; seq uint8[n_bases] numeric bases, nodes concatenated in id order;
 *   edge_start int64[n_nodes+1] / edges int32[n_edges]   successors (CSR, get_edges order);
 *   rev_start  int64[n_nodes+1] / rev_edges int32[n_edges] predecessors (CSR);
 *   is_ref uint8[n_nodes] (is_linear_ref_node_or_linear_ref_dummy_node); allele_freq double[n_nodes];
 *   position_base int64[n_nodes] or NULL: PositionId.get(node, 0) (kmer_finder.py:117); NULL = the
 *   exclusive prefix sum of node_size. */
typedef struct gki_graph gki_graph;
int gki_graph_create(gki_graph **out, int64_t n_nodes, const int32_t *h_node_size,
                     const uint8_t *h_seq, int64_t n_bases,
                     const int64_t *h_edge_start, const int32_t *h_edges,
                     const int64_t *h_rev_start, const int32_t *h_rev_edges, int64_t n_edges,
                     const uint8_t *h_is_ref, const double *h_allele_freq,
                     const int64_t *h_position_base);
/* Same, but seq already resident in HBM as uint8[n_bases] (synthetic generators, sharded loaders). */
int gki_graph_create_dseq(gki_graph **out, int64_t n_nodes, const int32_t *h_node_size,
                          const void *d_seq, int64_t n_bases,
                          const int64_t *h_edge_start, const int32_t *h_edges,
                          const int64_t *h_rev_start, const int32_t *h_rev_edges, int64_t n_edges,
                          const uint8_t *h_is_ref, const double *h_allele_freq,
                          const int64_t *h_position_base);
/* Re-run the device-side preparation (2-bit pack, node-start bitmap + rank) from the resident
 * uint8 sequence; gki_graph_create already did it once.  Exposed so a benchmark can time it. */
int gki_graph_prepare(gki_graph *g);
int gki_graph_destroy(gki_graph *g);
int64_t gki_graph_n_bases(const gki_graph *g);

/* critical_graph_paths.py:42-104 CriticalGraphPaths.from_graph (host-side walk over the
 * linear reference; O(#linear nodes)).  h_chrom_start: chromosome start nodes.
 * Outputs sized n_nodes by the caller.  GKI_ERR_BAD_ARG when the reference would raise
 * (not exactly one linear successor :96-100, or offset -1 :104). */
int gki_critical_paths(int64_t n_nodes, const int32_t *h_node_size,
                       const int64_t *h_edge_start, const int32_t *h_edges,
                       const int64_t *h_rev_start, const uint8_t *h_is_ref,
                       const int32_t *h_chrom_start, int n_chrom, int k,
                       uint32_t *h_out_nodes, uint16_t *h_out_offsets, int64_t *n_out);
/* The same walk on the device, over a resident graph: next-node table -> pointer-doubling jump tables -> the path of
 * every chromosome -> depth / bp_since_last_join as prefix sums over the path -> the test, compacted in walk order
 * (csrc/gki_critical.hip).  Same outputs (host arrays sized n_nodes by the caller) and same errors as
 * gki_critical_paths; at most 64 chromosomes. */
int gki_graph_critical_paths(gki_graph *g, const int32_t *h_chrom_start, int n_chrom, int k, uint32_t *h_out_nodes,
                             uint16_t *h_out_offsets, int64_t *n_out);

/* ---------------------------------------------------------------- DenseKmerFinder (A3-A5)
 * Replaces DenseKmerFinder.find() (kmer_finder.py:179-244: search_from :254-347,
 * _process_whole_node :349-381, _search_next_nodes :383-417, _add_kmer :128-168).
 * Output = the reference's record multiset, ordered by end node; inside a node first the windows
 * that reach into predecessors (walk order over the predecessor lists, offset ascending inside one
 * step, nodes of a window ascending), then the offsets whose window lies inside the node. */
/* Record order.  BY_NODE: by end node; inside a node first the windows that reach into predecessors, then the
 * offsets whose window lies inside the node (on a linear graph this is the reference's order).  SPLIT: all
 * inside-the-node records first (by position), then all others (by end node) -- the same multiset in two dense
 * streams, which is what the writes like best; meant for consumers that sort anyway (the index build). */
#define GKI_LAYOUT_BY_NODE 0
#define GKI_LAYOUT_SPLIT 1
/* Per-node flag word (uint16: GKI_NODE_* byte | history bound << 8) of gki_find_params.h_node_flags, computed by
 * gki_classify_nodes (host).  History bound: an upper bound (saturating at 255) on the distinct non-linear-ref nodes any
 * backward path holds in the k bases before the node -- when window + bound stay under the limit, any history will do.  The search of
 * kmer_finder.py:383-417 reaches a window iff SOME history satisfied the variant limit at every step into a node whose
 * entry is not free (free: linear-ref(-dummy) node, or a forced `only_follow_nodes` successor :386-388).  Walking
 * backwards, a history need not be enumerated past a node with GKI_NODE_T: a linear-ref node reachable through >= k
 * linear-ref bases (or from a chromosome start) -- that history adds no variant node to any later window and is always
 * allowed, so it dominates every other one. */
#define GKI_NODE_REF 1       /* is_linear_ref_node_or_linear_ref_dummy_node: not counted by the variant limit */
#define GKI_NODE_FORCED 2    /* member of only_follow_nodes */
#define GKI_NODE_T 4         /* see above */
#define GKI_NODE_SIMPLE 8    /* not T, but a predecessor is: the history through that predecessor dominates */
#define GKI_NODE_NESTED 16   /* not T, no T predecessor: histories are enumerated through its predecessors */
#define GKI_NODE_CHECK 32    /* has successors, none of them forced, and not exactly one linear-ref successor (:402) */
#define GKI_NODE_HFS 64      /* has a forced successor: its edges to other successors do not exist for the search */
#define GKI_NODE_DEAD 128    /* exact: the search never enters the node (no alive edge from an entered node, or no history
                                within the limit): no records, never stepped on */
/* Host, one pass in topological order, O(nodes + edges) plus a backward enumeration for every non-free NESTED node.
 * h_follow: uint8[n_nodes] membership of only_follow_nodes or NULL; h_roots: the nodes a search starts from with no
 * history -- chromosome starts and every critical node (kmer_finder.py:190-232: each critical point starts its own search).
 * h_out_flags uint16[n_nodes].  *general = 1 if any node is NESTED / CHECK / HFS / FORCED or cut off, i.e.
 * gki_finder_count needs the flags; 0 = the graph is in the class where "at most max_variant_nodes variant nodes in
 * the window" is the whole rule and h_node_flags may stay NULL.  GKI_ERR_BAD_ARG if the graph has a cycle. */
int gki_classify_nodes(int64_t n_nodes, const int32_t *h_node_size, const int64_t *h_edge_start, const int32_t *h_edges,
                       const int64_t *h_rev_start, const int32_t *h_rev_edges, const uint8_t *h_is_ref,
                       const uint8_t *h_follow, const int32_t *h_roots, int n_roots, int k, int max_variant_nodes,
                       uint16_t *h_out_flags, int32_t *general);

typedef struct {
    uint32_t struct_size;         /* sizeof(gki_find_params): a binding built against another layout is refused */
    int32_t k;                    /* 1..31 */
    int32_t max_variant_nodes;    /* kmer_finder.py:42 */
    int32_t one_node_per_kmer;    /* only_save_one_node_per_kmer :145-146 */
    int32_t layout;               /* GKI_LAYOUT_BY_NODE or GKI_LAYOUT_SPLIT: order of the records in the output */
    /* end positions processed: (node_begin, off_begin) inclusive .. (node_end, off_end) exclusive,
     * in (node id, offset) order -- the chunking of command_line_interface.py:588-601.  Whole
     * graph: node_begin = 0, off_begin = 0, node_end = n_nodes, off_end = 0. */
    int64_t node_begin, off_begin, node_end, off_end;
    /* per-node critical offset for lossy restarts (0 < c < k-1, SURVEY.md 8a' E1), uint16[n_nodes],
     * 0xFFFF = none; NULL when the graph has none. */
    const uint16_t *h_lossy_crit;
    /* host int32[n_nodes] or NULL.  NULL: node ids increase along every edge and the run is the id range
     * [node_begin, node_end].  Otherwise a topological rank of every node (gki_topological_rank): the run is the nodes
     * whose rank lies between the ranks of node_begin and node_end (node_end == n_nodes: to the end); costs a pass
     * over all nodes instead of over the run's. */
    const int32_t *h_node_rank;
    /* host uint16[n_nodes] from gki_classify_nodes, or NULL when it reported general == 0. */
    const uint16_t *h_node_flags;
    /* host uint8[n_nodes] membership of only_store_nodes (kmer_finder.py:153) or NULL: a record is written only for a
     * node in the set -- except the records of the bulk path (offsets k+2 .. size-2 of a node longer than 2k+3), which
     * the reference writes regardless (:370-374).  Needs h_node_flags (the general kernels apply it). */
    const uint8_t *h_store_nodes;
    /* The same four tables already RESIDENT on the finder's device (same dtypes and lengths); a non-NULL d_ pointer is
     * used as it is and its h_ twin is ignored: nothing is uploaded by gki_finder_count.  For chunked runs (one count per
     * chunk of critical-path numbers, every `index -t N` rank) and for flags that gki_graph_classify_nodes left on the
     * device.  The tables must stay valid until the emit that follows the count has completed.  With d_node_rank the
     * ranks of node_begin and node_end come in rank_begin / rank_end (node_end == n_nodes: INT32_MAX). */
    const uint16_t *d_lossy_crit;
    const int32_t *d_node_rank;
    const uint16_t *d_node_flags;
    const uint8_t *d_store_nodes;
    int32_t rank_begin, rank_end;
} gki_find_params;
/* sizeof(gki_find_params) of this build, for bindings to check at load time. */
int64_t gki_find_params_size(void);

/* gki_classify_nodes on the device, over a resident graph (csrc/gki_classify.hip): the classes are a least fixed point
 * ("entered" spreads from the search roots, the linear-ref run before a node and the variant bound only grow), reached
 * by relaxation sweeps instead of one pass in topological order; whether a nested non-free node has an admissible history
 * is enumerated on the device as well, in rounds with the relaxation (a node without one is dead, which takes histories
 * away from the nodes after it).  *needs_host = 1 when the call cannot finish the job -- a history deeper than 64 nodes
 * or longer than 2^20 steps, more than 48 rounds, or a dependency chain that outlasts the sweep budget -- then nothing
 * is written and the caller runs gki_classify_nodes.
 * h_out_flags (host uint16[n_nodes] or NULL) is filled when the graph is general, or always with always_copy_flags. */
int gki_graph_classify_nodes(gki_graph *g, const uint8_t *h_follow, const int32_t *h_roots, int n_roots, int k,
                             int max_variant_nodes, uint16_t *h_out_flags, int always_copy_flags, int32_t *general,
                             int32_t *needs_host);

/* Host: a topological rank of every node (Kahn; ties by node id), for gki_find_params.h_node_rank.  GKI_ERR_BAD_ARG if
 * the graph has a cycle. */
int gki_topological_rank(int64_t n_nodes, const int64_t *edge_start, const int32_t *edges, int32_t *out_rank);

typedef struct gki_finder gki_finder;
int gki_finder_create(gki_graph *g, gki_finder **out);
int gki_finder_destroy(gki_finder *f);
/* pass 1: counts records (device count pass + prefix sums); returns the total. */
int gki_finder_count(gki_finder *f, const gki_find_params *p, int64_t *n_records);
/* pass 2, FlatKmers layout after FlatKmers.from_multiple_flat_kmers (flat_kmers.py:71-90):
 * hashes uint64, nodes uint32, ref_offsets uint64 (= position id of the END position,
 * kmer_finder.py:116-117), allele_frequencies float32.  Any pointer may be NULL (column skipped). */
int gki_finder_emit_flat(gki_finder *f, void *d_hashes, void *d_nodes, void *d_ref_offsets, void *d_af32);
/* pass 2, get_flat_kmers(v="2") layout (kmer_finder.py:124-126): hashes int64, start_nodes int32,
 * start_offsets int16, nodes int32, allele_frequencies float64. */
int gki_finder_emit_v2(gki_finder *f, void *d_hashes, void *d_start_nodes, void *d_start_offsets,
                       void *d_nodes, void *d_af64);
/* Waits for the emit kernels; GKI_ERR_WINDOW_TOO_DEEP if an emit kernel had to leave records unwritten (all-nodes mode:
 * 64 consecutive nodes hold more than 2^32 records between them -- run such a graph in chunks of nodes). */
int gki_finder_synchronize(gki_finder *f);
/* HIP-event time of the most recent launch of one kernel of this finder (after synchronize).
 * which: 0 count-boundary, 1 emit-interior, 2 emit-boundary, 3 scans (sum), 4 graph prepare. */
int gki_finder_kernel_ms(gki_finder *f, int which, float *ms);
/* Records written by the interior kernel in the last emit (for roofline accounting). */
int64_t gki_finder_interior_records(const gki_finder *f);

/* Forward windows from given start positions: replaces DenseKmerFinder.find_only_kmers_starting_at_position
 * (kmer_finder.py:170-177): for every start position (d_nodes[i], d_offsets[i]) every forward path of exactly k
 * bases gives one window; records carry the END position like all finder records.  count -> emit; records of
 * start position i are [rec_start[i], rec_start[i+1]) in depth-first successor order (the reference's order).
 * d_nodes int32[n_pos], d_offsets int32[n_pos], d_rec_start int64[n_pos+1]; output = the v2 columns.
 * d_follow: uint8[n_nodes] membership of only_follow_nodes (:386-388) or NULL.
 * The first search on a graph (and the first after gki_graph_prepare) builds the search's per-node records: 32 bytes per
 * node, owned by the graph and freed with it.
 * gki_forward_count may leave the finished k-mers in a buffer owned by the graph (193 bytes per start position, both
 * modes); the next gki_forward_emit with the SAME arguments expands them instead of walking again and releases the buffer.
 * Any other sequence of calls (count(A), count(B), emit(A); a second emit) is answered by walking, and the emit pass
 * settles by itself whether it needs the slow path's deeper stacks: the results are the same.  The slow path's arena
 * (up to 5.6 GB) goes back to the library's pool when the emit call returns.  Calls on one graph handle must not overlap
 * in time (two host threads sharing a graph serialise their searches). */
int gki_forward_count(gki_graph *g, int k, int max_variant_nodes, int one_node, const void *d_follow,
                      const void *d_nodes, const void *d_offsets, int64_t n_pos, void *d_rec_start,
                      int64_t *n_records);
int gki_forward_emit(gki_graph *g, int k, int max_variant_nodes, int one_node, const void *d_follow,
                     const void *d_nodes, const void *d_offsets, int64_t n_pos, const void *d_rec_start, void *d_hashes,
                     void *d_start_nodes, void *d_start_offsets, void *d_nodes_out, void *d_af64);

/* ---------------------------------------------------------------- CollisionFreeKmerIndex (A9)
 * Build: replaces CollisionFreeKmerIndex.from_flat_kmers (collision_free_kmer_index.py:423-467)
 * and set_frequencies (:267-293).  Inputs are device columns of n records; outputs are device
 * arrays sized by the caller: hashes_to_index int32[modulo], n_kmers uint32[modulo], and the
 * permuted payload kmers uint64[n], nodes uint32[n], ref_offsets uint64[n], af float32[n],
 * frequencies uint16[n], and optionally the permutation uint32[n] (`sorting` of :435; NULL to skip)
 * so a caller can permute columns of other dtypes itself.  The sort is stable: inside a bucket
 * records keep their input order (np.argsort leaves it unspecified).
 * GKI_ERR_OVERFLOW if n >= 2^31 (the reference's directory is int32, :453). */
int gki_index_build(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                    int64_t n, uint64_t modulo, int skip_frequencies,
                    void *d_hashes_to_index, void *d_n_kmers,
                    void *d_out_kmers, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                    void *d_out_frequencies, void *d_out_permutation);

/* Bucket-range partitioned build.  Part p of n_parts owns the buckets
 * [modulo*p/n_parts, modulo*(p+1)/n_parts) (integer division): every GPU sorts and owns 1/n_parts of the directory,
 * which removes the redundant full sort of the all-gather build and the 2^31-records-per-index limit.
 * gki_partition_by_bucket_range: stable partition of n records (any number: more than 2^31 - 1 are taken in several
 *   passes whose runs of a part are laid out behind each other) by part; h_part_start[n_parts+1] (host) receives the
 *   slice boundaries in the output columns.  n_parts <= 256.  The output columns must not overlap the input.
 * gki_partition_by_bucket_range_chunked: the same with at most max_rows_per_pass records per pass (0: the default);
 *   bounds the pass's temporaries (a histogram entry per part and 4096 records).
 * gki_index_build_range: gki_index_build for the records of one slice: directory arrays int32 / uint32 [n_buckets],
 *   indexed by bucket - bucket_begin; a record whose bucket lies outside the range is GKI_ERR_BAD_ARG. */
int gki_partition_by_bucket_range(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                                  int64_t n, uint64_t modulo, int n_parts, void *d_out_kmers, void *d_out_nodes,
                                  void *d_out_ref_offsets, void *d_out_af32, int64_t *h_part_start);
int gki_partition_by_bucket_range_chunked(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets,
                                          const void *d_af32, int64_t n, uint64_t modulo, int n_parts,
                                          int64_t max_rows_per_pass, void *d_out_kmers, void *d_out_nodes,
                                          void *d_out_ref_offsets, void *d_out_af32, int64_t *h_part_start);
int gki_index_build_range(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                          int64_t n, uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                          void *d_hashes_to_index, void *d_n_kmers,
                          void *d_out_kmers, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                          void *d_out_frequencies, void *d_out_permutation);
/* Grouped partition and build: one sort pass less per slice on ONE GPU (the slices of a whole-genome index: 26 key bits at
 * 7 records per bucket = 7 bits grouped + one pass of 10 + 9 in LDS, against two passes without the grouping; the whole index
 * of the 3 Gbp graph in 187 ms against 195).
 * gki_partition_by_bucket_range_grouped: as _chunked, and the records of part p additionally leave grouped (stably) by
 *   the top group_bits bits of their key in that part (key = bucket - part begin; "top bits" = key >> (bits of the part's
 *   largest key - group_bits), 0 when the key has fewer bits): h_start[(n_parts << group_bits) + 1], entry
 *   p << group_bits | g = first row of group g of part p.  n_parts << group_bits <= 1024.  group_bits = 0: _chunked.
 * gki_index_build_range_grouped: gki_index_build_range for records that arrive so grouped: h_group_start[2^group_bits + 1]
 *   = the part's slice of that table, relative to the part's first row.  Same outputs, element by element. */
int gki_partition_by_bucket_range_grouped(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets,
                                          const void *d_af32, int64_t n, uint64_t modulo, int n_parts, int group_bits,
                                          int64_t max_rows_per_pass, void *d_out_kmers, void *d_out_nodes,
                                          void *d_out_ref_offsets, void *d_out_af32, int64_t *h_start);
int gki_index_build_range_grouped(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                                  int64_t n, uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets,
                                  int skip_frequencies, int group_bits, const int64_t *h_group_start,
                                  void *d_hashes_to_index, void *d_n_kmers,
                                  void *d_out_kmers, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                                  void *d_out_frequencies, void *d_out_permutation);
/* The same pair with the partitioned records kept as ROWS between the two calls -- what the whole-genome index on one GPU
 * uses (collision_free_kmer_index.PartitionedDeviceIndex): with hundreds of digits a run of a few records is one
 * contiguous piece as rows and four short ones as columns.
 * gki_partition_rows_by_bucket_range: d_rows uint64[3 * n] (per record: k-mer, ref offset, node | allele frequency bits
 *   << 32), d_keys uint32[n] (the bucket's offset in its part), both device buffers of the caller; h_start as above.
 * gki_index_build_range_from_rows: the slice build from one part's rows and keys (pointers into those buffers at the
 *   part's first record, n = its records, h_group_start relative to it).  No permutation output.  A key outside
 *   [0, n_buckets) is GKI_ERR_BAD_ARG; GKI_ERR_OUT_OF_DOMAIN when the records need the pair-sorting form. */
int gki_partition_rows_by_bucket_range(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets,
                                       const void *d_af32, int64_t n, uint64_t modulo, int n_parts, int group_bits,
                                       int64_t max_rows_per_pass, void *d_rows, void *d_keys, int64_t *h_start);
int gki_index_build_range_from_rows(const void *d_rows, const void *d_keys, int64_t n, uint64_t modulo,
                                    uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies, int group_bits,
                                    const int64_t *h_group_start, void *d_hashes_to_index, void *d_n_kmers,
                                    void *d_out_kmers, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                                    void *d_out_frequencies);
/* The build has two forms with identical results.  gki_index_build(_range) runs the row-carrying form (the 24-byte
 * payload travels with its key through stable partition passes, the last bits are sorted inside LDS; no random access)
 * and hands over to the pair-sorting form (stable LSD sort of (bucket, index) pairs, then one gather of the payload)
 * when a group of 2^L neighbouring buckets holds more than 2^22 records (an index that is a handful of buckets).
 * gki_index_build_pairs runs the pair-sorting form directly: same arguments, same outputs. */
int gki_index_build_pairs(const void *d_kmers, const void *d_nodes, const void *d_ref_offsets, const void *d_af32,
                          int64_t n, uint64_t modulo, uint64_t bucket_begin, uint64_t n_buckets, int skip_frequencies,
                          void *d_hashes_to_index, void *d_n_kmers,
                          void *d_out_kmers, void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32,
                          void *d_out_frequencies, void *d_out_permutation);

/* ---------------------------------------------------------------- ReverseKmerIndex
 * Replaces ReverseKmerIndex.from_flat_kmers (reverse_kmer_index.py:47-60): records stably sorted by
 * node.  Inputs: device columns nodes uint32[n], kmers uint64[n], ref_offsets uint64[n]; n_nodes =
 * max node id + 1.  Outputs (device, sized by the caller): index_positions uint32[n_nodes] (first
 * record of the node, 0 if none), n_hashes uint16[n_nodes] (records of the node, modulo 2^16 as the
 * NumPy assignment at :56 stores it), and the permuted kmers / ref_offsets uint64[n].
 * Built by the row-carrying form of the index build with the node id as the key; the pair-sorting form it replaced
 * answers what that form declines, and every call when the environment holds GKI_REVERSE_FORM=pairs (read per call;
 * for the parity tests of that path).  A node id >= n_nodes is GKI_ERR_BAD_ARG. */
int gki_reverse_index_build(const void *d_nodes, const void *d_kmers, const void *d_ref_offsets, int64_t n,
                            int64_t n_nodes, void *d_index_positions, void *d_n_hashes, void *d_out_kmers,
                            void *d_out_ref_offsets);

/* Probe: replaces a loop of CollisionFreeKmerIndex.get (:303-315) over q queries
 * (get_nodes_and_ref_offsets_from_multiple_kmers :354-376).  count -> emit.
 * d_hit_start int64[q+1]: hits of query i are [hit_start[i], hit_start[i+1]) in bucket order;
 * a query with no hit, or whose first hit has frequency > max_hits, contributes none.
 * Emit columns (any may be NULL): nodes uint32, ref_offsets uint64, query index int64
 * (`read_offsets` of :365), frequencies uint16, af float32, and the hit's position int64 in the
 * payload arrays (lets a caller gather from its own columns of any dtype). */
typedef struct {
    const void *d_hashes_to_index, *d_n_kmers, *d_kmers, *d_nodes, *d_ref_offsets, *d_frequencies, *d_af32;
    uint64_t modulo;
    int64_t n;
    /* bucket-range slice (gki_index_build_range): the directory arrays hold buckets [bucket_begin, bucket_begin +
     * n_buckets) only and a k-mer whose bucket lies outside is a miss; n_buckets == 0 means the whole [0, modulo). */
    uint64_t bucket_begin, n_buckets;
} gki_index_view;
int gki_index_lookup_count(const gki_index_view *ix, const void *d_queries, int64_t q, int64_t max_hits,
                           void *d_hit_start, int64_t *n_hits);
/* Node counting fused with the probe: d_counts[node] += 1 (uint32[n_counts], caller-zeroed or accumulated across
 * batches) for every hit of every query -- what kmer_mapper's map_kmers_to_graph_index gives KAGE
 * (collision_free_kmer_index.py:210-212) and CounterKmerIndex.get_node_counts (:39-40). */
int gki_index_count_nodes(const gki_index_view *ix, const void *d_queries, int64_t q, int64_t max_hits,
                          void *d_counts, int64_t n_counts);
int gki_index_lookup_emit(const gki_index_view *ix, const void *d_queries, int64_t q, int64_t max_hits,
                          const void *d_hit_start, void *d_hit_nodes, void *d_hit_ref_offsets,
                          void *d_hit_query, void *d_hit_frequencies, void *d_hit_af32, void *d_hit_position);

/* CollisionFreeKmerIndex.get (:303-315) for a few k-mers from HOST memory (q <= 64): one launch, one wave per query,
 * queries and results travel through pinned host memory the kernel addresses directly.  h_n_hits[i] = number of hits of
 * query i (0 for a miss or when the first hit's frequency exceeds max_hits); its first min(h_n_hits[i], 1024,
 * capacity_per_query) payload positions, in bucket order, are h_positions[i * capacity_per_query ...].  A caller that
 * sees more hits than it got positions for uses the batched gki_index_lookup_* pair. */
int gki_index_get_small(const gki_index_view *ix, const uint64_t *h_queries, int q, int64_t max_hits,
                        int64_t *h_n_hits, int64_t *h_positions, int64_t capacity_per_query);

/* ---------------------------------------------------------------- probe table (read-side hot loop)
 * A device-only re-layout of an index for counting: dir uint2[modulo] = {first record, count (16 bit, saturating)
 * | 16-bit fingerprint set << 16}, rows uint4[n] = {kmer, node, frequency}: one random 64-byte sector per query,
 * one more per candidate bucket (the reference layout touches five arrays per hit).  Costs modulo*8 + n*16 bytes
 * of HBM.  The view's d_n_kmers must outlive the probe (read for buckets of >= 65535 records).
 * gki_probe_count_nodes       = gki_index_count_nodes on the table; *n_hits (nullable) = hits counted.
 * gki_probe_reads_count_nodes = ReadKmers hashing (read_kmers.py:21-26,70) fused with the probe: reads are ASCII
 *   letters uint8[] with int64 read_start[n_reads+1]; strands bit 0 = forward k-mers, bit 1 = k-mers of the
 *   reverse-complemented read; no k-mer array is materialised.  *n_kmers (nullable) = k-mers probed. */
typedef struct gki_probe gki_probe;
int gki_probe_create(const gki_index_view *ix, gki_probe **out);
int gki_probe_destroy(gki_probe *p);
int gki_probe_count_nodes(gki_probe *p, const void *d_queries, int64_t q, int64_t max_hits, void *d_counts,
                          int64_t n_counts, int64_t *n_hits);
/* gki_index_lookup_count / _emit on the table: d_hit_start int64[q+1] as above; the emit pass writes the query index and
 * the hit's position in the payload arrays (int64 each), from which a caller gathers any column. */
int gki_probe_lookup_count(gki_probe *p, const void *d_queries, int64_t q, int64_t max_hits, void *d_hit_start,
                           int64_t *n_hits);
int gki_probe_lookup_emit(gki_probe *p, const void *d_queries, int64_t q, int64_t max_hits, const void *d_hit_start,
                          void *d_hit_query, void *d_hit_position);
/* `kmer in index` (collision_free_kmer_index.py:295-296) for q k-mers: d_flags uint8[q] = 1 / 0.  The whitelist test of
 * DenseKmerFinder (kmer_finder.py:130-132, 362-365); follow with gki_compact_flat. */
int gki_probe_contains(gki_probe *p, const void *d_queries, int64_t q, void *d_flags);
int gki_probe_reads_count_nodes(gki_probe *p, const void *d_reads, const void *d_read_start, int64_t n_reads, int k,
                                int strands, int64_t max_hits, void *d_counts, int64_t n_counts, int64_t *n_kmers,
                                int64_t *n_hits);

/* Measurement aid: independent random 8-byte loads per second the device sustains from a table of table_bytes (every
 * load that misses L2 is one 64-byte request -- the unit the probe kernels are bound by, not bytes).  Runs n_loads loads
 * twice and reports the second launch.  bench.py reports the read-side rate as a fraction of this. */
int gki_measure_random_loads(int64_t table_bytes, int64_t n_loads, double *loads_per_s);
/* The store ceiling of THIS device for the FlatKmers column pattern: writes n records of the four columns (8 + 4 + 8 + 4
 * bytes; the caller's buffers, e.g. the output columns before a run) with nothing else in the kernel, three launches,
 * best of the last two; bytes_per_s = 24 n / time.  bench.py prints it beside the 8 TB/s spec peak. */
int gki_measure_store_bw(void *d_hashes, void *d_nodes, void *d_ref_offsets, void *d_af32, int64_t n, double *bytes_per_s);
/* Device self-test of the kernels' wave prefix sum (DPP row shifts and broadcasts, csrc/gki_common.h) against the shuffle
 * form on random and extreme inputs: *n_bad = lanes that disagree (0 on a correct build).  No reference counterpart. */
int gki_selftest_wave_scan(int64_t *n_bad);
/* Read simulator on the device (benchmark input for BASELINE configs[4]): n_reads reads of read_len
 * letters (ASCII ACGT) sampled from a haplotype (uint8 codes 0..3 in HBM): uniform start, either strand, every base
 * substituted with p_substitution, a read replaced by uniform random letters with p_random_read.  Counter-based: read
 * first_read + q is a pure function of (seed, its index), so batches of one run and a NumPy restatement agree. */
int gki_simulate_reads(const void *d_haplotype, int64_t hap_len, int64_t n_reads, int read_len, uint64_t seed,
                       double p_substitution, double p_random_read, int64_t first_read, void *d_letters);

/* ---------------------------------------------------------------- multi-GPU exchange (RCCL over xGMI)
 * One process per GPU.  The reference gathers its per-process FlatKmers by pickling them through a
 * process pool and concatenating (command_line_interface.py:607-614, flat_kmers.py:71-90); here every
 * rank's finished columns are exchanged once, shard r landing at offset sum(counts[:r]), so that every
 * GPU holds the concatenation in rank order.  The 128-byte id comes from rank 0 and travels over the
 * caller's control plane (parallel.SocketControlPlane: plain TCP to rank 0); h_counts[world] likewise. */
/* Ordering: every exchange entry point first waits for all work already submitted to the device (the columns come
 * from finder / partition streams), runs on the communicator's own stream and returns when the exchange has completed;
 * a caller needs no synchronisation of its own on either side. */
#define GKI_COMM_ID_BYTES 128
typedef struct gki_comm gki_comm;
int gki_comm_get_unique_id(void *h_id);
int gki_comm_create(gki_comm **out, int world_size, int rank, const void *h_id);
int gki_comm_destroy(gki_comm *c);
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank). */
int gki_comm_info(gki_comm *c, int *rccl_world, int *rccl_rank);
/* Ranks that share ONE device.  RCCL refuses a communicator with two ranks on the same device (ncclInvalidUsage), so
 * a multi-rank run on a single GPU exchanges through HIP IPC instead: gki_device_bus_id tells the ranks apart
 * (PCI bus id of the current device), gki_ipc_export gives the 64-byte handle of the allocation that holds d_ptr and
 * d_ptr's offset inside it, gki_ipc_open maps a peer's allocation (returns its base), gki_ipc_close unmaps it.  The
 * handles travel over the caller's control plane (parallel.SharedDeviceComm). */
#define GKI_IPC_HANDLE_BYTES 64
int gki_device_bus_id(char *buf, int len);
int gki_ipc_export(const void *d_ptr, void *h_handle, int64_t *offset);
int gki_ipc_open(const void *h_handle, void **d_base);
int gki_ipc_close(void *d_base);
/* All-to-all(v) of FlatKmers columns for the bucket-range partitioned build: the send columns hold this rank's records
 * partitioned by destination, slice r = [h_send_start[r], h_send_start[r+1]); what rank r sends to this rank lands at
 * [h_recv_start[r], h_recv_start[r+1]) of the receive columns (both tables host int64[world+1]; the counts travel
 * over the caller's control plane first). */
int gki_comm_alltoall_flat(gki_comm *c, const int64_t *h_send_start, const void *d_hashes, const void *d_nodes,
                           const void *d_ref_offsets, const void *d_af32, const int64_t *h_recv_start, void *d_out_hashes,
                           void *d_out_nodes, void *d_out_ref_offsets, void *d_out_af32);
/* In-place sum over ranks of a uint32 device array (node counts of a bucket-partitioned lookup). */
int gki_comm_allreduce_u32(gki_comm *c, void *d_buf, int64_t n);
int gki_comm_allgather_flat(gki_comm *c, const int64_t *h_counts, const void *d_hashes, const void *d_nodes,
                            const void *d_ref_offsets, const void *d_af32, void *d_out_hashes, void *d_out_nodes,
                            void *d_out_ref_offsets, void *d_out_af32);

#ifdef __cplusplus
}
#endif
#endif
