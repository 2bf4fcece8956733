"""Sharding of the enumeration over ranks: contiguous ranges of critical-path numbers, exactly the
unit the reference's own process pool uses (command_line_interface.py:588-601,
kmer_finder.py:192-205).  No k-window crosses a critical point, so shards need no halo and no
exchange; the only collective of the build is the all-gather of the finished FlatKmers columns."""
import numpy as np


def critical_path_cuts(graph_arrays, critical_paths, n_shards):
    """Cut points 0 = c_0 <= c_1 <= ... <= c_n = len(critical_paths): shard i runs critical-path numbers
    [c_i, c_{i+1}) (start_at/stop_at of DenseKmerFinder), balanced by the number of bases covered."""
    n_crit = len(critical_paths)
    if n_shards <= 1:
        return [0, n_crit]
    if n_crit == 0:
        # no critical point: the run from the graph start is everything; (1, 1) starts past the last point = nothing
        return [0] + [1] * n_shards
    pos = graph_arrays.seq_start[np.asarray(critical_paths.nodes).astype(np.int64)] + \
        np.asarray(critical_paths.offsets).astype(np.int64)
    total = int(graph_arrays.seq_start[-1])
    targets = (np.arange(1, n_shards) * (total / n_shards)).astype(np.int64)
    # Only the first shard may start at critical path 0: a run with start 0 also owns the search from the graph start
    # (kmer_finder.py:208-211), a second one would emit it again.
    cuts = np.maximum(1, np.searchsorted(pos, targets, side="left"))
    # Never cut at a critical point at offset 0: the reference's run before it passes through such a point and the run
    # starting there is not rewound, so the two chunks would overlap (kmer_finder.py:231-232, 334-341) -- shards must
    # partition the records.  Move the cut to the next critical point with offset >= 1.
    offsets = np.asarray(critical_paths.offsets).astype(np.int64)
    seen = np.concatenate([np.nonzero(offsets >= 1)[0], [n_crit]])
    cuts = [int(seen[np.searchsorted(seen, c, side="left")]) for c in cuts]
    return [0] + cuts + [n_crit]


def shard_range(graph_arrays, critical_paths, rank, world_size):
    """(start_at_critical_path_number, stop_at_critical_path_number) of one rank."""
    cuts = critical_path_cuts(graph_arrays, critical_paths, world_size)
    return cuts[rank], cuts[rank + 1]
