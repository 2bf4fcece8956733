"""Graph literals used for golden vectors (data only) + canonical record digest.

The literals are the toy graphs of the reference's own tests (tests/test_kmer_finder.py,
tests/test_critical_graph_paths.py -- cited per entry) and of SURVEY.md Appendix A; each maps to
(node_sequences, edges, linear_ref_nodes, k, finder kwargs).  kwargs key `from_position` selects
`find_only_kmers_starting_at_position(node, offset)` instead of `find()`.
"""
import hashlib
import numpy as np

REFERENCE_TEST_GRAPHS = {
    # README toy with the 3->4 edge restored (BASELINE config 1)
    "readme_c1": ({1: "ACTG", 2: "A", 3: "G", 4: "CCCC"}, {1: [2, 3], 2: [4], 3: [4]}, [1, 2, 4], 5, {}),
    "appA2_one_snp": ({1: "ACGTAC", 2: "G", 3: "T", 4: "CATGCA"}, {1: [2, 3], 2: [4], 3: [4]}, [1, 2, 4], 4, {}),
    "appA2_one_snp_one_node": ({1: "ACGTAC", 2: "G", 3: "T", 4: "CATGCA"}, {1: [2, 3], 2: [4], 3: [4]},
                               [1, 2, 4], 4, {"only_save_one_node_per_kmer": True}),
    "appA3_two_close_snps": ({1: "ACGTAC", 2: "G", 3: "T", 4: "CA", 5: "A", 6: "C", 7: "TGCATG"},
                             {1: [2, 3], 2: [4], 3: [4], 4: [5, 6], 5: [7], 6: [7]}, [1, 2, 4, 5, 7], 4,
                             {"only_save_one_node_per_kmer": True}),
    "appA4_chain": ({1: "AC", 2: "GTACGT"}, {1: [2]}, [1, 2], 3, {}),
    # tests/test_critical_graph_paths.py:53-65 (lossy restart)
    "crit_test4": ({0: "A", 1: "CTTT", 2: "TAAGGGG", 3: "AA", 4: ""}, {0: [1], 1: [2, 4], 2: [3], 4: [3]},
                   [0, 1, 2, 3], 3, {}),
    # tests/test_kmer_finder.py:10-16
    "very_simple": ({0: "AAA", 1: "C", 2: "T", 3: "AAA"}, {0: [1, 2], 2: [3], 1: [3]}, [0, 1, 3], 3, {}),
    # :34-38
    "simple": ({0: "ACTGACTG", 1: "A", 2: "T", 3: "AAAAA", 4: "C", 5: "T", 6: "TGGGGG"},
               {0: [1, 2], 2: [3], 1: [3], 3: [4, 5], 4: [6], 5: [6]}, [0, 1, 3, 4, 6], 3, {}),
    # :51-62 (41 records)
    "nested_paths": ({0: "AAA", 1: "C", 2: "T", 3: "AAAA", 4: "C", 5: "G", 6: "AAA", 7: "TTT"},
                     {0: [1, 2, 7], 1: [3], 2: [3], 3: [4, 5], 4: [6], 5: [6], 7: [6]}, [0, 1, 3, 4, 6], 3, {}),
    # :66-71
    "long_node": ({1: "ATC", 2: "AAAAAAAA", 3: "T", 4: "CTA"}, {1: [2, 3], 2: [4], 3: [4]}, [1, 2, 4], 3, {}),
    # :82-87
    "empty_dummy_nodes": ({1: "ACT", 2: "C", 3: "", 4: "ACT"}, {1: [2, 3], 3: [4], 2: [4]}, [1, 2, 4], 3, {}),
    # :99-104
    "empty_dummy_nodes2": ({1: "AAAAA", 2: "", 3: "CCCCCC"}, {1: [2], 2: [3]}, [1, 3], 3, {}),
    # :118-129 (early stop)
    "empty_dummy_nodes4": ({1: "CC", 2: "", 3: "CCTCTG"}, {1: [2], 2: [3]}, [1, 3], 4, {"from_position": [1, 0]}),
    # :131-136
    "empty_dummy_nodes3": ({1: "AAAAA", 2: "G", 3: "", 4: "CCCCCC"}, {1: [2], 2: [3], 3: [4]}, [1, 2, 4], 3, {}),
    # :149-154
    "multiple_critical_points": (
        {1: "CCCCC", 2: "G", 3: "", 4: "ACT", 5: "", 6: "GC", 7: "A", 8: "T", 9: "G", 10: "GGG"},
        {1: [2, 3], 2: [4], 3: [4], 4: [5, 6], 5: [7], 6: [7], 7: [8, 9], 8: [10], 9: [10]},
        [1, 2, 4, 7, 8, 10], 3, {}),
    # :169-174 / :188-193
    "two_long_nodes1": ({1: "CCCCCCCCCC", 2: "AAAA"}, {1: [2]}, [1, 2], 3, {}),
    "two_long_nodes2": ({1: "CATGCATGCCTG", 2: "CCAAG"}, {1: [2]}, [1, 2], 5, {}),
    # :213-218
    "neighbouring_dummy_nodes": ({1: "ACT", 2: "", 3: "GGG", 4: "", 5: "A", 6: "CCC"},
                                 {1: [2, 3], 2: [4, 5], 3: [4, 5], 4: [6], 5: [6]}, [1, 5, 6], 3, {}),
    # :234-263
    "max_variant_nodes_0": ({1: "ACT", 2: "", 3: "GGG", 4: "", 5: "A", 6: "CCC"},
                            {1: [2, 3], 2: [4, 5], 3: [4, 5], 4: [6], 5: [6]}, [1, 5, 6], 3,
                            {"max_variant_nodes": 0}),
    "max_variant_nodes_1": ({1: "ACT", 2: "", 3: "GGG", 4: "", 5: "A", 6: "CCC"},
                            {1: [2, 3], 2: [4, 5], 3: [4, 5], 4: [6], 5: [6]}, [1, 5, 6], 3,
                            {"max_variant_nodes": 1}),
    # :267-272
    "snp_and_long_node": ({1: "ACTACTACTACT", 2: "G", 3: "C", 4: "GCAGCA"}, {1: [2, 3], 2: [4], 3: [4]},
                          [1, 2, 4], 3, {}),
    # :285-290 (k=31)
    "large_k": ({1: "G" * 100, 2: "C", 3: "T", 4: "G" * 10}, {1: [2, 3], 2: [4], 3: [4]}, [1, 2, 4], 31, {}),
    # :300-323 (early stop + only_store_nodes)
    "kmers_from_position_k3": ({1: "ACTACT", 2: "G", 3: "C", 4: "GCAGCA"}, {1: [2, 3], 2: [4], 3: [4]},
                               [1, 2, 4], 3, {"only_store_nodes": [2, 3], "from_position": [1, 4]}),
    "kmers_from_position_k5": ({1: "ACTACT", 2: "G", 3: "C", 4: "GCAGCA"}, {1: [2, 3], 2: [4], 3: [4]},
                               [1, 2, 4], 5, {"only_store_nodes": [2, 3], "from_position": [1, 5]}),
    # :328-346 (k=31 early stop across an indel)
    "special_case": ({1: "taacccctaacccctaaccctaaccctaac", 2: "", 3: "G", 4: "ccctaaccctaaccctaacccctaacccta"},
                     {1: [2, 3], 2: [4], 3: [4]}, [1, 4], 31,
                     {"only_store_nodes": [2, 3], "from_position": [1, 22]}),
    # :350-364
    "indel": ({1: "ACTGA", 2: "", 3: "C", 4: "GGGGGGGGG"}, {1: [2, 3], 2: [4], 3: [4]}, [1, 4], 9,
              {"only_store_nodes": [2, 3], "from_position": [1, 2]}),
    # :366-382
    "snp_and_indel": ({1: "ACTGAACTG", 2: "A", 3: "C", 4: "GGGG", 5: "", 6: "T", 7: "CCCCCC"},
                      {1: [3, 2], 2: [4], 3: [4], 4: [5, 6], 5: [7], 6: [7]}, [1, 2, 4, 6, 7], 13,
                      {"only_store_nodes": [5, 6], "max_variant_nodes": 5, "from_position": [1, 6]}),
    # :386-399
    "some_case": ({1: "AAAAAACTG", 2: "A", 3: "G", 4: "GC", 5: "T", 6: "C", 7: "TGAGCCCCC", 8: "A", 9: "T",
                   10: "AAAAA"},
                  {1: [2, 3], 2: [4], 3: [4], 4: [5, 6], 5: [7], 6: [7], 7: [8, 9], 9: [10], 8: [10]},
                  [1, 2, 4, 5, 7, 8, 10], 5, {}),
    # :402-410
    "case2": ({0: "AGTAGA", 1: "G", 2: "CT", 3: "A", 4: "CTA", 5: "G", 6: "A", 7: "TCATA"},
              {0: [1, 2], 1: [3], 2: [3], 3: [4], 4: [5, 6], 5: [7], 6: [7], 7: []}, [0, 1, 3, 4, 5, 7], 3, {}),
    # :412-475 (38 ordered records)
    "case1": ({0: "AGTAGA", 1: "G", 2: "CT", 3: "ACTA", 5: "G", 6: "A", 7: "TCATA"},
              {0: [1, 2], 1: [3], 2: [3], 3: [5, 6], 5: [7], 6: [7], 7: []}, [0, 1, 3, 5, 7], 3, {}),
}

# tests/test_kmer_finder.py:412-475: the reference's own ordered expectation for "case1"
CASE1_EXPECTED = [
    ("AGT", 0), ("GTA", 0), ("TAG", 0), ("AGA", 0), ("GAG", 0), ("GAG", 1), ("AGA", 0), ("AGA", 1), ("AGA", 3),
    ("GAC", 1), ("GAC", 3), ("GAC", 0), ("GAC", 2), ("ACT", 0), ("ACT", 2), ("CTA", 2), ("CTA", 3), ("TAC", 2),
    ("TAC", 3), ("ACT", 3), ("CTA", 3), ("TAG", 3), ("TAG", 5), ("AGT", 3), ("AGT", 5), ("AGT", 7), ("GTC", 5),
    ("GTC", 7), ("TAA", 3), ("TAA", 6), ("AAT", 3), ("AAT", 6), ("AAT", 7), ("ATC", 6), ("ATC", 7), ("TCA", 7),
    ("CAT", 7), ("ATA", 7)]

# tests/test_critical_graph_paths.py:6-94 known answers: name -> (graph literal, k, nodes, offsets)
CRITICAL_KATS = {
    "test_k3": (({0: "AAA", 1: "C", 2: "T", 3: "AAA"}, {0: [1, 2], 2: [3], 1: [3]}, [0, 1, 3]), 3, [0, 3], [2, 2]),
    "test_k4": (({0: "AAA", 1: "C", 2: "T", 3: "AAA"}, {0: [1, 2], 2: [3], 1: [3]}, [0, 1, 3]), 4, [], []),
    "test2": (({0: "AAACCCTTTT", 1: "CTTT", 2: "TAAGGGG", 3: "AAA"}, {0: [1, 2], 2: [3], 1: [3]}, [0, 1, 3]),
              3, [0, 3], [2, 2]),
    "test3": (({0: "ACTGACTG", 1: "A", 2: "T", 3: "AAAAA", 4: "C", 5: "T", 6: "TGGGGG"},
               {0: [1, 2], 2: [3], 1: [3], 3: [4, 5], 4: [6], 5: [6]}, [0, 1, 3, 4, 6]), 3, [0, 3, 6], [2, 2, 2]),
    "test4": (({0: "A", 1: "CTTT", 2: "TAAGGGG", 3: "AA", 4: ""}, {0: [1], 1: [2, 4], 2: [3], 4: [3]},
               [0, 1, 2, 3]), 3, [1], [1]),
    "test5": (({0: "ACTGACTG", 1: "A", 2: "T", 3: "AAAAA", 4: "C", 5: "T", 6: "TGGGGG", 100: ""},
               {0: [1, 2, 100], 2: [3], 1: [3], 3: [4, 5], 4: [6], 5: [6], 100: [6]}, [0, 1, 3, 4, 6]),
              3, [0, 6], [2, 2]),
    "test6": (({1: "AAAAA", 2: "", 3: "CCCCCC"}, {1: [2], 2: [3]}, [1, 3]), 3, [1], [2]),
}


def canonical_order(cols):
    """Sort permutation by (start_node, start_offset, kmer, node, allele_frequency)."""
    return np.lexsort((np.asarray(cols["allele_frequencies"]).astype(np.float64),
                       np.asarray(cols["nodes"]).astype(np.int64), np.asarray(cols["kmers"]).astype(np.int64),
                       np.asarray(cols["start_offsets"]).astype(np.int64),
                       np.asarray(cols["start_nodes"]).astype(np.int64)))


def canonical_digest(cols):
    """sha256 over the canonically sorted (kmer i64, node i32, start_node i32, start_offset i16, af f64)."""
    o = canonical_order(cols)
    h = hashlib.sha256()
    h.update(np.asarray(cols["kmers"]).astype(np.int64)[o].tobytes())
    h.update(np.asarray(cols["nodes"]).astype(np.int32)[o].tobytes())
    h.update(np.asarray(cols["start_nodes"]).astype(np.int32)[o].tobytes())
    h.update(np.asarray(cols["start_offsets"]).astype(np.int16)[o].tobytes())
    h.update(np.asarray(cols["allele_frequencies"]).astype(np.float64)[o].tobytes())
    return h.hexdigest()
