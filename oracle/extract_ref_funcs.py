#!/usr/bin/env python3
"""ORACLE — TEST INFRASTRUCTURE ONLY (build-time helper of `make -C oracle ref`).

Cuts the self-contained, libstdc++-only pieces of the reference's hot-path host logic out of
/root/reference/goldrush_path — by text anchors, at build time, in the build container — into
the git-ignored oracle/_ref/extract/*.inc, where oracle/ref_funcs_shim.cpp (ours: headers,
`using namespace std`, extern "C" wrappers; no reference code) includes them and g++ compiles
the reference's own lines.  Nothing of this is ever copied into the repository.

  goldrush_path.cpp   log_tile_states        whole function
                      sort_by_sec            whole function
                      find_longest_stretch   whole function
                      eval_flanks            whole function
                      calc_num_assigned_tiles  the function's TAIL: everything behind the per-tile
                                             query loop (threshold test, smoothing passes P1-P10,
                                             final count, `return`, closing brace).  The loop in
                                             front of it needs MIBloomFilter / sdsl and cannot be
                                             built here; the shim opens a function whose
                                             parameters carry the names of the locals the tail uses.
                      calc_num_assigned_tiles  two statement ranges INSIDE the per-tile loop that touch only
                                             std::set / std::map locals: the tabulation of a frame's unique
                                             IDs into the tile's count table, and the selection behind the
                                             frame loop (highest count, ties to the smallest ID; the list of
                                             IDs seen more than twice; its sort).  What feeds them — atRank,
                                             getData, the saturation bit — needs MIBloomFilter and stays restated.
                      main                   the hash-universe statements (HASH_UNIVERSE_COEFFICIENT ...)
  MIBloomFilter.hpp   calcOptimalSize        whole (static member) function
Round 4 (what process_read does around the miBF, and what an insert does to a rank):
  MIBloomFilter.hpp   s_mask, setData        the constant's declaration and the whole member function
  MIBFConstructSupport.hpp  insertMIBF (the hash_vec overload)  the `for` statement over the unique ranks: count,
                                             the reservoir test, the call of setData (:271-282)
  goldrush_path.cpp   log_info_struct        the struct
                      log_path_stat          whole function
                      silver_path_check      whole function
                      process_read           the function's BODY (everything behind the signature); the shim opens a
                                             function with the same parameter names over capture classes named like
                                             the reference's own (MIBloomFilter / MIBFConstructSupport: record the
                                             calls, hold m_data / m_counts for the reference's own setData and loop)
"""
import os
import re
import sys


def cut_function(text, signature_re):
    """A top-level function: from the line(s) of its return type / name to the closing brace in column 0."""
    m = re.search(signature_re, text, re.M)
    assert m, signature_re
    start = m.start()
    end = text.index("\n}\n", m.end()) + 3
    return text[start:end]


def main():
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    gp = open(os.path.join(ref, "goldrush_path.cpp")).read()
    mb = open(os.path.join(ref, "MIBloomFilter.hpp")).read()
    pieces = {}
    pieces["log_tile_states"] = cut_function(gp, r"^void\nlog_tile_states\(")
    pieces["sort_by_sec"] = cut_function(gp, r"^bool\nsort_by_sec\(")
    pieces["find_longest_stretch"] = cut_function(gp, r"^std::pair<ssize_t, ssize_t>\nfind_longest_stretch\(")
    pieces["eval_flanks"] = cut_function(gp, r"^std::tuple<bool, size_t, size_t>\neval_flanks\(")
    # tail of calc_num_assigned_tiles
    f0 = re.search(r"^size_t\ncalc_num_assigned_tiles\(", gp, re.M)
    assert f0
    f_end = gp.index("\n}\n", f0.end()) + 3
    body = gp[f0.start():f_end]
    anchor = "    tiles_assigned_all_id_vec[i] = id_counts_vec;\n  }\n"
    assert body.count(anchor) == 1
    tail = body[body.index(anchor) + len(anchor):]
    assert tail.lstrip().startswith("for (size_t i = 0; i < num_tiles; ++i) {") and tail.rstrip().endswith("return num_assigned_tiles;\n}")
    assert "miBF" not in tail and "hashed_values" not in tail
    pieces["smooth_tail"] = tail
    # inside the per-tile loop: tabulation of a frame's unique IDs, and the selection behind the frame loop
    a = body.index("      for (const auto& unique_id :")
    b = body.index("    uint32_t curr_id = 0;\n", a)
    tab = body[a:b]
    assert tab.rstrip().endswith("}\n    }") or tab.rstrip().endswith("}")
    # the range ends with the closing brace of the frame loop: keep only the `for (unique_id ...) { ... }` statement
    tab = tab[: tab.rindex("    }")]
    assert tab.count("id_counts") >= 3 and "miBF" not in tab and tab.strip().endswith("}")
    pieces["vote_tabulate"] = tab
    c = body.index("    sort(id_counts_vec.begin(), id_counts_vec.end(), sort_by_sec);\n", b)
    c = body.index("\n", c) + 1
    sel = body[b:c]
    assert "miBF" not in sel and "curr_id_count" in sel and "id_counts_vec.emplace_back" in sel
    pieces["vote_select"] = sel
    # hash universe statements of main()
    a = gp.index("      static const uint8_t BASES = 4;")
    b = gp.index("hash_universe_base * HASH_UNIVERSE_COEFFICIENT * opt::hash_num;", a)
    b = gp.index("\n", b) + 1
    pieces["hash_universe"] = gp[a:b]
    # calcOptimalSize
    a = mb.index("  static size_t calcOptimalSize(size_t entries,")
    b = mb.index("\n  }\n", a) + 5
    pieces["calc_optimal_size"] = mb[a:b]
    assert "log(1.0 - occupancy)" in pieces["calc_optimal_size"]
    # ---- round 4 -------------------------------------------------------------------------------------------
    cs = open(os.path.join(ref, "MIBFConstructSupport.hpp")).read()
    a = mb.index("  static const T s_mask = ")
    pieces["s_mask_decl"] = mb[a:mb.index("\n", a) + 1]
    a = mb.index("  void setData(uint64_t pos, T id)")
    pieces["set_data"] = mb[a:mb.index("\n  }\n", a) + 5]
    assert "__sync_bool_compare_and_swap(&m_data[pos], oldValue, id)" in pieces["set_data"] and "s_mask" in pieces["set_data"]
    a = cs.index("const std::vector<std::vector<uint64_t>>& hash_vec,")
    a = cs.index("#if _OPENMP\n#pragma omp parallel for\n#endif\n    for (size_t i = 0; i < unique_values.size(); ++i) {", a)
    b = cs.index("\n    }\n", a) + 7
    pieces["reservoir_loop"] = cs[a:b]
    assert "__sync_add_and_fetch(&m_counts[rank], 1)" in pieces["reservoir_loop"] and "miBF.setData(rank, id)" in pieces["reservoir_loop"] and "std::hash<T>{}" in pieces["reservoir_loop"]
    a = gp.index("struct log_info_struct {")
    pieces["log_info_struct"] = gp[a:gp.index("};\n", a) + 3]
    pieces["log_path_stat"] = cut_function(gp, r"^void\nlog_path_stat\(")
    pieces["silver_path_check"] = cut_function(gp, r"^void\nsilver_path_check\(")
    f = cut_function(gp, r"^inline void\nprocess_read\(")
    sig_end = f.index("log_info_struct& log_info)\n{\n") + len("log_info_struct& log_info)\n{\n")
    body = f[sig_end:]
    assert body.lstrip().startswith("if (record.seq.size() < min_seq_len) {") and body.rstrip().endswith("}")
    assert body.count("miBFCS.insertMIBF(") == 2 and body.count("silver_path_check(") == 2 and "calc_num_assigned_tiles(*miBF," in body
    pieces["process_read_body"] = body
    # the members / functions the mini reference (ref_mini_main.cpp) is made of, whole
    def member(text, start_marker, nth=0):
        a = -1
        for _ in range(nth + 1):
            a = text.index(start_marker, a + 1)
        if text[a:].split("\n", 1)[0].rstrip().endswith("}"):  # a one-line member
            return text[a:text.index("\n", a) + 1]
        return text[a:text.index("\n  }\n", a) + 5]
    a = mb.index("  static const T s_antiMask = ")
    pieces["s_antimask_decl"] = mb[a:mb.index("\n", a) + 1]
    pieces["at_rank_vec"] = member(mb, "  bool atRank(const vector<uint64_t>& hashes, vector<uint64_t>& rankPos) const")
    pieces["get_rank_pos_hash"] = member(mb, "  uint64_t getRankPos(const uint64_t hash) const")
    pieces["get_hash_num"] = member(mb, "  unsigned getHashNum() const")
    pieces["size_fn"] = member(mb, "  size_t size() const")
    pieces["get_data_vec"] = member(mb, "  vector<T> getData(const vector<uint64_t>& rankPos) const")
    pieces["reset_id_vector"] = member(mb, "  void reset_ID_vector()")
    for k_ in ("at_rank_vec", "get_rank_pos_hash", "get_data_vec"):
        assert "m_rankSupport" in pieces[k_] or "m_data" in pieces[k_]
    a = cs.index("  void insertMIBF(MIBloomFilter<T>& miBF,\n                  const std::vector<std::vector<uint64_t>>& hash_vec,")
    pieces["insert_mibf_whole"] = cs[a:cs.index("\n  }\n", a) + 5]
    assert "size_t vec_size = hash_vec[0].size();" in pieces["insert_mibf_whole"] and "values.set_empty_key(miBF.size());" in pieces["insert_mibf_whole"]
    pieces["reset_counts"] = member(cs, "  void reset_counts()")
    pieces["calc_num_assigned_tiles"] = cut_function(gp, r"^size_t\ncalc_num_assigned_tiles\(")
    for name, txt in pieces.items():
        with open(os.path.join(out, name + ".inc"), "w") as fp:
            fp.write(txt)


if __name__ == "__main__":
    main()
