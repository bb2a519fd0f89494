// wflign_hip.hpp -- host side of the align path above the C ABI.
//
// Mirrors the live part of the reference's wflign layer
// (src/common/wflign/src/wflign.cpp:19-483, wflign_swizzle.cpp:7-299,
// wflign_patch.cpp:139-283,2611-2734), re-organised for batches: the reference
// aligns one mapping record per Taskflow task; here every stage of
// do_biwfa_alignment runs over a whole batch so each stage is one
// wfm_align_batch call on the GPU:
//   stage 1    main BiWFA alignment                        (wflign.cpp:136-165)
//   stage 2+3  erosion scans + ends-free head and tail patches, one call for both
//                                                          (wflign.cpp:241-320, 323-418)
//   stage 4    swizzle + PAF record                         (wflign.cpp:423-454)
// and CIGARs are runs (count, op) from the device to the record's text: nothing is expanded to one byte per base.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/wfmash_hip.h"

namespace wflign {

// wflign_penalties_t (wflign_alignment.hpp:21)
struct wflign_penalties_t {
  int match = 0;
  int mismatch = 5;
  int gap_opening1 = 8;
  int gap_extension1 = 2;
  int gap_opening2 = 24;
  int gap_extension2 = 1;
};

using CigarOps = std::vector<std::pair<int, char>>;

// ---- CIGAR helpers (behaviour of the lambdas at wflign.cpp:174-238) ----
CigarOps parse_cigar(const std::string& cigar);
std::string cigar_to_string(const CigarOps& ops);
// long-form op string {M,X,I,D} -> run-length CIGAR with M written as '='
// (wfa_edit_cigar_to_string, wflign_swizzle.cpp:359-382; compress_cigar, wflign.cpp:183-208)
std::string compress_ops(const char* ops, size_t n);
std::string merge_adjacent_ops(const std::string& cigar1, const std::string& cigar2);  // wflign.cpp:211-238
std::string erode_short_matches_in_cigar(const std::string& cigar, int max_match_length = 3,
                                         bool is_head_cigar = true);                  // wflign.cpp:19-106

struct Erosion {
  uint64_t query_eroded = 0, target_eroded = 0;
  size_t erode_end_pos = 0;    // head: byte position in the CIGAR string after the eroded ops
  size_t erode_start_idx = 0;  // tail: index of the first eroded op
};
Erosion scan_head_erosion(const std::string& main_cigar);                       // wflign.cpp:241-276
Erosion scan_tail_erosion(const CigarOps& ops);                                 // wflign.cpp:331-364

// ---- the same on runs (the batch pipeline's form; "position in the text" = index of the run) ----
void ops_from_runs(const uint32_t* runs, size_t n, CigarOps& out);            // wfm_align_batch_rle runs -> (count, op), M as '='
bool erode_short_matches_ops(CigarOps& ops, int max_match_length, bool is_head_cigar);
void append_merged(CigarOps& dst, const CigarOps& src, size_t from, size_t to);
Erosion scan_head_erosion_ops(const CigarOps& ops);                           // erode_end_pos = runs eroded
bool try_swap_start_ops(CigarOps& ops, const char* q, int64_t qn, const char* t, int64_t tn);
bool try_swap_end_ops(CigarOps& ops, const char* q, int64_t qn, const char* t, int64_t tn);

// ---- swizzle (wflign_swizzle.cpp:217-299) ----
std::string try_swap_start_pattern(const std::string& cigar, const std::string& query_seq,
                                   const std::string& target_seq, int64_t query_start, int64_t target_start);
std::string try_swap_end_pattern(const std::string& cigar, const std::string& query_seq,
                                 const std::string& target_seq, int64_t query_start, int64_t target_start);

// ---- PAF record (wflign_patch.cpp:2611-2734) ----
double float2phred(double prob);
struct PafParams {
  float min_identity = 0.0f;
  uint64_t min_alignment_length = 32;
  float min_block_identity = 0.1f;
};
// Returns true and appends one PAF line (with the trailing tab of the
// reference writer, no newline) if the record passes the filters.
bool write_alignment_paf(std::string& out, const std::string& cigar_str, const std::string& query_name,
                         uint64_t query_total_length, uint64_t query_offset, uint64_t query_length, bool query_is_rev,
                         const std::string& target_name, uint64_t target_total_length, uint64_t target_offset,
                         const PafParams& pp, float mashmap_estimated_identity, int32_t chain_id, int32_t chain_length,
                         int32_t chain_pos);

// The same from runs, in the form the align driver writes in the end: fields joined by single tabs, closing newline
// (processMappingRecord re-tokenises the writer's text, computeAlignments.hpp:484-525).
bool write_alignment_paf_ops(std::string& out, const CigarOps& ops, const std::string& query_name, uint64_t query_total_length,
                             uint64_t query_offset, uint64_t query_length, bool query_is_rev, const std::string& target_name,
                             uint64_t target_total_length, uint64_t target_offset, const PafParams& pp,
                             float mashmap_estimated_identity, int32_t chain_id, int32_t chain_length, int32_t chain_pos);

// SAM record (wflign_patch.cpp:2480-2609) incl. the MD:Z tag (write_tag_and_md_string :2397-2478).
// `query` / `target` are the strand-adjusted query window and the target window (target offset 0).
bool write_alignment_sam(std::string& out, const std::string& cigar_str, const std::string& query_name,
                         uint64_t query_offset, bool query_is_rev, const std::string& target_name,
                         uint64_t target_offset, const PafParams& pp, float mashmap_estimated_identity,
                         bool no_seq_in_sam, bool emit_md_tag, const char* query, const char* target,
                         int32_t chain_id, int32_t chain_length, int32_t chain_pos);
std::string md_string(const std::string& cigar, int target_start, const char* target);  // "MD:Z:..."

// ---- batch form of do_biwfa_alignment (wflign.cpp:108-483) ----
struct BiwfaRecord {
  std::string query_name;
  const char* query = nullptr;       // strand-adjusted, upper-case, length query_length
  uint64_t query_total_length = 0, query_offset = 0, query_length = 0;
  bool query_is_rev = false;
  std::string target_name;
  const char* target = nullptr;      // points at rStartPos inside the fetched (padded) buffer
  uint64_t target_total_length = 0, target_offset = 0, target_length = 0;
  uint64_t target_avail = 0;         // bytes readable from `target` (length incl. tail padding)
  float mashmap_estimated_identity = 0;
  int32_t chain_id = -1, chain_length = 1, chain_pos = 1;
  // outputs
  bool ok = false;
  int32_t score = -1;
  CigarOps ops;                      // final CIGAR as runs (after patching + swizzle)
  std::string paf;                   // the record's line incl. newline, as the align driver writes it ("" if filtered)
  uint32_t tags = 0;                 // WFM_PF_* of the main alignment | of the head patch << 8 | of the tail patch << 16 (diagnostics: WFM_RECORD_TAGS)
};

struct BiwfaStats {
  uint64_t cells = 0;
  double ms_gpu = 0;
  uint64_t cells_tile = 0, tile_launches = 0;  // the tile kernels' share: unique cells, launches, summed launch durations
  double ms_tile = 0;
  uint64_t main_failed = 0, head_patches = 0, tail_patches = 0;
  std::string error;                 // the device call's message when do_biwfa_alignment_batch returns < 0
  std::vector<std::pair<double, double>> busy;  // when kernels of this batch's device calls ran (ms on the device's clock, wfm_get_busy_intervals)
};

struct OutputFormat {
  bool paf_format_else_sam = true;   // wflign.cpp:434
  bool no_seq_in_sam = false;
  bool emit_md_tag = false;
  int threads = 1;                   // host threads for the per-record CIGAR / PAF work of a batch
};

int do_biwfa_alignment_batch(wfm_handle_t* h, std::vector<BiwfaRecord>& recs, const wflign_penalties_t& penalties,
                             bool disable_chain_patching, const PafParams& pp, BiwfaStats* stats,
                             const OutputFormat& fmt = OutputFormat());

}  // namespace wflign
