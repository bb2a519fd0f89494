// aligner.hpp -- align driver above the C ABI: counterpart of align::Aligner
// (src/align/include/computeAlignments.hpp:142-738).  Same call surface
// (Parameters, MappingBoundaryRow, parseMashmapRow, compute) but records are
// processed in batches so that every wflign stage is one GPU launch set instead
// of one Taskflow task per record (computeAlignments.hpp:398-435).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "fasta.hpp"
#include "wflign_hip.hpp"

namespace align {

enum strnd : int16_t { FWD = 1, AMBIG = 0, REV = -1 };  // skch::strnd (base_types.hpp)

// align::Parameters (align_parameters.hpp:16), fields the live path reads
struct Parameters {
  int threads = 1;
  float min_identity = 0;                 // parse_args.hpp:566
  uint64_t min_alignment_length = 32;     // parse_args.hpp:572
  float min_block_identity = 0.1f;        // parse_args.hpp:583
  int wfa_patching_mismatch_score = 5;    // parse_args.hpp:290-294
  int wfa_patching_gap_opening_score1 = 8;
  int wfa_patching_gap_extension_score1 = 2;
  int wfa_patching_gap_opening_score2 = 24;
  int wfa_patching_gap_extension_score2 = 1;
  uint64_t wflign_max_len_minor = 128000; // windowLength * 128, parse_args.hpp:594
  std::vector<std::string> refSequences;
  std::vector<std::string> querySequences;
  std::string mashmapPafFile;
  std::string pafOutputFile;
  bool emit_md_tag = false;
  bool sam_format = false;
  bool no_seq_in_sam = false;
  bool disable_chain_patching = false;
  uint64_t target_padding = 1000;         // min(w, 5000), parse_args.hpp:608
  uint64_t query_padding = 1000;          // parse_args.hpp:620
  // batching (not in the reference): a batch is bounded by bases and by records.  1536 records of 50 kb fill the device
  // level by level (3 k workgroups per launch) and leave a mid-sized mapping file -- one rank's share of a pangenome,
  // 6 k records -- in enough batches for the workers' host stages to overlap the device (measured: 2 batches of 2.9 k
  // records 0.63 s, 6 batches 0.51 s; a 1.2 k-record file is best left whole)
  uint64_t batch_bases = 160ull << 20;
  size_t batch_records = 1536;
};

// MappingBoundaryRow (align_types.hpp:17)
struct MappingBoundaryRow {
  std::string qId, refId;
  int64_t qStartPos = 0, qEndPos = 0, rStartPos = 0, rEndPos = 0;
  int16_t strand = FWD;
  float mashmap_estimated_identity = 0;
  int32_t chain_id = -1, chain_length = 1, chain_pos = 1;
};

struct Summary {
  uint64_t records = 0;            // "total aligned records"
  uint64_t aligned_bp = 0;         // "total aligned bp" = sum of query spans (computeAlignments.hpp:481,528)
  uint64_t written = 0;            // PAF lines emitted
  uint64_t skipped = 0;            // invalid mapping rows
  uint64_t cells = 0;
  uint64_t cells_tile = 0, tile_launches = 0;  // the tile kernels' share of it (unique cells), their launches and summed launch durations
  double ms_tile = 0;
  double ms_gpu = 0, ms_total = 0;
  double ms_rows = 0, ms_fetch = 0, ms_wflign = 0, ms_text = 0;  // host stages, summed over the batches
  double ms_tags = 0;  // WFM_RECORD_TAGS (a diagnostic): writing the records' tags, lock wait included, summed over the batches
  uint64_t batches = 0;
  std::vector<std::pair<double, double>> busy;  // a worker's device-busy intervals (merged per device at the end of compute())
};

class Aligner {
 public:
  Aligner(const Parameters& p, wfm_handle_t* gpu);
  // one handle per GPU of the node: batches of records go to whichever device is free
  Aligner(const Parameters& p, const std::vector<wfm_handle_t*>& gpus);
  // throws std::runtime_error on malformed rows (computeAlignments.hpp:199-201,292-297)
  static void parseMashmapRow(const std::string& line, MappingBoundaryRow& row, uint64_t target_padding,
                              uint64_t query_padding = 0);
  Summary compute();  // streams param.mashmapPafFile through the GPUs into param.pafOutputFile
  // Aligns mapping lines already in memory on the first GPU; returns the PAF text.
  std::string align_lines(const std::vector<std::string>& lines, Summary& sum);
  // Bytes of the mapping file one batch may hold (~0: no limit beyond batch_records / batch_bases).  file_bytes = 0: the
  // file cannot be rewound or is empty.  rows / row_bytes / row_bases_sum: the first rows looked at (rows = 0: none).
  static uint64_t plan_batch_bytes(uint64_t file_bytes, uint64_t rows, uint64_t row_bytes, uint64_t row_bases_sum, uint64_t batch_records,
                                   uint64_t batch_bases, uint64_t nworkers, uint64_t ngpu, uint64_t min_batches, bool level);

 private:
  std::string align_batch(wfm_handle_t* gpu, std::vector<std::string>& lines, int threads, Summary& sum, uint64_t first_row = 0);
  static uint64_t row_bases(const std::string& line);
  const Parameters& param;
  std::vector<wfm_handle_t*> gpus;
  std::shared_ptr<wfmash_host::FastaStore> ref, query_own;
  const wfmash_host::FastaStore* query = nullptr;
};

}  // namespace align
