#include "aligner.hpp"
#include "parallel.hpp"
#include "../csrc/wfa_handle.h"

#include <algorithm>
#include <thread>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace align {

namespace {

std::vector<std::string> tokenize(const std::string& s) {  // whitespace split (computeAlignments.hpp:54-72)
  std::vector<std::string> t;
  size_t pos = 0;
  while (pos < s.size()) {
    while (pos < s.size() && std::isspace((unsigned char)s[pos])) ++pos;
    if (pos >= s.size()) break;
    const size_t st = pos;
    while (pos < s.size() && !std::isspace((unsigned char)s[pos])) ++pos;
    t.emplace_back(s.substr(st, pos - st));
  }
  return t;
}

std::vector<std::string> split(const std::string& s, char d) {  // computeAlignments.hpp:35-51
  std::vector<std::string> r;
  size_t pos = 0, f;
  while ((f = s.find(d, pos)) != std::string::npos) { r.emplace_back(s.substr(pos, f - pos)); pos = f + 1; }
  if (pos <= s.size()) r.emplace_back(s.substr(pos));
  return r;
}

bool is_a_number(const std::string& s) {  // utils.cpp:9-11
  return !s.empty() && s.find_first_not_of("0123456789.") == std::string::npos && std::count(s.begin(), s.end(), '.') < 2;
}

// makeUpperCaseAndValidDNA (commonFunc.hpp:132-142)
void upper_valid_dna(std::string& s) {
  for (auto& c : s) {
    if (c > 96 && c < 123) c -= 32;
    if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) c = 'N';
  }
}

// reverseComplement (commonFunc.hpp:74-83) on validated DNA
std::string revcomp(const std::string& s) {
  std::string r(s.size(), 'N');
  for (size_t i = 0; i < s.size(); ++i) {
    char c = s[i], o;
    switch (c) { case 'A': o = 'T'; break; case 'C': o = 'G'; break; case 'G': o = 'C'; break; case 'T': o = 'A'; break; default: o = c; }
    r[s.size() - 1 - i] = o;
  }
  return r;
}

struct Fetched {
  MappingBoundaryRow row;
  std::string ref;    // padded reference window
  std::string qry;    // strand-adjusted query window
  uint64_t ref_start = 0, ref_total = 0, q_total = 0;
  uint64_t row_no = 0;  // the row's number in the mapping file
};

}  // namespace

Aligner::Aligner(const Parameters& p, wfm_handle_t* g) : Aligner(p, std::vector<wfm_handle_t*>{g}) {}

Aligner::Aligner(const Parameters& p, const std::vector<wfm_handle_t*>& g) : param(p), gpus(g) {
  if (gpus.empty() || std::find(gpus.begin(), gpus.end(), nullptr) != gpus.end()) throw std::runtime_error("[wfmash::align] no GPU handle");
  if (param.refSequences.size() != 1 || param.querySequences.size() != 1)
    throw std::runtime_error("[wfmash::align] exactly one target and one query FASTA are expected");
  // (shared with whoever has the file open -- a map phase just before this leaves its sequences loaded: fasta.hpp, keep_until_next)
  ref = wfmash_host::open_shared(param.refSequences.front());
  if (param.querySequences.front() == param.refSequences.front()) query = ref.get();
  else { query_own = wfmash_host::open_shared(param.querySequences.front()); query = query_own.get(); }
}

void Aligner::parseMashmapRow(const std::string& line, MappingBoundaryRow& row, uint64_t target_padding, uint64_t query_padding) {
  const auto tokens = tokenize(line);
  if (tokens.size() < 13)
    throw std::runtime_error("[wfmash::align::parseMashmapRow] Error! Invalid mashmap mapping record: " + line);
  const auto idv = split(tokens[12], ':');
  const float mm_id = (!idv.empty() && is_a_number(idv.back())) ? std::stof(idv.back()) : 0.70f;  // fixed::percentage_identity
  int64_t chain_id = -1, chain_length = 1, chain_pos = 1;
  if (tokens.size() > 14) {
    const auto cv = split(tokens[14], ':');
    if (cv.size() == 3 && cv[0] == "ch" && cv[1] == "Z") {
      const auto parts = split(cv[2], '.');
      if (parts.size() == 3) {  // ch:Z:id.pos.len as the mapper writes it (mappingOutput.hpp:121)
        chain_id = std::stoll(parts[0]); chain_pos = std::stoll(parts[1]); chain_length = std::stoll(parts[2]);
      }
    }
  }
  row.qId = tokens[0];
  row.qStartPos = std::stoll(tokens[2]);
  row.qEndPos = std::stoll(tokens[3]);
  row.strand = tokens[4] == "+" ? FWD : REV;
  row.refId = tokens[5];
  const uint64_t ref_len = std::stoull(tokens[6]);
  row.chain_id = (int32_t)chain_id; row.chain_length = (int32_t)chain_length; row.chain_pos = (int32_t)chain_pos;
  uint64_t rs = (uint64_t)std::stoll(tokens[7]), re = (uint64_t)std::stoll(tokens[8]);
  uint64_t qs = (uint64_t)row.qStartPos, qe = (uint64_t)row.qEndPos;
  const uint64_t query_len = std::stoull(tokens[1]);
  if (target_padding > 0) {
    rs = rs >= target_padding ? rs - target_padding : 0;
    re = re + target_padding <= ref_len ? re + target_padding : ref_len;
  }
  if (query_padding > 0) {
    // padding only at the chain ends, and only STORED for the last piece (computeAlignments.hpp:268-289)
    if (chain_pos == 1) qs = qs >= query_padding ? qs - query_padding : 0;
    if (chain_pos == chain_length) {
      qe = qe + query_padding <= query_len ? qe + query_padding : query_len;
      row.qStartPos = (int64_t)qs;
      row.qEndPos = (int64_t)qe;
    }
  }
  if (rs >= ref_len || re > ref_len)
    throw std::runtime_error("[wfmash::align::parseMashmapRow] Error! Coordinates exceed reference length: " +
                             std::to_string(rs) + "-" + std::to_string(re) + " (ref_len=" + std::to_string(ref_len) + ")");
  row.rStartPos = (int64_t)rs;
  row.rEndPos = (int64_t)re;
  row.mashmap_estimated_identity = mm_id;
}

// One batch of mapping rows through the wflign pipeline on one GPU: createSeqRecord + processAlignment
// (computeAlignments.hpp:582-723) for every row, then the records' output text in row order.
namespace {
// WFM_RECORD_TAGS=<file>: one line per aligned record -- the row's number in the mapping file (0-based, empty lines not counted), the record's
// WFM_PF_* bits (main alignment | head patch << 8 | tail patch << 16) and its score.  The parity tests and bench.py draw their samples from it.
std::mutex g_tags_mu;
void write_record_tags(const std::vector<uint64_t>& row_no, const std::vector<wflign::BiwfaRecord>& recs) {
  const char* path = getenv("WFM_RECORD_TAGS");
  if (!path || !*path) return;
  std::lock_guard<std::mutex> lk(g_tags_mu);
  if (FILE* f = fopen(path, "a")) {
    for (size_t k = 0; k < recs.size(); ++k) fprintf(f, "%llu\t%u\t%d\t%d\n", (unsigned long long)row_no[k], recs[k].tags, recs[k].score, recs[k].ok ? 1 : 0);
    fclose(f);
  }
}
}  // namespace

std::string Aligner::align_batch(wfm_handle_t* gpu_handle, std::vector<std::string>& lines, int threads, Summary& sum, uint64_t first_row) {
  std::string out;
  const bool dbg = getenv("WFM_DEBUG") != nullptr;
  const auto tb0 = std::chrono::steady_clock::now();
  auto since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
  wflign::wflign_penalties_t pen;
  pen.match = 0;
  pen.mismatch = param.wfa_patching_mismatch_score;
  pen.gap_opening1 = param.wfa_patching_gap_opening_score1;
  pen.gap_extension1 = param.wfa_patching_gap_extension_score1;
  pen.gap_opening2 = param.wfa_patching_gap_opening_score2;
  pen.gap_extension2 = param.wfa_patching_gap_extension_score2;
  wflign::PafParams pp;
  pp.min_identity = param.min_identity;
  pp.min_alignment_length = param.min_alignment_length;
  pp.min_block_identity = param.min_block_identity;

  // rows first (cheap, in order), then the sequence fetches of the whole batch on `threads` threads
  std::vector<Fetched> rows;
  uint64_t row_no = first_row;
  for (const std::string& line : lines) {
    if (line.empty()) continue;
    Fetched f;
    f.row_no = row_no++;
    try {
      parseMashmapRow(line, f.row, param.target_padding, param.query_padding);
      const int64_t ref_size = ref->seq_len(f.row.refId);
      if (ref_size < 0) throw std::runtime_error("Reference sequence not found: " + f.row.refId);
      const int64_t query_size = query->seq_len(f.row.qId);
      if (query_size < 0) throw std::runtime_error("Query sequence not found: " + f.row.qId);
      f.ref_total = (uint64_t)ref_size; f.q_total = (uint64_t)query_size;
      rows.push_back(std::move(f));
    } catch (const std::exception& e) {
      std::cerr << "[wfmash::align] Error processing record: " << e.what() << std::endl;
      sum.skipped++;
    }
  }
  std::vector<std::string>().swap(lines);
  const double ms_parse = since(tb0);
  const auto tb1 = std::chrono::steady_clock::now();
  std::vector<std::string> fetch_error(rows.size());
  {
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (size_t k; (k = next.fetch_add(1)) < rows.size();) {
        Fetched& f = rows[k];
        try {
          const int64_t ref_size = (int64_t)f.ref_total;
          const uint64_t head_pad = (uint64_t)f.row.rStartPos >= param.wflign_max_len_minor ? param.wflign_max_len_minor : (uint64_t)f.row.rStartPos;
          const uint64_t tail_pad = (uint64_t)(ref_size - f.row.rEndPos) >= param.wflign_max_len_minor ? param.wflign_max_len_minor : (uint64_t)(ref_size - f.row.rEndPos);
          f.ref = ref->fetch(f.row.refId, f.row.rStartPos - (int64_t)head_pad, f.row.rEndPos + (int64_t)tail_pad - 1);
          if (f.ref.empty()) throw std::runtime_error("Failed to fetch reference sequence");
          std::string q = query->fetch(f.row.qId, f.row.qStartPos, f.row.qEndPos - 1);
          if (q.empty()) throw std::runtime_error("Failed to fetch query sequence");
          f.ref_start = (uint64_t)f.row.rStartPos - head_pad;
          upper_valid_dna(f.ref);
          upper_valid_dna(q);
          f.qry = f.row.strand == FWD ? std::move(q) : revcomp(q);
        } catch (const std::exception& e) {
          fetch_error[k] = e.what();
          if (fetch_error[k].empty()) fetch_error[k] = "error";
        }
      }
    };
    const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), rows.size());
    // (helpers from the process's pool, parallel.hpp: every one of them runs `work`, which shares the rows out by its own counter)
    wfmash_host::parallel_for((size_t)nt, nt, [&](size_t) { work(); });
  }
  std::vector<Fetched> fetched;
  fetched.reserve(rows.size());
  for (size_t k = 0; k < rows.size(); ++k) {
    if (!fetch_error[k].empty()) {
      std::cerr << "[wfmash::align] Error processing record: " << fetch_error[k] << std::endl;
      sum.skipped++;
      continue;
    }
    fetched.push_back(std::move(rows[k]));
  }
  if (fetched.empty()) return out;
  std::vector<wflign::BiwfaRecord> recs(fetched.size());
  for (size_t k = 0; k < fetched.size(); ++k) {
    const Fetched& f = fetched[k];
    wflign::BiwfaRecord& r = recs[k];
    r.query_name = f.row.qId;
    r.query = f.qry.data();
    r.query_total_length = f.q_total;
    r.query_offset = (uint64_t)f.row.qStartPos;
    r.query_length = f.qry.size();
    r.query_is_rev = f.row.strand != FWD;
    r.target_name = f.row.refId;
    const uint64_t skip = (uint64_t)f.row.rStartPos - f.ref_start;
    r.target = f.ref.data() + skip;
    r.target_total_length = f.ref_total;
    r.target_offset = (uint64_t)f.row.rStartPos;
    r.target_length = (uint64_t)(f.row.rEndPos - f.row.rStartPos);
    r.target_avail = f.ref.size() - skip;
    r.mashmap_estimated_identity = f.row.mashmap_estimated_identity;
    r.chain_id = f.row.chain_id; r.chain_length = f.row.chain_length; r.chain_pos = f.row.chain_pos;
  }
  const double ms_fetch = since(tb1);
  const auto tb2 = std::chrono::steady_clock::now();
  wflign::BiwfaStats st;
  wflign::OutputFormat fmt;
  fmt.paf_format_else_sam = !param.sam_format;
  fmt.no_seq_in_sam = param.no_seq_in_sam;
  fmt.emit_md_tag = param.emit_md_tag;
  fmt.threads = threads;
  const int rc = wflign::do_biwfa_alignment_batch(gpu_handle, recs, pen, param.disable_chain_patching, pp, &st, fmt);
  if (rc < 0) throw std::runtime_error(std::string("[wfmash::align] GPU alignment failed: ") + st.error);
  sum.cells += st.cells; sum.ms_gpu += st.ms_gpu;
  sum.cells_tile += st.cells_tile; sum.tile_launches += st.tile_launches; sum.ms_tile += st.ms_tile;
  sum.busy.insert(sum.busy.end(), st.busy.begin(), st.busy.end());
  if (getenv("WFM_RECORD_TAGS")) {
    const auto tt = std::chrono::steady_clock::now();
    std::vector<uint64_t> rn(fetched.size());
    for (size_t k = 0; k < fetched.size(); ++k) rn[k] = fetched[k].row_no;
    write_record_tags(rn, recs);
    sum.ms_tags += since(tt);
  }
  const double ms_biwfa = since(tb2);
  const auto tb3 = std::chrono::steady_clock::now();
  for (size_t k = 0; k < recs.size(); ++k) {
    sum.records++;
    sum.aligned_bp += (uint64_t)(fetched[k].row.qEndPos - fetched[k].row.qStartPos);
    if (recs[k].paf.empty()) continue;
    // PAF: the record is already in the form processMappingRecord gives it (fields re-joined with single tabs,
    // computeAlignments.hpp:484-525); SAM: no cg:Z: field -> the writer's line passes through unchanged
    out += recs[k].paf;
    sum.written++;
  }
  const double ms_text = since(tb3);
  sum.ms_rows += ms_parse; sum.ms_fetch += ms_fetch; sum.ms_wflign += ms_biwfa; sum.ms_text += ms_text; sum.batches++;
  if (dbg)
    fprintf(stderr, "[wfmash::align] batch of %zu records on %d threads: rows %.1f ms, fetch %.1f ms, wflign %.1f ms (device busy %.1f), text %.1f ms\n",
            recs.size(), threads, ms_parse, ms_fetch, ms_biwfa, st.ms_gpu, ms_text);
  return out;
}

std::string Aligner::align_lines(const std::vector<std::string>& lines, Summary& sum) {
  std::string out;
  size_t i = 0;
  while (i < lines.size()) {
    std::vector<std::string> batch;
    uint64_t bases = 0;
    while (i < lines.size() && batch.size() < param.batch_records && bases < param.batch_bases) {
      bases += row_bases(lines[i]);
      batch.push_back(lines[i++]);
    }
    const uint64_t first_row = i - batch.size();
    out += align_batch(gpus.front(), batch, param.threads, sum, first_row);
  }
  return out;
}

// bases a mapping row will align, padding aside (the reader sizes batches by it)
uint64_t Aligner::plan_batch_bytes(uint64_t file_bytes, uint64_t rows, uint64_t row_bytes, uint64_t row_bases_sum, uint64_t batch_records,
                                   uint64_t batch_bases, uint64_t nworkers, uint64_t ngpu, uint64_t min_batches, bool level) {
  if (file_bytes == 0) return ~0ull;
  uint64_t want = ngpu > 1 ? 8 * ngpu : std::max<uint64_t>(1, min_batches);
  if (level && nworkers > 1 && rows > 0 && row_bytes > 0) {
    const double est_rows = (double)file_bytes / ((double)row_bytes / (double)rows);
    const double est_bases = est_rows * ((double)row_bases_sum / (double)rows);
    const uint64_t need = (uint64_t)std::ceil(std::max(est_rows / (double)std::max<uint64_t>(1, batch_records), est_bases / (double)std::max<uint64_t>(1, batch_bases)));
    if (need >= 2 && need < 8 * nworkers) want = std::max<uint64_t>(want, (need + nworkers - 1) / nworkers * nworkers);
  }
  return want > 1 ? std::max<uint64_t>(1, file_bytes / want + 1) : ~0ull;
}

uint64_t Aligner::row_bases(const std::string& line) {
  uint64_t v[9] = {0};
  size_t pos = 0;
  for (int col = 0; col < 9; ++col) {
    while (pos < line.size() && std::isspace((unsigned char)line[pos])) ++pos;
    const size_t st = pos;
    while (pos < line.size() && !std::isspace((unsigned char)line[pos])) ++pos;
    if (st == pos) return 0;
    if (col == 2 || col == 3 || col == 7 || col == 8) v[col] = strtoull(line.c_str() + st, nullptr, 10);
  }
  return (v[3] > v[2] ? v[3] - v[2] : 0) + (v[8] > v[7] ? v[8] - v[7] : 0);
}

// The reference streams records from a reader through a pool of workers to a writer
// (computeAlignments.hpp:318-455).  Here the reader hands out batches of mapping rows, one worker per GPU
// takes the next batch whenever its device is free (the greedy least-loaded assignment of
// scripts/split_approx_mappings_in_chunks.py:19-27,47, taken at run time instead of from predicted weights),
// and finished batches are written in the order they were read: the output does not depend on the number of
// GPUs, and host memory holds the batches in flight, not the run.
Summary Aligner::compute() {
  Summary sum;
  const auto t0 = std::chrono::steady_clock::now();
  std::ifstream in(param.mashmapPafFile);
  if (!in.is_open()) throw std::runtime_error("[wfmash::align] Error! Failed to open input mapping file: " + param.mashmapPafFile);
  std::ofstream outstream(param.pafOutputFile);
  if (!outstream.is_open()) throw std::runtime_error("[wfmash::align] Error! Failed to open output file: " + param.pafOutputFile);
  if (param.sam_format) {  // write_sam_header (computeAlignments.hpp:725-736)
    for (int i = 0; i < ref->nseq(); ++i) outstream << "@SQ\tSN:" << ref->name(i) << "\tLN:" << ref->length(i) << "\n";
    outstream << "@PG\tID:wfmash\tPN:wfmash\tVN:" WFMASH_HIP_VERSION "\tCL:wfmash\n";
  }
  const size_t ngpu = gpus.size();
  // up to three batches per GPU in flight when there are host threads for it: the host stages of a batch (sequence
  // fetches, CIGAR surgery, PAF text) run while the device works on another
  static const size_t workers_env = getenv("WFM_ALIGN_WORKERS") ? (size_t)std::max(1, atoi(getenv("WFM_ALIGN_WORKERS"))) : 0;  // A/B runs
  // (four since round 5 where the host has 16 threads per GPU: with the round's kernels a batch of a pangenome rank is 100 ms of device time in 170 ms
  // of its worker's, and a full-size rank went from 1.37 - 1.42 s on three workers to 1.18 - 1.29 s on four -- gpurun_out/r5i.log; six gain nothing more)
  // (six since round 6 where the host has 48 threads per GPU: on the all-vs-all job at north_star's size -- 187 batches -- the device's busy time went from 7.58 to 7.30 s
  // and the align phase from 8.3 - 8.45 to 7.8 - 8.1 s, eight workers 8.2 - 8.3 s: gpurun_out/r6cb; a rank's 28 batches are the same 1.08 - 1.12 s on four, five or six)
  const size_t per_gpu = workers_env ? workers_env
                                     : ((size_t)param.threads >= 48 * ngpu ? 6 : ((size_t)param.threads >= 16 * ngpu ? 4 : ((size_t)param.threads >= 12 * ngpu ? 3 : ((size_t)param.threads >= 2 * ngpu ? 2 : 1))));
  const size_t nworkers_max = ngpu * per_gpu;
  uint64_t est_batches = ~0ull;  // (from the file's size and its first rows; unknown for a stream)
  // several GPUs: no batch may hold more than an eighth of one GPU's share of the file.  One GPU: a file of one batch
  // stays one batch (WFM_ALIGN_MIN_BATCHES cuts it for A/B runs); a file of a few batches is cut into a multiple of the
  // workers' number, level ones -- four batches on three workers are two rounds of which the second leaves the device to
  // one batch's tails.  (The number of records is estimated from the file's size and its first rows.)
  uint64_t batch_bytes = ~0ull;
  static const uint64_t min_batches = getenv("WFM_ALIGN_MIN_BATCHES") ? (uint64_t)std::max(1, atoi(getenv("WFM_ALIGN_MIN_BATCHES"))) : 1;
  static const bool level_batches = !(getenv("WFM_ALIGN_LEVEL") && atoi(getenv("WFM_ALIGN_LEVEL")) == 0);
  {
    // (a mapping file that cannot be rewound -- a FIFO, /dev/stdin -- is read as it comes: no size, no look ahead)
    in.seekg(0, std::ios::end);
    const std::streamoff end_at = in.fail() ? (std::streamoff)-1 : (std::streamoff)in.tellg();
    in.clear();
    if (end_at >= 0) in.seekg(0, std::ios::beg);
    const bool seekable = end_at >= 0 && !in.fail();
    in.clear();
    const uint64_t file_bytes = seekable ? (uint64_t)end_at : 0;
    uint64_t rows = 0, bytes = 0, bases = 0;
    if (level_batches && nworkers_max > 1 && file_bytes > 0) {
      std::string line;
      while (rows < 256 && std::getline(in, line)) {
        if (line.empty()) continue;
        ++rows; bytes += line.size() + 1; bases += row_bases(line);
      }
      in.clear();
      in.seekg(0, std::ios::beg);
    }
    batch_bytes = plan_batch_bytes(file_bytes, rows, bytes, bases, param.batch_records, param.batch_bases, nworkers_max, ngpu, min_batches, level_batches);
    // (no more workers than the file has batches: a worker beyond the first of a device brings a handle with arenas of its own, and the host threads are shared
    // out over the workers -- a mapping file of one batch keeps them all)
    if (rows > 0 && bytes > 0) {
      const double est_rows = (double)file_bytes / ((double)bytes / (double)rows);
      const double est_bases = est_rows * ((double)bases / (double)rows);
      uint64_t need = (uint64_t)std::ceil(std::max(est_rows / (double)std::max<uint64_t>(1, param.batch_records), est_bases / (double)std::max<uint64_t>(1, param.batch_bases)));
      if (batch_bytes != ~0ull) need = std::max<uint64_t>(need, (file_bytes + batch_bytes - 1) / batch_bytes);
      est_batches = std::max<uint64_t>(1, need);
    }
  }
  const size_t nworkers = (size_t)std::max<uint64_t>(ngpu, std::min<uint64_t>(nworkers_max, est_batches));
  std::mutex read_mu, write_mu;
  uint64_t next_seq = 0, next_write = 0;
  std::map<uint64_t, std::string> pending;
  std::string first_error;
  std::atomic<bool> failed{false};
  bool more_rows = true;  // (under read_mu) rows are left behind the batch handed out last
  std::vector<std::atomic<int>> in_flight(ngpu);  // batches on the device right now, per device
  for (auto& a : in_flight) a.store(0);
  uint64_t rows_read = 0;  // (under read_mu) non-empty rows handed out so far
  auto read_batch = [&](std::vector<std::string>& batch, uint64_t& first_row) -> int64_t {  // the batch's number, or -1 at the end
    std::lock_guard<std::mutex> lk(read_mu);
    batch.clear();
    first_row = rows_read;
    uint64_t bases = 0, bytes = 0;
    std::string line;
    while (batch.size() < param.batch_records && bases < param.batch_bases && bytes < batch_bytes && std::getline(in, line)) {
      if (line.empty()) continue;
      bases += row_bases(line);
      bytes += line.size() + 1;
      batch.push_back(std::move(line));
    }
    rows_read += batch.size();
    more_rows = !batch.empty() && in.peek() != std::char_traits<char>::eof();
    return batch.empty() ? -1 : (int64_t)next_seq++;
  };
  auto write_batch = [&](uint64_t seq, std::string&& text) {
    std::lock_guard<std::mutex> lk(write_mu);
    pending.emplace(seq, std::move(text));
    for (auto it = pending.begin(); it != pending.end() && it->first == next_write; it = pending.erase(it), ++next_write)
      outstream << it->second;
    outstream.flush();
  };
  std::vector<Summary> part(nworkers);
  const int threads_each = std::max(1, param.threads / (int)nworkers);
  // Every worker beyond the first of a device works on a handle of its own (own stream, own arenas): its batch's device
  // calls then really run beside the other workers' -- the few-workgroup tails of one batch (the patches that overflow
  // their score budget, the last leaves) under the wide levels of another -- instead of taking turns on one handle.
  // The extra handles stay with the process (two per device at most) and serve the next run as well: their arenas are
  // what a first use pays for, and a handle given back to the driver costs the next hipMalloc its scrubbing.
  std::vector<wfm_handle_t*> use(nworkers, nullptr);
  std::vector<std::pair<int, int>> borrowed;  // (device, slot) taken from the pool
  static const bool own_handles = !(getenv("WFM_ALIGN_OWN_HANDLES") && atoi(getenv("WFM_ALIGN_OWN_HANDLES")) == 0);
  struct Pool {
    std::mutex mu;
    std::map<int, std::vector<std::pair<wfm_handle_t*, bool>>> by_device;  // handle, in use
  };
  static Pool pool;
  for (size_t wk = 0; wk < nworkers; ++wk) {
    use[wk] = gpus[wk % ngpu];
    if (wk < ngpu || !own_handles) continue;
    const int dev = wfm_device(gpus[wk % ngpu]);
    std::lock_guard<std::mutex> lk(pool.mu);
    auto& v = pool.by_device[dev];
    int slot = -1;
    for (size_t q = 0; q < v.size(); ++q) if (!v[q].second) { slot = (int)q; break; }
    if (slot < 0 && v.size() + 1 < std::max<size_t>(3, per_gpu)) {
      wfm_handle_t* nh = nullptr;
      if (wfm_create(dev, &nh) == WFM_OK) { v.emplace_back(nh, false); slot = (int)v.size() - 1; }
    }
    if (slot >= 0) { v[(size_t)slot].second = true; use[wk] = v[(size_t)slot].first; borrowed.emplace_back(dev, slot); }
  }
  struct Return {
    Pool& p; std::vector<std::pair<int, int>>& b;
    ~Return() { std::lock_guard<std::mutex> lk(p.mu); for (auto& x : b) p.by_device[x.first][(size_t)x.second].second = false; }
  } give_back{pool, borrowed};
  // (the first batch of a device goes to its first worker, the one on the caller's handle: a file of one batch then runs
  // on arenas that are most likely there already)
  std::vector<std::atomic<int>> first_taken(ngpu);
  for (auto& a : first_taken) a.store(0);
  auto worker = [&](size_t wk) {
    try {
      std::vector<std::string> batch;
      if (wk >= ngpu) while (!first_taken[wk % ngpu].load() && !failed.load()) std::this_thread::yield();
      uint64_t first_row = 0;
      for (int64_t seq; !failed.load() && (seq = read_batch(batch, first_row)) >= 0;) {
        first_taken[wk % ngpu].store(1);
        const double at0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        // a batch that has its device to itself -- none beside it, none to come -- is cut into parts by the device layer
        // (wfm_set_concurrent_calls: its levels are chains of short launches, and one chain does not fill a device)
        // (both under the readers' lock: two workers that start together must not each see the device as theirs alone)
        bool more;
        int beside;
        { std::lock_guard<std::mutex> lk(read_mu); more = more_rows; beside = in_flight[wk % ngpu].fetch_add(1); }
        wfm_set_concurrent_calls(use[wk], beside + (more && per_gpu > 1 ? 1 : 0));
        struct Leave { std::atomic<int>& a; ~Leave() { a.fetch_sub(1); } } leave{in_flight[wk % ngpu]};
        std::string text = align_batch(use[wk], batch, threads_each, part[wk], first_row);
        const double at1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        write_batch((uint64_t)seq, std::move(text));
        if (getenv("WFM_DEBUG"))
          fprintf(stderr, "[wfmash::align] worker %zu: batch %lld from +%.0f ms to +%.0f ms, written at +%.0f ms\n", wk, (long long)seq, at0, at1,
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> lk(write_mu);
      if (first_error.empty()) first_error = e.what();
      failed.store(true);
    }
    first_taken[wk % ngpu].store(1);  // (nothing left for this device's first worker either: the others must not wait)
  };
  {
    std::vector<std::thread> pool;
    for (size_t wk = 1; wk < nworkers; ++wk) pool.emplace_back(worker, wk);
    worker(0);
    for (auto& t : pool) t.join();
  }
  if (failed.load()) throw std::runtime_error(first_error);
  {
    // per device: the time during which a kernel of any of its workers' calls was running (their calls overlap)
    std::vector<std::vector<std::pair<double, double>>> iv(ngpu);
    for (size_t wk = 0; wk < nworkers; ++wk) {
      const Summary& p = part[wk];
      sum.records += p.records; sum.aligned_bp += p.aligned_bp; sum.written += p.written; sum.skipped += p.skipped;
      sum.cells += p.cells;
      sum.cells_tile += p.cells_tile; sum.tile_launches += p.tile_launches; sum.ms_tile += p.ms_tile; sum.ms_tags += p.ms_tags;
      sum.ms_rows += p.ms_rows; sum.ms_fetch += p.ms_fetch; sum.ms_wflign += p.ms_wflign; sum.ms_text += p.ms_text; sum.batches += p.batches;
      iv[wk % ngpu].insert(iv[wk % ngpu].end(), p.busy.begin(), p.busy.end());
    }
    for (auto& v : iv) {
      std::sort(v.begin(), v.end());
      double total = 0, lo = 0, hi = -1;
      for (const auto& x : v) {
        if (x.first > hi) { if (hi > lo) total += hi - lo; lo = x.first; hi = x.second; }
        else hi = std::max(hi, x.second);
      }
      if (hi > lo) total += hi - lo;
      sum.ms_gpu = std::max(sum.ms_gpu, total);
      if (getenv("WFM_DEBUG") && !v.empty()) {
        // where the device had no kernel running: its busy share per twentieth of the span from its first kernel to its last (the intervals
        // are on the device's own clock; the run's wall time less that span is the head before the first kernel plus the tail after the last)
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double d_lo = v.front().first, d_hi = d_lo;
        for (const auto& x : v) d_hi = std::max(d_hi, x.second);
        const double span = std::max(d_hi - d_lo, 1e-6);
        std::vector<double> busy20(20, 0.0);
        auto add = [&](double a, double b) {
          for (int q = 0; q < 20; ++q) {
            const double qa = d_lo + span * q / 20, qb = d_lo + span * (q + 1) / 20;
            const double o = std::min(b, qb) - std::max(a, qa);
            if (o > 0) busy20[(size_t)q] += o;
          }
        };
        double mlo = 0, mhi = -1;
        for (const auto& x : v) {
          if (x.first > mhi) { if (mhi > mlo) add(mlo, mhi); mlo = x.first; mhi = x.second; }
          else mhi = std::max(mhi, x.second);
        }
        if (mhi > mlo) add(mlo, mhi);
        fprintf(stderr, "[wfmash::align] device: first kernel to last %.1f ms of the run's %.1f (head + tail %.1f), busy %.1f; busy share per twentieth of that span:", span, wall, wall - span, total);
        for (int q = 0; q < 20; ++q) fprintf(stderr, " %.2f", busy20[(size_t)q] / (span / 20));
        fprintf(stderr, "\n");
      }
    }
    // (a worker's own calls follow one another: the sum of their busy times is a lower bound of its device's)
    for (size_t wk = 0; wk < nworkers; ++wk) sum.ms_gpu = std::max(sum.ms_gpu, part[wk].ms_gpu);
  }
  outstream.close();
  sum.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::cerr << "[wfmash::align] total aligned records = " << sum.records << ", total aligned bp = " << sum.aligned_bp
            << ", completed in " << (uint64_t)(sum.ms_total / 1000.0) << " seconds" << std::endl;
  return sum;
}

}  // namespace align
