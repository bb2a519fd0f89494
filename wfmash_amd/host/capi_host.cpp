// capi_host.cpp -- C entry points of the host-side align driver (include/wfmash_host.h).
#include <algorithm>
#include <exception>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wfmash_host.h"
#include "../csrc/wfa_handle.h"
#include "../csrc/wfa_pack.h"
#include "aligner.hpp"
#include "fasta.hpp"
#include "map_stats.hpp"

extern "C" {

// CPU test hook of the packed extension (csrc/wfa_pack.h): both buffers are packed as wfm_upload_sequences' mirror is, the
// sequences begin at byte start_p / start_t of their buffers (any alignment), and the staged comparison of the tile kernel --
// 16 bases, 64, then 32 at a time, from window origins rounded down to a word -- is run from (v, h).  Returns the run length.
int wfmh_test_packed_lce(const uint8_t* buf_p, int64_t n_p, int64_t start_p, const uint8_t* buf_t, int64_t n_t, int64_t start_t, int v, int h, int maxn) {
  if (!buf_p || !buf_t || n_p < 0 || n_t < 0 || start_p < 0 || start_t < 0 || v < 0 || h < 0) return -1;
  std::vector<uint32_t> wp((size_t)(n_p + 15) / 16 + 16, 0u), wt((size_t)(n_t + 15) / 16 + 16, 0u);
  wfm::pack_words_model(buf_p, n_p, wp.data());
  wfm::pack_words_model(buf_t, n_t, wt.data());
  const int64_t ori_p = start_p & ~(int64_t)15, ori_t = start_t & ~(int64_t)15;
  return wfm::pk_lce_model(wp.data() + (ori_p >> 4), wt.data() + (ori_t >> 4), (uint32_t)(v + (start_p - ori_p)), (uint32_t)(h + (start_t - ori_t)), maxn);
}
// 1 when every byte is one of A C G T (upper case): the problems the packed kernels take
int wfmh_test_is_acgt(const uint8_t* seq, int64_t n) {
  for (int64_t i = 0; i < n; ++i) if (!wfm::pack_is_acgt(seq[i])) return 0;
  return 1;
}

void wfmh_align_default_params(wfmh_align_params_t* p) {
  if (!p) return;
  p->mismatch = 5; p->gap_open1 = 8; p->gap_ext1 = 2; p->gap_open2 = 24; p->gap_ext2 = 1;
  p->min_identity = 0.0f; p->min_alignment_length = 32; p->min_block_identity = 0.1f;
  p->target_padding = 1000; p->query_padding = 1000; p->wflign_max_len_minor = 128000;
  p->disable_chain_patching = 0;
  p->sam_format = 0; p->emit_md_tag = 0; p->no_seq_in_sam = 0;
  p->threads = 0; p->pad_ = 0;
}

int wfmh_align_paf(wfm_handle_t* h, const char* target_fasta, const char* query_fasta, const char* mapping_paf,
                   const char* out_paf, const wfmh_align_params_t* params, wfmh_align_summary_t* summary) {
  return wfmh_align_paf_multi(&h, 1, target_fasta, query_fasta, mapping_paf, out_paf, params, summary);
}

int wfmh_align_paf_multi(wfm_handle_t* const* handles, int n, const char* target_fasta, const char* query_fasta, const char* mapping_paf,
                         const char* out_paf, const wfmh_align_params_t* params, wfmh_align_summary_t* summary) {
  if (!handles || n < 1 || !target_fasta || !mapping_paf || !out_paf) return WFM_E_ARG;
  for (int i = 0; i < n; ++i) if (!handles[i]) return WFM_E_ARG;
  wfm_handle_t* h = handles[0];
  wfmh_align_params_t d;
  wfmh_align_default_params(&d);
  if (params) d = *params;
  try {
    align::Parameters ap;
    ap.refSequences.push_back(target_fasta);
    ap.querySequences.push_back(query_fasta ? query_fasta : target_fasta);
    ap.mashmapPafFile = mapping_paf;
    ap.pafOutputFile = out_paf;
    ap.wfa_patching_mismatch_score = d.mismatch;
    ap.wfa_patching_gap_opening_score1 = d.gap_open1;
    ap.wfa_patching_gap_extension_score1 = d.gap_ext1;
    ap.wfa_patching_gap_opening_score2 = d.gap_open2;
    ap.wfa_patching_gap_extension_score2 = d.gap_ext2;
    ap.min_identity = d.min_identity;
    ap.min_alignment_length = d.min_alignment_length;
    ap.min_block_identity = d.min_block_identity;
    ap.target_padding = d.target_padding;
    ap.query_padding = d.query_padding;
    ap.wflign_max_len_minor = d.wflign_max_len_minor;
    ap.disable_chain_patching = d.disable_chain_patching != 0;
    ap.sam_format = d.sam_format != 0; ap.emit_md_tag = d.emit_md_tag != 0; ap.no_seq_in_sam = d.no_seq_in_sam != 0;
    ap.threads = d.threads > 0 ? d.threads : (int)std::max(1u, std::thread::hardware_concurrency());
    align::Aligner aligner(ap, std::vector<wfm_handle_t*>(handles, handles + n));
    const align::Summary s = aligner.compute();
    if (summary) {
      summary->records = s.records; summary->aligned_bp = s.aligned_bp; summary->written = s.written;
      summary->skipped = s.skipped; summary->cells = s.cells; summary->ms_gpu = s.ms_gpu; summary->ms_total = s.ms_total;
      summary->ms_rows = s.ms_rows; summary->ms_fetch = s.ms_fetch; summary->ms_wflign = s.ms_wflign; summary->ms_text = s.ms_text;
      summary->batches = s.batches;
      summary->cells_tile = s.cells_tile; summary->tile_launches = s.tile_launches; summary->ms_tile = s.ms_tile;
      summary->ms_tags = s.ms_tags;
    }
    return WFM_OK;
  } catch (const std::exception& e) {
    std::cerr << e.what() << std::endl;
    wfm_set_error(h, e.what());
    return WFM_E_ARG;
  }
}

}  // extern "C"

// ---- test hooks: expose the pure host-side CIGAR functions to the CPU test-suite ----
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <vector>

extern "C" {

void wfmh_free(char* p) { free(p); }

// test hook: the align driver's batch plan (Aligner::plan_batch_bytes), pure arithmetic
unsigned long long wfmh_test_plan_batch_bytes(unsigned long long file_bytes, unsigned long long rows, unsigned long long row_bytes,
                                              unsigned long long row_bases_sum, unsigned long long batch_records, unsigned long long batch_bases,
                                              unsigned long long nworkers, unsigned long long ngpu, unsigned long long min_batches, int level) {
  return align::Aligner::plan_batch_bytes(file_bytes, rows, row_bytes, row_bases_sum, batch_records, batch_bases, nworkers, ngpu, min_batches, level != 0);
}

char* wfmh_test_cigar(const char* fn, const char* a, const char* b, const char* query, const char* target,
                      long long i0, long long i1) {
  std::string f = fn ? fn : "", sa = a ? a : "", sb = b ? b : "", q = query ? query : "", t = target ? target : "";
  std::string r;
  if (f == "erode") r = wflign::erode_short_matches_in_cigar(sa, (int)i0, i1 != 0);
  else if (f == "merge") r = wflign::merge_adjacent_ops(sa, sb);
  else if (f == "compress") r = wflign::compress_ops(sa.data(), sa.size());
  else if (f == "swap_start") r = wflign::try_swap_start_pattern(sa, q, t, 0, 0);
  else if (f == "swap_end") r = wflign::try_swap_end_pattern(sa, q, t, 0, 0);
  // ---- the batch pipeline's forms on runs (count, op): held against the text forms above by the CPU suite ----
  else if (f == "erode_ops") {
    wflign::CigarOps o = wflign::parse_cigar(sa);
    wflign::erode_short_matches_ops(o, (int)i0, i1 != 0);
    r = wflign::cigar_to_string(o);
  } else if (f == "merge_ops") {
    wflign::CigarOps o = wflign::parse_cigar(sa);
    const wflign::CigarOps o2 = wflign::parse_cigar(sb);
    wflign::append_merged(o, o2, 0, o2.size());
    r = wflign::cigar_to_string(o);
  } else if (f == "swap_start_ops") {
    wflign::CigarOps o = wflign::parse_cigar(sa);
    wflign::try_swap_start_ops(o, q.data(), (int64_t)q.size(), t.data(), (int64_t)t.size());
    r = wflign::cigar_to_string(o);
  } else if (f == "swap_end_ops") {
    wflign::CigarOps o = wflign::parse_cigar(sa);
    wflign::try_swap_end_ops(o, q.data(), (int64_t)q.size(), t.data(), (int64_t)t.size());
    r = wflign::cigar_to_string(o);
  } else if (f == "head_erosion_ops") {  // third value: the text position behind the eroded runs, for comparison
    const wflign::CigarOps o = wflign::parse_cigar(sa);
    const wflign::Erosion e = wflign::scan_head_erosion_ops(o);
    const size_t pos = wflign::cigar_to_string(wflign::CigarOps(o.begin(), o.begin() + (long)e.erode_end_pos)).size();
    r = std::to_string(e.query_eroded) + "," + std::to_string(e.target_eroded) + "," + std::to_string(pos);
  } else if (f == "runs") {  // a = runs as decimal numbers separated by commas -> CIGAR text
    std::vector<uint32_t> runs;
    std::stringstream ss(sa);
    std::string item;
    while (std::getline(ss, item, ',')) if (!item.empty()) runs.push_back((uint32_t)std::stoul(item));
    wflign::CigarOps o;
    wflign::ops_from_runs(runs.data(), runs.size(), o);
    r = wflign::cigar_to_string(o);
  } else if (f == "paf_ops") {
    std::vector<std::string> p;
    std::stringstream ss(sb);
    std::string item;
    while (std::getline(ss, item, '|')) p.push_back(item);
    if (p.size() == 12) {
      wflign::PafParams pp;
      wflign::write_alignment_paf_ops(r, wflign::parse_cigar(sa), p[0], std::stoull(p[1]), std::stoull(p[2]), std::stoull(p[3]), p[4] == "1", p[5],
                                      std::stoull(p[6]), std::stoull(p[7]), pp, std::stof(p[8]), std::stoi(p[9]), std::stoi(p[10]),
                                      std::stoi(p[11]));
    }
  }
  else if (f == "head_erosion") {
    const wflign::Erosion e = wflign::scan_head_erosion(sa);
    r = std::to_string(e.query_eroded) + "," + std::to_string(e.target_eroded) + "," + std::to_string(e.erode_end_pos);
  } else if (f == "tail_erosion") {
    const wflign::Erosion e = wflign::scan_tail_erosion(wflign::parse_cigar(sa));
    r = std::to_string(e.query_eroded) + "," + std::to_string(e.target_eroded) + "," + std::to_string(e.erode_start_idx);
  } else if (f == "paf") {
    // b = qname|qtotal|qoff|qlen|qrev|tname|ttotal|toff|mmid|chain_id|chain_len|chain_pos
    std::vector<std::string> p;
    std::stringstream ss(sb);
    std::string item;
    while (std::getline(ss, item, '|')) p.push_back(item);
    if (p.size() == 12) {
      wflign::PafParams pp;
      wflign::write_alignment_paf(r, sa, p[0], std::stoull(p[1]), std::stoull(p[2]), std::stoull(p[3]), p[4] == "1", p[5],
                                  std::stoull(p[6]), std::stoull(p[7]), pp, std::stof(p[8]), std::stoi(p[9]), std::stoi(p[10]),
                                  std::stoi(p[11]));
    }
  } else if (f == "min_hits") {   // a = "s,k,identity,ci"
    int sk = 0, k = 0; float id = 0, ci = 0;
    if (sscanf(sa.c_str(), "%d,%d,%f,%f", &sk, &k, &id, &ci) == 4)
      r = std::to_string(skch::Stat::estimateMinimumHits(sk, k, id)) + "," + std::to_string(skch::Stat::estimateMinimumHitsRelaxed(sk, k, id, ci));
  } else if (f == "sketch_cutoffs") {   // a = "s,k,ANIDiff,ANIDiffConf"
    int sk = 0, k = 0; float ad = 0, ac = 0;
    if (sscanf(sa.c_str(), "%d,%d,%f,%f", &sk, &k, &ad, &ac) == 4)
      for (int v : skch::Stat::sketch_cutoffs(sk, k, ad, ac)) { if (!r.empty()) r += ","; r += std::to_string(v); }
  } else if (f == "l2_tables") {   // a = "S,k,identity,ci": keep bits and nucIdentity x 1e4 for every (Q.sketchSize, shared)
    int S = 0, k = 0; float id = 0, ci = 0;
    if (sscanf(sa.c_str(), "%d,%d,%f,%f", &S, &k, &id, &ci) == 4) {
      std::vector<uint8_t> keep;
      std::vector<uint16_t> ident;
      skch::Stat::l2_identity_tables(S, k, id, true, ci, keep, ident);
      for (size_t i = 0; i < keep.size(); ++i) { if (i) r += ","; r += std::to_string((int)keep[i]) + ":" + std::to_string((int)ident[i]); }
    }
  } else if (f == "md") {
    r = wflign::md_string(sa, (int)i0, t.c_str());
  } else if (f == "parse_row") {
    try {
      align::MappingBoundaryRow row;
      align::Aligner::parseMashmapRow(sa, row, (uint64_t)i0, (uint64_t)i1);
      std::ostringstream os;
      os << row.qId << "," << row.qStartPos << "," << row.qEndPos << "," << (row.strand == align::FWD ? "+" : "-") << ","
         << row.refId << "," << row.rStartPos << "," << row.rEndPos << "," << row.mashmap_estimated_identity << ","
         << row.chain_id << "," << row.chain_length << "," << row.chain_pos;
      r = os.str();
    } catch (const std::exception& e) { r = std::string("ERROR"); }
  }
  char* out = (char*)malloc(r.size() + 1);
  memcpy(out, r.c_str(), r.size() + 1);
  return out;
}

char* wfmh_test_fasta(const char* path, const char* name, int64_t start, int64_t end_inclusive, int whole) {
  std::string r;
  try {
    wfmash_host::FastaStore fa(path ? path : "");
    if (!name) {
      std::ostringstream os;
      os << (fa.indexed() ? "indexed" : "in-memory") << "\n";
      for (int i = 0; i < fa.nseq(); ++i) os << fa.name(i) << "\t" << fa.length(i) << "\n";
      r = os.str();
    } else {
      const int i = fa.find(name);
      if (i < 0) r = "ERROR: no such sequence";
      else {
        if (whole) fa.preload({i}, std::max(2, whole));  // whole > 1: that many reader threads
        r = fa.fetch(name, start, end_inclusive);
      }
    }
  } catch (const std::exception& e) {
    r = std::string("ERROR: ") + e.what();
  }
  char* out = (char*)malloc(r.size() + 1);
  if (!out) return nullptr;
  memcpy(out, r.data(), r.size());
  out[r.size()] = 0;
  return out;
}

}  // extern "C"

void wfmh_release_sequences(void) { wfmash_host::release_kept(); }

char* wfmh_test_fasta_shared(const char* path, const char* name) {
  std::string r;
  try {
    std::shared_ptr<wfmash_host::FastaStore> fa = wfmash_host::open_shared(path ? path : "");
    const int i = name ? fa->find(name) : -1;
    if (i < 0) r = "ERROR: no such sequence";
    else { const wfmash_host::SeqView v = fa->sequence(i, 1); r.assign(v.data(), v.size()); }
    wfmash_host::keep_until_next({fa});
  } catch (const std::exception& e) {
    r = std::string("ERROR: ") + e.what();
  }
  char* out = (char*)malloc(r.size() + 1);
  if (!out) return nullptr;
  memcpy(out, r.data(), r.size());
  out[r.size()] = 0;
  return out;
}
