// capi_map.cpp -- C entry points of the map phase's host side (include/wfmash_host.h).
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <cstdio>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>

#include "../../include/wfmash_host.h"
#include "../csrc/wfa_handle.h"
#include "ani_estimate.hpp"
#include "capi_map.hpp"
#include "fasta.hpp"
#include "index_file.hpp"
#include "map_filter.hpp"
#include "mapper.hpp"
#include "sequence_ids.hpp"

namespace wfmash_host {

skch::Parameters to_parameters(const wfmh_map_params_t& c) {
  skch::Parameters p;
  p.kmerSize = c.kmer_size;
  p.windowLength = c.window_length;
  p.block_length = c.block_length;
  p.chain_gap = c.chain_gap;
  p.max_mapping_length = c.max_mapping_length;
  p.percentageIdentity = c.percentage_identity;
  p.sketchSize = c.sketch_size;
  p.filterMode = c.filter_mode;
  p.numMappingsForSegment = c.num_mappings_for_segment;
  p.numMappingsForScaffold = c.num_mappings_for_scaffold;
  p.dropRand = c.drop_rand != 0;
  p.split = c.split != 0;
  p.mergeMappings = c.merge_mappings != 0;
  p.skip_self = c.skip_self != 0;
  p.skip_prefix = c.skip_prefix != 0;
  p.lower_triangular = c.lower_triangular != 0;
  p.prefix_delim = c.prefix_delim;
  p.filterLengthMismatches = c.filter_length_mismatches != 0;
  p.sparsity_hash_threshold = c.sparsity_hash_threshold;
  p.overlap_threshold = c.overlap_threshold;
  p.scaffold_overlap_threshold = c.scaffold_overlap_threshold;
  p.scaffold_max_deviation = c.scaffold_max_deviation;
  p.scaffold_gap = c.scaffold_gap;
  p.scaffold_min_length = c.scaffold_min_length;
  p.legacy_output = c.legacy_output != 0;
  p.minimum_hits = c.minimum_hits;
  p.max_kmer_freq = c.max_kmer_freq;
  p.index_by_size = c.index_by_size;
  p.kmerComplexityThreshold = c.kmer_complexity_threshold;
  p.stage1_topANI_filter = c.stage1_topani_filter != 0;
  p.stage2_full_scan = c.stage2_full_scan != 0;
  p.ANIDiff = c.ani_diff;
  p.ANIDiffConf = c.ani_diff_conf;
  p.hgNumerator = c.hg_numerator;
  p.threads = c.threads;
  p.auto_pct_identity = c.auto_pct_identity != 0;
  p.ani_percentile = c.ani_percentile;
  p.ani_adjustment = c.ani_adjustment;
  if (c.target_prefix) p.target_prefix = c.target_prefix;
  if (c.target_list) p.target_list = c.target_list;
  if (c.query_list) p.query_list = c.query_list;
  if (c.index_file) { p.indexFilename = c.index_file; p.create_index_only = c.write_index != 0; }
  if (c.query_prefix) {  // CommonFunc::split(args::get(query_prefix), ',') (parse_args.hpp:204)
    std::stringstream ss(c.query_prefix);
    for (std::string tok; std::getline(ss, tok, ',');) p.query_prefix.push_back(tok);
  }
  return p;
}

}  // namespace wfmash_host

extern "C" {

void wfmh_map_default_params(wfmh_map_params_t* c) {
  if (!c) return;
  const skch::Parameters p;  // the defaults live in one place (map_types.hpp)
  std::memset(c, 0, sizeof(*c));
  c->kmer_size = p.kmerSize;
  c->window_length = p.windowLength;
  c->block_length = p.block_length;
  c->chain_gap = p.chain_gap;
  c->max_mapping_length = p.max_mapping_length;
  c->percentage_identity = p.percentageIdentity;
  c->sketch_size = p.sketchSize;
  c->filter_mode = p.filterMode;
  c->num_mappings_for_segment = p.numMappingsForSegment;
  c->num_mappings_for_scaffold = p.numMappingsForScaffold;
  c->drop_rand = p.dropRand;
  c->split = p.split;
  c->merge_mappings = p.mergeMappings;
  c->skip_self = p.skip_self;
  c->skip_prefix = p.skip_prefix;
  c->lower_triangular = p.lower_triangular;
  c->prefix_delim = p.prefix_delim;
  c->filter_length_mismatches = p.filterLengthMismatches;
  c->sparsity_hash_threshold = p.sparsity_hash_threshold;
  c->overlap_threshold = p.overlap_threshold;
  c->scaffold_overlap_threshold = p.scaffold_overlap_threshold;
  c->scaffold_max_deviation = p.scaffold_max_deviation;
  c->scaffold_gap = p.scaffold_gap;
  c->scaffold_min_length = p.scaffold_min_length;
  c->legacy_output = p.legacy_output;
  c->minimum_hits = p.minimum_hits;
  c->max_kmer_freq = p.max_kmer_freq;
  c->index_by_size = p.index_by_size;
  c->kmer_complexity_threshold = p.kmerComplexityThreshold;
  c->stage1_topani_filter = p.stage1_topANI_filter;
  c->stage2_full_scan = p.stage2_full_scan;
  c->ani_diff = p.ANIDiff;
  c->ani_diff_conf = p.ANIDiffConf;
  c->hg_numerator = p.hgNumerator;
  c->threads = p.threads;
  c->auto_pct_identity = p.auto_pct_identity;
  c->ani_percentile = p.ani_percentile;
  c->ani_adjustment = p.ani_adjustment;
}

int wfmh_map(wfm_handle_t* h, const char* target_fasta, const char* query_fasta, const char* out_paf, const wfmh_map_params_t* params,
             wfmh_map_summary_t* summary) {
  return wfmh_map_multi(&h, 1, target_fasta, query_fasta, out_paf, params, summary);
}

int wfmh_map_multi(wfm_handle_t* const* handles, int n, const char* target_fasta, const char* query_fasta, const char* out_paf,
                   const wfmh_map_params_t* params, wfmh_map_summary_t* summary) {
  if (!handles || n < 1 || !target_fasta || !out_paf) return WFM_E_ARG;
  for (int i = 0; i < n; ++i) if (!handles[i]) return WFM_E_ARG;
  wfm_handle_t* h = handles[0];
  try {
    wfmh_map_params_t def;
    wfmh_map_default_params(&def);
    skch::Parameters p = wfmash_host::to_parameters(params ? *params : def);
    p.refSequences = {std::string(target_fasta)};
    p.querySequences = {std::string(query_fasta ? query_fasta : target_fasta)};
    p.outFileName = out_paf;
    const auto t_open = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    int rc;
    double ms_opened = 0, ms_mapped = 0;
    {
      // the files stay open (and what is loaded of them stays loaded) from the identity estimate through the mapping
      std::vector<std::shared_ptr<wfmash_host::FastaStore>> files;
      for (const auto& f : p.refSequences) files.push_back(wfmash_host::open_shared(f));
      for (const auto& f : p.querySequences) files.push_back(wfmash_host::open_shared(f));
      ms_opened = ms_since(t_open);
      const auto t_call = std::chrono::steady_clock::now();
      double ms_identity = 0;
      // the normalised device copies of the chromosomes stay from the identity estimate's sketches to the index build and the queries' fragments
      // (wfm_map_sequence_cache); the scope ends before the files are let go
      struct SeqCacheScope {
        wfm_handle_t* h;
        explicit SeqCacheScope(wfm_handle_t* hh) : h(hh) { (void)wfm_map_sequence_cache(h, 1); }
        ~SeqCacheScope() { (void)wfm_map_sequence_cache(h, 0); }
      } seq_cache_scope(handles[0]);
      if (p.auto_pct_identity) {
        // main.cpp:72-128: estimate, then derive the sketch size from the estimate unless -s was given
        std::vector<std::string> target_prefix_vec;
        if (!p.target_prefix.empty()) target_prefix_vec.push_back(p.target_prefix);
        const skch::SequenceIdManager ids(p.querySequences, p.refSequences, p.query_prefix, target_prefix_vec,
                                          std::string(1, p.prefix_delim), p.query_list, p.target_list);
        p.percentageIdentity = skch::Stat::estimate_identity_for_groups(p, ids, std::vector<wfm_handle_t*>(handles, handles + n));
        ms_identity = ms_since(t_call);
      }
      skch::Map mapper(p, std::vector<wfm_handle_t*>(handles, handles + n));
      skch::MapSummary s;
      rc = mapper.mapQuery(&s);
      if (summary) {
        summary->targets = s.targets; summary->queries = s.queries; summary->subsets = s.subsets;
        summary->target_bp = s.target_bp; summary->query_bp = s.query_bp; summary->index_windows = s.index_windows;
        summary->fragments = s.fragments; summary->l2_mappings = s.l2_mappings; summary->written = s.written;
        summary->percentage_identity = mapper.parameters().percentageIdentity;
        summary->sketch_size = mapper.parameters().sketchSize;
        summary->ms_index = s.ms_index; summary->ms_map = s.ms_map; summary->ms_filter = s.ms_filter; summary->ms_total = s.ms_total;
        summary->ms_replicate = s.ms_replicate;
        summary->ms_identity = ms_identity;
        summary->ms_wall = ms_since(t_call);
      }
      ms_mapped = ms_since(t_open);
      // the sequences (gigabytes for a pangenome) stay loaded for the call that follows -- the align phase, as a rule, which
      // fetches its windows from the same files -- and what was kept before goes back to the system on a thread of its own
      wfmash_host::keep_until_next(std::move(files));
    }
    if (getenv("WFM_DEBUG"))
      fprintf(stderr, "[wfm] map call: files opened after %.1f ms, mapped after %.1f ms, returning after %.1f ms\n", ms_opened, ms_mapped, ms_since(t_open));
    return rc;
  } catch (const std::exception& e) {
    wfm_set_error(h, e.what());
    return WFM_E_ARG;
  }
}

char* wfmh_test_filter(const char* stage, const wfm_mapping_t* maps, int64_t n, const char* fasta, const char* query_name,
                       const wfmh_map_params_t* prm) {
  if (!stage || !fasta || !query_name || !prm || n < 0 || (n && !maps)) return nullptr;
  std::string text;
  try {
    const skch::Parameters p = wfmash_host::to_parameters(*prm);
    const std::string delim = p.prefix_delim ? std::string(1, p.prefix_delim) : std::string();
    const skch::SequenceIdManager ids({std::string(fasta)}, {std::string(fasta)}, {}, {std::string()}, delim);
    skch::MappingResultsVector_t v((size_t)n);
    if (n) std::memcpy(v.data(), maps, (size_t)n * sizeof(wfm_mapping_t));
    skch::set_filter_threads(getenv("WFM_FILTER_THREADS") ? atoi(getenv("WFM_FILTER_THREADS")) : 1);  // tests: the split passes
    const skch::seqno_t qid = ids.getSequenceId(query_name);
    const skch::offset_t qlen = ids.getSequenceLength(qid);
    std::ostringstream os;
    const std::string st(stage);
    if (st == "subset") {
      skch::MappingOutput::mappingBoundarySanityCheck(qlen, v, ids);
      skch::FilteredMappingsResult r = skch::filterSubsetMappings(v, p, ids, qlen);
      const bool merged = p.mergeMappings && p.split;
      skch::MappingOutput::reportReadMappings(merged ? r.mergedMappings : r.nonMergedMappings, merged ? r.mergedChainInfo : r.nonMergedChainInfo,
                                              query_name, os, ids, p, qlen);
    } else if (st == "onetoone") {
      skch::MappingResultsVector_t kept;
      skch::MappingFilterUtils::filterByGroup(v, kept, p.numMappingsForSegment - 1, true, ids, p);
      skch::MappingOutput::reportReadMappings(kept, query_name, os, ids, p, qlen);
    } else {
      return nullptr;
    }
    text = os.str();
  } catch (const std::exception& e) {
    text = std::string("ERROR: ") + e.what();
  }
  char* out = (char*)malloc(text.size() + 1);
  if (out) std::memcpy(out, text.c_str(), text.size() + 1);
  return out;
}

// Test hook for the on-disk index (host/index_file.cpp; no GPU needed).  op "ids": the id section alone
// (SequenceIdManager::exportIdMapping) of `fasta`'s sequences into out_path.  op "rewrite": every sub-index of
// in_path is read (read_sub_index, ids imported) and written again (write_sub_index) into out_path.
// Returns 0, or -1 with the message on stderr.
int wfmh_test_index_file(const char* op, const char* fasta, char prefix_delim, const char* in_path, const char* out_path) {
  if (!op || !fasta || !out_path) return -1;
  try {
    const std::string delim = prefix_delim ? std::string(1, prefix_delim) : std::string();
    skch::SequenceIdManager ids({std::string(fasta)}, {std::string(fasta)}, {}, {std::string()}, delim);
    std::ofstream out(out_path, std::ios::binary);
    if (!out) throw std::runtime_error("cannot open the output file");
    const std::string o(op);
    if (o == "ids") {
      ids.exportIdMapping(out);
    } else if (o == "rewrite") {
      if (!in_path) return -1;
      std::ifstream in(in_path, std::ios::binary);
      if (!in) throw std::runtime_error("cannot open the input file");
      uint64_t total = 1;
      for (uint64_t b = 0; b < total; ++b) {
        skch::SubIndex sub;
        skch::read_sub_index(in, sub, ids);
        total = sub.total_batches;
        skch::write_sub_index(out, sub, ids);
      }
    } else {
      return -1;
    }
    return out ? 0 : -1;
  } catch (const std::exception& e) {
    fprintf(stderr, "wfmh_test_index_file: %s\n", e.what());
    return -1;
  }
}

}  // extern "C"
