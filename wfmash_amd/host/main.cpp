// wfmash-hip -- command line front end of the MI355X build: map phase, align phase, or both, with
// the reference's flag names (src/interface/parse_args.hpp:61-138) for the options these phases read.
//
//   wfmash-hip target.fa [query.fa]            map, then align the mappings (PAF on stdout)
//   wfmash-hip -m target.fa [query.fa]         approximate mappings only (parse_args.hpp:77)
//   wfmash-hip -i map.paf target.fa [query.fa] align the mappings of a previous -m run (:119, :800-804)
//
//   wfmash-hip -W idx target.fa                build the target index, write it and stop (parse_args.hpp:745-751)
//   wfmash-hip -I idx target.fa [query.fa]     read the index instead of building it (:752-758)
//
// Differences: output goes to stdout or --out FILE; --device picks the GPU, --gpus N spreads the
// queries (map) and the mapping records (align) over N GPUs of the node.  Options of the reference
// that belong to subsystems outside this build (external seeds -K, wavefront plots -G/-u, scaffold
// dump --scaffold-out) are not accepted.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <limits>
#include <regex>
#include <string>
#include <vector>

#include "../../include/wfmash_host.h"

static int64_t handy_parameter(const std::string& v) {  // utils.cpp:13-29 ("50k", "1m", "2g")
  if (v.empty()) return -1;
  double mult = 1;
  std::string t = v;
  const char c = t.back();
  if (c == 'k' || c == 'K') { mult = 1e3; t.pop_back(); }
  else if (c == 'm' || c == 'M') { mult = 1e6; t.pop_back(); }
  else if (c == 'g' || c == 'G') { mult = 1e9; t.pop_back(); }
  if (t.empty() || t.find_first_not_of("0123456789.") != std::string::npos) return -1;
  return (int64_t)(atof(t.c_str()) * mult);
}

static void usage() {
  fprintf(stderr,
          "usage: wfmash-hip [options] target.fa [query.fa]\n"
          "  phases    -m approximate mappings only | -i FILE align mappings from FILE | (neither) map + align\n"
          "  mapping   -p PCT|aniN[+-X] identity [ani50-2]   -k INT k-mer [15]   -w INT window [1k]   -s INT sketch size [auto]\n"
          "            -n INT|inf mappings per segment [inf]   -l INT block length [0]   -c INT chain jump [2k]   -P INT max length [50k]\n"
          "            -N no split   -M no merge   -f no filter   -o one-to-one   -O FLOAT max overlap [0.95]   -x FLOAT sparsify [1.0]\n"
          "            -H INT L1 hits [3]   -F FLOAT high-frequency filter [0.0002]   -b SIZE target batch [all]\n"
          "            -W FILE build the index, write it and stop   -I FILE read the index from FILE\n"
          "            -S INT scaffold mass [10k]   -D INT scaffold dist [100k]   -j INT scaffold jump [100k]   -r INT per scaffold [1]\n"
          "            -Y C group delimiter [#]   -X self maps   -L lower triangular   -t INT threads [1]\n"
          "  alignment -g x,o1,e1,o2,e2 [5,8,2,24,1]   -E INT target padding   -U INT query padding   -a SAM   -d MD tag\n"
          "  other     --out FILE [stdout]   --device INT [0]   --gpus N|all [1] GPUs of this node, starting at --device\n");
}

int main(int argc, char** argv) {
  wfmh_align_params_t ap;
  wfmh_align_default_params(&ap);
  ap.threads = 1;  // -t, default 1 as in the reference (parse_args.hpp: thread_count); the C ABI default (0) means all cores
  wfmh_map_params_t mp;
  wfmh_map_default_params(&mp);
  std::string mapping_in, out = "/dev/stdout", target, query;
  bool approx_only = false, target_padding_given = false, query_padding_given = false;
  int device = 0, gpus = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&](const char* name) -> std::string {
      if (i + 1 >= argc) { fprintf(stderr, "[wfmash] ERROR: missing value for %s\n", name); exit(1); }
      return argv[++i];
    };
    auto size = [&](const char* name) -> int64_t {
      const int64_t v = handy_parameter(next(name));
      if (v < 0) { fprintf(stderr, "[wfmash] ERROR: %s expects a size such as 5000, 50k, 1m\n", name); exit(1); }
      return v;
    };
    if (a == "-m" || a == "--approx-mapping") approx_only = true;
    else if (a == "-i" || a == "--align-paf") mapping_in = next("-i");
    else if (a == "--out") out = next("--out");
    else if (a == "--device") device = atoi(next("--device").c_str());
    else if (a == "--gpus") { const std::string v = next("--gpus"); gpus = v == "all" ? 0 : atoi(v.c_str()); if (gpus < 0) gpus = 1; }
    else if (a == "-t" || a == "--threads") { mp.threads = atoi(next("-t").c_str()); ap.threads = mp.threads; }
    // ---- mapping (parse_args.hpp:71-115)
    else if (a == "-k" || a == "--kmer-size") mp.kmer_size = atoi(next("-k").c_str());
    else if (a == "-s" || a == "--sketch-size") mp.sketch_size = atoi(next("-s").c_str());
    else if (a == "-w" || a == "--window-size") {
      mp.window_length = size("-w");
      if (mp.window_length <= 0) { fprintf(stderr, "[wfmash] ERROR, skch::parseandSave, window size has to be a float value greater than 0.\n"); return 1; }
      if (mp.window_length < 100) {  // parse_args.hpp:324-328
        fprintf(stderr, "[wfmash] ERROR, skch::parseandSave, minimum window size is required to be >= 100 bp.\n");
        return 1;
      }
    }
    else if (a == "-p" || a == "--map-pct-id") {
      const std::string v = next("-p");
      std::smatch m;
      if (std::regex_match(v, m, std::regex("^ani(\\d+)([+-]\\d+)?$"))) {  // parse_args.hpp:345-366
        mp.auto_pct_identity = 1;
        mp.ani_percentile = atoi(m[1].str().c_str());
        if (mp.ani_percentile < 1 || mp.ani_percentile > 99) {  // :352-356
          fprintf(stderr, "[wfmash] ERROR: ANI percentile must be between 1 and 99, got: %d\n", mp.ani_percentile);
          return 1;
        }
        mp.ani_adjustment = m[2].matched ? (float)atof(m[2].str().c_str()) : 0.0f;
      } else if (v == "auto") {
        mp.auto_pct_identity = 1; mp.ani_percentile = 25; mp.ani_adjustment = 0.0f;
      } else {
        char* end = nullptr;
        const float pct = strtof(v.c_str(), &end);
        if (end == v.c_str()) {  // std::stof throws (:383-387)
          fprintf(stderr, "[wfmash] ERROR: Invalid value for -p/--map-pct-id: %s\n", v.c_str());
          return 1;
        }
        if (pct < 50) {  // :377-380
          fprintf(stderr, "[wfmash] ERROR: minimum nucleotide identity requirement should be >= 50%%.\n");
          return 1;
        }
        mp.auto_pct_identity = 0;
        mp.percentage_identity = pct / 100.0f;
      }
    }
    else if (a == "-n" || a == "--mappings") {
      const std::string v = next("-n");
      mp.num_mappings_for_segment = (v == "inf" || v == "-1") ? std::numeric_limits<uint32_t>::max() : (uint32_t)atoll(v.c_str());
    }
    else if (a == "-l" || a == "--block-length") mp.block_length = size("-l");
    else if (a == "-c" || a == "--chain-jump") mp.chain_gap = size("-c");
    else if (a == "-P" || a == "--max-length") mp.max_mapping_length = (uint64_t)size("-P");
    else if (a == "-N" || a == "--no-split") mp.split = 0;
    else if (a == "-M" || a == "--no-merge") mp.merge_mappings = 0;
    else if (a == "-f" || a == "--no-filter") mp.filter_mode = 3;
    else if (a == "-o" || a == "--one-to-one") mp.filter_mode = 2;
    else if (a == "-O" || a == "--overlap") mp.overlap_threshold = atof(next("-O").c_str());
    else if (a == "-x" || a == "--sparsify") {
      const double f = atof(next("-x").c_str());
      mp.sparsity_hash_threshold = f == 1 ? std::numeric_limits<uint64_t>::max() : (uint64_t)(f * (double)std::numeric_limits<uint64_t>::max());
    }
    else if (a == "-H" || a == "--l1-hits") mp.minimum_hits = atoi(next("-H").c_str());
    else if (a == "-F" || a == "--filter-freq") mp.max_kmer_freq = atof(next("-F").c_str());
    else if (a == "-b" || a == "--batch") mp.index_by_size = size("-b");
    else if (a == "-W" || a == "--write-index") { mp.index_file = argv[(next("-W"), i)]; mp.write_index = 1; }
    else if (a == "-I" || a == "--read-index") { mp.index_file = argv[(next("-I"), i)]; mp.write_index = 0; }
    else if (a == "-S" || a == "--scaffold-mass") mp.scaffold_min_length = size("-S");
    else if (a == "-D" || a == "--scaffold-dist") mp.scaffold_max_deviation = size("-D");
    else if (a == "-j" || a == "--scaffold-jump") mp.scaffold_gap = size("-j");
    else if (a == "-r" || a == "--retain-per-scaffold") {
      const std::string v = next("-r");
      mp.num_mappings_for_scaffold = (v == "inf" || v == "-1") ? std::numeric_limits<uint32_t>::max() : (uint32_t)atoll(v.c_str());
    }
    else if (a == "--scaffold-overlap") mp.scaffold_overlap_threshold = atof(next("--scaffold-overlap").c_str());
    else if (a == "-Y" || a == "--group-prefix") { const std::string v = next("-Y"); mp.prefix_delim = v.empty() ? '\0' : v[0]; mp.skip_prefix = mp.prefix_delim != '\0'; }
    else if (a == "-T" || a == "--target-prefix") mp.target_prefix = argv[(next("-T"), i)];
    else if (a == "-R" || a == "--target-list") mp.target_list = argv[(next("-R"), i)];
    else if (a == "-Q" || a == "--query-prefix") mp.query_prefix = argv[(next("-Q"), i)];
    else if (a == "-A" || a == "--query-list") mp.query_list = argv[(next("-A"), i)];
    else if (a == "-X" || a == "--self-maps") mp.skip_self = 0;
    else if (a == "-L" || a == "--lower-triangular") mp.lower_triangular = 1;
    // ---- alignment (parse_args.hpp:119-129)
    else if (a == "-E" || a == "--target-padding") { ap.target_padding = (uint64_t)size("-E"); target_padding_given = true; }
    else if (a == "-U" || a == "--query-padding") { ap.query_padding = (uint64_t)size("-U"); query_padding_given = true; }
    else if (a == "-g" || a == "--wfa-params") {
      const std::string v = next("-g");
      if (sscanf(v.c_str(), "%d,%d,%d,%d,%d", &ap.mismatch, &ap.gap_open1, &ap.gap_ext1, &ap.gap_open2, &ap.gap_ext2) != 5) {
        fprintf(stderr, "[wfmash] ERROR: --wfa-params expects 5 values\n");
        return 1;
      }
    }
    else if (a == "--min-length") ap.min_alignment_length = (uint64_t)atoll(next("--min-length").c_str());
    else if (a == "--min-block-id") ap.min_block_identity = (float)atof(next("--min-block-id").c_str());
    else if (a == "--no-patching") ap.disable_chain_patching = 1;
    else if (a == "-a" || a == "--sam") ap.sam_format = 1;
    else if (a == "-d" || a == "--md-tag") ap.emit_md_tag = 1;
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (a == "-v" || a == "--version") { std::printf("%s\n", WFMASH_HIP_VERSION); return 0; }
    else if (a[0] != '-') { if (target.empty()) target = a; else query = a; }
    else { fprintf(stderr, "[wfmash] unknown option %s\n", a.c_str()); usage(); return 1; }
  }
  if (target.empty()) { fprintf(stderr, "[wfmash] ERROR: need a target FASTA\n"); usage(); return 1; }
  if (approx_only && !mapping_in.empty()) { fprintf(stderr, "[wfmash] ERROR: -m and -i exclude each other\n"); return 1; }
  if (!mapping_in.empty() && mp.index_file) { fprintf(stderr, "[wfmash] ERROR: -i (align only) does not read or write an index (-W / -I)\n"); return 1; }
  if (!approx_only && !(mp.index_file && mp.write_index) && mp.window_length > 10000) {  // parse_args.hpp:330-335
    fprintf(stderr, "[wfmash] ERROR: window size (-w) must be <= 10kb when running alignment.\n"
                    "[wfmash] For larger values, use -m/--approx-mapping to generate mappings,\n"
                    "[wfmash] then align them with: wfmash ... -i mappings.paf\n");
    return 1;
  }
  if ((uint64_t)mp.window_length >= mp.max_mapping_length) {  // parse_args.hpp:485-488
    fprintf(stderr, "[wfmash] ERROR, skch::parseandSave, window size should not be larger than max mapping length.\n");
    return 1;
  }
  if ((uint64_t)mp.block_length >= mp.max_mapping_length) {  // :489-492
    fprintf(stderr, "[wfmash] ERROR, skch::parseandSave, block length should not be larger than max mapping length.\n");
    return 1;
  }
  // what the align phase derives from the segment length (parse_args.hpp:590-591, :600-620): the paddings default to
  // it, capped at 5000, and a mapping may be at most 128 segments long on its shorter axis
  if (!target_padding_given) ap.target_padding = (uint64_t)std::min<int64_t>(mp.window_length, 5000);
  if (!query_padding_given) ap.query_padding = (uint64_t)std::min<int64_t>(mp.window_length, 5000);
  ap.wflign_max_len_minor = (uint64_t)mp.window_length * 128;
  // one handle per GPU; every phase hands its batches to whichever device is free
  if (gpus == 0) { gpus = wfm_device_count() - device; if (gpus < 1) gpus = 1; }
  std::vector<wfm_handle_t*> hs;
  for (int g = 0; g < gpus; ++g) {
    wfm_handle_t* hg = nullptr;
    if (wfm_create(device + g, &hg) != WFM_OK) {
      fprintf(stderr, "[wfmash] ERROR: no usable MI355X device %d (there is no CPU fallback)\n", device + g);
      for (wfm_handle_t* o : hs) wfm_destroy(o);
      return 2;
    }
    hs.push_back(hg);
  }
  wfm_handle_t* h = hs.front();
  auto destroy_all = [&] { for (wfm_handle_t* o : hs) wfm_destroy(o); };
  const char* q = query.empty() ? nullptr : query.c_str();
  int rc = WFM_OK;
  std::string mapping = mapping_in;
  std::string temp;
  if (mapping.empty()) {
    if (approx_only) mapping = out;
    else {  // the hand-off file between the phases (temp_file::create, parse_args.hpp:805-808)
      char tmpl[] = "./wfmash-XXXXXX";
      const int fd = mkstemp(tmpl);
      if (fd < 0) { fprintf(stderr, "[wfmash] ERROR: cannot create a temporary file in the working directory\n"); destroy_all(); return 1; }
      close(fd);
      temp = tmpl;
      mapping = temp;
    }
    wfmh_map_summary_t ms;
    rc = wfmh_map_multi(hs.data(), (int)hs.size(), target.c_str(), q, mapping.c_str(), &mp, &ms);
    if (rc == WFM_OK)
      fprintf(stderr, "[wfmash::map] %llu queries x %llu targets (%llu subsets), identity %.2f%%, sketch %d: %llu fragments, %llu segment mappings, "
                      "%llu records; index %.0f ms (+ %.0f ms to copy it to the other GPUs), mapping %.0f ms, filtering %.0f ms, total %.0f ms\n",
              (unsigned long long)ms.queries, (unsigned long long)ms.targets, (unsigned long long)ms.subsets, ms.percentage_identity * 100.0,
              ms.sketch_size, (unsigned long long)ms.fragments, (unsigned long long)ms.l2_mappings, (unsigned long long)ms.written, ms.ms_index, ms.ms_replicate,
              ms.ms_map, ms.ms_filter, ms.ms_total);
    else fprintf(stderr, "[wfmash::map] ERROR: %s\n", wfm_last_error(h));
  }
  if (mp.index_file && mp.write_index) {  // -W: "index construction completed", nothing else runs (computeMap.hpp:405-415)
    if (!temp.empty()) unlink(temp.c_str());
    destroy_all();
    return rc == WFM_OK ? 0 : 3;
  }
  if (rc == WFM_OK && !approx_only) {
    wfmh_align_summary_t s;
    rc = wfmh_align_paf_multi(hs.data(), (int)hs.size(), target.c_str(), q, mapping.c_str(), out.c_str(), &ap, &s);
    if (rc == WFM_OK)
      fprintf(stderr, "[wfmash::align] %llu records, %llu aligned bp, %.1f ms GPU kernels, %.1f ms total => %.3g aligned bp/s\n",
              (unsigned long long)s.records, (unsigned long long)s.aligned_bp, s.ms_gpu, s.ms_total, s.aligned_bp / (s.ms_total * 1e-3));
    else fprintf(stderr, "[wfmash::align] ERROR: %s\n", wfm_last_error(h));
  }
  if (!temp.empty()) unlink(temp.c_str());
  destroy_all();
  return rc == WFM_OK ? 0 : 3;
}
