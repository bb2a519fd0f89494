// wfmash-hip -- command line front end of the MI355X align phase.
// Keeps the reference's flag names for the options the align path reads
// (src/interface/parse_args.hpp): -i/--align-input, -g/--wfa-params, -E, -U,
// -k (ignored), -o.  Mapping (the `-m` phase) is not part of this binary yet;
// supply approximate mappings with -i as the reference's two-phase restart does
// (parse_args.hpp:800-804).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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

int main(int argc, char** argv) {
  wfmh_align_params_t p;
  wfmh_align_default_params(&p);
  std::string mapping, out = "/dev/stdout", target, query;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&](const char* name) -> std::string { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", name); exit(1); } return argv[++i]; };
    if (a == "-i" || a == "--align-input") mapping = next("-i");
    else if (a == "-o" || a == "--output") out = next("-o");
    else if (a == "-E" || a == "--target-padding") p.target_padding = (uint64_t)handy_parameter(next("-E"));
    else if (a == "-U" || a == "--query-padding") p.query_padding = (uint64_t)handy_parameter(next("-U"));
    else if (a == "-g" || a == "--wfa-params") {
      const std::string v = next("-g");
      if (sscanf(v.c_str(), "%d,%d,%d,%d,%d", &p.mismatch, &p.gap_open1, &p.gap_ext1, &p.gap_open2, &p.gap_ext2) != 5) { fprintf(stderr, "[wfmash] ERROR: --wfa-params expects 5 values\n"); return 1; }
    }
    else if (a == "--device") device = atoi(next("--device").c_str());
    else if (a == "--no-patching") p.disable_chain_patching = 1;
    else if (a == "-a" || a == "--sam") p.sam_format = 1;
    else if (a == "-d" || a == "--md-tag") p.emit_md_tag = 1;
    else if (a == "-h" || a == "--help") {
      fprintf(stderr, "usage: wfmash-hip -i mappings.paf [-o out.paf] [-g x,o1,e1,o2,e2] [-E pad] [-U pad] target.fa [query.fa]\n");
      return 0;
    }
    else if (a[0] != '-') { if (target.empty()) target = a; else query = a; }
    else { fprintf(stderr, "[wfmash] unknown option %s\n", a.c_str()); return 1; }
  }
  if (target.empty() || mapping.empty()) { fprintf(stderr, "[wfmash] ERROR: need a target FASTA and -i mappings.paf\n"); return 1; }
  wfm_handle_t* h = nullptr;
  if (wfm_create(device, &h) != WFM_OK) { fprintf(stderr, "[wfmash] ERROR: no usable MI355X device (there is no CPU fallback)\n"); return 2; }
  wfmh_align_summary_t s;
  const int rc = wfmh_align_paf(h, target.c_str(), query.empty() ? nullptr : query.c_str(), mapping.c_str(), out.c_str(), &p, &s);
  if (rc == WFM_OK)
    fprintf(stderr, "[wfmash::align] %llu records, %llu aligned bp, %.1f ms GPU kernels, %.1f ms total => %.3g aligned bp/s\n",
            (unsigned long long)s.records, (unsigned long long)s.aligned_bp, s.ms_gpu, s.ms_total, s.aligned_bp / (s.ms_total * 1e-3));
  wfm_destroy(h);
  return rc == WFM_OK ? 0 : 3;
}
