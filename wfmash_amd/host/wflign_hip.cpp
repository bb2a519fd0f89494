// wflign_hip.cpp -- see wflign_hip.hpp.  Host-side CIGAR surgery, swizzle and PAF
// writer of the align path, restated from the reference's behaviour; all wavefront
// arithmetic is done on the GPU through wfm_align_batch (no CPU fallback).
#include "wflign_hip.hpp"
#include "parallel.hpp"
#include "../csrc/wfa_handle.h"

#include <algorithm>
#include <thread>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

namespace wflign {

namespace {
inline bool is_digit(char c) { return std::isdigit(static_cast<unsigned char>(c)) != 0; }
}  // namespace

CigarOps parse_cigar(const std::string& cigar) {
  CigarOps ops;
  ops.reserve(cigar.size() / 2 + 1);
  size_t i = 0;
  while (i < cigar.size()) {
    long long v = 0;
    while (i < cigar.size() && is_digit(cigar[i])) { v = v * 10 + (cigar[i] - '0'); ++i; }
    if (i >= cigar.size()) break;
    ops.emplace_back((int)v, cigar[i++]);
  }
  return ops;
}

std::string cigar_to_string(const CigarOps& ops) {
  std::string s;
  for (const auto& o : ops) { s += std::to_string(o.first); s += o.second; }
  return s;
}

std::string compress_ops(const char* ops, size_t n) {
  std::string out;
  size_t i = 0;
  while (i < n) {
    const char op = ops[i];
    size_t j = i;
    while (j < n && ops[j] == op) ++j;
    out += std::to_string(j - i);
    out += (op == 'M') ? '=' : op;
    i = j;
  }
  return out;
}

// Concatenate two CIGARs, fusing the last op of the first with the first op of
// the second when they are of the same type (wflign.cpp:211-238).
std::string merge_adjacent_ops(const std::string& cigar1, const std::string& cigar2) {
  if (cigar1.empty()) return cigar2;
  if (cigar2.empty()) return cigar1;
  const char op1 = cigar1.back();
  size_t end1 = cigar1.size() - 1;  // index of op char
  size_t start1 = end1;
  while (start1 > 0 && is_digit(cigar1[start1 - 1])) --start1;
  size_t pos2 = 0;
  while (pos2 < cigar2.size() && is_digit(cigar2[pos2])) ++pos2;
  if (pos2 >= cigar2.size()) return cigar1 + cigar2;
  const char op2 = cigar2[pos2];
  if (op1 == op2 && start1 < end1) {
    const long long c1 = std::stoll(cigar1.substr(start1, end1 - start1));
    const long long c2 = std::stoll(cigar2.substr(0, pos2));
    return cigar1.substr(0, start1) + std::to_string(c1 + c2) + op1 + cigar2.substr(pos2 + 1);
  }
  return cigar1 + cigar2;
}

std::string erode_short_matches_in_cigar(const std::string& cigar, int max_match_length, bool is_head_cigar) {
  if (cigar.length() < 6) return cigar;
  CigarOps ops = parse_cigar(cigar);
  if (ops.size() < 3) return cigar;
  size_t first = 1, last = ops.size() - 1;  // candidates are ops[first .. last)
  if (is_head_cigar) last = std::min(last, (size_t)3);
  else first = std::max(first, ops.size() - 3);
  bool modified = false;
  for (size_t i = first; i < last; ++i) {
    const char t = ops[i].second, a = ops[i - 1].second, b = ops[i + 1].second;
    const bool is_match = (t == 'M' || t == '=' || t == 'X');
    const bool opposite_indels = (a == 'I' && b == 'D') || (a == 'D' && b == 'I');
    if (is_match && ops[i].first <= max_match_length && opposite_indels &&
        ops[i - 1].first > ops[i].first && ops[i + 1].first > ops[i].first) {
      ops[i - 1].first += ops[i].first;
      ops[i + 1].first += ops[i].first;
      ops[i].first = 0;
      modified = true;
    }
  }
  if (!modified) return cigar;
  CigarOps merged;
  merged.reserve(ops.size());
  for (const auto& o : ops) {
    if (o.first <= 0) continue;
    if (!merged.empty() && merged.back().second == o.second) merged.back().first += o.first;
    else merged.push_back(o);
  }
  return cigar_to_string(merged);
}

namespace {
constexpr uint64_t MIN_PATCH_LENGTH = 128;        // wflign.cpp:169
constexpr uint64_t MAX_ERODE_LENGTH = 4096;       // wflign.cpp:170
constexpr int MIN_CONSECUTIVE_MATCHES = 11;       // wflign.cpp:171

inline void consume(char op, int count, uint64_t& q, uint64_t& t) {
  if (op == 'M' || op == 'X' || op == '=') { q += (uint64_t)count; t += (uint64_t)count; }
  else if (op == 'I') q += (uint64_t)count;
  else if (op == 'D') t += (uint64_t)count;
}
}  // namespace

Erosion scan_head_erosion(const std::string& main_cigar) {
  Erosion e;
  size_t pos = 0;
  bool found = false;
  while (pos < main_cigar.size()) {
    long long count = 0;
    while (pos < main_cigar.size() && is_digit(main_cigar[pos])) { count = count * 10 + (main_cigar[pos] - '0'); ++pos; }
    const char op = main_cigar[pos++];
    if (op == '=' && count >= MIN_CONSECUTIVE_MATCHES) found = true;
    if (found && e.query_eroded >= MIN_PATCH_LENGTH && e.target_eroded >= MIN_PATCH_LENGTH) break;
    if (e.query_eroded >= MAX_ERODE_LENGTH || e.target_eroded >= MAX_ERODE_LENGTH) break;
    consume(op, (int)count, e.query_eroded, e.target_eroded);
    e.erode_end_pos = pos;
  }
  return e;
}

Erosion scan_tail_erosion(const CigarOps& ops) {
  Erosion e;
  e.erode_start_idx = ops.size();
  bool found = false;
  for (int i = (int)ops.size() - 1; i >= 0; --i) {
    const int count = ops[i].first;
    const char op = ops[i].second;
    if (op == '=' && count >= MIN_CONSECUTIVE_MATCHES) found = true;
    if (found && e.query_eroded >= MIN_PATCH_LENGTH && e.target_eroded >= MIN_PATCH_LENGTH) break;
    if (e.query_eroded >= MAX_ERODE_LENGTH || e.target_eroded >= MAX_ERODE_LENGTH) break;
    consume(op, count, e.query_eroded, e.target_eroded);
    e.erode_start_idx = (size_t)i;
  }
  return e;
}

// ---------------------------------------------------------------------------
// swizzle
// ---------------------------------------------------------------------------
namespace {
std::string merge_cigar_ops(const std::string& cigar) {  // wflign_swizzle.cpp:7-37
  std::string merged;
  long long cur_count = 0;
  char cur_op = '\0';
  for (const auto& o : parse_cigar(cigar)) {
    if (o.second == cur_op) cur_count += o.first;
    else {
      if (cur_op != '\0') { merged += std::to_string(cur_count); merged += cur_op; }
      cur_op = o.second; cur_count = o.first;
    }
  }
  if (cur_op != '\0') { merged += std::to_string(cur_count); merged += cur_op; }
  return merged;
}

bool sequences_match(const std::string& q, const std::string& t, int64_t qs, int64_t ts, int n) {
  if (qs < 0 || ts < 0) return false;
  if (qs + n > (int64_t)q.size() || ts + n > (int64_t)t.size()) return false;
  return std::memcmp(q.data() + qs, t.data() + ts, (size_t)n) == 0;
}

// accepts only CIGARs made of '=' and 'D' (wflign_swizzle.cpp:61-105)
bool verify_eq_del_cigar(const std::string& cigar, const std::string& q, const std::string& t, int64_t qs, int64_t ts) {
  int64_t qp = qs, tp = ts;
  for (const auto& o : parse_cigar(cigar)) {
    const int v = o.first;
    if (o.second == '=') {
      if (qp < 0 || tp < 0 || qp + v > (int64_t)q.size() || tp + v > (int64_t)t.size()) return false;
      if (std::memcmp(q.data() + qp, t.data() + tp, (size_t)v) != 0) return false;
      qp += v; tp += v;
    } else if (o.second == 'D') {
      if (tp + v > (int64_t)t.size()) return false;
      tp += v;
    } else {
      return false;
    }
  }
  return true;
}
}  // namespace

std::string try_swap_start_pattern(const std::string& cigar, const std::string& query_seq, const std::string& target_seq,
                                   int64_t query_start, int64_t target_start) {
  // first two ops
  size_t i = 0;
  long long n = 0, dlen = 0;
  while (i < cigar.size() && is_digit(cigar[i])) { n = n * 10 + (cigar[i] - '0'); ++i; }
  if (i >= cigar.size()) return cigar;
  const char op1 = cigar[i++];
  while (i < cigar.size() && is_digit(cigar[i])) { dlen = dlen * 10 + (cigar[i] - '0'); ++i; }
  if (i >= cigar.size()) return cigar;
  const char op2 = cigar[i++];
  if (op1 == '=' && op2 == 'D' && sequences_match(query_seq, target_seq, query_start, target_start + dlen, (int)n)) {
    return merge_cigar_ops(std::to_string(dlen) + "D" + std::to_string(n) + "=" + cigar.substr(i));
  }
  return cigar;
}

std::string try_swap_end_pattern(const std::string& cigar, const std::string& query_seq, const std::string& target_seq,
                                 int64_t query_start, int64_t target_start) {
  const CigarOps ops = parse_cigar(cigar);
  if (ops.size() < 2) return cigar;
  // the reference's backwards parser also requires both ops to carry digits
  const auto& last = ops[ops.size() - 1];
  const auto& prev = ops[ops.size() - 2];
  if (!(prev.second == 'D' && last.second == '=')) return cigar;
  const int n = last.first, dlen = prev.first;
  // alignment_end_coords counts only '=' and 'D' (wflign_swizzle.cpp:192-215)
  int64_t end_q = query_start, end_t = target_start;
  for (const auto& o : ops) {
    if (o.second == '=') { end_q += o.first; end_t += o.first; }
    else if (o.second == 'D') end_t += o.first;
  }
  if (!sequences_match(query_seq, target_seq, end_q - n, end_t - n - dlen, n)) return cigar;
  // byte position where the second-to-last op starts
  size_t p = cigar.size();
  for (int k = 0; k < 2; ++k) {
    --p;  // op char
    while (p > 0 && is_digit(cigar[p - 1])) --p;
  }
  std::string swapped = merge_cigar_ops(cigar.substr(0, p) + std::to_string(n) + "=" + std::to_string(dlen) + "D");
  if (!verify_eq_del_cigar(swapped, query_seq, target_seq, query_start, target_start)) return cigar;
  return swapped;
}

// ---------------------------------------------------------------------------
// PAF writer
// ---------------------------------------------------------------------------
double float2phred(double prob) {
  if (prob == 1) return 255;
  const double p = -10 * std::log10(prob);
  if (p < 0 || p > 255) return 255;
  return p;
}

namespace {
struct CigarStats {
  uint64_t matches = 0, mismatches = 0, insertions = 0, inserted_bp = 0, deletions = 0, deleted_bp = 0,
           ref_len = 0, q_len = 0;
};
CigarStats cigar_stats(const CigarOps& ops, size_t b, size_t e) {  // process_compressed_cigar, wflign_patch.cpp:226-283
  CigarStats s;
  for (size_t i = b; i < e; ++i) {
    const uint64_t len = (uint64_t)ops[i].first;
    switch (ops[i].second) {
      case 'M': case '=': s.matches += len; s.ref_len += len; s.q_len += len; break;
      case 'X': s.mismatches += len; s.ref_len += len; s.q_len += len; break;
      case 'I': s.insertions++; s.inserted_bp += len; s.q_len += len; break;
      case 'D': s.deletions++; s.deleted_bp += len; s.ref_len += len; break;
      default: break;
    }
  }
  return s;
}
}  // namespace

bool write_alignment_paf(std::string& out, const std::string& cigar_str, const std::string& query_name,
                         uint64_t query_total_length, uint64_t query_offset, uint64_t query_length, bool query_is_rev,
                         const std::string& target_name, uint64_t target_total_length, uint64_t target_offset,
                         const PafParams& pp, float mashmap_estimated_identity, int32_t chain_id, int32_t chain_length,
                         int32_t chain_pos) {
  if (cigar_str.empty()) return false;
  const CigarOps ops = parse_cigar(cigar_str);
  // trim_indels (wflign_patch.cpp:139-223): strip leading / trailing I and D runs, shifting the coordinates
  size_t b = 0, e = ops.size();
  uint64_t new_ref_start = target_offset, new_query_start = query_offset;
  while (b < e && (ops[b].second == 'I' || ops[b].second == 'D')) {
    if (ops[b].second == 'I') new_query_start += (uint64_t)ops[b].first; else new_ref_start += (uint64_t)ops[b].first;
    ++b;
  }
  if (b < e) while (e > b && (ops[e - 1].second == 'I' || ops[e - 1].second == 'D')) --e;
  const CigarStats s = cigar_stats(ops, b, e);
  if (b >= e) return false;
  const double gap_compressed_identity = (double)s.matches / (double)(s.matches + s.mismatches + s.insertions + s.deletions);
  const double block_identity = (double)s.matches / (double)(s.matches + s.mismatches + s.inserted_bp + s.deleted_bp);
  if (!(gap_compressed_identity >= pp.min_identity && s.q_len >= pp.min_alignment_length && block_identity >= pp.min_block_identity))
    return false;
  uint64_t q_start, q_end;
  if (query_is_rev) {
    q_start = query_offset + (query_length - (new_query_start - query_offset) - s.q_len);
    q_end = query_offset + (query_length - (new_query_start - query_offset));
  } else {
    q_start = new_query_start;
    q_end = new_query_start + s.q_len;
  }
  const uint64_t aln_ref_pos = new_ref_start - target_offset;
  std::ostringstream os;  // default iostream formatting = the reference's (6 significant digits)
  os << query_name << "\t" << query_total_length << "\t" << q_start << "\t" << q_end << "\t"
     << (query_is_rev ? "-" : "+") << "\t" << target_name << "\t" << target_total_length << "\t"
     << target_offset + aln_ref_pos << "\t" << target_offset + aln_ref_pos + s.ref_len << "\t"
     << s.matches << "\t" << std::max(s.ref_len, s.q_len) << "\t" << std::round(float2phred(1.0 - block_identity)) << "\t"
     << "gi:f:" << gap_compressed_identity << "\t" << "bi:f:" << block_identity << "\t"
     << "md:f:" << mashmap_estimated_identity << "\t";
  if (chain_length > 0) os << "ch:Z:" << chain_id << "." << chain_length << "." << chain_pos << "\t";  // id.LENGTH.pos, wflign_patch.cpp:2708
  os << "cg:Z:";
  for (size_t i = b; i < e; ++i) os << ops[i].first << ops[i].second;
  os << "\t";
  out += os.str();
  return true;
}

// MD:Z string (write_tag_and_md_string, wflign_patch.cpp:2397-2478): every op but the last is
// handled by the "previous op" branch, the last one by the closing branch.
std::string md_string(const std::string& cigar, int target_start, const char* target) {
  std::ostringstream os;
  os << "MD:Z:";
  CigarOps raw = parse_cigar(cigar), ops;
  for (const auto& o : raw) {  // consecutive equal ops are summed by the reference's scanner
    if (!ops.empty() && ops.back().second == o.second) ops.back().first += o.first; else ops.push_back(o);
  }
  int t_off = target_start, l_md = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    const int len = ops[i].first;
    const char op = ops[i].second;
    const bool last = (i + 1 == ops.size());
    if (!last) {
      if (op == '=' || op == 'M') { l_md += len; t_off += len; }
      else if (op == 'X') { for (int j = 0; j < len; ++j) { os << l_md << target[t_off + j]; l_md = 0; } t_off += len; }
      else if (op == 'D') { os << l_md << "^"; for (int j = 0; j < len; ++j) os << target[t_off + j]; l_md = 0; t_off += len; }
    } else if (len) {
      if (op == '=' || op == 'M') os << len + l_md;
      else if (op == 'X') { for (int j = 0; j < len; ++j) { os << l_md << target[t_off + j]; l_md = 0; } os << "0"; }
      else if (op == 'I') os << l_md;
      else if (op == 'D') { os << l_md << "^"; for (int j = 0; j < len; ++j) os << target[t_off + j]; os << "0"; }
    }
  }
  return os.str();
}

bool write_alignment_sam(std::string& out, const std::string& cigar_str, const std::string& query_name,
                         uint64_t query_offset, bool query_is_rev, const std::string& target_name,
                         uint64_t target_offset, const PafParams& pp, float mashmap_estimated_identity,
                         bool no_seq_in_sam, bool emit_md_tag, const char* query, const char* target,
                         int32_t chain_id, int32_t chain_length, int32_t chain_pos) {
  if (cigar_str.empty()) return false;
  const CigarOps ops = parse_cigar(cigar_str);
  size_t b = 0, e = ops.size();
  uint64_t new_ref_start = target_offset, new_query_start = query_offset;
  while (b < e && (ops[b].second == 'I' || ops[b].second == 'D')) {
    if (ops[b].second == 'I') new_query_start += (uint64_t)ops[b].first; else new_ref_start += (uint64_t)ops[b].first;
    ++b;
  }
  if (b < e) while (e > b && (ops[e - 1].second == 'I' || ops[e - 1].second == 'D')) --e;
  if (b >= e) return false;
  const CigarStats s = cigar_stats(ops, b, e);
  const double gi = (double)s.matches / (double)(s.matches + s.mismatches + s.insertions + s.deletions);
  const double bi = (double)s.matches / (double)(s.matches + s.mismatches + s.inserted_bp + s.deleted_bp);
  if (!(gi >= pp.min_identity && s.q_len >= pp.min_alignment_length && bi >= pp.min_block_identity)) return false;
  std::string trimmed;
  for (size_t i = b; i < e; ++i) { trimmed += std::to_string(ops[i].first); trimmed += ops[i].second; }
  std::ostringstream os;
  os << query_name << "\t" << (query_is_rev ? "16" : "0") << "\t" << target_name << "\t" << new_ref_start + 1 << "\t"
     << std::round(float2phred(1.0 - bi)) << "\t" << trimmed << "\t" << "*\t0\t0\t";
  if (no_seq_in_sam) os << "*";
  else os.write(query + (new_query_start - query_offset), (std::streamsize)s.q_len);
  os << "\t*\t" << "NM:i:" << (s.mismatches + s.inserted_bp + s.deleted_bp) << "\t" << "gi:f:" << gi << "\t"
     << "bi:f:" << bi << "\t" << "md:f:" << mashmap_estimated_identity;
  if (chain_length > 0) {
    os << "\tci:i:" << chain_id;
    os << "\tch:Z:" << chain_id << "." << chain_length << "." << chain_pos;
  }
  if (emit_md_tag) os << "\t" << md_string(trimmed, 0, target);  // target offset 0 + aln.i (= 0), wflign_patch.cpp:2602-2604
  os << "\n";
  out += os.str();
  return true;
}

// ---------------------------------------------------------------------------
// the same helpers on runs (count, op) -- what the batch pipeline works on.  The reference compresses the aligner's op
// string at once (compress_cigar, wflign.cpp:183-208) and every later step reads runs out of the text again; here the
// device hands over runs (wfm_align_batch_rle) and the text is written once, at the end.  Each function is the
// string form above with "position in the text" read as "index of the run"; tests/test_host_logic_cpu.py holds the two
// forms against each other, tests/test_ref_wflign_gpu.py the whole pipeline against the reference's own wflign.cpp.
// ---------------------------------------------------------------------------
void ops_from_runs(const uint32_t* runs, size_t n, CigarOps& out) {
  static const char opc[4] = {'=', 'X', 'I', 'D'};  // M is written '=' (compress_cigar, wflign.cpp:201)
  out.clear();
  out.reserve(n + 8);
  for (size_t i = 0; i < n; ++i) out.emplace_back((int)WFM_RUN_LEN(runs[i]), opc[WFM_RUN_OP(runs[i])]);
}

// erode_short_matches_in_cigar (wflign.cpp:19-106); a CIGAR of three runs has at least six characters, so the text
// form's length test is implied by its run-count test
bool erode_short_matches_ops(CigarOps& ops, int max_match_length, bool is_head_cigar) {
  if (ops.size() < 3) return false;
  size_t first = 1, last = ops.size() - 1;
  if (is_head_cigar) last = std::min(last, (size_t)3);
  else first = std::max(first, ops.size() - 3);
  bool modified = false;
  for (size_t i = first; i < last; ++i) {
    const char t = ops[i].second, a = ops[i - 1].second, b = ops[i + 1].second;
    const bool is_match = (t == 'M' || t == '=' || t == 'X');
    const bool opposite_indels = (a == 'I' && b == 'D') || (a == 'D' && b == 'I');
    if (is_match && ops[i].first <= max_match_length && opposite_indels &&
        ops[i - 1].first > ops[i].first && ops[i + 1].first > ops[i].first) {
      ops[i - 1].first += ops[i].first;
      ops[i + 1].first += ops[i].first;
      ops[i].first = 0;
      modified = true;
    }
  }
  if (!modified) return false;
  size_t o = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    if (ops[i].first <= 0) continue;
    if (o > 0 && ops[o - 1].second == ops[i].second) ops[o - 1].first += ops[i].first;
    else ops[o++] = ops[i];
  }
  ops.resize(o);
  return true;
}

// merge_adjacent_ops (wflign.cpp:211-238): src[from, to) behind dst, only the two runs that meet are fused
void append_merged(CigarOps& dst, const CigarOps& src, size_t from, size_t to) {
  if (from >= to) return;
  if (!dst.empty() && dst.back().second == src[from].second) { dst.back().first += src[from].first; ++from; }
  dst.insert(dst.end(), src.begin() + (long)from, src.begin() + (long)to);
}

// wflign.cpp:241-276; erode_end_pos = number of runs eroded
Erosion scan_head_erosion_ops(const CigarOps& ops) {
  Erosion e;
  bool found = false;
  for (size_t i = 0; i < ops.size(); ++i) {
    const int count = ops[i].first;
    const char op = ops[i].second;
    if (op == '=' && count >= MIN_CONSECUTIVE_MATCHES) found = true;
    if (found && e.query_eroded >= MIN_PATCH_LENGTH && e.target_eroded >= MIN_PATCH_LENGTH) break;
    if (e.query_eroded >= MAX_ERODE_LENGTH || e.target_eroded >= MAX_ERODE_LENGTH) break;
    consume(op, count, e.query_eroded, e.target_eroded);
    e.erode_end_pos = i + 1;
  }
  return e;
}

namespace {
// merge_cigar_ops (wflign_swizzle.cpp:7-37): every pair of neighbours with the same op, over the whole CIGAR
void merge_all_adjacent(CigarOps& ops) {
  size_t o = 0;
  for (size_t i = 0; i < ops.size(); ++i) {
    if (o > 0 && ops[o - 1].second == ops[i].second) ops[o - 1].first += ops[i].first;
    else ops[o++] = ops[i];
  }
  ops.resize(o);
}
inline bool seq_match(const char* q, int64_t qn, const char* t, int64_t tn, int64_t qs, int64_t ts, int64_t n) {  // wflign_swizzle.cpp:39-59
  if (qs < 0 || ts < 0) return false;
  if (qs + n > qn || ts + n > tn) return false;
  return std::memcmp(q + qs, t + ts, (size_t)n) == 0;
}
inline char* put_u64(char* p, uint64_t v) {
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}
}  // namespace

// try_swap_start_pattern (wflign_swizzle.cpp:217-252), query_start = target_start = 0
bool try_swap_start_ops(CigarOps& ops, const char* q, int64_t qn, const char* t, int64_t tn) {
  if (ops.size() < 2) return false;
  if (!(ops[0].second == '=' && ops[1].second == 'D')) return false;
  const int64_t n = ops[0].first, dlen = ops[1].first;
  if (!seq_match(q, qn, t, tn, 0, dlen, n)) return false;
  std::swap(ops[0], ops[1]);
  merge_all_adjacent(ops);
  return true;
}

// try_swap_end_pattern (wflign_swizzle.cpp:254-299): the swapped CIGAR is only kept when it verifies, and the
// verification accepts nothing but '=' and 'D' (:61-105)
bool try_swap_end_ops(CigarOps& ops, const char* q, int64_t qn, const char* t, int64_t tn) {
  if (ops.size() < 2) return false;
  const std::pair<int, char> last = ops[ops.size() - 1], prev = ops[ops.size() - 2];
  if (!(prev.second == 'D' && last.second == '=')) return false;
  const int64_t n = last.first, dlen = prev.first;
  int64_t end_q = 0, end_t = 0;  // alignment_end_coords counts only '=' and 'D' (:192-215)
  bool only_eq_del = true;
  for (const auto& o : ops) {
    if (o.second == '=') { end_q += o.first; end_t += o.first; }
    else if (o.second == 'D') end_t += o.first;
    else only_eq_del = false;
  }
  if (!seq_match(q, qn, t, tn, end_q - n, end_t - n - dlen, n)) return false;
  if (!only_eq_del) return false;  // verify_cigar_alignment would refuse it
  CigarOps sw(ops.begin(), ops.end() - 2);
  sw.emplace_back((int)n, '=');
  sw.emplace_back((int)dlen, 'D');
  merge_all_adjacent(sw);
  int64_t qp = 0, tp = 0;
  for (const auto& o : sw) {
    const int64_t v = o.first;
    if (o.second == '=') {
      if (qp + v > qn || tp + v > tn) return false;
      if (std::memcmp(q + qp, t + tp, (size_t)v) != 0) return false;
      qp += v; tp += v;
    } else {
      if (tp + v > tn) return false;
      tp += v;
    }
  }
  ops.swap(sw);
  return true;
}

// write_alignment_paf (wflign_patch.cpp:2611-2724) from runs, as the line the align driver finally writes: its
// processMappingRecord splits the writer's text at white space and joins the fields with single tabs
// (computeAlignments.hpp:484-525), so the fields go out tab-separated with a closing newline straight away.
// Numbers: the reference streams doubles with the default ostream format = printf's %g.
bool write_alignment_paf_ops(std::string& out, const CigarOps& ops, const std::string& query_name, uint64_t query_total_length,
                             uint64_t query_offset, uint64_t query_length, bool query_is_rev, const std::string& target_name,
                             uint64_t target_total_length, uint64_t target_offset, const PafParams& pp,
                             float mashmap_estimated_identity, int32_t chain_id, int32_t chain_length, int32_t chain_pos) {
  if (ops.empty()) return false;
  size_t b = 0, e = ops.size();
  uint64_t new_ref_start = target_offset, new_query_start = query_offset;
  while (b < e && (ops[b].second == 'I' || ops[b].second == 'D')) {  // trim_indels (wflign_patch.cpp:139-223)
    if (ops[b].second == 'I') new_query_start += (uint64_t)ops[b].first; else new_ref_start += (uint64_t)ops[b].first;
    ++b;
  }
  if (b < e) while (e > b && (ops[e - 1].second == 'I' || ops[e - 1].second == 'D')) --e;
  const CigarStats s = cigar_stats(ops, b, e);
  if (b >= e) return false;
  const double gap_compressed_identity = (double)s.matches / (double)(s.matches + s.mismatches + s.insertions + s.deletions);
  const double block_identity = (double)s.matches / (double)(s.matches + s.mismatches + s.inserted_bp + s.deleted_bp);
  if (!(gap_compressed_identity >= pp.min_identity && s.q_len >= pp.min_alignment_length && block_identity >= pp.min_block_identity))
    return false;
  uint64_t q_start, q_end;
  if (query_is_rev) {
    q_start = query_offset + (query_length - (new_query_start - query_offset) - s.q_len);
    q_end = query_offset + (query_length - (new_query_start - query_offset));
  } else {
    q_start = new_query_start;
    q_end = new_query_start + s.q_len;
  }
  const uint64_t aln_ref_pos = new_ref_start - target_offset;
  char num[512];
  int m = snprintf(num, sizeof num, "\t%llu\t%llu\t%llu\t%c\t", (unsigned long long)query_total_length, (unsigned long long)q_start,
                   (unsigned long long)q_end, query_is_rev ? '-' : '+');
  out.reserve(out.size() + query_name.size() + target_name.size() + 256 + (e - b) * 6);
  out += query_name;
  out.append(num, (size_t)m);
  out += target_name;
  m = snprintf(num, sizeof num, "\t%llu\t%llu\t%llu\t%llu\t%llu\t%g\tgi:f:%g\tbi:f:%g\tmd:f:%g\t", (unsigned long long)target_total_length,
               (unsigned long long)(target_offset + aln_ref_pos), (unsigned long long)(target_offset + aln_ref_pos + s.ref_len),
               (unsigned long long)s.matches, (unsigned long long)std::max(s.ref_len, s.q_len), std::round(float2phred(1.0 - block_identity)),
               gap_compressed_identity, block_identity, (double)mashmap_estimated_identity);
  out.append(num, (size_t)m);
  if (chain_length > 0) {  // id.LENGTH.pos, wflign_patch.cpp:2708
    m = snprintf(num, sizeof num, "ch:Z:%d.%d.%d\t", chain_id, chain_length, chain_pos);
    out.append(num, (size_t)m);
  }
  out += "cg:Z:";
  char buf[4096];
  char* p = buf;
  for (size_t i = b; i < e; ++i) {
    p = put_u64(p, (uint64_t)(unsigned)ops[i].first);
    *p++ = ops[i].second;
    if (p > buf + sizeof buf - 32) { out.append(buf, (size_t)(p - buf)); p = buf; }
  }
  *p++ = '\n';
  out.append(buf, (size_t)(p - buf));
  return true;
}

// ---------------------------------------------------------------------------
// batch pipeline
// ---------------------------------------------------------------------------
namespace {
// fn(i) for i in [0, n) on up to `threads` threads; records are independent of each other (the reference runs one
// Taskflow task per record, computeAlignments.hpp:391-435)
template <typename F>
void for_each_record(size_t n, int threads, F&& fn) {
  const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), (n + 7) / 8);  // no thread for fewer than 8 records
  wfmash_host::parallel_for(n, nt, fn);  // (the process's pool: a pass used to start and join its own threads, 2 - 4 ms each -- parallel.hpp)
}

// A handle runs one batch at a time; several host threads may feed the same GPU (the align driver keeps up to three
// batches per device in flight so that the host stages of one overlap the device stages of another): calls are
// serialised here.
std::mutex& handle_lock(wfm_handle_t* h) {
  static std::mutex reg;
  static std::map<wfm_handle_t*, std::unique_ptr<std::mutex>> locks;
  std::lock_guard<std::mutex> lk(reg);
  auto& m = locks[h];
  if (!m) m.reset(new std::mutex());
  return *m;
}

struct GpuBatch {
  std::vector<wfm_problem_t> probs;
  std::vector<wfm_result_t> res;
  std::vector<uint32_t> flags;  // WFM_PF_* per problem (wfm_get_problem_flags), fetched while the handle is still ours
  uint32_t* runs = nullptr;
  std::string err;  // the handle's message, copied while the handle is still ours
  ~GpuBatch() { wfm_free_runs(runs); }
  int run(wfm_handle_t* h, const wfm_penalties_t& pen, BiwfaStats* st) {
    res.assign(probs.size(), wfm_result_t{});
    wfm_free_runs(runs);
    runs = nullptr;
    if (probs.empty()) return 0;
    std::lock_guard<std::mutex> lk(handle_lock(h));
    const int rc = wfm_align_batch_rle(h, &pen, probs.data(), probs.size(), res.data(), &runs, nullptr);
    if (rc < 0) err = wfm_last_error(h);
    flags.assign(probs.size(), 0u);
    if (rc >= 0) wfm_get_problem_flags(h, flags.data(), flags.size());
    if (rc >= 0 && st) {
      wfm_stats_t s;
      if (wfm_get_stats(h, &s) == WFM_OK) {
        st->cells += s.cells; st->ms_gpu += s.ms_any_busy;
        st->cells_tile += s.cells_tile_unique; st->tile_launches += s.tile_launches; st->ms_tile += s.ms_tile;
      }
      const size_t ni = wfm_get_busy_intervals(h, nullptr, 0);
      if (ni) {
        std::vector<double> iv(2 * ni);
        wfm_get_busy_intervals(h, iv.data(), ni);
        for (size_t q = 0; q < ni; ++q) st->busy.emplace_back(iv[2 * q], iv[2 * q + 1]);
      }
    }
    return rc;
  }
  void ops(size_t i, CigarOps& out) const { ops_from_runs(runs + res[i].ops_off, res[i].n_runs, out); }
};

// A guess of an upper bound of the score, for the device to cut its wavefronts with (wfm_problem_t::score_hint): the
// target window is the mapped range plus padding, so the alignment opens with and ends in a gap -- two gap openings
// and the length difference -- and in between it pays for the divergence mashmap estimated, at 6 per differing base
// (a mismatch costs 5), with 0.1 % and 1000 on top.  Too small a guess only costs that record a second run.
int32_t score_hint(const BiwfaRecord& r, const wflign_penalties_t& penalties) {
  double id = r.mashmap_estimated_identity > 1.0f ? r.mashmap_estimated_identity / 100.0 : r.mashmap_estimated_identity;
  id = std::min(1.0, std::max(0.5, id));
  const double len = (double)std::min(r.target_length, r.query_length);
  const double dl = std::fabs((double)r.target_length - (double)r.query_length);
  static const double per_base = getenv("WFM_HINT_PER_BASE") ? atof(getenv("WFM_HINT_PER_BASE")) : 6.0;
  static const double id_slack = getenv("WFM_HINT_ID_SLACK") ? atof(getenv("WFM_HINT_ID_SLACK")) : 0.001;
  // (1000 since round 5, 200 before: a root that runs past its guess is run again from scratch, and with the device time of a batch made of chains
  // of launches a second chain costs more than the cells a looser bound adds -- C2 55 -> 53 ms, scaled C4 rank 48.3 -> 46.5, 40 Mbp rank 214 -> 207 ms
  // of device time, gpurun_out: the hint sweep of round 5; 2000 gains nothing more)
  static const double konst = getenv("WFM_HINT_CONST") ? atof(getenv("WFM_HINT_CONST")) : 1000.0;
  const double hint = 2.0 * penalties.gap_opening2 + penalties.gap_extension2 * dl + (1.0 - id + id_slack) * len * per_base + konst;
  return hint < 1e9 ? (int32_t)hint : 0;
}

wfm_problem_t head_problem(const BiwfaRecord& r, const Erosion& e) {  // wflign.cpp:280-305
  wfm_problem_t p{};
  p.pattern = r.target; p.plen = (int32_t)e.target_eroded;
  p.text = r.query; p.tlen = (int32_t)e.query_eroded;
  p.mode = WFM_MODE_ENDSFREE;
  p.pattern_begin_free = (int32_t)e.target_eroded; p.pattern_end_free = 0;
  p.text_begin_free = (int32_t)e.query_eroded; p.text_end_free = 0;
  return p;
}
wfm_problem_t tail_problem(const BiwfaRecord& r, const Erosion& e) {  // wflign.cpp:368-397
  wfm_problem_t p{};
  p.pattern = r.target + r.target_length - e.target_eroded; p.plen = (int32_t)e.target_eroded;
  p.text = r.query + r.query_length - e.query_eroded; p.tlen = (int32_t)e.query_eroded;
  p.mode = WFM_MODE_ENDSFREE;
  p.pattern_begin_free = 0; p.pattern_end_free = (int32_t)e.target_eroded;
  p.text_begin_free = 0; p.text_end_free = (int32_t)e.query_eroded;
  return p;
}
// scan_tail_erosion that also says how far down it LOOKED (the run it stopped at is inspected, not eroded)
Erosion scan_tail(const CigarOps& ops, size_t* looked_from) {
  Erosion e;
  e.erode_start_idx = ops.size();
  *looked_from = ops.size();
  bool found = false;
  for (int i = (int)ops.size() - 1; i >= 0; --i) {
    *looked_from = (size_t)i;
    const int count = ops[(size_t)i].first;
    const char op = ops[(size_t)i].second;
    if (op == '=' && count >= MIN_CONSECUTIVE_MATCHES) found = true;
    if (found && e.query_eroded >= MIN_PATCH_LENGTH && e.target_eroded >= MIN_PATCH_LENGTH) break;
    if (e.query_eroded >= MAX_ERODE_LENGTH || e.target_eroded >= MAX_ERODE_LENGTH) break;
    consume(op, count, e.query_eroded, e.target_eroded);
    e.erode_start_idx = (size_t)i;
  }
  return e;
}
}  // namespace

// The reference patches a record's head, then scans the PATCHED CIGAR for its tail (wflign.cpp:323-364).  The tail scan
// walks up from the end and stops within 4096 bases, the head patch replaces the runs before erode_end_pos and may
// change the count of the one run it meets: whenever the tail scan of the unpatched CIGAR never looked at a run the head
// patch can touch -- every record longer than a few kilobases -- both scans see what they would have seen in turn, and
// the two patches of all records of the batch go to the device in ONE call.  The remaining records (short ones, where
// the scans overlap) get their tail scanned after their head is in place and share a second, small call.
int do_biwfa_alignment_batch(wfm_handle_t* h, std::vector<BiwfaRecord>& recs, const wflign_penalties_t& penalties,
                             bool disable_chain_patching, const PafParams& pp, BiwfaStats* stats, const OutputFormat& fmt) {
  const wfm_penalties_t pen{penalties.mismatch, penalties.gap_opening1, penalties.gap_extension1,
                            penalties.gap_opening2, penalties.gap_extension2};
  const bool dbg = getenv("WFM_DEBUG") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto ts0 = now();
  const int nt = fmt.threads;
  const size_t n = recs.size();
  // the handle's message was copied while the handle was still this thread's (another batch may be using it by now)
  auto fail = [&](GpuBatch& g, int rc) { if (stats) stats->error = g.err; return rc; };
  // ---- stage 1: main end-to-end BiWFA (wflign.cpp:136-165) ----
  struct Work {
    Erosion he, te;
    size_t tail_looked_from = 0;
    int head_slot = -1, tail_slot = -1;  // problem index in the patch call
    bool tail_later = false;             // the scans overlap: the tail is scanned on the head-patched CIGAR
  };
  std::vector<Work> wk(n);
  {
    GpuBatch g;
    g.probs.reserve(n);
    for (const auto& r : recs) {
      wfm_problem_t p{};
      p.pattern = r.target; p.plen = (int32_t)r.target_length;
      p.text = r.query; p.tlen = (int32_t)r.query_length;
      p.mode = WFM_MODE_END2END_BIWFA;
      p.score_hint = score_hint(r, penalties);
      g.probs.push_back(p);
    }
    const int rc = g.run(h, pen, stats);
    if (rc < 0) return fail(g, rc);
    for_each_record(n, nt, [&](size_t i) {
      BiwfaRecord& r = recs[i];
      r.ok = (g.res[i].status == 0);  // status != 0: the reference drops the record silently (wflign.cpp:150-152)
      r.score = g.res[i].score;
      r.tags = g.flags[i] & 0xffu;
      r.paf.clear();
      r.ops.clear();
      if (!r.ok) return;
      g.ops(i, r.ops);
      if (disable_chain_patching) return;
      Work& w = wk[i];
      w.he = scan_head_erosion_ops(r.ops);
      w.te = scan_tail(r.ops, &w.tail_looked_from);
    });
  }
  if (stats)
    for (const auto& r : recs) stats->main_failed += !r.ok;
  const auto ts1 = now();
  size_t later = 0;
  if (!disable_chain_patching) {
    // ---- stages 2 + 3: head patches (wflign.cpp:241-320) and tail patches (wflign.cpp:323-418) ----
    GpuBatch g;
    for (size_t i = 0; i < n; ++i) {
      if (!recs[i].ok) continue;
      Work& w = wk[i];
      const bool head = w.he.query_eroded > 3 || w.he.target_eroded > 3;
      if (head) { w.head_slot = (int)g.probs.size(); g.probs.push_back(head_problem(recs[i], w.he)); }
      // independent of the head patch: without one, or when the tail scan stopped above every run the patch can touch
      const bool apart = !head || w.tail_looked_from > w.he.erode_end_pos;
      if (!apart) { w.tail_later = true; ++later; continue; }
      if (w.te.query_eroded > 3 || w.te.target_eroded > 3) { w.tail_slot = (int)g.probs.size(); g.probs.push_back(tail_problem(recs[i], w.te)); }
    }
    int rc = g.run(h, pen, stats);
    if (rc < 0) return fail(g, rc);
    std::atomic<uint64_t> n_head{0}, n_tail{0};
    for_each_record(n, nt, [&](size_t i) {
      BiwfaRecord& r = recs[i];
      Work& w = wk[i];
      if (!r.ok || (w.head_slot < 0 && w.tail_slot < 0)) return;
      const bool head_ok = w.head_slot >= 0 && g.res[(size_t)w.head_slot].status == 0;
      const bool tail_ok = w.tail_slot >= 0 && g.res[(size_t)w.tail_slot].status == 0;
      if (w.head_slot >= 0) r.tags |= (g.flags[(size_t)w.head_slot] & 0xffu) << 8;
      if (w.tail_slot >= 0) r.tags |= (g.flags[(size_t)w.tail_slot] & 0xffu) << 16;
      if (!head_ok && !tail_ok) return;
      CigarOps out, patch;
      size_t from = 0, to = r.ops.size();
      if (head_ok) {
        g.ops((size_t)w.head_slot, out);
        erode_short_matches_ops(out, 3, true);
        from = w.he.erode_end_pos;
        ++n_head;
      }
      if (tail_ok) to = w.te.erode_start_idx;
      out.reserve(out.size() + (to - from) + 64);
      append_merged(out, r.ops, from, to);
      if (tail_ok) {
        g.ops((size_t)w.tail_slot, patch);
        erode_short_matches_ops(patch, 3, false);
        append_merged(out, patch, 0, patch.size());
        ++n_tail;
      }
      r.ops.swap(out);
    });
    if (later) {  // short records: the tail scan on the head-patched CIGAR, as the reference runs it
      GpuBatch g2;
      std::vector<size_t> owner;
      for (size_t i = 0; i < n; ++i) {
        Work& w = wk[i];
        if (!recs[i].ok || !w.tail_later) continue;
        size_t looked;
        w.te = scan_tail(recs[i].ops, &looked);
        if (w.te.query_eroded > 3 || w.te.target_eroded > 3) { owner.push_back(i); g2.probs.push_back(tail_problem(recs[i], w.te)); }
      }
      rc = g2.run(h, pen, stats);
      if (rc < 0) return fail(g2, rc);
      for_each_record(owner.size(), nt, [&](size_t j) {
        recs[owner[j]].tags |= (g2.flags[j] & 0xffu) << 16;
        if (g2.res[j].status != 0) return;
        BiwfaRecord& r = recs[owner[j]];
        CigarOps patch;
        g2.ops(j, patch);
        erode_short_matches_ops(patch, 3, false);
        r.ops.resize(wk[owner[j]].te.erode_start_idx);
        append_merged(r.ops, patch, 0, patch.size());
        ++n_tail;
      });
    }
    if (stats) { stats->head_patches += n_head.load(); stats->tail_patches += n_tail.load(); }
  }
  const auto ts2 = now();
  // ---- stage 4: swizzle + record (wflign.cpp:423-454) ----
  for_each_record(n, nt, [&](size_t ri) {
    BiwfaRecord& r = recs[ri];
    if (!r.ok) return;
    const int64_t qn = (int64_t)r.query_length, tn = (int64_t)(r.target_avail ? r.target_avail : r.target_length);
    try_swap_start_ops(r.ops, r.query, qn, r.target, tn);
    try_swap_end_ops(r.ops, r.query, qn, r.target, tn);
    if (fmt.paf_format_else_sam)
      write_alignment_paf_ops(r.paf, r.ops, r.query_name, r.query_total_length, r.query_offset, r.query_length,
                              r.query_is_rev, r.target_name, r.target_total_length, r.target_offset, pp,
                              r.mashmap_estimated_identity, r.chain_id, r.chain_length, r.chain_pos);
    else
      write_alignment_sam(r.paf, cigar_to_string(r.ops), r.query_name, r.query_offset, r.query_is_rev, r.target_name, r.target_offset, pp,
                          r.mashmap_estimated_identity, fmt.no_seq_in_sam, fmt.emit_md_tag, r.query, r.target,
                          r.chain_id, r.chain_length, r.chain_pos);
  });
  if (dbg)
    fprintf(stderr, "[wflign] %zu records: main alignment + runs + scans %.1f ms (incl. waiting for the device), patches %.1f ms (%zu tails scanned after their heads), swizzle + records %.1f ms\n",
            n, ms(ts0, ts1), ms(ts1, ts2), later, ms(ts2, now()));
  return 0;
}

}  // namespace wflign
