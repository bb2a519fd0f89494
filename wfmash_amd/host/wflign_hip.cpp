// wflign_hip.cpp -- see wflign_hip.hpp.  Host-side CIGAR surgery, swizzle and PAF
// writer of the align path, restated from the reference's behaviour; all wavefront
// arithmetic is done on the GPU through wfm_align_batch (no CPU fallback).
#include "wflign_hip.hpp"

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
// batch pipeline
// ---------------------------------------------------------------------------
namespace {
// fn(i) for i in [0, n) on up to `threads` threads; records are independent of each other (the reference runs one
// Taskflow task per record, computeAlignments.hpp:391-435)
template <typename F>
void for_each_record(size_t n, int threads, F&& fn) {
  const int nt = (int)std::min<size_t>((size_t)std::max(1, threads), n);
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<size_t> next{0};
  auto work = [&] { for (size_t i; (i = next.fetch_add(1)) < n;) fn(i); };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

// A handle runs one batch at a time; several host threads may feed the same GPU (the align driver keeps two batches
// per device in flight so that the host stages of one overlap the device stages of the other): calls are serialised here.
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
  std::vector<char> arena;
  int run(wfm_handle_t* h, const wfm_penalties_t& pen, BiwfaStats* st) {
    res.assign(probs.size(), wfm_result_t{});
    if (probs.empty()) return 0;
    arena.resize(wfm_align_arena_bytes(probs.data(), probs.size()) + 8);
    std::lock_guard<std::mutex> lk(handle_lock(h));
    const int rc = wfm_align_batch(h, &pen, probs.data(), probs.size(), res.data(), arena.data(), arena.size());
    if (rc >= 0 && st) {
      wfm_stats_t s;
      if (wfm_get_stats(h, &s) == WFM_OK) { st->cells += s.cells; st->ms_gpu += s.ms_any_busy; }
    }
    return rc;
  }
  std::string cigar(size_t i) const { return compress_ops(arena.data() + res[i].ops_off, res[i].ops_len); }
};
}  // namespace

int do_biwfa_alignment_batch(wfm_handle_t* h, std::vector<BiwfaRecord>& recs, const wflign_penalties_t& penalties,
                             bool disable_chain_patching, const PafParams& pp, BiwfaStats* stats, const OutputFormat& fmt) {
  const wfm_penalties_t pen{penalties.mismatch, penalties.gap_opening1, penalties.gap_extension1,
                            penalties.gap_opening2, penalties.gap_extension2};
  GpuBatch g;
  const bool dbg = getenv("WFM_DEBUG") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto ts0 = now();
  // ---- stage 1: main end-to-end BiWFA (wflign.cpp:136-165) ----
  g.probs.reserve(recs.size());
  for (const auto& r : recs) {
    wfm_problem_t p{};
    p.pattern = r.target; p.plen = (int32_t)r.target_length;
    p.text = r.query; p.tlen = (int32_t)r.query_length;
    p.mode = WFM_MODE_END2END_BIWFA;
    {
      // A guess of an upper bound of the score, for the device to cut its wavefronts with (wfm_problem_t::score_hint): the
      // target window is the mapped range plus padding, so the alignment opens with and ends in a gap -- two gap openings
      // and the length difference -- and in between it pays for the divergence mashmap estimated, at 6 per differing base
      // (a mismatch costs 5), with 0.1 % and 200 on top.  Too small a guess only costs that record a second run.
      double id = r.mashmap_estimated_identity > 1.0f ? r.mashmap_estimated_identity / 100.0 : r.mashmap_estimated_identity;
      id = std::min(1.0, std::max(0.5, id));
      const double len = (double)std::min(r.target_length, r.query_length);
      const double dl = std::fabs((double)r.target_length - (double)r.query_length);
      static const double per_base = getenv("WFM_HINT_PER_BASE") ? atof(getenv("WFM_HINT_PER_BASE")) : 6.0;
      static const double id_slack = getenv("WFM_HINT_ID_SLACK") ? atof(getenv("WFM_HINT_ID_SLACK")) : 0.001;
      static const double konst = getenv("WFM_HINT_CONST") ? atof(getenv("WFM_HINT_CONST")) : 200.0;
      const double hint = 2.0 * penalties.gap_opening2 + penalties.gap_extension2 * dl + (1.0 - id + id_slack) * len * per_base + konst;
      p.score_hint = hint < 1e9 ? (int32_t)hint : 0;
    }
    g.probs.push_back(p);
  }
  int rc = g.run(h, pen, stats);
  if (rc < 0) return rc;
  const auto ts1 = now();
  const int nt = fmt.threads;
  for_each_record(recs.size(), nt, [&](size_t i) {
    recs[i].ok = (g.res[i].status == 0);  // status != 0: the reference drops the record silently (wflign.cpp:150-152)
    recs[i].score = g.res[i].score;
    recs[i].paf.clear();
    if (recs[i].ok) recs[i].cigar = g.cigar(i);
  });
  if (stats)
    for (const auto& r : recs) stats->main_failed += !r.ok;
  const auto ts2 = now();
  if (!disable_chain_patching) {
    // ---- stage 2: head patches (wflign.cpp:241-320) ----
    std::vector<size_t> owner;
    std::vector<Erosion> ero;
    g.probs.clear();
    std::vector<Erosion> scanned(recs.size());
    for_each_record(recs.size(), nt, [&](size_t i) { if (recs[i].ok) scanned[i] = scan_head_erosion(recs[i].cigar); });
    for (size_t i = 0; i < recs.size(); ++i) {
      if (!recs[i].ok) continue;
      const Erosion& e = scanned[i];
      if (e.query_eroded > 3 || e.target_eroded > 3) {
        wfm_problem_t p{};
        p.pattern = recs[i].target; p.plen = (int32_t)e.target_eroded;
        p.text = recs[i].query; p.tlen = (int32_t)e.query_eroded;
        p.mode = WFM_MODE_ENDSFREE;
        p.pattern_begin_free = (int32_t)e.target_eroded; p.pattern_end_free = 0;
        p.text_begin_free = (int32_t)e.query_eroded; p.text_end_free = 0;
        g.probs.push_back(p); owner.push_back(i); ero.push_back(e);
      }
    }
    rc = g.run(h, pen, stats);
    if (rc < 0) return rc;
    for_each_record(owner.size(), nt, [&](size_t j) {
      if (g.res[j].status != 0) return;
      BiwfaRecord& r = recs[owner[j]];
      std::string head = erode_short_matches_in_cigar(g.cigar(j), 3, true);
      r.cigar = merge_adjacent_ops(head, r.cigar.substr(ero[j].erode_end_pos));
    });
    if (stats)
      for (size_t j = 0; j < owner.size(); ++j) stats->head_patches += g.res[j].status == 0;
    // ---- stage 3: tail patches (wflign.cpp:323-418), on the head-patched CIGAR ----
    owner.clear(); ero.clear(); g.probs.clear();
    std::vector<CigarOps> parsed;
    std::vector<CigarOps> all_ops(recs.size());
    for_each_record(recs.size(), nt, [&](size_t i) {
      if (!recs[i].ok) return;
      all_ops[i] = parse_cigar(recs[i].cigar);
      scanned[i] = scan_tail_erosion(all_ops[i]);
    });
    for (size_t i = 0; i < recs.size(); ++i) {
      if (!recs[i].ok) continue;
      CigarOps& ops = all_ops[i];
      const Erosion& e = scanned[i];
      if (e.query_eroded > 3 || e.target_eroded > 3) {
        wfm_problem_t p{};
        p.pattern = recs[i].target + recs[i].target_length - e.target_eroded; p.plen = (int32_t)e.target_eroded;
        p.text = recs[i].query + recs[i].query_length - e.query_eroded; p.tlen = (int32_t)e.query_eroded;
        p.mode = WFM_MODE_ENDSFREE;
        p.pattern_begin_free = 0; p.pattern_end_free = (int32_t)e.target_eroded;
        p.text_begin_free = 0; p.text_end_free = (int32_t)e.query_eroded;
        g.probs.push_back(p); owner.push_back(i); ero.push_back(e); parsed.push_back(std::move(ops));
      }
    }
    rc = g.run(h, pen, stats);
    if (rc < 0) return rc;
    for_each_record(owner.size(), nt, [&](size_t j) {
      if (g.res[j].status != 0) return;
      BiwfaRecord& r = recs[owner[j]];
      std::string tail = erode_short_matches_in_cigar(g.cigar(j), 3, false);
      CigarOps keep(parsed[j].begin(), parsed[j].begin() + (long)ero[j].erode_start_idx);
      r.cigar = merge_adjacent_ops(cigar_to_string(keep), tail);
    });
    if (stats)
      for (size_t j = 0; j < owner.size(); ++j) stats->tail_patches += g.res[j].status == 0;
  }
  const auto ts3 = now();
  // ---- stage 4: swizzle + PAF (wflign.cpp:423-454) ----
  for_each_record(recs.size(), nt, [&](size_t ri) {
    BiwfaRecord& r = recs[ri];
    if (!r.ok) return;
    const std::string q(r.query, r.query_length);
    const std::string t(r.target, r.target_avail ? r.target_avail : r.target_length);
    std::string sw = try_swap_start_pattern(r.cigar, q, t, 0, 0);
    if (sw != r.cigar) r.cigar = sw;
    sw = try_swap_end_pattern(r.cigar, q, t, 0, 0);
    if (sw != r.cigar) r.cigar = sw;
    if (fmt.paf_format_else_sam)
      write_alignment_paf(r.paf, r.cigar, r.query_name, r.query_total_length, r.query_offset, r.query_length,
                          r.query_is_rev, r.target_name, r.target_total_length, r.target_offset, pp,
                          r.mashmap_estimated_identity, r.chain_id, r.chain_length, r.chain_pos);
    else
      write_alignment_sam(r.paf, r.cigar, r.query_name, r.query_offset, r.query_is_rev, r.target_name, r.target_offset, pp,
                          r.mashmap_estimated_identity, fmt.no_seq_in_sam, fmt.emit_md_tag, r.query, r.target,
                          r.chain_id, r.chain_length, r.chain_pos);
  });
  if (dbg)
    fprintf(stderr, "[wflign] %zu records: main alignment call %.1f ms (incl. waiting for the device), CIGAR strings %.1f ms, patches %.1f ms, swizzle + records %.1f ms\n",
            recs.size(), ms(ts0, ts1), ms(ts1, ts2), ms(ts2, ts3), ms(ts3, now()));
  return 0;
}

}  // namespace wflign
