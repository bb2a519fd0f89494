// fasta.hpp -- in-memory FASTA store used by the align driver.
//
// Replaces the reference's faigz/htslib random access (src/common/faigz.h:221-505,
// faidx_meta_load / faidx_reader_fetch_seq) for the hot path's needs: sequence
// names in file order, lengths, and substring fetches.  Plain and gzip/bgzip
// FASTA are read through zlib (a BGZF file is a series of gzip members).
// Random access through .fai/.gzi without loading the file is SURVEY 8(f) "next".
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace wfmash_host {

class FastaStore {
 public:
  // Throws std::runtime_error if the file cannot be read.
  explicit FastaStore(const std::string& path);
  int nseq() const { return (int)names_.size(); }
  const std::string& name(int i) const { return names_[i]; }
  int64_t seq_len(const std::string& name) const;  // -1 if absent (faidx_meta_seq_len)
  // Bases [start, end_inclusive] of `name` (faigz uses an inclusive end), clamped
  // to the sequence; empty string if absent.
  std::string fetch(const std::string& name, int64_t start, int64_t end_inclusive) const;
  const std::string& sequence(int i) const { return seqs_[i]; }
  // index of `name`, -1 if absent
  int find(const std::string& name) const { auto it = index_.find(name); return it == index_.end() ? -1 : it->second; }

 private:
  std::vector<std::string> names_;
  std::vector<std::string> seqs_;
  std::unordered_map<std::string, int> index_;
};

}  // namespace wfmash_host
