#include "fasta.hpp"

#include <zlib.h>

#include <stdexcept>

namespace wfmash_host {

FastaStore::FastaStore(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open FASTA: " + path);
  gzbuffer(f, 1 << 20);
  std::vector<char> buf(1 << 20);
  std::string* cur = nullptr;
  bool in_header = false;
  std::string header;
  int n;
  while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
    for (int i = 0; i < n; ++i) {
      const char c = buf[i];
      if (in_header) {
        if (c == '\n') {
          in_header = false;
          // name = header up to the first whitespace (faidx convention)
          size_t e = 0;
          while (e < header.size() && header[e] != ' ' && header[e] != '\t' && header[e] != '\r') ++e;
          names_.push_back(header.substr(0, e));
          seqs_.emplace_back();
          cur = &seqs_.back();
          header.clear();
        } else {
          header.push_back(c);
        }
      } else if (c == '>') {
        in_header = true;
      } else if (c != '\n' && c != '\r') {
        if (cur) cur->push_back(c);
      }
    }
  }
  gzclose(f);
  for (size_t i = 0; i < names_.size(); ++i) index_.emplace(names_[i], (int)i);
}

int64_t FastaStore::seq_len(const std::string& name) const {
  auto it = index_.find(name);
  return it == index_.end() ? -1 : (int64_t)seqs_[it->second].size();
}

std::string FastaStore::fetch(const std::string& name, int64_t start, int64_t end_inclusive) const {
  auto it = index_.find(name);
  if (it == index_.end()) return std::string();
  const std::string& s = seqs_[it->second];
  if (start < 0) start = 0;
  int64_t end = end_inclusive + 1;
  if (end > (int64_t)s.size()) end = (int64_t)s.size();
  if (start >= end) return std::string();
  return s.substr((size_t)start, (size_t)(end - start));
}

}  // namespace wfmash_host
