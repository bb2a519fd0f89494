// fasta.hpp -- FASTA access for the map and align drivers.
//
// Replaces the reference's faigz/htslib layer (src/common/faigz.h:221-505: faidx_meta_load,
// faidx_meta_seq_len, faidx_reader_fetch_seq) with what the hot path needs: sequence names in
// file order, lengths, and substring fetches.
//
// Two modes, chosen per file:
//  * indexed -- `<path>.fai` exists and the file is plain text or BGZF: nothing is read up front.
//    fetch() is true random access (pread of the lines that hold the range; for BGZF the blocks
//    that hold it, found through `<path>.gzi` or, without one, a scan of the block headers);
//    sequence(i) loads one whole sequence on first use, thread-safe, so callers can load many
//    side by side (preload()).  A long sequence (a chromosome) goes into a block of its own -- 2 MB
//    aligned and marked for huge pages -- that several threads fill side by side, each its own
//    range of the bases (the .fai gives every base its place in the file): one thread strips the
//    line ends of about 1 GB/s, and first-touching a quarter of a gigabyte of 4 kB pages costs as
//    much again.
//  * in-memory -- no .fai, or a gzip stream that is not BGZF: the file is read once through zlib.
//    (htslib would build the .fai here, FAI_CREATE; this reader does not write next to its inputs.)
// Both modes return identical bytes; tests/test_fasta_cpu.py holds them against each other.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace wfmash_host {

// the bases of a whole sequence, held by the FastaStore they came from
struct SeqView {
  const char* p = nullptr;
  size_t n = 0;
  const char* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
};

class FastaStore {
 public:
  // Throws std::runtime_error if the file (or a present but unusable index) cannot be read.
  explicit FastaStore(const std::string& path);
  ~FastaStore();
  FastaStore(const FastaStore&) = delete;
  FastaStore& operator=(const FastaStore&) = delete;

  int nseq() const { return (int)names_.size(); }
  const std::string& name(int i) const { return names_[i]; }
  int64_t length(int i) const { return lens_[i]; }
  int64_t seq_len(const std::string& name) const;  // -1 if absent (faidx_meta_seq_len)
  // Bases [start, end_inclusive] of `name` (faigz uses an inclusive end), clamped
  // to the sequence; empty string if absent.
  std::string fetch(const std::string& name, int64_t start, int64_t end_inclusive) const;
  // The whole sequence; in indexed mode it is read on first use and kept.  helpers: further threads a long sequence
  // may be read with (-1: up to 15, as the machine has them).
  SeqView sequence(int i, int helpers = -1) const;
  // Reads the listed sequences (all if empty) with up to `threads` readers: sequences side by side, and the threads
  // that are left over inside the long ones.
  void preload(const std::vector<int>& which, int threads) const;
  // index of `name`, -1 if absent
  int find(const std::string& name) const { auto it = index_.find(name); return it == index_.end() ? -1 : it->second; }
  bool indexed() const { return fd_ >= 0; }
  // is the file (and its .fai) still the one this store was opened on?  (device, inode, size, modification time)
  bool same_file() const;
  // bytes of whole sequences held in memory right now
  int64_t resident_bytes() const;

 private:
  struct FaiEntry { int64_t offset, line_bases, line_width; };
  void load_stream(const std::string& path);
  bool open_indexed(const std::string& path);
  // bytes [off, off+n) of the uncompressed file
  void read_text(int64_t off, int64_t n, char* dst) const;
  // bases [start, end) of sequence i appended to out / written to dst
  void read_bases(int i, int64_t start, int64_t end, std::string& out) const;
  void read_bases_to(int i, int64_t start, int64_t end, char* dst) const;
  void load_block(int i, int helpers) const;
  struct Block { char* base = nullptr; size_t map_bytes = 0; char* p = nullptr; };  // an anonymous mapping; p: 2 MB aligned

  struct FileId { uint64_t dev = 0, ino = 0; int64_t size = -1, mtime_ns = 0; bool operator==(const FileId& o) const { return dev == o.dev && ino == o.ino && size == o.size && mtime_ns == o.mtime_ns; } };
  static FileId file_id(const std::string& path);
  FileId id_, id_fai_, id_gzi_;
  std::string path_;
  std::vector<std::string> names_;
  std::vector<int64_t> lens_;
  std::unordered_map<std::string, int> index_;
  mutable std::vector<std::string> seqs_;
  mutable std::vector<Block> blocks_;  // indexed mode: the long sequences (seqs_[i] stays empty)
  // indexed mode
  int fd_ = -1;
  bool bgzf_ = false;
  std::vector<FaiEntry> fai_;
  std::vector<int64_t> block_coff_, block_uoff_;  // BGZF: start of each block in the file / in the text
  int64_t text_size_ = 0;
  mutable std::unique_ptr<std::once_flag[]> once_;
  mutable std::unique_ptr<std::atomic<bool>[]> loaded_;  // indexed mode: seqs_[i] holds the whole sequence (set after the load)
};

// One FastaStore per file for as long as somebody holds it: the identity estimate and the mapper of one run ask for the same
// files one after the other (main.cpp:72-128 then computeMap.hpp:147-230), and a whole-sequence load of a pangenome is
// gigabytes -- whoever spans both (wfmh_map_multi) keeps the pointers, and the second one finds the sequences loaded.
std::shared_ptr<FastaStore> open_shared(const std::string& path);
// Drops the pointers on a thread of its own (WFM_FASTA_RELEASE_LATER=0: here and now): handing a few gigabytes of pages
// back to the system takes a tenth of a second that the caller need not wait for.
void release_later(std::vector<std::shared_ptr<FastaStore>> files);
// The stores of the call that just ended stay open until the next call hands in its own (or release_kept()): the align
// phase that follows a map phase asks for the same files, and finds the sequences loaded (open_shared; a file that was
// rewritten in between is opened afresh).  Stores holding more than WFM_FASTA_KEEP_GB (32) are let go at once,
// WFM_FASTA_KEEP=0 keeps nothing.
void keep_until_next(std::vector<std::shared_ptr<FastaStore>> files);
void release_kept();

}  // namespace wfmash_host
