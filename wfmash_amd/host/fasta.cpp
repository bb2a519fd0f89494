#include "fasta.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <thread>

namespace wfmash_host {

namespace {

void pread_all(int fd, void* dst, size_t n, int64_t off, const std::string& path) {
  char* p = static_cast<char*>(dst);
  while (n > 0) {
    const ssize_t r = pread(fd, p, n, (off_t)off);
    if (r <= 0) throw std::runtime_error("short read from " + path + " (stale .fai/.gzi?)");
    p += r; n -= (size_t)r; off += r;
  }
}

uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t le32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t le64(const unsigned char* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// A BGZF block is a gzip member whose extra field carries the subfield 'B','C' = total block size - 1
// (SAM specification, section 4.1).  Returns the block size and where the deflate data begins; 0 if the
// bytes at `off` are not a BGZF block header.
int64_t bgzf_block_size(int fd, int64_t off, int64_t file_size, int* data_begin) {
  unsigned char h[12];
  if (off + 12 > file_size) return 0;
  if (pread(fd, h, 12, (off_t)off) != 12) return 0;
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
  const int xlen = le16(h + 10);
  if (off + 12 + xlen > file_size) return 0;
  std::vector<unsigned char> x((size_t)xlen);
  if (xlen && pread(fd, x.data(), (size_t)xlen, (off_t)(off + 12)) != xlen) return 0;
  for (int p = 0; p + 4 <= xlen;) {
    const int slen = le16(x.data() + p + 2);
    if (x[(size_t)p] == 'B' && x[(size_t)p + 1] == 'C' && slen == 2 && p + 6 <= xlen) {
      if (data_begin) *data_begin = 12 + xlen;
      return (int64_t)le16(x.data() + p + 4) + 1;
    }
    p += 4 + slen;
  }
  return 0;
}

}  // namespace

FastaStore::FileId FastaStore::file_id(const std::string& path) {
  FileId f;
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0) return f;  // absent: size -1
  f.dev = (uint64_t)sb.st_dev; f.ino = (uint64_t)sb.st_ino; f.size = (int64_t)sb.st_size;
  f.mtime_ns = (int64_t)sb.st_mtim.tv_sec * 1000000000ll + (int64_t)sb.st_mtim.tv_nsec;
  return f;
}

bool FastaStore::same_file() const { return file_id(path_) == id_ && file_id(path_ + ".fai") == id_fai_ && file_id(path_ + ".gzi") == id_gzi_; }

int64_t FastaStore::resident_bytes() const {
  int64_t t = 0;
  for (size_t i = 0; i < seqs_.size(); ++i) {
    if (fd_ >= 0 && !loaded_[i].load(std::memory_order_acquire)) continue;
    t += fd_ >= 0 && blocks_[i].p ? (int64_t)blocks_[i].map_bytes : (int64_t)seqs_[i].capacity();
  }
  return t;
}

FastaStore::FastaStore(const std::string& path) : path_(path) {
  id_ = file_id(path);
  id_fai_ = file_id(path + ".fai");
  id_gzi_ = file_id(path + ".gzi");  // (BGZF inputs: the block table is part of what the store was opened on; absent: size -1 on both sides)
  if (!open_indexed(path)) load_stream(path);
  for (size_t i = 0; i < names_.size(); ++i) index_.emplace(names_[i], (int)i);  // the first of equal names wins, as in faidx
}

FastaStore::~FastaStore() {
  if (fd_ >= 0) close(fd_);
  // (in pieces: an unmap holds the address space's lock, and other threads' page faults with it, for as long as it takes)
  const size_t piece = (size_t)64 << 20;
  for (Block& b : blocks_)
    for (size_t off = 0; b.base && off < b.map_bytes; off += piece) munmap(b.base + off, std::min(piece, b.map_bytes - off));
}

bool FastaStore::open_indexed(const std::string& path) {
  std::ifstream fai(path + ".fai");
  if (!fai.is_open()) return false;
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) throw std::runtime_error("cannot open FASTA: " + path);
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); throw std::runtime_error("cannot stat FASTA: " + path); }
  const int64_t file_size = (int64_t)sb.st_size;
  unsigned char magic[2] = {0, 0};
  const bool gz = file_size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
  if (gz) {
    int db = 0;
    if (bgzf_block_size(fd, 0, file_size, &db) == 0) { close(fd); return false; }  // plain gzip: no random access, stream it
    // block table: <path>.gzi = u64 count, then (compressed offset, uncompressed offset) of every block but the first
    std::ifstream gzi(path + ".gzi", std::ios::binary);
    bool have = false;
    if (gzi.is_open()) {
      unsigned char b[16];
      if (gzi.read(reinterpret_cast<char*>(b), 8)) {
        const uint64_t n = le64(b);
        block_coff_.assign(1, 0); block_uoff_.assign(1, 0);
        have = true;
        for (uint64_t i = 0; i < n; ++i) {
          if (!gzi.read(reinterpret_cast<char*>(b), 16)) { have = false; break; }
          block_coff_.push_back((int64_t)le64(b)); block_uoff_.push_back((int64_t)le64(b + 8));
        }
        if (have && (block_coff_.back() >= file_size || !std::is_sorted(block_coff_.begin(), block_coff_.end()))) have = false;
      }
    }
    if (have) {
      // the text ends after the last data block; an index may or may not list the empty EOF block
      int64_t off = block_coff_.back(), uoff = block_uoff_.back();
      while (off < file_size) {
        const int64_t bs = bgzf_block_size(fd, off, file_size, nullptr);
        if (bs == 0) { close(fd); throw std::runtime_error("corrupt BGZF block in " + path); }
        unsigned char t[4];
        pread_all(fd, t, 4, off + bs - 4, path);
        off += bs;
        uoff += le32(t);
        if (off < file_size) { block_coff_.push_back(off); block_uoff_.push_back(uoff); }
      }
      text_size_ = uoff;
    } else {
      block_coff_.clear(); block_uoff_.clear();
      int64_t off = 0, uoff = 0;
      while (off < file_size) {
        const int64_t bs = bgzf_block_size(fd, off, file_size, nullptr);
        if (bs == 0) { close(fd); throw std::runtime_error("corrupt BGZF block in " + path); }
        unsigned char t[4];
        pread_all(fd, t, 4, off + bs - 4, path);
        block_coff_.push_back(off); block_uoff_.push_back(uoff);
        off += bs;
        uoff += le32(t);
      }
      text_size_ = uoff;
    }
    bgzf_ = true;
  } else {
    text_size_ = file_size;
  }
  std::string line;
  while (std::getline(fai, line)) {
    if (line.empty()) continue;
    std::istringstream is(line);
    std::string name;
    int64_t len = 0;
    FaiEntry e{0, 0, 0};
    std::getline(is, name, '\t');
    if (!(is >> len >> e.offset >> e.line_bases >> e.line_width) || len < 0 || e.offset < 0 || e.line_width < e.line_bases ||
        (len > 0 && e.line_bases <= 0)) {
      close(fd);
      throw std::runtime_error("malformed line in " + path + ".fai: " + line);
    }
    if (len > 0) {
      const int64_t last = e.offset + (len - 1) / e.line_bases * e.line_width + (len - 1) % e.line_bases;
      if (last >= text_size_) { close(fd); throw std::runtime_error(path + ".fai does not match the FASTA (sequence " + name + " ends past the file)"); }
    }
    names_.push_back(name); lens_.push_back(len); fai_.push_back(e);
  }
  fd_ = fd;
  seqs_.resize(names_.size());
  blocks_.resize(names_.size());
  once_.reset(new std::once_flag[names_.size()]);
  loaded_.reset(new std::atomic<bool>[names_.size()]);
  for (size_t i = 0; i < names_.size(); ++i) loaded_[i].store(false, std::memory_order_relaxed);
  return true;
}

void FastaStore::load_stream(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open FASTA: " + path);
  gzbuffer(f, 1 << 20);
  std::vector<char> buf(1 << 20);
  std::string* cur = nullptr;
  bool in_header = false;
  std::string header;
  int n;
  while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) {
    int i = 0;
    while (i < n) {
      if (in_header) {
        const char* nl = static_cast<const char*>(memchr(buf.data() + i, '\n', (size_t)(n - i)));
        const int e = nl ? (int)(nl - buf.data()) : n;
        header.append(buf.data() + i, (size_t)(e - i));
        i = e;
        if (nl) {
          ++i;
          in_header = false;
          // name = header up to the first whitespace (faidx convention)
          size_t w = 0;
          while (w < header.size() && header[w] != ' ' && header[w] != '\t' && header[w] != '\r') ++w;
          names_.push_back(header.substr(0, w));
          seqs_.emplace_back();
          cur = &seqs_.back();
          header.clear();
        }
      } else if (buf[(size_t)i] == '>') {
        in_header = true;
        ++i;
      } else {
        // a run of sequence bytes up to the end of the line
        const char* nl = static_cast<const char*>(memchr(buf.data() + i, '\n', (size_t)(n - i)));
        int e = nl ? (int)(nl - buf.data()) : n;
        int stop = e;
        if (stop > i && buf[(size_t)stop - 1] == '\r') --stop;
        if (cur && stop > i) cur->append(buf.data() + i, (size_t)(stop - i));
        i = nl ? e + 1 : n;
      }
    }
  }
  gzclose(f);
  for (const auto& s : seqs_) lens_.push_back((int64_t)s.size());
}

void FastaStore::read_text(int64_t off, int64_t n, char* dst) const {
  if (n <= 0) return;
  if (off < 0 || off + n > text_size_) throw std::runtime_error("read past the end of " + path_);
  if (!bgzf_) { pread_all(fd_, dst, (size_t)n, off, path_); return; }
  size_t b = (size_t)(std::upper_bound(block_uoff_.begin(), block_uoff_.end(), off) - block_uoff_.begin()) - 1;
  std::vector<unsigned char> comp, plain;
  struct stat sb;
  fstat(fd_, &sb);
  while (n > 0) {
    if (b >= block_coff_.size()) throw std::runtime_error("BGZF block table of " + path_ + " is too short");
    int db = 0;
    const int64_t bs = bgzf_block_size(fd_, block_coff_[b], (int64_t)sb.st_size, &db);
    if (bs == 0) throw std::runtime_error("corrupt BGZF block in " + path_);
    comp.resize((size_t)bs);
    pread_all(fd_, comp.data(), (size_t)bs, block_coff_[b], path_);
    const uint32_t isize = le32(comp.data() + bs - 4);
    plain.resize(isize);
    if (isize) {
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib inflateInit2 failed");
      zs.next_in = comp.data() + db;
      zs.avail_in = (uInt)(bs - db - 8);
      zs.next_out = plain.data();
      zs.avail_out = isize;
      const int zr = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (zr != Z_STREAM_END || zs.total_out != isize) throw std::runtime_error("corrupt BGZF block in " + path_);
    }
    const int64_t in_block = off - block_uoff_[b];
    const int64_t take = std::min<int64_t>(n, (int64_t)isize - in_block);
    if (take > 0) {
      memcpy(dst, plain.data() + in_block, (size_t)take);
      dst += take; off += take; n -= take;
    }
    ++b;
  }
}

void FastaStore::read_bases_to(int i, int64_t start, int64_t end, char* dst) const {
  if (start >= end) return;
  const FaiEntry& e = fai_[(size_t)i];
  const int64_t first = e.offset + start / e.line_bases * e.line_width + start % e.line_bases;
  const int64_t last = e.offset + (end - 1) / e.line_bases * e.line_width + (end - 1) % e.line_bases;  // inclusive
  // in slices that stay in the cache between the read and the pass that drops the line ends
  const int64_t slice = 4ll << 20;
  std::vector<char> buf((size_t)std::min(slice, last + 1 - first));
  for (int64_t off = first; off <= last; off += slice) {
    const int64_t n = std::min(slice, last + 1 - off);
    read_text(off, n, buf.data());
    // keep the bytes whose position within its line is < line_bases
    int64_t col = (off - e.offset) % e.line_width;
    int64_t p = 0;
    while (p < n) {
      if (col < e.line_bases) {
        const int64_t run = std::min(e.line_bases - col, n - p);
        memcpy(dst, buf.data() + p, (size_t)run);
        dst += run; p += run; col += run;
      } else {
        const int64_t skip = std::min(e.line_width - col, n - p);
        p += skip; col += skip;
      }
      if (col == e.line_width) col = 0;
    }
  }
}

void FastaStore::read_bases(int i, int64_t start, int64_t end, std::string& out) const {
  if (start >= end) return;
  const size_t at = out.size();
  out.resize(at + (size_t)(end - start));
  read_bases_to(i, start, end, &out[at]);
}

namespace {
constexpr int64_t kHuge = 2ll << 20;
// sequences from this length on get a block of their own (WFM_FASTA_BLOCK_MIN: the tests set it low)
int64_t block_min() {
  const char* e = getenv("WFM_FASTA_BLOCK_MIN");
  return e ? std::max<int64_t>(1, atoll(e)) : (int64_t)8 << 20;
}
// the ranges the readers of one block take: whole huge pages (WFM_FASTA_BLOCK_ALIGN: the tests set it low)
int64_t block_align() {
  const char* e = getenv("WFM_FASTA_BLOCK_ALIGN");
  return e ? std::max<int64_t>(1, atoll(e)) : kHuge;
}
}  // namespace

// a long sequence into a block of its own, filled by 1 + helpers threads
void FastaStore::load_block(int i, int helpers) const {
  const int64_t len = lens_[(size_t)i];
  Block b;
  b.map_bytes = (size_t)((len + kHuge - 1) / kHuge * kHuge + kHuge);
  void* m = mmap(nullptr, b.map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) throw std::bad_alloc();
  b.base = static_cast<char*>(m);
  b.p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(b.base) + (uintptr_t)kHuge - 1) / (uintptr_t)kHuge * (uintptr_t)kHuge);
  const char* he = getenv("WFM_FASTA_HUGE");
  const bool huge = he ? atoi(he) != 0 : true;
  // a hint (4 kB pages work as well, only slower): on the MI355X host the identity estimate of 9 x 249 Mbp took 130 - 134 ms
  // with it and 161 - 250 ms without, the whole map call of a C4 rank 0.73 s against 0.80 - 0.94 s (WFM_FASTA_HUGE=0 turns it off)
  if (huge) (void)madvise(b.p, b.map_bytes - (size_t)(b.p - b.base), MADV_HUGEPAGE);
  const int64_t al = block_align();
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)helpers + 1, (len + al - 1) / al));
  std::vector<int64_t> cut((size_t)T + 1);
  for (int t = 0; t <= T; ++t) cut[(size_t)t] = t == T ? len : len / T * t / al * al;
  std::vector<std::string> errors((size_t)T);
  auto part = [&](int t) {
    try { read_bases_to(i, cut[(size_t)t], cut[(size_t)t + 1], b.p + cut[(size_t)t]); }
    catch (const std::exception& e) { errors[(size_t)t] = e.what(); if (errors[(size_t)t].empty()) errors[(size_t)t] = "read failed"; }
  };
  {
    std::vector<std::thread> pool;
    try {
      for (int t = 1; t < T; ++t) pool.emplace_back(part, t);
    } catch (const std::exception&) {  // no more threads to be had: this one does the rest
      for (int t = (int)pool.size() + 1; t < T; ++t) part(t);
    }
    part(0);
    for (auto& th : pool) th.join();
  }
  for (const auto& e : errors)
    if (!e.empty()) { munmap(b.base, b.map_bytes); throw std::runtime_error(e); }
  blocks_[(size_t)i] = b;
}

int64_t FastaStore::seq_len(const std::string& name) const {
  auto it = index_.find(name);
  return it == index_.end() ? -1 : lens_[(size_t)it->second];
}

SeqView FastaStore::sequence(int i, int helpers) const {
  if (fd_ >= 0)
    std::call_once(once_[(size_t)i], [&] {
      const int64_t len = lens_[(size_t)i];
      if (len >= block_min()) {
        if (helpers < 0) helpers = (int)std::min<unsigned>(15u, std::max(1u, std::thread::hardware_concurrency()) - 1u);
        load_block(i, helpers);
      } else {
        std::string s;
        read_bases(i, 0, len, s);
        seqs_[(size_t)i] = std::move(s);
      }
      loaded_[(size_t)i].store(true, std::memory_order_release);  // fetch() may look at the sequence from now on
    });
  if (fd_ >= 0 && blocks_[(size_t)i].p) return SeqView{blocks_[(size_t)i].p, (size_t)lens_[(size_t)i]};
  return SeqView{seqs_[(size_t)i].data(), seqs_[(size_t)i].size()};
}

void FastaStore::preload(const std::vector<int>& which, int threads) const {
  if (fd_ < 0) return;
  std::vector<int> todo = which;
  if (todo.empty())
    for (int i = 0; i < nseq(); ++i) todo.push_back(i);
  // longest first
  std::sort(todo.begin(), todo.end(), [&](int a, int b) { return lens_[(size_t)a] > lens_[(size_t)b]; });
  // the threads go to the bases: a long sequence gets its share of them as readers of its block, and so many sequences
  // are in the making at a time that the shares add up to `threads`
  int64_t total = 0;
  for (int i : todo) total += lens_[(size_t)i];
  const int T = std::max(1, threads);
  std::vector<int> helpers(todo.size(), 0);
  int nt = (int)std::min<size_t>((size_t)T, todo.size());
  if (total > 0 && lens_[(size_t)todo[0]] >= block_min()) {
    int used = 0;
    nt = 0;
    for (size_t j = 0; j < todo.size(); ++j) {
      const int share = (int)std::max<int64_t>(1, std::min<int64_t>(T, (int64_t)((double)lens_[(size_t)todo[j]] / (double)total * T)));
      helpers[j] = share - 1;
      if (used < T) { used += share; ++nt; }
    }
  }
  std::atomic<size_t> next{0};
  auto work = [&] {
    for (size_t j; (j = next.fetch_add(1)) < todo.size();) (void)sequence(todo[j], helpers[j]);
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

std::string FastaStore::fetch(const std::string& name, int64_t start, int64_t end_inclusive) const {
  auto it = index_.find(name);
  if (it == index_.end()) return std::string();
  const int i = it->second;
  if (start < 0) start = 0;
  int64_t end = end_inclusive + 1;
  if (end > lens_[(size_t)i]) end = lens_[(size_t)i];
  if (start >= end) return std::string();
  // a whole sequence that another thread is loading right now (sequence(i)) is not touched: this fetch reads its range
  // from the file instead
  if (fd_ < 0) return seqs_[(size_t)i].substr((size_t)start, (size_t)(end - start));
  if (loaded_[(size_t)i].load(std::memory_order_acquire)) {
    const Block& b = blocks_[(size_t)i];
    return b.p ? std::string(b.p + start, (size_t)(end - start)) : seqs_[(size_t)i].substr((size_t)start, (size_t)(end - start));
  }
  std::string out;
  read_bases(i, start, end, out);
  return out;
}

std::shared_ptr<FastaStore> open_shared(const std::string& path) {
  static std::mutex mu;
  static std::unordered_map<std::string, std::weak_ptr<FastaStore>> open_files;
  std::lock_guard<std::mutex> lk(mu);
  auto it = open_files.find(path);
  if (it != open_files.end())
    if (auto sp = it->second.lock())
      if (sp->same_file()) return sp;  // (a file rewritten since -- a kept store outlives the call it was opened in -- is opened afresh)
  auto sp = std::make_shared<FastaStore>(path);
  open_files[path] = sp;
  return sp;
}

void release_later(std::vector<std::shared_ptr<FastaStore>> files) {
  const char* e = getenv("WFM_FASTA_RELEASE_LATER");
  if (e && atoi(e) == 0) { files.clear(); return; }
  try {
    std::thread([held = std::move(files)]() mutable { held.clear(); }).detach();
  } catch (const std::exception&) {
    // no thread to be had: the caller's own `files` (moved from, or still whole) goes out of scope as before
  }
}

namespace {
// (heap singletons that are never destroyed: a handle that goes during static teardown -- an interpreter's exit -- still calls release_kept)
std::mutex& g_kept_mu = *new std::mutex;
std::vector<std::shared_ptr<FastaStore>>& g_kept = *new std::vector<std::shared_ptr<FastaStore>>;
}  // namespace

void keep_until_next(std::vector<std::shared_ptr<FastaStore>> files) {
  const char* e = getenv("WFM_FASTA_KEEP");
  const char* g = getenv("WFM_FASTA_KEEP_GB");
  const double cap_gb = g ? atof(g) : 32.0;
  int64_t bytes = 0;
  for (const auto& f : files) if (f) bytes += f->resident_bytes();
  std::vector<std::shared_ptr<FastaStore>> old;
  {
    std::lock_guard<std::mutex> lk(g_kept_mu);
    old.swap(g_kept);
    if (!(e && atoi(e) == 0) && (double)bytes <= cap_gb * 1e9) g_kept = files;
  }
  // what the new set does not hold on to goes back to the system (the same stores, as a rule, when a run repeats)
  release_later(std::move(old));
  release_later(std::move(files));
}

// (the device layer calls this when the last handle of the process goes: wfm_destroy)
}  // namespace wfmash_host
void wfm_set_last_handle_hook(void (*f)());  // csrc/wfa_handle.h
namespace wfmash_host {
static const int g_hook_registered = (wfm_set_last_handle_hook(&release_kept), 0);

void release_kept() {
  std::vector<std::shared_ptr<FastaStore>> old;
  {
    std::lock_guard<std::mutex> lk(g_kept_mu);
    old.swap(g_kept);
  }
  old.clear();
}

}  // namespace wfmash_host
