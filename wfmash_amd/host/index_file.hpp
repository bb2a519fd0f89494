// index_file.hpp -- the on-disk index of the map phase (`-W` write, `-I` read), SURVEY 8f-2.
//
// Layout of the reference (src/map/include/winSketch.hpp:569-660 write side, :682-960 read side;
// sequenceIds.hpp:101-212 for the id section).  A file is one sub-index per target subset, appended
// one after the other; native endianness and type widths (x86-64: size_t = 8 bytes):
//   u64  magic 0xDEADBEEFCAFEBABE
//   u64  batch_idx, u64 total_batches, i64 batch_size (-b)
//   u64  n_names, then (u64 length, bytes) per target name of the subset
//   id section: u64 n, then (u64 length, bytes, i32 id) per sequence, then i32 next id
//   i64  windowLength, i32 sketchSize, i32 kmerSize
//   u64  n_minmers, then n x MinmerInfo (32 B: u64 hash, i64 wpos, i64 wpos_end, i32 seqId, i16 strand, 2 B padding)
//   u64  n_keys, then per key: u64 hash, u64 n_points, n x IntervalPoint (24 B: i64 pos, u64 hash, i32 seqId,
//        i8 side, 3 B padding)
// The reference writes the keys in the iteration order of its hash map, i.e. in the order its indexing threads
// first met them, which depends on its thread count; this writer uses the order of a key's first interval in
// minmerIndex (what one indexing thread produces).  Readers -- the reference's and this one -- take any order.
// Padding bytes are zero here (the reference leaves them uninitialised).
#pragma once

#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"
#include "map_types.hpp"
#include "sequence_ids.hpp"

namespace skch {

constexpr uint64_t kIndexMagic = 0xDEADBEEFCAFEBABEull;

struct SubIndex {
  uint64_t batch_idx = 0, total_batches = 1;
  int64_t batch_size = 0;
  std::vector<std::string> names;              // target sequences of the subset
  offset_t windowLength = 0;
  int sketchSize = 0, kmerSize = 0;
  std::vector<wfm_minmer_t> minmers;           // minmerIndex
  std::vector<uint64_t> uhash;                 // keys, ascending (as wfm_index_download / wfm_index_upload hold them)
  std::vector<int64_t> poff;                   // uhash.size() + 1 offsets into points
  std::vector<wfm_interval_point_t> points;
};

// appends one sub-index; throws std::runtime_error on I/O errors
void write_sub_index(std::ostream& out, const SubIndex& ix, const SequenceIdManager& ids);
// reads the next sub-index (the id section goes into `ids`); throws std::runtime_error when the stream does not
// hold one
void read_sub_index(std::istream& in, SubIndex& ix, SequenceIdManager& ids);
// batch size and number of subsets of an index file (what Map::mapQuery peeks at, computeMap.hpp:349-376)
void peek_index_file(const std::string& path, int64_t* batch_size, uint64_t* total_batches);

}  // namespace skch
