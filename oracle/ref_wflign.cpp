// oracle/ref_wflign.cpp -- TEST INFRASTRUCTURE ONLY.
// Thin C wrapper around the REFERENCE'S OWN wflign sources, compiled where they lie (oracle/Makefile, target `ref`):
//   src/common/wflign/src/{wflign.cpp, wflign_patch.cpp, wflign_alignment.cpp, wflign_swizzle.cpp, rkmh.cpp,
//   murmur3.cpp, lodepng.cpp} and deps/atomic_image.cpp
// The only header they need that the reference tree does not hold is WFA2-lib's `WFAligner.hpp`
// (wflign_alignment.hpp:7); the include path gives them the PRODUCT's drop-in for that seam,
// wfmash_amd/host/WFAligner.hpp, so every alignment the reference code asks for runs on the GPU through
// wfm_align_batch, and everything around it -- do_biwfa_alignment's erosion / patching / merging (wflign.cpp:19-483),
// the swizzle (wflign_swizzle.cpp), write_alignment_paf / write_alignment_sam / the MD tag (wflign_patch.cpp) -- is
// the reference's own code.  `wflign_git_version.hpp` is produced by the reference's own
// scripts/generate_git_version.sh into oracle/_ref/gen/ (it only feeds the @PG line of a SAM header).
// Nothing of the reference is copied into this repository.
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

#include "wflign.hpp"

extern "C" {

// One mapping record through the reference's do_biwfa_alignment (wflign.cpp:108-483), called the way
// Aligner::processAlignment does (computeAlignments.hpp:661-723): `target_buf` is the padded reference window (NUL
// terminated, the swizzle reads past the mapping's end), the mapping starts `target_skip` bases into it.
// Returns the record text (malloc'd, may be empty); free with ref_wflign_free.
char* ref_do_biwfa_alignment(const char* query_name, const char* query, uint64_t query_total_length, uint64_t query_offset,
                             uint64_t query_length, int query_is_rev, const char* target_name, const char* target_buf,
                             uint64_t target_skip, uint64_t target_total_length, uint64_t target_offset, uint64_t target_length,
                             int mismatch, int o1, int e1, int o2, int e2, int emit_md_tag, int paf_format_else_sam,
                             int no_seq_in_sam, int disable_chain_patching, float min_identity, uint64_t min_alignment_length,
                             float min_block_identity, uint64_t wflign_max_len_minor, float mashmap_estimated_identity,
                             int32_t chain_id, int32_t chain_length, int32_t chain_pos) {
  std::string text;
  try {
    // the reference hands out writable buffers (char* const): keep private copies
    std::string q(query, query_length);
    std::string t(target_buf);
    wflign_penalties_t pen;
    pen.match = 0; pen.mismatch = mismatch; pen.gap_opening1 = o1; pen.gap_extension1 = e1; pen.gap_opening2 = o2; pen.gap_extension2 = e2;
    std::stringstream out;
    wflign::wavefront::do_biwfa_alignment(query_name, &q[0], query_total_length, query_offset, query_length, query_is_rev != 0,
                                          target_name, &t[0] + target_skip, target_total_length, target_offset, target_length, out, pen,
                                          emit_md_tag != 0, paf_format_else_sam != 0, no_seq_in_sam != 0, disable_chain_patching != 0,
                                          min_identity, min_alignment_length, min_block_identity, wflign_max_len_minor,
                                          mashmap_estimated_identity, chain_id, chain_length, chain_pos);
    text = out.str();
  } catch (const std::exception& e) {
    text = std::string("ERROR: ") + e.what();
  }
  char* r = (char*)malloc(text.size() + 1);
  if (r) memcpy(r, text.c_str(), text.size() + 1);
  return r;
}

void ref_wflign_free(char* p) { free(p); }

}  // extern "C"
