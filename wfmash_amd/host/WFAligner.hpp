// WFAligner.hpp -- drop-in replacement for WFA2-lib's C++ binding header as wfmash includes it
// (src/common/wflign/src/wflign_alignment.hpp:7, include path src/common/wflign/CMakeLists.txt:73-74).
// Provides exactly the surface the live align path uses (wflign.cpp:136-148,280-309,368-401;
// wflign_alignment.cpp:665-678) and forwards every alignment to the C ABI of libwfmash_hip.so.
// One problem per call: this seam keeps wflign.cpp unchanged; the batch seam
// (wflign_hip.hpp) is the fast one.
//
// The reference's own wflign.cpp / wflign_patch.cpp / wflign_alignment.cpp / wflign_swizzle.cpp compile and link
// against this header unmodified (oracle/Makefile target `ref`, oracle/ref_wflign.cpp); tests/test_ref_wflign_gpu.py
// runs their do_biwfa_alignment on the GPU through it.  The members the dormant hierarchical WFlign class uses
// (wflign.cpp:1097-1168, wflign_patch.cpp:354-362,441-478) are here too: WFAlignerGapAffine, setHeuristicWFmash,
// setMaxAlignmentSteps, getAlignmentStatus, and the match-callback alignEnd2End, which has no device form and throws.
#pragma once

// wavefront_align.h status codes as the reference tests them (wflign.cpp:150,1168; wflign_patch.cpp:362,447)
#ifndef WF_STATUS_ALG_COMPLETED
#define WF_STATUS_ALG_COMPLETED 0
#define WF_STATUS_ALG_PARTIAL 1
#define WF_STATUS_MAX_STEPS_REACHED (-100)
#define WF_STATUS_OOM (-200)
#define WF_STATUS_UNATTAINABLE (-300)
#endif

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/wfmash_hip.h"

namespace wfa {

class WFAligner {
 public:
  enum AlignmentScope { Score, Alignment };
  enum MemoryModel { MemoryHigh, MemoryMed, MemoryLow, MemoryUltralow };
  enum AlignmentStatus { StatusAlgCompleted = 0, StatusAlgPartial = 1, StatusMaxStepsReached = -100, StatusOOM = -200 };

  virtual ~WFAligner() = default;
  void setHeuristicNone() {}  // the GPU path is always exact (wflign.cpp:145,289,377)
  // adaptive band of the dormant WFlambda aligner (wflign.cpp:1106-1110): the exact alignment is a valid answer
  void setHeuristicWFmash(int /*min_wavefront_length*/, int /*max_distance_threshold*/) {}
  // an alignment whose score passes `steps` ends with WF_STATUS_MAX_STEPS_REACHED (wflign_patch.cpp:354,441,474)
  void setMaxAlignmentSteps(int steps) { max_steps_ = steps; }
  // match-callback form of the dormant hierarchical WFlign (wflign.cpp:1163): the callback is host code per cell
  int alignEnd2End(int (*)(int, int, void*), void*, int, int) {
    throw std::logic_error("wfa::WFAligner: the match-callback alignEnd2End (dormant WFlign path) has no device form");
  }

  int alignEnd2End(const char* pattern, int plen, const char* text, int tlen) {
    return run(pattern, plen, text, tlen, mem_ == MemoryUltralow ? WFM_MODE_END2END_BIWFA : WFM_MODE_END2END_UNI, 0, 0, 0, 0);
  }
  int alignEnd2End(std::string& pattern, std::string& text) {
    return alignEnd2End(pattern.data(), (int)pattern.size(), text.data(), (int)text.size());
  }
  int alignEndsFree(const char* pattern, int plen, int pBeginFree, int pEndFree, const char* text, int tlen, int tBeginFree, int tEndFree) {
    return run(pattern, plen, text, tlen, WFM_MODE_ENDSFREE, pBeginFree, pEndFree, tBeginFree, tEndFree);
  }
  int alignEndsFree(std::string& pattern, int pBeginFree, int pEndFree, std::string& text, int tBeginFree, int tEndFree) {
    return alignEndsFree(pattern.data(), (int)pattern.size(), pBeginFree, pEndFree, text.data(), (int)text.size(), tBeginFree, tEndFree);
  }
  int getAlignmentStatus() const { return res_.status; }
  int getAlignmentScore() const { return -res_.score; }  // WFA2-lib reports penalties as negative scores
  // pointer into aligner-owned storage, the caller copies (wflign_alignment.cpp:671-677)
  void getAlignment(char** ops, int* len) { *ops = arena_.data(); *len = (int)res_.ops_len; }
  std::string getAlignment() { return std::string(arena_.data(), res_.ops_len); }  // long form (wflign.cpp:309)

 protected:
  WFAligner(int mismatch, int go1, int ge1, int go2, int ge2, MemoryModel m) : pen_{mismatch, go1, ge1, go2, ge2}, mem_(m) {}

 private:
  static wfm_handle_t* handle() {
    static thread_local struct H {
      wfm_handle_t* h = nullptr;
      H() { if (wfm_create(0, &h) != WFM_OK) throw std::runtime_error("wfa::WFAligner: no usable MI355X device (there is no CPU fallback)"); }
      ~H() { wfm_destroy(h); }
    } holder;
    return holder.h;
  }
  int run(const char* p, int pl, const char* t, int tl, int mode, int pbf, int pef, int tbf, int tef) {
    wfm_problem_t pr{p, pl, t, tl, mode, pbf, pef, tbf, tef};
    arena_.assign((size_t)pl + tl + 1, 0);
    res_ = wfm_result_t{};
    const int rc = wfm_align_batch(handle(), &pen_, &pr, 1, &res_, arena_.data(), arena_.size());
    if (rc < 0) res_.status = StatusOOM;
    else if (res_.status == 0 && max_steps_ >= 0 && res_.score > max_steps_) res_.status = StatusMaxStepsReached;
    return res_.status;  // 0 == WF_STATUS_ALG_COMPLETED (wflign.cpp:150,307,399)
  }
  wfm_penalties_t pen_;
  MemoryModel mem_;
  wfm_result_t res_{};
  std::vector<char> arena_;
  int max_steps_ = -1;
};

// gap-affine (one piece) = two identical pieces; only the dormant WFlign class builds it (wflign.cpp:1097-1125)
class WFAlignerGapAffine : public WFAligner {
 public:
  WFAlignerGapAffine(int mismatch, int gapOpening, int gapExtension, AlignmentScope, MemoryModel memoryModel)
      : WFAligner(mismatch, gapOpening, gapExtension, gapOpening, gapExtension, memoryModel) {}
};

class WFAlignerGapAffine2Pieces : public WFAligner {
 public:
  WFAlignerGapAffine2Pieces(int /*match, always 0 in wfmash*/, int mismatch, int gapOpening1, int gapExtension1,
                            int gapOpening2, int gapExtension2, AlignmentScope, MemoryModel memoryModel)
      : WFAligner(mismatch, gapOpening1, gapExtension1, gapOpening2, gapExtension2, memoryModel) {}
};

}  // namespace wfa
