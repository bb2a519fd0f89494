// wfa_handle.h -- accessors of the opaque wfm_handle for the other translation units.
#ifndef WFM_WFA_HANDLE_H_
#define WFM_WFA_HANDLE_H_
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/wfmash_hip.h"

hipStream_t wfm_stream(wfm_handle_t* h);
int wfm_device(const wfm_handle_t* h);
void wfm_set_error(wfm_handle_t* h, const std::string& msg);
// one object another translation unit keeps with the handle; destroyed with it
void* wfm_attachment(wfm_handle_t* h);
void wfm_set_attachment(wfm_handle_t* h, void* p, void (*destroy)(void*));
#endif
