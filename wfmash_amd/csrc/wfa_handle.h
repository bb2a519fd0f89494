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
// called after the LAST handle of the process has been destroyed (the host layer lets go of the sequence stores it keeps between calls)
void wfm_set_last_handle_hook(void (*f)());
#endif
