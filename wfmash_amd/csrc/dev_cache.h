// dev_cache.h -- device blocks of the map path (and, since round 4, the arenas and sequence buffers of the align path) come from, and go
// back to, a per-device cache instead of hipMalloc / hipFree.  Round 6: blocks of a megabyte and more come from one heap per device on mapped
// chunks of a reserved address range (dev_cache.hip), the size-class cache described below is what the small blocks keep.
//
// Why: on this driver a hipMalloc of memory the process has not had mapped before costs 30 - 40 ms per GB and a hipFree of
// gigabytes makes the next allocation wait for the scrubbing (profiles/r3_cold_start.md); the map path works on whole
// chromosomes -- 0.25 GB of bases, 2 GB of hashes, 2 GB of sort space per sequence -- and asked for them sequence after
// sequence: 2 of 8 sketches of a C4 rank's identity estimate took 360 - 390 ms instead of 20, the index stage's emit 10 or
// 220 ms, by chance.  With 288 GB of HBM the blocks can simply stay: sizes are rounded up to 1/8 of a power of two (so that
// chromosomes of slightly different lengths share a class), a freed block waits in its class, and only an allocation
// failure (or wfm_trim_device_cache) gives memory back to the driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

hipError_t wfm_dmalloc(void** p, size_t bytes);  // on the current device
// The block goes back to the cache.  hipFree waits for the device; wfm_dfree does the same (hipDeviceSynchronize) so that it
// can replace it anywhere; callers that have synchronised the stream the block was used on call wfm_dfree_nosync.
void wfm_dfree(void* p);
void wfm_dfree_nosync(void* p);
size_t wfm_dcache_trim(void);                    // every cached small block and the free end of every device's heap back to the driver; returns the bytes released
void wfm_dcache_warm(void);                      // makes the current device's heap (and maps its first WFM_POOL_GB): wfm_create calls it
size_t wfm_dcache_stats(int dev, size_t* committed, size_t* live, size_t* peak_live);  // the heap of a device: bytes mapped / handed out / most ever handed out at once
