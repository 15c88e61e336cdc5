// guard_alloc.h — development aid (ALEGO_DEBUG_CANARY=1): every persistent device allocation of the library gets 4 KB of 0xA5
// in front of it and behind it; alego_debug_check_guards() reads them back and reports the ones a kernel wrote into.  Out-of-bounds
// WRITES of up to a page are caught this way whatever lies next to the buffer; without the variable these are plain hipMalloc / hipFree.
#ifndef ALEGO_GUARD_ALLOC_H_
#define ALEGO_GUARD_ALLOC_H_
#include <hip/hip_runtime.h>

#include <string>

hipError_t guard_malloc(void** p, size_t bytes);
hipError_t guard_free(void* p);
// number of allocations whose guards are damaged (-1: guards are not enabled); `report` lists them (size, offset of the first bad byte)
int guard_check(std::string* report);
#endif
