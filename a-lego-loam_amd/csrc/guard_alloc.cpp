// guard_alloc.cpp — see guard_alloc.h
#include "guard_alloc.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {
constexpr size_t G = 4096;
struct Rec { size_t bytes; };
std::mutex mu;
std::map<void*, Rec> live;   // user pointer -> size
bool enabled() { static const bool on = getenv("ALEGO_DEBUG_CANARY") != nullptr; return on; }
}  // namespace

hipError_t guard_malloc(void** p, size_t bytes) {
  if (!enabled()) return hipMalloc(p, bytes);
  void* base = nullptr;
  const size_t padded = (bytes + 15) / 16 * 16;
  hipError_t e = hipMalloc(&base, padded + 2 * G);
  if (e != hipSuccess) return e;
  e = hipMemset(base, 0xA5, padded + 2 * G);
  if (e != hipSuccess) { (void)hipFree(base); return e; }
  *p = (char*)base + G;
  std::lock_guard<std::mutex> l(mu);
  live[*p] = Rec{bytes};
  return hipSuccess;
}

hipError_t guard_free(void* p) {
  if (!enabled() || !p) return hipFree(p);
  {
    std::lock_guard<std::mutex> l(mu);
    auto it = live.find(p);
    if (it == live.end()) return hipFree(p);
    live.erase(it);
  }
  return hipFree((char*)p - G);
}

int guard_check(std::string* report) {
  if (!enabled()) return -1;
  if (hipDeviceSynchronize() != hipSuccess) { if (report) *report += "device error before the check; "; }
  std::lock_guard<std::mutex> l(mu);
  int bad = 0;
  std::vector<unsigned char> buf(G);
  for (auto& kv : live) {
    const size_t bytes = kv.second.bytes;
    for (int side = 0; side < 2; ++side) {
      // behind: from the end of the user's bytes (the rounding slack is part of the guard) to the end of the page behind it
      const char* src = side == 0 ? (const char*)kv.first - G : (const char*)kv.first + bytes;
      const size_t n = side == 0 ? G : G;
      if (hipMemcpy(buf.data(), src, n, hipMemcpyDeviceToHost) != hipSuccess) { ++bad; continue; }
      for (size_t i = 0; i < n; ++i) {
        if (buf[i] != 0xA5) {
          ++bad;
          if (report) *report += "allocation of " + std::to_string(bytes) + " B: guard " + (side == 0 ? "in front" : "behind") + " damaged at offset " + std::to_string(side == 0 ? (long)i - (long)G : (long)i) + "; ";
          break;
        }
      }
    }
  }
  return bad;
}
