// Diagnostic: random / mutated PointCloud2 headers and payloads through alego_pc2_to_points under AddressSanitizer + UBSan.
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -Iinclude tests/diagnostics/pc2_fuzz.cpp a-lego-loam_amd/csrc/pc2.cpp -o /tmp/pc2_fuzz && /tmp/pc2_fuzz 2000000
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "alego_mi355x.h"

int main(int argc, char** argv) {
  const long iters = argc > 1 ? std::atol(argv[1]) : 200000;
  std::mt19937_64 rng(12345);
  auto R = [&](long lo, long hi) { return (long)(lo + rng() % (unsigned long)(hi - lo + 1)); };
  long ok = 0, bad = 0;
  const char* names[] = {"x", "y", "z", "intensity", "ring", "t", ""};
  for (long it = 0; it < iters; ++it) {
    uint32_t w = (uint32_t)R(0, 40), h = (uint32_t)R(0, 3), ps = (uint32_t)(R(0, 3) ? (R(0, 1) ? 16 : 32) : R(0, 64));
    uint32_t rs = R(0, 3) ? w * ps + (uint32_t)(R(0, 3) ? 0 : 8) : (uint32_t)R(0, 3000);
    alego_pc2_field f[8];
    int nf = (int)R(0, 6);
    for (int i = 0; i < nf; ++i) {
      f[i].name = R(0, 9) ? names[i < 4 ? i : R(0, 6)] : nullptr;
      const uint32_t offs[] = {0, 4, 8, 12, 16, 20, ps, ps - 1, ps - 3, 0x7fffffffu, 0xfffffffdu, (uint32_t)R(0, 80)};
      f[i].offset = offs[R(0, 11)];
      f[i].datatype = (uint8_t)(R(0, 4) ? 7 : R(0, 12));
      f[i].count = (uint32_t)R(0, 2);
    }
    uint64_t len = R(0, 3) ? (uint64_t)h * rs : (uint64_t)R(0, 4000);
    if (R(0, 5) == 0 && len > 0) len -= (uint64_t)R(1, (long)std::min<uint64_t>(len, 40));
    std::vector<uint8_t> data(len);
    for (auto& b : data) b = (uint8_t)rng();
    if (R(0, 20) == 0) { w = 0xffffffffu; }
    if (R(0, 20) == 0) { h = 0xffffffffu; }
    const long n = (w > 100000u || h > 100000u) ? 5000 : (long)w * (long)h;
    const int cap = (int)(R(0, 3) ? std::min<long>(n < 0 ? 0 : n, 5000) : R(0, 50));
    std::vector<alego_point> out((size_t)std::max(cap, 1));
    const int rc = alego_pc2_to_points(R(0, 30) ? data.data() : nullptr, len, w, h, ps, rs, (int)R(0, 1), R(0, 30) ? f : nullptr, nf, R(0, 30) ? out.data() : nullptr, cap);
    if (rc >= 0) { ++ok; if (rc > cap) { std::printf("rc %d > cap %d\n", rc, cap); return 1; } } else ++bad;
  }
  std::printf("accepted %ld rejected %ld\n", ok, bad);
  return 0;
}
