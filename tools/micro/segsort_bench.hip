// micro-benchmark: rocprim::segmented_radix_sort_pairs cost vs segment count / sizes / total size
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
int main() {
  struct Case { int nseg; int seglen; int cap; } cases[] = {{320, 0, 250000}, {320, 2000, 250000}, {128, 0, 600000}, {128, 60000, 600000}, {16, 60000, 600000}, {2, 60000, 600000}, {2, 60000, 60000}, {128, 60000, 60000}};
  for (auto c : cases) {
    size_t total = (size_t)c.nseg * c.cap;
    unsigned *ka, *kb; int *va, *vb, *sb, *se;
    hipMalloc(&ka, total * 4); hipMalloc(&kb, total * 4); hipMalloc(&va, total * 4); hipMalloc(&vb, total * 4);
    hipMalloc(&sb, c.nseg * 4); hipMalloc(&se, c.nseg * 4);
    std::vector<unsigned> hk(total); for (size_t i = 0; i < total; ++i) hk[i] = (unsigned)((i * 2654435761u) >> 12);
    hipMemcpy(ka, hk.data(), total * 4, hipMemcpyHostToDevice);
    std::vector<int> b(c.nseg), e(c.nseg);
    for (int s = 0; s < c.nseg; ++s) { b[s] = s * c.cap; e[s] = b[s] + c.seglen; }
    hipMemcpy(sb, b.data(), c.nseg * 4, hipMemcpyHostToDevice); hipMemcpy(se, e.data(), c.nseg * 4, hipMemcpyHostToDevice);
    size_t bytes = 0; void* tmp = nullptr;
    rocprim::segmented_radix_sort_pairs(nullptr, bytes, ka, kb, va, vb, (unsigned)total, (unsigned)c.nseg, sb, se, 0, 32, 0);
    hipMalloc(&tmp, bytes);
    hipEvent_t a, z; hipEventCreate(&a); hipEventCreate(&z);
    for (int bits : {32, 20}) {
      for (int it = 0; it < 3; ++it) rocprim::segmented_radix_sort_pairs(tmp, bytes, ka, kb, va, vb, (unsigned)total, (unsigned)c.nseg, sb, se, 0, bits, 0);
      hipEventRecord(a, 0);
      for (int it = 0; it < 10; ++it) rocprim::segmented_radix_sort_pairs(tmp, bytes, ka, kb, va, vb, (unsigned)total, (unsigned)c.nseg, sb, se, 0, bits, 0);
      hipEventRecord(z, 0); hipEventSynchronize(z);
      float ms; hipEventElapsedTime(&ms, a, z);
      printf("nseg %4d seglen %6d cap %7d total %9zu bits %d tmp %zu KB: %.1f us/sort\n", c.nseg, c.seglen, c.cap, total, bits, bytes / 1024, ms * 100);
    }
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(sb); hipFree(se); hipFree(tmp);
  }
}
