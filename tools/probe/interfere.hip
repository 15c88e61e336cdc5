// interfere.hip — resource-interference probe (development tool, not part of the product).
// Runs one kind of synthetic load next to bench.py (a second process on the same GPU) for a given time:
//   valu  N : N wavefronts per SIMD of dependent-free f32 FMAs (no memory, no LDS)
//   lds   N : N wavefronts per SIMD of ds_read / ds_write traffic
//   mem   G : streaming copy through HBM, G gigabytes per launch
// The drop of bench.py's throughput per kind says which shared resource the pipeline is sensitive to.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void __launch_bounds__(64) k_valu(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 2.f, g = 3.f, h = 4.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { a = a * b + c; d = d * b + e; f = f * b + g; h = h * b + a; c = c * b + d; e = e * b + f; g = g * b + h; b = b * 0.99999f + 1e-6f; }
  }
  if (a + c + d + e + f + g + h == 123.f) out[0] = a;
}
__global__ void __launch_bounds__(64) k_lds(float* out, int iters) {
  __shared__ float s[64 * 17];
  for (int q = threadIdx.x; q < 64 * 17; q += 64) s[q] = q;
  float acc = 0;
  int idx = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc += s[idx]; s[idx] = acc; idx = (idx + 67) % (64 * 17); }
  }
  if (acc == 123.f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_mem(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: interfere valu|lds|mem <N> <seconds>\n"); return 2; }
  const char* kind = argv[1]; const double N = atof(argv[2]), secs = atof(argv[3]);
  float* out; hipMalloc(&out, 4);
  float4 *a = nullptr, *b = nullptr; size_t n = 0;
  if (!strcmp(kind, "mem")) { n = (size_t)(N * 1e9 / 32); hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 0, n * 16); }
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    if (!strcmp(kind, "valu")) hipLaunchKernelGGL(k_valu, dim3((int)(1024 * N)), dim3(64), 0, 0, out, 20000);
    else if (!strcmp(kind, "lds")) hipLaunchKernelGGL(k_lds, dim3((int)(1024 * N)), dim3(64), 0, 0, out, 4000);
    else hipLaunchKernelGGL(k_mem, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipDeviceSynchronize();
    ++launches;
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!strcmp(kind, "mem")) std::printf("mem: %.0f GB/s\n", launches * n * 32.0 / dt / 1e9);
  else std::printf("%s: %ld launches in %.1f s (%.2f ms each)\n", kind, launches, dt, 1e3 * dt / launches);
  return 0;
}
