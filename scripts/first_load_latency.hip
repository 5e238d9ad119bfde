// first_load_latency.hip -- what does the FIRST trip to memory of a kernel cost on an idle SIMD?  One wavefront per SIMD
// (1024 workgroups of 64), every lane loads one double from a 2 MB buffer that the previous launch of the same kernel has
// just read (so it sits in L2 / the Infinity Cache), then a dependent second load; shader-clock stamps around both.
//   hipcc --offload-arch=gfx950 -O3 scripts/first_load_latency.hip -o scripts/first_load_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
__global__ void probe(const double* __restrict__ a, const int* __restrict__ idx, long long* out, double* sink) {
  const long long t0 = clock64();
  const int lane = threadIdx.x, b = blockIdx.x;
  const double v = a[(size_t)b * 256 + lane];            // coalesced 512 B per wavefront
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  const int j = idx[(b * 64 + lane) & 0xFFFF];           // second, independent address: another page
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = clock64();
  const double w = a[(size_t)((j + (int)v) & 0x3FFFF)];  // dependent gather
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t3 = clock64();
  if (lane == 0) {
    out[b * 3 + 0] = t1 - t0;
    out[b * 3 + 1] = t2 - t1;
    out[b * 3 + 2] = t3 - t2;
  }
  if (w == 123.456) *sink = w;
}
int main() {
  const int B = 1024;
  double* a; int* idx; long long* out; double* sink;
  hipMalloc(&a, sizeof(double) * (1 << 18));
  hipMalloc(&idx, sizeof(int) * (1 << 16));
  hipMalloc(&out, sizeof(long long) * B * 3);
  hipMalloc(&sink, 8);
  hipMemset(a, 0, sizeof(double) * (1 << 18));
  std::vector<int> h(1 << 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (int)((i * 2654435761u) & 0x3FFFF);
  hipMemcpy(idx, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(probe, dim3(B), dim3(64), 0, 0, a, idx, out, sink);
  hipDeviceSynchronize();
  std::vector<long long> o(B * 3);
  hipMemcpy(o.data(), out, sizeof(long long) * B * 3, hipMemcpyDeviceToHost);
  for (int k = 0; k < 3; ++k) {
    std::vector<long long> v(B);
    for (int b = 0; b < B; ++b) v[b] = o[b * 3 + k];
    std::sort(v.begin(), v.end());
    printf("%s: median %lld  p10 %lld  p90 %lld shader cycles\n", k == 0 ? "first load of the kernel (coalesced)" : k == 1 ? "second load (another array)" : "third load (dependent gather)", v[B / 2], v[B / 10], v[9 * B / 10]);
  }
  return 0;
}
