// kernarg_latency.hip -- how long does a wavefront wait for its kernel arguments?  One wavefront per SIMD (1024 workgroups of
// 64): a shader-clock stamp at entry, one after the first use of a pointer argument (the compiler's s_load of the kernarg
// segment has returned), one after a coalesced load through it.  Built twice: as is, and with the arguments preloaded into
// SGPRs by the dispatcher (-mllvm -amdgpu-kernarg-preload-count=N), and timed back to back on the host as well.
//   hipcc --offload-arch=gfx950 -O3 scripts/kernarg_latency.hip -o scripts/kernarg_latency.bin
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 scripts/kernarg_latency.hip -o scripts/kernarg_latency_preload.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void probe(const double* __restrict__ a, long long* __restrict__ out, double* __restrict__ sink, int stride) {
  const long long t0 = clock64();
  asm volatile("" ::"s"(a), "s"(out), "s"(stride));  // the arguments are needed HERE
  const long long t1 = clock64();
  const int lane = threadIdx.x, b = blockIdx.x;
  const double v = a[(size_t)b * stride + lane];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = clock64();
  if (lane == 0) {
    out[b * 2 + 0] = t1 - t0;
    out[b * 2 + 1] = t2 - t1;
  }
  if (v == 123.456) *sink = v;
}
int main() {
  const int B = 1024;
  double* a; long long* out; double* sink;
  hipMalloc(&a, sizeof(double) * (1 << 18));
  hipMalloc(&out, sizeof(long long) * B * 2);
  hipMalloc(&sink, 8);
  hipMemset(a, 0, sizeof(double) * (1 << 18));
  for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe, dim3(B), dim3(64), 0, 0, a, out, sink, 256);
  hipDeviceSynchronize();
  std::vector<long long> o(B * 2);
  hipMemcpy(o.data(), out, sizeof(long long) * B * 2, hipMemcpyDeviceToHost);
  for (int k = 0; k < 2; ++k) {
    std::vector<long long> v(B);
    for (int b = 0; b < B; ++b) v[b] = o[b * 2 + k];
    std::sort(v.begin(), v.end());
    printf("%s: median %lld  p10 %lld  p90 %lld shader cycles\n", k == 0 ? "kernel arguments in SGPRs after" : "first load after that (coalesced)",
           v[B / 2], v[B / 10], v[9 * B / 10]);
  }
  const int N = 2000;
  const auto w0 = std::chrono::steady_clock::now();
  for (int rep = 0; rep < N; ++rep) hipLaunchKernelGGL(probe, dim3(B), dim3(64), 0, 0, a, out, sink, 256);
  hipDeviceSynchronize();
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count() / N;
  printf("back to back: %.2f us per launch\n", us);
  return 0;
}
