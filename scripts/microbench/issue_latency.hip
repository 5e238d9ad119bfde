// dependent-issue latencies of the f64 VALU operations the solve kernels chain, one wavefront per SIMD (scratch experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_lat(double* out, unsigned long long* clk, double a, double b, int n) {
  __shared__ double lds[256];
  lds[threadIdx.x] = a * threadIdx.x;
  __syncthreads();
  double x = out[threadIdx.x];
  unsigned long long t[12];
  unsigned long long w0 = wall_clock64();
  t[0] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) x = __builtin_fma(x, a, b);
  }
  asm volatile("" : "+v"(x));
  t[1] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) x = x * a;
  }
  asm volatile("" : "+v"(x));
  t[2] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) x = __builtin_amdgcn_rsq(x) + 2.0;
  }
  asm volatile("" : "+v"(x));
  t[3] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) x = __builtin_amdgcn_rcp(x);
  }
  asm volatile("" : "+v"(x));
  t[4] = __builtin_readcyclecounter();
  // two independent chains interleaved
  double y = x + 1.0;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      x = __builtin_fma(x, a, b);
      y = __builtin_fma(y, a, b);
    }
  }
  asm volatile("" : "+v"(x), "+v"(y));
  t[5] = __builtin_readcyclecounter();
  double z = x + 2.0, w = x + 3.0;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      x = __builtin_fma(x, a, b);
      y = __builtin_fma(y, a, b);
      z = __builtin_fma(z, a, b);
      w = __builtin_fma(w, a, b);
    }
  }
  asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
  t[6] = __builtin_readcyclecounter();
  // LDS pointer chase
  int idx = threadIdx.x;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) idx = (int)lds[idx & 255] & 255;
  }
  asm volatile("" : "+v"(idx));
  t[7] = __builtin_readcyclecounter();
  // ds_bpermute chain
  int v = idx;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) v = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 4) << 2, v) + 1;
  }
  asm volatile("" : "+v"(v));
  t[8] = __builtin_readcyclecounter();
  // f32 fma chain for comparison
  float f = (float)x;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) f = __builtin_fmaf(f, (float)a, (float)b);
  }
  asm volatile("" : "+v"(f));
  t[9] = __builtin_readcyclecounter();
  unsigned long long w1 = wall_clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x + y + z + w + idx + v + f;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    for (int k = 0; k < 10; ++k) clk[k] = t[k];
    clk[10] = w1 - w0;
  }
}

int main() {
  double* out;
  unsigned long long* clk;
  const int blocks = 2048;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMemset(out, 0, sizeof(double) * 64 * blocks);
  hipMalloc(&clk, sizeof(unsigned long long) * 16);
  const int n = 64;
  for (int grid : {1, 1024, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_lat, dim3(grid), dim3(64), 0, 0, out, clk, 0.999, 1e-3, n);
      hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"fma f64 chain", "mul f64 chain", "rsq f64 + add chain", "rcp f64 chain", "2 fma chains", "4 fma chains",
                           "LDS pointer chase (read + cvt + and)", "ds_bpermute + add chain", "fma f32 chain"};
    const double ops[] = {1, 1, 2, 1, 2, 4, 1, 1, 1};
    printf("grid %d wavefronts: s_memtime clocks %llu, wall_clock64 ticks %llu (100 MHz: %.2f us) -> s_memtime at %.1f MHz\n", grid,
           h[9] - h[0], h[10], h[10] / 100.0, (double)(h[9] - h[0]) / (h[10] / 100.0));
    for (int k = 0; k < 9; ++k)
      printf("  %-40s %7.2f clocks per chain step (%.2f per instruction)\n", names[k], (double)(h[k + 1] - h[k]) / (n * 32.0),
             (double)(h[k + 1] - h[k]) / (n * 32.0) / ops[k]);
  }
  return 0;
}
