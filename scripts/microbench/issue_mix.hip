// what scalar instructions between the f64 VALU instructions of a lone wavefront cost (scratch experiment)
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

__global__ void k_mix(double* out, unsigned long long* clk, double a, double b, int n) {
  double x = out[threadIdx.x];
  unsigned long long t[8];
  t[0] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));)
  }
  t[1] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s20, 0x3ff12345\n s_mov_b32 s21, 0x12345678" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21");)
  }
  t[2] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_readlane_b32 s20, %3, 3\n v_readlane_b32 s21, %3, 4" : "+v"(x) : "v"(a), "v"(b), "v"(threadIdx.x) : "s20", "s21");)
  }
  t[3] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s20, 0x3ff12345" : "+v"(x) : "v"(a), "v"(b) : "s20");)
  }
  t[4] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_cndmask_b32 %3, %3, %3, vcc" : "+v"(x) : "v"(a), "v"(b), "v"(threadIdx.x));)
  }
  t[5] = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    REP32(asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 0" : "+v"(x) : "v"(a), "v"(b));)
  }
  t[6] = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0)
    for (int k = 0; k < 7; ++k) clk[k] = t[k];
}

int main() {
  double* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, sizeof(double) * 64 * 2048);
  (void)hipMemset(out, 0, sizeof(double) * 64 * 2048);
  (void)hipMalloc(&clk, sizeof(unsigned long long) * 16);
  const int n = 64;
  for (int grid : {1024, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_mix, dim3(grid), dim3(64), 0, 0, out, clk, 0.999, 1e-3, n);
      (void)hipDeviceSynchronize();
    }
    unsigned long long h[16];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"fma f64", "fma f64 + 2 s_mov_b32 (literal)", "fma f64 + 2 v_readlane", "fma f64 + 1 s_mov_b32", "fma f64 + v_cndmask_b32", "fma f64 + s_nop 0"};
    printf("grid %d wavefronts\n", grid);
    for (int k = 0; k < 6; ++k) printf("  %-36s %7.2f clocks per group\n", names[k], (double)(h[k + 1] - h[k]) / (n * 32.0));
  }
  return 0;
}
