// store-pattern microbenchmark: 65536 paths x 10 segments x 320 B of coefficients, written by 4096 wavefronts of 16 paths in
// ten "steps" (segment 9 first), with a dependent FMA chain of `work` instructions between steps standing in for the solve
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64, 1) void k(double* out, int S, int work, int n_paths) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x, pl = lane >> 2, dim = lane & 3;
  const int q = blockIdx.x * 16 + pl;
  double c[10];
  for (int k = 0; k < 10; ++k) c[k] = lane * 0.001 + k;
  double acc = 1.0 + lane * 1e-9;
  for (int v = S - 1; v >= 0; --v) {
    for (int w = 0; w < work; ++w) acc = fma(acc, 1.0000001, 1e-9);   // dependent chain: ~8 cycles each
    for (int k = 0; k < 10; ++k) c[k] += acc;
    if (MODE == 0) {        // as the kernel: a lane owns 80 B, 5 x 16 B
      double2* o = reinterpret_cast<double2*>(out + ((size_t)(q * S + v) * 4 + dim) * 10);
      for (int k = 0; k < 10; k += 2) o[k / 2] = make_double2(c[k], c[k + 1]);
    } else if (MODE == 1) { // perfectly coalesced garbage layout: instruction k writes 64 consecutive 16-B pieces
      double2* o = reinterpret_cast<double2*>(out + ((size_t)(blockIdx.x * S + v) * 640));
      for (int k = 0; k < 10; k += 2) o[(k / 2) * 64 + lane] = make_double2(c[k], c[k + 1]);
    } else if (MODE == 2) { // staged through LDS: 16 lanes x 16 B = 256 consecutive bytes of one path
      double2* st = reinterpret_cast<double2*>(lds + pl * 40 + dim * 10);
      for (int k = 0; k < 10; k += 2) st[k / 2] = make_double2(c[k], c[k + 1]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      for (int g = 0; g < 4; ++g) {
        const int P = 4 * g + (lane >> 4), piece = lane & 15;
        const double2 val = *reinterpret_cast<const double2*>(lds + P * 40 + piece * 2);
        *reinterpret_cast<double2*>(out + ((size_t)((blockIdx.x * 16 + P) * S + v) * 40 + piece * 2)) = val;
      }
      const double2 val = *reinterpret_cast<const double2*>(lds + pl * 40 + 32 + dim * 2);
      *reinterpret_cast<double2*>(out + ((size_t)(q * S + v) * 40 + 32 + dim * 2)) = val;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    } else if (MODE == 3) { // 8-byte stores, 10 per lane
      double* o = out + ((size_t)(q * S + v) * 4 + dim) * 10;
      for (int k = 0; k < 10; ++k) o[k] = c[k];
    } else if (MODE == 4) { // nothing stored
      for (int k = 0; k < 10; ++k) asm volatile("" ::"v"(c[k]));
    }
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int MODE>
float run(double* d, int P, int S, int work, int lds_bytes, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(P / 16), dim3(64), lds_bytes, 0, d, S, work, P);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(P / 16), dim3(64), lds_bytes, 0, d, S, work, P);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; if (ms < best) best = ms;
  }
  return sum / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int P = 65536, S = 10;
  double* d; CK(hipMalloc(&d, (size_t)P * S * 320));
  const int lds = 31232;  // the kernel's LDS per wavefront: five wavefronts per CU
  for (int work : {0, 100, 400, 1000}) {
    printf("work %4d FMA/step: lane-owned 5x16B %.1f us | coalesced %.1f | LDS-staged 256B/path %.1f | 10x8B %.1f | no stores %.1f  (211 MB written)\n", work,
           run<0>(d, P, S, work, lds, 10), run<1>(d, P, S, work, lds, 10), run<2>(d, P, S, work, lds, 10), run<3>(d, P, S, work, lds, 10), run<4>(d, P, S, work, lds, 10));
  }
  return 0;
}
