// fp64_latency.hip -- single-wave latency / issue interval of the FP64 instructions the solve kernels live on
// (dependent chain vs. independent streams), in shader cycles per instruction.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int NSTREAM, int OP>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double a, double b) {
  double acc[NSTREAM];
#pragma unroll
  for (int i = 0; i < NSTREAM; ++i) acc[i] = a + i + threadIdx.x;
  // per-lane values the compiler must keep in VGPRs
  double va = a + 1e-9 * threadIdx.x, vb = b - 1e-9 * threadIdx.x;
  asm volatile("" : "+v"(va), "+v"(vb));
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NSTREAM; ++i) {
        if (OP == 0) acc[i] = __builtin_fma(acc[i], b, a);
        else if (OP == 1) acc[i] = acc[i] * b;
        else if (OP == 2) acc[i] = acc[i] + b;
        else if (OP == 3) acc[i] = __builtin_amdgcn_rsq(acc[i]);
        else if (OP == 4) acc[i] = __builtin_amdgcn_rcp(acc[i]);
        else if (OP == 5) acc[i] = 1.0 / acc[i];
        else if (OP == 6) acc[i] = sqrt(acc[i]);
        else if (OP == 7) acc[i] = __builtin_fma(acc[i], vb, va);           // three VGPR-pair sources
        else if (OP == 8) acc[i] = acc[i] * vb;                             // two VGPR-pair sources
        else if (OP == 9) acc[i] = __builtin_fma(acc[i], vb, a);            // two VGPR pairs + one SGPR pair
        else if (OP == 10) acc[i] = __builtin_fma(-acc[(i + 1) % NSTREAM], vb, acc[i]);  // as in a Schur update: three VGPR pairs, all different
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NSTREAM; ++i) s += acc[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NSTREAM, int OP>
static int run(const char* name) {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NSTREAM, OP>), dim3(1), dim3(64), 0, 0, out, cyc, 1.000001, 0.999999);
  CK(hipDeviceSynchronize());
  long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-10s streams %d: %.2f cycles per instruction (%.2f per dependent step)\n", name, NSTREAM, (double)h / (64.0 * 16 * NSTREAM), (double)h / (64.0 * 16));
  return 0;
}

int main() {
  run<1, 0>("fma"); run<2, 0>("fma"); run<4, 0>("fma"); run<8, 0>("fma");
  run<1, 1>("mul"); run<4, 1>("mul");
  run<1, 2>("add"); run<4, 2>("add");
  run<1, 3>("rsq"); run<4, 3>("rsq");
  run<1, 4>("rcp"); run<4, 4>("rcp");
  run<1, 5>("div"); run<4, 5>("div");
  run<1, 6>("sqrt"); run<4, 6>("sqrt");
  run<1, 7>("fma vvv"); run<2, 7>("fma vvv"); run<4, 7>("fma vvv"); run<8, 7>("fma vvv");
  run<1, 8>("mul vv"); run<4, 8>("mul vv");
  run<1, 9>("fma vvs"); run<4, 9>("fma vvs");
  run<4, 10>("fma -v*v+v"); run<8, 10>("fma -v*v+v");
  return 0;
}
