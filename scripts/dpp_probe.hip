// dpp_probe.hip -- gfx950 probes behind the row-broadcast elimination of the tile kernel (phase B) and behind the FP64
// peaks quoted in bench.py's rooflines.  Not part of the library.
//   1. semantics of the double-precision DPP forms the assembler accepts for gfx950: v_fmac_f64_dpp / v_mov_b64_dpp /
//      v_rcp_f64_dpp with row_newbcast:N (source operand 0 is read from lane N of the lane's own row of 16);
//   2. single-wavefront issue interval of v_fmac_f64_dpp, dependent and independent, next to plain v_fma_f64;
//   3. whole-GPU FP64 peak of the vector pipe (v_fma_f64) and of the matrix pipe (v_mfma_f64_16x16x4_f64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define FMAC_BCAST(acc, a, b, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b))
#define FMAC_NEG_BCAST(acc, a, b, N) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b))

__global__ void semantics(const double* a, const double* b, double* out) {
  const int i = threadIdx.x;
  double x = a[i], y = b[i];
  double acc = 100.0 + i;
  FMAC_BCAST(acc, x, y, 3);          // acc + a[row*16 + 3] * b[i]
  out[i] = acc;
  double acc2 = 0.5 * i;
  FMAC_NEG_BCAST(acc2, x, y, 11);    // acc2 - a[row*16 + 11] * b[i]
  out[64 + i] = acc2;
  double m;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(x));
  out[128 + i] = m;
  double r;
  asm volatile("v_rcp_f64_dpp %0, %1 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  out[192 + i] = r;
  // half of the lanes switched off: does a broadcast from a disabled lane still deliver its register?
  double acc3 = 7.0;
  if ((i & 15) >= 8) FMAC_BCAST(acc3, x, y, 2);   // lane 2 of every row is inactive here
  out[256 + i] = acc3;
}

template <int NSTREAM, int OP>
__global__ __launch_bounds__(64) void issue(double* out, long long* cyc, double a, double b) {
  double acc[NSTREAM];
#pragma unroll
  for (int i = 0; i < NSTREAM; ++i) acc[i] = a + i + threadIdx.x;
  double va = a + 1e-9 * threadIdx.x, vb = 1e-3 * b - 1e-9 * threadIdx.x;
  asm volatile("" : "+v"(va), "+v"(vb));
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NSTREAM; ++i) {
        if (OP == 0) acc[i] = __builtin_fma(acc[i], vb, va);
        else if (OP == 1) FMAC_BCAST(acc[i], va, vb, 5);                       // broadcast operand is loop invariant
        else if (OP == 2) FMAC_NEG_BCAST(acc[i], acc[(i + 1) % NSTREAM], vb, 5);  // broadcast operand produced by the neighbouring stream
        else if (OP == 3) FMAC_NEG_BCAST(acc[i], acc[i], vb, 5);               // acc -= bcast(acc) * m : the elimination update
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
static int run_issue(const char* name) {
  double* out; long long* cyc;
  CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 8));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((issue<NSTREAM, OP>), dim3(1), dim3(64), 0, 0, out, cyc, 1.000001, 0.999999);
  CK(hipDeviceSynchronize());
  long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-28s streams %d: %.2f cycles per instruction\n", name, NSTREAM, (double)h / (64.0 * 16 * NSTREAM));
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

// whole-GPU peaks
__global__ __launch_bounds__(256) void peak_fma(double* out, int iters, double a, double b) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = a + i + threadIdx.x;
  double vb = b - 1e-12 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], vb, a);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void peak_mfma(double* out, int iters, double a, double b) {
  double4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = double4_t{a, a, a, a};
  const double va = a + 1e-9 * threadIdx.x, vb = b - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  // ---- 1. semantics
  std::vector<double> ha(64), hb(64), ho(320);
  for (int i = 0; i < 64; ++i) { ha[i] = 1.0 + 0.25 * i; hb[i] = 2.0 + 0.125 * i; }
  double *a, *b, *o;
  CK(hipMalloc(&a, 512)); CK(hipMalloc(&b, 512)); CK(hipMalloc(&o, 320 * 8));
  CK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, a, b, o);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ho.data(), o, 320 * 8, hipMemcpyDeviceToHost));
  int bad[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 64; ++i) {
    const int row = i / 16;
    if (ho[i] != std::fma(ha[row * 16 + 3], hb[i], 100.0 + i)) bad[0]++;
    if (ho[64 + i] != std::fma(-ha[row * 16 + 11], hb[i], 0.5 * i)) bad[1]++;
    if (ho[128 + i] != ha[row * 16 + 7]) bad[2]++;
    if (std::fabs(ho[192 + i] * ha[row * 16 + 9] - 1.0) > 1e-6) bad[3]++;
    const double want = ((i & 15) >= 8) ? std::fma(ha[row * 16 + 2], hb[i], 7.0) : 7.0;
    if (ho[256 + i] != want) bad[4]++;
  }
  printf("semantics: fmac_dpp %s, fmac_dpp(neg) %s, mov_b64_dpp %s, rcp_dpp %s, broadcast from an exec-disabled lane %s (lane 8 of row 0 got %.6f, fma would give %.6f)\n",
         bad[0] ? "WRONG" : "ok", bad[1] ? "WRONG" : "ok", bad[2] ? "WRONG" : "ok", bad[3] ? "WRONG" : "ok",
         bad[4] ? "DIFFERS" : "delivers the register", ho[256 + 8], std::fma(ha[2], hb[8], 7.0));

  // ---- 2. single-wavefront issue intervals
  run_issue<1, 0>("fma vvv dependent"); run_issue<4, 0>("fma vvv"); run_issue<8, 0>("fma vvv");
  run_issue<1, 1>("fmac_dpp dependent acc"); run_issue<4, 1>("fmac_dpp"); run_issue<8, 1>("fmac_dpp");
  run_issue<2, 2>("fmac_dpp bcast(neighbour)"); run_issue<8, 2>("fmac_dpp bcast(neighbour)");
  run_issue<1, 3>("fmac_dpp acc-=bcast(acc)*m"); run_issue<8, 3>("fmac_dpp acc-=bcast(acc)*m");

  // ---- 3. whole-GPU FP64 peaks
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 8, iters = 20000;
  double* big;
  CK(hipMalloc(&big, (size_t)blocks * 256 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(peak_fma, dim3(blocks), dim3(256), 0, 0, big, iters, 1.000001, 0.999999);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double fl_fma = 2.0 * 16 * (double)iters * blocks * 256;
  printf("v_fma_f64 peak: %.1f TFLOP/s (%d CUs, clock %d MHz, %.2f ms)\n", fl_fma / (ms * 1e-3) / 1e12, prop.multiProcessorCount,
         prop.clockRate / 1000, ms);
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(peak_mfma, dim3(blocks), dim3(256), 0, 0, big, iters, 1.000001, 0.999999);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  // one v_mfma_f64_16x16x4_f64 = 16*16*4 multiply-adds per wavefront
  const double fl_mfma = 2.0 * 16 * 16 * 4 * 4 * (double)iters * blocks * 4;
  printf("v_mfma_f64_16x16x4_f64 peak: %.1f TFLOP/s (%.2f ms)\n", fl_mfma / (ms * 1e-3) / 1e12, ms);
  return 0;
}
