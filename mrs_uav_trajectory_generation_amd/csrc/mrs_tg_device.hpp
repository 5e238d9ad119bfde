// mrs_tg_device.hpp -- device-side numerics of the batched minimum-derivative QP (gfx950).
//
// What the reference computes per path (all citations relative to /root/reference/include/
// eth_trajectory_generation/impl/): per-segment A^-1 and H = A^-T Q A^-1
// (polynomial_optimization_linear_impl.h:113-177,311-320,606-618), the reduced system
// d_p = -R_pp^-1 R_pf d_f (:341-373) and the coefficients c = A^-1 C d (:264-282).
//
// How it is computed here (DESIGN.md "numerical formulation"): time-normalised exact constants
// (mrs_tg_constants.h) so no matrix is inverted at run time, and R_pp is never formed: it is block
// tridiagonal over vertices (one 4x4 block of the free derivatives v,a,j,s per vertex; the position
// slot is always constrained by every caller of the reference, src/mrs_trajectory_generation.cpp:944,
// 963,967), so one sweep of block Cholesky over the vertices eliminates it.  Constrained slots are
// masked to identity rows, which keeps every path on the same instruction stream.
#pragma once
#include <hip/hip_runtime.h>

#include "mrs_tg_constants.h"
#include "mrs_tg_launch.h"

namespace mrs_tg {

constexpr int kN = 10;     // coefficients per polynomial
constexpr int kHalf = 5;   // derivative slots per vertex
constexpr int kD = 4;      // dimensions
constexpr int kNB = 4;     // free-candidate slots per vertex (velocity..snap)
constexpr int kSlot0 = 1;  // first free-candidate slot
constexpr double kTimeLowerBound = 0.01;  // kOptimizationTimeLowerBound (polynomial_optimization_nonlinear.h:304)

// optional phase clocks: the micro-benchmarks under scripts/ define MRS_TG_PHASE_CLOCKS and read g_phase_clock;
// compiled out of the library
#ifdef MRS_TG_PHASE_CLOCKS
__device__ long long g_phase_clock[32];
#define MRS_TG_PHASE_MARK_T(i, thread)                                                          \
  do {                                                                                          \
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == (thread)) g_phase_clock[i] = clock64();   \
  } while (0)
#else
#define MRS_TG_PHASE_MARK_T(i, thread)
#endif
#define MRS_TG_PHASE_MARK(i) MRS_TG_PHASE_MARK_T(i, 0)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the global-memory counter
// (s_waitcnt vmcnt(0)), which turns a prefetch issued before the barrier into a wait at the barrier.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// blockIdx -> work index such that the workgroups of one XCD (blockIdx % 8: round-robin dispatch over the eight XCDs of
// an MI355X) own a contiguous range of work items and therefore share cache lines in that XCD's L2
__device__ __forceinline__ int xcd_contiguous_index(unsigned bid, unsigned grid) {
  constexpr unsigned kXcds = 8;
  const unsigned xcd = bid % kXcds, idx = bid / kXcds, per = grid / kXcds, rem = grid % kXcds;
  return (int)(xcd * per + (xcd < rem ? xcd : rem) + idx);
}

// LDS accumulation (ds_add_f64); the callers add exactly two terms per slot, so the result does not depend on the order
__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

static __constant__ double c_abar_inv[kN][kN] = MRS_TG_ABAR_INV_INIT;
static __constant__ double c_hbar[kHalf][kN][kN] = MRS_TG_HBAR_INIT;

__device__ __forceinline__ double rsqrt_refined(double x) {
  // v_rsq_f64 is good to 5e-8; one third-order step y (1 + e/2 + 3 e^2/8), e = 1 - x y^2, leaves O(e^3) ~ 1e-21 and
  // is four dependent operations (two Newton steps: six) -- scripts/rsq_accuracy.hip
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  const double r = fma(y * e, fma(e, 0.375, 0.5), y);
  // Every caller is a Cholesky pivot.  R_pp is positive definite, but with segment times spanning eleven decades (a
  // path whose feasibility scaling ran away: 2.7 s next to 9e11 s, T^-7 apart by 1e80) a Schur complement can come
  // out <= 0; 1/sqrt of that would turn the whole path into NaN.  A zero here makes L_cc = 0, z_c = 0, W_c. = 0 and
  // x_c = 0: the variable is left at zero, which is what the reference's rank-revealing QR does with a vanishing
  // pivot, and every output stays finite.
  return (x > 0.0) ? r : 0.0;
}

// index into a packed lower-triangular 4x4 (r >= c)
__device__ __forceinline__ constexpr int tri(int r, int c) { return r * (r + 1) / 2 + c; }
// index into the packed upper triangle of the symmetric 10x10 (a <= b)
__device__ __forceinline__ constexpr int sym10(int a, int b) {
  return (a <= b) ? (a * kN - a * (a - 1) / 2 + (b - a)) : (b * kN - b * (b - 1) / 2 + (a - b));
}
constexpr int kSym10 = 55;

// p2[m] = T^(m + 1 - 2d), m = 0..8: H(a,b) = HBAR[a][b] * p2[(a%5)+(b%5)]
__device__ __forceinline__ void hessian_powers(double T, int d, double (&p2)[9]) {
  const double t2 = T * T;
  double td = 1.0;
  if (d == 1) td = T;
  else if (d == 2) td = t2;
  else if (d == 3) td = t2 * T;
  else if (d == 4) td = t2 * t2;
  p2[0] = T / (td * td);
#pragma unroll
  for (int m = 1; m < 9; ++m) p2[m] = p2[m - 1] * T;
}

// Hs[sym10(a,b)] for one segment, from the exact unit-time constants
__device__ __forceinline__ void hessian_from_time(double T, int d, double (&Hs)[kSym10]) {
  double p2[9];
  hessian_powers(T, d, p2);
#pragma unroll
  for (int a = 0; a < kN; ++a)
#pragma unroll
    for (int b = a; b < kN; ++b) Hs[sym10(a, b)] = c_hbar[d][a][b] * p2[(a % kHalf) + (b % kHalf)];
}

// ---------------------------------------------------------------------------------------------
// Forward elimination over the vertices of one path for ND of the 4 dimensions.
//
// State while standing on vertex v (before its segment v is absorbed):
//   Sm  : Schur complement accumulated so far for vertex v's 4x4 block (packed lower)
//   y   : right-hand side accumulated so far, per dimension
//   qf  : sum over segments of f^T H f (f = constrained part of the end-point derivatives)
//   red : sum over vertices of |L^-1 y|^2 ; the optimal cost is 0.5 * (qf - red)
template <int ND>
struct Elim {
  double Sm[10];
  double y[kNB][ND];
  double Linv[kNB];  // reciprocals of the diagonal of the most recent Cholesky factor
  double qf, red;

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int i = 0; i < 10; ++i) Sm[i] = 0.0;
#pragma unroll
    for (int k = 0; k < kNB; ++k)
#pragma unroll
      for (int q = 0; q < ND; ++q) y[k][q] = 0.0;
    qf = 0.0;
    red = 0.0;
  }

  // mask the current vertex (bit k of free_mask set = slot kSlot0+k is free), factor its block,
  // z = L^-1 y.  L (packed lower) and z are returned for the caller to keep if it back-substitutes.
  __device__ __forceinline__ void factor_vertex(unsigned free_mask, double (&L)[10], double (&z)[kNB][ND]) {
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
      const bool fr = (free_mask >> r) & 1u;
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        const bool fc = (free_mask >> c) & 1u;
        double v = Sm[tri(r, c)];
        if (r == c) v = fr ? v : 1.0;
        else v = (fr && fc) ? v : 0.0;
        Sm[tri(r, c)] = v;
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) y[r][q] = fr ? y[r][q] : 0.0;
    }
    // Cholesky, column by column
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
      double dsum = Sm[tri(c, c)];
#pragma unroll
      for (int m = 0; m < c; ++m) dsum -= L[tri(c, m)] * L[tri(c, m)];
      // 1/sqrt(pivot): hardware estimate + one third-order step, then sqrt(pivot) = pivot * rsqrt(pivot); ~4x fewer
      // instructions than IEEE sqrt followed by IEEE division.  (Returning the reciprocal diagonal so that the back
      // substitution multiplies instead of dividing was measured: 2 % faster on the fused lane kernels, 12 % slower
      // on the 512-VGPR blocks variant, whose register allocation it upsets -- not kept.)
      const double inv = rsqrt_refined(dsum);
      // a rejected pivot (inv = 0, see rsqrt_refined) leaves z_c = 0, W_c. = 0 and L_.c = 0, so the back substitution
      // arrives at 0 / L_cc for it: a tiny positive stand-in keeps that 0 instead of 0 / 0
      L[tri(c, c)] = fmax(dsum * inv, 1.0e-300);
      Linv[c] = inv;
#pragma unroll
      for (int r = c + 1; r < kNB; ++r) {
        double s = Sm[tri(r, c)];
#pragma unroll
        for (int m = 0; m < c; ++m) s -= L[tri(r, m)] * L[tri(c, m)];
        L[tri(r, c)] = s * inv;
      }
    }
    // z = L^-1 y
#pragma unroll
    for (int q = 0; q < ND; ++q) {
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        double s = y[r][q];
#pragma unroll
        for (int m = 0; m < r; ++m) s -= L[tri(r, m)] * z[m][q];
        z[r][q] = s * Linv[r];
        red += z[r][q] * z[r][q];
      }
    }
  }

  // Absorb segment v (between the current vertex v and vertex v+1):
  //   Hs        packed symmetric 10x10 Hessian of the segment
  //   fs, fe    constrained derivative values (0 where free) of vertex v / v+1, [slot][dim]
  //   free_s/e  free masks of vertex v / v+1
  // On return the state stands on vertex v+1.  L, z, W describe vertex v for back-substitution:
  //   x_v = L^-T (z - W x_{v+1}).
  __device__ __forceinline__ void absorb_segment(const double (&Hs)[kSym10], const double (&fs)[kHalf][ND],
                                                 const double (&fe)[kHalf][ND], unsigned free_s, unsigned free_e,
                                                 double (&L)[10], double (&z)[kNB][ND], double (&W)[kNB][kNB]) {
    // u = H [fs; fe]; the free rows feed the right-hand sides, all rows feed qf.
    // Every vertex constrains its position, and the other constrained derivatives are zero except for an
    // initial state, so the two position columns are always processed and the eight derivative columns
    // only where some constrained derivative value is non-zero (one branch per segment, uniform across the
    // wavefront in practice): 20 instead of 100 multiply-adds per dimension.
    bool deriv_nonzero = false;
#pragma unroll
    for (int q = 0; q < ND; ++q)
#pragma unroll
      for (int b = 1; b < kHalf; ++b) deriv_nonzero = deriv_nonzero || (fs[b][q] != 0.0) || (fe[b][q] != 0.0);
    double u[kN][ND];
#pragma unroll
    for (int q = 0; q < ND; ++q) {
#pragma unroll
      for (int a = 0; a < kN; ++a) u[a][q] = Hs[sym10(a, 0)] * fs[0][q] + Hs[sym10(a, kHalf)] * fe[0][q];
      qf += fs[0][q] * u[0][q] + fe[0][q] * u[kHalf][q];
    }
    if (deriv_nonzero) {
#pragma unroll
      for (int q = 0; q < ND; ++q) {
        double extra = 0.0;
#pragma unroll
        for (int a = 0; a < kN; ++a) {
          double s = 0.0;
#pragma unroll
          for (int b = 1; b < kHalf; ++b) s += Hs[sym10(a, b)] * fs[b][q] + Hs[sym10(a, kHalf + b)] * fe[b][q];
          u[a][q] += s;
        }
        // qf = f^T u over all ten rows: the position rows were added above with the incomplete u, fix them up
#pragma unroll
        for (int b = 1; b < kHalf; ++b) extra += fs[b][q] * u[b][q] + fe[b][q] * u[kHalf + b][q];
        double pos_fix = 0.0;
#pragma unroll
        for (int b = 1; b < kHalf; ++b)
          pos_fix += fs[0][q] * (Hs[sym10(0, b)] * fs[b][q] + Hs[sym10(0, kHalf + b)] * fe[b][q]) +
                     fe[0][q] * (Hs[sym10(kHalf, b)] * fs[b][q] + Hs[sym10(kHalf, kHalf + b)] * fe[b][q]);
        qf += extra + pos_fix;
      }
    }
    double ue[kNB][ND];
#pragma unroll
    for (int q = 0; q < ND; ++q) {
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        y[r][q] -= u[kSlot0 + r][q];
        ue[r][q] = u[kHalf + kSlot0 + r][q];
      }
    }
    // start block of this segment completes vertex v's diagonal block
#pragma unroll
    for (int r = 0; r < kNB; ++r)
#pragma unroll
      for (int c = 0; c <= r; ++c) Sm[tri(r, c)] += Hs[sym10(kSlot0 + r, kSlot0 + c)];
    factor_vertex(free_s, L, z);
    // W = L^-1 E~,  E~ = coupling block with constrained rows (vertex v) / columns (vertex v+1) zeroed
#pragma unroll
    for (int c = 0; c < kNB; ++c) {
      const bool fc = (free_e >> c) & 1u;
#pragma unroll
      for (int r = 0; r < kNB; ++r) {
        const bool fr = (free_s >> r) & 1u;
        double s = (fr && fc) ? Hs[sym10(kSlot0 + r, kHalf + kSlot0 + c)] : 0.0;
#pragma unroll
        for (int m = 0; m < r; ++m) s -= L[tri(r, m)] * W[m][c];
        W[r][c] = s * Linv[r];
      }
    }
    // next vertex: Sm = Hee - W^T W ; y = -ue - W^T z
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
#pragma unroll
      for (int c = 0; c <= r; ++c) {
        double s = Hs[sym10(kHalf + kSlot0 + r, kHalf + kSlot0 + c)];
#pragma unroll
        for (int m = 0; m < kNB; ++m) s -= W[m][r] * W[m][c];
        Sm[tri(r, c)] = s;
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) {
        double s = -ue[r][q];
#pragma unroll
        for (int m = 0; m < kNB; ++m) s -= W[m][r] * z[m][q];
        y[r][q] = s;
      }
    }
  }
};

// x = L^-T (z - W x_next)   (x_next = 0 and W ignored for the last vertex)
template <int ND>
__device__ __forceinline__ void back_substitute(const double (&L)[10], const double (&z)[kNB][ND],
                                                const double (&W)[kNB][kNB], const double (&xn)[kNB][ND], bool last,
                                                double (&x)[kNB][ND]) {
#pragma unroll
  for (int q = 0; q < ND; ++q) {
    double t[kNB];
#pragma unroll
    for (int r = 0; r < kNB; ++r) {
      double s = z[r][q];
      if (!last) {
#pragma unroll
        for (int c = 0; c < kNB; ++c) s -= W[r][c] * xn[c][q];
      }
      t[r] = s;
    }
#pragma unroll
    for (int r = kNB - 1; r >= 0; --r) {
      double s = t[r];
#pragma unroll
      for (int m = r + 1; m < kNB; ++m) s -= L[tri(m, r)] * x[m][q];
      x[r][q] = s / L[tri(r, r)];
    }
  }
}

// c_k = T^-k * sum_j ABAR_INV[k][j] * T^(j%5) * d_j   for one dimension; d = [d_start(5); d_end(5)]
__device__ __forceinline__ void coefficients_from_time(double T, const double (&dv)[kN], double (&c)[kN]) {
  double w[kHalf];
  w[0] = 1.0;
#pragma unroll
  for (int k = 1; k < kHalf; ++k) w[k] = w[k - 1] * T;
  double db[kN];
#pragma unroll
  for (int j = 0; j < kN; ++j) db[j] = dv[j] * w[j % kHalf];
  const double ti = 1.0 / T;
  double tik = 1.0;
#pragma unroll
  for (int k = 0; k < kN; ++k) {
    double s = 0.0;
    if (k < kHalf) {
      s = c_abar_inv[k][k] * db[k];  // the upper half of ABAR_INV is diag(1/k!)
    } else {
#pragma unroll
      for (int j = 0; j < kN; ++j) s += c_abar_inv[k][j] * db[j];
    }
    c[k] = s * tik;
    tik *= ti;
  }
}

// Horner with the derivative table (same evaluation scheme as Polynomial::evaluate,
// include/eth_trajectory_generation/polynomial.h:150-163)
__device__ __forceinline__ double poly_eval(const double* c, double t, int derivative) {
  // B[derivative][j] = j!/(j-derivative)!
  auto bc = [derivative](int j) {
    double v = 1.0;
    for (int n = 0; n < derivative; ++n) v *= (double)(j - n);
    return v;
  };
  double acc = bc(kN - 1) * c[kN - 1];
  for (int j = kN - 2; j >= derivative; --j) acc = acc * t + bc(j) * c[j];
  return acc;
}

// B[D][i] B[D][j] / (i + j - 2D + 1): 0.5 c^T Q c = T^(1-2D) sum_ij G_ij (c_i T^i)(c_j T^j)
// (computeQuadraticCostJacobian, linear_impl.h:606-618, with the factor 2 of Q against the 0.5 of the cost)
template <int D>
__device__ __forceinline__ double cost_quadratic_form(const double (&cb)[kN]) {
  double total = 0.0;
#pragma unroll
  for (int i = D; i < kN; ++i) {
    double bi = 1.0;
#pragma unroll
    for (int n = 0; n < D; ++n) bi *= (double)(i - n);
    double inner = 0.0;
#pragma unroll
    for (int j = D; j < kN; ++j) {
      double bj = 1.0;
#pragma unroll
      for (int n = 0; n < D; ++n) bj *= (double)(j - n);
      inner = fma(bi * bj / (double)(i + j - 2 * D + 1), cb[j], inner);
    }
    total = fma(cb[i], inner, total);
  }
  return total;
}

// cb_k = c_k T^k of one polynomial -> its cost 0.5 c^T Q c = T^(1-2d) * form (the reference's computeCost,
// linear_impl.h:128-141, for one segment and dimension)
__device__ __forceinline__ double cost_quadratic_form_d(int d, const double (&cb)[kN]) {
  switch (d) {
    case 0: return cost_quadratic_form<0>(cb);
    case 1: return cost_quadratic_form<1>(cb);
    case 2: return cost_quadratic_form<2>(cb);
    case 3: return cost_quadratic_form<3>(cb);
    default: return cost_quadratic_form<4>(cb);
  }
}

// violation scaling of one segment: max(1, v, sqrt(a), cbrt(j))  (trajectory.cpp:625-642)
__device__ __forceinline__ double violation_scaling(const double* __restrict__ mx, const double* __restrict__ lim) {
  double viol[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double h = mx[k * 3 + 0] / lim[k * 3 + 0];
    const double v = mx[k * 3 + 1] / lim[k * 3 + 1];
    const double y = mx[k * 3 + 2] / lim[k * 3 + 2];
    viol[k] = fmax(fmax(h, v), y);
  }
  return fmax(1.0, fmax(fmax(viol[0], sqrt(viol[1])), cbrt(viol[2])));
}

// ---------------------------------------------------------------------------------------------
// batch addressing shared by all kernels

struct PathRef {
  int p;   // path index in the caller's order
  int s0;  // first segment (CSR)
  int S;   // number of segments
  int v0;  // first vertex
};

__device__ __forceinline__ PathRef path_at(const BatchView& b, int q) {
  PathRef r;
  if (b.uniform_S > 0) {  // every path has the same segment count: the stable sort left the order alone, offsets are arithmetic
    r.p = q;              // (no dependent loads before a kernel can touch its inputs)
    r.S = b.uniform_S;
    r.s0 = q * b.uniform_S;
    r.v0 = r.s0 + q;
    return r;
  }
  r.p = b.order[q];
  r.s0 = b.seg_offsets[r.p];
  r.S = b.seg_offsets[r.p + 1] - r.s0;
  r.v0 = r.s0 + r.p;
  return r;
}

// constrained values (0 where free) and the free mask of the candidate slots of one vertex
template <int ND>
__device__ __forceinline__ unsigned load_vertex(const uint8_t* __restrict__ mask, const double* __restrict__ vals, int v,
                                                int dim0, double (&f)[kHalf][ND], bool& position_fixed) {
  // all loads are unconditional (the ABI guarantees the value array exists for every slot; entries of
  // unconstrained slots are ignored), so they issue back to back instead of one branch per slot
  const uint8_t* __restrict__ mrow = mask + (size_t)v * kHalf;
  const double* __restrict__ vrow = vals + (size_t)v * kHalf * kD + dim0;
  uint8_t mk[kHalf];
  double raw[kHalf][ND];
#pragma unroll
  for (int k = 0; k < kHalf; ++k) mk[k] = mrow[k];
#pragma unroll
  for (int k = 0; k < kHalf; ++k)
#pragma unroll
    for (int q = 0; q < ND; ++q) raw[k][q] = vrow[k * kD + q];
  unsigned free_bits = 0;
#pragma unroll
  for (int k = 0; k < kHalf; ++k) {
    const bool fixed = mk[k] != 0;
    if (k == 0) position_fixed = fixed;
    if (k >= kSlot0 && !fixed) free_bits |= 1u << (k - kSlot0);
#pragma unroll
    for (int q = 0; q < ND; ++q) f[k][q] = fixed ? raw[k][q] : 0.0;
  }
  return free_bits;
}

// Per-path status written by the solve kernels: -2 when a vertex leaves its position free (unsupported);
// otherwise 1 for the plain linear solve, or the outer loop's stopping reason when one is handed in
// (a start rejected by the optimiser, recorded there as -2, surfaces as FAILURE -1 like the reference's
// caught NLopt exception, nonlinear_impl.h:193-197).
__device__ __forceinline__ int merge_status(bool pos_ok, const int32_t* __restrict__ status_in, int p) {
  if (!pos_ok) return -2;
  if (!status_in) return 1;
  const int s = status_in[p];
  return s == -2 ? -1 : s;
}

constexpr int kWsPerVertex = 10 + kNB * kD + kNB * kNB;  // L, z, W = 42 doubles per vertex

}  // namespace mrs_tg
