/* solve_batch_host.c -- the C ABI from plain C99 (no C++, no HIP headers, no torch): two rest-to-rest paths, minimum snap,
 * Mellinger time allocation, sampled at 0.2 s -- on one device, then sharded over a device list.
 * tests/test_gpu_cpp_host.py builds it with gcc and checks the output.
 *
 *   gcc -std=c99 -I include examples/solve_batch_host.c -o solve_batch_host -L mrs_uav_trajectory_generation_amd -lmrs_tg
 */
#include <stdio.h>
#include <string.h>

#include "mrs_tg.h"

#define N_PATHS 2
#define CAP 512

int main(void) {
  /* path 0: the reference tests' four waypoints; path 1: a straight 20 m line in two segments */
  const int32_t seg_offsets[N_PATHS + 1] = {0, 3, 5};
  const double wp[(3 + 1 + 2 + 1) * 4] = {-5, -5, 5, 1, -5, 5, 5, 2, 5, -5, 5, 3, 5, 5, 5, 4, /* path 1 */ 0, 0, 2, 0, 10, 0, 2, 0, 20, 0, 2, 0};
  const int n_vertices = 7;
  uint8_t mask[7 * 5];
  double vals[7 * 5 * 4];
  memset(vals, 0, sizeof(vals));
  for (int v = 0; v < n_vertices; ++v) {
    const int is_end = (v == 0 || v == 3 || v == 4 || v == 6);
    for (int k = 0; k < 5; ++k) mask[v * 5 + k] = (k == 0 || is_end) ? 1 : 0; /* makeStartOrEnd(SNAP) at the ends */
    for (int d = 0; d < 4; ++d) vals[(v * 5 + 0) * 4 + d] = wp[v * 4 + d];
  }
  double limits[N_PATHS * 9];
  const double lim9[9] = {2, 2, 1, 2, 2, 2, 20, 20, 20};
  for (int p = 0; p < N_PATHS; ++p) memcpy(limits + p * 9, lim9, sizeof(lim9));

  mrs_tg_ctx* ctx = NULL;
  if (mrs_tg_create(0, &ctx) != MRS_TG_OK) {
    fprintf(stderr, "mrs_tg_create: %s\n", mrs_tg_last_error(NULL));
    return 1;
  }
  mrs_tg_options opt;
  mrs_tg_default_options(&opt);
  opt.derivative_to_optimize = 4;
  opt.time_alloc_method = MRS_TG_TIME_ALLOC_MELLINGER;
  opt.estimate_times = 1;
  opt.sampling_dt = 0.2;
  opt.sample_capacity = CAP;

  double seg_times[5] = {0}, coeffs[5 * 4 * 10], cost[N_PATHS];
  static double samples[N_PATHS * CAP * 4];
  int32_t status[N_PATHS], n_samples[N_PATHS];
  const int rc = mrs_tg_solve_batch(ctx, N_PATHS, seg_offsets, wp, mask, vals, limits, &opt, seg_times, coeffs, status, cost,
                                    n_samples, samples);
  if (rc != MRS_TG_OK) {
    fprintf(stderr, "mrs_tg_solve_batch: %s\n", mrs_tg_last_error(ctx));
    mrs_tg_destroy(ctx);
    return 1;
  }
  printf("{\"abi\": %d, \"paths\": [", mrs_tg_abi_version());
  for (int p = 0; p < N_PATHS; ++p) {
    printf("%s{\"status\": %d, \"cost\": %.9g, \"n_samples\": %d, \"times\": [", p ? ", " : "", status[p], cost[p], n_samples[p]);
    for (int s = seg_offsets[p]; s < seg_offsets[p + 1]; ++s) printf("%s%.9g", s > seg_offsets[p] ? ", " : "", seg_times[s]);
    const double* last = samples + ((size_t)p * CAP + (n_samples[p] - 1)) * 4;
    printf("], \"first\": [%.9g, %.9g, %.9g], \"last\": [%.9g, %.9g, %.9g]}", samples[(size_t)p * CAP * 4], samples[(size_t)p * CAP * 4 + 1],
           samples[(size_t)p * CAP * 4 + 2], last[0], last[1], last[2]);
  }
  printf("]");
  mrs_tg_destroy(ctx);

  /* the same batch over a device list (here: device 0 twice -- on an 8-GPU node {0, 1, .., 7}): every device solves its
   * shard on its own host thread, the results land in the caller's buffers; same numbers as the single-device call */
  {
    const int devices[2] = {0, 0};
    mrs_tg_multi* multi = NULL;
    double seg_times2[5] = {0}, coeffs2[5 * 4 * 10], cost2[N_PATHS];
    static double samples2[N_PATHS * CAP * 4];
    int32_t status2[N_PATHS], n_samples2[N_PATHS], shard[N_PATHS];
    if (mrs_tg_create_multi(devices, 2, &multi) != MRS_TG_OK) {
      fprintf(stderr, "mrs_tg_create_multi: %s\n", mrs_tg_last_error(NULL));
      return 1;
    }
    mrs_tg_multi_shard(multi, N_PATHS, seg_offsets, shard);
    if (mrs_tg_multi_solve_batch(multi, N_PATHS, seg_offsets, wp, mask, vals, limits, &opt, seg_times2, coeffs2, status2, cost2,
                                 n_samples2, samples2) != MRS_TG_OK) {
      fprintf(stderr, "mrs_tg_multi_solve_batch: %s\n", mrs_tg_multi_last_error(multi));
      mrs_tg_destroy_multi(multi);
      return 1;
    }
    printf(", \"multi\": {\"devices\": %d, \"shard\": [%d, %d], \"identical\": %s}", mrs_tg_multi_n_devices(multi), shard[0], shard[1],
           (memcmp(coeffs, coeffs2, sizeof(coeffs)) == 0 && memcmp(seg_times, seg_times2, sizeof(seg_times)) == 0 &&
            memcmp(status, status2, sizeof(status)) == 0 && memcmp(n_samples, n_samples2, sizeof(n_samples)) == 0)
               ? "true"
               : "false");
    mrs_tg_destroy_multi(multi);
  }
  printf("}\n");
  return 0;
}
