// mrs_tg_multi.hip -- a batch over several devices (include/mrs_tg.h, "several devices").
//
// Paths are independent units -- the reference itself solves one path per request on one worker thread
// (/root/reference/src/mrs_trajectory_generation.cpp:1064-1083, 1513) -- so a batch shards with no exchange between the
// devices: every device gets a contiguous range of paths (uniform batches: the shard is a pointer offset into the
// caller's buffers) or a balanced subset (ragged batches: longest path first onto the least loaded device, packed into a
// shard-local batch and scattered back), one host thread drives each device through mrs_tg_solve_batch, and the "gather"
// is the device-to-host copy of each shard into the caller's output buffers.  Host code only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mrs_tg.h"

struct mrs_tg_multi {
  std::vector<mrs_tg_ctx*> ctx;
  std::string last_error;
};

namespace {

// contiguous shards whose sizes differ by at most one (uniform batches)
void contiguous_shards(int n_paths, int n_dev, std::vector<int32_t>& shard) {
  const int base = n_paths / n_dev, rem = n_paths % n_dev;
  int p = 0;
  for (int r = 0; r < n_dev; ++r) {
    const int n = base + (r < rem ? 1 : 0);
    for (int i = 0; i < n; ++i) shard[p++] = r;
  }
}

// longest-processing-time-first on the segment count (ragged batches)
void balanced_shards(int n_paths, const int32_t* so, int n_dev, std::vector<int32_t>& shard) {
  std::vector<int> order(n_paths);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [so](int a, int b) { return (so[a + 1] - so[a]) > (so[b + 1] - so[b]); });
  std::vector<long long> load(n_dev, 0);
  for (int p : order) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    shard[p] = r;
    load[r] += so[p + 1] - so[p];
  }
}

bool is_uniform(int n_paths, const int32_t* so) {
  for (int p = 1; p < n_paths; ++p)
    if (so[p + 1] - so[p] != so[1] - so[0]) return false;
  return true;
}

}  // namespace

extern "C" {

int mrs_tg_create_multi(const int* device_ordinals, int n_devices, mrs_tg_multi** multi_out) {
  if (!multi_out) return MRS_TG_ERR_INVALID_ARG;
  *multi_out = nullptr;
  if (!device_ordinals || n_devices < 1) return MRS_TG_ERR_INVALID_ARG;
  mrs_tg_multi* m = new (std::nothrow) mrs_tg_multi();
  if (!m) return MRS_TG_ERR_NOMEM;
  for (int i = 0; i < n_devices; ++i) {
    mrs_tg_ctx* c = nullptr;
    const int rc = mrs_tg_create(device_ordinals[i], &c);
    if (rc != MRS_TG_OK) {
      mrs_tg_destroy_multi(m);
      return rc;
    }
    m->ctx.push_back(c);
  }
  *multi_out = m;
  return MRS_TG_OK;
}

void mrs_tg_destroy_multi(mrs_tg_multi* multi) {
  if (!multi) return;
  for (mrs_tg_ctx* c : multi->ctx) mrs_tg_destroy(c);
  delete multi;
}

int mrs_tg_multi_n_devices(const mrs_tg_multi* multi) { return multi ? (int)multi->ctx.size() : 0; }

mrs_tg_ctx* mrs_tg_multi_context(mrs_tg_multi* multi, int index) {
  if (!multi || index < 0 || index >= (int)multi->ctx.size()) return nullptr;
  return multi->ctx[index];
}

const char* mrs_tg_multi_last_error(const mrs_tg_multi* multi) { return multi ? multi->last_error.c_str() : ""; }

int mrs_tg_multi_shard(const mrs_tg_multi* multi, int32_t n_paths, const int32_t* seg_offsets, int32_t* shard_out) {
  if (!multi || !seg_offsets || !shard_out || n_paths < 0) return MRS_TG_ERR_INVALID_ARG;
  std::vector<int32_t> shard((size_t)n_paths);
  const int n_dev = (int)multi->ctx.size();
  if (is_uniform(n_paths, seg_offsets)) contiguous_shards(n_paths, n_dev, shard);
  else balanced_shards(n_paths, seg_offsets, n_dev, shard);
  std::copy(shard.begin(), shard.end(), shard_out);
  return MRS_TG_OK;
}

int mrs_tg_multi_solve_batch(mrs_tg_multi* multi, int32_t n_paths, const int32_t* so, const double* wp, const uint8_t* mask,
                             const double* vals, const double* limits, const mrs_tg_options* opt, double* seg_times,
                             double* coeffs, int32_t* status, double* cost, int32_t* n_samples, double* samples) {
  if (!multi) return MRS_TG_ERR_INVALID_ARG;
  multi->last_error.clear();
  if (!so || !mask || !vals || !opt || !seg_times || !coeffs || !status || n_paths < 0) {
    multi->last_error = "seg_offsets, fixed_mask, fixed_values, options, seg_times, coeffs_out, status_out are required";
    return MRS_TG_ERR_INVALID_ARG;
  }
  if (n_paths == 0) return MRS_TG_OK;
  const int n_dev = (int)multi->ctx.size();
  std::vector<int32_t> shard((size_t)n_paths);
  const bool uniform = is_uniform(n_paths, so);
  if (uniform) contiguous_shards(n_paths, n_dev, shard);
  else balanced_shards(n_paths, so, n_dev, shard);
  const size_t cap = (opt->sampling_dt > 0 && samples) ? (size_t)opt->sample_capacity : 0;

  std::vector<int> rcs((size_t)n_dev, MRS_TG_OK);
  std::vector<std::string> errs((size_t)n_dev);
  auto run_shard = [&](int r) {
    mrs_tg_ctx* ctx = multi->ctx[r];
    std::vector<int> mine;
    for (int p = 0; p < n_paths; ++p)
      if (shard[p] == r) mine.push_back(p);
    if (mine.empty()) return;
    const int n = (int)mine.size();
    int rc;
    if (uniform) {
      // a contiguous range: the shard's arrays are slices of the caller's
      const int p0 = mine.front();
      const size_t s0 = (size_t)so[p0], v0 = s0 + (size_t)p0;
      std::vector<int32_t> lso((size_t)n + 1);
      for (int i = 0; i <= n; ++i) lso[i] = so[p0 + i] - so[p0];
      rc = mrs_tg_solve_batch(ctx, n, lso.data(), wp ? wp + v0 * 4 : nullptr, mask + v0 * 5, vals + v0 * 20,
                              limits ? limits + (size_t)p0 * 9 : nullptr, opt, seg_times + s0, coeffs + s0 * 40, status + p0,
                              cost ? cost + p0 : nullptr, n_samples ? n_samples + p0 : nullptr,
                              cap ? samples + (size_t)p0 * cap * 4 : nullptr);
    } else {
      // a subset: pack, solve, scatter back
      std::vector<int32_t> lso((size_t)n + 1, 0);
      for (int i = 0; i < n; ++i) lso[i + 1] = lso[i] + (so[mine[i] + 1] - so[mine[i]]);
      const size_t nS = (size_t)lso[n], nV = nS + (size_t)n;
      std::vector<double> lwp(wp ? nV * 4 : 0), lvals(nV * 20), llim(limits ? (size_t)n * 9 : 0), lt(nS), lc(nS * 40),
          lcost((size_t)n), lsmp(cap ? (size_t)n * cap * 4 : 0);
      std::vector<uint8_t> lmask(nV * 5);
      std::vector<int32_t> lst((size_t)n), lns((size_t)n);
      for (int i = 0; i < n; ++i) {
        const int p = mine[i];
        const size_t s0 = (size_t)so[p], S = (size_t)(so[p + 1] - so[p]), v0 = s0 + (size_t)p, l0 = (size_t)lso[i],
                     lv0 = l0 + (size_t)i;
        if (wp) std::memcpy(lwp.data() + lv0 * 4, wp + v0 * 4, sizeof(double) * 4 * (S + 1));
        std::memcpy(lmask.data() + lv0 * 5, mask + v0 * 5, 5 * (S + 1));
        std::memcpy(lvals.data() + lv0 * 20, vals + v0 * 20, sizeof(double) * 20 * (S + 1));
        if (limits) std::memcpy(llim.data() + (size_t)i * 9, limits + (size_t)p * 9, sizeof(double) * 9);
        std::memcpy(lt.data() + l0, seg_times + s0, sizeof(double) * S);
      }
      rc = mrs_tg_solve_batch(ctx, n, lso.data(), wp ? lwp.data() : nullptr, lmask.data(), lvals.data(),
                              limits ? llim.data() : nullptr, opt, lt.data(), lc.data(), lst.data(), lcost.data(),
                              n_samples ? lns.data() : nullptr, cap ? lsmp.data() : nullptr);
      if (rc == MRS_TG_OK) {
        for (int i = 0; i < n; ++i) {
          const int p = mine[i];
          const size_t s0 = (size_t)so[p], S = (size_t)(so[p + 1] - so[p]), l0 = (size_t)lso[i];
          std::memcpy(seg_times + s0, lt.data() + l0, sizeof(double) * S);
          std::memcpy(coeffs + s0 * 40, lc.data() + l0 * 40, sizeof(double) * 40 * S);
          status[p] = lst[i];
          if (cost) cost[p] = lcost[i];
          if (n_samples) n_samples[p] = lns[i];
          if (cap) std::memcpy(samples + (size_t)p * cap * 4, lsmp.data() + (size_t)i * cap * 4, sizeof(double) * 4 * cap);
        }
      }
    }
    rcs[r] = rc;
    if (rc != MRS_TG_OK) errs[r] = mrs_tg_last_error(ctx);
  };
  std::vector<std::thread> workers;
  for (int r = 1; r < n_dev; ++r) workers.emplace_back(run_shard, r);
  run_shard(0);
  for (std::thread& t : workers) t.join();
  for (int r = 0; r < n_dev; ++r)
    if (rcs[r] != MRS_TG_OK) {
      multi->last_error = "device " + std::to_string(r) + ": " + errs[r];
      return rcs[r];
    }
  return MRS_TG_OK;
}

}  // extern "C"
