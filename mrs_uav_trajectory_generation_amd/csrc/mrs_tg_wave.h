// mrs_tg_wave.h -- launch interface of the one-wavefront-per-path outer-loop kernel (mrs_tg_wave.hip; internal to the library).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mrs_tg_launch.h"
#include "mrs_tg_nonlinear.h"

namespace mrs_tg {

constexpr int kWaveMaxS = 12;  // 4 (S + 4) lanes <= 64: both directions of the two-sided evaluation in one wavefront

// small batches (the plan's dim_split == 4) whose longest path has at most kWaveMaxS segments; MRS_TG_WAVE_KERNEL=0 disables
bool wave_kernel_applies(const BatchView& b, int dim_split);
// the whole outer loop of every path of the batch: seg_times in / out (last evaluated point), opt_status out (stopping reason,
// -2: start rejected); the events, if any, ride on the launch
hipError_t launch_optimize_wave(const BatchView& b, const NonlinearParams& prm, const uint8_t* mask, const double* vals,
                                double* seg_times, int32_t* opt_status, hipStream_t stream, hipEvent_t ev_start,
                                hipEvent_t ev_stop);

}  // namespace mrs_tg
