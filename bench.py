#!/usr/bin/env python3
"""bench.py -- headline benchmark: trajectories/s for a batch of 10-segment min-snap paths.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch whose inputs already sit in HBM.  Default
workload = BASELINE.json configs[1]: 1024 random 10-segment order-10 min-snap paths, fixed (Euclidean)
segment times, linear QP only: the Hessian/mapping-block assembly kernel + the block-Cholesky solve
kernel per step.  `--workload nonlinear` times configs[2] (Mellinger outer loop + feasibility scaling +
sampling) instead.  Weak scaling: every rank owns `--paths` paths and there is no data-path collective
(paths are independent); as at N = 1 the results of a step stay resident in the HBM of the GPU that
computed them, and the job's one collective -- the RCCL gather of the final step's coefficients / times /
status to rank 0 ("exactly one gather at the end") -- runs inside the timed region.
`--gather every` gathers after every step instead (double-buffered on a side stream); for N > 1 that
rate is reported next to the headline as extras.gather_every_step.

Steps are independent batches, so `--in-flight` of them (default 4 = the HIP runtime's hardware queues per process)
are kept in flight per GPU, each on its own HIP stream with its own context and plan: a 1024-path batch leaves most of
an MI355X idle (one wavefront per CU in the serial phase of the solve), and the assembly of one batch overlaps the
solves of the others.  Every step still does all
of its work; `extras.one_batch_in_flight` is the same measurement with one stream (each step waits for the previous).

Prints ONE JSON line (rank 0) with `roofline` (assembly kernel, HBM-write bound, HIP-event timed on the
launch stream) and `cpu_baseline` (the C oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ASSEMBLY_BYTES_PER_SEGMENT = 8 + 800 + 800   # read T, write full 10x10 f64 H and A^-1


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["linear", "nonlinear"], default="linear")
    ap.add_argument("--paths", type=int, default=1024, help="paths per GPU")
    ap.add_argument("--segments", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="single-core work of the cpu_baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (other workload) measurement")
    ap.add_argument("--gather", choices=["final", "every"], default="final",
                    help="N > 1: gather the results to rank 0 once at the end of the timed steps (default) or after every step")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (tests the RCCL path)")
    ap.add_argument("--in-flight", type=int, default=4,
                    help="independent batches in flight per GPU: steps are issued round-robin on this many HIP streams "
                         "(one context + plan each); 1 = every step waits for the previous one")
    return ap.parse_args()


def time_steps(step_fn, steps, warmup, dist, torch, final_fn=None):
    """W untimed steps, then exactly K steps (+ final_fn, the job's closing gather) between barrier + synchronize
    pairs; returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    if final_fn is not None:
        final_fn()          # warm the collective up as well (communicator set-up is not part of a step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    if final_fn is not None:
        final_fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def kernel_event_ms(launch_fn, reps, torch):
    """Average duration of one launch, HIP events on the launch stream around every single launch."""
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
    for i in range(reps):
        starts[i].record()
        launch_fn()
        stops[i].record()
    torch.cuda.synchronize()
    per = sorted(s.elapsed_time(e) for s, e in zip(starts, stops))
    return float(np.mean(per)), float(per[len(per) // 2])


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        if args.force_dist and "RANK" not in os.environ:   # stand-alone single-rank rendezvous
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist is not None else 0)

    from mrs_uav_trajectory_generation_amd import api, problem as pr

    # every rank generates its own shard: path p of rank r is seeded with r * paths + p
    batch = pr.random_batch(args.paths, args.segments, seed0=rank * args.paths)
    ctx = api.Context(dev.index)
    ctx.use_torch_stream()
    plan = api.Plan(ctx, batch.seg_offsets)
    db = api.DeviceBatch(batch, dev, sample_capacity=512)
    nS, P = batch.n_segments, batch.n_paths

    # initial (Euclidean) segment times are computed once on the device and kept: "fixed times"
    est = api.default_options(derivative_to_optimize=4, estimate_times=1)
    plan.solve(est, db.fixed_mask, db.fixed_values, db.seg_times, db.coeffs, db.status, db.cost,
               waypoints=db.waypoints, limits=db.limits)
    torch.cuda.synchronize()
    t_init = db.seg_times.clone()
    t_fixed = t_init.clone()   # linear mode never writes the times

    opt_lin = api.default_options(derivative_to_optimize=4)
    opt_nl = api.default_options(derivative_to_optimize=4, time_alloc_method=api.TIME_ALLOC_MELLINGER,
                                 sampling_dt=0.2, sample_capacity=512)

    from mrs_uav_trajectory_generation_amd import shard

    # Output double buffer: results of step k are gathered to rank 0 on a side stream while step k+1 computes
    # (the gather is the job's only collective; RCCL over xGMI).  Coefficients, times and status share one
    # f64 buffer per slot so that the gather is a single collective.
    n_lanes = max(1, args.in_flight)   # batches in flight: lane 0 = torch's current stream (ctx, plan), lanes 1.. = side streams
    lane_stream = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_lanes - 1)]
    lane_ctx, lane_plan = [ctx], [plan]
    for st in lane_stream[1:]:
        with torch.cuda.stream(st):
            c = api.Context(dev.index)
            c.use_torch_stream()
            lane_ctx.append(c)
            lane_plan.append(api.Plan(c, batch.seg_offsets))
    n_slots = max(n_lanes, 2 if dist is not None else 1)   # slot s runs on lane s % n_lanes
    active_lanes = [n_lanes]                               # extras.one_batch_in_flight sets this to 1
    packed = [torch.zeros(nS * 41 + P, dtype=torch.float64, device=dev) for _ in range(n_slots)]
    out_coeffs = [p[:nS * 40].view(nS, 4, 10) for p in packed]
    out_times = [p[nS * 40:nS * 41] for p in packed]
    status_i32 = [torch.zeros(P, dtype=torch.int32, device=dev) for _ in range(n_slots)]
    for tt in out_times:
        tt.copy_(t_init)
    db.coeffs, db.seg_times, db.status = out_coeffs[0], out_times[0], status_i32[0]
    comm_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    recv = [[torch.empty_like(packed[0]) for _ in range(world)] if (dist is not None and rank == 0) else None
            for _ in range(n_slots)]
    slot_free = [None] * n_slots      # event: the gather that read this slot has finished
    step_no = [0]
    # per-slot side outputs and the pre-bound solve calls (arguments converted once: a few us of host time per step)
    slot_cost = [torch.zeros(P, dtype=torch.float64, device=dev) for _ in range(n_slots)]
    slot_nsamp = [torch.zeros(P, dtype=torch.int32, device=dev) for _ in range(n_slots)]
    slot_samples = [db.samples] + [torch.zeros_like(db.samples) for _ in range(n_slots - 1)]
    bound = {}

    def slot_call(kind, slot, lane):
        key = (kind, slot, lane)
        if key not in bound:
            if kind == "linear":
                bound[key] = lane_plan[lane].bind_solve(opt_lin, db.fixed_mask, db.fixed_values, t_fixed, out_coeffs[slot],
                                                        status_i32[slot], slot_cost[slot])
            else:
                bound[key] = lane_plan[lane].bind_solve(opt_nl, db.fixed_mask, db.fixed_values, out_times[slot], out_coeffs[slot],
                                                        status_i32[slot], slot_cost[slot], limits=db.limits,
                                                        n_samples=slot_nsamp[slot], samples=slot_samples[slot])
        return bound[key]

    def finish_step(slot, lane):
        if dist is None:
            return
        with torch.cuda.stream(lane_stream[lane]):
            packed[slot][nS * 41:].copy_(status_i32[slot])          # int32 -> f64 tail of the packed buffer
            ready = torch.cuda.Event()
            ready.record()
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            shard.gather_to_root(packed[slot], dist, bufs=recv[slot])
            done = torch.cuda.Event()
            done.record()
        slot_free[slot] = done

    def begin_step():
        """next slot and the lane (stream) it runs on; the lane waits until the gather that read the slot is done"""
        lanes = active_lanes[0]
        slots = max(lanes, 2 if dist is not None else 1)
        slot = step_no[0] % slots
        lane = slot % lanes
        step_no[0] += 1
        if slot_free[slot] is not None:
            lane_stream[lane].wait_event(slot_free[slot])
            slot_free[slot] = None
        return slot, lane

    gather_every = [args.gather == "every"]
    last_slot = [(0, 0)]

    def step_linear():
        slot, lane = begin_step()
        slot_call("linear", slot, lane)()
        last_slot[0] = (slot, lane)
        if gather_every[0]:
            finish_step(slot, lane)

    def step_nonlinear():
        slot, lane = begin_step()
        with torch.cuda.stream(lane_stream[lane]):
            out_times[slot].copy_(t_init)   # the outer loop overwrites the times: restart from the same point
        slot_call("nonlinear", slot, lane)()
        last_slot[0] = (slot, lane)
        if gather_every[0]:
            finish_step(slot, lane)

    def final_gather():
        """the job's closing collective: results of the last step -> rank 0 (no-op at N = 1 without --force-dist)"""
        if dist is None or gather_every[0]:
            return
        finish_step(*last_slot[0])
        torch.cuda.current_stream().wait_event(slot_free[last_slot[0][0]])

    steps_fn = {"linear": step_linear, "nonlinear": step_nonlinear}
    elapsed = time_steps(steps_fn[args.workload], args.steps, args.warmup, dist, torch, final_gather)
    total_paths = P * world * args.steps
    value = total_paths / elapsed

    # ---- roofline of the assembly kernel (rank 0's device; HIP events on the launch stream) ----
    Hbuf = torch.empty(plan.block_doubles, dtype=torch.float64, device=dev)
    Abuf = torch.empty(plan.block_doubles, dtype=torch.float64, device=dev)

    def launch_assemble():
        plan.assemble(4, t_init, Hbuf, Abuf)

    for _ in range(10):
        launch_assemble()
    torch.cuda.synchronize()
    # (a) two events around a run of back-to-back launches: mean launch duration without per-launch event cost
    reps = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch_assemble()
    e1.record()
    torch.cuda.synchronize()
    mean_ms = e0.elapsed_time(e1) / reps
    # (b) an event pair around every single launch (adds the event packets' own time to each sample)
    per_launch_mean_ms, per_launch_med_ms = kernel_event_ms(launch_assemble, 200, torch)
    alg_bytes = ASSEMBLY_BYTES_PER_SEGMENT * nS
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
    # HBM traffic from PMC counters (separate rocprofv3 passes, profiles/round1_pmc_assemble_hbm_traffic.csv):
    # WRITE_SIZE calibrated on this kernel's store pattern (8 B per lane, 512 B per wave instruction) with a pure
    # fill of known size (k_fill8 in scripts/k1_variants.hip: exact); FETCH_SIZE is doubled as MI355X_MICROARCH.md
    # prescribes for gfx950.  Measured for the 1024 x 10 launch: 16000 KiB written + 2 x 105 KiB fetched.
    traffic = (16000 + 2 * 105) * 1024 if (P == 1024 and args.segments == 10) else None
    roofline = dict(kernel="assemble_blocks_uniform_kernel", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, bytes_per_launch=alg_bytes,
                    avg_launch_us=mean_ms * 1e3, per_launch_event_us=per_launch_mean_ms * 1e3,
                    per_launch_event_median_us=per_launch_med_ms * 1e3,
                    note="16 MB per launch: launch-ramp bound, see extras.roofline_large for the same kernel at 1 GB")

    extras = {}
    if not args.no_extras and rank == 0:
        # the same kernel on a batch that is not launch-ramp dominated (65536 x 10 segments = 1.05 GB per launch)
        big_P = 65536
        so_big = (np.arange(big_P + 1, dtype=np.int64) * args.segments).astype(np.int32)
        plan_big = api.Plan(ctx, so_big)
        t_big = t_init.repeat((big_P * args.segments + nS - 1) // nS)[:big_P * args.segments].contiguous()
        Hb = torch.empty(plan_big.block_doubles, dtype=torch.float64, device=dev)
        Ab = torch.empty(plan_big.block_doubles, dtype=torch.float64, device=dev)
        for _ in range(3):
            plan_big.assemble(4, t_big, Hb, Ab)
        torch.cuda.synchronize()
        m_big, _ = kernel_event_ms(lambda: plan_big.assemble(4, t_big, Hb, Ab), 20, torch)
        bytes_big = ASSEMBLY_BYTES_PER_SEGMENT * big_P * args.segments
        extras["roofline_large"] = dict(kernel="assemble_blocks_uniform_kernel", paths=big_P, bytes_per_launch=bytes_big,
                                        avg_launch_us=m_big * 1e3, achieved=bytes_big / (m_big * 1e-3) / 1e9,
                                        unit="GB/s", frac=bytes_big / (m_big * 1e-3) / 1e9 / HBM_PEAK_GBS)
        del Hb, Ab
        plan_big.close()
        # PCIe-inclusive rate of the one-call host interface (plan creation + H2D + kernels + D2H); never the headline
        times_host = t_init.cpu().numpy()
        ctx.solve_batch(batch, times_host)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.solve_batch(batch, times_host)
        extras["host_buffer_call"] = dict(value=5 * P / (time.perf_counter() - t0), unit="trajectories/s",
                                          note="mrs_tg_solve_batch with host buffers, linear QP, includes PCIe copies")
    if n_lanes > 1 and not args.no_extras:
        # the same steps with one batch in flight: every step waits for the previous one (single stream)
        torch.cuda.synchronize()
        active_lanes[0] = 1
        step_no[0] = 0
        el1 = time_steps(steps_fn[args.workload], args.steps, 3, dist, torch, final_gather)
        active_lanes[0] = n_lanes
        step_no[0] = 0
        if rank == 0:
            extras["one_batch_in_flight"] = dict(value=P * world * args.steps / el1, unit="trajectories/s",
                                                 ms_per_step=el1 / args.steps * 1e3)
    if not args.no_extras:
        other = "nonlinear" if args.workload == "linear" else "linear"
        k2 = max(5, args.steps // 10) if other == "nonlinear" else args.steps
        el2 = time_steps(steps_fn[other], k2, 3, dist, torch, final_gather)
        if rank == 0:
            extras[other] = dict(value=P * world * k2 / el2, unit="trajectories/s", steps=k2, ms_per_step=el2 / k2 * 1e3)
        if dist is not None and not gather_every[0]:
            # the same workload with the results of EVERY step gathered to rank 0 (bound by the xGMI links into the root)
            gather_every[0] = True
            el3 = time_steps(steps_fn[args.workload], args.steps, 3, dist, torch)
            gather_every[0] = False
            if rank == 0:
                extras["gather_every_step"] = dict(value=P * world * args.steps / el3, unit="trajectories/s",
                                                   ms_per_step=el3 / args.steps * 1e3,
                                                   bytes_per_rank_per_step=int(packed[0].numel() * 8))

    # ---- parity of this very batch against the oracle (max-coeff err vs CPU ref) + CPU baseline ----
    cpu = None
    err = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        step_no[0] = 0
        steps_fn["linear"]()
        torch.cuda.synchronize()
        db.coeffs = out_coeffs[0]
        times = t_init.cpu().numpy()
        n_cpu = min(P, 1024)
        sub = batch.select(range(n_cpu)) if n_cpu < P else batch
        sub_t = times[:sub.n_segments]
        t0 = time.perf_counter()
        ref = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, sub_t, deriv=4)
        dt1 = time.perf_counter() - t0
        # bounded sample of ~10 s of single-core work: the same batch solved again and again
        reps_cpu = max(1, min(2000, int(args.cpu_seconds / max(dt1, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(reps_cpu):
            po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, sub_t, deriv=4)
        dt1 = (time.perf_counter() - t0) / reps_cpu
        cores = os.cpu_count() or 1
        reps_all = max(1, reps_cpu // 8)
        t0 = time.perf_counter()
        for _ in range(reps_all):
            po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, sub_t, deriv=4,
                           n_threads=cores)
        dtn = (time.perf_counter() - t0) / reps_all
        gpu_c = db.coeffs.cpu().numpy()[:sub.n_segments]
        worst = 0.0
        for p in range(sub.n_paths):
            a, b = sub.seg_offsets[p], sub.seg_offsets[p + 1]
            worst = max(worst, float(np.max(np.abs(gpu_c[a:b] - ref["coeffs"][a:b])) / np.max(np.abs(ref["coeffs"][a:b]))))
        err = worst
        cpu = dict(value=n_cpu / dt1, unit="trajectories/s", cores=1, kind="port",
                   sample="%d x the first %d of the %d paths of this batch (%.1f s of one core), linear QP, C oracle "
                          "(reference-style arithmetic, dense QR); all-core figure: %d x the same batch"
                          % (reps_cpu, n_cpu, P, reps_cpu * dt1, reps_all),
                   value_all_cores=n_cpu / dtn, cores_all=cores)
        if args.workload == "nonlinear" or not args.no_extras:
            n_nl = min(P, 256)
            subn = batch.select(range(n_nl))

            def cpu_nonlinear():
                po.solve_batch(subn.seg_offsets, subn.waypoints, subn.fixed_mask, subn.fixed_values, subn.limits,
                               times[:subn.n_segments], deriv=4, time_alloc_method=2, sampling_dt=0.2, sample_capacity=512)
            t0 = time.perf_counter()
            cpu_nonlinear()
            dtnl = time.perf_counter() - t0
            reps_nl = max(1, min(100, int(0.5 * args.cpu_seconds / max(dtnl, 1e-6))))
            t0 = time.perf_counter()
            for _ in range(reps_nl):
                cpu_nonlinear()
            dtnl = (time.perf_counter() - t0) / reps_nl
            cpu["nonlinear_value"] = n_nl / dtnl
            cpu["nonlinear_sample"] = ("%d x %d paths (%.1f s of one core), Mellinger outer loop + scaling + sampling, 1 thread"
                                       % (reps_nl, n_nl, reps_nl * dtnl))

    if rank == 0:
        line = dict(metric="trajectories/sec (batch of N-seg min-snap paths)", value=value, unit="trajectories/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                    config=dict(workload=("BASELINE configs[1]: %d random %d-segment order-10 min-snap paths per GPU, "
                                          "fixed times, linear QP" if args.workload == "linear" else
                                          "BASELINE configs[2]: %d random %d-segment paths per GPU, Mellinger outer loop "
                                          "(<=10 evaluations) + feasibility scaling + sampling dt 0.2") % (P, args.segments),
                                paths_per_gpu=P, segments=args.segments, batches_in_flight=n_lanes,
                                parallelism=("independent paths sharded per rank, no data-path collective; RCCL gather of "
                                             "the results to rank 0 %s, inside the timed region"
                                             % ("after every step" if args.gather == "every" else "once, after the last step"))
                                if world > 1 else "single GPU"),
                    max_coeff_err_vs_cpu_ref=err, roofline=roofline, cpu_baseline=cpu, extras=extras)
        import ctypes
        ctypes.CDLL(None).fflush(None)   # RCCL's banner sits in C stdio: keep the JSON line the last thing printed
        print(json.dumps(line), flush=True)
    for pl in lane_plan[1:]:
        pl.close()
    for c in lane_ctx[1:]:
        c.close()
    plan.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
