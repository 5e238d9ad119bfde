#!/usr/bin/env python3
"""bench.py -- headline benchmark: trajectories/s for a batch of 10-segment min-snap paths.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a torch.distributed environment starts N ranks itself:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`
as a CHILD process, before this process has imported torch or touched the GPU, and relays the child's JSON line and
exit code.  Launched by torch.distributed.run (RANK / WORLD_SIZE in the environment) it is one of the ranks.

A "step" is one pass of the hot path over one batch whose inputs already sit in HBM.  Headline workload = BASELINE.json
configs[1]: 1024 random 10-segment order-10 min-snap paths per GPU, fixed (Euclidean) segment times, linear QP only, solved
by the library's default linear solve (mrs_tg_plan_solve, flags 0: every lane forms its column of the reduced system from
the segment times in registers -- no block is written to memory).  The Hessian / mapping-block ASSEMBLY kernel
(mrs_tg_plan_assemble, 1608 B per segment into HBM) is the `roofline` object; the step that runs it and then solves from
the materialised blocks is timed as extras.materialized_blocks_step.  `--workload nonlinear` times configs[2] (Mellinger
outer loop + feasibility scaling + sampling) instead.

Weak scaling: every rank owns `--paths` paths and there is no data-path collective (paths are independent); as at N = 1 the
results of a step stay resident in the HBM of the GPU that computed them, and the job's one collective -- the RCCL gather of
the final step's coefficients / times / status to rank 0 ("exactly one gather at the end") -- is timed on its
own right behind the K steps (`gather_ms`; `value_including_gather` has it inside): its fixed cost (one RCCL launch + 3.4 MB
per rank into the root) is as long as 20 of the 5 us steps and would otherwise be what a short run measures.  Each rank
times its own K steps between its own synchronizes, after the opening barrier and before the closing one; `value` uses the
MAX over ranks.  BASELINE configs[3] in its own terms -- ONE batch of 65536 nonlinear-time paths cut into contiguous shards
over the N ranks, results gathered to rank 0 INSIDE its timed region -- is measured in the same run at every N and reported
as extras.config3 and, in short, as the top-level `config3_strong_scaling` (strong scaling of a fixed batch).

Steps are independent batches -- every step is a full pass over its own batch into its own output arrays -- so `--in-flight`
of them (default 20) are kept in flight per GPU.  `--issue grouped` (default): `--group-size` (10) consecutive steps go out as
ONE dispatch (mrs_tg_bound_solve_launch_group: the single-batch solve kernel's body, its workgroups divided among the
batches), the dispatches alternating over two HIP streams; a runtime launch costs the host 3 us, so twenty one-step launches
were half of the driver's 20-step timed region.  WHICH two streams (round 6): the runtime maps HIP streams onto a few hardware
queues, and whether two streams' dispatches overlap depends on the pair and on the process (profiles/round6_stream_pairs.txt:
49, 57-60 or 66 us for the same two dispatches); `--stream-candidates` (8) streams are created and an untimed calibration of the
two-dispatch round picks the pair (config.issue_policy.stream_pair holds the table).  `--issue streams` is round 2's method -- one dispatch per step, round-robin over
`--streams` (4 = the hardware queues the runtime gives a process) HIP streams with their own contexts and plans -- and is
reported as extras.streams_in_flight; extras.one_batch_in_flight is one stream, every step waiting for the previous.  The
nonlinear workload keeps one batch in flight per stream.

Prints ONE JSON line (rank 0): `roofline` (assembly kernel, HBM-write bound; `achieved` uses the PER-DISPATCH duration --
events attached to the kernel launch itself, what rocprofv3 --kernel-trace reports for the dispatch; the back-to-back
launch interval is given beside it and labelled), `roofline_solve` / `roofline_outer_loop` (FP64, flop model of SURVEY.md 8d
over the per-dispatch duration), and `cpu_baseline` (rank 0 of an N = 1 run only; the C oracle on this box's host cores: one core, and all cores through
its persistent thread pool on the configs[3]-sized batch -- with as many threads as the container's cgroup grants CPUs,
`cpu_quota_cpus`, when that is less than the hardware threads it sees).
"""
import argparse
import gc
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz; scripts/dpp_probe.hip measures it (sustained)
ASSEMBLY_BYTES_PER_SEGMENT = 8 + 800 + 800   # read T, write full 10x10 f64 H and A^-1
SOLVE_FLOP_PER_SEGMENT = 6.0e3               # linear solve ~6e3 S flop per path (6e4 at S = 10)
NONLINEAR_FLOP_PER_PATH_S10 = 7.0e6          # mode 2, <= 121 linear solves at S = 10
CONFIG3_PATHS = 65536


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--regions", type=int, default=9,
                    help="timed regions of --steps steps each (barrier + synchronize on both sides of every one): `value` is "
                         "the MEDIAN region's throughput (BASELINE.md section 3: median of >= 5 repetitions)")
    ap.add_argument("--workload", choices=["linear", "nonlinear"], default="linear")
    ap.add_argument("--paths", type=int, default=1024, help="paths per GPU")
    ap.add_argument("--segments", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="single-core work of the cpu_baseline sample")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--gather", choices=["final", "every"], default="final",
                    help="N > 1: gather the results to rank 0 once at the end of the timed steps (default) or after every step")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for one rank (tests the RCCL path)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: the collectives go through host memory (lets several ranks share one GPU in a test)")
    ap.add_argument("--issue-threads", type=int, default=1,
                    help="host threads of the library's issue loop (mrs_tg_bound_solve_launch_many_mt): one runtime launch "
                         "costs the host more than four concurrent kernels take to retire one")
    ap.add_argument("--in-flight", type=int, default=20,
                    help="independent batches (sets of input / output arrays) in flight per GPU; 1 = every step waits for the "
                         "previous one.  --issue grouped: --group-size of them per dispatch; --issue streams and the "
                         "nonlinear workload: one per HIP stream, at most --streams of them")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams (one context + plan each) per GPU: the hardware queues the runtime gives a process")
    ap.add_argument("--stream-candidates", type=int, default=8,
                    help="grouped issue: HIP streams created, of which the pair whose two dispatches overlap best carries the timed "
                         "region (the runtime maps streams onto a few hardware queues; two streams on one queue serialize). "
                         "<= 2: no calibration, the first two lanes as until round 6")
    ap.add_argument("--issue", choices=["grouped", "streams"], default="grouped",
                    help="how the K linear steps are issued: 'grouped' packs the steps of a round (one per batch in flight) into "
                         "ONE dispatch on one stream (mrs_tg_bound_solve_launch_group); 'streams' issues one dispatch per step, "
                         "round-robin over --in-flight HIP streams (the round-2 method, reported as extras.streams_in_flight)")
    ap.add_argument("--group-size", type=int, default=10,
                    help="--issue grouped: batches per dispatch; the --in-flight batches are spread over in-flight / group-size "
                         "streams, each stream's dispatches carry group-size steps")
    ap.add_argument("--shared-inputs", dest="distinct_slots", action="store_false",
                    help="every batch in flight reads slot 0's input arrays (the round-3 set-up; default: each slot has its own batch)")
    ap.add_argument("--config3-paths", type=int, default=CONFIG3_PATHS, help="size of the fixed batch of extras.config3")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# self-launch (N > 1)

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(n_gpus, argv, port, python=sys.executable, script=os.path.abspath(__file__)):
    """The torch.distributed.run command that starts `n_gpus` ranks of this script with the caller's own arguments."""
    return [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def self_launch(args, argv):
    """Start the ranks as a child process (this process has not initialised the GPU), relay the JSON line, return its rc."""
    cmd = launcher_command(args.gpus, argv, _free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


# ---------------------------------------------------------------------------------------------------------------------
# helpers (run inside a rank)

LAST_OWN_ELAPSED = [0.0]
LAST_OWN_GATHER = [0.0]
LAST_OWN_REGIONS = [[0.0]]


def pick_stream_pair(n_candidates, measure_us):
    """The pair of lanes (a < b) with the shortest two-dispatch round: measure_us(a, b) -> microseconds.  Returns (best pair,
    {(a, b): us}).  (The choice among HIP streams of bench.py's grouped issue path, see calibrate_group_lanes in main().)"""
    table = {}
    for a in range(n_candidates):
        for b in range(a + 1, n_candidates):
            table[(a, b)] = float(measure_us(a, b))
    best = min(sorted(table), key=table.get)
    return best, table


def time_regions(step_fn, steps, warmup, dist, torch, regions=1, final_fn=None, block_fn=None, gather_inside=False):
    """W untimed steps, then `regions` timed regions of exactly K steps each, every region between its own barrier +
    synchronize pairs ("steady-state median of >= 5 repetitions after 1 warm-up").  Every rank reads its
    own clock right after its own synchronize (before the closing barrier, whose cost is a property of the collective library,
    not of the K steps); the figures returned are, per region, the MAX over ranks.  final_fn, the job's closing gather, is
    timed on its own right behind the LAST region's steps (second return value, MAX over ranks; 0.0 without a collective)
    unless gather_inside, where it belongs to every timed region (configs[3]: the fixed batch is not done before rank 0
    holds the results).  block_fn(n), when given, issues n steps (the host's issue loop in C,
    mrs_tg_bound_solve_launch_many) and replaces n calls of step_fn.  Returns ([elapsed per region], gather_s)."""
    if block_fn is not None:
        def run(n):
            if n > 0:
                block_fn(n)
    else:
        def run(n):
            for _ in range(n):
                step_fn()
    run(warmup)
    if final_fn is not None:
        final_fn()          # warm the collective up as well (communicator set-up is not part of a step)
    own = []
    gc_was_on = gc.isenabled()
    gc.disable()            # (a collection inside a 60 us region would be the measurement)
    for _ in range(max(1, regions)):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        if gather_inside and final_fn is not None:
            final_fn()
        torch.cuda.synchronize()
        own.append(time.perf_counter() - t0)
    if gc_was_on:
        gc.enable()
    LAST_OWN_REGIONS[0] = list(own)   # this rank's own figures, before the MAX over ranks
    LAST_OWN_ELAPSED[0] = own[0]
    gather_s = 0.0
    if final_fn is not None and not gather_inside and dist is not None:
        t1 = time.perf_counter()
        final_fn()
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - t1
    LAST_OWN_GATHER[0] = gather_s   # this rank's own closing gather
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor(own + [gather_s], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        own, gather_s = [float(v) for v in t[:-1].tolist()], float(t[-1].item())
    return own, gather_s


def time_steps(step_fn, steps, warmup, dist, torch, final_fn=None, block_fn=None, gather_inside=False):
    """one timed region of K steps (the secondary measurements): (elapsed, gather_s) of time_regions(..., regions=1)"""
    own, gather_s = time_regions(step_fn, steps, warmup, dist, torch, 1, final_fn, block_fn, gather_inside)
    return own[0], gather_s


def median_of(values):
    """the median as a value that was measured (the upper of the two middle ones for an even count)"""
    v = sorted(values)
    return v[len(v) // 2]


def dispatch_stats(ctx, kernel_id, launch_fn, reps, torch):
    """Per-dispatch durations of `reps` QUEUED launches of one kernel: the library attaches a pair of timing events to every
    kernel launch itself (mrs_tg_set_profiling / mrs_tg_kernel_ms_history), so each value is that dispatch's own start-to-end
    time -- what rocprofv3 --kernel-trace reports for the same launches of the same command.  Returns (mean, median, min) ms."""
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    try:
        for _ in range(reps):
            launch_fn()
        vals = ctx.kernel_ms_history(kernel_id, min(reps, 512))
    finally:
        ctx.set_profiling(False)
    torch.cuda.synchronize()
    vals.sort()
    return sum(vals) / len(vals), vals[len(vals) // 2], vals[0]


def isolated_dispatch_stats(ctx, kernel_id, launch_fn, reps, torch):
    """dispatch_stats with every launch waiting for the previous one to finish: the dispatch's own duration with no neighbour in
    flight (queued launches of a 5 us kernel overlap -- the next dispatch's ramp runs into this one's tail -- and each one's
    events then span more than its own work; rocprofv3 --kernel-trace sees the launches apart as well: the tool's work per
    launch separates them).  Returns (mean, median, min) ms."""
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    try:
        for _ in range(reps):
            launch_fn()
            torch.cuda.current_stream().synchronize()
        vals = ctx.kernel_ms_history(kernel_id, min(reps, 512))
    finally:
        ctx.set_profiling(False)
    vals.sort()
    return sum(vals) / len(vals), vals[len(vals) // 2], vals[0]


def back_to_back_ms(launch_fn, reps, torch):
    """(two events around `reps` back-to-back launches) / reps: the launch-to-launch interval, in which the tail of one
    dispatch overlaps the ramp of the next -- NOT a kernel duration."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch_fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def measured_traffic_solve_quad(n_paths, n_seg):
    """HBM traffic of solve_quad_kernel from the newest committed PMC summary (scripts/pmc_solve_quad.sh); not measured in this run"""
    best = None
    # (bench.py's solves state MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: the counters of solve_quad_kernel<true>)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_solve_quad_wp_hbm_traffic.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("paths") == n_paths and d.get("segments") == n_seg:
            best = (d, os.path.relpath(f, ROOT))
    return best


def cores_for_checks():
    visible, quota = host_parallelism()
    return max(1, min(visible, int(quota + 0.5))) if quota else min(visible, 16)


def host_parallelism():
    """(threads the process may run on, CPUs' worth of time its cgroup grants or None): the box reports 256 hardware threads,
    but a container's cpu.max quota is what bounds an all-core CPU baseline (16 CPUs on the GPU boxes of this pool)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        visible = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            quota = None
    return visible, quota


def measured_traffic(n_paths, n_seg):
    """HBM traffic of the assembly kernel from the newest committed PMC summary (separate rocprofv3 --pmc passes,
    scripts/pmc_assemble.py; MI355X_MICROARCH.md corrections applied there).  Not measured in this run."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_assemble_hbm_traffic.json"))):
        try:
            with open(f) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        if d.get("paths") == n_paths and d.get("segments") == n_seg:
            best = (d, os.path.relpath(f, ROOT))
    return best


def fill_ceiling(n_paths, n_seg):
    """Per-dispatch duration of the fastest PURE FILL of the assembly kernel's bytes (scripts/k1_variants.hip under
    rocprofv3 --kernel-trace --stats, committed): what any kernel that writes 1024 x 10 segments' blocks can reach at most."""
    if (n_paths, n_seg) != (1024, 10):
        return None
    import csv
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_k1_fill_ceiling_kernel_stats.csv"))):
        try:
            with open(f) as fh:
                rows = [r for r in csv.DictReader(fh) if "k_fill" in r["Name"]]
        except (OSError, KeyError):
            continue
        if rows:
            r = min(rows, key=lambda r: float(r["AverageNs"]))
            best = dict(kernel=r["Name"].split("(")[0].replace("void ", ""), avg_launch_us=float(r["AverageNs"]) / 1e3,
                        source=os.path.relpath(f, ROOT))
    return best


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, argv))   # nothing below has run: no torch import, no GPU call in this process

    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < n_dev else local_rank % max(n_dev, 1)   # gloo test: ranks share the GPU
    dist = None
    rccl_version = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        if args.force_dist and "RANK" not in os.environ:   # stand-alone single-rank rendezvous
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if args.dist_backend == "nccl" and world > n_dev:
            # fail fast, with a message: RCCL cannot put two ranks on one GPU, and a rank that wrapped around onto a device
            # another rank owns would hang in the communicator set-up instead of saying so
            raise SystemExit("bench.py: --gpus %d under RCCL needs %d visible GPUs, torch.cuda.device_count() = %d "
                             "(rank %d, local rank %d; --dist-backend gloo lets ranks share a GPU in a test)"
                             % (args.gpus, world, n_dev, rank, local_rank))
        torch.cuda.set_device(dev_index)
        if args.dist_backend == "nccl":
            try:
                rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as ex:   # (diagnostics only)
                rccl_version = "unknown (%s)" % type(ex).__name__
            if rank == 0:
                print("bench.py: RCCL %s, %d rank(s), HSA_ENABLE_IPC_MODE_LEGACY=%s" %
                      (rccl_version, world, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")), file=sys.stderr, flush=True)
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist_mod.init_process_group("gloo")
        dist = dist_mod
        ranks_seen = dist.get_world_size()
        if ranks_seen != max(args.gpus, 1) and not args.force_dist:
            raise SystemExit("bench.py: --gpus %d but torch.distributed reports %d ranks" % (args.gpus, ranks_seen))
    else:
        torch.cuda.set_device(0)
        dev_index = 0
        ranks_seen = 1
    dev = torch.device("cuda", dev_index)
    gloo = dist is not None and args.dist_backend == "gloo"

    from mrs_uav_trajectory_generation_amd import api, problem as pr, shard

    # every rank generates its own shard: path p of rank r is seeded with r * paths + p
    batch = pr.random_batch(args.paths, args.segments, seed0=rank * args.paths)
    ctx = api.Context(dev.index)
    ctx.use_torch_stream()
    # extras.roofline_large's output buffers (2 x 524 MB), allocated FIRST: where a process's buffers land decides the rate of a
    # 1 GB streaming write -- the first few GB a process allocates are written at 0.81-0.88 of the HBM peak, later allocations at
    # 0.62-0.65, stably, per buffer (profiles/round6_assembly_placement.txt: the "unexplained" 0.64-0.81 of rounds 4-5).  Both
    # placements are measured and reported.
    BIG_P = 65536
    early_big = None
    if rank == 0 and not args.no_extras:
        early_big = (torch.empty(BIG_P * args.segments * 100, dtype=torch.float64, device=dev),
                     torch.empty(BIG_P * args.segments * 100, dtype=torch.float64, device=dev))
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

    # [0]: the batch has the device to itself; [1]: several batches in flight (MRS_TG_FLAG_SHARED_DEVICE, a launch-shape hint)
    # (MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: every vertex of these batches has its waypoint as position constraint, as every
    # vertex findTrajectory builds; the saturated-device solve then reads the compact waypoint array -- checked at bind time)
    opt_lin = [api.default_options(derivative_to_optimize=4, flags=f | api.FLAG_POSITIONS_ARE_WAYPOINTS) for f in (0, api.FLAG_SHARED_DEVICE)]
    opt_blocks = [api.default_options(derivative_to_optimize=4, flags=api.FLAG_MATERIALIZED_BLOCKS | f)
                  for f in (0, api.FLAG_SHARED_DEVICE)]
    # the nonlinear step is findTrajectory's: segment-time estimate from the waypoints, outer loop from there, scaling,
    # sampling (estimate_times=1: every step starts from the same point without a reset of the times in front of it)
    opt_nl = [api.default_options(derivative_to_optimize=4, time_alloc_method=api.TIME_ALLOC_MELLINGER, estimate_times=1,
                                  sampling_dt=0.2, sample_capacity=512, flags=f | api.FLAG_POSITIONS_ARE_WAYPOINTS)
              for f in (0, api.FLAG_SHARED_DEVICE)]

    def gather(tensor, bufs):
        """the job's one collective: equally shaped per-rank results -> rank 0"""
        if gloo:
            shard.gather_to_root(tensor.cpu(), dist)
        else:
            shard.gather_to_root(tensor, dist, bufs=bufs)

    # Output double buffer: results of step k are gathered to rank 0 on a side stream while step k+1 computes
    # (the gather is the job's only collective; RCCL over xGMI).  Coefficients, times and status share one
    # f64 buffer per slot so that the gather is a single collective.
    # lanes = HIP streams (lane 0 = torch's current stream with ctx / plan, lanes 1.. = side streams with their own); slots =
    # sets of output arrays.  Stream-issued steps keep one batch in flight per lane; grouped steps --in-flight of them
    n_lanes = max(1, min(args.in_flight, args.streams))
    # grouped issue: a few more streams than it uses; group_lanes[g] = the lane that carries the g-th dispatch of a round, chosen by
    # calibrate_group_lanes() below (lanes beyond n_lanes are candidates only: stream-issued steps never run on them)
    n_cand = max(n_lanes, args.stream_candidates) if (args.issue == "grouped" and args.workload == "linear" and n_lanes > 1) else n_lanes
    group_lanes = list(range(n_cand))
    lane_stream = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(n_cand - 1)]
    lane_ctx, lane_plan = [ctx], [plan]
    for st in lane_stream[1:]:
        with torch.cuda.stream(st):
            c = api.Context(dev.index)
            c.use_torch_stream()
            lane_ctx.append(c)
            lane_plan.append(api.Plan(c, batch.seg_offsets))
    n_group_slots = max(1, args.in_flight)                                  # grouped issue: slot s on lane s // group size
    n_slots = max(n_lanes, 2 if dist is not None else 1, n_group_slots)     # stream issue: slot s runs on lane s % n_lanes
    active_lanes = [n_lanes]                               # extras.one_batch_in_flight sets this to 1
    packed = [torch.zeros(nS * 41 + P, dtype=torch.float64, device=dev) for _ in range(n_slots)]
    out_coeffs = [p[:nS * 40].view(nS, 4, 10) for p in packed]
    out_times = [p[nS * 40:nS * 41] for p in packed]
    status_i32 = [torch.zeros(P, dtype=torch.int32, device=dev) for _ in range(n_slots)]
    for tt in out_times:
        tt.copy_(t_init)
    db.coeffs, db.seg_times, db.status = out_coeffs[0], out_times[0], status_i32[0]
    comm_stream = torch.cuda.Stream(device=dev) if dist is not None else None
    recv = [[torch.empty_like(packed[0]) for _ in range(world)] if (dist is not None and rank == 0 and not gloo) else None
            for _ in range(n_slots)]
    slot_free = [None] * n_slots      # event: the gather that read this slot has finished
    step_no = [0]
    # per-slot side outputs and the pre-bound solve calls (arguments converted once: a few us of host time per step)
    slot_cost = [torch.zeros(P, dtype=torch.float64, device=dev) for _ in range(n_slots)]
    slot_nsamp = [torch.zeros(P, dtype=torch.int32, device=dev) for _ in range(n_slots)]
    slot_samples = [db.samples] + [torch.zeros_like(db.samples) for _ in range(n_slots - 1)]
    # Fixed-times steps: every batch in flight has its OWN inputs as well -- slot s of rank r holds the paths seeded
    # (r * slots + s) * paths + p (slot 0 of rank 0 = the paths 0 .. P-1 every other figure of the line is quoted on), its own
    # masks, constrained values and Euclidean times, so twenty batches in flight read twenty different batches.
    slot_batch, slot_db, slot_t = [batch], [db], [t_fixed]
    for sl in range(1, n_slots if args.distinct_slots else 1):
        bs = pr.random_batch(args.paths, args.segments, seed0=(rank * n_slots + sl) * args.paths)
        dbs = api.DeviceBatch(bs, dev, sample_capacity=0)
        plan.solve(est, dbs.fixed_mask, dbs.fixed_values, dbs.seg_times, dbs.coeffs, dbs.status, dbs.cost,
                   waypoints=dbs.waypoints, limits=dbs.limits)
        torch.cuda.synchronize()
        slot_batch.append(bs)
        slot_db.append(dbs)
        slot_t.append(dbs.seg_times.clone())

    def slot_inputs(slot):
        i = slot if slot < len(slot_db) else 0
        return slot_db[i].fixed_mask, slot_db[i].fixed_values, slot_t[i], slot_db[i].waypoints

    bound = {}

    def slot_call(kind, slot, lane):
        sh = 1 if active_lanes[0] > 1 else 0
        key = (kind, slot, lane, sh)
        if key not in bound:
            if kind in ("linear", "blocks"):
                s_mask, s_vals, s_times, s_wp = slot_inputs(slot)
                bound[key] = lane_plan[lane].bind_solve(opt_lin[sh] if kind == "linear" else opt_blocks[sh], s_mask, s_vals,
                                                        s_times, out_coeffs[slot], status_i32[slot], slot_cost[slot],
                                                        waypoints=s_wp)
            else:
                bound[key] = lane_plan[lane].bind_solve(opt_nl[sh], db.fixed_mask, db.fixed_values, out_times[slot], out_coeffs[slot],
                                                        status_i32[slot], slot_cost[slot], waypoints=db.waypoints,
                                                        limits=db.limits, n_samples=slot_nsamp[slot], samples=slot_samples[slot])
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
            gather(packed[slot], recv[slot])
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

    def make_linear_step(kind):
        def step():
            slot, lane = begin_step()
            slot_call(kind, slot, lane)()
            last_slot[0] = (slot, lane)
            if gather_every[0]:
                finish_step(slot, lane)
        return step

    round_robin = {}

    grouped_mode = [args.issue == "grouped"]

    def make_linear_block(kind):
        """n steps of a fixed-times workload through the library's own issue loop: step k runs slot k % slots.  streams:
        on lane slot % lanes, one dispatch per step (mrs_tg_bound_solve_launch_many).  grouped (the default solve only): all
        slots are bound to lane 0's plan and the steps of a round -- one per batch in flight -- go out as ONE dispatch
        (mrs_tg_bound_solve_launch_group).  Not for --gather every (a collective follows every step there)."""
        def block(n):
            lanes = active_lanes[0]
            grouped = grouped_mode[0] and kind == "linear" and lanes > 1 and n_group_slots > 1
            slots = n_group_slots if grouped else max(lanes, 2 if dist is not None else 1)
            for slot in range(slots):       # a slot whose last result is still being gathered: its lane waits for that
                if slot_free[slot] is not None:
                    lane_stream[group_lanes[min(slot // max(1, min(args.group_size, slots)), lanes - 1)] if grouped else slot % lanes].wait_event(slot_free[slot])
                    slot_free[slot] = None
            key = (kind, lanes, slots, grouped, tuple(group_lanes[:lanes]) if grouped else None)
            if key not in round_robin:
                if grouped:   # slot sl on the lane of its group (sl // group size): consecutive slots of a lane go out as one dispatch
                    gs = max(1, min(args.group_size, slots))
                    round_robin[key] = api.RoundRobin([slot_call(kind, sl, group_lanes[min(sl // gs, lanes - 1)]) for sl in range(slots)], grouped=True)
                else:
                    round_robin[key] = api.RoundRobin([slot_call(kind, sl, sl % lanes) for sl in range(slots)],
                                                      threads=min(args.issue_threads, slots))
            round_robin[key](n)
            step_no[0] = n
            last_slot[0] = ((n - 1) % slots, group_lanes[min(((n - 1) % slots) // max(1, min(args.group_size, slots)), lanes - 1)] if grouped
                            else ((n - 1) % slots) % lanes)
        return block

    def step_nonlinear():
        slot, lane = begin_step()
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

    def verify_gather():
        """rank 0: what the closing gather delivered for rank 0 is bit for bit what rank 0 computed, and every peer's buffer
        holds finite coefficients and status words of the library (not stale memory)"""
        if rank != 0 or gloo:
            return None
        torch.cuda.synchronize()
        slot = last_slot[0][0]
        mine = bool(torch.equal(recv[slot][0], packed[slot]))
        peers = all(bool(torch.isfinite(r).all()) and bool((r[nS * 41:].abs() <= 6).all()) for r in recv[slot])
        return dict(root_equals_local=mine, peers_finite_with_valid_status=peers, ranks=len(recv[slot]),
                    bytes_per_rank=int(packed[slot].numel() * 8))

    steps_fn = {"linear": make_linear_step("linear"), "blocks": make_linear_step("blocks"), "nonlinear": step_nonlinear}
    blocks_fn = {"linear": make_linear_block("linear"), "blocks": make_linear_block("blocks")}

    def block_for(kind):
        return None if (gather_every[0] or kind not in blocks_fn) else blocks_fn[kind]

    # The clocks of an idle MI355X take a few milliseconds of work to come up, and the driver's default run is 5 warm-up
    # + 20 timed steps of ~10 us: an untimed ramp of the same step keeps the timed region from measuring the ramp.
    # (by time, not by count: 300 steps of the current kernels are 1 ms; the ramp runs for 30 ms)
    ramp_steps = 0
    t_ramp = time.perf_counter()
    ramp_block = block_for(args.workload)
    while time.perf_counter() - t_ramp < 0.030:
        if ramp_block is not None:   # the timed region's own issue path: the same dispatch shapes (a kernel's first launch
            ramp_block(96)           # ever costs hundreds of microseconds of module set-up, which is not a step either)
        else:
            for _ in range(96):
                steps_fn[args.workload]()
        ramp_steps += 96
        torch.cuda.synchronize()
    step_no[0] = 0

    # Which two streams?  The runtime maps HIP streams onto a handful of hardware queues, and which pairs share one differs from
    # process to process: on one box torch's current stream and the first side stream -- the pair of rounds 4-6 -- took 67 us for
    # the two dispatches of a 20-step region in every process, other pairs 49, 60 or 66 us (one stream twice: 55;
    # profiles/round6_stream_pairs.txt).  A host that keeps two dispatches in flight picks its streams: the pair with the
    # shortest two-dispatch round of the timed region's own issue path, measured here, untimed, once.
    stream_pairs = None

    def calibrate_group_lanes():
        gs = max(1, min(args.group_size, n_group_slots))
        blk = block_for("linear")
        if blk is None or n_cand <= 2 or args.stream_candidates <= 2 or n_group_slots < 2 * gs:
            return None
        def round_us(a_, b_):
            group_lanes[:] = [a_, b_] + [l for l in range(n_cand) if l not in (a_, b_)]
            for _ in range(3):
                blk(2 * gs)
            torch.cuda.synchronize()
            ts = []
            for _ in range(9):
                t0_ = time.perf_counter()
                blk(2 * gs)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0_)
            return median_of(ts) * 1e6

        best, table = pick_stream_pair(n_cand, round_us)
        group_lanes[:] = [best[0], best[1]] + [l for l in range(n_cand) if l not in best]
        step_no[0] = 0
        return dict(candidates=n_cand, chosen=list(best), two_dispatch_round_us={"%d,%d" % k: round(v, 1) for k, v in sorted(table.items())},
                    lane_0="torch's current stream", note="lanes 1.. are torch side streams in creation order; rounds 4-6 used the pair 0,1")

    if args.workload == "linear" and grouped_mode[0] and not gather_every[0]:
        stream_pairs = calibrate_group_lanes()

    # R regions of K steps each, every one between its own barrier + synchronize pairs; the line's figure is the MEDIAN region
    # (one 20-step region of the fixed-times workload is 60-80 us: a single one measures the moment, not the kernel)
    region_s, gather_s = time_regions(steps_fn[args.workload], args.steps, args.warmup, dist, torch, max(1, args.regions),
                                      final_gather, block_fn=block_for(args.workload))
    total_paths = P * world * args.steps
    elapsed = median_of(region_s)
    value = total_paths / elapsed
    own_regions_main = list(LAST_OWN_REGIONS[0])   # this rank's own regions (later measurements overwrite the global)
    own_elapsed_main = median_of(own_regions_main)
    # what every rank saw, so that a first multi-GPU curve can be read from the line alone: its own K-step time, the device it
    # ran on, and whether the dmabuf IPC mode the pool's driver needs was set in its environment
    props = torch.cuda.get_device_properties(dev)
    rank_info = dict(rank=rank, local_rank=local_rank, device_index=dev_index, device=props.name,
                     arch=getattr(props, "gcnArchName", None), ms_per_step=own_elapsed_main / args.steps * 1e3,
                     gather_ms=(LAST_OWN_GATHER[0] * 1e3 if dist is not None else None),
                     hsa_enable_ipc_mode_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    per_rank = [rank_info]
    if dist is not None:
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, rank_info)
    gather_check = verify_gather() if dist is not None else None

    # ---- the same issue path over 200 steps (the driver's run is 20): kept as a top-level key so that rounds stay comparable
    value_200 = None
    if args.steps != 200 and args.workload == "linear" and not args.no_extras:
        el200, _ = time_steps(steps_fn[args.workload], 200, 0, dist, torch, None, block_fn=block_for(args.workload))
        value_200 = P * world * 200 / el200

    # ---- roofline of the kernel the timed region RUNS (rank 0's device): the dispatches of the headline's own issue path,
    # each with its own pair of events (the library times the launches of kernel family 1 on the lane's context) ----
    roofline_headline = None
    if args.workload == "linear" and block_for("linear") is not None:
        group_n = max(1, min(args.group_size, n_group_slots)) if (grouped_mode[0] and n_lanes > 1 and n_group_slots > 1) else 1
        api.kernel_trace_reset()
        block_for("linear")(n_group_slots if group_n > 1 else n_lanes)          # one round: what kernels does a dispatch run?
        traced = api.kernel_trace()
        torch.cuda.synchronize()
        head_kernel = traced[-1] if traced else "unknown"
        head_base = head_kernel.split("<")[0]   # (the counters under profiles/ carry the kernel's name without template arguments)
        for c in lane_ctx:
            c.set_profiling(True)
        try:
            block_for("linear")(max(1000, 50 * n_group_slots))
            torch.cuda.synchronize()
            hv = []
            for c in lane_ctx:
                hv += c.kernel_ms_history(api.KERNEL_SOLVE_LINEAR, 512)
        finally:
            for c in lane_ctx:
                c.set_profiling(False)
        hv.sort()
        h_mean = sum(hv) / max(len(hv), 1)
        paths_disp = P * group_n
        comp_bytes = paths_disp * (40 * args.segments + 288 + 328 * args.segments)
        ach = comp_bytes / (h_mean * 1e-3) / 1e9
        # over the timed region itself: compulsory bytes of all K steps / the region's wall time (dispatches overlap on two streams)
        ach_region = (P * args.steps * (40 * args.segments + 288 + 328 * args.segments)) / own_elapsed_main / 1e9
        roofline_headline = dict(
            kernel=head_kernel, kernels_of_one_round=traced, in_timed_region=True, bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS,
            paths_per_dispatch=paths_disp, steps_per_dispatch=group_n,
            compulsory_bytes_per_dispatch=comp_bytes,
            bytes_model="SURVEY.md 8d: (40 S + 288) B in + 328 S B out per path = %d B at S = %d; nothing is materialised"
                        % (40 * args.segments + 288 + 328 * args.segments, args.segments),
            avg_dispatch_us=h_mean * 1e3, median_dispatch_us=(hv[len(hv) // 2] * 1e3 if hv else None), dispatches_timed=len(hv),
            achieved=ach, frac=ach / HBM_PEAK_GBS,
            achieved_over_timed_region=ach_region, frac_over_timed_region=ach_region / HBM_PEAK_GBS,
            timing="per dispatch: a pair of events attached to each launch of the headline's own issue path (two streams, so two "
                   "dispatches overlap as in the timed region); *_over_timed_region: all K steps' compulsory bytes / the timed "
                   "region's wall time",
            traffic=None, traffic_over_compulsory=None, valu_issue_frac=None)
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_solve_*_group_hbm_traffic.json"))):
            try:
                with open(f) as fh:
                    d = json.load(fh)
            except (OSError, ValueError):
                continue
            if d.get("paths_per_dispatch") == paths_disp and d.get("segments") == args.segments and d.get("kernel") == head_base:
                roofline_headline.update(traffic=d["hbm_bytes_per_dispatch"],
                                         traffic_over_compulsory=d["hbm_bytes_per_dispatch"] / float(comp_bytes),
                                         traffic_source=os.path.relpath(f, ROOT) + " (separate rocprofv3 --pmc passes; from "
                                                                                   "profiles/, not this run)")
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_sq_solve_*_group.json"))):
            try:
                with open(f) as fh:
                    d = json.load(fh)
            except (OSError, ValueError):
                continue
            if d.get("paths") == paths_disp and d.get("segments") == args.segments and head_base in d.get("kernel", ""):
                c = d["counters"]
                issue_peak_h = 256 * 4 * 2.4e9 / 4.0
                roofline_headline.update(
                    valu_issue_frac=c["SQ_INSTS_VALU"] / (h_mean * 1e-3) / issue_peak_h,
                    valu_instructions_per_dispatch=c["SQ_INSTS_VALU"], lds_instructions_per_dispatch=c.get("SQ_INSTS_LDS"),
                    wait_share_of_wave_cycles=(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None),
                    valu_active_share_of_wave_cycles=(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None),
                    counters_source=d["source"] + " (separate rocprofv3 --pmc passes; from profiles/, not this run)")

    # ---- roofline of the assembly kernel (rank 0's device) ----
    Hbuf = torch.empty(plan.block_doubles, dtype=torch.float64, device=dev)
    Abuf = torch.empty(plan.block_doubles, dtype=torch.float64, device=dev)

    def launch_assemble():
        plan.assemble(4, t_init, Hbuf, Abuf)

    for _ in range(20):
        launch_assemble()
    torch.cuda.synchronize()
    q_mean, q_med, q_min = dispatch_stats(ctx, api.KERNEL_ASSEMBLE, launch_assemble, 300, torch)     # queued: neighbours overlap
    asm_mean, asm_med, asm_min = isolated_dispatch_stats(ctx, api.KERNEL_ASSEMBLE, launch_assemble, 300, torch)
    asm_b2b = back_to_back_ms(launch_assemble, 200, torch)
    alg_bytes = ASSEMBLY_BYTES_PER_SEGMENT * nS
    achieved = alg_bytes / (asm_mean * 1e-3) / 1e9
    traffic, traffic_source = None, "no PMC summary for this batch shape under profiles/"
    tm = measured_traffic(P, args.segments)
    if tm is not None:
        traffic = tm[0]["hbm_bytes_per_launch"]
        traffic_source = "%s (separate rocprofv3 --pmc passes; from profiles/, not this run)" % tm[1]
    roofline = dict(kernel="assemble_blocks_uniform_kernel", in_timed_region=False,
                    scope="SURVEY.md 8d's contract kernel (the HBM-bound Hessian / mapping-block assembly behind mrs_tg_plan_assemble and "
                          "MRS_TG_FLAG_MATERIALIZED_BLOCKS), measured on its own AFTER the timed region: the headline's default solve "
                          "forms the blocks in registers and never launches it -- the kernel the timed steps run is roofline_headline",
                    bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_source, bytes_per_launch=alg_bytes,
                    avg_launch_us=asm_mean * 1e3, median_launch_us=asm_med * 1e3, min_launch_us=asm_min * 1e3,
                    timing="per dispatch: a pair of events attached to each of 300 launches, every launch issued when the previous "
                           "one has finished (round 6; the dispatch's own duration, which is also what rocprofv3 --kernel-trace "
                           "reports in its run of the command: the tool's work per launch keeps the dispatches apart).  Until round 5 "
                           "the 300 launches were QUEUED: the next dispatch's ramp then runs into this one's tail and each pair of "
                           "events spans more than its own dispatch -- 5.2 to 6.3 us from run to run where isolated dispatches take "
                           "5.0 to 5.2; those figures are queued_*",
                    queued_avg_launch_us=q_mean * 1e3, queued_median_launch_us=q_med * 1e3,
                    queued_frac=alg_bytes / (q_mean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    back_to_back_interval_us=asm_b2b * 1e3,
                    back_to_back_note="launch-to-launch interval of 200 queued launches; one dispatch's tail overlaps the "
                                      "next one's ramp, so this is NOT a kernel duration (%.3f of peak if it were)"
                                      % (alg_bytes / (asm_b2b * 1e-3) / 1e9 / HBM_PEAK_GBS),
                    note="16 MB per launch: launch-ramp bound, see extras.roofline_large for the same kernel at 1 GB")
    ceil = fill_ceiling(P, args.segments)
    if ceil is not None:
        ceil["frac"] = alg_bytes / (ceil["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        ceil["note"] = ("fastest pure fill of the same bytes, per dispatch, from profiles/ (not this run): the ceiling of this "
                        "launch size")
        roofline["fill_ceiling"] = ceil

    # ---- FP64 rooflines of the solve kernel and of the outer-loop kernel (per dispatch, flop model of SURVEY.md 8d) ----
    db.coeffs, db.seg_times, db.status = out_coeffs[0], out_times[0], status_i32[0]

    def launch_solve():
        plan.solve(opt_lin[0], db.fixed_mask, db.fixed_values, t_fixed, out_coeffs[0], status_i32[0], slot_cost[0], waypoints=db.waypoints)

    def launch_nl():
        plan.solve(opt_nl[0], db.fixed_mask, db.fixed_values, out_times[0], out_coeffs[0], status_i32[0], slot_cost[0],
                   waypoints=db.waypoints, limits=db.limits, n_samples=slot_nsamp[0], samples=slot_samples[0])

    for _ in range(10):
        launch_solve()
    torch.cuda.synchronize()
    sol_mean, sol_med, sol_min = dispatch_stats(ctx, api.KERNEL_SOLVE_LINEAR, launch_solve, 200, torch)
    solve_flop = SOLVE_FLOP_PER_SEGMENT * nS
    def newest_sq(pattern):
        """SQ counters of a kernel at this batch shape from the newest committed PMC passes (scripts/pmc_sq.sh, pmc_sq_json.py)"""
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
            try:
                with open(f) as fh:
                    d = json.load(fh)
            except (OSError, ValueError):
                continue
            if d.get("paths") == P and d.get("segments") == args.segments:
                return d
        return None

    sq = newest_sq("round*_pmc_sq_solve_rows.json")
    issue_peak = 256 * 4 * 2.4e9 / 4.0   # wavefront VALU instructions per second: one per 4 cycles per SIMD
    counted = None
    if sq is not None:
        c = sq["counters"]
        counted = dict(source=sq["source"] + " (separate rocprofv3 --pmc passes; from profiles/, not this run)",
                       valu_instructions_per_launch=c["SQ_INSTS_VALU"], lds_instructions_per_launch=c["SQ_INSTS_LDS"],
                       valu_issue_rate_ginst_s=c["SQ_INSTS_VALU"] / (sol_mean * 1e-3) / 1e9,
                       valu_issue_peak_ginst_s=issue_peak / 1e9,
                       valu_issue_frac=c["SQ_INSTS_VALU"] / (sol_mean * 1e-3) / issue_peak,
                       executed_fp64_tflops_upper_bound=2.0 * 64 * c["SQ_INSTS_VALU"] / (sol_mean * 1e-3) / 1e12,
                       wait_share_of_wave_cycles=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                       valu_active_share_of_wave_cycles=c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"])
    roofline_solve = dict(kernel="solve_rows_kernel", bound="fp64 vector", unit="TFLOP/s", peak=FP64_VECTOR_PEAK_TFLOPS,
                          counted=counted,
                          flop_per_launch=solve_flop, avg_launch_us=sol_mean * 1e3, median_launch_us=sol_med * 1e3,
                          achieved=solve_flop / (sol_mean * 1e-3) / 1e12,
                          frac=solve_flop / (sol_mean * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                          note="flop model of SURVEY.md 8d (6e3 flop per segment, which counts the reference's two dense "
                               "10^3 products per segment; the kernel forms the blocks from exact constants and executes "
                               "roughly a quarter of that); the kernel is bound by one dependent chain per path, not by "
                               "FP64 issue: see DESIGN.md")
    for _ in range(3):
        launch_nl()
    torch.cuda.synchronize()
    nl_mean, nl_med, nl_min = dispatch_stats(ctx, api.KERNEL_NONLINEAR, launch_nl, 50, torch)
    nl_flop = NONLINEAR_FLOP_PER_PATH_S10 * (args.segments / 10.0) ** 2 * P
    sq_nl = newest_sq("round*_pmc_sq_outer_loop.json")
    if sq_nl is not None:
        # executed work: every VALU instruction counted as a 64-lane FP64 FMA -- an upper bound of the flops the kernel ran
        c = sq_nl["counters"]
        nl_exec = 2.0 * 64 * c["SQ_INSTS_VALU"]
        roofline_outer = dict(kernel=sq_nl["kernel"], bound="fp64 vector issue", unit="wavefront VALU instructions/s",
                              peak=issue_peak, avg_launch_us=nl_mean * 1e3, median_launch_us=nl_med * 1e3,
                              achieved=c["SQ_INSTS_VALU"] / (nl_mean * 1e-3),
                              valu_issue_frac=c["SQ_INSTS_VALU"] / (nl_mean * 1e-3) / issue_peak,
                              valu_lane_flop_upper_bound_per_launch=nl_exec,
                              counted=dict(source=sq_nl["source"] + " (separate rocprofv3 --pmc passes; from profiles/, not this run)",
                                           valu_instructions_per_launch=c["SQ_INSTS_VALU"],
                                           lds_instructions_per_launch=c["SQ_INSTS_LDS"],
                                           valu_issue_frac=c["SQ_INSTS_VALU"] / (nl_mean * 1e-3) / issue_peak,
                                           wait_share_of_wave_cycles=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                           valu_active_share_of_wave_cycles=c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                                           lds_bank_conflict_share=(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
                                                                    if c.get("SQ_LDS_IDX_ACTIVE") else None)),
                              reference_work_flop=nl_flop,
                              note="an ISSUE rate, not a flop rate: counted VALU instructions per second against one per 4 cycles per SIMD "
                                   "(valu_lane_flop_upper_bound_per_launch = 2 x 64 x that count, if every instruction were a full FP64 FMA); "
                                   "reference_work_flop is SURVEY.md 8d's model of what the reference would execute for the same "
                                   "batch (7e6 flop per 10-segment path = 121 linear solves), which the kernel undercuts by running "
                                   "forward-only cost sweeps on exact constants and stopping on ftol / xtol -- not a rate")
    else:
        roofline_outer = dict(kernel="optimize_wave_kernel / optimize_split_kernel / optimize_lean_kernel", bound="fp64 vector issue",
                              unit="wavefront VALU instructions/s", peak=issue_peak, avg_launch_us=nl_mean * 1e3,
                              median_launch_us=nl_med * 1e3, achieved=None, valu_issue_frac=None, reference_work_flop=nl_flop,
                              note="no committed SQ counters for this batch shape (scripts/pmc_sq.sh); reference_work_flop is "
                                   "SURVEY.md 8d's model of the reference's work, not what the kernel executes")

    extras = {}
    if not args.no_extras and rank == 0:
        # the same kernel on a batch that is not launch-ramp dominated (65536 x 10 segments = 1.05 GB per launch)
        big_P = 65536
        so_big = (np.arange(big_P + 1, dtype=np.int64) * args.segments).astype(np.int32)
        plan_big = api.Plan(ctx, so_big)
        t_big = t_init.repeat((big_P * args.segments + nS - 1) // nS)[:big_P * args.segments].contiguous()
        assert big_P == BIG_P and early_big is not None and early_big[0].numel() == plan_big.block_doubles
        bytes_big = ASSEMBLY_BYTES_PER_SEGMENT * big_P * args.segments

        def large_stats(Hx, Ax):
            for _ in range(3):
                plan_big.assemble(4, t_big, Hx, Ax)
            torch.cuda.synchronize()
            return dispatch_stats(ctx, api.KERNEL_ASSEMBLE, lambda: plan_big.assemble(4, t_big, Hx, Ax), 20, torch)

        # Where a buffer lands decides the rate of a 1 GB streaming write: 0.62 .. 0.88 of the HBM peak from buffer to buffer,
        # stable for as long as the buffer lives (profiles/round6_assembly_placement.txt: not warm-up, not the clocks, not the skew
        # between the two outputs) -- the "unexplained" 0.64-0.81 from run to run of rounds 4-5, which measured ONE pair.  Five pairs
        # here -- the one allocated when the process started and four allocated now -- and the MEDIAN pair is the figure.
        pairs_big = [early_big] + [(torch.empty(plan_big.block_doubles, dtype=torch.float64, device=dev),
                                    torch.empty(plan_big.block_doubles, dtype=torch.float64, device=dev)) for _ in range(4)]
        stats_big = [large_stats(Hx, Ax) for (Hx, Ax) in pairs_big]
        order = sorted(range(len(stats_big)), key=lambda k: stats_big[k][1])
        m_big, med_big, min_big = stats_big[order[len(order) // 2]]
        extras["roofline_large"] = dict(kernel="assemble_blocks_uniform_kernel", paths=big_P, bytes_per_launch=bytes_big,
                                        avg_launch_us=m_big * 1e3, median_launch_us=med_big * 1e3, min_launch_us=min_big * 1e3,
                                        achieved=bytes_big / (med_big * 1e-3) / 1e9,
                                        unit="GB/s", frac=bytes_big / (med_big * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        frac_of_min_launch=bytes_big / (min_big * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        frac_of_mean_launch=bytes_big / (m_big * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        timing="per dispatch, 20 queued launches per pair of output buffers; achieved / frac from the "
                                               "MEDIAN launch of the median pair of five",
                                        frac_by_buffer_pair=[bytes_big / (st[1] * 1e-3) / 1e9 / HBM_PEAK_GBS for st in stats_big],
                                        placement="frac_by_buffer_pair: [the pair allocated when the process started, four pairs allocated "
                                                  "here].  Where a buffer lands decides the rate of a 1 GB streaming write -- 0.62 to 0.88 of "
                                                  "the peak from buffer to buffer, stable per buffer (profiles/round6_assembly_placement.txt); "
                                                  "rounds 1-5 measured whichever their one pair happened to get (0.64-0.81 from run to run)")
        del pairs_big
        early_big = None
        plan_big.close()
        # PCIe-inclusive rate of the one-call host interface (H2D + kernels + D2H; the plan and the context's transfer arenas
        # are kept after the first call).  Three callers: numpy arrays allocated per call (what api.solve_batch does by
        # default: every call faults in 3.4 MB of fresh output pages), the same pageable arrays re-used, and arrays in
        # pinned memory (mrs_tg_host_alloc), which the DMA engines read / write in place.
        times_host = t_init.cpu().numpy()

        def host_rate(fn, reps=30):
            fn()
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            return reps * P / (time.perf_counter() - t0)

        fresh = host_rate(lambda: ctx.solve_batch(batch, times_host))
        keep = ctx.solve_batch(batch, times_host)
        reused = host_rate(lambda: ctx.solve_batch(batch, times_host, out=keep))
        pb = pr.Batch(batch.seg_offsets, api.pinned_copy(batch.waypoints), api.pinned_copy(batch.fixed_mask),
                      api.pinned_copy(batch.fixed_values), api.pinned_copy(batch.limits), batch.derivative_to_optimize)
        pin_out = dict(times=api.pinned_copy(times_host), coeffs=api.pinned_empty(keep["coeffs"].shape),
                       status=api.pinned_empty(P, np.int32), cost=api.pinned_empty(P), n_samples=None, samples=None)
        pinned = host_rate(lambda: ctx.solve_batch(pb, times_host, out=pin_out))
        same = bool(np.array_equal(pin_out["coeffs"], keep["coeffs"]) and np.array_equal(pin_out["status"], keep["status"]))
        bytes_in = int(batch.fixed_mask.nbytes + batch.fixed_values.nbytes + times_host.nbytes)
        bytes_out = int(keep["coeffs"].nbytes + times_host.nbytes + keep["status"].nbytes + keep["cost"].nbytes)
        extras["host_buffer_call"] = dict(value=pinned, unit="trajectories/s", caller="arrays in pinned host memory (mrs_tg_host_alloc), re-used",
                                          pageable_reused=reused, pageable_allocated_per_call=fresh,
                                          us_per_call=dict(pinned=P / pinned * 1e6, pageable_reused=P / reused * 1e6,
                                                           pageable_allocated_per_call=P / fresh * 1e6),
                                          bytes_host_to_device=bytes_in, bytes_device_to_host=bytes_out,
                                          pinned_equals_pageable=same,
                                          note="mrs_tg_solve_batch with host buffers, linear QP, PCIe copies included; never the headline")
        # one request, findTrajectory()'s own signature (mrs_tg_find_trajectory: one 10-segment path, Mellinger outer loop,
        # feasibility scaling, sampling dt 0.2): the latency a drop-in nodelet sees per service call
        wp1 = pr.random_box_waypoints(args.segments, 12345)
        for _ in range(3):
            ctx.find_trajectory(wp1, sample_capacity=1024)
        lat_py = []
        for _ in range(30):
            t0 = time.perf_counter()
            ctx.find_trajectory(wp1, sample_capacity=1024)
            lat_py.append(time.perf_counter() - t0)
        lat_py.sort()
        call1, _ = ctx.bind_find_trajectory(wp1, sample_capacity=1024)   # the ctypes arguments marshalled once
        for _ in range(3):
            call1()
        lat = []
        for _ in range(100):
            t0 = time.perf_counter()
            call1()
            lat.append(time.perf_counter() - t0)
        lat.sort()
        extras["single_request_latency"] = dict(median_us=lat[len(lat) // 2] * 1e6, min_us=lat[0] * 1e6, max_us=lat[-1] * 1e6,
                                                python_wrapper_median_us=lat_py[len(lat_py) // 2] * 1e6,
                                                call="mrs_tg_find_trajectory through ctypes, one %d-segment path, Mellinger + "
                                                     "scaling + sampling dt 0.2, host buffers in and out; median_us: the foreign "
                                                     "call with its arguments marshalled once (what a C++ host pays, cf. "
                                                     "examples/request_latency_host.cpp); python_wrapper_median_us: "
                                                     "Context.find_trajectory, which builds the ctypes structures and output arrays "
                                                     "per call (the figure of rounds 1-3)" % args.segments)
    if not args.no_extras and args.workload == "linear":
        # the step that materialises the blocks: assembly kernel + solve from the blocks in HBM
        elb, _ = time_steps(steps_fn["blocks"], args.steps, 3, dist, torch, final_gather, block_fn=block_for("blocks"))
        if rank == 0:
            extras["materialized_blocks_step"] = dict(value=P * world * args.steps / elb, unit="trajectories/s",
                                                      ms_per_step=elb / args.steps * 1e3,
                                                      note="MRS_TG_FLAG_MATERIALIZED_BLOCKS: assemble_blocks kernel (16.5 MB to "
                                                           "HBM) + solve_tile_kernel reading the blocks back; %d in flight" % n_lanes)
    if n_lanes > 1 and not args.no_extras and args.workload == "linear" and grouped_mode[0]:
        # the same K steps issued one dispatch per step, round-robin over the streams (the headline's method up to round 2)
        grouped_mode[0] = False
        els, _ = time_steps(steps_fn["linear"], args.steps, 3, dist, torch, final_gather, block_fn=block_for("linear"))
        grouped_mode[0] = True
        if rank == 0:
            extras["streams_in_flight"] = dict(value=P * world * args.steps / els, unit="trajectories/s", ms_per_step=els / args.steps * 1e3,
                                               note="one dispatch per step on %d HIP streams (mrs_tg_bound_solve_launch_many), "
                                                    "MRS_TG_FLAG_SHARED_DEVICE" % n_lanes)
    if n_lanes > 1 and not args.no_extras:
        # the same steps with one batch in flight: every step waits for the previous one (single stream)
        torch.cuda.synchronize()
        active_lanes[0] = 1
        step_no[0] = 0
        el1, _ = time_steps(steps_fn[args.workload], args.steps, 3, dist, torch, final_gather, block_fn=block_for(args.workload))
        el1b = (time_steps(steps_fn["blocks"], args.steps, 3, dist, torch, final_gather, block_fn=block_for("blocks"))[0]
                if args.workload == "linear" else None)
        active_lanes[0] = n_lanes
        step_no[0] = 0
        if rank == 0:
            extras["one_batch_in_flight"] = dict(value=P * world * args.steps / el1, unit="trajectories/s",
                                                 ms_per_step=el1 / args.steps * 1e3)
            if el1b is not None:
                extras["one_batch_in_flight"]["materialized_blocks_ms_per_step"] = el1b / args.steps * 1e3
    if not args.no_extras:
        other = "nonlinear" if args.workload == "linear" else "linear"
        k2 = max(5, args.steps // 10) if other == "nonlinear" else args.steps
        el2, _ = time_steps(steps_fn[other], k2, 3, dist, torch, final_gather, block_fn=block_for(other))
        if rank == 0:
            extras[other] = dict(value=P * world * k2 / el2, unit="trajectories/s", steps=k2, ms_per_step=el2 / k2 * 1e3)
        if dist is not None and not gather_every[0]:
            # the same workload with the results of EVERY step gathered to rank 0 (bound by the xGMI links into the root)
            gather_every[0] = True
            el3, _ = time_steps(steps_fn[args.workload], args.steps, 3, dist, torch)
            gather_every[0] = False
            if rank == 0:
                extras["gather_every_step"] = dict(value=P * world * args.steps / el3, unit="trajectories/s",
                                                   ms_per_step=el3 / args.steps * 1e3,
                                                   bytes_per_rank_per_step=int(packed[0].numel() * 8))

    # ---- BASELINE configs[3]: ONE batch of 65536 nonlinear-time paths, contiguous shards over the ranks, gather to rank 0
    if not args.no_extras:
        total3 = args.config3_paths
        a3, b3 = shard.contiguous_shard(total3, rank, world)
        n3 = b3 - a3
        # path p of the batch is seeded with p, whatever the number of ranks: the same 65536 paths at every N
        batch3 = pr.random_batch(n3, args.segments, seed0=a3)
        plan3 = api.Plan(ctx, batch3.seg_offsets)
        db3 = api.DeviceBatch(batch3, dev, sample_capacity=512)
        plan3.solve(est, db3.fixed_mask, db3.fixed_values, db3.seg_times, db3.coeffs, db3.status, db3.cost,
                    waypoints=db3.waypoints, limits=db3.limits)
        torch.cuda.synchronize()
        t3 = db3.seg_times.clone()
        nS3 = batch3.n_segments
        pk3 = torch.zeros(shard.packed_doubles(n3, nS3), dtype=torch.float64, device=dev)
        c3, tt3, st3_f64 = shard.packed_views(pk3, n3, nS3)
        st3 = torch.zeros(n3, dtype=torch.int32, device=dev)
        # equal shards gather in one collective; an uneven cut pads to the largest shard (shard.gather_packed_shards, the code
        # tests/test_dist_gloo.py runs on four gloo ranks with 65535 paths)
        cap3 = shard.shard_capacity(total3, world)
        pad3 = (torch.zeros(shard.packed_doubles(cap3, cap3 * args.segments), dtype=torch.float64, device=dev)
                if dist is not None else None)
        recv3 = ([torch.empty_like(pad3) for _ in range(world)] if (dist is not None and rank == 0 and not gloo) else None)
        call3 = plan3.bind_solve(opt_nl[0], db3.fixed_mask, db3.fixed_values, tt3, c3, st3, db3.cost, waypoints=db3.waypoints,
                                 limits=db3.limits, n_samples=db3.n_samples, samples=db3.samples)

        def step3():
            call3()

        def gather3():
            if dist is None:
                return
            st3_f64.copy_(st3)
            shard.gather_packed_shards(pk3, pad3, dist, bufs=recv3, via_host=gloo)

        k3 = max(3, min(10, args.steps // 20))
        el4, _ = time_steps(step3, k3, 2, dist, torch, gather3, gather_inside=True)
        # paths the pipeline did not hand back as successes: ROUNDOFF_LIMITED = the feasibility scaling ran away (mrs_tg.h)
        bad3 = torch.tensor([int((st3 == api.STATUS_ROUNDOFF_LIMITED).sum().item()), int((st3 <= 0).sum().item())], dtype=torch.int64, device=dev)
        if dist is not None:
            if gloo:
                bad3 = bad3.cpu()
            dist.all_reduce(bad3)
        c3_check = None
        if dist is not None and rank == 0 and recv3 is not None:
            torch.cuda.synchronize()
            c3_check = bool(torch.equal(recv3[0][:pk3.numel()], pk3))
        if rank == 0:
            extras["config3"] = dict(workload="BASELINE configs[3]: one batch of %d random %d-segment paths, Mellinger outer loop + "
                                              "feasibility scaling + sampling, contiguous shards over %d rank(s), results gathered "
                                              "to rank 0 after the last step (inside the timed region)" % (total3, args.segments, world),
                                     value=total3 * k3 / el4, unit="trajectories/s", scaling="strong", steps=k3,
                                     ms_per_step=el4 / k3 * 1e3, paths_per_rank=n3, n_gpus=world,
                                     gather_bytes_per_rank=int(pk3.numel() * 8), gather_root_equals_local=c3_check,
                                     runaway_paths=int(bad3[0].item()), paths_not_successful=int(bad3[1].item()))
        # the fixed-times solve of the same shard, per dispatch: the solve kernel of a SATURATED device beside roofline_solve's
        # single small batch (four lanes per path with the factors in LDS once a launch carries >= 6144 paths, mrs_tg_quad.hip)
        if rank == 0:
            for _ in range(2):
                plan3.solve(opt_lin[0], db3.fixed_mask, db3.fixed_values, t3, c3, st3, db3.cost, waypoints=db3.waypoints)
            torch.cuda.synchronize()
            api.kernel_trace_reset()
            plan3.solve(opt_lin[0], db3.fixed_mask, db3.fixed_values, t3, c3, st3, db3.cost, waypoints=db3.waypoints)
            sat_kernel = (api.kernel_trace() or ["unknown"])[-1]   # (what the trace says: quad from 20480 paths, duo below, rows below 6144)
            m_sat, med_sat, _ = dispatch_stats(ctx, api.KERNEL_SOLVE_LINEAR,
                                               lambda: plan3.solve(opt_lin[0], db3.fixed_mask, db3.fixed_values, t3, c3, st3, db3.cost,
                                                                   waypoints=db3.waypoints), 10, torch)
            flop_sat = SOLVE_FLOP_PER_SEGMENT * n3 * args.segments
            extras["roofline_solve_saturated"] = dict(
                kernel=sat_kernel, paths=n3, bound="fp64 vector", unit="TFLOP/s",
                peak=FP64_VECTOR_PEAK_TFLOPS, flop_per_launch=flop_sat, avg_launch_us=m_sat * 1e3, median_launch_us=med_sat * 1e3,
                achieved=flop_sat / (m_sat * 1e-3) / 1e12, frac=flop_sat / (m_sat * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                trajectories_per_s=n3 / (m_sat * 1e-3),
                compulsory_bytes_per_launch=int(n3 * (40 * args.segments + 288 + 328 * args.segments)),
                hbm_frac_of_compulsory_bytes=n3 * (40 * args.segments + 288 + 328 * args.segments) / (m_sat * 1e-3) / 1e9 / HBM_PEAK_GBS,
                note="flop model of SURVEY.md 8d (6e3 flop per segment: it counts the reference's two dense 10^3 products per segment, "
                     "which no kernel here executes), per dispatch with events on the launch; compulsory bytes = SURVEY 8d's "
                     "(40 S + 288) in + 328 S out per path")
            tq = measured_traffic_solve_quad(n3, args.segments) if sat_kernel.startswith("solve_quad_kernel") else None
            if tq is not None:
                extras["roofline_solve_saturated"].update(
                    traffic=tq[0]["hbm_bytes_per_launch"], traffic_source=tq[1] + " (separate rocprofv3 --pmc passes; from profiles/, not this run)",
                    traffic_over_compulsory=tq[0]["hbm_bytes_per_launch"] / float(n3 * (40 * args.segments + 288 + 328 * args.segments)),
                    traffic_note="writes 211 MB = the coefficients; vertex positions from the compact waypoint array "
                                 "(MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS): from the [vertex][5][4] value array the kernel reads 137 MB for "
                                 "45 MB of inputs (ratio 1.34, profiles/round*_pmc_solve_quad_hbm_traffic.json) -- at the same speed")
        plan3.close()
        del db3, pk3, pad3, recv3

    # ---- two figures beside the BASELINE configs (rank 0, after every other measurement so that they disturb none)
    if not args.no_extras and rank == 0 and world == 1:
        # the policy layer around the solver (mrs_tg_optimize_paths = the reference's optimize(): preprocessing, solve, length
        # check, spatial validation, mid-point subdivision rounds) for a batch of requests, host arrays in and out
        try:
            n_req = 1024
            req = [pr.random_box_waypoints(4 + (i % 8), 7000 + i) for i in range(n_req)]
            pol_out = api.optimize_paths(ctx, req, sample_capacity=2048)
            t0 = time.perf_counter()
            for _ in range(3):
                api.optimize_paths(ctx, req, sample_capacity=2048, out=pol_out)   # (the response arrays are kept, as a server's are)
            dt_pol = (time.perf_counter() - t0) / 3
            t0 = time.perf_counter()
            api.optimize_paths(ctx, req, sample_capacity=2048)
            dt_pol_fresh = time.perf_counter() - t0
            extras["policy_layer"] = dict(value=n_req / dt_pol, unit="requests/s", requests=n_req, ms_per_call=dt_pol * 1e3,
                                          value_with_response_arrays_allocated_per_call=n_req / dt_pol_fresh,
                                          succeeded=int(pol_out["success"].sum()), rounds_mean=float(pol_out["iterations"].mean()),
                                          waypoints_out_mean=float(pol_out["n_waypoints"].mean()),
                                          call="mrs_tg_optimize_paths through api.optimize_paths (its numpy marshalling included; the response "
                                               "arrays re-used from call to call -- rounds 1-5 allocated 67 MB of them per call), "
                                               "box-generator requests of 4-11 waypoints, the reference's default policy "
                                               "(min-acceleration, max deviation 0.05 m, up to 6 subdivision rounds)")
        except Exception as exc:   # (an extra: never the reason a bench line is missing)
            extras["policy_layer"] = dict(error=repr(exc))
        # the Mellinger pipeline under the nodelet's default objective (min-acceleration; the BASELINE configs are min-snap)
        try:
            b2 = pr.random_batch(P, args.segments, seed0=0, derivative_to_optimize=2)
            db2 = api.DeviceBatch(b2, dev, sample_capacity=512)
            plan2 = api.Plan(ctx, b2.seg_offsets)
            est2 = api.default_options(derivative_to_optimize=2, estimate_times=1)
            plan2.solve(est2, db2.fixed_mask, db2.fixed_values, db2.seg_times, db2.coeffs, db2.status, db2.cost,
                        waypoints=db2.waypoints, limits=db2.limits)
            torch.cuda.synchronize()
            t_start2 = db2.seg_times.clone()
            opt2 = api.default_options(derivative_to_optimize=2, time_alloc_method=api.TIME_ALLOC_MELLINGER, sampling_dt=0.2,
                                       sample_capacity=512)

            def step2():
                db2.seg_times.copy_(t_start2)
                plan2.solve(opt2, db2.fixed_mask, db2.fixed_values, db2.seg_times, db2.coeffs, db2.status, db2.cost,
                            waypoints=db2.waypoints, limits=db2.limits, n_samples=db2.n_samples, samples=db2.samples)
            for _ in range(3):
                step2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                step2()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / 10
            extras["nonlinear_min_acceleration"] = dict(value=P / dt2, unit="trajectories/s", ms_per_step=dt2 * 1e3, steps=10,
                                                        workload="%d x %d segments, derivative_to_optimize = 2 (the nodelet's default "
                                                                 "config), Mellinger outer loop + feasibility scaling + sampling, one "
                                                                 "batch in flight" % (P, args.segments))
            plan2.close()
        except Exception as exc:
            extras["nonlinear_min_acceleration"] = dict(error=repr(exc))
    # ---- parity of this very batch against the oracle (max-coeff err vs CPU ref) + CPU baseline ----
    cpu = None
    err = None
    err_exact = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # (the contract: rank 0, at N = 1 only)
        from oracle import pyoracle as po
        active_lanes[0] = n_lanes
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
        gpu_c = db.coeffs.cpu().numpy()[:sub.n_segments]
        worst = 0.0
        for p in range(sub.n_paths):
            a, b = sub.seg_offsets[p], sub.seg_offsets[p + 1]
            worst = max(worst, float(np.max(np.abs(gpu_c[a:b] - ref["coeffs"][a:b])) / np.max(np.abs(ref["coeffs"][a:b]))))
        err = worst
        # every batch in flight against the oracle: each slot solved ITS OWN batch (seeds (rank * slots + slot) * paths + p)
        slots_checked = None
        if len(slot_db) > 1:
            slots_run = n_group_slots if (grouped_mode[0] and n_lanes > 1 and n_group_slots > 1) else n_lanes
            # one more round through the headline's own issue path (the extras above have used the slots for other workloads
            # since the timed region), from zeroed outputs: what is compared is what that round wrote
            for sl in range(min(slots_run, len(slot_db))):
                out_coeffs[sl].zero_()
                status_i32[sl].zero_()
            torch.cuda.synchronize()
            step_no[0] = 0
            if block_for("linear") is not None:
                block_for("linear")(slots_run)
            else:
                for _ in range(slots_run):
                    steps_fn["linear"]()
            torch.cuda.synchronize()
            worst_slot, worst_q, all_q = [], 0.0, []
            for sl in range(min(slots_run, len(slot_db))):
                bs = slot_batch[sl]
                ts = slot_t[sl].cpu().numpy()
                rs = po.solve_batch(bs.seg_offsets, bs.waypoints, bs.fixed_mask, bs.fixed_values, bs.limits, ts,
                                    deriv=4, n_threads=cores_for_checks())
                # ... and against the oracle's 113-bit route: the HIP path's own error on every path the timed region solves
                po.lib().mto_set_arithmetic(po.QUAD_PRECISION)
                try:
                    rq = po.solve_batch(bs.seg_offsets, bs.waypoints, bs.fixed_mask, bs.fixed_values, bs.limits, ts, deriv=4,
                                        n_threads=cores_for_checks())
                finally:
                    po.lib().mto_set_arithmetic(po.REFERENCE_ARITHMETIC)
                gc = out_coeffs[sl].cpu().numpy()
                st_ok = bool((status_i32[sl].cpu().numpy() == 1).all())
                w = max(float(np.max(np.abs(gc[a:b] - rs["coeffs"][a:b])) / np.max(np.abs(rs["coeffs"][a:b])))
                        for a, b in zip(bs.seg_offsets[:-1], bs.seg_offsets[1:]))
                eqs = [float(np.max(np.abs(gc[a:b] - rq["coeffs"][a:b])) / np.max(np.abs(rq["coeffs"][a:b])))
                       for a, b in zip(bs.seg_offsets[:-1], bs.seg_offsets[1:])]
                all_q += eqs
                worst_slot.append((w, st_ok))
            slots_checked = dict(slots=len(worst_slot), max_coeff_err_vs_cpu_ref=max(w for w, _ in worst_slot),
                                 max_coeff_err_vs_113bit_ref=float(np.max(all_q)), median_coeff_err_vs_113bit_ref=float(np.median(all_q)),
                                 paths_above_1e_8_vs_113bit_ref=int(np.sum(np.asarray(all_q) > 1e-8)), paths=len(all_q),
                                 every_status_success=all(ok for _, ok in worst_slot),
                                 distinct_inputs=True, seeds="(rank * slots + slot) * paths + p",
                                 note="max_coeff_err_vs_cpu_ref is set by ONE path (slot 15, path 237: a 0.179 s segment between 4.7 s "
                                      "and 4.0 s ones) on which the reference-style double oracle is 5.4e-7 off the 60-digit solution and "
                                      "the HIP path 4.8e-8 (tests/golden bench_slot15_path237_short_segment, "
                                      "tests/test_gpu_headline_kernel.py)")
        # the same comparison against the oracle's 113-bit route (the reference's algorithm without its rounding: what is left is
        # the HIP path's own error, where the figure above is dominated by the double-precision oracle's)
        po.lib().mto_set_arithmetic(po.QUAD_PRECISION)
        try:
            refq = po.solve_batch(sub.seg_offsets, sub.waypoints, sub.fixed_mask, sub.fixed_values, sub.limits, sub_t, deriv=4,
                                  n_threads=cores_for_checks())
        finally:
            po.lib().mto_set_arithmetic(po.REFERENCE_ARITHMETIC)
        eq = np.array([float(np.max(np.abs(gpu_c[a:b] - refq["coeffs"][a:b])) / np.max(np.abs(refq["coeffs"][a:b])))
                       for a, b in zip(sub.seg_offsets[:-1], sub.seg_offsets[1:])])
        err_exact = dict(max=float(eq.max()), median=float(np.median(eq)), share_below_1e_11=float((eq < 1e-11).mean()),
                         reference="oracle linear solve in 113-bit arithmetic (oracle/mto_linear.c, mto_set_arithmetic(2)), "
                                   "%d paths of this batch" % sub.n_paths)
        # all cores: the oracle's persistent thread pool on a configs[3]-sized batch (the 1024 paths tiled 64 times)
        visible, quota = host_parallelism()
        # as many threads as the cgroup grants CPUs (more only queue behind the quota's throttling); all visible ones without a quota
        cores = max(1, min(visible, int(quota + 0.5))) if quota else visible
        tile = max(1, CONFIG3_PATHS // n_cpu)
        parts = [sub.path(p) for p in range(sub.n_paths)] * tile
        big = pr.assemble_batch(parts, np.tile(sub.limits, (tile, 1)))
        big_t = np.tile(sub_t, tile)
        po.solve_batch(big.seg_offsets, big.waypoints, big.fixed_mask, big.fixed_values, big.limits, big_t, deriv=4,
                       n_threads=cores)   # creates the pool, touches the memory
        t0 = time.perf_counter()
        reps_all = 0
        while reps_all < 3 or (time.perf_counter() - t0 < 0.3 * args.cpu_seconds and reps_all < 50):
            po.solve_batch(big.seg_offsets, big.waypoints, big.fixed_mask, big.fixed_values, big.limits, big_t, deriv=4,
                           n_threads=cores)
            reps_all += 1
        dtn = (time.perf_counter() - t0) / reps_all
        cpu = dict(value=n_cpu / dt1, unit="trajectories/s", cores=1, kind="port",
                   sample="%d x the first %d of the %d paths of this batch (%.1f s of one core), linear QP, C oracle "
                          "(reference-style arithmetic, dense QR)" % (reps_cpu, n_cpu, P, reps_cpu * dt1),
                   value_all_cores=big.n_paths / dtn, cores_all=cores, host_threads_visible=visible,
                   cpu_quota_cpus=quota,
                   sample_all_cores="%d x a batch of %d paths (the same %d paths tiled), %d threads of the oracle's "
                                    "persistent pool, dynamic chunks" % (reps_all, big.n_paths, n_cpu, cores))
        if args.workload == "nonlinear" or not args.no_extras:
            n_nl = min(P, 256)
            subn = batch.select(range(n_nl))

            def cpu_nonlinear(b_, t_, threads):
                po.solve_batch(b_.seg_offsets, b_.waypoints, b_.fixed_mask, b_.fixed_values, b_.limits, t_, deriv=4,
                               time_alloc_method=2, sampling_dt=0.2, sample_capacity=512, n_threads=threads)
            tn = times[:subn.n_segments]
            t0 = time.perf_counter()
            cpu_nonlinear(subn, tn, 1)
            dtnl = time.perf_counter() - t0
            reps_nl = max(1, min(100, int(0.5 * args.cpu_seconds / max(dtnl, 1e-6))))
            t0 = time.perf_counter()
            for _ in range(reps_nl):
                cpu_nonlinear(subn, tn, 1)
            dtnl = (time.perf_counter() - t0) / reps_nl
            cpu["nonlinear_value"] = n_nl / dtnl
            cpu["nonlinear_sample"] = ("%d x %d paths (%.1f s of one core), Mellinger outer loop + scaling + sampling, 1 thread"
                                       % (reps_nl, n_nl, reps_nl * dtnl))
            tile_n = max(1, 8192 // n_nl)
            bign = pr.assemble_batch([subn.path(p) for p in range(n_nl)] * tile_n, np.tile(subn.limits, (tile_n, 1)))
            bign_t = np.tile(tn, tile_n)
            cpu_nonlinear(bign, bign_t, cores)
            t0 = time.perf_counter()
            reps_na = 0
            while reps_na < 2 or (time.perf_counter() - t0 < 0.3 * args.cpu_seconds and reps_na < 20):
                cpu_nonlinear(bign, bign_t, cores)
                reps_na += 1
            cpu["nonlinear_value_all_cores"] = bign.n_paths * reps_na / (time.perf_counter() - t0)
            cpu["nonlinear_sample_all_cores"] = "%d x %d paths, %d threads" % (reps_na, bign.n_paths, cores)

    if rank == 0:
        lin_desc = ("BASELINE configs[1]: %d random %d-segment order-10 min-snap paths per GPU, fixed times, linear QP"
                    % (P, args.segments))
        nl_desc = ("BASELINE configs[2]: %d random %d-segment paths per GPU, Mellinger outer loop (<=10 evaluations) + "
                   "feasibility scaling + sampling dt 0.2" % (P, args.segments))
        line = dict(metric="trajectories/sec (batch of N-seg min-snap paths)", value=value, unit="trajectories/s",
                    n_gpus=world, ranks_seen=ranks_seen, steps=args.steps, warmup=args.warmup,
                    ms_per_step=elapsed / args.steps * 1e3,
                    regions=len(region_s), ms_per_step_min=min(region_s) / args.steps * 1e3,
                    ms_per_step_median=elapsed / args.steps * 1e3, ms_per_step_max=max(region_s) / args.steps * 1e3,
                    ms_per_step_regions=[r / args.steps * 1e3 for r in region_s],
                    value_first_region=total_paths / region_s[0], value_best_region=total_paths / min(region_s),
                    ms_per_step_by_rank=dict(min=min(r["ms_per_step"] for r in per_rank), max=max(r["ms_per_step"] for r in per_rank)),
                    ranks=per_rank,
                    value_definition="MEDIAN over the R timed regions of: paths of all ranks x K / MAX over ranks of a rank's own time "
                                     "between its two synchronizes around the region's K steps (value_first_region: the single-region "
                                     "figure of rounds 1-5); the closing gather is timed on its own (gather_ms) and is inside "
                                     "value_including_gather -- compare THAT figure across N when the collective matters",
                    gather_ms=(gather_s * 1e3 if dist is not None else None),
                    value_including_gather=(total_paths / (elapsed + gather_s) if dist is not None else None),
                    gather_check=gather_check,
                    config3_strong_scaling=({k: extras["config3"][k] for k in ("value", "unit", "scaling", "n_gpus", "ms_per_step",
                                                                              "paths_per_rank")}
                                            if "config3" in extras else None),
                    one_batch_in_flight=(extras["one_batch_in_flight"]["value"] if "one_batch_in_flight" in extras else None),
                    value_200_steps=(value if args.steps == 200 else value_200),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                    config=dict(workload=lin_desc if args.workload == "linear" else nl_desc,
                                paths_per_gpu=P, segments=args.segments,
                                batches_in_flight=(n_group_slots if (grouped_mode[0] and args.workload == "linear" and n_lanes > 1) else n_lanes),
                                hip_streams=n_lanes,
                                positions=("MRS_TG_FLAG_POSITIONS_ARE_WAYPOINTS: every vertex of these batches has its waypoint as position "
                                           "constraint (as every vertex findTrajectory builds) and the bound solves say so -- checked by "
                                           "the library at bind time; the saturated-device solve then reads the compact [vertex][4] "
                                           "waypoint array instead of 8 bytes of every 160 of fixed_values: 1.05 instead of 1.34 times the "
                                           "compulsory bytes, same bits, no faster (HISTORY.md)"),
                                issue_policy=dict(batches_in_flight=args.in_flight, steps_per_dispatch=args.group_size, issue=args.issue,
                                                  frozen="since round 4: 20 batches in flight, 10 steps per dispatch, two streams; "
                                                         "rounds 1-3 used 4 / 4 / 16 in flight (HISTORY.md) -- compare rounds on "
                                                         "one_batch_in_flight (one batch, one dispatch per step: the strict reading "
                                                         "of configs[1]) and value_200_steps, both top-level keys",
                                                  stream_pair=(stream_pairs if stream_pairs is not None else
                                                               ("the first two lanes (torch's current stream and the first side stream), as in rounds 4-6"
                                                                if (grouped_mode[0] and args.workload == "linear" and n_lanes > 1) else None)),
                                                  stream_pair_note="round 6: WHICH two streams carry the dispatches is chosen by an untimed calibration "
                                                                   "of the two-dispatch round among --stream-candidates HIP streams (the runtime maps streams "
                                                                   "onto a few hardware queues; pairs that share one serialize); the policy -- 20 in flight, "
                                                                   "10 steps per dispatch, two streams -- is unchanged; --stream-candidates 2 = the pair of rounds 4-6"),
                                linear_solve="default of mrs_tg_plan_solve, blocks formed in registers (nothing materialised): "
                                             "solve_rows_kernel for a launch of one batch; a dispatch that carries >= 6144 paths (the "
                                             "grouped steps of the headline: %d x %d) runs solve_duo_group_kernel (eight lanes per path, "
                                             "the vertex chain eliminated from both ends; round 6) below 20480 paths and "
                                             "solve_quad_group_kernel (four lanes per path) from there on -- roofline_headline.kernel is "
                                             "what the trace says --, solve_rows_group_kernel a smaller group; the assembly kernel is timed on its "
                                             "own (roofline) and inside extras.materialized_blocks_step"
                                             % (max(1, min(args.group_size, n_group_slots)), P),
                                clock_ramp_steps=ramp_steps, clock_ramp_ms=30,
                                slot_inputs=("every batch in flight has its own masks, constrained values and segment times (path seeds "
                                             "(rank * slots + slot) * paths + p) and its own outputs" if len(slot_db) > 1 else
                                             "the batches in flight share ONE set of input arrays (--shared-inputs); outputs per slot"),
                                step_issue=(("mrs_tg_bound_solve_launch_group: %d consecutive steps (each a full pass over the batch of "
                                             "its slot, see slot_inputs) go out as ONE dispatch; the dispatches alternate "
                                             "over %d HIP stream(s)" % (max(1, min(args.group_size, n_group_slots)),
                                                                       min(n_lanes, (n_group_slots + max(1, min(args.group_size, n_group_slots)) - 1)
                                                                           // max(1, min(args.group_size, n_group_slots)))) if (grouped_mode[0] and args.workload == "linear" and n_lanes > 1)
                                             else "mrs_tg_bound_solve_launch_many_mt: the K steps are issued round-robin over the "
                                                  "streams by the library's C loop on %d host thread(s)" % min(args.issue_threads, n_lanes))
                                            if block_for(args.workload) is not None else "one Python call per step"),
                                launch_hint=("MRS_TG_FLAG_SHARED_DEVICE (several batches in flight: two paths per wavefront "
                                             "so that four launches fit the SIMDs side by side)"
                                             if (n_lanes > 1 and not (grouped_mode[0] and args.workload == "linear")) else "none"),
                                parallelism=("independent paths sharded per rank, no data-path collective; %s gather of "
                                             "the results to rank 0 %s"
                                             % ("gloo (host)" if gloo else "RCCL",
                                                "after every step, inside the timed region" if args.gather == "every" else
                                                "once, after the last step: timed on its own (gather_ms) right behind the K "
                                                "steps; value_including_gather has it inside"))
                                if world > 1 else "single GPU"),
                    max_coeff_err_vs_cpu_ref=err, max_coeff_err_vs_113bit_ref=err_exact,
                    in_flight_slots_vs_cpu_ref=(slots_checked if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None),
                    rccl_version=rccl_version,
                    roofline=roofline, roofline_headline=roofline_headline, roofline_solve=roofline_solve,
                    roofline_outer_loop=roofline_outer, cpu_baseline=cpu, extras=extras)
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
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
