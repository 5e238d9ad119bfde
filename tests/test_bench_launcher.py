"""bench.py's multi-GPU entry: `--gpus N` without a torch.distributed environment must start N ranks itself, as a child
process, before the parent has imported torch or touched a GPU (a process that has initialised the GPU must never be
replaced by another program on this pool)."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launcher_command_is_the_drivers_command():
    bench = _bench()
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.launcher_command(8, argv, 29411, python="python3", script="/x/bench.py")
    assert cmd == ["python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", "29411", "/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    args = bench.parse_args(argv)
    assert (args.gpus, args.steps, args.warmup, args.workload, args.paths, args.in_flight, args.streams, args.group_size) == (8, 20, 5, "linear", 1024, 20, 4, 10)
    assert bench.parse_args([]).gpus == 1


def test_importing_bench_does_not_import_torch():
    """The decision to self-launch is taken before torch is imported: importing the module and parsing arguments must not
    pull torch in."""
    code = ("import sys, importlib.util; spec = importlib.util.spec_from_file_location('b', %r); m = importlib.util.module_from_spec(spec); "
            "spec.loader.exec_module(m); m.parse_args(['--gpus', '4']); print('torch' in sys.modules)" % os.path.join(ROOT, "bench.py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    assert out.stdout.strip() == "False"


def test_self_launch_relays_the_result_line(monkeypatch, capsys):
    """The child's JSON line is the only thing the parent prints on stdout; its exit code is the parent's."""
    bench = _bench()
    seen = {}

    class Done:
        returncode = 0
        stdout = 'NCCL banner\n{"metric": "x", "value": 1}\n'

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        seen["env"] = kw["env"]
        return Done()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    args = bench.parse_args(["--gpus", "2", "--steps", "3"])
    assert bench.self_launch(args, ["--gpus", "2", "--steps", "3"]) == 0
    assert seen["cmd"][1:6] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2"]
    assert seen["cmd"][-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "x", "value": 1}' and "NCCL banner" in out.err


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself_on_one_gpu():
    """`bench.py --gpus 2` with no RANK in the environment: two ranks (sharing the box's single GPU, collectives over
    gloo), configs[3]'s fixed batch cut in two, one JSON line that says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "5",
                          "--warmup", "2", "--paths", "256", "--config3-paths", "2048", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["config"]["paths_per_gpu"] == 256 and line["value"] > 0
    c3 = line["extras"]["config3"]
    assert c3["n_gpus"] == 2 and c3["paths_per_rank"] == 1024 and c3["scaling"] == "strong" and c3["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["roofline_solve"]["achieved"] > 0 and line["roofline_outer_loop"]["avg_launch_us"] > 0
    # the closing gather is timed on its own; the per-rank throughput of the two ranks sharing one GPU is reported as is
    assert line["gather_ms"] is not None and line["value_including_gather"] <= line["value"]


@pytest.mark.gpu
def test_single_rank_rccl_path_runs_and_the_gather_delivers_the_local_results():
    """The only way to execute RCCL's initialisation and `dist.gather` on a one-GPU lease: `--force-dist --dist-backend nccl`
    with one rank.  The line must say one rank was seen, carry configs[3]'s object, time the closing gather on its own and
    confirm that what the gather put into rank 0's receive buffers is bit for bit what rank 0 computed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl",
                          "--steps", "5", "--warmup", "2", "--config3-paths", "4096", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert line["gather_ms"] is not None and line["gather_ms"] > 0 and line["value_including_gather"] < line["value"]
    assert line["gather_check"] == dict(root_equals_local=True, peers_finite_with_valid_status=True, ranks=1,
                                        bytes_per_rank=(1024 * 10 * 41 + 1024) * 8)
    c3 = line["extras"]["config3"]
    assert c3 is not None and c3["value"] > 0 and c3["scaling"] == "strong" and c3["paths_per_rank"] == 4096
    assert c3["gather_root_equals_local"] is True
    assert line["config3_strong_scaling"]["value"] == c3["value"]
    assert "gather_every_step" in line["extras"] and line["extras"]["gather_every_step"]["value"] > 0
