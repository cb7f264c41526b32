"""CPU tier: bench.py's rank launch / join protocol (the driver's contract is `python bench.py --gpus N`; the reference's scripts
launch through torchrun, scripts/inference_any_image_pose.sh:4).  `--selftest-launch` runs only that protocol -- self-spawn of the
ranks under torch.distributed.run, process group (gloo without a GPU), join count -- and prints its JSON line; no model, no kernels."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_bench_spawns_its_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--selftest-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["selftest"] == "launch"


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run(["--gpus", "2", "--selftest-launch"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1", "--selftest-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1


def test_per_gpu_work_is_the_same_at_every_n():
    """weak scaling of the headline config: the default run keeps ONE frame per GPU and batch at N = 1, 2, 4, 8 (the driver computes
    its scaling efficiency from the per-N values); the 8-frames-per-GPU shapes are explicit"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert [bench.workload_for(None, n)[0] for n in (1, 2, 4, 8)] == [1, 1, 1, 1]
    assert bench.workload_for(None, 1)[1] == "configs[1]" and bench.workload_for(8, 1)[1] == "configs[2]"
    assert bench.workload_for(8, 8)[1] == "configs[3]" and "configs[1]" in bench.workload_for(None, 8)[1]
