"""GPU tier, ONE rank: the multi-GPU code path (magicdance_amd/parallel.py, ddim.FusedStepRunner.run_steps / enqueue_chunk) on a
real RCCL process group of world size 1 -- blocked reference-KV table, in-place ``all_gather_into_tensor`` of every table chunk on
the table stream while captured step graphs replay on the step stream, decoded-frame / latent all-gather -- and its result against
the single-process route, BIT for bit (same kernels, same batches, deterministic reductions; the collectives move bytes only).
The 8-GPU run itself is the driver's (SCALE_rNN.json); world 2 / 3 / 4 logic incl. uneven and empty shards is covered on gloo
(tests/test_parallel_gloo.py)."""
import os
import socket

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def group(dev):
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    yield dist.group.WORLD
    dist.destroy_process_group()


def _inputs(side, frames, dev, seed=3):
    from magicdance_amd import synthetic
    inp = synthetic.synth_inputs((side, side), frames=frames, seed=seed, device=dev)
    return inp["pose"], inp["ctx"], inp["ref"], inp["x_T"]


def test_sharded_route_on_a_one_rank_rccl_group_is_bit_identical(dev, group):
    from magicdance_amd import parallel
    model = H.build_hip_model(64, 2, seed=0, device=dev, image_size=16)
    frames, steps = 3, 10
    pose, ctx, ref, x_T = _inputs(16, frames, dev)
    xs = x_T.repeat(frames, 1, 1, 1)
    # identical appearance batches on both routes: 5 timesteps per pass = per row block (2 chunks of 5 rows over 1 rank)
    single = parallel.FrameShardedSampler(model, rank=0, world=1)
    single._runner().bank_chunk = 5
    z_ref = single.sample(pose, ctx, ref, xs, ddim_steps=steps, scale=7.0)
    st = model._fused
    assert (st.per, st.nblocks) == (5, 2)
    model._fused = None
    sharded = parallel.FrameShardedSampler(model, rank=0, world=1, group=group, force_sharded=True)
    sharded._runner().bank_chunk = 5
    sharded._runner().table_chunks = 2
    z = sharded.sample(pose, ctx, ref, xs, ddim_steps=steps, scale=7.0)          # table all-gathers + latent all-gather
    st = model._fused
    assert st.table_mode and (st.per, st.nblocks, st.n_chunks(1)) == (5, 2, 2) and st.graph is not None
    torch.cuda.synchronize()
    assert z.shape == z_ref.shape and bool(torch.isfinite(z).all())
    assert torch.equal(z, z_ref)
    # the sequence entry point (the scripts' route): uneven batches, results gathered with counts; table filled once
    model._fused = None
    z_seq = sharded.sample_sequence(pose, ctx, ref, x_T, frames_per_batch=2, ddim_steps=steps, scale=7.0, gather_counts=[frames])
    torch.cuda.synchronize()
    model._fused = None
    single2 = parallel.FrameShardedSampler(model, rank=0, world=1)
    z_seq_ref = single2.sample_sequence(pose, ctx, ref, x_T, frames_per_batch=2, ddim_steps=steps, scale=7.0)
    # (the two routes batch the appearance timesteps differently here -- 16 per pass vs 5-row blocks -- so the bank rows differ by
    #  accumulation-order noise of the batched GEMMs; the collectives themselves are exact: see the bit-identical check above)
    err = float((z_seq - z_seq_ref).abs().max() / z_seq_ref.abs().max())
    assert z_seq.shape == (frames, 4, 16, 16) and err <= 2e-3, err


def test_full_size_sharded_route_one_rank(dev, group):
    """configs[3]'s per-rank work at the real geometry (512x512, 50 steps, 2 frames on this rank): two 25-row table chunks, each
    all-gathered in place (1.15 GB) on the table stream while the first chunk's step graphs replay; decoded-frame gather."""
    from magicdance_amd import parallel
    model = H.build_hip_model(320, 8, seed=0, device=dev, image_size=64)
    pose, ctx, ref, x_T = _inputs(64, 2, dev, seed=0)
    xs = x_T.repeat(2, 1, 1, 1)
    sharded = parallel.FrameShardedSampler(model, rank=0, world=1, group=group, force_sharded=True)
    z = sharded.sample(pose, ctx, ref, xs, ddim_steps=50, scale=7.0)
    torch.cuda.synchronize()
    st = model._fused
    assert (st.per, st.nblocks) == (25, 2)
    model._fused = None
    z_ref = parallel.FrameShardedSampler(model, rank=0, world=1).sample(pose, ctx, ref, xs, ddim_steps=50, scale=7.0)
    err = float((z - z_ref).abs().max() / z_ref.abs().max())
    assert bool(torch.isfinite(z).all()) and err <= 1.5e-3, err   # appearance batches of 16 (+9) vs 16/16/16/2 rows: rounding only


def test_bench_multi_gpu_code_path_on_a_one_rank_group(dev):
    """bench.py's N > 1 branches (process group, join count, barriers, max-over-ranks timing, the sharded sampler, the configs[3]-
    shaped `extra` leg run by every rank, rank-0 epilogue, group teardown) cannot run on a one-GPU box as N > 1 -- ``--force-sharded``
    runs exactly those branches on a 1-rank RCCL group.  A subprocess: bench.py owns its process group."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-sharded", "--size", "16", "--ddim-steps", "6",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], cwd=root, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("{"), "the JSON line must be the LAST line on stdout (after RCCL's banner): " + r.stdout[-500:]
    line = json.loads(last)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["frames_per_gpu"] == 1 and "scaling_reference" in line
    (name, e8), = line["extra"].items()
    assert "configs[3]" in name and e8["n_gpus"] == 1 and e8["value"] > 0 and e8["steps"] == 3
