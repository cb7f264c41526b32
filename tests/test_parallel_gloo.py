"""CPU tier, world_size 2 over gloo: the frame-sharded sampler (bank table computed round-robin over ranks, broadcast,
per-rank step loop, latent all-gather) must reproduce the single-process result.  Kernels are emulated on CPU
(tests/hip_emulator.py); what is under test is the sharding / collective logic of magicdance_amd/parallel.py."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import torch.distributed as dist
    from _pytest.monkeypatch import MonkeyPatch
    from tests import helpers as H, hip_emulator
    from magicdance_amd import ddim, parallel, synthetic
    mpatch = MonkeyPatch()
    hip_emulator.install(mpatch)
    orig = ddim.FusedStepRunner.__init__

    def init(self, model):
        orig(self, model)
        self.use_graph = False
    mpatch.setattr(ddim.FusedStepRunner, "__init__", init)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = H.build_hip_model(64, 2, seed=0, device="cpu", image_size=8)
        fpg, steps = 1, 5   # odd: the table is padded to 6 rows, rank 1 owns rows 3..4 of its block of 3
        inp = synthetic.synth_inputs((8, 8), frames=fpg * world, seed=3)
        x_T = inp["x_T"].repeat(fpg, 1, 1, 1)
        my = slice(rank * fpg, (rank + 1) * fpg)
        sharded = parallel.FrameShardedSampler(model, rank=rank, world=world)
        z_all = sharded.sample(inp["pose"][my].contiguous(), inp["ctx"], inp["ref"], x_T, ddim_steps=steps, scale=7.0)
        if rank == 0:
            model._fused = None
            single = parallel.FrameShardedSampler(model, rank=0, world=1)
            z_ref = single.sample(inp["pose"], inp["ctx"], inp["ref"], inp["x_T"].repeat(fpg * world, 1, 1, 1),
                                  ddim_steps=steps, scale=7.0)
            torch.save({"z_all": z_all, "z_ref": z_ref}, out_path)
    finally:
        dist.destroy_process_group()
        mpatch.undo()


@pytest.mark.timeout(600)
def test_frame_sharding_matches_single_process(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["z_all"].shape == r["z_ref"].shape == (2, 4, 8, 8)
    # same arithmetic per frame (bank via table vs inline, frames batched vs alone); the CPU emulation picks different
    # conv algorithms for different batch sizes, so fp16-storage rounding flips show up (on the GPU the difference is 0)
    err = float((r["z_all"] - r["z_ref"]).abs().max() / r["z_ref"].abs().max())
    assert err <= 6e-3, err
