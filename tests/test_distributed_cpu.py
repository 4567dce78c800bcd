"""The N>1 plumbing on CPU: two gloo ranks each hold the tile-major block rt_render_device would produce for
their shard, rank 0 gathers and re-assembles; the result must be the unsharded image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import cases
import oracle
from raytracer_amd import api


def shard_block(img: np.ndarray, rank: int, world: int) -> np.ndarray:
    """What rank `rank` of `world` would hand to the gather: its tiles in ascending global order, padded."""
    h, w, _ = img.shape
    slots, max_tiles, _, _ = api.tile_owner_slots(w, h, world)
    blk = np.zeros((max_tiles, api.TILE, api.TILE, 3), dtype=np.uint8)
    for (tx, ty), (o, k) in slots.items():
        if o != rank:
            continue
        y0, x0 = ty * api.TILE, tx * api.TILE
        t = img[y0:y0 + api.TILE, x0:x0 + api.TILE]
        blk[k, :t.shape[0], :t.shape[1]] = t
    return blk.reshape(-1)


def _worker(rank: int, world: int, port: int, img: np.ndarray, out_path: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from raytracer_amd import dist as rdist
    import torch.distributed as dist
    r, w, _ = rdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    text = rdist.broadcast_bytes(b"scene-bytes" if rank == 0 else None)
    assert text == b"scene-bytes"
    # scene broadcast: rank 0 parses and builds, the others rebuild the C-ABI arrays from the blob
    from raytracer_amd import scenegen as sg
    cfg, _ = api.parse_args(["-t"])
    mine = api.LoadedScene(sg.glass_scene((64, 36), True), cfg, str(cases.TEXTURES)) if rank == 0 else None
    got = rdist.broadcast_scene(mine, True)
    ref = api.LoadedScene(sg.glass_scene((64, 36), True), cfg, str(cases.TEXTURES))
    assert bytes(api.pack_scene(got, True)) == bytes(api.pack_scene(ref, True))
    assert got.scene.n_textures == 2 and got.bvh().n_nodes == ref.bvh().n_nodes
    block = torch.from_numpy(shard_block(img, rank, world))
    gathered = rdist.gather_blocks(block, dst=0)
    if rank == 0:
        blocks = [c.numpy() for c in gathered.chunk(world)]
        np.save(out_path, api.assemble_tiles_numpy(blocks, img.shape[1], img.shape[0]))
    else:
        assert gathered is None
    # the asynchronous form the frame pipeline uses: two gathers in flight into preallocated buffers, waited in order
    outs = [torch.empty(world * block.numel(), dtype=torch.uint8) if rank == 0 else None for _ in range(2)]
    second = block.flip(0).contiguous()
    h0 = rdist.gather_blocks_async(block, outs[0]); h1 = rdist.gather_blocks_async(second, outs[1])
    h0.wait(); h1.wait()
    if rank == 0:
        assert torch.equal(outs[0], gathered)
        assert torch.equal(outs[1].view(world, -1)[rank], second) and not torch.equal(outs[0], outs[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_and_assemble_world_n(world, tmp_path):
    img, _ = oracle.render(cases.reference_scene(200, 112), [])     # sizes that are not multiples of the tile
    img = img.astype(np.uint8)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    out = str(tmp_path / "assembled.npy")
    mp.spawn(_worker, args=(world, port, img, out), nprocs=world, join=True)
    assert np.array_equal(np.load(out), img)


def test_tile_ownership_is_balanced_and_complete():
    for (w, h) in [(1920, 1080), (3840, 2160), (200, 112), (31, 33)]:
        for world in (1, 2, 4, 8):
            slots, max_tiles, tx_n, ty_n = api.tile_owner_slots(w, h, world)
            assert len(slots) == tx_n * ty_n
            counts = np.bincount([o for o, _ in slots.values()], minlength=world)
            assert counts.max() == max_tiles and counts.max() - counts.min() <= max(1, ty_n)
            for r in range(world):                       # slot indices are 0..count-1 in ascending tile order
                ks = [k for (tx, ty), (o, k) in sorted(slots.items(), key=lambda kv: (kv[0][1], kv[0][0])) if o == r]
                assert ks == list(range(len(ks)))


def test_bench_starts_its_own_ranks_when_asked_for_several_gpus():
    """`python bench.py --gpus 2` without a torchrun environment: the script starts the ranks itself (a child
    torch.distributed.run), the ranks initialise the process group (gloo here) and reduce one number, rank 0's line comes
    back through the parent, which passes the child's exit status on."""
    import json
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(cases.ROOT / "bench.py"), "--gpus", "2", "--single-device", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"launch_check": "ok", "world": 2, "sum_of_ranks_plus_one": 3.0}
    # a --gpus count that contradicts the launcher's world size is refused, with the child's status passed on
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(cases.ROOT / "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=120, env=env2)
    assert p.returncode == 2
