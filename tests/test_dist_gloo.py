"""N > 1 path on CPU: two gloo ranks each produce their round-robin row bands, one gather puts the image
together on rank 0, and it must equal the 1-rank image byte for byte.  The per-rank renderer here is the CPU
oracle (this is a test of the partition + gather plumbing, not of the kernels)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, band_rows, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    import raytracer_amd as R
    from raytracer_amd.dist import BandGather

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W, H = 61, 45
    scene = R.stock_scene("cube_spheres", W, H, tri_reflective=0.3, sph_reflective=0.3)
    bg = BandGather(H, W, band_rows, world, rank, torch.device("cpu"))
    part, st = O.render_rgb(scene, n_ranks=world, band_rows=band_rows, rank=rank)
    assert part.shape[0] == bg.n_local
    bg.local[: bg.n_local] = torch.from_numpy(part)
    rays = torch.tensor([st.rays], dtype=torch.int64)
    dist.all_reduce(rays)
    img = bg.gather()
    if rank == 0:
        full, fst = O.render_rgb(scene)
        assert rays.item() == fst.rays
        np.save(out_path, np.stack([img.numpy(), full]))
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,band_rows", [(2, 4), (3, 1)])
def test_two_rank_bands_gather_to_the_one_rank_image(tmp_path, world, band_rows):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), band_rows, out), nprocs=world, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)


def _exchange_worker(rank, world, port, band_rows, rounds, n_frames, out_dir):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    import raytracer_amd as R
    from raytracer_amd.dist import FrameExchange

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W, H = 45, 37
    # every frame is a different image, so a frame that ends up on the wrong rank or in the wrong slot shows
    scenes = [R.stock_scene("cube_spheres", W, H, tri_reflective=0.1 * f, sph_reflective=0.05 * f) for f in range(n_frames)]
    fx = FrameExchange(H, W, band_rows, world, rank, torch.device("cpu"), rounds=rounds, slots=2)
    got = {}
    pending = [None, None]
    base = [0, 0]
    f, slot = 0, 0
    while f < n_frames:
        if pending[slot] is not None:
            for j, img in fx.finish(pending[slot], slot):
                got[base[slot] + j] = img.numpy().copy()
        n = min(fx.group, n_frames - f)
        for j in range(n):
            part, _ = O.render_rgb(scenes[f + j], n_ranks=world, band_rows=band_rows, rank=rank)
            fx.target(slot, j)[: fx.n_local] = torch.from_numpy(part)
        pending[slot], base[slot] = fx.exchange_async(slot, n), f
        f += n
        slot ^= 1
    for k in (slot, slot ^ 1):
        if pending[k] is not None:
            for j, img in fx.finish(pending[k], k):
                got[base[k] + j] = img.numpy().copy()
    assert sorted(got) == [i for i in range(n_frames) if i % world == rank]
    for i, img in got.items():
        full, _ = O.render_rgb(scenes[i])
        assert np.array_equal(img, full), f"frame {i} on rank {rank}"
    count = torch.tensor([len(got)], dtype=torch.int64)
    dist.all_reduce(count)
    if rank == 0:
        with open(os.path.join(out_dir, "frames.txt"), "w") as fh:
            fh.write(str(int(count.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,band_rows,rounds,n_frames", [(2, 4, 2, 9), (3, 1, 1, 4), (1, 1, 2, 3)])
def test_frame_exchange_puts_frame_f_on_rank_f_mod_n(tmp_path, world, band_rows, rounds, n_frames):
    mp.spawn(_exchange_worker, args=(world, _free_port(), band_rows, rounds, n_frames, str(tmp_path)), nprocs=world,
             join=True)
    assert int(open(tmp_path / "frames.txt").read()) == n_frames


def test_band_helpers():
    from raytracer_amd.dist import default_band_rows, rank_rows

    assert default_band_rows(1080, 8) == 16
    assert default_band_rows(20, 8) >= 1
    ys = [rank_rows(50, 4, 3, r) for r in range(3)]
    assert sorted(np.concatenate(ys).tolist()) == list(range(50))


def _bench_loop_worker(rank, world, port, n_frames, out_dir):
    """bench.py's own run_frames() loop (the timed region of `bench.py --gpus N`), with the CPU oracle as the
    renderer and gloo as the transport: every frame must end up complete on exactly one rank."""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import bench
    import oracle_lib as O
    import raytracer_amd as R
    from raytracer_amd.dist import FrameExchange, default_band_rows

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    W, H = 45, 37
    scene = R.stock_scene("cube", W, H)
    band_rows = default_band_rows(H, world)
    fx = FrameExchange(H, W, band_rows, world, rank, torch.device("cpu"), rounds=2, slots=2)
    part, _ = O.render_rgb(scene, n_ranks=world, band_rows=band_rows, rank=rank)
    part = torch.from_numpy(part)
    seen = []

    def render(target):
        target[: fx.n_local] = part

    real_finish = fx.finish

    def finish(handle, slot=0):
        frames = real_finish(handle, slot)
        seen.extend(img.numpy().copy() for _, img in frames)
        return frames

    fx.finish = finish
    held = bench.run_frames(render, fx, n_frames)
    assert held == len(seen)
    full, _ = O.render_rgb(scene)
    for img in seen:
        assert np.array_equal(img, full)
    count = torch.tensor([held], dtype=torch.int64)
    dist.all_reduce(count)
    if rank == 0:
        with open(os.path.join(out_dir, "held.txt"), "w") as fh:
            fh.write(str(int(count.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 11), (3, 7)])
def test_bench_run_frames_loop_over_gloo(tmp_path, world, n_frames):
    mp.spawn(_bench_loop_worker, args=(world, _free_port(), n_frames, str(tmp_path)), nprocs=world, join=True)
    assert int(open(tmp_path / "held.txt").read()) == n_frames
