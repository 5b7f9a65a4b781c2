"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares (no compute without a GPU), and the N>1 path (tile sharding + one gather + assembly) is exercised
with two gloo ranks."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest

import hexray_amd as hx
from conftest import ROOT, scene_path


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hx[rh]_[a-z0-9_]+)\s*\(", text)) - {"hxr_progress_cb"})


def test_hip_library_exports_every_declared_symbol():
    names = declared_functions(os.path.join(ROOT, "include", "hexray_hip.h"))
    assert len(names) >= 19
    lib = ctypes.CDLL(hx.HIP_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libhexray_hip.so does not export {n}"


def test_host_library_exports_every_declared_symbol():
    names = declared_functions(os.path.join(ROOT, "hexray_amd", "host", "host_api.h"))
    lib = ctypes.CDLL(hx.HOST_LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"libhexray_host.so does not export {n}"


def test_error_codes_without_a_gpu_or_scene():
    """Entry points reject bad arguments with negative codes and a message; nothing crosses the ABI as an exception."""
    lib = hx.hip_lib()
    p = hx.frame_params(width=1920, height=1080, mode=hx.MODE_SINGLE, tile_size=20)   # not a multiple of 8
    assert lib.hxr_shard_tile_count(None, ctypes.byref(p)) == -1
    assert b"tile_size" in lib.hxr_last_error()
    p = hx.frame_params(width=4000, height=1080, mode=hx.MODE_SINGLE)                 # beyond VFB_MAX_SIZE
    assert lib.hxr_shard_tile_count(None, ctypes.byref(p)) == -1
    p = hx.frame_params(mode=hx.MODE_SINGLE)                                          # size must be explicit without a scene
    assert lib.hxr_shard_tile_count(None, ctypes.byref(p)) == -1
    assert lib.hxr_render(None, ctypes.byref(p), None, 0, None, None, None) == -1
    assert lib.hxr_get_counters(None, None) == -1


def test_shard_bookkeeping_matches_python_mirror():
    from hexray_amd.distributed import shard_tile_count, tile_position, tiles_in_frame
    lib = hx.hip_lib()
    for (w, h, tile, world) in ((1920, 1080, 64, 1), (1920, 1080, 32, 8), (640, 480, 64, 3), (100, 60, 16, 4)):
        tx, ty = tiles_in_frame(w, h, tile)
        seen = np.zeros((h, w), dtype=np.int32)
        for rank in range(world):
            p = hx.frame_params(width=w, height=h, mode=hx.MODE_SINGLE, tile_size=tile, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
            n = lib.hxr_shard_tile_count(None, ctypes.byref(p))
            assert n == shard_tile_count(w, h, tile, rank, world)
            assert lib.hxr_output_bytes(None, ctypes.byref(p), 0) == n * tile * tile * 12
            for i in range(n):
                r = [ctypes.c_int() for _ in range(4)]
                assert lib.hxr_shard_tile_rect(None, ctypes.byref(p), i, *r) == 0
                x0, y0, x1, y1 = (v.value for v in r)
                t = i * world + rank
                cx, cy = tile_position(t, tx, world)
                assert (x0, y0) == (cx * tile, cy * tile)
                if world == 1:
                    assert (cx, cy) == (t % tx, t // tx)          # a single shard walks the frame row by row
                seen[y0:y1, x0:x1] += 1
        assert seen.min() == 1 and seen.max() == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, w, h, tile, out_path):
    """One rank of the sharded pipeline on the CPU: the per-rank render is stood in for by cropping this
    rank's tiles out of an oracle frame; the sharding, the gather and the assembly are the real code."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import hexray_amd as hx
    from hexray_amd.distributed import assemble_tiles_numpy, gather_tiles, padded_tiles_per_rank
    from oracle_binding import oracle_render

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(w, h))
    full, _ = oracle_render(hs, hx.frame_params(mode=hx.MODE_SINGLE), threads=2)
    lib = hx.hip_lib()
    p = hx.frame_params(width=w, height=h, mode=hx.MODE_SINGLE, tile_size=tile, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
    n = lib.hxr_shard_tile_count(None, ctypes.byref(p))
    slots = padded_tiles_per_rank(w, h, tile, world)
    local = np.zeros((slots, tile, tile, 3), dtype=np.float32)
    for i in range(n):
        r = [ctypes.c_int() for _ in range(4)]
        lib.hxr_shard_tile_rect(None, ctypes.byref(p), i, *r)
        x0, y0, x1, y1 = (v.value for v in r)
        local[i, :y1 - y0, :x1 - x0] = full[y0:y1, x0:x1]
    gathered = gather_tiles(torch.from_numpy(local.reshape(-1)), world, rank)
    if rank == 0:
        frame = assemble_tiles_numpy(gathered.numpy(), w, h, tile, world)
        np.save(out_path, np.stack([frame, full]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame(tmp_path):
    import torch.multiprocessing as mp
    w, h, tile, world = 104, 72, 16, 2
    out = str(tmp_path / "frames.npy")
    mp.spawn(_rank_main, args=(world, _free_port(), w, h, tile, out), nprocs=world, join=True)
    frame, full = np.load(out)
    assert np.array_equal(frame, full)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE) must start N ranks itself, as its FIRST action, and report the number
    of ranks that joined; under a launcher (WORLD_SIZE set) it must not launch again.  --rendezvous-check stops before any GPU work."""
    import importlib.util
    import json
    import subprocess
    import sys

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    saved = os.environ.pop("WORLD_SIZE", None)
    try:
        args = bench.parse_args(["--gpus", "4", "--steps", "2"])
        cmd = bench.launch_command(args, ["--gpus", "4", "--steps", "2"], 29999)
        assert cmd is not None and cmd[1:3] == ["-m", "torch.distributed.run"]
        assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
        assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
        assert bench.launch_command(bench.parse_args(["--gpus", "1"]), [], 0) is None
        os.environ["WORLD_SIZE"] = "4"
        assert bench.launch_command(args, [], 0) is None          # already a rank: the driver's torch.distributed.run form
    finally:
        os.environ.pop("WORLD_SIZE", None)
        if saved is not None:
            os.environ["WORLD_SIZE"] = saved
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-check"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec == {"rendezvous_check": True, "n_gpus": 2, "asked": 2}
    assert "starting the ranks" in r.stderr
    # a rank count that does not match --gpus is refused, not reported
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-check"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "ranks joined" in (r.stderr + r.stdout)
