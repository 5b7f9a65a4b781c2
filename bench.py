"""Benchmark of the hot path: frames of BASELINE.json's headline workload through the C-ABI HIP module.

    python bench.py --gpus N --steps K --warmup W

A "step" is one frame of `data/meshes.hexray` at 1920x1080 with 16 samples per pixel (Camera `dof on,
numSamples 16` -- SURVEY.md section 8d, config 3b), i.e. one pass of the per-pixel / per-sample raytrace
loop over 33,177,600 camera samples.  The scene is resident in HBM before the timed region starts and the
frame stays in HBM (the PCIe-inclusive rate is noted in DESIGN.md).  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the frame's 32x32 tiles are dealt round-robin to the ranks, each
rank renders its tiles, and ONE RCCL gather brings them to rank 0, which assembles the frame: the total
work per step is fixed, so scaling is "strong".

Rank 0 prints one JSON line.  `roofline` is the one that binds this kernel -- FP64 vector-ALU issue (SURVEY.md section 8d names it as
the secondary roofline; the working set lives in L2 / Infinity Cache, so the HBM line of 8d does not bind): `achieved` = useful FP64
flop per launch (FMA / MUL / ADD wave-instructions x 64 lanes x the measured VALU lane utilisation) / the render kernel's mean
duration from HIP events on the launch stream, `peak` = 78.6 TFLOP/s, so `frac` <= 1 by construction.  The counters behind it
(and `traffic`, the HBM bytes per launch) are collected BY THIS RUN: three `rocprofv3 --pmc` child processes (counters only, one
pass each for the SQ set, FETCH_SIZE and WRITE_SIZE) render the same workload before this process touches the GPU.  Where that is not
possible the committed profile is used, labelled, and only when it was taken on the kernel sources of this checkout (`kernel_build`).
SURVEY.md 8d's algorithmic-bytes figure stays beside it as `hbm_algorithmic`.  `cpu_baseline` times the CPU oracle (test
infrastructure, a port of the reference's algorithm) on this host's cores for the same frame, N = 1 only.

What `value` is and is not: the timed frames are steady-state re-renders of the same frame into a device buffer -- from the
second frame on the packet schedule is seeded by the previous frame's per-packet clocks.  `other_timings`
carries what that leaves out: the FIRST frame of a fresh scene handle (no cost map; what a one-shot render() gets), the steady
state with the adaptive schedule switched off, and frames through hxr_render(), which add the copy to a host buffer.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, overlay kwargs, description)
    "meshes16": ("meshes.hexray", dict(width=1920, height=1080, dof=True, num_samples=16),
                 "data/meshes.hexray 1920x1080 x 16 spp (Camera dof on, numSamples 16; renderWithMonteCarlo)"),
    "meshes_aa": ("meshes.hexray", dict(width=1920, height=1080),
                  "data/meshes.hexray 1920x1080 adaptive 5-tap AA (as shipped; renderWithoutMonteCarlo)"),
    "meshes1": ("meshes.hexray", dict(width=1920, height=1080, want_aa=False),
                "data/meshes.hexray 1920x1080 x 1 spp (pass 1 of renderWithoutMonteCarlo alone)"),
    "simple1": ("simple.hexray", dict(width=1920, height=1080, want_aa=False),
                "data/simple.hexray 1920x1080 x 1 spp"),
    "heightfield16": ("heightfield.hexray", dict(width=1920, height=1080, dof=True, num_samples=16),
                      "data/heightfield.hexray 1920x1080 x 16 spp (Camera dof on, numSamples 16)"),
    "smallpt256": ("smallpt.hexray", dict(width=1920, height=1080, num_paths=256), "data/smallpt.hexray 1920x1080 x 256 paths"),
    "smallpt16": ("smallpt.hexray", dict(width=1920, height=1080, num_paths=16), "data/smallpt.hexray 1920x1080 x 16 paths"),
    # the two catch-all kernel variants (every geometry kind compiled in; with / without path tracing)
    "cornell16": ("cornell_box.hexray", dict(width=1920, height=1080, num_paths=16), "data/cornell_box.hexray 1920x1080 x 16 paths (GI over meshes)"),
    "kdtree16": ("kdtree_test.hexray", dict(width=1920, height=1080, dof=True, num_samples=16), "data/kdtree_test.hexray 1920x1080 x 16 spp (CSG + kd mesh)"),
    # no nodes at all: the fixed per-packet cost (sample generation, camera, reduction, store)
    "empty16": ("../tests/golden/tiny.hexray", dict(width=1920, height=1080, dof=True, num_samples=16), "empty scene 1920x1080 x 16 spp (overhead probe)"),
}
TILE = 64          # single GPU: only the order of the packets depends on it (the reference's bucket size; tools/tile_sweep.py: 8.80-8.85 ms at 64 / 128 / 8, 8.92 at 32)
SHARD_TILE = 8     # N > 1: tiles are dealt round-robin to the ranks; 8x8 spreads the expensive image regions best (rank time max/mean 1.13 vs 1.35 at 32)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 FP64 FMA lanes x 2 flop x 2.4 GHz
# rocprofv3 --pmc passes of the live counter collection (MI355X_MICROARCH.md, "rocprofv3 PMC slots": 8 SQ + 2 GRBM per pass; FETCH_SIZE and
# WRITE_SIZE do not fit one pass).  Counters only: no tracing domain is combined with --pmc.
PMC_PASSES = (
    ("sq", ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
            "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "GRBM_GUI_ACTIVE")),
    ("fetch", ("FETCH_SIZE",)),
    ("write", ("WRITE_SIZE",)),
)
PMC_CHILD_FRAMES = 4    # frames a counter pass renders; the LAST dispatch of the render kernel is read (steady state: scheduled from a cost map)


def algorithmic_bytes(c, width, height):
    """bytes the algorithm must read / write for one frame, from its work counters."""
    return (24 * c["kd_visits"] + 4 * c["leaf_indices"] + 100 * c["tri_tests"] + 192 * c["mesh_hits"] + 48 * c["tex_bilinear"]
            + 48 * c["env_bilinear"] + 4 * (c["hf_highest"] + c["hf_height"]) + 96 * c["hf_normals"] + 12 * width * height)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="meshes16", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sah-side", action="store_true", help="skip the SAH-tree side measurement (profiling runs: one kernel configuration only)")
    ap.add_argument("--no-other-timings", action="store_true", help="skip the first-frame / no-schedule / host-buffer / deferred-ray side measurements (profiling runs)")
    ap.add_argument("--no-config5", action="store_true", help="N > 1: skip the smallpt x 256 paths sub-record (BASELINE.json's 8-GPU configuration)")
    ap.add_argument("--accel", default="reference", choices=["reference", "sah"],
                    help="kd-trees walked by the timed frames: the reference's own (default) or the module's SAH rebuild")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not start the rocprofv3 --pmc child passes (roofline then comes from the committed profile, if it matches this checkout)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # what the counter passes run: a few frames of the workload, nothing else
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="launch / join the ranks, agree on the world size over gloo, print it and stop before any GPU work (CPU test of the launcher)")
    return ap.parse_args(argv)


def launch_command(args, argv, port):
    """The command that starts one rank per GPU, or None when this process IS a rank (WORLD_SIZE set: the driver's
    `python -m torch.distributed.run ... bench.py --gpus N` form) or a single GPU is asked for.  The reference's analogue is the
    thread fan-out of render() (src/main.cpp:273-275,364-367)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as children of THIS process, which has not imported torch,
    loaded the HIP module or touched a GPU (a process that has must never be replaced by another program), and leave with their
    exit code: non-zero as soon as one rank fails."""
    import subprocess
    cmd = launch_command(args, argv, free_port())
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench.py: --gpus %d without WORLD_SIZE: starting the ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def kernel_build_id():
    """Identifies the kernel sources and their build flags: sha256 over hexray_amd/csrc/* and build.sh (first 16 hex digits).  A committed
    PMC profile carries the id it was taken on; bench.py uses it only when the ids match."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hexray_amd", "csrc")
    for name in sorted(os.listdir(d)) + ["../../build.sh"]:
        path = os.path.normpath(os.path.join(d, name))
        if os.path.isfile(path):
            h.update(name.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_child(args):
    """What one counter pass runs under rocprofv3: a few frames of the workload through the C ABI into a host buffer (no torch)."""
    import hexray_amd as hx
    scene_name, ov, _ = WORKLOADS[args.workload]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    hx.init(0)
    ds = hx.DeviceScene(hs, accel=hx.ACCEL_SAH if args.accel == "sah" else hx.ACCEL_REFERENCE)
    params = hx.frame_params(tile_size=TILE)
    for _ in range(PMC_CHILD_FRAMES):
        ds.render(params)
    print("pmc child: last kernel ms %.4f" % ds.last_kernel_ms())


def live_pmc(args):
    """The counters of the render kernel for this workload, collected NOW: one `rocprofv3 --pmc` child process per pass of PMC_PASSES, started
    before this process has loaded torch or touched the GPU.  Returns {counter: value of the last render-kernel dispatch} or None with a reason."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # never under another profiler: a counter pass started from a traced process would inherit its tool library and combine tracing with --pmc
    profiler_env = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_", "ROCTX", "ROCPROFILER"))]
    if profiler_env or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler (%s): no nested counter passes" % ", ".join(profiler_env[:3] or ["LD_PRELOAD"])
    out = {}
    work = tempfile.mkdtemp(prefix="hxr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for tag, counters in PMC_PASSES:
            d = os.path.join(work, tag)
            cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                                                      "--workload", args.workload, "--accel", args.accel]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=90)
            except subprocess.TimeoutExpired:
                return None, "pass %s timed out" % tag
            if r.returncode != 0:
                return None, "pass %s: rocprofv3 exit %d: %s" % (tag, r.returncode, r.stdout.decode(errors="replace")[-300:].replace("\n", " | "))
            rows = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "hxr_render_kernel" in row.get("Kernel_Name", ""):
                        key = (row["Counter_Name"], int(row["Dispatch_Id"]))
                        rows[key] = rows.get(key, 0.0) + float(row["Counter_Value"])
            if not rows:
                return None, "pass %s: no render-kernel rows in the counter CSV" % tag
            last = max(k[1] for k in rows)
            for (name, dispatch), v in rows.items():
                if dispatch == last:
                    out[name] = v
    finally:
        shutil.rmtree(work, ignore_errors=True)
    missing = [c for _, cs in PMC_PASSES for c in cs if c not in out]
    if missing:
        return None, "counters missing: %s" % " ".join(missing)
    return out, "live: %d rocprofv3 --pmc child passes of this bench.py run (%d frames each, last dispatch read)" % (len(PMC_PASSES), PMC_CHILD_FRAMES)


def committed_pmc(workload):
    """The committed summary of the same counters (profiles/pmc_live_<workload>.json, written by tools/profile_collect.py from the round's profile run),
    usable only when it was taken on the kernel sources of this checkout."""
    path = os.path.join(ROOT, "profiles", "pmc_live_%s.json" % workload)
    try:
        rec = json.load(open(path))
    except Exception:
        return None, "no committed profile for this workload"
    if rec.get("kernel_build") != kernel_build_id():
        return None, "committed profile %s is of kernel build %s, this checkout is %s: not used" % (os.path.relpath(path, ROOT), rec.get("kernel_build"), kernel_build_id())
    return rec["counters"], "committed profile %s (kernel build %s)" % (os.path.relpath(path, ROOT), rec["kernel_build"])


def roofline_record(pmc, source, kernel_ms, abytes, samples_per_frame):
    """The bench line's `roofline`: FP64 vector-ALU issue (what binds), with HBM traffic and SURVEY.md 8d's algorithmic-bytes line beside it."""
    ksec = kernel_ms * 1e-3
    rec = {"bound": "fp64-valu", "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
           "kernel": "hxr_render_kernel<false>", "kernel_ms": round(kernel_ms, 4), "source": source, "kernel_build": kernel_build_id()}
    if pmc:
        fma, mul, add = pmc["SQ_INSTS_VALU_FMA_F64"], pmc["SQ_INSTS_VALU_MUL_F64"], pmc["SQ_INSTS_VALU_ADD_F64"]
        lane_util = pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0)
        nominal = (2 * fma + mul + add) * 64.0          # flop per launch if every lane of every FP64 wave-instruction did useful work
        useful = nominal * lane_util                     # ... scaled by the lanes that were switched on (all VALU instructions: the FP64 ones are not counted apart)
        traffic = (2 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0   # KB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
        simd_cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0              # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs; SQ_ACTIVE_INST_* count quad-cycles
        rec.update({
            "achieved": round(useful / ksec / 1e12, 3), "frac": round(useful / ksec / 1e12 / FP64_VALU_PEAK_TFLOPS, 4),
            "traffic": int(traffic),
            "fp64_tflops_nominal": round(nominal / ksec / 1e12, 3), "frac_nominal": round(nominal / ksec / 1e12 / FP64_VALU_PEAK_TFLOPS, 4),
            "valu_busy_frac": round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 4), "valu_lane_util": round(lane_util, 4),
            "salu_per_valu_instruction": round(pmc["SQ_INSTS_SALU"] / pmc["SQ_INSTS_VALU"], 3),
            "hbm_measured_frac": round(traffic / ksec / 1e9 / HBM_PEAK_GBS, 4),
            "pmc": {k: pmc[k] for k in sorted(pmc)},
            "note": "achieved = (2 FMA + MUL + ADD FP64 wave-instructions) x 64 x valu_lane_util / kernel time (HIP events of THIS run); "
                    "counters are per launch (last dispatch of a counter pass); traffic = (2 FETCH_SIZE + WRITE_SIZE) KB: register-spill scratch, the tables stay in cache",
        })
    if abytes is not None:
        achieved = abytes / ksec / 1e9
        rec["hbm_algorithmic"] = {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                                  "algorithmic_bytes_per_launch": int(abytes), "bytes_per_sample": round(abytes / samples_per_frame, 2),
                                  "note": "SURVEY.md 8d's algorithmic bytes (per-ray record sizes x this frame's work counters) priced against HBM peak: a side record -- "
                                          "the packet walk fetches a record once per WAVE from cache-resident tables, so this line does not bind and can exceed 1"}
    return rec


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if launch_command(args, argv, 0) is not None:      # FIRST thing: no torch, no HIP call before the ranks exist
        raise SystemExit(self_launch(args, argv))
    if args.pmc_child:
        pmc_child(args)
        return
    pmc, pmc_source = None, "not collected"
    if args.gpus == 1 and os.environ.get("WORLD_SIZE", "1") == "1" and not args.rendezvous_check:
        # counter passes as child processes, BEFORE torch or the HIP module is loaded here (a process that has touched the GPU starts no other program)
        if not args.no_live_pmc:
            pmc, pmc_source = live_pmc(args)
        if pmc is None:
            why = pmc_source
            pmc, pmc_source = committed_pmc(args.workload)
            pmc_source = "%s (live collection: %s)" % (pmc_source, why)

    import numpy as np
    import torch
    import torch.distributed as dist

    import hexray_amd as hx
    from hexray_amd.distributed import gather_tiles, padded_tiles_per_rank

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal on a box with fewer GPUs than ranks (HXR_BENCH_REHEARSAL=1): every rank renders on GPU 0 and the gather
    # goes through gloo on host copies.  RCCL refuses two ranks on one device, so this is the only way to run the
    # sharded pipeline end to end on a single-GPU box; the numbers it prints are not a measurement.
    rehearsal = os.environ.get("HXR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if args.rendezvous_check:
        # CPU test of the launch path (tests/test_abi_and_sharding.py): the ranks meet over gloo and count themselves
        if world > 1:
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            joined = dist.get_world_size()
            assert int(t.item()) == joined
            dist.destroy_process_group()
        else:
            joined = 1
        if rank == 0:
            print(json.dumps({"rendezvous_check": True, "n_gpus": joined, "asked": args.gpus}))
        if joined != args.gpus:
            raise SystemExit("bench.py: --gpus %d but %d ranks joined" % (args.gpus, joined))
        return
    if world > 1:
        if not rehearsal and torch.cuda.device_count() < world:   # counting devices does not initialise the GPU
            raise SystemExit("bench.py: %d ranks but %d GPUs visible (HXR_BENCH_REHEARSAL=1 runs every rank on GPU 0 over gloo)"
                             % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()          # the ranks that actually joined: this is what `n_gpus` reports
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) joined (WORLD_SIZE=%s): refusing to print a line whose n_gpus is not what was asked"
                         % (args.gpus, world, os.environ.get("WORLD_SIZE")))

    scene_name, ov, description = WORKLOADS[args.workload]
    t_prep = time.perf_counter()
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))   # parse + prepareTriangles + reference kd build + flatten
    host_prep_s = time.perf_counter() - t_prep
    hx.init(local_rank)
    torch.cuda.set_device(local_rank)
    accel = hx.ACCEL_SAH if args.accel == "sah" else hx.ACCEL_REFERENCE
    t_prep = time.perf_counter()
    ds = hx.DeviceScene(hs, accel=accel)
    device_prep_s = time.perf_counter() - t_prep        # the first call in a process also brings up the HIP runtime and loads the code objects
    t_prep = time.perf_counter()
    hx.DeviceScene(hs, accel=accel).close()
    device_prep_again_s = time.perf_counter() - t_prep
    W, H = hs.settings.frame_width, hs.settings.frame_height
    stream = torch.cuda.current_stream().cuda_stream

    probe = hx.frame_params()
    # resolve the samples per pixel the way render() does (reference src/main.cpp:421-424)
    spp = max(hs.camera.num_samples if hs.camera.dof else 0, hs.settings.num_paths if hs.settings.gi else 0)
    samples_per_frame = W * H * (spp if spp > 0 else 1)

    if world == 1:
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        params = hx.frame_params(tile_size=TILE)

        def step():
            ds.render_device(params, frame.data_ptr(), W * 12, stream)
    else:
        T = SHARD_TILE
        slots = padded_tiles_per_rank(W, H, T, world)
        local = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if rank == 0 else None
        landing = torch.empty(world * local.numel(), dtype=torch.float32, device="cuda") if rank == 0 and not rehearsal else None
        params = hx.frame_params(tile_size=T, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)

        def step():
            ds.render_device(params, local.data_ptr(), 0, stream)
            if rehearsal:
                torch.cuda.synchronize()
                got = gather_tiles(local.cpu(), world, rank)
                gathered = got.cuda() if rank == 0 else None
            else:
                gathered = gather_tiles(local, world, rank, out=landing)
            if rank == 0:
                ds.assemble_frame(params, gathered.data_ptr(), slots * T * T * 12, frame.data_ptr(), W * 12, stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, outside warm-up and timing: two frames that bring the process to its steady state whatever --warmup says -- the first
    # launch loads the code objects and faults the scene in (14 ms), the second is the first one scheduled from a cost map.  What a
    # first frame costs is reported on its own (other_timings.first_frame_kernel_ms).
    SETUP_FRAMES = 2
    for _ in range(SETUP_FRAMES):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if world == 1:
            pass
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # render-kernel duration (HIP events on the launch stream): measured over separate, individually
    # synchronised launches of the same frame so that every event pair is read back
    for _ in range(min(args.steps, 10)):
        step()
        torch.cuda.synchronize()
        kernel_ms.append(ds.last_kernel_ms())
    kernel_ms_mean = float(np.mean(kernel_ms))

    # work counters of one frame (instrumented kernel build, outside the timed region)
    cparams = hx.frame_params(tile_size=TILE, collect_counters=True)
    timed_equals_counted = None
    if world == 1:
        timed_frame = frame.clone()                       # what the timed kernel wrote
        ds.render_device(cparams, frame.data_ptr(), W * 12, stream)
        torch.cuda.synchronize()
        counters = ds.counters()
        timed_equals_counted = bool(torch.equal(timed_frame, frame))   # the timed variant did the counted variant's work: bit-equal frames
        del timed_frame
    else:
        counters = None

    if world > 1 and rank == 0 and os.environ.get("HXR_BENCH_VERIFY") == "1":
        # the sharded frame must equal the unsharded one bit for bit (random numbers are keyed on pixel and sample)
        whole = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        ds.render_device(hx.frame_params(tile_size=TILE), whole.data_ptr(), W * 12, stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(whole, frame))
        print("sharded frame identical to the single-GPU frame:", same, file=sys.stderr)
        if not same:
            raise SystemExit("sharded frame differs from the single-GPU frame")

    config5 = None
    if world > 1 and not args.no_config5 and args.workload == "meshes16":
        # BASELINE.json's 8-GPU configuration next to the headline line: data/smallpt.hexray, 1920x1080, 256 paths per pixel, the same
        # tile shards and the same single gather; 3 timed frames after one set-up frame (max over ranks, like the main line)
        config5 = sharded_frames(hx, torch, dist, "smallpt256", world, rank, rehearsal, stream, frames=3, setup=1)

    other = None
    if world == 1 and rank == 0 and not args.no_other_timings:
        other = other_timings(hx, hs, accel, W, H, stream, torch, np, min(args.steps, 10))

    result = None
    if rank == 0:
        value = samples_per_frame * args.steps / elapsed / 1e6
        result = {
            "metric": "Msamples/s (primary+secondary rays) at 1920x1080",
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "scene asset data/%s shipped with the reference (no external dataset); Monte-Carlo modes: frames agree with the CPU oracle run on the same "
                    "counter-based generator, agreement with the reference's own mt19937 statistics is not verifiable here (DESIGN.md section 4)%s"
                    % (scene_name, "; this scene's EXR cubemap is decoded by the module's own PIZ reader, whose texels are UNPINNED (no OpenEXR decoder in the image "
                                   "to compare with: DESIGN.md section 4)" if scene_name in ("simple.hexray", "heightfield.hexray") else ""),
            "config": {"workload": description, "width": W, "height": H, "spp": spp if spp > 0 else 1, "tile_size": TILE if world == 1 else SHARD_TILE,
                       "accel": "reference kd-tree (as handed over by the host)" if accel == hx.ACCEL_REFERENCE else "SAH kd-tree rebuilt by the module",
                       "parallelism": "tile-shard x%d, one RCCL gather" % world if world > 1 else "single GPU", "rng_seed": int(probe.rng_seed)},
        }
        if counters is not None:
            abytes = algorithmic_bytes(counters, W, H)
            rays = counters["raycasts"] + counters["shadow_queries"]
            result["roofline"] = roofline_record(pmc, pmc_source, kernel_ms_mean, abytes, samples_per_frame)
            result["timed_frame_equals_counted_frame"] = timed_equals_counted
            result["rays_per_sample"] = round(rays / counters["camera_samples"], 3)
            result["Mrays_per_s"] = round(rays / (kernel_ms_mean * 1e-3) / 1e6, 1)
        else:
            result["roofline"] = roofline_record(None, "per-rank kernel time of rank 0; counters are collected at N=1 only", kernel_ms_mean, None, samples_per_frame)
        if config5 is not None:
            result["config5_smallpt256"] = config5
        if other is not None:
            result["other_timings"] = other
            result["value_is"] = ("steady-state frames (schedule seeded by the previous identical frame; %d untimed set-up frames before the warm-up), "
                                  "device buffer; see other_timings for a first frame and host-buffer frames" % SETUP_FRAMES)
        # outside the timed region, reported separately: scene preparation
        result["scene_prep_s"] = {"host_parse_kd_build_flatten": round(host_prep_s, 3), "hxr_scene_create_first_call_in_process": round(device_prep_s, 3),
                                  "hxr_scene_create_and_destroy_again": round(device_prep_again_s, 4)}
        if world == 1 and not args.no_other_timings:
            # the same preparation with Mesh::buildKD and Heightfield::fillProperties (blur, box, normals, maxima)
            # left to the module, which runs them as kernels inside hxr_scene_create
            t_prep = time.perf_counter()
            bare = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov), gpu_kd_build=True, gpu_heightfield_blur=True, gpu_bitmap_prep=True)
            t_mid = time.perf_counter()
            hx.DeviceScene(bare, accel=hx.ACCEL_REFERENCE_DEVICE_BUILD).close()
            result["scene_prep_s"]["prepared_on_gpu"] = {"host_parse_flatten": round(t_mid - t_prep, 3),
                                                         "hxr_scene_create_and_destroy": round(time.perf_counter() - t_mid, 4)}
            bare.close()
        if world == 1 and accel == hx.ACCEL_REFERENCE and not args.no_sah_side:
            # SURVEY.md 8(f) rank 1, measured beside the headline: the same frame through the module's SAH kd-trees
            ref_frame = frame.clone()
            t_sah = time.perf_counter()
            alt = hx.DeviceScene(hs, accel=hx.ACCEL_SAH)      # the trees are built by kernels inside hxr_scene_create_ex (hxr_kd_build.hip)
            sah_create_s = time.perf_counter() - t_sah
            alt_ms = []
            for i in range(2 + min(args.steps, 10)):
                alt.render_device(params, frame.data_ptr(), W * 12, stream)
                torch.cuda.synchronize()
                if i >= 2:
                    alt_ms.append(alt.last_kernel_ms())
            diff = (frame - ref_frame).abs()
            alt.render_device(cparams, frame.data_ptr(), W * 12, stream)
            torch.cuda.synchronize()
            ac = alt.counters()
            result["sah_tree"] = {"kernel_ms": round(float(np.mean(alt_ms)), 4), "Msamples_per_s_kernel": round(samples_per_frame / (float(np.mean(alt_ms)) * 1e-3) / 1e6, 1),
                                  "reference_tree_kernel_ms": round(kernel_ms_mean, 4),
                                  "max_abs_diff_vs_reference_tree_frame": float(diff.max().item()), "pixels_differing": int((diff.amax(dim=2) > 0).sum().item()),
                                  "tri_tests_per_sample": round(ac["tri_tests"] / ac["camera_samples"], 2), "kd_visits_per_sample": round(ac["kd_visits"] / ac["camera_samples"], 2)}
            # SURVEY.md 8(f1), sanctioned side value: the same frame (0 pixels differ) through the module's own SAH trees, kernel time; the headline `value`
            # stays the reference tree's.  What building those trees costs is part of scene preparation, outside the timed region like the reference's beginRender.
            result["value_sah_tree"] = result["sah_tree"]["Msamples_per_s_kernel"]
            result["scene_prep_s"]["hxr_scene_create_with_sah_trees_built_on_device"] = round(sah_create_s, 4)
            del alt
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene_name, ov, samples_per_frame)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def sharded_frames(hx, torch, dist, workload, world, rank, rehearsal, stream, frames, setup):
    """`frames` frames of another workload through the N-rank pipeline of the main line (round-robin tile shards, one gather, assembly on
    rank 0), timed the same way: barrier + synchronize on both sides, max over ranks.  Returns the sub-record on every rank."""
    from hexray_amd.distributed import gather_tiles, padded_tiles_per_rank
    scene_name, ov, description = WORKLOADS[workload]
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    ds = hx.DeviceScene(hs, accel=hx.ACCEL_REFERENCE)
    W, H = hs.settings.frame_width, hs.settings.frame_height
    spp = max(hs.camera.num_samples if hs.camera.dof else 0, hs.settings.num_paths if hs.settings.gi else 0) or 1
    T = SHARD_TILE
    slots = padded_tiles_per_rank(W, H, T, world)
    local = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if rank == 0 else None
    landing = torch.empty(world * local.numel(), dtype=torch.float32, device="cuda") if rank == 0 and not rehearsal else None
    params = hx.frame_params(tile_size=T, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)

    def step():
        ds.render_device(params, local.data_ptr(), 0, stream)
        if rehearsal:
            torch.cuda.synchronize()
            got = gather_tiles(local.cpu(), world, rank)
            gathered = got.cuda() if rank == 0 else None
        else:
            gathered = gather_tiles(local, world, rank, out=landing)
        if rank == 0:
            ds.assemble_frame(params, gathered.data_ptr(), slots * T * T * 12, frame.data_ptr(), W * 12, stream)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(setup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(frames):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms = ds.last_kernel_ms()
    ds.close()
    hs.close()
    return {"workload": description, "frames": frames, "ms_per_frame": round(elapsed / frames * 1e3, 3),
            "Msamples_per_s": round(W * H * spp * frames / elapsed / 1e6, 2), "rank0_kernel_ms_last_frame": round(float(kernel_ms), 3),
            "tile_size": T, "parallelism": "tile-shard x%d, one gather" % world}


def other_timings(hx, hs, accel, W, H, stream, torch, np, frames):
    """What `value` leaves out (N = 1): a first frame without history, the steady state without the adaptive schedule, and
    host-buffer frames through hxr_render() (the entry INTEGRATION.md tells a host to call)."""
    out = {}
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    params = hx.frame_params(tile_size=TILE)
    first = []
    for _ in range(3):                     # three fresh scene handles: kernels are loaded, buffers are not, no cost map exists
        cold = hx.DeviceScene(hs, accel=accel)
        cold.render_device(params, frame.data_ptr(), W * 12, stream)
        torch.cuda.synchronize()
        first.append(cold.last_kernel_ms())
        cold.close()
    out["first_frame_kernel_ms"] = round(float(np.median(first)), 4)
    plain = hx.DeviceScene(hs, accel=accel, adaptive_schedule=False)
    ms = []
    for i in range(2 + frames):
        plain.render_device(params, frame.data_ptr(), W * 12, stream)
        torch.cuda.synchronize()
        if i >= 2:
            ms.append(plain.last_kernel_ms())
    out["no_adaptive_schedule_kernel_ms"] = round(float(np.mean(ms)), 4)
    plain.close()
    warm = hx.DeviceScene(hs, accel=accel)
    for _ in range(3):
        warm.render(params)
    t0 = time.perf_counter()
    for _ in range(frames):
        warm.render(params)                # blocking: kernel + 24.9 MB copy into a host array per frame
    out["host_buffer_ms_per_frame"] = round((time.perf_counter() - t0) / frames * 1e3, 4)
    out["host_buffer_kernel_ms"] = round(float(warm.last_kernel_ms()), 4)
    warm.close()
    # the same frames with secondary rays deferred to the continuation pass (hxr_scene_options.deferred_rays)
    try:
        dq = hx.DeviceScene(hs, accel=accel, deferred_rays=True)
        rows = []
        for i in range(2 + min(frames, 5)):
            dq.render_device(params, frame.data_ptr(), W * 12, stream)
            torch.cuda.synchronize()
            if i >= 2:
                rows.append(dq.frame_timing())
        if rows and rows[-1]["deferred_rays"] > 0:
            out["deferred_rays"] = {k: round(float(np.mean([r[k] for r in rows])), 4) for k in ("total_ms", "render_ms", "continue_ms", "resolve_ms")}
            out["deferred_rays"]["rays"] = rows[-1]["deferred_rays"]
        dq.close()
    except hx.HipError:
        pass
    return out


def usable_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(scene_name, ov, samples_per_frame):
    """The CPU oracle (a port of the reference's algorithm; oracle/hxo_oracle.c) on this host's cores: the
    workload frame (at reduced resolution when the full frame would take minutes), one warm-up plus repeats
    until about 10 s of CPU wall time have been spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hexray_amd as hx
    from oracle_binding import oracle_lib, oracle_render
    threads = min(usable_cpus(), oracle_lib().hxo_max_threads())
    ov = dict(ov)
    what = "the whole workload frame"
    if samples_per_frame > 40_000_000:   # e.g. 256 paths per pixel: minutes on a CPU
        ov["width"], ov["height"] = ov["width"] // 4, ov["height"] // 4
        samples_per_frame //= 16
        what = "the workload at %dx%d (1/16 of the pixels, same spp)" % (ov["width"], ov["height"])
    hs = hx.HostScene(os.path.join(ROOT, "data", scene_name), hx.overlay_text(**ov))
    oracle_render(hs, hx.frame_params(), threads=threads)   # warm-up (thread pool, page faults)
    times = []
    t_all = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_all < 12.0 or len(times) < 2):
        t = time.perf_counter()
        oracle_render(hs, hx.frame_params(), threads=threads)
        times.append(time.perf_counter() - t)
    med = sorted(times)[len(times) // 2]
    return {"value": round(samples_per_frame / med / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%s, %d runs after one warm-up, median %.3f s (min %.3f, max %.3f); gcc -O3 -ffast-math -march=haswell, OpenMP"
                      % (what, len(times), med, min(times), max(times))}


if __name__ == "__main__":
    main()
