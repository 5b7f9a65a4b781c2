"""Round 4: the 8-way shard bound of the headline frame (slowest rank, sum of ranks; steady state, all on this GPU one rank at a time) under schedule settings.
usage: python3 tools/exp_shard_sched.py [world]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import hexray_amd as hx
from hexray_amd.distributed import padded_tiles_per_rank
hx.init(0)
hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(os.environ.get("HXR_TILE", "8"))
slots = padded_tiles_per_rank(1920, 1080, T, world)
buf = torch.zeros(slots * T * T * 3, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def bound(env):
    for k, v in env.items(): os.environ[k] = v
    ds = hx.DeviceScene(hs)
    per_rank = []
    for rank in range(world):
        p = hx.frame_params(tile_size=T, shard_rank=rank, shard_count=world, layout=hx.LAYOUT_TILES)
        ms = []
        for i in range(8):
            ds.render_device(p, buf.data_ptr(), 0, st); torch.cuda.synchronize()
            ms.append(ds.last_kernel_ms())
        per_rank.append(float(np.median(ms[3:])))
    ds.close()
    for k in env: del os.environ[k]
    return per_rank
cases = [("default", {})]
if os.environ.get("EXP_PARTS"):   # how many replicating blocks, what they take
    cases = [("default", {})] + [("parts grid %s" % g, {"HXR_PARTS_GRID": g}) for g in ("64", "96", "192", "256", "384")] + [("no parts kernel", {"HXR_NO_PARTS_KERNEL": "1"}), ("first_x 4, split_x 8", {"HXR_SCHED_SPLIT_X": "8"}), ("resident 16", {"HXR_SCHED_RESIDENT": "16"}), ("resident 8", {"HXR_SCHED_RESIDENT": "8"})]
elif os.environ.get("EXP_INSTANCE"):   # throughput instance (four waves per SIMD) against the latency instance (three, touch prefetch, replicating blocks, priorities)
    cases = [("four waves", {"HXR_MESH_MC_FOUR": "1"}), ("three waves", {"HXR_MESH_MC_FOUR": "0"}), ("three waves, first_x 4", {"HXR_MESH_MC_FOUR": "0", "HXR_SCHED_FIRST_X": "4"}),
             ("four waves, first_x 4", {"HXR_MESH_MC_FOUR": "1", "HXR_SCHED_FIRST_X": "4"}), ("three waves, priority", {"HXR_MESH_MC_FOUR": "0", "HXR_FRONT_PRIORITY": "1"}),
             ("three waves, priority, first_x 4", {"HXR_MESH_MC_FOUR": "0", "HXR_FRONT_PRIORITY": "1", "HXR_SCHED_FIRST_X": "4"})]
elif os.environ.get("EXP_FIRST_X_ONLY"):
    cases += [("first_x %s" % fx, {"HXR_SCHED_FIRST_X": fx}) for fx in ("5", "4", "3")]
else:
  for fx in ("4", "3", "2", "1"): cases.append(("first_x %s" % fx, {"HXR_SCHED_FIRST_X": fx}))
if not os.environ.get("EXP_FIRST_X_ONLY"):
 for sx in ("8", "4"): cases.append(("split_x %s" % sx, {"HXR_SCHED_SPLIT_X": sx}))
 for fd in ("2", "4"): cases.append(("frame_div %s" % fd, {"HXR_SCHED_FRAME_DIV": fd}))
 cases.append(("first_x 2, frame_div 2", {"HXR_SCHED_FIRST_X": "2", "HXR_SCHED_FRAME_DIV": "2"}))
for name, env in cases + cases[:1]:
    r = bound(env)
    print("%-24s slowest %.3f  sum %.2f   ranks %s" % (name, max(r), sum(r), " ".join("%.3f" % v for v in r)), flush=True)
