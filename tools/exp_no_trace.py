import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import hexray_amd as hx
hx.init(0)
ROOT="/root/repo"
hs = hx.HostScene(os.path.join(ROOT, "data", "meshes.hexray"), hx.overlay_text(width=1920, height=1080, dof=True, num_samples=16))
st = torch.cuda.current_stream().cuda_stream
frame = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
p = hx.frame_params(tile_size=64)
ds = hx.DeviceScene(hs)
ms=[]
for i in range(10):
    ds.render_device(p, frame.data_ptr(), 1920 * 12, st); torch.cuda.synchronize(); ms.append(ds.last_kernel_ms())
print(os.environ.get("HXR_HIP_LIB","shipped"), " ".join("%.3f" % m for m in ms))
