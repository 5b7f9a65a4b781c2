import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, hexray_amd as hx
hx.init(0)
hs = hx.HostScene('tests/golden/all_meshes.hexray')
for mi in (7, 8):
    nodes, leaf, depth, ms = hx.kd_build_device(hs, mi)
    np.save('gpurun_out/kd_nodes_%d.npy' % mi, nodes); np.save('gpurun_out/kd_leaf_%d.npy' % mi, leaf)
    print(mi, len(nodes), len(leaf), depth, ms)
for mi in range(9):
    t = [hx.kd_build_device(hs, mi)[3] for _ in range(3)]
    print("mesh", mi, "build ms", t)
