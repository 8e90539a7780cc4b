"""Generates tests/golden/demo_clouds.npz from the reference's demo_data/*.pcd (run in the build
container, where /root/reference exists).  The fixture is DATA: the parsed x y z and the unpacked
r g b bytes of the two ASCII PCD files (`FIELDS x y z rgb`, rgb stored as a U32), i.e. exactly what
pcl::io::loadPCDFile hands to the demo driver (main_cvo_gpu_align_two_color_pcd.cpp:46-53)."""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_clouds.npz")


def read_ascii_pcd(path):
    xyz, rgb = [], []
    with open(path) as f:
        data = False
        for line in f:
            if not data:
                if line.startswith("DATA"):
                    assert line.split()[1] == "ascii"
                    data = True
                continue
            a = line.split()
            if len(a) < 4:
                continue
            xyz.append([np.float32(a[0]), np.float32(a[1]), np.float32(a[2])])
            u = int(a[3])
            rgb.append([(u >> 16) & 255, (u >> 8) & 255, u & 255])
    return np.asarray(xyz, np.float32), np.asarray(rgb, np.uint8)


sx, sr = read_ascii_pcd(os.path.join(REF, "demo_data", "source.pcd"))
tx, tr = read_ascii_pcd(os.path.join(REF, "demo_data", "target.pcd"))
print(sx.shape, tx.shape)
np.savez_compressed(OUT, src_xyz=sx, src_rgb=sr, tgt_xyz=tx, tgt_rgb=tr)
print("wrote", OUT)
