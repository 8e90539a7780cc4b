#!/usr/bin/env python
"""Registers / LDS / occupancy of every gfx950 kernel of the backend (hipcc -Rpass-analysis=kernel-resource-usage).
CPU only: hipcc cross-compiles.  usage: kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unified_cvo_amd import build as B  # noqa: E402

flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared",)]
cmd = [B._hipcc()] + flags + os.environ.get("CVO_EXTRA_HIPCC_FLAGS", "").split() + ["-I", os.path.join(ROOT, "include"), "-c", B.sources()[0], "-o", "/tmp/_kres.o",
                              "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    if flt in k:
        print(f"{k:58s} VGPRs {v.get('VGPRs', '?'):>4} AGPRs {v.get('AGPRs', '?'):>3} SGPRs {v.get('TotalSGPRs', '?'):>4} "
              f"occupancy {v.get('Occupancy', '?'):>2} LDS {v.get('LDS Size', '?'):>6} scratch {v.get('ScratchSize', '?')}")
