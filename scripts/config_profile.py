"""One align() of one named case (cases.<name>, optional n), for rocprofv3 --kernel-trace --stats.
usage: config_profile.py config1|config2|config3|config4|scene [n] [max_iterations]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import cases
from unified_cvo_amd import CvoGPU
name = sys.argv[1]
kw = dict(n=int(sys.argv[2])) if len(sys.argv) > 2 and name != "config1" else {}
its = int(sys.argv[3]) if len(sys.argv) > 3 else 0
P, src, tgt, init = getattr(cases, name)(**kw)
g = CvoGPU(params=P)
da, db = g.upload(src), g.upload(tgt)
g.align(da, db, init, max_iterations=20)
g.set_option("VERBOSE", os.environ.get("PROBE_VERBOSE", "1"))
r = g.align(da, db, init, max_iterations=its)
print(name, kw, r.iterations, r.ret, f"{r.seconds*1e3:.2f} ms, {r.seconds*1e6/r.iterations:.1f} us/it", g.debug_list_builds(), flush=True)
