"""development: the multi-sequence forward kernel (k_fwdn) on a few windows of config C's shape: statuses, error sites, and bytes against the oracle"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from vechat_amd import capi
from vechat_amd.engine import HipContext
import oracle_api as oa
n = int(os.environ.get("DBG_N", "64")); reps = int(os.environ.get("DBG_REPS", "1"))
b = capi.synth_batch(capi.synth_cfg(1002, 500, 64), 0, n)
c = HipContext(device=0)
ref = None
for r in range(reps):
    cons, st = c.consensus(b, retry_overflow=False)
    bad = [w for w in range(n) if int(st[w]) > 1]
    if ref is None: ref, pol, ost = oa.oracle_run(b, c.params)
    diff = [w for w in range(n) if cons[w] != ref[w]]
    ei = c.errinfo()
    sites = {}
    for w in bad: sites[ei[w][0]] = sites.get(ei[w][0], 0) + 1
    print("rep", r, "bad status", len(bad), "of", n, "| bytes differ from the oracle:", len(diff), "| cells", c.stats()["cells"], ost.cells, "| sites", sites, "| first bad", bad[:8], [ei[w] for w in bad[:4]])
