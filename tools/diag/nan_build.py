"""Diagnostic: one device builder on triangles with NaN / infinite vertices, step by step, progress appended to a log."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nexus_amd import capi, pod, scenegen

radius, log = int(sys.argv[1]), sys.argv[2]
kinds = sys.argv[3] if len(sys.argv) > 3 else "nan,inf,ninf"


def say(msg):
    with open(log, "a") as f:
        f.write("[r=%d %s] %.2f %s\n" % (radius, kinds, time.time() - T0, msg))


T0 = time.time()
tris = np.ascontiguousarray(scenegen.random_soup(4000, seed=12, extent=1.0, size=0.05), dtype=pod.TRI_DT)
bad = tris.copy()
rng = np.random.RandomState(3)
victims = rng.choice(len(bad), 40, replace=False)
if "nan" in kinds.split(","):
    bad["pos0"][victims[:15], 0] = np.nan
if "inf" in kinds.split(","):
    bad["pos1"][victims[15:30]] = np.inf
if "ninf" in kinds.split(","):
    bad["pos2"][victims[30:], 2] = -np.inf
ctx = capi.Context(32, 32)
ctx.set_device_builder(radius)
say("context ready")
bid = ctx.build_blas(bad)
say("build_blas returned")
nodes, idx = ctx.read_blas(bid, len(bad))
say("read_blas: %d nodes, permutation %s" % (len(nodes), sorted(idx.tolist()) == list(range(len(bad)))))
ident = np.eye(4, dtype=np.float32).reshape(16)
inst = np.array([capi.instance_init(bid, 0, ident, nodes[0])], dtype=pod.INST_DT)
inst["boundsMin"], inst["boundsMax"] = -4.0, 4.0
say("instance made, root p %s e %s" % (nodes[0]["p"], nodes[0]["e"]))
tn, ti = capi.tlas_build(inst)
say("host tlas built")
ctx.set_tlas(tn, ti, inst)
say("tlas set")
rays = scenegen.interior_rays(2000, seed=5, extent=1.0)
got = ctx.trace_batch(rays)
say("trace_batch returned: %d hits" % int((got["hitDistance"] < 1e29).sum()))
