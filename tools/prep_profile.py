import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from sparc_amd import synth
from sparc_amd.graph import ClusterBatch
dev = torch.device("cuda:0")
shape = synth.SHAPES["amazon2m"]
graph, bounds = synth.planted_cluster_graph(int(shape["n"] * 0.1), int(shape["edges"] * 0.1), 500, dev, seed=0)
nodes_of, cluster_ids = synth.cluster_batches(bounds, 50, seed=0, return_clusters=True)
def prep(b):
    coords, values, shp, _ = graph.subblock(nodes_of[b], mode=2, binarize=False, diag_lambda=0.01)
    csz = np.diff(bounds.cpu().numpy())[cluster_ids[b]]
    cb = ClusterBatch.from_device_coo(shp[0], coords, values, cluster_sizes=csz).warm(47, measure=True)
    torch.cuda.synchronize()
    return cb
prep(0)
t0 = time.perf_counter(); prep(1); print("one batch", time.perf_counter() - t0)
pr = cProfile.Profile(); pr.enable(); prep(2); prep(3); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
