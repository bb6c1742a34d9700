#!/usr/bin/env python3
"""Per-wave timeline of one inner pass of the windowed kernel (developer tool; needs SPARC_DBG=64):
start / barrier / end cycle stamps of every wave, summarised per workgroup."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparc_amd import synth
from sparc_amd._lib import lib
from sparc_amd.graph import ClusterBatch
dev = torch.device("cuda:0")
b, k = synth.shape_batch(os.environ.get("WORKLOAD", "amazon2m"), dev, bsize=int(os.environ.get("BSIZE", "0")) or None)
ClusterBatch.tile_vec = int(os.environ.get("VEC", "0")) or None
t = b.tiling(k, True)
Y = torch.randn(b.m, k, device=dev)
for _ in range(3):
    b.lap(Y, True)
buf = torch.zeros(t.ndesc * 16 * 4, dtype=torch.int64, device=dev)
lib.sparc_debug_stamps.argtypes = [ctypes.c_void_p]
lib.sparc_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
b.lap(Y, True)
torch.cuda.synchronize()
lib.sparc_debug_stamps(None)
s = buf.cpu().numpy().reshape(t.ndesc, 16, 4)
nw = t.threads // 64
if os.environ.get("DUMP"):   # raw stamps and descriptors for offline analysis
    np.savez(os.environ["DUMP"], stamps=s[:, :nw], desc=t.desc, slot_L=t.plan["slot_L"], slot_O=t.plan["slot_O"])
s = s[:, :nw]
live = s[:, 0, 3] > 0
s = s[live]
t0 = s[:, :, 0].min()
start = (s[:, :, 0] - t0); bar = (s[:, :, 1] - t0); end = (s[:, :, 2] - t0)
print(f"workgroups {live.sum()} of {t.ndesc}; counter ticks (100 MHz constant clock if s_memrealtime, else shader cycles)")
for name, v in (("start", start), ("barrier", bar), ("end", end), ("stage = barrier - start", bar - start), ("loop = end - barrier", end - bar)):
    print(f"{name:26s} min {v.min():8d}  p10 {int(np.percentile(v, 10)):8d}  median {int(np.median(v)):8d}  p90 {int(np.percentile(v, 90)):8d}  max {v.max():8d}")
slots = s[:, 0, 3]
per_wave_rounds = np.ceil(slots / nw)
loop_wg = (end - bar).max(axis=1)
print("loop ticks per round (median over workgroups):", float(np.median(loop_wg / per_wave_rounds)))
# second pass launched right after the first overwrites the stamps: the numbers are the second inner pass
order = np.argsort(s[:, 0, 0])
late = s[order[-20:], 0, 0] - t0
print("start of the 20 last workgroups:", late.tolist())
