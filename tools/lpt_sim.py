#!/usr/bin/env python3
"""Longest-first block order from a temporal cost map, simulated (tools/drain_sim.py wave-level model on the real
per-ray sample counts of C1): blocks of a launch ordered by the longest ray of the same block in the previous launch,
by its true longest ray, and with the map dilated.  Result: profiles/r06_lpt_sim.jsonl -- and the chip's answer to the
same question, which is a different one: profiles/r06_cost_order.jsonl (EXPERIMENTS.md round 6).
    python tools/lpt_sim.py > profiles/r06_lpt_sim.jsonl     (CPU, ~10 min)"""
import sys, json, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import drain_sim as ds
ds.A, ds.B = 1900.0, 0.6
from oracle import binding as ob
from volrend_amd import synth
cfg = synth.CONFIGS["C1"]; tree = synth.make_config_tree("C1"); th = ob.TreeHandle(tree)
W,H,focal = cfg["width"],cfg["height"],cfg["focal"]; poses = synth.make_poses(200)
cache={}
def fb(pi):
    if pi not in cache:
        tr = synth.c2w_to_transform(poses[pi%200])
        s,_,_ = ob.render_maps(th, ob.make_camera(tr,W,H,focal), ob.default_options())
        cache[pi]=s.reshape(H//8,8,W//8,8).transpose(0,2,1,3).reshape(-1, 64).astype(np.int32)
    return cache[pi]
def run(frames, order):
    rays = np.concatenate([fb(f)[b] for b in order for f in frames])
    rays = rays[rays>0]
    c,_ = ds.run_phase(rays, 5120, "none", 0, 10**9)
    return c/2.2/1e3
nb = 10000
for first,nf,lag in ((5,1,1),(24,1,1),(60,1,1),(100,1,4),(150,1,1),(5,2,2),(5,4,4),(40,4,4),(5,20,5)):
    frames = list(range(first, first+nf))
    prev = [f-lag for f in frames]
    nat = np.arange(nb)
    cost_true = np.max([fb(f).max(1) for f in frames],axis=0)
    cost_prev = np.max([fb(f).max(1) for f in prev],axis=0)
    res = {"first":first,"frames":nf,"prev_lag":lag,"longest":int(cost_true.max()),
           "natural_us": round(run(frames, nat),1),
           "by_prev_max_us": round(run(frames, np.argsort(-cost_prev, kind="stable")),1),
           "by_prev_max_16buckets_us": round(run(frames, np.argsort(-(np.minimum(cost_prev,255)//16), kind="stable")),1),
           "by_true_max_us": round(run(frames, np.argsort(-cost_true, kind="stable")),1)}
    print(json.dumps(res), flush=True)
print("--- dilation", flush=True)
from scipy.ndimage import maximum_filter
for first,nf,lag in ((5,4,4),(40,4,4),(5,2,2),(5,1,1),(24,1,1),(100,1,4),(5,8,8)):
    frames = list(range(first, first+nf)); prev=[f-lag for f in frames]
    cost_prev = np.max([fb(f).max(1) for f in prev],axis=0).reshape(100,100)
    res={"first":first,"frames":nf,"lag":lag}
    for r in (0,1,2,4,8):
        c = maximum_filter(cost_prev, size=(1,2*r+1)) if r else cost_prev
        res[f"dil_h{r}"] = round(run(frames, np.argsort(-c.reshape(-1), kind="stable")),1)
    c = maximum_filter(cost_prev, size=(5,9)); res["dil_5x9"] = round(run(frames, np.argsort(-c.reshape(-1), kind="stable")),1)
    # per (block, frame) ordering by the previous launch's same-index frame cost (frame f predicted by frame f - lag), dilated
    print(json.dumps(res), flush=True)
