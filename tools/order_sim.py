#!/usr/bin/env python3
"""Ray order of a multi-frame launch against its tail (simulation): tools/drain_sim.py run_phase on the real per-ray
sample counts of consecutive C1 poses, ids block major / frame minor, blocks visited in the natural order, rows / columns
from the image centre outwards, radially, and by their true longest ray.  python tools/order_sim.py 20 > profiles/r06_order_sim.jsonl"""
import sys, json, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import drain_sim as ds
ds.A, ds.B = 1900.0, 0.6
from oracle import binding as ob
from volrend_amd import synth
cfg = synth.CONFIGS["C1"]; tree = synth.make_config_tree("C1"); th = ob.TreeHandle(tree)
W,H,focal = cfg["width"],cfg["height"],cfg["focal"]; poses = synth.make_poses(200)
NF = int(sys.argv[1]) if len(sys.argv)>1 else 20
def frame_blocks(pi):
    tr = synth.c2w_to_transform(poses[pi%200])
    s,_,_ = ob.render_maps(th, ob.make_camera(tr,W,H,focal), ob.default_options())
    return s.reshape(H//8,8,W//8,8).transpose(0,2,1,3).reshape(H//8, W//8, 64).astype(np.int32)
fb = np.stack([frame_blocks(5+i) for i in range(NF)])  # [NF, 100, 100, 64]
nby, nbx = fb.shape[1], fb.shape[2]
def centre_out(n):
    mid = n//2; out=[]
    for r in range(n):
        out.append(mid + ((r+1)//2 if r%2 else -(r//2)))
    return [o for o in out if 0<=o<n]
def order_rays(kind):
    # block major, frame minor
    if kind=="natural": blocks = [(y,x) for y in range(nby) for x in range(nbx)]
    elif kind=="rows_centre_out": blocks = [(y,x) for y in centre_out(nby) for x in range(nbx)]
    elif kind=="stripes_cols_centre_out":
        S=(nby+7)//8; blocks=[]
        for s0 in range(0,nby,S):
            for x in centre_out(nbx):
                for y in range(s0,min(s0+S,nby)): blocks.append((y,x))
    elif kind=="radial":
        cy,cx=(nby-1)/2,(nbx-1)/2
        blocks = sorted([(y,x) for y in range(nby) for x in range(nbx)], key=lambda b:(b[0]-cy)**2+(b[1]-cx)**2)
    elif kind=="true_max":
        mx = fb.max(axis=(0,3))
        blocks = sorted([(y,x) for y in range(nby) for x in range(nbx)], key=lambda b:-mx[b])
    rays = np.concatenate([fb[f,y,x] for (y,x) in blocks for f in range(NF)])
    return rays[rays>0]
for kind in ("natural","rows_centre_out","stripes_cols_centre_out","radial","true_max"):
    rays = order_rays(kind)
    c,_ = ds.run_phase(rays, 5120, "none", 0, 10**9)
    ideal = rays.sum()/ (5120*64)
    print(json.dumps({"frames":NF,"order":kind,"us":round(c/2.2/1e3,1),"us_per_frame":round(c/2.2/1e3/NF,1),"rays":len(rays)}), flush=True)
