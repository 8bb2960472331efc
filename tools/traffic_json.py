#!/usr/bin/env python3
"""profiles/rNN_traffic.json from a tools/pmc.sh pass directory (groups fetch write rdsize
tcc sq1 are needed).   tools/traffic_json.py gpurun_out/pmc_<tag> <frames_per_launch> <summary path>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KB;
read bytes are taken from the request-size histogram (RDREQ_128B*128 + _64B*64 + _32B*32)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, frames = sys.argv[1], int(sys.argv[2])
source = sys.argv[3] if len(sys.argv) > 3 else d
kernel = sys.argv[4] if len(sys.argv) > 4 else "render_kernel<0, 16, 0>"
vals = defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    per = defaultdict(float)
    for row in csv.DictReader(open(f)):
        if kernel in row["Kernel_Name"]:
            per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (_, name), v in per.items():
        vals[(os.path.basename(f).split("_")[0], name)].append(v)


def mean(group, name):
    v = vals[(group, name)]
    return sum(v) / len(v)


rd = (mean("rdsize", "TCC_EA0_RDREQ_128B_sum") * 128 + mean("rdsize", "TCC_EA0_RDREQ_64B_sum") * 64
      + mean("rdsize", "TCC_EA0_RDREQ_32B_sum") * 32)
wr = mean("write", "WRITE_SIZE") * 1024
fetch = mean("fetch", "FETCH_SIZE") * 1024
out = {
    "config": "C1", "fp_mode": "strict", "frames_per_launch": frames,
    "kernel": f"vr::{kernel} (FAST, SH16)",
    "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
    "read_bytes_per_frame": rd / frames, "write_bytes_per_frame": wr / frames,
    "FETCH_SIZE_KB_per_launch": fetch / 1024, "fetch_size_undercount": fetch / rd,
    "method": "rocprofv3 --pmc, separate passes (tools/pmc.sh groups rdsize, write, fetch): read "
              "bytes = TCC_EA0_RDREQ_128B*128 + _64B*64 + _32B*32 (every request of this kernel "
              "is 128 B, so FETCH_SIZE = RDREQ*64 B under-counts by 2x, the gfx950 correction of "
              "MI355X_MICROARCH.md); write bytes = WRITE_SIZE (KB, uncalibrated, <1 % of the total)",
    "l2_hit_rate": mean("tcc", "TCC_HIT_sum") / mean("tcc", "TCC_REQ_sum"),
    "valu_insts_per_frame": mean("sq1", "SQ_INSTS_VALU") / frames,
    "valu_lane_utilisation": mean("sq1", "SQ_THREAD_CYCLES_VALU") /
                             (64 * mean("sq1", "SQ_ACTIVE_INST_VALU")),
    "wave_wait_fraction": mean("sq1", "SQ_WAIT_ANY") / mean("sq1", "SQ_WAVE_CYCLES"),
    "source": source,
}
print(json.dumps(out, indent=1))
