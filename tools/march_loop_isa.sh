#!/bin/bash
# ISA of the march round of the fused SH16 kernel (render_kernel<strict, SH16, FAST>): the loop from its
# header to the first shade round, with instruction counts by unit.  Static counts -- the descent loop below the
# lookup structure and the guard's rarely taken blocks are in there -- next to the measured per-frame totals of
# profiles/r03_traffic_C1.json (98 M vector, 41 M scalar wave-instructions).
#   bash tools/march_loop_isa.sh > profiles/r03_march_loop_isa.txt
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -x hip \
  -I include -I volrend_amd/csrc -S --cuda-device-only volrend_amd/csrc/vr_kernels.hip -o /tmp/vr_isa.s 2>/dev/null
awk '/^_ZN2vr12_GLOBAL__N_113render_kernelILi0ELi16ELi0ELb0EEEvNS_7KParamsE:/,/s_endpgm/' /tmp/vr_isa.s > /tmp/vr_isa_k16.s
python3 - <<'PY'
import re
lines = open('/tmp/vr_isa_k16.s').read().split('\n')
hdr = [i for i, l in enumerate(lines) if 'Loop Header: Depth=2' in l and 'Inner' not in l][0]
d3 = [i for i, l in enumerate(lines) if 'Depth=3' in l and i > hdr]
seg = [l for l in lines[hdr - 8: d3[1]] if l.strip() and not l.strip().startswith(';')]
cnt = {}
for l in seg:
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m:
        continue
    op = m.group(1)
    unit = ('VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') and not op.startswith(('s_waitcnt', 's_cbranch', 's_branch', 's_nop'))
            else 'branch' if op.startswith(('s_cbranch', 's_branch')) else 'wait/nop' if op.startswith('s_') else
            'LDS' if op.startswith('ds_') else 'VMEM')
    cnt[unit] = cnt.get(unit, 0) + 1
whole = sum(1 for l in lines if re.match(r'\s+[a-z]', l))
print("render_kernel<strict, SH16, FAST>: %d instructions in the kernel; march round (loop header .. first shade round): %s"
      % (whole, ", ".join(f"{k} {v}" for k, v in sorted(cnt.items()))))
print()
print('\n'.join(l.split(';')[0].rstrip() if not l.startswith('.L') else l for l in seg))
PY
