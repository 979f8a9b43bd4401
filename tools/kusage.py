#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one HIP source: hipcc -Rpass-analysis=kernel-resource-usage, parsed.
usage: python tools/kusage.py csrc/conv_mfma.hip [filter-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::|\(.*", "", name)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("vspill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':70s} vgpr agpr sgpr occ vspill scratch   lds")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('vgpr',0):4d} {r.get('agpr',0):4d} {r.get('sgpr',0):4d} {r.get('occ',0):3d} {r.get('vspill',0):6d} {r.get('scratch',0):7d} {r.get('lds',0):6d}")
