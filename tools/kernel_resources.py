"""Development: register / scratch / LDS / occupancy of every kernel of one csrc file (hipcc -Rpass-analysis), one row each.
usage: python tools/kernel_resources.py csrc/gemm.hip [extra hipcc flags]   (cross-compiles, no GPU needed)"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                     ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*", "", name)
    print("%-70s vgpr %3d agpr %3d scratch %4d lds %6d occ %d" % (name[:70], r.get("vgpr", -1), r.get("agpr", -1),
                                                                  r.get("scratch", -1), r.get("lds", -1), r.get("occ", -1)))
