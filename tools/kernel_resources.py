#!/usr/bin/env python3
"""Print VGPR/AGPR/SGPR/spill/scratch/LDS/occupancy per kernel of the HIP sources (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "figaroh_plus_amd", "csrc")
files = sys.argv[1:] or ["figh_regressor.hip", "figh_linalg.hip"]
for f in files:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(root, "include"),
                        "-I" + csrc, "-ffp-contract=fast", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(csrc, f), "-o", "/dev/null"],
                       capture_output=True, text=True)
    cur = None
    rows = {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark: .*?:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void figh::", "")
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for k, r in rows.items():
        print("%-50s VGPR %4s AGPR %4s SGPR %4s spillS %4s spillV %4s scratch %6s LDS %6s occ %s" % (
            k[:50], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
