#!/usr/bin/env python3
"""Register / scratch / occupancy of the hot kernels as hipcc reports them (-Rpass-analysis=kernel-resource-usage), without a GPU:
  tools/kernel_resources.py [extra -D flags]     -> one line per instantiation of the scan, sweep and raycast kernels"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value",
           "-Rpass-analysis=kernel-resource-usage", "-I" + os.path.join(ROOT, "include"), "-o", "/tmp/_se_res.so"] + sys.argv[1:] + [os.path.join(ROOT, "supereight_amd", "csrc", "se_hip_api.hip")]
    t = subprocess.run(cmd, capture_output=True, text=True).stderr
    names = []
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
        name = b.split()[0]
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
        rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
        names.append(name)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for (name, v, a, s, sc, occ, lds), d in zip(rows, dem):
        d = re.sub(r"\(.*", "", d).replace("void ", "")
        if any(k in d for k in ("k_alloc_scan", "k_raycast", "k_integrate", "k_icp", "k_alloc_commit", "k_occ_commit")):
            print(f"{d:<60} VGPR {v:>3} AGPR {a:>3} SGPR {s:>3} scratch {sc:>4} occ {occ} lds {lds}")


if __name__ == "__main__":
    main()
