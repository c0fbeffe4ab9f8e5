#!/usr/bin/env python3
"""Copy one gpurun_out/<tag>/ profile set (tools/profile_round5.sh) into profiles/ under the round's prefix:
    python tools/collect_profiles.py gpurun_out/round5b round5
pmc_traffic.json keeps its name (bench.py reads it), mfma_util.json becomes <prefix>_mfma_util.json, everything else <prefix>_<file>."""
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
for f in sorted(os.listdir(src)):
    if f.endswith(".err"):
        continue
    name = f if f == "pmc_traffic.json" else f"{prefix}_{f}"
    shutil.copy(os.path.join(src, f), os.path.join(dst, name))
    print(name)
