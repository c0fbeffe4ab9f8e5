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
sys.path.insert(0, root)
from dex_tts_amd import synth  # noqa: E402
import json  # noqa: E402
import subprocess  # noqa: E402
head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
with open(os.path.join(dst, "profiles_head.json"), "w") as fh:      # what the spliced rocprof / PMC numbers of the bench line were measured on
    json.dump({"round_prefix": prefix, "commit_at_collection": head, "kernel_sources_sha": synth.kernel_sources_sha(root),
               "note": "kernel_sources_sha = sha1 of dex_tts_amd/csrc at collection time; bench.py reports profiles_stale when the tree differs"}, fh, indent=1)
for f in sorted(os.listdir(src)):
    if f.endswith(".err"):
        continue
    name = f if f == "pmc_traffic.json" else f"{prefix}_{f}"
    shutil.copy(os.path.join(src, f), os.path.join(dst, name))
    print(name)
