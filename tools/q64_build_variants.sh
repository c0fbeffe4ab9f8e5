#!/bin/bash
# Build tools/attnq64_x_<name> stamp binaries for generator experiments: tools/q64_build_variants.sh "name:ENV=VAL ENV=VAL" ...
# (the committed .inc is generated with no Q64GEN_* variable set; run them all with tools/q64_variants.sh on the GPU box)
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  inc=/tmp/q64_$name.inc
  env Q64GEN_OUT=$inc $envs python tools/gen_attn_q64.py > /dev/null || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -DQ64_STAMP -DQ64_CORE_INC="\"$inc\"" -I dex_tts_amd/csrc -I include tools/attnq64.hip \
      -L dex_tts_amd/lib -ldexamd -Wl,-rpath,'$ORIGIN/../dex_tts_amd/lib' -o tools/attnq64_x_$name || exit 1
  echo built tools/attnq64_x_$name
done
