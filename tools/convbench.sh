# build + run tools/convbench (isolated 64->64 conv kernels with phase counters)
set -e
cd /root/repo/tools
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEX_TIMING -I../include -I../dex_tts_amd/csrc -o convbench convbench.hip ../dex_tts_amd/csrc/conv3x3_stream.hip ../dex_tts_amd/csrc/conv3x3_bf16.hip ../dex_tts_amd/csrc/conv3x3_regw.hip 2>/dev/null
