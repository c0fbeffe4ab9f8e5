# build + run tools/rwbench (the weights-in-registers 128 -> 128 convolution alone, with phase counters)
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDEX_TIMING -fno-slp-vectorize $RW_DEFS -I../include -I../dex_tts_amd/csrc -o rwbench rwbench.hip ../dex_tts_amd/csrc/conv3x3_regw.hip 2>/dev/null || exit 1
./rwbench "$@"
