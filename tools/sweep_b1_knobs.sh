cd /root/repo
bash tools/ab_e2e.sh DEX_LINATTN_NSUB "0 1 2 3 4" gedex_b1 > /dev/null 2>&1; cp gpurun_out/e2e_DEX_LINATTN_NSUB.txt gpurun_out/sw1.txt
bash tools/ab_e2e.sh DEX_CONV_SMALL_MAX "256 64 1024" gedex_b1 > /dev/null 2>&1; cat gpurun_out/e2e_DEX_CONV_SMALL_MAX.txt >> gpurun_out/sw1.txt
bash tools/ab_e2e.sh DEX_CONV_STREAM "1 2" gedex_b1 > /dev/null 2>&1; cat gpurun_out/e2e_DEX_CONV_STREAM.txt >> gpurun_out/sw1.txt
bash tools/ab_e2e.sh DEX_CONV_PP "1 2" gedex_b1 > /dev/null 2>&1; cat gpurun_out/e2e_DEX_CONV_PP.txt >> gpurun_out/sw1.txt
bash tools/ab_e2e.sh DEX_PATCH_FUSED "1 0" gedex_b1 > /dev/null 2>&1; cat gpurun_out/e2e_DEX_PATCH_FUSED.txt >> gpurun_out/sw1.txt
bash tools/ab_e2e.sh DEX_GEMM_NWALK "1 0" gedex_b1 > /dev/null 2>&1; cat gpurun_out/e2e_DEX_GEMM_NWALK.txt >> gpurun_out/sw1.txt
cat gpurun_out/sw1.txt
