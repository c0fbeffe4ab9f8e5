# Same-box end-to-end sweep of the batch launchers' grid knobs around their defaults (round 6): bash tools/sweep_batch_knobs.sh workload...
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/sweep_batch.txt; : > $O
for w in "$@"; do
  for kv in "DEX_CONVT_WGS:512 256 1024" "DEX_CONV_DOWN_WGS:512 256 1024" "DEX_DWCONV_CAP:1024 512 2048" "DEX_FINAL_CAP:1536 768 3072" "DEX_FIRST_CAP:4096 2048 8192" "DEX_REGW_WGS:256 512" "DEX_TIV_CAP:512 256 1024" "DEX_NWALK_SPLIT:0 1 2 4" "DEX_POS_CT:0 1 2 3" "DEX_CONV_TH8:1 0"; do
    K=${kv%%:*}; V=${kv#*:}
    bash tools/ab_e2e.sh $K "$V" $w > /dev/null 2>&1; cat gpurun_out/e2e_$K.txt >> $O
  done
done
cat $O
