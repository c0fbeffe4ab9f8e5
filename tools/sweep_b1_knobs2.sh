cd /root/repo
O=gpurun_out/sweep_b1b.txt; : > $O
for kv in "DEX_FIRST_CAP:4096 320 640 1280" "DEX_FINAL_CAP:1536 160 320 640" "DEX_CONVT_WGS:512 160 256" "DEX_CONV_DOWN_WGS:512 160 256" "DEX_FIRST_MFMA:1 0" "DEX_DIT_CLUSTER_LOCAL:1 0" "DEX_H_BF16:1 0" "DEX_LP_INTER:1 0"; do
  K=${kv%%:*}; V=${kv#*:}
  bash tools/ab_e2e.sh $K "$V" gedex_b1 > /dev/null 2>&1; cat gpurun_out/e2e_$K.txt >> $O
done
cat $O
