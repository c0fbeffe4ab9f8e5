# pos_conv: column-workgroup form on/off x chunk width, per workload: tools/possweep.sh <workload>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/possweep; mkdir -p $O
WL=$1
for col in 0 1; do for ct in 0 1 2; do
  rm -rf /tmp/p3
  env DEX_POS_COL=$col DEX_POS_CT=$ct rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o t -- python $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/trace_step.py /tmp/p3/t_kernel_trace.csv > $O/${WL}_${col}_${ct}.txt
  echo "COL=$col CT=$ct: $(grep -h -E 'pos_conv_direct' $O/${WL}_${col}_${ct}.txt | head -1)"
done; done
