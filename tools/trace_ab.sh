# one Euler step per launch for a workload under two environment settings: tools/trace_ab.sh <workload> <VAR> <val0> <val1>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/trace_ab; mkdir -p $O
WL=$1; VAR=$2
for v in $3 $4; do
  rm -rf /tmp/p3
  env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o t -- python $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/trace_step.py /tmp/p3/t_kernel_trace.csv > $O/${WL}_${VAR}_$v.txt
  echo "== $VAR=$v"; grep -E "first_conv|pp64|step:" $O/${WL}_${VAR}_$v.txt | head -12
done
