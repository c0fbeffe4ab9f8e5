# convt_up: pixel-tile count / workgroup-count sweep on one workload: tools/cusweep.sh <workload>
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/cusweep; mkdir -p $O
WL=$1
for mt in 1 2; do for wgs in 256 512 1024 2048; do
  rm -rf /tmp/p3
  env DEX_CONVT_MT=$mt DEX_CONVT_WGS=$wgs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o t -- python $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/trace_step.py /tmp/p3/t_kernel_trace.csv > $O/${WL}_${mt}_${wgs}.txt
  echo "MT=$mt WGS=$wgs: $(grep -h -E 'convt_up' $O/${WL}_${mt}_${wgs}.txt | head -1)"
done; done
