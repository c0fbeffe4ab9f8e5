# A/B of the packed-fp32 prologue arithmetic of one kernel file: tools/pkab.sh <workload> <file.hip> [kernel-name pattern]
# (traces one Euler step with the library as built, rebuilds <file.hip> with -DDEX_NO_PK -fno-slp-vectorize, traces again)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pkab; mkdir -p $O
WL=$1; F=$2; PAT=${3:-pp64|stream64|step:}
for v in packed single; do
  if [ $v = single ]; then
    touch $R/dex_tts_amd/csrc/$F
    (cd $R && DEX_FILE_FLAGS="$F=-DDEX_NO_PK -fno-slp-vectorize" python -c "import dex_tts_amd.build as b; b.build(verbose=False)") 2>&1 | grep -iE "error" | head -5
  fi
  rm -rf /tmp/p3
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -o t -- python $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/trace_step.py /tmp/p3/t_kernel_trace.csv > $O/${WL}_$v.txt
  echo "== $v"; grep -E "$PAT" $O/${WL}_$v.txt | head -14
done
