# A/B of the SLP vectorizer (packed fp32 arithmetic) for one kernel file: tools/ab_slp.sh <file.hip> <workload> [<workload> ..]
# (the library as built, then <file.hip> rebuilt with -fno-slp-vectorize; one Euler step traced under rocprofv3 each)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; F=$1; shift
for v in packed single; do
  if [ $v = single ]; then
    touch $R/dex_tts_amd/csrc/$F
    (cd $R && DEX_FILE_FLAGS="$F=-fno-slp-vectorize" python -c "import dex_tts_amd.build as b; b.build()") 2>&1 | grep -iE " error|built" | head -3
  fi
  for w in "$@"; do
    rm -rf /tmp/ps_$w
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$w -o t -- python $R/bench.py --workload $w --steps 2 --warmup 1 --graph off --no-cpu-baseline --no-profile > /dev/null 2>&1
    echo "== $v $w"; python $R/tools/trace_step.py $(find /tmp/ps_$w -name "*kernel_trace.csv" | head -1) | grep -E "step:|rowchain" | tail -4
  done
done
