cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/r1f; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p3_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/p3_$c -name "*counter_collection.csv" | head -1) $c > $O/b1_pmc_$c.txt 2>&1
done
head -5 $O/b1_pmc_FETCH_SIZE.txt
