# all bench workloads, bf16 mode, one box: workload value ms_per_step
for w in gedex_b1 gedex_b32 dex_b1 dex_b32 dex_esd_b32_n100 gedex_long gedex_b1_t800; do
  echo -n "$w: "; python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('ms_per_euler_step'))"
done
