# Round-6 evidence in one gpurun call: the default bench line, rocprofv3 kernel stats + one-step traces, the DiT attention as its own
# launch, PMC HBM traffic and MFMA-busy counters for every BASELINE config IN THE MODE IT NAMES (configs[4] in fp16; configs[3]
# gets its own rows).  Usage (on the GPU box, from the repo root): bash tools/profile_round6.sh [tag]
set -x
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-round6}; O=$R/gpurun_out/$TAG; mkdir -p $O
B="--no-cpu-baseline --no-profile"
prec() { case $1 in gedex_long) echo "--precision fp16";; *) echo "--precision bf16";; esac; }
WL="gedex_b1 gedex_b32 dex_b32 dex_esd_b32_n100 gedex_long dex_b32_t512"
for w in $WL; do
  rm -rf /tmp/p_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -o t -- python $R/bench.py --workload $w $(prec $w) --steps 2 --warmup 1 --graph off $B > $O/${w}_bench_under_rocprof.json 2>/dev/null
  cp $(find /tmp/p_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  python $R/tools/trace_step.py $(find /tmp/p_$w -name "*kernel_trace.csv" | head -1) > $O/${w}_one_euler_step_trace.txt
done
# the DiT attention as its own launch (profiler evidence for roofline_attention)
for w in $WL; do
  rm -rf /tmp/pa_$w
  DEX_ATTN_SEPARATE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_$w -o t -- python $R/bench.py --workload $w $(prec $w) --steps 1 --warmup 1 --graph off $B > /dev/null 2>&1
  grep -E "Name|attn_direct|attn_q64" $(find /tmp/pa_$w -name "*kernel_stats.csv" | head -1) > $O/${w}_attention_separate_kernel_stats.csv
done
# HBM traffic: one --pmc pass per counter, no tracing
rm -f $O/pmc_traffic.json
for w in $WL; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${w}_$c
    rocprofv3 --pmc $c --output-format csv -d /tmp/pm_${w}_$c -o pmc -- python $R/bench.py --workload $w $(prec $w) --steps 1 --warmup 0 --graph off $B > /dev/null 2>&1
  done
  python $R/tools/pmc_json.py $w $(find /tmp/pm_${w}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pm_${w}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/${w}_pmc_top.txt 2>&1
done
# MFMA utilisation of the product kernels: SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES) per kernel symbol (its own --pmc pass);
# once with the fused DiT block, once with the attention as its own launch
for w in $WL; do
  for sep in 0 1; do
    rm -rf /tmp/pq_${w}_$sep
    DEX_ATTN_SEPARATE=$sep rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pq_${w}_$sep -o pmc -- python $R/bench.py --workload $w $(prec $w) --steps 1 --warmup 0 --graph off $B > /dev/null 2>&1
    python $R/tools/pmc_mfma.py $w $sep $(find /tmp/pq_${w}_$sep -name "*counter_collection.csv" | head -1) $O/mfma_util.json > $O/${w}_mfma_util_sep$sep.txt 2>&1
  done
done
# diagnostic counter sets for the batch row chain / attention (dex_b32 only)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pd_$n
  rocprofv3 --pmc $set --output-format csv -d /tmp/pd_$n -o pmc -- python $R/bench.py --workload dex_b32 --precision bf16 --steps 1 --warmup 0 --graph off $B > /dev/null 2>&1
  python - "$(find /tmp/pd_$n -name "*counter_collection.csv" | head -1)" >> $O/dex_b32_diag_counters.txt <<'PY'
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("dex::", "").split("(")[0][:60]
    if not any(t in k for t in ("dit_rowchain", "attn_direct", "attn_q64", "conv3x3_rw", "pos_conv")): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in agg.items():
    print(k, {a: round(b / 1e6, 2) for a, b in sorted(c.items())})
PY
done
# the split-weight mode (fp16x2): kernel stats + one-step traces of the B = 1 and B = 32 jobs (GeDEX and DEX), MFMA utilisation of its kernels
for w in gedex_b1 gedex_b32 dex_b32; do
  rm -rf /tmp/px_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$w -o t -- python $R/bench.py --workload $w --precision fp16x2 --steps 2 --warmup 1 --graph off $B > $O/${w}_fp16x2_bench_under_rocprof.json 2>/dev/null
  cp $(find /tmp/px_$w -name "*kernel_stats.csv" | head -1) $O/${w}_fp16x2_kernel_stats.csv
  python $R/tools/trace_step.py $(find /tmp/px_$w -name "*kernel_trace.csv" | head -1) > $O/${w}_fp16x2_one_euler_step_trace.txt
  rm -rf /tmp/pqx_$w
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pqx_$w -o pmc -- python $R/bench.py --workload $w --precision fp16x2 --steps 1 --warmup 0 --graph off $B > /dev/null 2>&1
  python $R/tools/pmc_mfma.py ${w}_fp16x2 0 $(find /tmp/pqx_$w -name "*counter_collection.csv" | head -1) $O/mfma_util.json > $O/${w}_fp16x2_mfma_util.txt 2>&1
done
python $R/bench.py > $O/bench_stdout_line.json 2> $O/bench_default.err; cp $R/gpurun_out/bench_full_gedex_b1_n1.json $O/bench_default.json   # (the FULL record: tests/test_bench_line.py reads profiles/round*_bench_default.json)
ls -la $O
