python -m pytest tests/test_vocoder.py -m gpu -q --timeout 600 2>&1 | tail -8
python - <<'PY'
import json, torch, bench
dev = torch.device("cuda", 0); st = torch.cuda.Stream(dev)
for prec in ("fp32", "bf16", "fp16"):
    print(json.dumps(bench.vocoder_block(dev, st, precision=prec)))
print(json.dumps(bench.vocoder_block(dev, st, big=True, precision="bf16")))
PY
grep vocoder gpurun_out/parity_measured.jsonl | tail -6
