mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "qk_norm or attention" < /dev/null 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_next_rows_gpu.py -m gpu -x -q < /dev/null 2>&1 | tail -3
for i in 1 2; do
ESME_ATTN_QP=0 timeout 300 python bench.py --model esmc_600m --tokens 32064 --seq-len 1002 --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/qpc_off_$i.json
timeout 300 python bench.py --model esmc_600m --tokens 32064 --seq-len 1002 --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/qpc_on_$i.json
done
for f in $O/qpc_off_*.json $O/qpc_on_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d.get('kernel_ms_per_step'), d.get('attention',{}).get('frac'))
PY
done
