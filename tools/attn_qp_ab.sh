mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attention or attn or q_scale or rotary" < /dev/null > $O/qp_tests.txt 2>&1; tail -12 $O/qp_tests.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_next_rows_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q < /dev/null > $O/qp_model_tests.txt 2>&1; tail -5 $O/qp_model_tests.txt
for i in 1 2; do
ESME_ATTN_QP=0 timeout 300 python bench.py --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/qp_off_$i.json
timeout 300 python bench.py --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/qp_on_$i.json
done
ESME_ATTN_QP=0 timeout 300 python bench.py --batch proteome --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/qp_off_prot.json
timeout 300 python bench.py --batch proteome --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 > $O/qp_on_prot.json
for f in $O/qp_off_*.json $O/qp_on_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d.get('kernel_ms_per_step'), d.get('attention',{}).get('frac'))
PY
done
