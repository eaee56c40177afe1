mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_model_gpu.py -m gpu -x -q > $O/ord_tests.txt 2>&1 < /dev/null
tail -3 $O/ord_tests.txt
for i in 1 2; do
ESME_ATTN_ORDER=0 timeout 300 python bench.py --batch proteome --no-cpu-baseline < /dev/null > $O/ord_prot_off_$i.json 2>/dev/null
timeout 300 python bench.py --batch proteome --no-cpu-baseline < /dev/null > $O/ord_prot_on_$i.json 2>/dev/null
done
ESME_ATTN_ORDER=0 timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/ord_uni_off.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/ord_uni_on.json 2>/dev/null
for f in $O/ord_prot_*.json $O/ord_uni_*.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], d.get('kernel_ms_per_step'))
PY
done
