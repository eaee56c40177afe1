#!/bin/bash
# Same-box A/B of two whole trees (HEAD vs a checkout of another commit under gpurun_in_<tag>/), separate processes, interleaved:
#   tools/lab/ab_trees.sh r5 "--precision half"   -> gpurun_out/ab/<tag>_{A,B}_<i>.json
TAG=$1; shift
ARGS="$*"
O=/root/repo/gpurun_out/ab; mkdir -p $O
for i in 1 2 3; do
  (timeout 600 python bench.py $ARGS --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_A_$i.json 2>/dev/null)
  (cd /root/repo/gpurun_in_$TAG && timeout 600 python bench.py $ARGS --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_B_$i.json 2>/dev/null)
done
python - <<'P'
import json, glob, os
O='/root/repo/gpurun_out/ab'
for f in sorted(glob.glob(O+'/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable', e); continue
    print(os.path.basename(f), d['ms_per_step'], d.get('kernel_ms_per_step') or d.get('kernels'))
P
