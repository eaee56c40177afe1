#!/bin/bash
# precision 'half' with the attention kernel's fixed-reference form on / off (ESME_HALF_QP), separate processes, interleaved, one box
O=/root/repo/gpurun_out/half_qp; mkdir -p $O
run() { ESME_HALF_QP=$1 timeout 600 python bench.py --precision half --no-cpu-baseline --steps 10 --warmup 3 $3 > $O/$2_qp$1_$4.json 2>/dev/null; }
for i in 1 2; do
  for q in 1 0; do
    run $q uniform "" $i
    run $q proteome "--batch proteome" $i
    run $q esmc600m "--model esmc_600m --tokens 32064 --seq-len 1002" $i
  done
done
for q in 1 0; do run $q esm2_3b "--model esm2_3b --steps 5" 1; run $q esm2_150m "--model esm2_150m --tokens 8192 --seq-len 512 --steps 30" 1; done
python - <<'P'
import json, glob, os
for f in sorted(glob.glob('/root/repo/gpurun_out/half_qp/*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable'); continue
    print(f"{os.path.basename(f):34s} {d['ms_per_step']:8.3f} ms  {100*d['e2e']['frac_bf16_mfma_peak']:.1f} %  attn {d.get('kernel_ms_per_step',{}).get('attn')}  parity {d.get('parity',{}).get('rel_fro_hip_vs_oracle_fp32')}  plan qp {d.get('plan',{}).get('fixed_reference_attention') if isinstance(d.get('plan'),dict) else None}")
P
