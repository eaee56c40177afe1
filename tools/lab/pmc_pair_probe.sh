#!/bin/bash
# PMC counters of the pair-stream residual GEMM (K = 5120) in this tree and in gpurun_in_r5: is the K-loop slowdown instruction fetch?
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/pmc_pair; rm -rf $O; mkdir -p $O
for T in repo gpurun_in_r5; do
  TREE=/root/repo; [ $T = gpurun_in_r5 ] && TREE=/root/repo/gpurun_in_r5
  for G in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
    n=$(echo $G | cut -d' ' -f1)
    TREE=$TREE KS=5120 ROUNDS=1 ITERS=3 timeout 300 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/${T}_$n -- python /root/repo/tools/lab/pair_gemm_probe.py > $O/${T}_$n.log 2>&1
  done
done
python - <<'P'
import csv, glob, os, collections
O='/root/repo/gpurun_out/pmc_pair'
for d in sorted(glob.glob(O+'/*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_bf16_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(os.path.basename(d.rstrip('/')), {k: round(sum(v)/len(v)) for k,v in acc.items()}, 'launches', {k:len(v) for k,v in acc.items()})
P
