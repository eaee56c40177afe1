#!/bin/bash
# SQ / LDS counters of the attention kernel variants (tools/attn_lab.py), one rocprofv3 --pmc pass.
# usage (GPU box): bash tools/attn_pmc.sh <tag> [extra attn_lab args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out/attn_pmc_$tag
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
    --output-format csv -d $out -- python /root/repo/tools/attn_lab.py --rounds 1 --iters 2 "$@" > $out.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM \
    --output-format csv -d ${out}b -- python /root/repo/tools/attn_lab.py --rounds 1 --iters 2 "$@" >> $out.log 2>&1
python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/attn_pmc_$tag.md $out ${out}b
