cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z_]*\|TCP_[A-Z_]*\|SQ_[A-Z_]*\|TA_[A-Z_]*" | sort -u > /root/repo/gpurun_out/counters.txt
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /root/repo/gpurun_out/pmc_sq2 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_sq2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum --output-format csv -d /root/repo/gpurun_out/pmc_tcc -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_tcc.log 2>&1
ls /root/repo/gpurun_out/pmc_sq2/* /root/repo/gpurun_out/pmc_tcc/* 2>&1 | head
