#!/bin/bash
# Round-4 measurement batch (GPU box).  Everything lands under gpurun_out/r04/; tools/make_profiles_r04.py turns it
# into the committed profiles/r04_* files.
set -x
exec < /dev/null            # nothing here reads stdin: a stray read must fail, not wait
cd /root/repo
O=gpurun_out/r04
mkdir -p $O
make -C esm-efficient_amd/csrc TRACE=1 > $O/make_trace.log 2>&1      # the instrumented library must match the sources (it is not built by __graft_entry__.build())
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --spawn --no-cpu-baseline > $O/bench_spawn.json 2>/dev/null
timeout 900 python bench.py --batch proteome --no-cpu-baseline > $O/bench_proteome.json 2>/dev/null
timeout 900 python bench.py --high-precision --no-cpu-baseline > $O/bench_high_precision.json 2>/dev/null
timeout 900 python bench.py --precision half --no-cpu-baseline > $O/bench_half.json 2>/dev/null
timeout 900 python bench.py --precision half --batch proteome --no-cpu-baseline > $O/bench_half_proteome.json 2>/dev/null
timeout 900 python bench.py --precision half --model esmc_600m --tokens 32064 --seq-len 1002 --no-cpu-baseline > $O/bench_half_esmc600m.json 2>/dev/null
timeout 900 python bench.py --precision half --model esm2_3b --tokens 50000 --no-cpu-baseline --steps 5 > $O/bench_half_3b.json 2>/dev/null
timeout 900 python bench.py --precision half --model esm2_150m --tokens 8192 --seq-len 512 --no-cpu-baseline --steps 30 > $O/bench_half_150m.json 2>/dev/null
timeout 900 python bench.py --precision exact > $O/bench_exact.json 2>/dev/null
timeout 900 python bench.py --precision exact --model esmc_600m --tokens 32064 --seq-len 1002 --no-cpu-baseline > $O/bench_exact_esmc600m.json 2>/dev/null
timeout 900 python bench.py --model esm2_3b --tokens 50000 --no-cpu-baseline --steps 5 > $O/bench_3b.json 2>/dev/null
timeout 900 python bench.py --model esmc_600m --tokens 32064 --seq-len 1002 --no-cpu-baseline > $O/bench_esmc600m.json 2>/dev/null
timeout 900 python bench.py --model esm2_150m --tokens 8192 --seq-len 512 --no-cpu-baseline --steps 30 > $O/bench_150m.json 2>/dev/null
ESME_NO_C_FORWARD=1 timeout 900 python bench.py --model esm2_150m --tokens 8192 --seq-len 512 --no-cpu-baseline --steps 30 > $O/bench_150m_pyloop.json 2>/dev/null
timeout 900 python bench.py --model esm2_150m --tokens 8192 --seq-len 512 --no-cpu-baseline --steps 30 --graph > $O/bench_150m_graph.json 2>/dev/null
timeout 900 python bench.py --model esm1b --no-cpu-baseline > $O/bench_esm1b.json 2>/dev/null
timeout 900 python bench.py --quantization 4bit --no-cpu-baseline > $O/bench_650m_q4.json 2>/dev/null
timeout 900 python tools/attn_lab.py > $O/attn_lab_uniform.txt 2>&1
timeout 900 python tools/attn_lab.py --batch proteome --rounds 3 > $O/attn_lab_proteome.txt 2>&1
timeout 900 python tools/attn_lab.py --seq-len 2000 --rounds 3 > $O/attn_lab_s2000.txt 2>&1
ESME_HIP_LIB=/root/repo/esm-efficient_amd/esme/libesme_hip_trace.so PERSIST=0 timeout 900 python tools/gemm_phase_trace.py > $O/gemm_phase_trace.txt 2>&1
timeout 900 python tools/gemm_persist_check.py > $O/gemm_persist.txt 2>&1
LAB1W_VARIANTS=0,1,3,5,7,9,17 LAB_SHAPES='[("normal",50000,5120,1280),("normal",50000,1280,5120)]' timeout 900 python tools/lab/run_gemm_1w.py > $O/gemm_1w_lab.txt 2>&1
timeout 900 python tools/attn_lab.py --qp --variants 1,4,8 > $O/attn_lab_qp_uniform.txt 2>&1
timeout 900 python tools/gemm_small_m.py > $O/gemm_small_m.txt 2>&1
timeout 900 python bench.py --gpus 1 --spawn --no-cpu-baseline --model esm2_3b --steps 5 > $O/bench_3b_spawn.json 2>/dev/null
timeout 900 python tools/gemm_epi_bench.py > $O/gemm_epi_bench.txt 2>&1
timeout 900 python tools/attn_power_probe.py > $O/attn_power_probe.txt 2>&1
timeout 900 python tools/power_probe.py > $O/power_probe.txt 2>&1
timeout 900 python tools/qk_norm_bench.py > $O/qk_norm_bench.txt 2>&1
for p in mfma_issue_probe dma_role_probe mfma_shape_probe; do [ -x tools/lab/bin/$p ] && tools/lab/bin/$p > $O/$p.txt 2>&1; done
ESME_GEMM_PERSIST=0 timeout 900 python bench.py --no-cpu-baseline > $O/bench_nopersist.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/$O/prof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_half -- python /root/repo/bench.py --precision half --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/$O/prof_half.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /root/repo/$O/pmc_sq -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_sq.log 2>&1
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /root/repo/$O/pmc_sq2 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_sq2.log 2>&1
cd /root/repo
ls $O/prof/* $O/pmc_fetch/* | head
