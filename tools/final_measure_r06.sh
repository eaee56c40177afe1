#!/bin/bash
# Round-5 measurement batch (GPU box).  Everything lands under gpurun_out/r06/final/; tools/make_profiles_r06.py turns it
# into the committed profiles/r06_* files.
set -x
exec < /dev/null            # nothing here reads stdin: a stray read must fail, not wait
cd /root/repo
O=gpurun_out/r06/final
mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 900 python bench.py --gpus 1 --spawn --no-cpu-baseline > $O/bench_spawn.json 2>/dev/null
timeout 900 python bench.py --batch proteome --no-cpu-baseline > $O/bench_proteome.json 2>/dev/null
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
timeout 900 python bench.py --model esm1b --no-cpu-baseline > $O/bench_esm1b.json 2>/dev/null
timeout 900 python bench.py --quantization 4bit --no-cpu-baseline > $O/bench_650m_q4.json 2>/dev/null
timeout 900 python tools/proteome_e2e.py --out $O/proteome_e2e.json > /dev/null 2>&1
timeout 900 python tools/proteome_e2e.py --precision half --out $O/proteome_e2e_half.json > /dev/null 2>&1
L=33 E=1280 timeout 900 python tools/half_outlier_probe.py > $O/half_outlier_probe_33x1280.txt 2>&1
timeout 900 python tools/half_outlier_probe.py > $O/half_outlier_probe_12x640.txt 2>&1
timeout 900 python tools/half_robust_breakdown.py > $O/half_robust_breakdown.json 2>/dev/null
timeout 900 python tools/attn_lab.py --qp --variants 1,4,8 > $O/attn_lab_qp_uniform.txt 2>&1
# round 6: a 15B-width model (E = 5 120, 40 heads of 128: ESM2-15B's geometry, first 4 of 48 layers), fast and half; head dim 128 attention alone
timeout 900 python bench.py --model esm2_15b --layers 4 --tokens 50000 --no-cpu-baseline --steps 5 > $O/bench_15b_width.json 2>/dev/null
timeout 900 python bench.py --model esm2_15b --layers 4 --tokens 50000 --no-cpu-baseline --steps 5 --precision half > $O/bench_half_15b_width.json 2>/dev/null
timeout 900 python tools/attn_lab.py --d 128 --heads 40 --variants 1 --rounds 3 --iters 10 > $O/attn_lab_d128_s500.txt 2>&1
timeout 900 python tools/attn_lab.py --d 128 --heads 40 --variants 1 --rounds 3 --iters 10 --seq-len 2000 > $O/attn_lab_d128_s2000.txt 2>&1
timeout 900 python tools/tail_round_probe.py > $O/tail_round_probe.txt 2>&1
timeout 900 python tools/half_guard_cost.py > $O/half_guard_cost.txt 2>&1
BATCH=proteome timeout 900 python tools/half_guard_cost.py > $O/half_guard_cost_proteome.txt 2>&1
timeout 900 python tools/attn_ragged_table.py --json $O/attn_ragged_table.json > $O/attn_ragged_table.txt 2>&1
timeout 900 python tools/attn_sb_bench.py > $O/attn_sb_bench.txt 2>&1
CALIB=residues timeout 900 python tools/half_token_outlier_probe.py > $O/token_outlier_guard.txt 2>&1
timeout 900 python tools/half_token_outlier_probe.py > $O/token_outlier_vocab.txt 2>&1
timeout 900 python tools/qk_norm_bench.py > $O/qk_norm_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-half > /root/repo/$O/prof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_half -- python /root/repo/bench.py --precision half --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/$O/prof_half.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-half > /root/repo/$O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-half > /root/repo/$O/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /root/repo/$O/pmc_sq -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-half > /root/repo/$O/pmc_sq.log 2>&1
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /root/repo/$O/pmc_sq2 -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-half > /root/repo/$O/pmc_sq2.log 2>&1
# precision 'half': the mode's own traffic / SQ passes (profiles/r06_traffic_half.json, r06_pmc_traffic_half.md, r06_pmc_counters_half.md)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_fetch_half -- python /root/repo/bench.py --precision half --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_fetch_half.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_write_half -- python /root/repo/bench.py --precision half --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_write_half.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /root/repo/$O/pmc_sq_half -- python /root/repo/bench.py --precision half --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/$O/pmc_sq_half.log 2>&1
cd /root/repo
# keep the merge-back small: the per-dispatch traces are large, the stats / counter CSVs are what the profiles are made from
find $O -name '*kernel_trace.csv' -size +8M -delete
ls $O | head -50
