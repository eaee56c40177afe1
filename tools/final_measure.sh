set -x
cd /root/repo
python bench.py > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err
python bench.py --model esm2_150m --tokens 8192 --seq-len 512 --no-cpu-baseline > gpurun_out/bench_150m.json 2>/dev/null
python bench.py --model esm2_3b --tokens 50000 --no-cpu-baseline --steps 5 > gpurun_out/bench_3b.json 2>/dev/null
python bench.py --batch proteome --no-cpu-baseline > gpurun_out/bench_proteome.json 2>/dev/null
python bench.py --model esm1b --no-cpu-baseline > gpurun_out/bench_esm1b.json 2>/dev/null
python bench.py --quantization 4bit --no-cpu-baseline > gpurun_out/bench_650m_q4.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r01e -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_e.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_fetch_e -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_fetch_e.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/pmc_write_e -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_write_e.log 2>&1
cd /root/repo
ls gpurun_out/prof_r01e/* | head
