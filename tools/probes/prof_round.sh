export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
python bench.py --steps 4 --warmup 2 > $O/bench_L352.json 2> $O/bench_L352.err
python bench.py --samples 12 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b12.json 2>> $O/bench.err
python bench.py --samples 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b1.json 2>> $O/bench.err
python bench.py --samples 13 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b13.json 2>> $O/bench.err
python bench.py --samples 25 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b25.json 2>> $O/bench.err
python bench.py --samples 50 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b50.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-op-profile > $O/prof.log 2>&1
python tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-op-profile > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-op-profile > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/prof
python tools/probes/clock_probe.py 3 > $O/clock.txt 2>&1
bash tools/pmc_run.sh tri r03i_tri; python tools/pmc_reduce.py gpurun_out/r03i_tri tri_attn4 > $O/pmc_triattn4.txt
bash tools/pmc_run.sh contract r03i_contract; python tools/pmc_reduce.py gpurun_out/r03i_contract gemm3_kernel > $O/pmc_contract.txt
rm -rf gpurun_out/r03i_tri_* gpurun_out/r03i_contract_*
ls -la $O
