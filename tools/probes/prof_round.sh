# One profiling round on the GPU box (bash tools/probes/prof_round.sh <tag>): headline bench with CPU baseline, the per-GPU batch sizes of
# strong scaling, rocprofv3 kernel stats + HBM traffic passes of the same command, in-kernel clocks, counter passes of the dominant kernels.
# GIT_COMMIT=<hash> in the environment stamps profiles/pmc_traffic.json (the GPU box has no .git: pass it from the calling side).
TAG=${1:-r04}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 4 --warmup 2 > $O/bench_L352.json 2> $O/bench_L352.err
for b in 50 25 13 12 1; do python bench.py --samples $b --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_b$b.json 2>> $O/bench.err; done
python bench.py --workload 6ct7like --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_6ct7like.json 2>> $O/bench.err
python bench.py --workload 6qd7like --samples 32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_6qd7like.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof -o x -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-op-profile > $O/prof.log 2>&1
python tools/rocprof_summary.py $(find $O/prof -name "*results.db" | head -1) $O/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-op-profile > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-op-profile > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json samples=100 L=352 git_commit=${GIT_COMMIT:-unknown} > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/prof
python tools/probes/clock_probe.py 3 > $O/clock.txt 2>&1
bash tools/pmc_run.sh tri ${TAG}_tri; python tools/pmc_reduce.py gpurun_out/${TAG}_tri tri_attn8 > $O/pmc_triattn8.txt
bash tools/pmc_run.sh mlp ${TAG}_mlp; python tools/pmc_reduce.py gpurun_out/${TAG}_mlp gemm3_mlp > $O/pmc_mlp.txt
bash tools/pmc_run.sh qkv ${TAG}_qkv; python tools/pmc_reduce.py gpurun_out/${TAG}_qkv 'gemm_as_kernel<0, true' > $O/pmc_qkvg.txt
bash tools/pmc_run.sh glu ${TAG}_glu; python tools/pmc_reduce.py gpurun_out/${TAG}_glu 'gemm_as_kernel<1' > $O/pmc_glu.txt
bash tools/pmc_run.sh gtail ${TAG}_gtail; python tools/pmc_reduce.py gpurun_out/${TAG}_gtail 'gemm3_gtail' > $O/pmc_gtail.txt
rm -rf gpurun_out/${TAG}_tri_* gpurun_out/${TAG}_mlp_* gpurun_out/${TAG}_qkv_* gpurun_out/${TAG}_glu_* gpurun_out/${TAG}_gtail_*
python tools/ab_lib.py tools/probes/bin/libabx_stamp.so tools/probes/tri_stamps.py 20 352 > $O/triattn8_stamps.txt 2>&1
for c in config2 config5 config4; do python tools/e2e_bench.py $c > $O/e2e_$c.json 2>> $O/e2e.err; done
ls -la $O
