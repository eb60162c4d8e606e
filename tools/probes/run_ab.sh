cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "$@"; do
if [ "$v" = "base" ]; then LIBARG=""; else LIBARG="tools/ab_lib.py tools/probes/bin/libabx_$v.so"; fi
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$v -o x -- python $LIBARG tools/kbench.py --bc 100 --L 352 --only ipa > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_$v/x_results.db gpurun_out/prof_$v/stats.csv > /dev/null; echo $v; grep -E "ipa_(pair|weights)" gpurun_out/prof_$v/stats.csv | awk -F"\"," '{print substr($1,1,22), $2}'
done; done
