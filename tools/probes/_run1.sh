export TMPDIR=/tmp
for r in 1 2; do
ABX_HIP_LIB=$PWD/tools/probes/bin/libabx_base.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-op-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['result_digest'][:12], d['ms_per_step'], d['value'])"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-op-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['result_digest'][:12], d['ms_per_step'], d['value'])"
done
