export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "ipa_tail" -x 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_call_matches or graph_replay or chunk or large_shape" -x 2>&1 | tail -4
for b in 1 12 100; do
ABX_IPA_SPLITK=0 python bench.py --samples $b --steps 4 --warmup 2 --no-cpu-baseline --no-op-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail walks K  B=$b', d['ms_per_step'], d['value'])"
python bench.py --samples $b --steps 4 --warmup 2 --no-cpu-baseline --no-op-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split-K x 11   B=$b', d['ms_per_step'], d['value'])"
done
