export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "heads_tail or ipa_tail" -x 2>&1 | tail -5
python -m pytest tests/test_gpu_model.py -q -m gpu -k "fused_heads or graph_replay or full_call_matches" -x 2>&1 | tail -5
for b in 1 12; do
ABX_NO_FUSED_HEADS=1 python bench.py --samples $b --steps 5 --warmup 2 --no-cpu-baseline --no-op-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('separate B=$b', d['ms_per_step'], d['value'])"
python bench.py --samples $b --steps 5 --warmup 2 --no-cpu-baseline --no-op-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused    B=$b', d['ms_per_step'], d['value'])"
done
