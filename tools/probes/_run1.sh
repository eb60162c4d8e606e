export TMPDIR=/tmp
python tools/ab_lib.py tools/probes/bin/libabx_early.so tools/probes/kb_bits.py 2>&1 | grep sha
for r in 1 2; do
python tools/probes/kb_store.py 20 768 ab 0 2>&1 | grep -v amdgpu.ids | awk -v v=head '{printf "%s %s %s | ", v, $1, $(NF-1)} END {print ""}'
python tools/ab_lib.py tools/probes/bin/libabx_early.so tools/probes/kb_store.py 20 768 ab 0 2>&1 | grep -v amdgpu.ids | awk -v v=early '{printf "%s %s %s | ", v, $1, $(NF-1)} END {print ""}'
done
python tools/probes/kb_mlp.py 100 2>&1 | grep "fused mlp (2" | sed 's/^/head /'
python tools/ab_lib.py tools/probes/bin/libabx_early.so tools/probes/kb_mlp.py 100 2>&1 | grep "fused mlp (2" | sed 's/^/early /'
