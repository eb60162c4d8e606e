export TMPDIR=/tmp
for r in 1 2; do
python tools/probes/kb_store.py 20 768 ab 0 2>&1 | grep -v amdgpu.ids | awk -v v=head '{printf "%s %s %s | ", v, $1, $(NF-1)} END {print ""}'
for v in prio1 prio3; do
python tools/ab_lib.py tools/probes/bin/libabx_$v.so tools/probes/kb_store.py 20 768 ab 0 2>&1 | grep -v amdgpu.ids | awk -v v=$v '{printf "%s %s %s | ", v, $1, $(NF-1)} END {print ""}'
done
done
