# Reproducibility of one model call (per-op checksums, tools/probes/op_checksums.py) while ANOTHER process uses the same GPU:
#   (a) a plain torch matmul loop, started before and while our process starts  (b) a second copy of our process, started together
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
OUT=/tmp/solo.txt REPS=1 python tools/probes/op_checksums.py 2>/dev/null | tail -1
cat > /tmp/mm.py <<EOF
import torch, time
a = torch.randn(8192, 8192, device='cuda'); b = torch.randn(8192, 8192, device='cuda')
t0 = time.time()
while time.time() - t0 < float(__import__('sys').argv[1]):
    for _ in range(20): c = a @ b
    torch.cuda.synchronize()
EOF
for t in 1 2 3 4; do
python /tmp/mm.py 25 > /dev/null 2>&1 &
sleep $((t % 2 * 6))
OUT=/tmp/mm$t.txt REPS=1 python tools/probes/op_checksums.py > /dev/null 2>&1
wait
echo "matmul neighbour, trial $t: $(diff /tmp/solo.txt /tmp/mm$t.txt | grep -c '^>') differing ops; first: $(diff /tmp/solo.txt /tmp/mm$t.txt | grep '^>' | head -1 | cut -c1-40)"
done
for t in 1 2 3 4; do
(OUT=/tmp/bg$t.txt REPS=1 python tools/probes/op_checksums.py > /dev/null 2>&1) &
OUT=/tmp/co$t.txt REPS=1 python tools/probes/op_checksums.py > /dev/null 2>&1
wait
for f in /tmp/co$t.txt /tmp/bg$t.txt; do echo "own copy started together, trial $t: $(diff /tmp/solo.txt $f | grep -c '^>') differing ops; first: $(diff /tmp/solo.txt $f | grep '^>' | head -1 | cut -c1-40)"; done
done
