export TMPDIR=/tmp
mkdir -p gpurun_out/r04h
python bench.py --samples 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04h/b1.json 2>/dev/null
python bench.py --samples 12 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r04h/b12.json 2>/dev/null
