set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r04f
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fused_transition or range_word or split_accuracy" -x 2>&1 | tail -5
python tools/probes/kb_store.py 20 > gpurun_out/r04f/kb_store.txt 2>&1
python tools/ab_lib.py tools/probes/bin/libabx_base.so tools/probes/kb_mlp.py 20 > gpurun_out/r04f/kb_mlp_base_20.txt 2>&1
python tools/probes/kb_mlp.py 20 > gpurun_out/r04f/kb_mlp_new_20.txt 2>&1
python tools/ab_lib.py tools/probes/bin/libabx_base.so tools/probes/kb_mlp.py 100 > gpurun_out/r04f/kb_mlp_base_100.txt 2>&1
python tools/probes/kb_mlp.py 100 > gpurun_out/r04f/kb_mlp_new_100.txt 2>&1
tail -n 8 gpurun_out/r04f/*.txt
