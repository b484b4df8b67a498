timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k cgconv 2>&1 | tail -2
for i in 1 2; do
echo "== half"; python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "^bwd:"
echo "== nohalf"; MDL_CG_NO_HALF=1 python tools/bench_kernels.py --which fwd,bwd --iters 20 2>&1 | grep -E "^bwd:"
done
