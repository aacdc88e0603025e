export RMR_BAM_LEVEL=1
echo "== default"; timeout 300 python tests/manual/prof_infer_cli.py 6000 6 fp32 1 2>&1 | grep "procs/gpu"
echo "== OMP 2, PACK 2"; OMP_NUM_THREADS=2 RMR_PACK_THREADS=2 timeout 300 python tests/manual/prof_infer_cli.py 6000 6 fp32 1 2>&1 | grep "procs/gpu"
echo "== OMP 2, PACK 2, BAM_THREADS 3"; OMP_NUM_THREADS=2 RMR_PACK_THREADS=2 RMR_BAM_THREADS=3 timeout 300 python tests/manual/prof_infer_cli.py 6000 6 fp32 1 2>&1 | grep "procs/gpu"
echo "== default P=5"; timeout 300 python tests/manual/prof_infer_cli.py 6000 5 fp32 1 2>&1 | grep "procs/gpu"
