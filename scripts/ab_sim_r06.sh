cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sim_parity.py -x -q 2>&1 | tail -5
timeout 600 python scripts/sim_fuzz.py 40 200 2>&1 | tail -3
timeout 600 python scripts/sim_fuzz.py 30 200 packed 2>&1 | tail -3
for i in 1 2 3; do
  echo "== new"; python scripts/bench_sim.py --E 16384 --blocks 64 --policy cruise 2>&1 | tail -1
  echo "== r05"; python scripts/bench_sim.py --E 16384 --blocks 64 --policy cruise --lib copo_amd/lib/libcopo_hip_r05.so 2>&1 | tail -1
done
echo "== new E256"; python scripts/bench_sim.py --E 256 --blocks 1024 --policy cruise 2>&1 | tail -1
echo "== r05 E256"; python scripts/bench_sim.py --E 256 --blocks 1024 --policy cruise --lib copo_amd/lib/libcopo_hip_r05.so 2>&1 | tail -1
