cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sim_parity.py -x -q 2>&1 | tail -3
timeout 300 python scripts/sim_fuzz.py 30 150 2>&1 | tail -2
echo "=== shipped Tollgate populations: no buildings / buildings (crash on touch, seen by the LiDAR) / buildings + the other MetaDrive rules"
python scripts/eval_f4_populations.py '{}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"toll_buildings": 1}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"toll_buildings": 1, "toll_early_exit": 1, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "speed_reward": 0.0}}' 2>&1 | grep tollgate
for mb in 512 1024; do echo "== fused step, $mb rows per minibatch"; COPO_BENCH_MB=$mb python scripts/bench_fused.py 300 2>&1 | grep "fused sgd"; done
bash scripts/fidelity_r06.sh "tollb" "0 1 2 3" > gpurun_out/r06_fid_tollb2.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_tollb2.txt 2>/dev/null
bash scripts/fidelity_r06.sh "dp" "0 1 2 3" > gpurun_out/r06_fid_dp.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_dp.txt 2>/dev/null
