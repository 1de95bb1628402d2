cd $GRAFT_REPO_ROOT
# (the Tollgate keys pin the scene defaults of the time this pass was run; see scripts/fidelity_r06.sh)
timeout 600 python -m pytest tests/test_gpu_sim_parity.py -x -q 2>&1 | tail -3
timeout 300 python scripts/sim_fuzz.py 30 150 2>&1 | tail -2
echo "=== shipped Tollgate populations: no buildings / buildings (crash on touch, seen by the LiDAR) / buildings + the other MetaDrive rules"
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.1, "toll_speed_limit": 0.0, "overspeed_penalty": 0.0, "toll_early_exit": 0, "toll_buildings": 0, "lidar_range": 40.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.1, "toll_speed_limit": 0.0, "overspeed_penalty": 0.0, "toll_early_exit": 0, "toll_buildings": 1, "lidar_range": 40.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.0, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "toll_early_exit": 1, "toll_buildings": 1, "lidar_range": 40.0}}' 2>&1 | grep tollgate
for mb in 512 1024; do echo "== fused step, $mb rows per minibatch"; COPO_BENCH_MB=$mb python scripts/bench_fused.py 300 2>&1 | grep "fused sgd"; done
bash scripts/fidelity_r06.sh "tollb" "0 1 2 3" > gpurun_out/r06_fid_tollb2.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_tollb2.txt 2>/dev/null
bash scripts/fidelity_r06.sh "dp" "0 1 2 3" > gpurun_out/r06_fid_dp.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_dp.txt 2>/dev/null
