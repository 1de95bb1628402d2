cd $GRAFT_REPO_ROOT
# (the Tollgate keys pin the scene defaults of the time this pass was run; see scripts/fidelity_r06.sh)
timeout 600 python -m pytest tests/test_gpu_sim_parity.py -x -q 2>&1 | tail -2
echo "=== shipped Tollgate populations, LiDAR 20 m, buildings the LiDAR does NOT see (+ booth rules)"
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.0, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "toll_early_exit": 1, "toll_buildings": 2, "lidar_range": 20.0}}' 2>&1 | grep tollgate
bash scripts/fidelity_r06.sh "hidden" "0 1 2 3" > gpurun_out/r06_fid_hidden.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_hidden.txt 2>/dev/null
