cd $GRAFT_REPO_ROOT
# (the Tollgate keys pin the scene defaults of the time this pass was run; see scripts/fidelity_r06.sh)
echo "=== shipped Tollgate populations with the LiDAR at 20 m: alone / + buildings / + buildings + booth rules"
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.1, "toll_speed_limit": 0.0, "overspeed_penalty": 0.0, "toll_early_exit": 0, "toll_buildings": 0, "lidar_range": 20.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.1, "toll_speed_limit": 0.0, "overspeed_penalty": 0.0, "toll_early_exit": 0, "toll_buildings": 1, "lidar_range": 20.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.0, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "toll_early_exit": 1, "toll_buildings": 1, "lidar_range": 20.0}}' 2>&1 | grep tollgate
bash scripts/fidelity_r06.sh "tolll" "0 1 2 3" > gpurun_out/r06_fid_tolll.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_tolll.txt 2>/dev/null
