cd $GRAFT_REPO_ROOT
echo "=== shipped Tollgate populations with the LiDAR at 20 m: alone / + buildings / + buildings + booth rules"
python scripts/eval_f4_populations.py '{"tollgate": {"lidar_range": 20.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"lidar_range": 20.0, "toll_buildings": 1}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"lidar_range": 20.0, "toll_buildings": 1, "toll_early_exit": 1, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "speed_reward": 0.0}}' 2>&1 | grep tollgate
bash scripts/fidelity_r06.sh "tolll" "0 1 2 3" > gpurun_out/r06_fid_tolll.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_tolll.txt 2>/dev/null
