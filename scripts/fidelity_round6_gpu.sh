cd $GRAFT_REPO_ROOT
# (first pass.  Its `toll_buildings: 1` was a road-coordinate box test the LiDAR did not see; that implementation was replaced by static boxes -- on the final
#  code 1 = boxes the LiDAR SEES (the second pass, fidelity_round6_gpu3.sh), 2 = boxes it does not see (fidelity_round6_gpu5.sh): re-running this file repeats the second pass)
# (the Tollgate keys pin the scene defaults of the time this pass was run; see scripts/fidelity_r06.sh)
timeout 600 python -m pytest tests/test_gpu_sim_parity.py -x -q 2>&1 | tail -3
echo "=== shipped populations: as before / booth buildings / all MetaDrive rules"
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.1, "toll_speed_limit": 0.0, "overspeed_penalty": 0.0, "toll_early_exit": 0, "toll_buildings": 0, "lidar_range": 40.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.1, "toll_speed_limit": 0.0, "overspeed_penalty": 0.0, "toll_early_exit": 0, "toll_buildings": 1, "lidar_range": 40.0}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"speed_reward": 0.0, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "toll_early_exit": 1, "toll_buildings": 1, "lidar_range": 40.0}}' 2>&1 | grep tollgate
bash scripts/fidelity_r06.sh "tollb" "0 1 2 3" > gpurun_out/r06_fid_tollb.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_tollb.txt
bash scripts/fidelity_r06.sh "inter" "0 1 2 3 4 5 6 7" > gpurun_out/r06_fid_inter.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_inter.txt
