cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sim_parity.py -x -q 2>&1 | tail -3
echo "=== shipped populations: as before / booth buildings / all MetaDrive rules"
python scripts/eval_f4_populations.py '{}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"toll_buildings": 1}}' 2>&1 | grep tollgate
python scripts/eval_f4_populations.py '{"tollgate": {"toll_buildings": 1, "toll_early_exit": 1, "toll_speed_limit": 0.8333333, "overspeed_penalty": 0.5, "speed_reward": 0.0}}' 2>&1 | grep tollgate
bash scripts/fidelity_r06.sh "tollb" "0 1 2 3" > gpurun_out/r06_fid_tollb.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_tollb.txt
bash scripts/fidelity_r06.sh "inter" "0 1 2 3 4 5 6 7" > gpurun_out/r06_fid_inter.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_inter.txt
