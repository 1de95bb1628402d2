cd $GRAFT_REPO_ROOT
bash scripts/fidelity_r06.sh "interref" "0 1 2 3 4 5 6 7" > gpurun_out/r06_fid_interref.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_interref.txt
bash scripts/fidelity_r06.sh "dp" "0 1 2 3" > gpurun_out/r06_fid_dp.txt 2>&1
python scripts/fidelity_summary.py gpurun_out/r06_fid_dp.txt
python scripts/fidelity_dynamics_sweep.py 2>&1 | grep -v amdgpu > gpurun_out/r06_fid_dynamics.txt; cat gpurun_out/r06_fid_dynamics.txt
for mb in 512 1024; do echo "== fused step, $mb rows per minibatch"; COPO_BENCH_MB=$mb python scripts/bench_fused.py 300 2>&1 | grep "fused sgd"; done
