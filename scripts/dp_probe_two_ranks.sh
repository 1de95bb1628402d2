#!/bin/bash
# Two ranks of copo_amd/dp_probe.py sharing cuda:0 over gloo (a one-GPU box): the data-parallel tile exchange at a given
# learner shape.  usage: scripts/dp_probe_two_ranks.sh [hidden] [obs_dim] [timeout_s] [nets] [minibatch]
H=${1:-256}; O=${2:-92}; T=${3:-150}; N=${4:-2}; MB=${5:-128}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 WORLD_SIZE=2 LOCAL_RANK=0 COPO_DIST_BACKEND=gloo COPO_DP_PROBE_VERBOSE=1
export COPO_DP_PROBE_HIDDEN=$H COPO_DP_PROBE_OBS=$O COPO_DP_PROBE_NETS=$N COPO_DP_PROBE_MB=$MB
RANK=0 timeout $T python -m copo_amd.dp_probe & p0=$!
RANK=1 timeout $T python -m copo_amd.dp_probe & p1=$!
wait $p0; r0=$?; wait $p1; r1=$?
echo "exit codes: $r0 $r1"
