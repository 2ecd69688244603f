#!/bin/bash
# round 6, GPU session 15: the bf16 engine's new comm-stream work (per-bucket casts, input ahead) with a communicator:
# a one-rank RCCL communicator (bit-identical to no communicator) and the two-rank launch rehearsal on one GPU
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s15
mkdir -p $O
cd $R
echo -n "no communicator:       "; python tools/bf16_bench.py 256 60 1 2>/dev/null | tail -1
echo -n "one-rank communicator: "; DV_FORCE_COMM=1 python tools/bf16_bench.py 256 60 1 2>/dev/null | tail -1
DEBVADER_AMD_LIB=$R/debvader_amd/lib/libdebvader_hip_debug.so DV_DEBUG_SAME_GPU=1 DV_DEBUG_FAKE_PEERS=1 timeout -k 10 300 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 \
  --config 2 --steps 20 --warmup 5 --no-roofline --no-secondary --no-cpu-baseline > $O/rehearsal_bf16.json 2> $O/rehearsal_bf16.err
echo "rehearsal rc=$?"
tail -c 900 $O/rehearsal_bf16.json
