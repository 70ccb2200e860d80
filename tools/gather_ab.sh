#!/bin/bash
# One-rank RCCL group on the box's GPU: the overlapped output all_gather of bench.py against the in-line one.
set -e
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517
for i in 1 2; do
  timeout -k 10 200 python bench.py --force-gather --overlap-gather --no-cpu-baseline --no-insitu --steps 10 | tee -a gpurun_out/gather_ab.txt
  timeout -k 10 200 python bench.py --force-gather --no-cpu-baseline --no-insitu --steps 10 | tee -a gpurun_out/gather_ab.txt
done
