#!/bin/bash
# GPU box: the evidence set of a round -- kernel traces (three streams / one stream), counter summary, layer report, the default
# bench line, and the single-rank RCCL run beside a plain run on the same box.   usage: tools/final_profile.sh [tag]   (default r05c)
TAG=${1:-r06a}
tools/profile_step.sh ${TAG}_c2_resnet50_b256_bf16 > gpurun_out/${TAG}_profile.log 2>&1; tail -2 gpurun_out/${TAG}_profile.log
# the counter summary carries the build id of the library it profiled: placed under profiles/ (on this box too) the default
# bench line below replays it -- and only it (bench.py: a summary of another build is named as stale_profile, not replayed)
cp gpurun_out/${TAG}_c2_resnet50_b256_bf16_pmc.json profiles/${TAG}_pmc.json
KT_ONLY=1 MSFWSI_DUAL_STREAM=0 tools/profile_step.sh ${TAG}_c2_resnet50_b256_bf16_one_stream > gpurun_out/${TAG}_profile1.log 2>&1
timeout -k 10 400 python bench.py --steps 3 --warmup 2 --layer-report gpurun_out/${TAG}_layer_report.tsv > gpurun_out/${TAG}_bench_layers.json 2>/dev/null
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_default.json 2>gpurun_out/${TAG}_bench_default.err
PORT=29517
MSFWSI_FORCE_SYNC=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT timeout -k 10 400 python bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_force_sync_rccl.json 2>gpurun_out/${TAG}_force_sync.err
timeout -k 10 400 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_plain_same_box.json 2>/dev/null
# two ranks on the one card over gloo (RCCL refuses two ranks on one device): the multi-rank constructor (rank-0 broadcast of
# differently seeded replicas, collective probes, communicators) and the sharded step at a CPU-transport-sized batch
MSFWSI_BENCH_BACKEND=gloo MSFWSI_BENCH_DEVICE=0 timeout -k 10 600 python bench.py --gpus 2 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_gloo_rehearsal.json 2>gpurun_out/${TAG}_2rank.err
cut -c1-200 gpurun_out/${TAG}_bench_2rank_gloo_rehearsal.json
cut -c1-200 gpurun_out/${TAG}_bench_default.json gpurun_out/${TAG}_bench_force_sync_rccl.json gpurun_out/${TAG}_bench_plain_same_box.json
