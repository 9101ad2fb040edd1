#!/bin/bash
# Second half of a round's profile set, after scripts/gpu_profile.sh <tag>: SQ and TA/TCP counter passes on ONE pass of 32 spp at 1080p
# (66 M paths: the bench's launch size), the kernel trace + FETCH / WRITE passes of the three other BASELINE configs at full size, and the
# timing of one GPU's share of an 8-GPU frame (128 spp).  usage: gpu_round_counters.sh <tag>.  Results under gpurun_out/.
set -e -o pipefail
TAG=${1:-r03e}
cd "$GRAFT_REPO_ROOT"
SQ="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES"
TA="GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
mkdir -p gpurun_out/$TAG
PROF_SPP=32 PROF_S=32 bash scripts/gpu_pmc.sh ${TAG}_sq32 "$SQ" > gpurun_out/$TAG/sq32.log 2>&1
PROF_SPP=32 PROF_S=32 bash scripts/gpu_pmc.sh ${TAG}_ta32 "$TA" > gpurun_out/$TAG/ta32.log 2>&1
if [ -z "$SKIP_CONFIGS" ]; then
    bash scripts/gpu_profile_config.sh $TAG soup 64 > gpurun_out/$TAG/soup.log 2>&1
    bash scripts/gpu_profile_config.sh $TAG instanced 32 > gpurun_out/$TAG/instanced.log 2>&1
    bash scripts/gpu_profile_config.sh $TAG spheres 64 > gpurun_out/$TAG/spheres.log 2>&1
    for c in instanced spheres; do
        PROF_SPP=8 PROF_S=8 PROF_CONFIG=$c bash scripts/gpu_pmc.sh ${TAG}_${c}_sq "$SQ" > gpurun_out/$TAG/${c}_sq.log 2>&1
    done
fi
timeout -k 10 200 python3 bench.py --spp 128 --steps 10 --warmup 2 --cpu-spp 0 --no-alt --no-others > gpurun_out/$TAG/share128.json 2> gpurun_out/$TAG/share128.err
echo done
