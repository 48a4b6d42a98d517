#!/bin/bash
# round-4 GPU batch 1: the new in-flight test, the bench with lane-private inputs (+ A/B against shared inputs), the mixing
# kernel walking its items in reverse (Infinity-Cache experiment), BEV phase stamps and PMC passes
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4b1; mkdir -p $out
timeout -k 10 400 python3 -m pytest tests/test_graph_gpu.py -x -q -m gpu > $out/test_graph.log 2>&1; echo "test_graph rc=$?"; tail -3 $out/test_graph.log
timeout -k 10 500 python3 bench.py --no-cpu-baseline --no-stress > $out/bench_distinct.json 2> $out/bench_distinct.err; echo "bench distinct rc=$?"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-stress --same-inputs-per-lane > $out/bench_same.json 2> $out/bench_same.err; echo "bench same rc=$?"
python3 - <<'PY'
import json
for n in ("distinct","same"):
    try:
        d=json.load(open(f"gpurun_out/r4b1/bench_{n}.json"))
        print(n, "value", d["value"], "one", d["one_sample_in_flight"]["value"], "match", d.get("lanes_match_single_plan_bitwise"), d["config"].get("distinct_inputs_per_lane"))
    except Exception as e: print(n, "ERR", e)
PY
AB_ARGS="--in-flight 1" tools/ab_bench.sh r4b1_ab default build/lib_mixrev.so default build/lib_mixrev.so
AB_ARGS="" tools/ab_bench.sh r4b1_ab4 default build/lib_mixrev.so
RACFORMER_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_bevstamps.so timeout -k 10 200 python3 tools/bev_phase_split.py $out/bev_phase_split.json > $out/bev_phase.log 2>&1; echo "phase rc=$?"; tail -5 $out/bev_phase.log
PMC_JSON=$GRAFT_REPO_ROOT/$out/pmc_bev.json tools/gpu_pmc_kernel.sh r4b1_pmc bev_sampling \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL" \
  "TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
for v in default msmv_pk msmv_pk_loadsfirst; do
  if [ $v = default ]; then unset RACFORMER_HIP_LIB; else export RACFORMER_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$v.so; fi
  timeout -k 10 240 python3 tools/race_victims.py $out/race_$v.json > $out/race_$v.log 2>&1; echo "race $v rc=$?"; cat $out/race_$v.log | grep "deviating"
done
unset RACFORMER_HIP_LIB
