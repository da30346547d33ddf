#!/bin/bash
# the streamed and the two-phase host-visible call on C5 under settings of the HIP runtime's copy path (process-wide environment
# variables of the runtime, read at start-up): does any of them keep the device-to-host copies off the compute units?
mkdir -p gpurun_out/knobs
run() { # name, env assignments...
  local name=$1; shift
  env "$@" UGS_PROBE_ONLY_DEFAULT=1 timeout -k 10 120 python tools/streamed_call_probe.py c5_er_1m 5 > gpurun_out/knobs/$name.json 2> gpurun_out/knobs/$name.err || { echo "$name FAILED"; tail -2 gpurun_out/knobs/$name.err; return 0; }
  python -c "
import json; d=json.load(open('gpurun_out/knobs/$name.json')); print('$name', 'two-phase', d['two_phase']['median_ms'], 'streamed', d['streamed']['default(rows/8)']['median_ms'], 'equal', d['streamed_equals_two_phase'])"
}
run default UGS_DUMMY=1
run limit_blit_wg_16 DEBUG_CLR_LIMIT_BLIT_WG=16
run limit_blit_wg_64 DEBUG_CLR_LIMIT_BLIT_WG=64
run blit_engine_2 GPU_BLIT_ENGINE_TYPE=2
run force_blit_size_0 GPU_FORCE_BLIT_COPY_SIZE=0
run sdma_on HSA_ENABLE_SDMA=1
