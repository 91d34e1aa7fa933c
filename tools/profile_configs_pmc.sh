#!/bin/bash
# Hardware counters of tools/bench_configs.py for one named configuration.  Results: gpurun_out/cfgpmc_<name>/<pass>/
# usage: profile_configs_pmc.sh <config>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"
cd /tmp && export TMPDIR=/tmp
run() {
  pass=$1; shift
  OUT="$REPO/gpurun_out/cfgpmc_$name/$pass"; mkdir -p "$OUT"
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT" -o pmc --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$name" > "$OUT/../$pass.log" 2>&1 || { echo "pass $pass failed"; tail -5 "$OUT/../$pass.log"; exit 1; }
  echo "pass $pass done"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
