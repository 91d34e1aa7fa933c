#!/bin/bash
# Texture-addresser / L1 counters of tools/bench_configs.py for one configuration.  Results: gpurun_out/cfgmem_<name>/<pass>/
REPO="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"
cd /tmp && export TMPDIR=/tmp
run() {
  pass=$1; shift
  OUT="$REPO/gpurun_out/cfgmem_$name/$pass"; mkdir -p "$OUT"
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d "$OUT" -o pmc --output-format csv -- python3 "$REPO/tools/bench_configs.py" "$name" > "$OUT/../$pass.log" 2>&1 || { echo "pass $pass failed"; tail -5 "$OUT/../$pass.log"; exit 1; }
  echo "pass $pass done"
}
run mem1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
run mem2 TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run mem3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
