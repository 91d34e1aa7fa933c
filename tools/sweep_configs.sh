#!/bin/bash
# Runs tools/bench_configs.py once per library variant built by sweep_build.sh.
cd "$(dirname "$0")/.."
for f in raytrace_amd/variants/*.so; do
  echo "== $f"
  RT_LIB_PATH=$PWD/$f timeout -k 10 200 python -u tools/bench_configs.py "$@" 2>/dev/null | cut -c1-60 || exit 1
done
