cd "$(dirname "$0")/.."
for rep in 1 2; do for f in raytrace_amd/variants/*.so; do for W in 1 8; do
  RT_LIB_PATH=$PWD/$f timeout -k 10 200 python bench.py --emulate-world $W --steps 6 --warmup 2 --no-cpu-baseline --no-culling-leg --no-cold-frame 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f W=$W', round(d['ms_per_step'],2), 'ms/step')"
done; done; done
