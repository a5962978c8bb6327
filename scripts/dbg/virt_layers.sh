for v in 1 72,96 72 ; do
  NOISE_VIRTUAL_DW=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])" || exit 1
done
