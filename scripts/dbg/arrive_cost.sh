# cost of a fence + arrival counter at the end of every pw_gemm workgroup (diagnostic build)
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export NOISE_DIAG_ARRIVE=1; else unset NOISE_DIAG_ARRIVE; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --breakdown 2>&1 >/dev/null | grep -E "^noise_pw_fwd " | sed "s/^/diag=$v /" || exit 1
done
