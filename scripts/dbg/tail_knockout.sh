# which of the tail kernel's reads bound it: bit 0 = a0 from one pixel, bit 1 = d2 / a1 from one pixel, bit 2 = log-magnitudes
# from one sample (diagnostic build: results are wrong with any bit set)
for v in 0 1 2 4 7; do
  NOISE_TAIL_DBG=$v timeout -k 10 200 python scripts/bench_infer.py A 256 bf16 2>&1 | grep -E "fused_tail" | sed "s/^/dbg=$v /" || exit 1
done
