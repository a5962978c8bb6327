# pw_virt_bwd_kernel at one workgroup per CU (extra dynamic LDS on a diagnostic build) against the shipped two
for v in 0 25000; do
  export NOISE_DIAG_SMEM=$v
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --warmup 4 --breakdown 2>&1 >/dev/null | grep -E "noise_pw_bwd_virt" | sed "s/^/extra_lds=$v /" || exit 1
done
