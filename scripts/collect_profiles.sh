#!/bin/bash
# Copy the summaries written by scripts/profile_round.sh (gpurun_out/prof) into profiles/, named per round.
R=${1:-r2}
P=gpurun_out/prof
cp $P/bench_line.json profiles/${R}_bench_line.json
cp $P/entry_point_breakdown.txt profiles/${R}_entry_point_breakdown.txt
cp $P/kt/p_kernel_stats.csv profiles/${R}_train_step_kernel_stats.csv
cp $P/hbm_traffic.txt profiles/${R}_hbm_traffic.txt
cp $P/hbm_traffic_per_kernel.json profiles/${R}_hbm_traffic_per_kernel.json
cp $P/mfma_utilisation.txt profiles/${R}_mfma_utilisation.txt
cp $P/infer_A.txt profiles/${R}_infer_320_160.txt
cp $P/infer_B.txt profiles/${R}_infer_512_128.txt
cp $P/ikt/p_kernel_stats.csv profiles/${R}_infer_bf16_kernel_stats.csv
cp $P/infer_hbm_traffic.txt profiles/${R}_infer_bf16_hbm_traffic.txt
cp $P/infer_hbm_traffic_per_kernel.json profiles/${R}_infer_bf16_hbm_traffic_per_kernel.json
cp $P/infer_mfma_utilisation.txt profiles/${R}_infer_bf16_mfma_utilisation.txt
# written by scripts/profile_round_extra.sh
for f in synth_1024.txt synth_4096.txt stft_accuracy.txt walk_occupancy.txt synth_hbm_traffic.txt synth_hbm_traffic_per_kernel.json \
         measured_peaks.json other_configs.json augment_256x10s.txt ofa_forward.txt wave_unet.txt; do
  [ -f $P/$f ] && cp $P/$f profiles/${R}_$f
done
[ -f $P/skt/p_kernel_stats.csv ] && cp $P/skt/p_kernel_stats.csv profiles/${R}_synth_kernel_stats.csv
[ -f $P/wkt/p_kernel_stats.csv ] && cp $P/wkt/p_kernel_stats.csv profiles/${R}_wave_unet_kernel_stats.csv
A=gpurun_out/prof_aug
[ -f $A/kt/p_kernel_stats.csv ] && cp $A/kt/p_kernel_stats.csv profiles/${R}_augment_kernel_stats.csv
[ -f $A/hbm_traffic.txt ] && cp $A/hbm_traffic.txt profiles/${R}_augment_hbm_traffic.txt
true
