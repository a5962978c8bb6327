#!/bin/bash
# Second half of the evidence under gpurun_out/prof (run on the GPU box from the repo root, after profile_round.sh):
#   1. sharded synthesis (SNR mix + clip check + STFT 320/160 + STFT 512/128) on resident audio: entry-point table,
#      kernel stats, HBM traffic in separate counter passes
#   2. accuracy and rates of the FFT STFT kernels with library / hardware-rate transcendental functions
#   3. stencil walkers against the number of resident workgroups
#   4. measured peaks, the other BASELINE configurations, waveform augmentation (+ its rocprofv3 passes), OFA forward
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
timeout -k 10 200 python3 scripts/bench_synth.py 1024 1024 > $O/synth_1024.txt 2>&1
timeout -k 10 200 python3 scripts/bench_synth.py 4096 1024 > $O/synth_4096.txt 2>&1
timeout -k 10 200 python3 scripts/stft_accuracy.py > $O/stft_accuracy.txt 2>&1
WALK_SWEEP=occupancy timeout -k 10 200 python3 scripts/walk_sweep.py > $O/walk_occupancy.txt 2>&1
timeout -k 10 200 python3 scripts/peaks.py > $O/peaks.log 2>&1 && cp gpurun_out/measured_peaks.json $O/measured_peaks.json
timeout -k 10 300 python3 scripts/measure_configs.py > $O/other_configs.json 2> $O/other_configs.err
timeout -k 10 200 python3 scripts/bench_augment.py 256 1.0 --cpu > $O/augment_256x10s.txt 2>&1
timeout -k 10 200 python3 scripts/bench_ofa.py 64 7 > $O/ofa_forward.txt 2>&1
timeout -k 10 200 python3 scripts/bench_ofa.py 64 3 >> $O/ofa_forward.txt 2>&1
timeout -k 10 300 python3 scripts/bench_wave.py 1280 8000 > $O/wave_unet.txt 2>&1
timeout -k 10 300 python3 scripts/bench_wave.py 64 160000 >> $O/wave_unet.txt 2>&1
timeout -k 10 300 python3 scripts/bench_wave.py 256 8000 >> $O/wave_unet.txt 2>&1
bash scripts/profile_augment.sh > $O/profile_augment.log 2>&1
cd /tmp && export TMPDIR=/tmp
S="$R/scripts/bench_synth.py 512 512"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/skt -o p -- python3 $S > $O/skt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/spmc/f -o p -- python3 $S > $O/spmc_f.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/spmc/w -o p -- python3 $S > $O/spmc_w.log 2>&1
python3 $R/scripts/pmc_traffic.py $O/spmc $O/synth_hbm_traffic_per_kernel.json > $O/synth_hbm_traffic.txt
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wkt -o p -- python3 $R/scripts/bench_wave.py 256 8000 > $O/wkt.log 2>&1
cat $O/synth_1024.txt
