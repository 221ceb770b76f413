set -o pipefail
mkdir -p gpurun_out/r04b
hipcc -O3 --offload-arch=gfx950 -o /tmp/fprobe tools/dev/fp32_mfma_filler_probe.hip && timeout -k 10 300 /tmp/fprobe > gpurun_out/r04b/fp32_mfma_filler_probe.txt 2>&1
cat gpurun_out/r04b/fp32_mfma_filler_probe.txt
GUARD_LOWPASS=1 timeout -k 10 400 python tools/dev/guard_explore.py 20:5 60:10 150:25 300:50 600:100 > gpurun_out/r04b/guard_explore_lowpass.txt 2>&1; tail -6 gpurun_out/r04b/guard_explore_lowpass.txt
