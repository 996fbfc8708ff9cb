# build and run the wave-FFT lab on the GPU box; output -> gpurun_out/<tag>_wave_fft_lab.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R/scripts/wave_fft_lab
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -munsafe-fp-atomics -Wno-unused-function -o /tmp/wave_fft_lab lab.hip || exit 1
/tmp/wave_fft_lab 2>&1 | tee $O/${1:-lab}_wave_fft_lab.txt
