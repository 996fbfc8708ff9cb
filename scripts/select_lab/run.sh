# builds and runs the selection lab on the GPU box: bash scripts/select_lab/run.sh
D=$(cd $(dirname $0) && pwd)
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$D/../../idiaptts_amd/csrc -o /tmp/select_lab $D/lab.hip && /tmp/select_lab
