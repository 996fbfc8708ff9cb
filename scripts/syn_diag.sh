# diagnostic builds of the wave-per-pulse kernel (GPU box): bash scripts/syn_diag.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v synth.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for V in "-DSYN_WAVE_DIAG=0" "-DSYN_WAVE_DIAG=1" $2; do
  hipcc $FLAGS $V -c $R/idiaptts_amd/csrc/synth.hip -o /tmp/synth_v.o || exit 1
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/synth_v.o || exit 1
  echo "== $V" | tee -a $O/$1_syn_diag.txt
  python3 $R/scripts/syn_only.py 256 3 2>&1 | tail -6 | tee -a $O/$1_syn_diag.txt
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
