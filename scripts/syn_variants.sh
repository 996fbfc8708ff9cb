# Per-kernel time of the synthesis with the wave-per-pulse kernel built at one and at two waves per
# SIMD (run on the GPU box): bash scripts/syn_variants.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v synth.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for OCC in 1 2; do
  hipcc $FLAGS -DSYN_WAVE_OCC=$OCC -c $R/idiaptts_amd/csrc/synth.hip -o /tmp/synth_$OCC.o || exit 1
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/synth_$OCC.o || exit 1
  echo "== SYN_WAVE_OCC $OCC" | tee -a $O/$1_syn_variants.txt
  bash $R/scripts/world_prof.sh 16000 256 2>&1 | grep -i "syn_\|mgc2sp\|decode_ap\|gemm_f64" | head -14 | tee -a $O/$1_syn_variants.txt
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
