# What the (boundary + 1)-largest selection costs inside d4c_kernel: the kernel timed with the selection
# rounds cut out (results wrong, lab only).  GPU box: bash scripts/d4c_topk_lab.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v world_f0ap.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
sed 's/for (int round = 0; round <= boundary; ++round) {/for (int round = 0; round < 0; ++round) {/' $R/idiaptts_amd/csrc/world_f0ap.hip > $R/idiaptts_amd/csrc/_lab_f0ap.hip
hipcc $FLAGS -c $R/idiaptts_amd/csrc/_lab_f0ap.hip -o /tmp/f0ap_lab.o || exit 1
rm -f $R/idiaptts_amd/csrc/_lab_f0ap.hip
for V in orig lab; do
  if [ $V = lab ]; then hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/f0ap_lab.o || exit 1; fi
  for FS in 16000 48000; do
    echo "== $V $FS" | tee -a $O/$1_d4c_topk.txt
    bash $R/scripts/analysis_prof.sh $1_$V 256 $FS 2>&1 | grep -i "d4c_kernel\|total kernel" | tee -a $O/$1_d4c_topk.txt
  done
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
