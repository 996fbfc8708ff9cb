# CheapTrick wave kernel at different workgroup sizes (run on the GPU box): bash scripts/ct_variants.sh <tag> "512 768 1024"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v world_frame.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for V in $2; do
  hipcc $FLAGS -DCTW_THREADS_N=$V -c $R/idiaptts_amd/csrc/world_frame.hip -o /tmp/wfr_$V.o || exit 1
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/wfr_$V.o || exit 1
  echo "== CTW_THREADS $V" | tee -a $O/$1_ct_variants.txt
  bash $R/scripts/analysis_prof.sh $1_t$V 256 16000 2>&1 | grep -i "cheaptrick\|total kernel" | tee -a $O/$1_ct_variants.txt
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
