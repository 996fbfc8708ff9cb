# Solver time of the WORLD analysis with the pivot row's first k/16 entries through scalar
# registers instead of LDS (run on the GPU box): bash scripts/solve_variants.sh <tag> "0 2 4 6 8"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v mcep_lockstep.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for V in $2; do
  hipcc $FLAGS -DLS_READLANE_16THS=$V -c $R/idiaptts_amd/csrc/mcep_lockstep.hip -o /tmp/mcls_$V.o || exit 1
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/mcls_$V.o || exit 1
  echo "== LS_READLANE_16THS $V" | tee -a $O/$1_solve_variants.txt
  bash $R/scripts/analysis_prof.sh $1_v$V 256 16000 2>&1 | grep -i "mcls_solve\|total kernel" | tee -a $O/$1_solve_variants.txt
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
