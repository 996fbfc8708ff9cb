# Solver variants (GPU box): segment length and reciprocal form.  bash scripts/solve_variants2.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
L=$R/idiaptts_amd/_lib
cp $L/libidiaptts_amd.so /tmp/lib_orig.so
OBJS=$(ls $L/*.o | grep -v mcep_lockstep.o)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=on -munsafe-fp-atomics"
for V in "8 0" "8 1" "4 0" "4 1" "2 1"; do
  set -- $V; SEG=$1; RCP=$2
  hipcc $FLAGS -DLS_SEG=$SEG -DLS_RCP=$RCP -c $R/idiaptts_amd/csrc/mcep_lockstep.hip -o /tmp/mcls_v.o || exit 1
  hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o $L/libidiaptts_amd.so $OBJS /tmp/mcls_v.o || exit 1
  echo "== LS_SEG $SEG LS_RCP $RCP" | tee -a $O/$TAG_solve_variants2.txt
  bash $R/scripts/analysis_prof.sh sv2 256 16000 2>&1 | grep -i "mcls_solve\|total kernel" | tee -a $O/$TAG_solve_variants2.txt
done
cp /tmp/lib_orig.so $L/libidiaptts_amd.so
