# Same-box A/B of the MLPG call: the library as built against one with scripts/ff_ab/mlpg_old.o (a copy of
# idiaptts_amd/_lib/mlpg.o taken before the change under test; see run.sh).  Round 4: the solve kernel forced to
# three waves per SIMD (168 registers, 2 spills) is 8-12 % SLOWER per call than at two (172 registers): 0.387 /
# 0.412 / 0.438 against 0.358 / 0.361 / 0.389 ms; the reduce kernel held to two waves per SIMD: no difference;
# the solve kernel held to ONE workgroup per CU (100 KB of unused dynamic LDS): 3-10 % slower.  Two waves per SIMD it stays.
R=$GRAFT_REPO_ROOT; L=$R/idiaptts_amd/_lib; cd /tmp
cp $L/libidiaptts_amd.so /tmp/lib_new.so
OBJS=$(ls $L/*.o | grep -v "/mlpg.o")
hipcc -shared -fPIC -pthread --offload-arch=gfx950 -o /tmp/lib_old.so $OBJS $R/scripts/ff_ab/mlpg_old.o
for rep in 1 2 3; do
  cp /tmp/lib_new.so $L/libidiaptts_amd.so; echo new; python3 $R/scripts/mlpg_curve.py 2>/dev/null | tail -3
  cp /tmp/lib_old.so $L/libidiaptts_amd.so; echo old; python3 $R/scripts/mlpg_curve.py 2>/dev/null | tail -3
done
cp /tmp/lib_new.so $L/libidiaptts_amd.so
