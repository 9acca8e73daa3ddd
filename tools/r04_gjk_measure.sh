#!/bin/bash
# Round 4: what RP_CFG_HULL_GJK costs - k_prep2 phase clocks and bench lines with and without it, both action distributions.   gpurun -- 'bash tools/r04_gjk_measure.sh <tag>'
TAG=${1:-x}
OUT=gpurun_out/r04_gjk_$TAG
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "hull_gjk or split_pipeline or rollout_200 or reset_parity or solver_slot or torsional" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
RP_PLAYROOM_LIB=$PWD/tools/clocks2.so python tools/gpu_clocks_prep.py > $OUT/clocks_gjk.txt 2>&1
RP_NO_GJK=1 RP_PLAYROOM_LIB=$PWD/tools/clocks2.so python tools/gpu_clocks_prep.py > $OUT/clocks_nogjk.txt 2>&1
DIST=A RP_PLAYROOM_LIB=$PWD/tools/clocks2.so python tools/gpu_clocks_prep.py > $OUT/clocks_gjk_A.txt 2>&1
RP_NO_GJK=1 DIST=A RP_PLAYROOM_LIB=$PWD/tools/clocks2.so python tools/gpu_clocks_prep.py > $OUT/clocks_nogjk_A.txt 2>&1
python bench.py --no-cpu-baseline > $OUT/bench_gjk.json 2>&1
RP_NO_GJK=1 python bench.py --no-cpu-baseline > $OUT/bench_nogjk.json 2>&1
grep -h "GJK\|hull pairs" $OUT/clocks_gjk.txt $OUT/clocks_gjk_A.txt
