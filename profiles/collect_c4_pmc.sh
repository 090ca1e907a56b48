#!/bin/bash
# HBM-side traffic of a train step (configs[3]) from two rocprofv3 PMC passes -> profiles/c4_step_traffic.json
#   bash profiles/collect_c4_pmc.sh ROUND COMMIT     (on the GPU box, from the repository root; the JSON is also copied to gpurun_out/)
set -o pipefail
export TMPDIR=/tmp CASV_BENCH_NO_CALIBRATION=1 CASV_PROFILE_ROUND=$1 CASV_PROFILE_COMMIT=$2
OUT=$PWD/gpurun_out/r0$1prof
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rm -rf $OUT/c4_${n}
  ( timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/c4_${n} -o $n -- python3 bench.py --workload c4 --steps 1 --warmup 1 --no-others --no-cpu-baseline ) > $OUT/c4_${n}.log 2>&1 || echo "FAILED $c"
done
f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python3 profiles/pmc_traffic_step.py $(f c4_fetch) $(f c4_write) 3 c4_step_traffic.json gemm_ | cut -c1-400
cp profiles/c4_step_traffic.json $OUT/c4_step_traffic.json
