#!/bin/bash
# Profiles of a round (run on the GPU box from the repository root): `bash profiles/collect.sh ROUND COMMIT [what ...]`
#   what = stats (rocprofv3 --kernel-trace --stats of c3 / c2 / c4 / page), pmc (FETCH_SIZE / WRITE_SIZE passes, separate runs, of
#   c3 / page / c4 / c2 -> profiles/*_traffic.json through pmc_traffic.py / pmc_traffic_step.py); default: both.
# Raw outputs go to gpurun_out/rNNprof (scratch); the summaries to be judged are COPIED into profiles/ by hand afterwards.
set -o pipefail
ROOT=$PWD
ROUND=$1; COMMIT=$2; shift 2
WHAT=${*:-stats pmc}
OUT=$ROOT/gpurun_out/r0${ROUND}prof
mkdir -p $OUT
export TMPDIR=/tmp
export CASV_PROFILE_ROUND=$ROUND CASV_PROFILE_COMMIT=$COMMIT
run() {  # name, rocprof args..., -- program args
  name=$1; shift
  ( cd $ROOT && timeout -k 10 400 rocprofv3 "$@" ) > $OUT/$name.log 2>&1 || echo "FAILED $name" | tee -a $OUT/failed.log
  echo "done $name"
}
export CASV_BENCH_NO_CALIBRATION=1
for what in $WHAT; do
  if [ $what = stats ]; then
    run c3_stats --kernel-trace --stats --output-format csv -d $OUT/c3 -o c3 -- python3 bench.py --steps 5 --warmup 2 --no-others --no-cpu-baseline
    run c3_fp32_stats --kernel-trace --stats --output-format csv -d $OUT/c3_fp32 -o c3_fp32 -- python3 bench.py --steps 3 --warmup 1 --no-others --no-cpu-baseline --arithmetic fp32
    run c2_stats --kernel-trace --stats --output-format csv -d $OUT/c2 -o c2 -- python3 bench.py --workload c2 --steps 20 --warmup 3 --no-others --no-cpu-baseline
    run c4_stats --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 bench.py --workload c4 --steps 5 --warmup 2 --no-others --no-cpu-baseline
    run page_stats --kernel-trace --stats --output-format csv -d $OUT/page -o page -- python3 bench.py --workload page --steps 3 --warmup 1 --no-others --no-cpu-baseline
  fi
  if [ $what = pmc ]; then
    for w in c3 page c4 c2; do
      run ${w}_fetch --pmc FETCH_SIZE --output-format csv -d $OUT/${w}_fetch -o fetch -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-others --no-cpu-baseline
      run ${w}_write --pmc WRITE_SIZE --output-format csv -d $OUT/${w}_write -o write -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-others --no-cpu-baseline
    done
    f() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
    python3 profiles/pmc_traffic.py $(f c3_fetch) $(f c3_write) gemm_split256_kernel split256_gemm_traffic.json 2 > $OUT/c3_traffic.txt 2>&1
    python3 profiles/pmc_traffic.py $(f page_fetch) $(f page_write) gemm_split256_kernel page_split_gemm_traffic.json 2 > $OUT/page_traffic.txt 2>&1
    python3 profiles/pmc_traffic.py $(f c2_fetch) $(f c2_write) persist_ persist_decode_traffic.json 1 > $OUT/c2_traffic.txt 2>&1
    python3 profiles/pmc_traffic_step.py $(f c4_fetch) $(f c4_write) 3 c4_step_traffic.json gemm_ > $OUT/c4_traffic.txt 2>&1
    cat $OUT/*_traffic.txt
    cp profiles/split256_gemm_traffic.json profiles/page_split_gemm_traffic.json profiles/persist_decode_traffic.json profiles/c4_step_traffic.json $OUT/   # (only gpurun_out travels back)
  fi
done
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*_stats.csv" | head -40
du -sh $OUT
