#!/bin/bash
# All the evidence of round 3 for profiles/r03 (run on the GPU box through gpurun; writes gpurun_out/<out>/):
#   scripts/profile_round3.sh <outdir-under-gpurun_out>
#  bench_c2.json + bench_c2_pmc/   the headline bench line; its roofline comes from rocprofv3 --pmc child passes (raw CSVs kept)
#  bench_c2_kernel_stats*.csv      rocprofv3 --kernel-trace --stats of the same command (--no-pmc: profilers do not nest), pipelined and not
#  bench_{head,c4,c3_256spp}.json (+ _pmc/), bench_c3_full.json, bench_c5_full.json   the other BASELINE configs
#  bench_head_static_boxes.json    HEAD Book-1 on the reference's boxes (RTX_MOTION=0 RTX_MOTION_TOPOLOGY=0): what the time-aware boxes buy
out=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$out
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 --keep-pmc $O/bench_c2_pmc > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
echo "[profile] c2 done"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline > $O/bench_c2_stats_run.json 2> $O/stats.err ) || exit 1
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/bench_c2_kernel_stats.csv && rm -rf $O/stats
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --steps 20 --warmup 5 --no-pmc --no-extras --no-cpu-baseline --frames-in-flight 1 > $O/bench_c2_stats_run_unpipelined.json 2> $O/stats1.err ) || exit 1
cp $(ls $O/stats1/*/*_kernel_stats.csv | head -1) $O/bench_c2_kernel_stats_unpipelined.csv && rm -rf $O/stats1
echo "[profile] stats done"
python3 bench.py --workload head --steps 5 --warmup 1 --no-cpu-baseline --no-extras --keep-pmc $O/bench_head_pmc > $O/bench_head.json 2>> $O/bench.err || exit 1
RTX_MOTION=0 RTX_MOTION_TOPOLOGY=0 python3 bench.py --workload head --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-pmc > $O/bench_head_static_boxes.json 2>> $O/bench.err || exit 1
python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --keep-pmc $O/bench_c4_pmc > $O/bench_c4.json 2>> $O/bench.err || exit 1
echo "[profile] head, c4 done"
python3 bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-pmc > $O/bench_c3_full.json 2>> $O/bench.err || exit 1
python3 bench.py --workload c3 --spp 256 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --keep-pmc $O/bench_c3_pmc > $O/bench_c3_256spp.json 2>> $O/bench.err || exit 1
echo "[profile] c3 done"
python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --no-extras --no-pmc --no-count > $O/bench_c5_full.json 2>> $O/bench.err || exit 1
python3 bench.py --gpus 8 --single-process --same-device --workload c2 --steps 5 --warmup 1 > $O/bench_single_process_8_shards_one_device.json 2>> $O/bench.err || exit 1
echo "[profile] c5, single-process rehearsal done"
