#!/bin/bash
# rocprofv3 --kernel-trace --stats of one non-headline bench workload: scripts/profile_workload.sh <tag> <bench args...>
# -> gpurun_out/prof_<tag>/{bench.json,kernel_stats.csv}
tag=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/bench_stats.log 2>&1 || exit 1
grep '^{' $O/bench_stats.log > $O/bench.json
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/stats
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"
