#!/bin/bash
# Round 4, VERDICT item 3: what does the N > 1 code path (RCCL all-gather on the tail stream) cost on ONE GPU?  Plain and forced-
# collective bench lines interleaved in one lease (same box), then RCCL channel caps, then a kernel trace of the forced run.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out; O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
run() { ( timeout 300 env "$@" $B 2>$O/$TAG.err ) > $O/$TAG.json; cut -c1-120 $O/$TAG.json | sed "s/^/$TAG /"; }
TAG=plain_a run X=1
TAG=coll_a run SLIME_BENCH_FORCE_COLLECTIVE=1
TAG=plain_b run X=1
TAG=coll_b run SLIME_BENCH_FORCE_COLLECTIVE=1
TAG=coll_ch1 run SLIME_BENCH_FORCE_COLLECTIVE=1 NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1
TAG=coll_ch4 run SLIME_BENCH_FORCE_COLLECTIVE=1 NCCL_MAX_NCHANNELS=4 NCCL_MIN_NCHANNELS=1
TAG=coll_nopipe ; ( timeout 300 env SLIME_BENCH_FORCE_COLLECTIVE=1 $B --no-pipeline 2>$O/$TAG.err ) > $O/$TAG.json; cut -c1-120 $O/$TAG.json | sed "s/^/$TAG /"
TAG=plain_nopipe ; ( timeout 300 $B --no-pipeline 2>$O/$TAG.err ) > $O/$TAG.json; cut -c1-120 $O/$TAG.json | sed "s/^/$TAG /"
cd /tmp; export TMPDIR=/tmp
( SLIME_BENCH_FORCE_COLLECTIVE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_coll" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline 2>"$R/$O/prof_coll.err" ) > "$R/$O/prof_coll.json"
cd "$R"
find $O/prof_coll -name "*kernel_stats.csv" -exec cp {} $O/coll_kernel_stats.csv \;
# the RCCL kernels' rows of the trace (name, grid, workgroup size, duration) before deleting the big file
python - <<'EOF' > gpurun_out/r4c/coll_rccl_launches.txt 2>&1
import csv, glob, collections
for f in glob.glob("gpurun_out/r4c/prof_coll/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if "ccl" in n.lower() or "gather" in n.lower() or "copy" in n.lower():
            agg[(n[:100], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        print(k, "launches", len(v), "avg_us", sum(v) / len(v) / 1e3, "max_us", max(v) / 1e3)
EOF
rm -rf $O/prof_coll
cat $O/coll_rccl_launches.txt; head -12 $O/coll_kernel_stats.csv | cut -c1-160
