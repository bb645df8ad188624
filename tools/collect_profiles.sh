#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box:  bash tools/collect_profiles.sh round3_a [extra bench.py flags]
# Every pass is its own run (kernel-trace/stats, FETCH_SIZE, WRITE_SIZE, three SQ counter groups); the VALU issue time
# is measured on the SAME box by tools/microbench/valu_rate.  The per-kernel passes run the fleet on ONE stream
# (--groups 1: one full-fleet launch per kernel and step, nothing overlapping - per-kernel times and counters mean what
# they say); one more kernel-trace pass runs the default stream groups and records how far their kernels overlap.
# Outputs: gpurun_out/profiles_<tag>/<tag>_*.{csv,json,txt}; copy the summaries (not the raw per-dispatch PMC rows) into profiles/.
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
tag=$1; shift || true
extra="$@"
out=gpurun_out/profiles_$tag; mkdir -p $out
BENCH="python3 bench.py --no-cpu-baseline --no-single --groups 1 $extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- $BENCH --steps 10 --warmup 2 > $out/${tag}_bench_under_rocprof.json 2> $out/stats.err
cp $out/stats/r_kernel_stats.csv $out/${tag}_kernel_stats.csv
python3 tools/timeline_gaps.py $out/stats/r_kernel_trace.csv > $out/${tag}_timeline_gaps.txt 2>&1 || true
# the default schedule (stream groups): which kernels run side by side, and each kernel's duration in that company
rocprofv3 --kernel-trace --stats --output-format csv -d $out/groups -o r -- python3 bench.py --no-cpu-baseline --no-single $extra --steps 20 --warmup 2 > $out/${tag}_bench_groups_under_rocprof.json 2> $out/groups.err
cp $out/groups/r_kernel_stats.csv $out/${tag}_kernel_stats_groups.csv
python3 tools/overlap_trace.py $out/groups/r_kernel_trace.csv > $out/${tag}_overlap_groups.txt 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o r -- $BENCH --steps 3 --warmup 1 > $out/$c.json 2> $out/$c.err
  lc=$(echo $c | tr A-Z a-z)
  cp $out/$c/r_counter_collection.csv $out/${tag}_pmc_$lc.csv
done
# SQ passes (8 SQ slots per pass): instruction counts; active cycles; where the waves wait + LDS bank conflicts
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$n -o r -- $BENCH --steps 3 --warmup 1 > $out/$n.json 2> $out/$n.err || { echo "PMC pass $n failed"; tail -3 $out/$n.err; }
  [ -f $out/$n/r_counter_collection.csv ] && cp $out/$n/r_counter_collection.csv $out/${tag}_pmc_$(echo $n | tr A-Z a-z).csv
done
# the chip's VALU issue time per wave-instruction and SIMD, measured here and now
if [ ! -x tools/microbench/valu_rate ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate; fi
./tools/microbench/valu_rate > $out/${tag}_valu_rate_microbench.txt
python3 tools/pmc_summary.py "$out" "$tag"
# which build these numbers belong to (bench.py quotes the profile profiles/CURRENT.json names and flags it when the sources have moved on)
python3 tools/profile_tag.py "$tag" > $out/${tag}_profile_tag.json
echo "copy $out/${tag}_*.{csv,json,txt} into profiles/ and ${tag}_profile_tag.json to profiles/CURRENT.json"
