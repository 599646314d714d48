# Round profile on the GPU box (outputs under gpurun_out/prof_<TAG>/, summaries copied to profiles/ by hand).
#   gpurun -- 'bash tools/profile_round.sh r03 "C*"'            kernel trace + the two HBM-traffic PMC passes + SQ counters
#   gpurun -- 'bash tools/profile_round.sh r03 "C*-2x64" mfma'  ... + the matrix-pipe counters of the two-layer kernels
# Counters are collected in their own runs (no trace options beside --pmc); the program itself follows `--`.
TAG=${1:-r}; WL=${2:-C*}; EXTRA=$3
R=$GRAFT_REPO_ROOT
N=$(echo "$WL" | sed 's/\*/star/')
O=$R/gpurun_out/prof_${TAG}_$N
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PSVO_WORKLOAD="$WL"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o train -- python3 $R/bench.py --workload "$WL" --no-cpu-baseline > $O/bench_under_trace.json 2> $O/trace.err
echo "kernel trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -- python3 $R/tools/traffic_probe.py > $O/pmc_f.log 2>&1; echo "fetch pass rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -- python3 $R/tools/traffic_probe.py > $O/pmc_w.log 2>&1; echo "write pass rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- python3 $R/tools/traffic_probe.py > $O/pmc_sq.log 2>&1; echo "sq pass rc=$?"
if [ "$EXTRA" = "mfma" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/pmc_mfma -- python3 $R/tools/traffic_probe.py > $O/pmc_mfma.log 2>&1; echo "mfma pass rc=$?"
fi
cd $R
python3 tools/traffic_report.py $O/pmc_f $O/pmc_w $O/hbm_traffic.json > $O/traffic_report.log 2>&1; echo "traffic report rc=$?"; tail -2 $O/traffic_report.log
python3 tools/pmc_sq_report.py $O/pmc_sq $O/sq_counters.json > $O/sq_report.log 2>&1; echo "sq report rc=$?"
# keep the summaries only (the raw traces exceed what gpurun copies back)
find $O/trace -type f ! -name "*kernel_stats.csv" -delete
find $O/pmc_f $O/pmc_w $O/pmc_sq $O/pmc_mfma -type f ! -name "*counter_collection.csv" -delete 2>/dev/null
du -sh $O
