# Round profile on the GPU box: kernel trace of the default bench, the two PMC traffic passes, the bench line.
#   gpurun -- 'bash tools/profile_round.sh'      (outputs under gpurun_out/, summaries copied to profiles/ by hand)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o train -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_train_bench.json 2> $R/gpurun_out/prof_train.err
echo "kernel trace rc=$?"; tail -c 300 $R/gpurun_out/prof_train.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/tools/traffic_probe.py > $R/gpurun_out/pmc_f.log 2>&1
echo "fetch pass rc=$?"; tail -c 200 $R/gpurun_out/pmc_f.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/tools/traffic_probe.py > $R/gpurun_out/pmc_w.log 2>&1
echo "write pass rc=$?"; tail -c 200 $R/gpurun_out/pmc_w.log
cd $R
python3 tools/traffic_report.py gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/hbm_traffic.json > gpurun_out/traffic_report.log 2>&1
echo "report rc=$?"; tail -3 gpurun_out/traffic_report.log
du -sh gpurun_out/* | sort -h | tail -5
# keep the summaries only (the raw traces exceed what gpurun copies back)
find gpurun_out/prof_train -type f ! -name "*kernel_stats.csv" -delete
find gpurun_out/pmc_f gpurun_out/pmc_w -type f ! -name "*counter_collection.csv" -delete
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "bench rc=$?"; tail -c 400 gpurun_out/bench_final.json
du -sh gpurun_out
