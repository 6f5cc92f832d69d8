mkdir -p gpurun_out/r3o
bash tools/power_trace.sh $PWD/gpurun_out/r3o/power_trace.txt
grep -c "t=" gpurun_out/r3o/power_trace.txt; sed -n 1,6p gpurun_out/r3o/power_trace.txt; grep -A3 "load starts" gpurun_out/r3o/power_trace.txt | head -8; tail -5 gpurun_out/r3o/power_trace.txt
python tools/pmc_traffic.py $PWD/gpurun_out/r3o/pmc > gpurun_out/r3o/pmc_summary.txt 2>&1
tail -30 gpurun_out/r3o/pmc_summary.txt
cp gpurun_out/r3o/pmc/r03_pmc_traffic.json gpurun_out/r3o/ 2>/dev/null
rm -rf gpurun_out/r3o/pmc
bash tools/profile_bench.sh r03 > gpurun_out/r3o/profile_bench.log 2>&1
tail -12 gpurun_out/r3o/profile_bench.log
