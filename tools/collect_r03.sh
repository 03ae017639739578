#!/bin/bash
# Round-3 measurement records (run on the GPU box from the repo root): full GPU suite, smoke, headline bench, kernel stats,
# PMC traffic of the sparse engines, the other workloads.  Outputs under gpurun_out/r03/ (copied to profiles/r03/ afterwards).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
# PMC traffic of the sparse engines FIRST: bench.py reports roofline.traffic only from a record whose fingerprint matches this tree
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 tools/prof3d.py --bench-batch --steps 3 --warmup 1 > $O/prof3d_fetch.log 2>&1; echo "pmc fetch rc=$?" | tee -a $O/summary.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 tools/prof3d.py --bench-batch --steps 3 --warmup 1 > $O/prof3d_write.log 2>&1; echo "pmc write rc=$?" | tee -a $O/summary.txt
ALG=$(grep algorithmic_bytes_per_step $O/prof3d_fetch.log | awk '{print $2}')
# profiled steps: 1 warm-up + 1 accounting step + 3 timed = 5
python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv 5 16 $ALG "$1" > $O/traffic_3d.json 2> $O/traffic.err; echo "traffic rc=$?" | tee -a $O/summary.txt
mkdir -p profiles/r03 && cp $O/traffic_3d.json profiles/r03/traffic_3d.json
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
MM_BENCH_LAYERS=1 python bench.py --steps 10 --warmup 3 > /dev/null 2> $O/layers.err; grep "\[layer\]" $O/layers.err > $O/engines_per_layer_final.txt
rocprofv3 --kernel-trace --stats -d $O/kstats -o ks --output-format csv -- python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_profiled.json 2> $O/bench_n1_profiled.err; echo "rocprof rc=$?" | tee -a $O/summary.txt
cp $O/kstats/ks_kernel_stats.csv $O/bench_n1_kernel_stats.csv 2>/dev/null
# the same with the sparse backward on ONE stream: per-kernel durations that the roofline leg's per-call event times must agree with
MM_SPCONV_BWD_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $O/kstats_serial -o ks --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_n1_serial_profiled.json 2> $O/bench_n1_serial_profiled.err; echo "rocprof serial rc=$?" | tee -a $O/summary.txt
cp $O/kstats_serial/ks_kernel_stats.csv $O/bench_n1_serial_kernel_stats.csv 2>/dev/null
python bench.py --workload c4 --steps 10 --warmup 3 > $O/bench_c4.json 2>/dev/null; python bench.py --workload c5 --steps 10 --warmup 3 > $O/bench_c5.json 2>/dev/null
python bench.py --workload c5 --sparse-act fp16 --precision fp16 --steps 10 --warmup 3 > $O/bench_c5_fp16.json 2>/dev/null
python bench.py --precision fp16 --steps 20 --warmup 5 --no-extras > $O/bench_n1_fp16.json 2>/dev/null
python bench.py --image 400x225 --steps 15 --warmup 4 > $O/bench_n1_image_400x225.json 2>/dev/null
rm -rf $O/kstats/*trace* $O/kstats_serial/*trace* $O/pmc_fetch/*trace* $O/pmc_write/*trace* 2>/dev/null
ls -la $O | tail -24
