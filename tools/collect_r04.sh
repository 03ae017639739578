#!/bin/bash
# Round-4 measurement records (run on the GPU box from the repo root; $1 = git commit of the tree, the box has no .git):
# smoke, headline bench, kernel stats (two-stream and serial), PMC traffic of the sparse engines, SQ counters of the 2D kernels,
# the other workloads, probes.  Outputs under gpurun_out/r04/ (copied to profiles/r04/ afterwards).  The full GPU test suite is a
# separate call (8 minutes).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MM_GIT_HEAD="$1"
O=gpurun_out/r04; mkdir -p $O profiles/r04
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
# PMC records FIRST: bench.py quotes roofline.traffic / mfma_busy only from records whose source fingerprints match this tree
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 tools/prof3d.py --bench-batch --steps 3 --warmup 1 > $O/prof3d_fetch.log 2>&1; echo "pmc fetch rc=$?" | tee -a $O/summary.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 tools/prof3d.py --bench-batch --steps 3 --warmup 1 > $O/prof3d_write.log 2>&1; echo "pmc write rc=$?" | tee -a $O/summary.txt
ALG=$(grep algorithmic_bytes_per_step $O/prof3d_fetch.log | awk '{print $2}')
python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv 5 16 $ALG "$1" > $O/traffic_3d.json 2> $O/traffic.err; echo "traffic rc=$?" | tee -a $O/summary.txt
cp $O/traffic_3d.json profiles/r04/traffic_3d.json
rm -rf $O/pmc_fetch $O/pmc_write   # raw counter files: tens of MB each, gpurun merges at most 64 MiB back
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -- python3 bench.py --steps 3 --warmup 2 --no-extras > $O/pmc_sq_bench.json 2> $O/pmc_sq.err; echo "pmc sq rc=$?" | tee -a $O/summary.txt
python tools/pmc_sq.py $O/pmc_sq fp16 > $O/pmc_sq_step.json 2>> $O/pmc_sq.err; echo "pmc_sq reduce rc=$?" | tee -a $O/summary.txt
cp $O/pmc_sq_step.json profiles/r04/pmc_sq_step.json
rm -rf $O/pmc_sq
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
MM_BENCH_LAYERS=1 python bench.py --steps 10 --warmup 3 > /dev/null 2> $O/layers.err; grep "\[layer\]" $O/layers.err > $O/engines_per_layer_final.txt
rocprofv3 --kernel-trace --stats -d $O/kstats -o ks --output-format csv -- python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_profiled.json 2> $O/bench_n1_profiled.err; echo "rocprof rc=$?" | tee -a $O/summary.txt
cp $O/kstats/ks_kernel_stats.csv $O/bench_n1_kernel_stats.csv 2>/dev/null
rm -rf $O/kstats
# the same with the sparse backward on ONE stream: per-kernel durations that the roofline leg's per-call event times must agree with
MM_SPCONV_BWD_OVERLAP=0 rocprofv3 --kernel-trace --stats -d $O/kstats_serial -o ks --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_n1_serial_profiled.json 2> $O/bench_n1_serial_profiled.err; echo "rocprof serial rc=$?" | tee -a $O/summary.txt
cp $O/kstats_serial/ks_kernel_stats.csv $O/bench_n1_serial_kernel_stats.csv 2>/dev/null
rm -rf $O/kstats_serial
python bench.py --workload c4 --steps 10 --warmup 3 > $O/bench_c4.json 2>/dev/null; python bench.py --workload c5 --steps 10 --warmup 3 > $O/bench_c5.json 2>/dev/null
python bench.py --precision bf16 --steps 20 --warmup 5 --no-extras > $O/bench_n1_bf16.json 2>/dev/null
python bench.py --image 400x225 --steps 15 --warmup 4 --no-extras > $O/bench_n1_image_400x225.json 2>/dev/null
MM_BN2D_FUSED=1 MM_BN_FUSED=1 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_n1_bn_as_under_ddp.json 2>/dev/null
for v in 0 1; do MM_BN2D_PRE=$v python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_bn2d_pre_$v.json 2>/dev/null; done
python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_bn2d_pre_auto.json 2>/dev/null
MM_CONV_WHOLE_ITEMS=1 python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_conv_whole_items.json 2>/dev/null
python tools/bench_bn2d.py > $O/bn2d_layer_set.txt 2>&1
[ -x tools/_bin/conv3x3_diag ] && timeout -k 10 200 tools/_bin/conv3x3_diag > $O/conv3x3_diag.txt 2>&1
[ -x tools/_bin/mfma_ceiling ] && timeout -k 10 200 tools/_bin/mfma_ceiling > $O/mfma_ceiling.txt 2>&1
du -sh $O; ls -la $O | tail -40
