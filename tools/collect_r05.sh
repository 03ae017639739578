#!/bin/bash
# Round-5 measurement records (run on the GPU box from the repo root; $1 = git commit of the tree, the box has no .git):
# smoke, PMC traffic of the sparse engines, SQ counters of the 2D kernels (both BEFORE the bench: bench.py quotes roofline.traffic /
# mfma_busy only from records whose source fingerprints match this tree), headline bench, kernel stats (default and serial) with
# their family break-down, the other workloads, A/B records of this round's switches, the co-residency stress (product build and
# the diagnostic round-4 resource request), in-kernel clocks, host profile.  Outputs under gpurun_out/r05/final/ (copied to
# profiles/r05/ afterwards).  The full GPU test suite is a separate call.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MM_GIT_HEAD="$1"
O=gpurun_out/r05/final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 tools/prof3d.py --bench-batch --steps 3 --warmup 1 > $O/prof3d_fetch.log 2>&1; echo "pmc fetch rc=$?" | tee -a $O/summary.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 tools/prof3d.py --bench-batch --steps 3 --warmup 1 > $O/prof3d_write.log 2>&1; echo "pmc write rc=$?" | tee -a $O/summary.txt
ALG=$(grep algorithmic_bytes_per_step $O/prof3d_fetch.log | awk '{print $2}')
python tools/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv 5 16 $ALG "$1" > $O/traffic_3d.json 2> $O/traffic.err; echo "traffic rc=$?" | tee -a $O/summary.txt
mkdir -p profiles/r05; cp $O/traffic_3d.json profiles/r05/traffic_3d.json
rm -rf $O/pmc_fetch $O/pmc_write   # raw counter files: tens of MB each, gpurun merges at most 64 MiB back
MM_GRAPH2D=0 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq -- python3 bench.py --steps 3 --warmup 2 --no-extras > $O/pmc_sq_bench.json 2> $O/pmc_sq.err; echo "pmc sq rc=$?" | tee -a $O/summary.txt
python tools/pmc_sq.py $O/pmc_sq fp16 > $O/pmc_sq_step.json 2>> $O/pmc_sq.err; echo "pmc_sq reduce rc=$?" | tee -a $O/summary.txt
cp $O/pmc_sq_step.json profiles/r05/pmc_sq_step.json
rm -rf $O/pmc_sq
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_driver_cmd.json 2> $O/bench_n1_driver_cmd.err; echo "bench (driver command) rc=$?" | tee -a $O/summary.txt
export MM_BENCH_NO_CPU=1
rocprofv3 --kernel-trace --stats -d $O/kstats -o ks --output-format csv -- python3 bench.py --steps 20 --warmup 5 > $O/bench_n1_profiled.json 2> $O/bench_n1_profiled.err; echo "rocprof rc=$?" | tee -a $O/summary.txt
cp $O/kstats/ks_kernel_stats.csv $O/bench_n1_kernel_stats.csv 2>/dev/null; rm -rf $O/kstats
# the same with everything on ONE stream (no dW side stream, no rulebook side stream): per-kernel durations that the roofline leg's
# per-call event times must agree with
MM_SPCONV_BWD_OVERLAP=0 MM_META_SIDE=0 rocprofv3 --kernel-trace --stats -d $O/kstats_serial -o ks --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_n1_serial_profiled.json 2> $O/bench_n1_serial_profiled.err; echo "rocprof serial rc=$?" | tee -a $O/summary.txt
cp $O/kstats_serial/ks_kernel_stats.csv $O/bench_n1_serial_kernel_stats.csv 2>/dev/null; rm -rf $O/kstats_serial
python tools/kstats_categories.py $O/bench_n1_serial_kernel_stats.csv 16 --top 40 > $O/kernel_families_serial.txt 2>&1
python bench.py --workload c4 --steps 10 --warmup 3 > $O/bench_c4.json 2>/dev/null; python bench.py --workload c5 --steps 10 --warmup 3 > $O/bench_c5.json 2>/dev/null
python bench.py --precision bf16 --steps 20 --warmup 5 --no-extras > $O/bench_n1_bf16.json 2>/dev/null
MM_BN2D_FUSED=1 MM_BN_FUSED=1 MM_GRAPH2D=0 MM_META_SIDE=0 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_n1_as_under_ddp.json 2>/dev/null
# A/B records of the round's switches (same box, back to back)
for i in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_default_$i.json 2>/dev/null
  MM_GRAPH2D=0 python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_graph2d_off_$i.json 2>/dev/null
  MM_META_SIDE=0 python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_meta_side_off_$i.json 2>/dev/null
  MM_OVERLAP_BRANCHES=2 python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_overlap_branches_2_$i.json 2>/dev/null
  MM_CONV_WGRAD_BATCH=0 python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_wgrad_batch_off_$i.json 2>/dev/null
  MM_BN2D_POOL=1 python bench.py --steps 20 --warmup 5 --no-extras > $O/ab_bn2d_pool_on_$i.json 2>/dev/null
done
python - > $O/ab_summary.txt <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05/final/ab_*.json") + glob.glob("gpurun_out/r05/final/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); c = d["config"]
        print(f.split("/")[-1].ljust(38), "value", d["value"], d["unit"], "| ms/step", d["ms_per_step"], "| p10/p50/p90", c.get("step_ms_p10_p50_p90"),
              "| host enqueue (empty queue)", c.get("host_enqueue_ms_empty_queue"), "| final loss", c.get("final_loss"), "| roofline.frac", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
# co-residency stress: the product build, then the diagnostic build with the round-4 resource request of k_conv3x3w
[ -f tools/_bin/libsquat.so ] || hipcc -O2 --offload-arch=gfx950 -shared -fPIC tests/helpers/squatter.hip -o tools/_bin/libsquat.so
timeout -k 10 300 python tools/corun_units.py > $O/corun_units_product.txt 2>&1
[ -f tools/_bin/libmm2d3d_hip_sharedcu.so ] && MM_LIB_PATH=tools/_bin/libmm2d3d_hip_sharedcu.so timeout -k 10 300 python tools/corun_units.py > $O/corun_units_round4_request.txt 2>&1
timeout -k 10 300 python tools/corun_net.py 1 1024 1024 20000 400 > $O/corun_net_product.txt 2>&1
[ -f tools/_bin/libmm2d3d_hip_sharedcu.so ] && MM_LIB_PATH=tools/_bin/libmm2d3d_hip_sharedcu.so timeout -k 10 300 python tools/conv_corun.py 1 > $O/conv_corun_round4_request.txt 2>&1
[ -f tools/_bin/libmm2d3d_hip_clock.so ] && MM_LIB_PATH=tools/_bin/libmm2d3d_hip_clock.so MM_GRAPH2D=0 timeout -k 10 300 python tools/clock_probe.py > $O/clock_probe.txt 2>&1
timeout -k 10 300 python tools/hosttime2.py 2>&1 | grep -v amdgpu.ids > $O/host_profile.txt
python tools/bench_bn2d.py > $O/bn2d_layer_set.txt 2>&1
du -sh $O; cat $O/summary.txt; cat $O/ab_summary.txt
