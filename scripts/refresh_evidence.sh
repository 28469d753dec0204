#!/bin/bash
# How the files under profiles/r03_* are made (two gpurun calls, then a local copy). Usage, from the repo root:
#   gpurun --timeout 1200 -- 'TA_COMMIT=<short hash> bash scripts/refresh_evidence.sh gpu1'
#   bash scripts/refresh_evidence.sh collect1          # pmc_traffic.json first: bench.py reads its stamp
#   gpurun --timeout 1200 -- 'bash scripts/refresh_evidence.sh gpu2'
#   bash scripts/refresh_evidence.sh collect2
set -u
TAG=r03
case "${1:-}" in
gpu1)   # the GPU test suite, then rocprofv3 kernel trace + PMC passes of bench.py
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_final.log 2>&1; tail -3 gpurun_out/t_final.log
  bash scripts/profile_bench.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; tail -2 gpurun_out/prof_$TAG.log
  ;;
collect1)
  [ -f gpurun_out/prof_$TAG/pmc_traffic.json ] || { echo "no profile under gpurun_out/prof_$TAG"; exit 1; }
  cp gpurun_out/prof_$TAG/summary.txt profiles/${TAG}_rocprofv3_summary.txt
  cp "$(ls -t gpurun_out/prof_$TAG/trace/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_kernel_stats.csv
  cp gpurun_out/prof_$TAG/pmc_traffic.json profiles/pmc_traffic.json
  ;;
gpu2)   # the bench line, the configs, EAM / ADP / GRAP profiles, the MD-step traces, the calculator
  python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
  python scripts/bench_configs.py > gpurun_out/${TAG}_configs.json 2> gpurun_out/${TAG}_configs.err
  bash scripts/profile_configs.sh $TAG > gpurun_out/pc_$TAG.log 2>&1
  ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/md_trace_$TAG -- python3 $ROOT/scripts/md_loop.py 60 > $ROOT/gpurun_out/md_trace_$TAG.log 2>&1
  export TA_MD_SKIN=0
  rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/nl_trace_$TAG -- python3 $ROOT/scripts/md_loop.py 60 > $ROOT/gpurun_out/nl_trace_$TAG.log 2>&1
  unset TA_MD_SKIN; cd $ROOT
  python scripts/md_trace_summary.py "$(ls gpurun_out/md_trace_$TAG/*/*kernel_trace.csv | tail -1)" "filter_group" > gpurun_out/${TAG}_md_step_trace.txt 2>&1
  python scripts/md_trace_summary.py "$(ls gpurun_out/nl_trace_$TAG/*/*kernel_trace.csv | tail -1)" "bin_atoms" > gpurun_out/${TAG}_newlist_step_trace.txt 2>&1
  python scripts/bench_calculator.py --calls 300 > gpurun_out/${TAG}_calculator_call.json
  python scripts/bench_calculator.py --calls 300 --skin 0 > gpurun_out/${TAG}_calculator_call_skin0.json
  tail -c 200 gpurun_out/${TAG}_bench_n1.json
  ;;
collect2)
  for f in bench_n1.json configs.json md_step_trace.txt newlist_step_trace.txt calculator_call.json calculator_call_skin0.json; do
    [ -s gpurun_out/${TAG}_$f ] || { echo "missing gpurun_out/${TAG}_$f"; exit 1; }
  done
  for c in eam adp grap; do for n in 1 64; do
    [ -s gpurun_out/prof_${TAG}_${c}_${n}/summary.txt ] || { echo "missing profile $c $n"; exit 1; }
  done; done
  for f in bench_n1.json configs.json md_step_trace.txt newlist_step_trace.txt calculator_call.json calculator_call_skin0.json; do
    cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
  done
  for c in eam adp grap; do for n in 1 64; do
    ( echo "# scripts/run_config.py $c $n (plain run, then rocprofv3 kernel trace + PMC passes; scripts/profile_configs.sh)"
      cat gpurun_out/prof_${TAG}_${c}_${n}/plain.json; echo; cat gpurun_out/prof_${TAG}_${c}_${n}/summary.txt ) > profiles/${TAG}_${c}_${n}.txt
  done; done
  ;;
*) sed -n 2,7p "$0" ;;
esac
