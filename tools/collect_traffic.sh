#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE: separate runs, counters only) for the bench workloads -> profiles/<round>_pmc_traffic_<tag>.json
# Usage (on the GPU box): bash tools/collect_traffic.sh r03 [workload ...]   workloads: r50 r50_640 d0 d2 (default: r50 d0 d2)
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=${1:-r03}; shift
W=${@:-r50 d0 d2}
cd /tmp && export TMPDIR=/tmp
for w in $W; do
  case $w in
    r50)     A="";                                               TAG=retinanet_r50_512_b16_bf16 ;;
    r50_640) A="--size 640";                                     TAG=retinanet_r50_640_b16_bf16 ;;
    d0)      A="--arch efficientdet-d0";                         TAG=efficientdet_d0_512_b16_bf16 ;;
    d2)      A="--arch efficientdet-d2 --size 768 --batch 8";    TAG=efficientdet_d2_768_b8_bf16 ;;
    *) echo "unknown workload $w"; exit 2 ;;
  esac
  CMD="bench.py $A --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-infer --graph 0"
  rm -rf $R/gpurun_out/pmc_fetch_$w $R/gpurun_out/pmc_write_$w
  (cd $R && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$w -o run -- python $CMD > $R/gpurun_out/pmc_fetch_$w.log 2>&1) &&
  (cd $R && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$w -o run -- python $CMD > $R/gpurun_out/pmc_write_$w.log 2>&1) &&
  (cd $R && python tools/collect_traffic.py gpurun_out/pmc_fetch_$w gpurun_out/pmc_write_$w 5 gpurun_out/${RND}_pmc_traffic_$TAG.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python $CMD") || { echo "traffic pass for $w failed"; tail -5 $R/gpurun_out/pmc_fetch_$w.log $R/gpurun_out/pmc_write_$w.log; exit 1; }
  find $R/gpurun_out/pmc_fetch_$w $R/gpurun_out/pmc_write_$w -name "*.csv" -size +20M -delete
done
