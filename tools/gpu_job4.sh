#!/bin/bash
# One GPU-box session of round 4 (run through gpurun from the repo root): tools/gpu_job4.sh <tag> <steps...>
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
for step in "$@"; do
  case $step in
    mb_node4) timeout 300 gaudi_amd/ngemm4_mb > $out/ngemm4_mb.txt 2>&1 ;;
    mb_node_gx) timeout 300 gaudi_amd/ngemm4_mb g > $out/ngemm_gx.txt 2>&1 ;;
    tests_new) timeout 1500 python3 -m pytest tests/test_gpu_round4.py -q -m gpu > $out/tests_new.txt 2>&1 ;;
    tests_all) timeout 3000 python3 -m pytest tests -x -q -m gpu > $out/tests_all.txt 2>&1 ;;
    tests_core) timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_split.py tests/test_gpu_fullsize.py -x -q -m gpu > $out/tests_core.txt 2>&1 ;;
    cpu_scaling) timeout 900 python3 tools/cpu_threads_probe.py > $out/cpu_scaling.txt 2>&1 ;;
    smoke) timeout 300 python3 __graft_entry__.py smoke > $out/smoke.txt 2>&1 ;;
    bench) timeout 1200 python3 bench.py --steps 2 --warmup 1 > $out/bench.json 2> $out/bench.err ;;
    bench_quick) timeout 600 $B > $out/bench_quick.json 2> $out/bench_quick.err ;;
    bench_c4x) GAUDI_DEBUG_PLAN=1 timeout 900 python3 bench.py --workload c4x --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $out/bench_c4x.json 2> $out/bench_c4x.err ;;
    bench_c4x_v4g) GAUDI_GN8=0 timeout 900 python3 bench.py --workload c4x --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $out/bench_c4x_v4g.json 2> $out/bench_c4x_v4g.err ;;
    bench_c2) timeout 600 $B --workload c2 > $out/bench_c2.json 2> $out/bench_c2.err ;;
    bench_c4) GAUDI_DEBUG_PLAN=1 timeout 600 $B --workload c4 > $out/bench_c4.json 2> $out/bench_c4.err ;;
    bench_b1024) GAUDI_DEBUG_PLAN=1 timeout 600 $B --batch 1024 > $out/bench_b1024.json 2> $out/bench_b1024.err ;;
    bench_b1024_nopairs) GAUDI_PAIRS=0 timeout 600 $B --batch 1024 > $out/bench_b1024_nopairs.json 2> $out/bench_b1024_nopairs.err ;;
    bench_c4_nopairs) GAUDI_PAIRS=0 timeout 600 $B --workload c4 > $out/bench_c4_nopairs.json 2> $out/bench_c4_nopairs.err ;;
    bench_c2_b1024) timeout 600 $B --workload c2 --batch 1024 > $out/bench_c2_b1024.json 2> $out/bench_c2_b1024.err ;;
    bench_c2_b1024_nopairs) GAUDI_PAIRS=0 timeout 600 $B --workload c2 --batch 1024 > $out/bench_c2_b1024_nopairs.json 2> $out/bench_c2_b1024_nopairs.err ;;
    bench_exp) for v in $GAUDI_VARIANTS; do [ -f gaudi_amd/libgaudi_var_$v.so ] && GAUDI_LIB=$PWD/gaudi_amd/libgaudi_var_$v.so timeout 600 $B > $out/bench_$v.json 2> $out/bench_$v.err; done ;;
    bench_dist) timeout 600 $B --dist > $out/bench_dist.json 2> $out/bench_dist.err ;;
    bench_gloo2) GAUDI_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
                   bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --batch 256 > $out/bench_gloo2.json 2> $out/bench_gloo2.err ;;
    *) echo "unknown step $step" ;;
  esac
done
ls -la $out
