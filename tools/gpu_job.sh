#!/bin/bash
# One GPU-box session of round 3 (run through gpurun from the repo root): tools/gpu_job.sh <tag> <steps...>
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
for step in "$@"; do
  case $step in
    mb_node) timeout 300 gaudi_amd/ngemm_mb > $out/ngemm_mb.txt 2>&1 ;;
    mb_split) for v in ${GAUDI_MB:-base}; do [ -x gaudi_amd/split_mb_$v ] && timeout 300 gaudi_amd/split_mb_$v t > $out/split_mb_$v.txt 2>&1; done ;;
    tests_new) timeout 1500 python3 -m pytest tests/test_gpu_round3.py -x -q -m gpu -s > $out/tests_new.txt 2>&1
               timeout 1500 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_stability.py -x -q -m gpu -k "nan or fresh or main_from_checkpoint or analyze or phi_vs_reference or predictor_forward_and_gradient or reproducible" >> $out/tests_new.txt 2>&1 ;;
    tests_all) timeout 3000 python3 -m pytest tests -x -q -m gpu > $out/tests_all.txt 2>&1 ;;
    bench_variants) for v in old t44 gs; do [ -f gaudi_amd/libgaudi_var_$v.so ] && GAUDI_LIB=$PWD/gaudi_amd/libgaudi_var_$v.so timeout 600 $B > $out/bench_$v.json 2> $out/bench_$v.err; done
                    timeout 600 $B > $out/bench_new.json 2> $out/bench_new.err ;;
    bench_exp) for v in $GAUDI_VARIANTS; do [ -f gaudi_amd/libgaudi_var_$v.so ] && GAUDI_LIB=$PWD/gaudi_amd/libgaudi_var_$v.so timeout 600 $B > $out/bench_$v.json 2> $out/bench_$v.err; done ;;
    tests_r3) timeout 2400 python3 -m pytest tests/test_gpu_round3.py -x -q -m gpu -s > $out/tests_r3.txt 2>&1
              timeout 600 python3 -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "nan or fresh" >> $out/tests_r3.txt 2>&1 ;;
    repro_gn) for w in tiny default; do for e in "GAUDI_WAVES=4" "GAUDI_FORCE_GN=1"; do echo "== $w $e" >> $out/repro_gn.txt; env $e timeout 300 python3 tools/repro_gn.py $w >> $out/repro_gn.txt 2>&1; done; done ;;
    stamps) GAUDI_LIB=$PWD/gaudi_amd/libgaudi_hip_stamps.so GAUDI_PRINT_STAMPS=1 timeout 900 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --diffusion-steps 100 > $out/stamps.json 2> $out/stamps.txt ;;
    spl) for k in 25 100 250; do timeout 600 $B --steps-per-launch $k > $out/bench_spl$k.json 2> $out/bench_spl$k.err; done ;;
    bench) timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/bench.err ;;
    bench_dist) timeout 600 $B --dist > $out/bench_dist.json 2> $out/bench_dist.err ;;
    *) echo "unknown step $step" ;;
  esac
done
ls -la $out
