#!/bin/bash
# One GPU-box session of round 6 (run through gpurun from the repo root): tools/gpu_job6.sh <tag> <steps...>
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
ab() {  # ab <name> <bench args...>: variant libraries named in GAUDI_VARIANTS, then the tree's own library, twice each (A B A B)
  name=$1; shift
  for rep in 1 2; do
    for v in $GAUDI_VARIANTS; do
      [ -f gaudi_amd/libgaudi_var_$v.so ] && GAUDI_LIB=$PWD/gaudi_amd/libgaudi_var_$v.so timeout 900 $B "$@" > $out/${name}_${v}_$rep.json 2> $out/${name}_${v}_$rep.err
    done
    timeout 900 $B "$@" > $out/${name}_new_$rep.json 2> $out/${name}_new_$rep.err
  done
}
for step in "$@"; do
  case $step in
    ab_c3) ab c3 ;;
    ab_c3_nogate) ab c3 --no-parity-gate ;;
    ab_c2) ab c2 --workload c2 ;;
    ab_c4) ab c4 --workload c4 ;;
    ab_c4x) ab c4x --workload c4x --diffusion-steps 200 ;;
    ab_b1024) ab c3b1024 --batch 1024 --diffusion-steps 250 ;;
    mb_r) timeout 300 gaudi_amd/ngemmr_mb > $out/ngemmr_mb.txt 2>&1
          for f in gaudi_amd/ngemmr_mb_*; do timeout 60 $f r > $out/$(basename $f).txt 2>&1; done ;;
    mb_rv) for f in gaudi_amd/ngemmr_mb_*; do timeout 60 $f r > $out/$(basename $f).txt 2>&1; done ;;
    mb_r1) timeout 300 gaudi_amd/ngemmr_mb > $out/ngemmr_mb.txt 2>&1 ;;
    tests_r6) timeout 1800 python3 -m pytest tests/test_gpu_round6.py -q -m gpu > $out/tests_r6.txt 2>&1 ;;
    c4x) timeout 900 $B --workload c4x > $out/c4x_new.json 2> $out/c4x_new.err
         GAUDI_FAMILY_SPLIT=0 timeout 900 $B --workload c4x > $out/c4x_nosplit.json 2> $out/c4x_nosplit.err
         GAUDI_FAMILY_SPLIT=0 GAUDI_GN8_PACK=0 timeout 900 $B --workload c4x > $out/c4x_r5.json 2> $out/c4x_r5.err ;;
    c4x_pq) for rep in 1 2; do
           GAUDI_GN8_PQ=0 timeout 900 $B --workload c4x > $out/c4x_allglobal_$rep.json 2> $out/c4x_allglobal_$rep.err
           timeout 900 $B --workload c4x > $out/c4x_hybrid_$rep.json 2> $out/c4x_hybrid_$rep.err
         done ;;
    xcd) for rep in 1 2; do
           for wl in c4x c4; do
             GAUDI_XCD_ORDER=0 timeout 900 $B --workload $wl > $out/${wl}_plain_$rep.json 2> $out/${wl}_plain_$rep.err
             GAUDI_XCD_ORDER=1 timeout 900 $B --workload $wl > $out/${wl}_xcd_$rep.json 2> $out/${wl}_xcd_$rep.err
           done
         done ;;
    wide_full) for rep in 1 2; do
           GAUDI_WIDE_FULL=0 GAUDI_DEBUG_PLAN=1 timeout 900 $B --batch 1024 --diffusion-steps 250 > $out/b1024_half_$rep.json 2> $out/b1024_half_$rep.err
           GAUDI_DEBUG_PLAN=1 timeout 900 $B --batch 1024 --diffusion-steps 250 > $out/b1024_full_$rep.json 2> $out/b1024_full_$rep.err
         done
         grep -h "\[plan\]" $out/b1024_half_1.err | tail -1; grep -h "\[plan\]" $out/b1024_full_1.err | tail -1 ;;
    tests_core) timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_split.py tests/test_gpu_round5.py -x -q -m gpu > $out/tests_core.txt 2>&1 ;;
    tests_v8g) timeout 2400 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -m gpu -k "v8g or large or n40 or dense or callback_targets" > $out/tests_v8g.txt 2>&1 ;;
    tests_all) timeout 3400 python3 -m pytest tests -x -q -m gpu > $out/tests_all.txt 2>&1 ;;
    bench) timeout 1200 python3 bench.py > $out/bench.json 2> $out/bench.err ;;
    *) echo "unknown step $step" ;;
  esac
done
ls -la $out
for f in $out/*.json; do [ -f "$f" ] && echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)"; done
tail -3 $out/tests_*.txt 2>/dev/null
cat $out/ngemmr_mb*.txt 2>/dev/null | grep -v "^numerics.*OK$" | head -150
