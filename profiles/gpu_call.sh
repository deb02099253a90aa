#!/bin/bash
# ONE script for what a gpurun call of this repository does (replaces the per-call one-offs of round 4).  Every stage writes
# under gpurun_out/<tag>_*; copy what is to be judged into profiles/ afterwards.
#
#   gpurun --timeout 2400 -- 'bash profiles/gpu_call.sh <tag> <stage> [<stage> ...]'
#
# stages (run in the order given):
#   tests[:<pytest args>]   pytest -m gpu (default: the whole suite), tail into <tag>_pytest.txt
#   bench[:<bench args>]    python bench.py <args> -> <tag>_bench.json (+ a one-line summary)
#   driver                  the driver's own command: bench.py --gpus 1 --steps 20 --warmup 5 -> <tag>_bench_driver_cmd.json
#   prof[:<bench args>]     rocprofv3 --kernel-trace --stats of bench.py (default: headline leg only) -> <tag>_prof/
#   ab:<lib.so>[:<rounds>]  profiles/ab_lib.py: headline leg, in-tree library against <lib.so>, alternating on this box
#   packed[:<S list>[:<U>]] seed-pack leg only (bench.run_seed_pack_leg) for S in the list at U updates per step
#   packed_prof:<S>[:<U>]   rocprofv3 kernel table of the packed iteration at S seeds -> <tag>_packed/S<S>.txt
#   gaps:<S>[:<U>]          rocprofv3 kernel trace of the packed iteration: per kernel duration and gap to the next launch
#   sweep                   bench.py --sweep (env kernels at N = 2^12 .. 2^24) -> <tag>_sweep.json
#   py:<script and args>    any profiles/*.py probe -> <tag>_<script>.txt
set +e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
TAG=$1; shift
summary() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline", {})
    print("BENCH %.4f ms/iter  %.2f M env-steps/s  step_push %.2f us" % (d["ms_per_step"], d["value"] / 1e6, r.get("launch_us", 0)))
    for k in ("seed_pack", "seed_pack_utd_1_256"):
        if isinstance(d.get(k), list):
            print(k, [(x.get("seeds_per_gpu"), round(x.get("ms_per_packed_iteration", 0), 4), round(x.get("value", 0) / 1e6, 2),
                       x.get("grad_steps_per_s")) for x in d[k]])
    if isinstance(d.get("config4"), dict):
        print("config4", {k: (round(v.get("ms_per_iteration", 0), 2), round(v.get("value", 0))) for k, v in d["config4"].items()
                          if isinstance(v, dict)})
    rs = d.get("roofline_stages", {})
    if "by_group" in rs:
        print("stages", rs.get("launches"), rs.get("stand_alone_sum_us"),
              [(g["group"], g["launches"], round(g["us"], 1), round(g["frac"], 4)) for g in rs["by_group"]])
except Exception as e:
    print("bench parse failed:", e)
P
}
for stage in "$@"; do
  kind=${stage%%:*}; arg=""; [ "$kind" != "$stage" ] && arg=${stage#*:}
  echo "=== $stage"
  case $kind in
    tests)
      (timeout 2400 python -m pytest ${arg:-tests} -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -30) > gpurun_out/${TAG}_pytest.txt 2>&1
      tail -12 gpurun_out/${TAG}_pytest.txt | cut -c1-300 ;;
    bench)
      timeout 1500 python bench.py $arg > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
      summary gpurun_out/${TAG}_bench.json; tail -c 400 gpurun_out/${TAG}_bench.err ;;
    driver)
      timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> gpurun_out/${TAG}_bench_driver_cmd.err
      summary gpurun_out/${TAG}_bench_driver_cmd.json ;;
    prof)
      bash profiles/bench_profile.sh ${arg:---no_legs --no_planner --min_seconds 1} 2>&1 | tail -32
      rm -rf gpurun_out/${TAG}_prof; mv gpurun_out/bench_prof gpurun_out/${TAG}_prof ;;
    ab)
      lib=${arg%%:*}; rounds=3; [ "$lib" != "$arg" ] && rounds=${arg#*:}
      timeout 1200 python profiles/ab_lib.py $lib $rounds > gpurun_out/${TAG}_ab.json 2> gpurun_out/${TAG}_ab.err
      cat gpurun_out/${TAG}_ab.json | cut -c1-600 ;;
    packed)
      seeds=${arg%%:*}; u=1; [ "$seeds" != "$arg" ] && u=${arg#*:}
      timeout 1500 python profiles/packed_probe.py $u ${seeds:-1,2,4,8} > gpurun_out/${TAG}_packed_u$u.json 2> gpurun_out/${TAG}_packed_u$u.err
      cut -c1-1200 gpurun_out/${TAG}_packed_u$u.json; tail -c 300 gpurun_out/${TAG}_packed_u$u.err ;;
    packed_prof)
      s=${arg%%:*}; u=16; [ "$s" != "$arg" ] && u=${arg#*:}
      bash profiles/packed_prof.sh $u "$s" ${TAG}_packed 2>&1 | tail -36 ;;
    gaps)
      # kernel-by-kernel durations and gaps of the packed iteration at S seeds (U updates): gaps:<S>[:<U>]
      s=${arg%%:*}; u=1; [ "$s" != "$arg" ] && u=${arg#*:}
      ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp$s && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp$s -o p -- python $OLDPWD/profiles/packed_probe.py $u $s 200 > /tmp/gp$s.json 2>/tmp/gp$s.err )
      python profiles/trace_gaps.py $(find /tmp/gp$s -name "*kernel_trace.csv" | head -1) step_push > gpurun_out/${TAG}_gaps_S${s}_U$u.txt 2>&1
      cat gpurun_out/${TAG}_gaps_S${s}_U$u.txt | cut -c1-160 ;;
    sweep)
      timeout 900 python bench.py --sweep > gpurun_out/${TAG}_sweep.json 2> gpurun_out/${TAG}_sweep.err
      python -c "
import json; d=json.loads(open('gpurun_out/${TAG}_sweep.json').read().strip().splitlines()[-1])
for k,v in d.items():
    if isinstance(v,list): print(k, [(x.get('n'), round(x.get('frac',0),4), round(x.get('us',0),1)) for x in v][-5:])" ;;
    py)
      name=$(basename ${arg%% *} .py)
      timeout 1500 python profiles/$arg > gpurun_out/${TAG}_$name.txt 2> gpurun_out/${TAG}_$name.err
      tail -25 gpurun_out/${TAG}_$name.txt | cut -c1-400; tail -c 300 gpurun_out/${TAG}_$name.err ;;
    *) echo "unknown stage $stage" ;;
  esac
done
