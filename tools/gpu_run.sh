#!/bin/bash
# ONE parameterised lease script (replaces the 27 one-off tools/gpu_r04*.sh of round 4).  Run through gpurun from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <tag> <step> [<step> ...]'
# Everything a step writes goes to gpurun_out/<tag>/ (scratch; copy what should be judged into profiles/).  Steps:
#   smoke      __graft_entry__.smoke()
#   tests      pytest -m gpu (whole suite)                 tests:<expr>  pytest -m gpu -k <expr>
#   bench      the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5 (stdout -> bench.json, stderr -> bench.err)
#   bench2     the same once more (-> bench_run2.json)
#   profile    tools/profile_round.sh <tag> (kernel trace + FETCH / WRITE / SQ counter passes; -> gpurun_out/profile_<tag>/)
#   exp:<args> python tools/archive/exp_r05.py <args with ',' for spaces>   (-> exp_<args>.jsonl)
#   sh:<file>  bash <file> (an ad-hoc fragment under tools/scratch/: git-ignored, but it travels to the box — gpurun_out/ does not)
TAG=${1:?tag}; shift
O=gpurun_out/$TAG; mkdir -p "$O"
export TMPDIR=/tmp
for step in "$@"; do
  case "$step" in
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?";;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log;;
    tests:*) timeout 1500 python -m pytest tests -m gpu -x -q -k "${step#tests:}" > $O/pytest_k.log 2>&1; echo "pytest -k rc=$?"; tail -8 $O/pytest_k.log;;
    bench|bench2)
      f=$O/bench; [ "$step" = bench2 ] && f=$O/bench_run2
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $f.json 2> $f.err; echo "$step rc=$? final line $(tail -1 $f.json | wc -c) bytes"
      cp bench_details.json ${f}_details.json 2>/dev/null
      tail -1 $f.json | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('value',r['value'],'frac',r['roofline']['frac'],'host_path',r.get('host_path'))
for k in r['roofline']['kernels']: print('  %-48s %-70s %8.2f us %.4f' % (k['kernel'][:48],k['config'][:70],k['us'],k['frac']), {x:k[x] for x in ('api_over_kernels','pair_us','pair_frac','deferred_check_us','bit_exact') if x in k})
print('cpu_baseline',{x:r['cpu_baseline'][x] for x in ('value','cores','kind')} if 'cpu_baseline' in r else None)
";;
    profile) bash tools/profile_round.sh $TAG > $O/profile.log 2>&1; echo "profile rc=$?";;
    exp:*) a="${step#exp:}"; timeout 900 python tools/archive/exp_r05.py ${a//,/ } >> "$O/exp_${a%%,*}.jsonl" 2>> $O/exp.err; echo "exp $a rc=$?"; tail -n 40 "$O/exp_${a%%,*}.jsonl";;
    sh:*) bash "${step#sh:}"; echo "sh rc=$?";;
    *) echo "unknown step $step";;
  esac
done
