# round 5, GPU call 10: why is the five-launch expert layer not faster on the wall clock? graph replay A/B + main-stream timeline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for g in 1 0; do for v in 0 1; do echo "== COVER_PI0_GRAPH=$g COVER_DEFER_NORM=$v"; COVER_PI0_GRAPH=$g COVER_DEFER_NORM=$v timeout 600 python bench.py --profile pi0 --steps 20 --warmup 4 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done; done
for v in 0 1; do
  COVER_DEFER_NORM=$v rocprofv3 --kernel-trace -d gpurun_out/r05/pi0_dn$v -o pi0 -- python3 bench.py --profile pi0 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python tools/timeline.py gpurun_out/r05/pi0_dn$v/pi0_results.db > gpurun_out/r05/call10_pi0_dn${v}_timeline.txt 2>&1
  head -60 gpurun_out/r05/call10_pi0_dn${v}_timeline.txt
  rm -rf gpurun_out/r05/pi0_dn$v
done
