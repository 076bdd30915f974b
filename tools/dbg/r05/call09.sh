# round 5, GPU call 9: kernel trace of the P1 decision with the deferred norm off / on; rest of the -m gpu suite
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in 0 1; do
  COVER_DEFER_NORM=$v rocprofv3 --kernel-trace --stats -d gpurun_out/r05/pi0_dn$v -o pi0 -- python3 bench.py --profile pi0 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
  python tools/rocpd_stats.py gpurun_out/r05/pi0_dn$v/pi0_results.db patchify_k > gpurun_out/r05/call09_pi0_dn${v}_kernel_stats.txt 2>&1
  head -40 gpurun_out/r05/call09_pi0_dn${v}_kernel_stats.txt
  rm -rf gpurun_out/r05/pi0_dn$v
done
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r05/call09_gputests.txt
