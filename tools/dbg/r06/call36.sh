# round 6, GPU call 36: value-column split of the fused decode attention (COVER_DA_VSPLIT=1 / 2 / 4; 4 is the default at N = 32) on the headline
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for vs in 4 2 1; do
  echo "== COVER_DA_VSPLIT=$vs (rep $rep)"; COVER_DA_VSPLIT=$vs timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c36_vsplit_ab.txt
