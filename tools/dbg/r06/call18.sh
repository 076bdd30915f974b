# round 6, GPU call 18: which lane's block scale the scaled MFMA applies to which operand bytes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 300 python tools/dbg/mx_probe2.py 2>&1 | grep -v amdgpu.ids | tail -20 | tee $O/c18_mx_probe2.txt
