# round 6, GPU call 17: probe of the MX GEMM's operand / scale assumptions
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
M=256 N=512 K=512 timeout 300 python tools/dbg/mx_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/c17_mx_probe.txt
M=512 N=4096 K=11008 timeout 300 python tools/dbg/mx_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a $O/c17_mx_probe.txt
