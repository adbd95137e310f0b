cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PIPE=0 timeout 120 tools/xcd_chain_bench.bin > gpurun_out/r03_xcd_probe6.log 2>&1; echo "probe rc=$?" >> gpurun_out/r03_xcd_probe6.log
PIPE=0 VARIANT=48 timeout 120 tools/xcd_chain_bench.bin > gpurun_out/r03_xcd_probe6b.log 2>&1; echo "probe rc=$?" >> gpurun_out/r03_xcd_probe6b.log
PIPE=0 VARIANT=48 timeout 120 tools/xcd_chain_bench_fasttanh.bin > gpurun_out/r03_xcd_probe6c.log 2>&1; echo "probe rc=$?" >> gpurun_out/r03_xcd_probe6c.log
for f in gpurun_out/r03_xcd_probe6.log gpurun_out/r03_xcd_probe6b.log gpurun_out/r03_xcd_probe6c.log; do echo == $f; grep "^\[4\] B=45\|^\[4\] B=100.*xcd-local, 1\|^\[4\] B=100.*variant\|rc=\|phase ticks" $f; done
